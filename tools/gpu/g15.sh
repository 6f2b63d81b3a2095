cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 1500 python bench.py > gpurun_out/r4/g15_bench_default.json 2> gpurun_out/r4/g15_bench_default.err
bash tools/kstats.sh r4final > gpurun_out/r4/g15_kstats.txt 2>&1
cp gpurun_out/kstats_r4final.csv gpurun_out/r4/
timeout 900 bash tools/pmc_util.sh gpurun_out/r4/g15_pmc_util.txt > /dev/null 2>&1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > gpurun_out/r4/g15_bench_driver.json 2> gpurun_out/r4/g15_bench_driver.err
echo done
