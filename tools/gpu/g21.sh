cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 1500 python bench.py > gpurun_out/r4/g21_bench_default.json 2> gpurun_out/r4/g21_bench_default.err
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > gpurun_out/r4/g21_bench_driver.json 2> gpurun_out/r4/g21_bench_driver.err
bash tools/kstats.sh r4final2 > gpurun_out/r4/g21_kstats.txt 2>&1
cp gpurun_out/kstats_r4final2.csv gpurun_out/r4/
timeout 900 bash tools/pmc_util.sh gpurun_out/r4/g21_pmc_util.txt > /dev/null 2>&1
timeout 900 python bench.py --workload mv5 --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > gpurun_out/r4/g21_mv5_fused.json 2> gpurun_out/r4/g21_mv5_fused.err
timeout 900 python bench.py --workload mv5 --mv-shard --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > gpurun_out/r4/g21_mv5_shard.json 2> gpurun_out/r4/g21_mv5_shard.err
timeout 900 python bench.py --workload train --dtype bf16 --steps 10 --warmup 3 > gpurun_out/r4/g21_train_bf16.json 2> gpurun_out/r4/g21_train_bf16.err
echo done
