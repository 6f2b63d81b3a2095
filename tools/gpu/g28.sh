cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 1500 python bench.py > gpurun_out/r4/g28_bench_default.json 2> gpurun_out/r4/g28_bench_default.err
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > gpurun_out/r4/g28_bench_driver.json 2> gpurun_out/r4/g28_bench_driver.err
bash tools/kstats.sh r4final3 > gpurun_out/r4/g28_kstats.txt 2>&1
cp gpurun_out/kstats_r4final3.csv gpurun_out/r4/
timeout 900 bash tools/pmc_util.sh gpurun_out/r4/g28_pmc_util.txt > /dev/null 2>&1
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4/g28_smoke.txt 2>&1
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | grep -v Warning | tail -40 > gpurun_out/r4/g28_pytest.txt
echo done
