cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -v Warning | tail -30 > gpurun_out/r4/g19_pytest.txt
timeout 900 python tools/parity_table.py > gpurun_out/r4/g19_parity.txt 2>&1
timeout 600 python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > gpurun_out/r4/g19_bench.json 2> gpurun_out/r4/g19_bench.err
echo done
