cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 900 python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > gpurun_out/r4/g12_bench.json 2> gpurun_out/r4/g12_bench.err
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -v Warning | tail -40 > gpurun_out/r4/g12_pytest.txt
echo done
