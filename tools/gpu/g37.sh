cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k attention 2>&1 | grep -v Warning | tail -3 > gpurun_out/r4/g37_pytest.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 >> gpurun_out/r4/g37_pytest.txt
timeout 600 python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > gpurun_out/r4/g37_bench.json 2> gpurun_out/r4/g37_bench.err
echo done
