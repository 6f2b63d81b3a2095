cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 2400 python -m pytest tests/test_gpu_backward.py tests/test_gpu_bf16.py tests/test_gpu_nvs.py tests/test_gpu_bench.py -q 2>&1 | grep -v Warning | tail -15 > gpurun_out/r4/g27_pytest.txt
echo done
