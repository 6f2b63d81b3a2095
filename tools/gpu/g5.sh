cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_ops.py -q -x -s -k "context_gradient or folded_into" 2>&1 | grep -v Warning | tail -120 > gpurun_out/r4/g5_pytest.txt
echo done
