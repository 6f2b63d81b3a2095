cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_nvs.py tests/test_gpu_ops.py -q -x -k "gradients_flow or c_input_reaches or pingpong or logit_jumps" 2>&1 | grep -v Warning | tail -70 > gpurun_out/r4/g7_pytest.txt
echo done
