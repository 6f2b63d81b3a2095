cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 900 python tools/bench_attn_nqb.py 5 > gpurun_out/r4/g34_attn_nqb.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k attention 2>&1 | grep -v Warning | tail -5 > gpurun_out/r4/g34_pytest.txt
timeout 1200 python -m pytest tests/test_gpu_unet.py tests/test_gpu_backward.py tests/test_gpu_text.py -q 2>&1 | grep -v Warning | tail -5 >> gpurun_out/r4/g34_pytest.txt
echo done
