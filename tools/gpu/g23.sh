cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests/test_gpu_harness.py -q -x 2>&1 | grep -v Warning | tail -40 > gpurun_out/r4/g23_pytest.txt
echo done
