cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_nvs.py -q -x -k "mv_v5_concat or gradients_flow" 2>&1 | grep -v Warning | tail -60 > gpurun_out/r4/g13_pytest.txt
echo done
