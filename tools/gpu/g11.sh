cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py -q -k "sixteen_channel or time_mlp or splitk or full_8x16 or unet_mid" 2>&1 | grep -v Warning | tail -30 > gpurun_out/r4/g11_pytest.txt
timeout 600 python tools/bench_shapes.py --tiles all --only "l3" > gpurun_out/r4/g11_shapes_l3.txt 2>&1
bash tools/kstats.sh r4b > gpurun_out/r4/g11_kstats.txt 2>&1
echo done
