cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests/test_gpu_unet.py tests/test_gpu_sampler.py -q 2>&1 | grep -v Warning | tail -5 > gpurun_out/r4/g31_pytest.txt
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "tile" 2>&1 | grep -v Warning | tail -3 >> gpurun_out/r4/g31_pytest.txt
timeout 600 python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > gpurun_out/r4/g31_bench.json 2> gpurun_out/r4/g31_bench.err
echo done
