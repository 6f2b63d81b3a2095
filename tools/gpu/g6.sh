cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_backward.py tests/test_gpu_nvs.py tests/test_gpu_ffn.py -q -k "attention or context_gradient or folded_into or c_input or proj_out or group_sums" 2>&1 | grep -v Warning | tail -40 > gpurun_out/r4/g6_pytest.txt
LR_BENCH_MODES=0 timeout 300 python tools/bench_attn_pp.py 5 > gpurun_out/r4/g6_attn_new.txt 2>&1
LR_BENCH_MODES=0 LEFTREFILL_LIB_PATH=$PWD/leftrefill_amd/lib/variants/libleftrefill_hip_foldmax.so timeout 300 python tools/bench_attn_pp.py 5 > gpurun_out/r4/g6_attn_foldmax.txt 2>&1
LR_BENCH_MODES=0 timeout 300 python tools/bench_attn_pp.py 5 >> gpurun_out/r4/g6_attn_new.txt 2>&1
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 --no-roofline --no-cpu-baseline > gpurun_out/r4/g6_bench_new.json 2> gpurun_out/r4/g6_bench_new.err
echo done
