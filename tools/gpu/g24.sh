cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "fused_skip or conv_golden or tile_plan" 2>&1 | grep -v Warning | tail -30 > gpurun_out/r4/g24_pytest.txt
timeout 1500 python -m pytest tests/test_gpu_unet.py -q -x 2>&1 | grep -v Warning | tail -30 >> gpurun_out/r4/g24_pytest.txt
b() { timeout 600 env $2 python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > gpurun_out/r4/g24_bench_$1.json 2> gpurun_out/r4/g24_bench_$1.err; }
b fused X=1
b unfused LEFTREFILL_SKIP_FUSED=0
b fused2 X=1
b unfused2 LEFTREFILL_SKIP_FUSED=0
bash tools/kstats.sh r4e > gpurun_out/r4/g24_kstats_fused.txt 2>&1
bash tools/kstats.sh r4e0 LEFTREFILL_SKIP_FUSED=0 > gpurun_out/r4/g24_kstats_unfused.txt 2>&1
echo done
