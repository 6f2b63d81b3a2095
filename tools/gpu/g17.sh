cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 900 python tools/bench_attn_tr.py 5 > gpurun_out/r4/g17_attn_tr.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "attention or out_block" 2>&1 | grep -v Warning | tail -15 > gpurun_out/r4/g17_pytest.txt
timeout 900 python -m pytest tests/test_gpu_sampler.py -q -x 2>&1 | grep -v Warning | tail -15 >> gpurun_out/r4/g17_pytest.txt
b() { timeout 600 python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > gpurun_out/r4/g17_bench_$1.json 2> gpurun_out/r4/g17_bench_$1.err; }
b new
LEFTREFILL_EMB_TABLE=0 b noembtable
LEFTREFILL_OUT_FUSED=0 b nooutfused
b new2
echo done
