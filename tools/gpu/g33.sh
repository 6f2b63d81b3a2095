cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 900 python tools/bench_attn_nqb.py 5 > gpurun_out/r4/g33_attn_nqb.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k attention 2>&1 | grep -v Warning | tail -5 > gpurun_out/r4/g33_pytest.txt
b() { timeout 600 env $2 python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > gpurun_out/r4/g33_bench_$1.json 2> gpurun_out/r4/g33_bench_$1.err; }
b auto X=1
b nqb2 LR_ATTN_NQB=2
b auto2 X=1
b nqb2b LR_ATTN_NQB=2
echo done
