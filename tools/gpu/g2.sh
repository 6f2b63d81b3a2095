cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/r4/g2_pytest.txt
LR_BENCH_MODES=0 timeout 300 python tools/bench_attn_pp.py 5 > gpurun_out/r4/g2_attn_fold.txt 2>&1
LR_BENCH_MODES=0 LEFTREFILL_LIB_PATH=$PWD/leftrefill_amd/lib/variants/libleftrefill_hip_nofold.so timeout 300 python tools/bench_attn_pp.py 5 > gpurun_out/r4/g2_attn_nofold.txt 2>&1
LR_BENCH_MODES=0 timeout 300 python tools/bench_attn_pp.py 5 >> gpurun_out/r4/g2_attn_fold.txt 2>&1
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 --no-roofline --no-cpu-baseline > gpurun_out/r4/g2_bench_new.json 2> gpurun_out/r4/g2_bench_new.err
LEFTREFILL_GN_GROUPS=0 LEFTREFILL_ST_GN_FOLD=0 timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 --no-roofline --no-cpu-baseline > gpurun_out/r4/g2_bench_nogn.json 2> gpurun_out/r4/g2_bench_nogn.err
LEFTREFILL_LIB_PATH=$PWD/leftrefill_amd/lib/variants/libleftrefill_hip_nofold.so timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 --no-roofline --no-cpu-baseline > gpurun_out/r4/g2_bench_nofold.json 2> gpurun_out/r4/g2_bench_nofold.err
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 --no-roofline --no-cpu-baseline > gpurun_out/r4/g2_bench_new2.json 2> gpurun_out/r4/g2_bench_new2.err
echo done
