cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x 2>&1 | grep -v Warning | tail -30 > gpurun_out/r4/g25_pytest.txt
timeout 1500 python -m pytest tests/test_gpu_unet.py -q -x 2>&1 | grep -v Warning | tail -30 >> gpurun_out/r4/g25_pytest.txt
b() { timeout 600 env $2 python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > gpurun_out/r4/g25_bench_$1.json 2> gpurun_out/r4/g25_bench_$1.err; }
b comp X=1
b nocomp LEFTREFILL_FF_PROJ=0
b comp2 X=1
b nocomp2 LEFTREFILL_FF_PROJ=0
timeout 900 python tools/parity_table.py > gpurun_out/r4/g25_parity.txt 2>&1
echo done
