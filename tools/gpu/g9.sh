cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 1200 python -m pytest tests/test_gpu_unet.py tests/test_gpu_bench.py -q -k "multiview or mv_shard or mv" 2>&1 | grep -v Warning | tail -30 > gpurun_out/r4/g9_pytest.txt
timeout 900 python bench.py --workload mv5 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > gpurun_out/r4/g9_mv5_fused.json 2> gpurun_out/r4/g9_mv5_fused.err
timeout 900 python bench.py --workload mv5 --mv-shard --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r4/g9_mv5_shard_rank0.json 2> gpurun_out/r4/g9_mv5_shard_rank0.err
LEFTREFILL_MV_SIM_RANK=2 timeout 900 python bench.py --workload mv5 --mv-shard --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r4/g9_mv5_shard_rank2.json 2> gpurun_out/r4/g9_mv5_shard_rank2.err
timeout 900 python bench.py --workload train --steps 10 --warmup 3 --dtype bf16 > gpurun_out/r4/g9_train_bf16.json 2> gpurun_out/r4/g9_train_bf16.err
echo done
