cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 900 python bench.py --workload mv5 --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > gpurun_out/r4/g29_mv5_fused.json 2> gpurun_out/r4/g29_mv5_fused.err
timeout 900 python bench.py --workload mv5 --mv-shard --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > gpurun_out/r4/g29_mv5_shard0.json 2> gpurun_out/r4/g29_mv5_shard0.err
LEFTREFILL_MV_SIM_RANK=2 timeout 900 python bench.py --workload mv5 --mv-shard --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > gpurun_out/r4/g29_mv5_shard2.json 2> gpurun_out/r4/g29_mv5_shard2.err
timeout 900 python bench.py --workload train --dtype bf16 --steps 10 --warmup 3 > gpurun_out/r4/g29_train_bf16.json 2> gpurun_out/r4/g29_train_bf16.err
echo done
