cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "piece_major" 2>&1 | grep -v Warning | tail -15 > gpurun_out/r4/g8_pytest.txt
timeout 900 python tools/bench_shapes.py --tiles table > gpurun_out/r4/g8_shapes_nat.txt 2>&1
LR_BENCH_PM=1 timeout 900 python tools/bench_shapes.py --tiles table > gpurun_out/r4/g8_shapes_pm.txt 2>&1
timeout 900 python tools/bench_shapes.py --tiles table >> gpurun_out/r4/g8_shapes_nat.txt 2>&1
for nb in 512 1024 2048 4096; do LR_GN_APPLY_BLOCKS=$nb timeout 300 python tools/bench_gn_apply.py >> gpurun_out/r4/g8_gn_apply.txt 2>&1; done
echo done
