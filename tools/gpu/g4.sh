cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_ops.py -q -k "context_gradient or folded_into or pingpong or group_sums or statistics_from" 2>&1 | grep -v Warning | tail -80 > gpurun_out/r4/g4_pytest.txt
bash tools/kstats.sh new2 > gpurun_out/r4/g4_kstats_new.txt 2>&1
echo done
