cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
LEFTREFILL_LIB_PATH=$GRAFT_REPO_ROOT/leftrefill_amd/lib/variants/libleftrefill_hip_stagger.so timeout 1200 python tools/bench_stagger.py > gpurun_out/r4/g22_stagger.txt 2>&1
echo done
