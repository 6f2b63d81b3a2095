cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 900 python tools/parity_table.py gpurun_out/r4/parity_default.txt > /dev/null 2>&1
LEFTREFILL_MV_LN_FOLD=0 timeout 900 python tools/parity_table.py gpurun_out/r4/parity_nomvfold.txt > /dev/null 2>&1
LEFTREFILL_LIB_PATH=$PWD/leftrefill_amd/lib/variants/libleftrefill_hip_nofold.so timeout 900 python tools/parity_table.py gpurun_out/r4/parity_noattnfold.txt > /dev/null 2>&1
LEFTREFILL_GN_GROUPS=0 LEFTREFILL_ST_GN_FOLD=0 timeout 900 python tools/parity_table.py gpurun_out/r4/parity_nogn.txt > /dev/null 2>&1
echo done
