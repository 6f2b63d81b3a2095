cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/r4/g3_pytest.txt
bash tools/kstats.sh new > gpurun_out/r4/g3_kstats_new.txt 2>&1
bash tools/kstats.sh nogn LEFTREFILL_GN_GROUPS=0 LEFTREFILL_ST_GN_FOLD=0 > gpurun_out/r4/g3_kstats_nogn.txt 2>&1
bash tools/kstats.sh nofoldst LEFTREFILL_ST_GN_FOLD=0 > gpurun_out/r4/g3_kstats_nostfold.txt 2>&1
cp gpurun_out/kstats_*.csv gpurun_out/r4/ 2>/dev/null
echo done
