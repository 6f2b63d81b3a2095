cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 900 bash tools/pmc_attn_tr.sh gpurun_out/r4/pmc_attn_tr > gpurun_out/r4/g18_pmc_attn_tr.log 2>&1
bash tools/kstats.sh r4c > gpurun_out/r4/g18_kstats_vt.txt 2>&1
bash tools/kstats.sh r4c_tr LEFTREFILL_ATTN_VT=0 > gpurun_out/r4/g18_kstats_tr.txt 2>&1
bash tools/kstats.sh r4c_old LEFTREFILL_EMB_TABLE=0 LEFTREFILL_OUT_FUSED=0 > gpurun_out/r4/g18_kstats_old.txt 2>&1
echo done
