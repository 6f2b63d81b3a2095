#!/bin/bash
# PMC counters of the level-0 self-attention (B 8, 5 heads, 8192 x 8192): pre-transposed V^T (lr_attention_vt_f16) vs natural V gathered
# with the LDS transpose read (LEFTREFILL_ATTN_VT=0 -> lr_attention_f16).  Two separate --pmc passes each, no trace domains.
#   bash tools/pmc_attn_tr.sh [outdir]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$R/gpurun_out/pmc_attn_tr}
mkdir -p $OUT
rm -f $OUT/summary.txt
for v in vt tr; do
  E="LEFTREFILL_ATTN_VT=1"
  [ $v = tr ] && E="LEFTREFILL_ATTN_VT=0"
  rm -rf /tmp/p1_$v /tmp/p2_$v
  env $E rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS \
      -d /tmp/p1_$v -- python $R/tools/bench_attn.py 8 5 8192 8192 3 > /dev/null 2>&1
  env $E rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES \
      -d /tmp/p2_$v -- python $R/tools/bench_attn.py 8 5 8192 8192 3 > /dev/null 2>&1
  echo "== $v" | tee -a $OUT/summary.txt
  python $R/tools/pmc_kernel.py attention_ /tmp/p1_$v /tmp/p2_$v | tee -a $OUT/summary.txt
done
