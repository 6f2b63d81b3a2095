#!/bin/bash
# PMC counters of the attention kernels (two separate --pmc passes, no trace domains), pipelined vs reference schedule.
#   bash tools/pmc_attn.sh [outdir]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$R/gpurun_out/pmc_attn}
mkdir -p $OUT
for v in pipe old; do
  E="LR_X=0"; [ $v = old ] && E="LR_ATTN_NO_PIPE=1"
  env $E rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS \
      -d /tmp/p1_$v -- python $R/tools/bench_attn.py 8 5 8192 8192 3 > /dev/null 2>&1
  env $E rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES \
      -d /tmp/p2_$v -- python $R/tools/bench_attn.py 8 5 8192 8192 3 > /dev/null 2>&1
  env $E rocprofv3 --output-format csv --pmc SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU2 SQ_THREAD_CYCLES_VALU SQ_LDS_DATA_FIFO_FULL \
      -d /tmp/p3_$v -- python $R/tools/bench_attn.py 8 5 8192 8192 3 > /dev/null 2>&1
  echo "== $v" | tee -a $OUT/summary.txt
  python $R/tools/pmc_kernel.py attention_ /tmp/p1_$v /tmp/p2_$v /tmp/p3_$v | tee -a $OUT/summary.txt
done
