#!/bin/bash
# PMC utilisation per kernel family over one eager UNet step (N = 8, 64x128): three separate --pmc passes (no trace domains),
# aggregated by tools/pmc_util.py.   bash tools/pmc_util.sh [out.txt]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$R/gpurun_out/pmc_util.txt}; case $OUT in /*) ;; *) OUT=$R/$OUT;; esac
rm -rf /tmp/pu1 /tmp/pu2 /tmp/pu3
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS \
    -d /tmp/pu1 -- python $R/tools/pmc_step.py run > /dev/null 2>&1
rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
    -d /tmp/pu2 -- python $R/tools/pmc_step.py run > /dev/null 2>&1
python $R/tools/pmc_util.py /tmp/pu1 /tmp/pu2 | tee $OUT
