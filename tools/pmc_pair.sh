#!/bin/bash
# PMC passes over the fused ResBlock step (tools/bench_pair.py shapes $1) for RVCX_PAIR_VARIANT=$2: LDS / wait / MFMA counters
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
[ -n "${2:-}" ] && export RVCX_PAIR_VARIANT=$2
OUT=/tmp/pmc_pair_$$
for set in "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_BUSY_CYCLES" \
           "SQ_WAVE_CYCLES SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES"; do
  rm -rf $OUT
  rocprofv3 --pmc $set --output-format csv -d $OUT -- python3 tools/bench_pair.py 1 1 $1 > /dev/null 2>&1
  python3 tools/pmc_summary.py $OUT | grep -A10 resblock_pair
done
