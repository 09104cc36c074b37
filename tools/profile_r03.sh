#!/bin/bash
# Round-3 profile suite (run on the GPU box through gpurun): kernel trace + PMC passes of the bench command in
# SERIAL mode (every launch on the library's one stream: a kernel's duration is its own), summaries -> gpurun_out/.
# rocprofv3 gets the python interpreter directly after `--` (no env / sh -c hop: the profiler initialises the GPU).
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp RVCX_SERIAL=1
OUT=gpurun_out/prof_r03
mkdir -p $OUT
CMD="python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-children"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F16 \
  --output-format csv -d $OUT/pmc_mfma -- $CMD > /dev/null 2> $OUT/pmc_mfma.err
python3 tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_traffic_r03.json
python3 tools/pmc_summary.py $OUT/pmc_mfma > $OUT/pmc_mfma_r03.txt
python3 tools/kernel_stats.py $OUT/trace $OUT/rocprof_r03_serial_kernel_stats.txt
ls -la $OUT
