#!/bin/bash
# Round-6 evidence for every driver-timed line (run on the GPU box through gpurun): serial-mode kernel traces of the C2
# (split-fp16 and exact-fp32), C3 and C5 workloads, and the HBM-traffic PMC passes for C2 (C3: PMC_C3=1, very slow) -> gpurun_out/prof_r06/
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp RVCX_SERIAL=1
OUT=gpurun_out/prof_r06
mkdir -p $OUT
trace() {   # tag, bench args...
  local tag=$1; shift
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tag -- python3 bench.py --no-cpu-baseline --no-children --no-roofline "$@" \
    > $OUT/bench_$tag.json 2> $OUT/trace_$tag.err
  python3 tools/kernel_stats.py /tmp/tr_$tag $OUT/rocprof_r06_${tag}_kernel_stats.txt > /dev/null
  rm -rf /tmp/tr_$tag
}
trace c2 --steps 5 --warmup 1
RVCX_H3=0 RVCX_ATT_H3=0 trace c2_fp32 --steps 5 --warmup 1
trace c3 --workload c3 --steps 1 --warmup 1
trace c5 --workload c5 --steps 1 --warmup 1
pmc() {     # tag, bench args...
  local tag=$1; shift
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf_$tag -- python3 bench.py --no-cpu-baseline --no-children --no-roofline "$@" > /dev/null 2> $OUT/pmc_fetch_$tag.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw_$tag -- python3 bench.py --no-cpu-baseline --no-children --no-roofline "$@" > /dev/null 2> $OUT/pmc_write_$tag.err
  python3 tools/pmc_traffic.py /tmp/pf_$tag /tmp/pw_$tag $OUT/pmc_traffic_r06_$tag.json
  rm -rf /tmp/pf_$tag /tmp/pw_$tag
}
pmc c2 --steps 3 --warmup 1
[ -n "${PMC_C3:-}" ] && pmc c3 --workload c3 --steps 1 --warmup 1    # > 40 minutes under the counter passes: off by default
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F16 \
  --output-format csv -d /tmp/pm_c2 -- python3 bench.py --no-cpu-baseline --no-children --no-roofline --steps 3 --warmup 1 > /dev/null 2> $OUT/pmc_mfma.err
python3 tools/pmc_summary.py /tmp/pm_c2 > $OUT/pmc_mfma_r06.txt
ls -la $OUT
