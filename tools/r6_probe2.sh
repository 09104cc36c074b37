#!/bin/bash
# round 6: gemm_bd against the LDS-staged tiles; HuBERT alone B = 1 / 16; API call counts of one C2 step
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r6d; mkdir -p $O
export RVCX_DEBUG=1
timeout 600 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_audio.py tests/test_gpu_round6.py -q -m gpu > $O/tests.log 2>&1
timeout 300 python tools/bench_gemm.py 20 > $O/bench_gemm.txt 2>&1
for bd in 1 0; do RVCX_GEMM_BD=$bd timeout 300 python tools/prof_hubert.py 16 > $O/prof_hubert_bd${bd}_B16.txt 2>&1; done
RVCX_GEMM_BD=0 timeout 300 python bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-600 > $O/c3_bd0.txt
timeout 300 python bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-600 > $O/c3_bd1.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rpa; timeout 300 rocprofv3 --hip-runtime-trace --kernel-trace --stats --output-format csv -d /tmp/rpa -- python3 $OLDPWD/bench.py --steps 6 --warmup 2 --no-children --no-cpu-baseline --no-roofline > /dev/null 2>&1
find /tmp/rpa -name "*stats*.csv" | head -5 > $OLDPWD/$O/rpa_files.txt
for f in $(find /tmp/rpa -name "*hip_api_stats.csv" -o -name "*hip_stats.csv" | head -2); do cp $f $OLDPWD/$O/; done
python3 $OLDPWD/tools/kernel_stats.py /tmp/rpa $OLDPWD/$O/rocprof_c2_kernels.txt > /dev/null
