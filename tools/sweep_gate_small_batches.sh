#!/bin/bash
# round 6: HuBERT's wait behind the F0 U-Net -- always (RVCX_HUBERT_GATE=1), never (0), the default rule (unset) -- on calls of
# 2 / 4 clips (one exposed micro-batch), C3 (four micro-batches of 16), C5, C2; one box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r6t; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps ${STEPS:-2} --warmup 1 --no-cpu-baseline --no-roofline --no-children $ARGS 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$ARGS $tag', round(d['value'],1))" >> $O/gate_rule.txt; }
for ARGS in "--batch 2" "--batch 4"; do STEPS=8; for i in 1 2; do run always RVCX_HUBERT_GATE=1; run never RVCX_HUBERT_GATE=0; run rule X=1; done; done
for ARGS in "--workload c3" "--workload c5"; do STEPS=2; for i in 1 2; do run always RVCX_HUBERT_GATE=1; run never RVCX_HUBERT_GATE=0; run rule X=1; done; done
ARGS=""; STEPS=10; run always RVCX_HUBERT_GATE=1; run rule X=1
cat $O/gate_rule.txt
