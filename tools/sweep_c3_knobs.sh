#!/bin/bash
# round 6: HuBERT's wait behind the F0 U-Net (RVCX_HUBERT_GATE: 1 always = round 5, unset = single utterances only) on C3 / C5 / C2, one box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r6n; mkdir -p $O
run() { tag=$1; wl=$2; shift; shift; env "$@" timeout 300 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-children 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl $tag', round(d['value'],1), round(d['stage_ms']['total'],1))" >> $O/gate.txt; }
for i in 1 2; do
run gate_always c3 RVCX_HUBERT_GATE=1
run gate_single_only c3 X=1
done
for i in 1 2; do
run gate_always c5 RVCX_HUBERT_GATE=1
run gate_single_only c5 X=1
done
run gate_always c2 RVCX_HUBERT_GATE=1
run gate_single_only c2 X=1
cat $O/gate.txt
