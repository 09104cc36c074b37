#!/bin/bash
# attn_h3 with one piece removed at a time (-DRVCX_ATT_ABL bits: 1 no MFMAs, 2 no fetch of the next key tile, 4 one commit only,
# 8 no exp, 16 no first barrier; built by tools/build_variant.sh abl<N> attention.hip -DRVCX_ATT_ABL=<N>): kernel durations
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for v in "" abl1 abl2 abl4 abl8 abl16 abl31; do
  export RVCX_LIBRARY=$PWD/polgen-rvc_amd/librvcx${v:+_$v}.so
  OUT=/tmp/abl_att_$$_$v
  rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 tools/bench_attention.py > /dev/null 2>&1
  echo "== ${v:-shipping}"; python3 tools/kernel_stats.py $OUT | grep "attn_h3" | cut -c1-40,110-175
done
