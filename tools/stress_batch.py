#!/usr/bin/env python3
"""Stress of the pipelined batch path: many utterances of random lengths under two chunk geometries -- (1,1,2,3): most
clips are cut into chunks and only equal lengths share a micro-batch; (1,6,38,41): nothing is cut and the lengths fall
into ragged length classes -- both F0 back-ends, retrieval on / off: every utterance of the batched call must be
bit-identical to converting it alone.  usage: stress_batch.py [rounds=3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polgen_rvc_amd  # noqa
from polgen_rvc_amd import _lib, synthetic as S, weights as W

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ctx = _lib.Context(0)
hcfg, rcfg, scfg = S.HUBERT_CFG_TINY, S.RMVPE_CFG_TINY, S.SYNTH_CFG_TINY
ctx.load_hubert(W.hubert_cfg_struct(hcfg), S.hubert_state(hcfg, 3))
ctx.load_rmvpe(W.rmvpe_cfg_struct(rcfg), S.rmvpe_state(rcfg, 3))
sd = S.fcpe_state(S.FCPE_CFG_TINY, 401)
ctx.load_fcpe(W.fcpe_cfg_struct(W.fcpe_cfg_from_state(sd)), sd)
mid = ctx.load_synth(W.synth_cfg_struct(scfg, hcfg["embed_dim"]), S.synth_state(scfg, 3, input_dim=hcfg["embed_dim"]))
rng = np.random.default_rng(11)
bad = 0
for r in range(rounds):
    lens = list(rng.choice([1.3, 1.9, 2.6, 3.3, 5.1, 7.7], size=20)) + [float(rng.uniform(1.2, 9.0)) for _ in range(6)]
    rng.shuffle(lens)
    clips = [S.make_clip(100 * r + i, float(t)) for i, t in enumerate(lens)]
    for geo in ((1, 1, 2, 3), (1, 6, 38, 41)):
        for method in (_lib.F0_RMVPE, _lib.F0_FCPE):
            for index_rate in (0.0, 0.6):
                if index_rate:
                    ctx.load_index(S.make_index(1024, hcfg["embed_dim"], r))
                p = _lib.Params(1.0, 50.0, 1100.0, index_rate, 0.33, 0.7, 0, *geo, 40 + r)
                p.f0_method = method
                batch = ctx.convert_batch(mid, clips, p)
                mbs = ctx.last_micro_batches()
                for i, c in enumerate(clips):
                    q = _lib.Params(1.0, 50.0, 1100.0, index_rate, 0.33, 0.7, 0, *geo, 40 + r + i)
                    q.f0_method = method
                    alone = ctx.convert_batch(mid, [c], q)[0]
                    if not np.array_equal(alone, batch[i]):
                        bad += 1
                        print(f"MISMATCH round {r} geometry {geo} method {method} index {index_rate} clip {i} len {lens[i]:.2f}", flush=True)
                if index_rate:
                    ctx.load_index(None)
                print(f"  round {r} geometry {geo} method {method} index {index_rate}: micro-batches {sorted(mbs, reverse=True)}", flush=True)
    print(f"round {r}: {len(clips)} clips x 8 configurations checked, mismatches so far {bad}", flush=True)
print("STRESS_OK" if bad == 0 else f"STRESS_FAILED {bad}")
sys.exit(1 if bad else 0)
