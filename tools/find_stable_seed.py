"""Parallel search for a synthetic-RMVPE seed on which EVERY frame of a long clip has a well-conditioned f0 decision
(oracle.rmvpe.unstable_frames) and no coarse-quantisation tie -- the precondition tools/gen_golden.py puts on every
waveform fixture (an argmax flip shifts the sine source's phase for the rest of the chunk; the reference itself is not
reproducible across BLAS builds on such frames).  A 95 s clip has 9 700 frames, so ~1 % of the seeds qualify: this tool
spreads the scan over processes.  Usage: python tools/find_stable_seed.py --seconds 95 --clip 3 --start 3000 --n 160 --procs 4
Prints one line per seed and the qualifying seeds at the end; `gen_golden.py --full --only pipe_long95` takes the seed
through RVCX_LONG95_SEED."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def scan(args):
    seeds, seconds, clip, threads, rel, tie_margin = args
    import numpy as np
    import torch
    torch.set_num_threads(threads)
    import polgen_rvc_amd  # noqa: F401
    from polgen_rvc_amd import synthetic as S
    from oracle import pipeline as OP, rmvpe as OR
    rcfg = S.RMVPE_CFG_FULL
    a = OP.highpass(S.make_clip(clip, seconds).astype(np.float64))
    a = np.pad(a, (16000, 16000), mode="reflect").astype(np.float32)
    out = []
    for seed in seeds:
        sd = S.to_torch(S.rmvpe_state(rcfg, seed))
        f0, hid, _ = OR.infer_f0(sd, rcfg, a, 0.03, 50, 1100, return_hidden=True)
        bad = OR.unstable_frames(hid, 0.03, 50, 1100, rel=rel)
        f0m = 1127 * np.log(1 + f0 / 700)
        m0, m1 = 1127 * np.log(1 + 50 / 700), 1127 * np.log(1 + 1100 / 700)
        q = (f0m - m0) * 254 / (m1 - m0) + 1
        tie = (f0 > 0) & (np.abs(q - np.floor(q) - 0.5) < tie_margin)
        vf = float((f0 > 0).mean())
        ok = len(bad) == 0 and not tie.any() and 0.15 < vf < 0.995
        print(f"seed {seed}: unstable {len(bad)}, ties {int(tie.sum())}, voiced {vf:.2f}{'  <== OK' if ok else ''}", flush=True)
        out.append((seed, len(bad), int(tie.sum()), vf, ok))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=95.0)
    ap.add_argument("--clip", type=int, default=3)
    ap.add_argument("--start", type=int, default=3000)
    ap.add_argument("--n", type=int, default=160)
    ap.add_argument("--procs", type=int, default=4)
    ap.add_argument("--rel", type=float, default=2e-4)
    ap.add_argument("--tie-margin", type=float, default=1e-3)
    a = ap.parse_args()
    import multiprocessing as mp
    seeds = [a.start + 100 * k for k in range(a.n)]
    parts = [(seeds[i::a.procs], a.seconds, a.clip, max(1, 8 // a.procs), a.rel, a.tie_margin) for i in range(a.procs)]
    with mp.get_context("spawn").Pool(a.procs) as pool:
        res = sum(pool.map(scan, parts), [])
    good = sorted(r[0] for r in res if r[4])
    print("qualifying seeds:", good)


if __name__ == "__main__":
    main()
