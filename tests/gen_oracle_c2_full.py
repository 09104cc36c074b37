#!/usr/bin/env python3
"""Writes tests/golden/oracle_c2_30s_48k_all.npz: EVERY float sample of the pinned CPU oracle's C2 conversion
(30 s, 48 k, rmvpe+, full-size models of seed 1900, clip 0, noise seed 11).

The oracle needs ~10 minutes of host time for this clip, so the GPU test (tests/test_gpu_c3_full.py) reads the stored
waveform instead of running it on the GPU box; the Gaussian noise is NOT stored -- the test redraws it from the same
torch generator in the reference's order (z, then source, chunk by chunk).  Run from the repo root:
    python tests/gen_oracle_c2_full.py
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "polgen-rvc_amd"))
import polgen_rvc_amd  # noqa: E402,F401
from oracle import pipeline as OP  # noqa: E402
from polgen_rvc_amd import synthetic as S  # noqa: E402

SEED, CLIP, SECONDS, NOISE_SEED = 1900, 0, 30.0, 11


def main():
    hcfg, rcfg, scfg = S.HUBERT_CFG_BASE, S.RMVPE_CFG_FULL, S.SYNTH_CFG_48K
    hs, rs, ss = S.hubert_state(hcfg, SEED), S.rmvpe_state(rcfg, SEED), S.synth_state(scfg, SEED)
    audio = S.make_clip(CLIP, SECONDS)
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    models = OP.Models(S.to_torch(hs), hcfg, S.to_torch(rs), rcfg, S.to_torch(ss), scfg)
    t0 = time.time()
    pcm, parts = OP.pipeline(models, OP.Geometry(48000, 1, 6, 38, 41), audio, 0.0, 0, None, 0.0, 1.0, 0.33, 50, 1100,
                             seed=NOISE_SEED, return_parts=True)
    print(f"oracle: {time.time() - t0:.0f} s, {len(pcm)} samples")
    assert np.array_equal(OP.to_int16(np.asarray(parts["audio_f32"], np.float32)), pcm)   # the test derives the PCM
    out = os.path.join(ROOT, "tests", "golden", "oracle_c2_30s_48k_all.npz")
    np.savez_compressed(out, audio_f32=np.asarray(parts["audio_f32"], np.float32),
                        noise_shapes=np.array([[int(np.prod(z.shape)), int(np.prod(s.shape))] for z, s in parts["noises"]]),
                        model_seed=SEED, clip=CLIP, seconds=SECONDS, noise_seed=NOISE_SEED)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
