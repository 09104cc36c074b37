import sys, time
sys.path.insert(0,'/root/repo')
import numpy as np
import polgen_rvc_amd
from polgen_rvc_amd import _lib, synthetic as S
ctx = _lib.Context(0)
ctx.load_crepe(S.crepe_state("full", 3))
x = np.pad(S.make_clip(5, 30.0), (16000, 16000), mode="reflect")
for hop in (128, 160):
    F = ctx.crepe_frames(len(x), hop)
    d = np.zeros(F, np.float32)
    ctx.crepe_predict(x, hop, 50, 1100, dither=d)
    t0 = time.perf_counter(); ctx.crepe_predict(x, hop, 50, 1100, dither=d); t = time.perf_counter() - t0
    print(f"crepe-full, 32 s padded clip, hop {hop}: {F} frames in {t*1e3:.1f} ms")
