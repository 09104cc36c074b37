#!/usr/bin/env python3
"""Driver of the host-side sanitizer build (make host-asan; tools/host_asan.sh runs it with the ASan runtime preloaded and
RVCX_LIBRARY=build/asan/librvcx_asan.so).  The library's HIP runtime is tools/hipstub: "device" memory is host memory and
kernels do not run, so outputs are meaningless -- what is exercised under AddressSanitizer + UBSan is the host code:
checkpoint folding / packing for all five model kinds (+ both index kinds), weight regions (free / reload / clone /
adopt), the micro-batch planner over BASELINE configs[4]'s 256 lengths, bucket lengths, the chunk planner of clips longer
than x_max, the f0-file track, arena arithmetic, error paths, the FLAC codec.  Prints HOST_ASAN_OK at the end."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RVCX_DEBUG", "1")
import numpy as np

import polgen_rvc_amd  # noqa: F401
from polgen_rvc_amd import _lib, synthetic as S, weights as W

assert "asan" in _lib.LIB_PATH, "run through tools/host_asan.sh (RVCX_LIBRARY must name the sanitizer build)"
L = _lib.lib()
name, total = _lib.device_info(0)
assert "hipstub" in name, name


def params(**kw):
    p = _lib.Params(0.0, 50.0, 1100.0, 0.0, 0.33, 1.0, 0, 1, 6, 38, 41, 7)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def load_all(ctx, full):
    hcfg = S.HUBERT_CFG_BASE if full else S.HUBERT_CFG_TINY
    rcfg = S.RMVPE_CFG_FULL if full else S.RMVPE_CFG_TINY
    ctx.load_hubert(W.hubert_cfg_struct(hcfg), S.hubert_state(hcfg, 1))
    ctx.load_rmvpe(W.rmvpe_cfg_struct(rcfg), S.rmvpe_state(rcfg, 1))
    sd = S.fcpe_state(S.FCPE_CFG_FULL if full else S.FCPE_CFG_TINY, 2)
    ctx.load_fcpe(W.fcpe_cfg_struct(W.fcpe_cfg_from_state(sd)), sd)
    ctx.load_crepe(S.crepe_state("full" if full else "tiny", 3))
    mids = []
    for scfg in ([S.SYNTH_CFG_48K, S.SYNTH_CFG_40K] if full else [S.SYNTH_CFG_TINY]):
        mids.append(ctx.load_synth(W.synth_cfg_struct(scfg, hcfg["embed_dim"]), S.synth_state(scfg, 4, input_dim=hcfg["embed_dim"])))
    return hcfg, mids


# ---- tiny models with real (random) values: folding, packing, fp16 images, legacy weight-norm names, fp16 tensors
ctx = _lib.Context(0)
hcfg, mids = load_all(ctx, full=False)
ss = S.synth_state(S.SYNTH_CFG_TINY, 5, input_dim=hcfg["embed_dim"])
legacy = {k.replace(".parametrizations.weight.original0", ".weight_g").replace(".parametrizations.weight.original1", ".weight_v"):
          np.asarray(v).astype(np.float16) for k, v in ss.items()}
mids.append(ctx.load_synth(W.synth_cfg_struct(S.SYNTH_CFG_TINY, hcfg["embed_dim"]), legacy))
regions, layout = ctx.weights_regions()
assert len(regions) >= 5 and all(n > 0 for _, n in regions)
big = S.make_index(2048, hcfg["embed_dim"], 1)
ctx.load_index(big)
g = np.random.Generator(np.random.PCG64(1))
cent = big[g.choice(len(big), 16, replace=False)]
assign = np.argmin(((big[:, None, :] - cent[None]) ** 2).sum(-1), axis=1).astype(np.int32)
ctx.load_index_ivf(big, cent, assign, 1)
ctx.load_index(None)
# conversions: single clip, ragged batch, a clip long enough to be cut into chunks, f0 file, fcpe / crepe back-ends
clips = [S.make_clip(i, s) for i, s in enumerate([1.2, 2.5, 2.5, 1.7, 3.1])]
for m in mids:
    ctx.convert_batch(m, clips, params())
    ctx.convert_batch(m, clips[:1], params(volume_envelope=0.25, resample_sr=16000), want_f32=True)
long_clip = S.make_clip(9, 47.0)
ctx.convert_batch(mids[0], [long_clip, long_clip], params())
ctx.convert_batch(mids[0], clips[:2], params(f0_method=_lib.F0_FCPE))
tab = np.stack([np.linspace(0, 2, 21), np.full(21, 220.0)], axis=1).astype(np.float32)
ctx.convert_batch(mids[0], clips[:2], params(), inp_f0=[tab, tab])
assert len(_lib.f0_file_track(tab)) > 0 and len(_lib.f0_file_track(tab[:1])) >= 0
ctx.get_f0_x_ex(np.zeros(16000 * 3, np.float32), 300, params(), tab)
ctx.rmvpe_f0(clips[1], 0.03, 50, 1100)
ctx.hubert_features(clips[1], hcfg["embed_dim"])
ctx.resample(np.stack([clips[1], clips[1]], axis=1).astype(np.float64), 44100, 16000)
# region free / reload cycle, clone into a second context, adopt
for _ in range(3):
    L.rvcx_unload_synth(ctx._h, mids[-1])
    mids[-1] = ctx.load_synth(W.synth_cfg_struct(S.SYNTH_CFG_TINY, hcfg["embed_dim"]), ss)
ctx2 = _lib.Context(0)
with S.shapes_only():
    load_all(ctx2, full=False)
    ctx2.load_synth(W.synth_cfg_struct(S.SYNTH_CFG_TINY, hcfg["embed_dim"]), S.synth_state(S.SYNTH_CFG_TINY, 5, input_dim=hcfg["embed_dim"]))
ctx2.weights_clone(ctx)
ctx2.weights_adopt()
ctx2.convert_batch(0, clips[:2], params())
# error paths: nothing may be read or written out of bounds on the way to the error code
for bad in (lambda: ctx.convert_batch(99, clips[:1], params()), lambda: ctx.convert_batch(mids[0], [np.zeros(0, np.float32)], params()),
            lambda: ctx2.weights_clone(ctx2) if False else ctx.load_synth(W.synth_cfg_struct(S.SYNTH_CFG_TINY, hcfg["embed_dim"]), {"x": np.zeros(3, np.float32)})):
    try:
        bad()
    except _lib.RvcxError:
        pass
ctx2.close()
ctx.close()

# ---- full-size models from shape-only placeholders (what ranks != 0 load): the real layouts, region chunking, the planner
ctx = _lib.Context(0)
with S.shapes_only():
    hcfg, (m48, m40) = load_all(ctx, full=True)
g = np.random.Generator(np.random.PCG64(5))
lengths = [int(round(s * 100)) * 160 for s in g.uniform(3.0, 15.0, 256)]          # bench.py c5_lengths()
p = params()
seen = set()
for n in lengths:
    mb = ctx.micro_batch(m48, n, p)
    bl = int(L.rvcx_bucket_length(ctx._h, m40, n, _lib.C.byref(p)))
    assert 1 <= mb <= 16 and n <= bl < n + 128 * 160, (n, mb, bl)      # the longest clip of the length class (128 frames wide)
    seen.add(bl)
assert 5 < len(seen) < 256
zero_clips = [np.zeros(n, np.float32) for n in lengths[:40]]
pcm = ctx.convert_batch(m40, zero_clips, p)
assert len(pcm) == 40 and sum(ctx.last_micro_batches()) == 40
ctx.convert_batch(m48, [np.zeros(16000 * 30, np.float32)] * 3, params(index_rate=0.0))
ctx.close()

# ---- the FLAC codec is host code too
pcm = (np.sin(np.arange(30000) * 0.01) * 20000).astype(np.int16)
blob = _lib.flac_encode(np.stack([pcm, pcm[::-1]], axis=1), 48000)
back, sr, bits = _lib.flac_decode(blob)
assert sr == 48000 and bits == 16 and np.array_equal(back[:, 0], pcm)
for cut in (10, 50, len(blob) // 2, len(blob) - 3):
    try:
        _lib.flac_decode(blob[:cut])
    except _lib.RvcxError:
        pass
print("HOST_ASAN_OK launches", int(L.hipstub_launches()) if hasattr(L, "hipstub_launches") else -1)
