import sys, time
sys.path.insert(0,'/root/repo')
import numpy as np, torch
import polgen_rvc_amd
from polgen_rvc_amd import _lib, synthetic as S
from oracle import crepe as OC
cap = sys.argv[1] if len(sys.argv) > 1 else "tiny"
st = S.crepe_state(cap, 3)
sd = S.to_torch(st)
x = S.make_clip(5, 3.0)
hop = 128
F = OC.n_frames(len(x), hop)
noise = np.random.default_rng(0).triangular(-20, 0, 20, size=F).astype(np.float32)
xq = x.astype(np.float32); xq = xq / np.quantile(np.abs(xq), 0.999)
opitch, parts = OC.predict(sd, xq, hop, 50, 1100, noise, batch_size=2*hop, return_parts=True)
ctx = _lib.Context(0)
ctx.load_crepe(st)
t0 = time.time()
pitch, probs, bins = ctx.crepe_predict(x, hop, 50, 1100, dither=noise, return_parts=True)
print("gpu s", time.time() - t0, "F", F)
print("probs max abs err", np.abs(probs - parts["probs"]).max(), "oracle probs range", parts["probs"].min(), parts["probs"].max())
print("bins equal frac", (bins == parts["bins"]).mean(), "distinct", len(set(bins.tolist())))
ob, op = [], []
for i in range(0, F, 2*hop):
    b, p_ = OC.decode_batch(torch.from_numpy(probs[i:i+2*hop]), 50, 1100, noise[i:i+2*hop]); ob.append(b); op.append(p_)
ob = np.concatenate(ob); op = np.concatenate(op)
print("decode of the GPU's own probs: bins equal", np.array_equal(ob, bins), "pitch max rel err", np.abs(op - pitch).max() / op.max())
p2, b2 = ctx.crepe_decode(parts["probs"], 2*hop, 50, 1100, noise)
print("op decode of the oracle's probs: bins equal", np.array_equal(b2, parts["bins"]), "pitch rel", np.abs(p2 - opitch).max() / opitch.max())
params = _lib.Params(2.0, 50.0, 1100.0, 0.0, 0.33, 1.0, 0, 1, 6, 38, 41, 5, _lib.F0_CREPE, 0, hop, 0)
p_len = len(x) // 160
coarse, f0 = ctx.get_f0_crepe_x(x, p_len, params, dither=noise)
of0 = OC.get_f0_crepe(sd, x, 50, 1100, p_len, hop, noise) * 2 ** (2.0 / 12)
print("get_f0: f0 max rel err", np.abs(f0 - of0).max() / of0.max(), "frames differing > 1e-4 rel", (np.abs(f0 - of0) > 1e-4 * of0.max()).sum(), "of", p_len)
