"""RMVPE F0 (rvcx_rmvpe_f0) and HuBERT features (rvcx_hubert_features) on the GPU vs the committed
reference goldens and the CPU oracle.  fp32; tolerances stated per test."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import rms

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_bigru(ctx):
    """nn.GRU(384, 256, bidirectional) forward (RMVPE.py:125-137) vs torch-CPU; tolerance 1e-5 rel."""
    from polgen_rvc_amd import synthetic as S
    sd = S.rmvpe_state(S.RMVPE_CFG_TINY, 2)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 45, 384, generator=g)
    gru = torch.nn.GRU(384, 256, batch_first=True, bidirectional=True)
    gru.load_state_dict({k[len("fc.0.gru."):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith("fc.0.gru.")})
    with torch.no_grad():
        ref = gru(x)[0].numpy()
    got = ctx.bigru(x.numpy(), sd)
    assert rms(got - ref) / rms(ref) < 1e-5


@pytest.mark.parametrize("tag", ["tiny", "full_1s"])
def test_rmvpe_vs_reference_golden(ctx, tag):
    """hidden (salience) within 1e-4 relative RMS of the reference's E2E; f0 within 1e-3 relative on
    frames both call voiced, voicing decisions identical except where the salience max is within
    1e-5 of the 0.03 threshold."""
    from polgen_rvc_amd import synthetic as S, weights as W
    d = np.load(os.path.join(GOLD, f"rmvpe_{tag}.npz"))
    cfg = json.loads(str(d["cfg"]))
    ctx.load_rmvpe(W.rmvpe_cfg_struct(cfg), S.rmvpe_state(cfg, int(d["seed"])))
    f0, hid = ctx.rmvpe_f0(d["audio"], return_hidden=True)
    st = int(d["stride"])
    e = rms(hid[0, ::st] - d["hidden"]) / rms(d["hidden"])
    print(f"rmvpe {tag}: hidden rel err {e:.3e}; voiced {int((f0 > 0).sum())}/{f0.size}")
    assert e < 1e-4
    ref = d["f0"]
    both = (ref > 0) & (f0[0] > 0)
    assert (np.abs(f0[0][both] - ref[both]) / ref[both]).max() < 1e-3
    assert ((ref > 0) != (f0[0] > 0)).mean() < 0.01


def test_rmvpe_batch_equals_single(ctx):
    from polgen_rvc_amd import synthetic as S, weights as W
    cfg = S.RMVPE_CFG_TINY
    ctx.load_rmvpe(W.rmvpe_cfg_struct(cfg), S.rmvpe_state(cfg, 1))
    a = np.stack([S.make_clip(20 + i, 1.0) for i in range(3)])
    fb, hb = ctx.rmvpe_f0(a, return_hidden=True)
    for i in range(3):
        f1, h1 = ctx.rmvpe_f0(a[i], return_hidden=True)
        assert rms(hb[i] - h1[0]) / rms(h1[0]) < 1e-5


@pytest.mark.parametrize("tag", ["tiny", "base_1s"])
def test_hubert_vs_hf_twin_golden(ctx, tag):
    """HuBERT parity is UNPINNED by the reference (fairseq not vendored); the golden is the
    architecture-identical transformers.HubertModel run on the same synthetic weights.
    Tolerance 1e-4 relative RMS on the layer-L output."""
    from polgen_rvc_amd import synthetic as S, weights as W
    d = np.load(os.path.join(GOLD, f"hubert_{tag}.npz"))
    cfg = json.loads(str(d["cfg"]))
    ctx.load_hubert(W.hubert_cfg_struct(cfg), S.hubert_state(cfg, int(d["seed"])))
    got = ctx.hubert_features(d["wav"], cfg["embed_dim"], cfg["layers"])
    e = rms(got - d["out"]) / rms(d["out"])
    got1 = ctx.hubert_features(d["wav"], cfg["embed_dim"], 1)
    e1 = rms(got1 - d["out_l1"]) / rms(d["out_l1"])
    print(f"hubert {tag}: rel err layer L {e:.3e}, layer 1 {e1:.3e}")
    assert e < 1e-4 and e1 < 1e-4
