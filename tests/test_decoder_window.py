"""CPU: the receptive field the decoder window relies on (round 6).

VC.pipeline throws t_pad_tgt samples of every decoder call away at both ends (rvc/infer/pipeline.py:432-447).  The product
evaluates the NSF decoder only on the frames that can reach the kept samples: frames [skip, len - skip) with
skip <= t_pad frames - synth_dec_rf(cfg) (csrc/synth.hip: SynthIO::dec_skip; weights.synth_dec_rf is its host-side twin).
Here the bound itself is checked on the ORACLE's decoder (oracle/synth.py: nsf_decoder, pinned against the reference's
GeneratorNSF by tests/test_oracle_golden.py): a change of z at one frame must leave every output sample further than
synth_dec_rf frames away untouched -- exactly, the decoder being a stack of local convolutions over z (the harmonic source,
whose phase is a prefix sum, is evaluated whole by the product and is not perturbed here)."""
import math

import numpy as np
import pytest
import torch

import polgen_rvc_amd  # noqa: F401
from polgen_rvc_amd import synthetic as S, weights as W
from oracle import synth as OS


@pytest.mark.parametrize("name,T", [("tiny", 120), ("48k", 48), ("40k", 48)])
def test_a_frame_of_z_reaches_no_further_than_the_receptive_field_bound(name, T):
    cfg = {"tiny": S.SYNTH_CFG_TINY, "48k": S.SYNTH_CFG_48K, "40k": S.SYNTH_CFG_40K}[name]
    rf = W.synth_dec_rf(cfg)
    inter, gin, ups = cfg[2], cfg[16], cfg[12]
    upp = math.prod(ups)
    sd = {k: torch.from_numpy(v) for k, v in S.synth_state(cfg, 5).items() if k.startswith("dec.")}
    g = torch.Generator().manual_seed(1)
    z = torch.randn(1, inter, T, generator=g)
    f0 = 180.0 + 40.0 * torch.sin(torch.arange(T) / 7.0)[None]
    gv = torch.randn(1, gin, 1, generator=g) * 0.1
    sn = torch.randn(1, T * upp, 1, generator=g)
    with torch.no_grad():
        y0 = OS.nsf_decoder(sd, cfg, z, f0, gv, sn)[0, 0].numpy()
        extents = []
        for j in (T // 2, T // 2 + 1):
            z1 = z.clone()
            z1[0, :, j] += 1.0
            y1 = OS.nsf_decoder(sd, cfg, z1, f0, gv, sn)[0, 0].numpy()
            ch = np.nonzero(y1 != y0)[0]
            assert ch.size, "the perturbation must be visible"
            lo, hi = ch.min() / upp - j, ch.max() / upp - j       # reach in frames, relative to the perturbed frame
            extents.append((lo, hi))
            assert ch.min() >= (j - rf) * upp and ch.max() < (j + rf + 1) * upp, (name, j, lo, hi, rf)
    print(f"{name}: bound {rf} frames each side; measured reach {extents}")
    # and the bound is not idle: the measured reach is most of it (the ConvTranspose1d terms and the + 2 are the slack)
    assert max(-extents[0][0], extents[0][1]) > 0.75 * rf - 2


def test_the_formula_on_the_three_shipped_geometries():
    assert W.synth_dec_rf(S.SYNTH_CFG_48K) == 14 and W.synth_dec_rf(S.SYNTH_CFG_40K) == 15
    # t_pad = 100 frames at x_pad = 1: the window drops (100 - 14) & ~3 = 84 frames at each end of a 48 k decoder call
    assert (100 - W.synth_dec_rf(S.SYNTH_CFG_48K)) & ~3 == 84
