"""The time-major Linear kernels of the transformer sections (csrc/gemm.hip) and their LayerNorm (ops.hip:
layernorm_tm) against plain torch fp32 references of the same ops.

Tolerances: split-fp16 GEMM <= 2e-6 relative RMS (three fp16 MFMAs of an 11 + 11 bit split: ~2^-21 per product, the
size of fp32 rounding; torch's own fp32 matmul sits at ~2e-7 of the float64 result), exact-fp32 GEMM <= 1e-6,
LayerNorm <= 1e-6; the decoded split-form output within 2^-21 relative of the fp32 output; every tile bit-identical."""
import numpy as np
import pytest
import torch

from conftest import rms

pytestmark = pytest.mark.gpu


def _ref(x_cf, w, bias, res, act):
    x = torch.from_numpy(x_cf).double().permute(0, 2, 1).reshape(-1, x_cf.shape[1])        # (B T, Cin)
    y = x @ torch.from_numpy(w).double().T
    if bias is not None:
        y = y + torch.from_numpy(bias).double()
    if act == 3:
        y = torch.nn.functional.gelu(y)
    elif act == 2:
        y = torch.relu(y)
    if res is not None:
        y = y + torch.from_numpy(res).double()
    return y.numpy()


@pytest.mark.parametrize("B,T,cin,cout,act,use_res", [
    (1, 1599, 768, 2304, 0, False),      # HuBERT q/k/v of a 30 s clip
    (1, 1599, 3072, 768, 0, True),       # fc2 + residual
    (1, 399, 768, 3072, 3, False),       # fc1 + GELU
    (2, 77, 128, 160, 3, True),          # tiny HuBERT shapes, batch folded into the rows, ragged tiles
    (3, 33, 160, 128, 2, False),
    (1, 5, 48, 36, 0, True),             # Cin = 3 chunks (odd stage count), Cout not a multiple of 32
    (5, 333, 768, 768, 0, True),         # rows not a multiple of the 256-row B-direct tile, ragged last tile
    (2, 700, 176, 256, 3, False),        # Cin = 11 chunks: odd stage count on the B-direct tile
])
def test_gemm_tm_vs_torch(ctx, B, T, cin, cout, act, use_res):
    from polgen_rvc_amd import _lib
    g = np.random.Generator(np.random.PCG64(B * 1000 + T + cin))
    x = g.standard_normal((B, cin, T)).astype(np.float32)
    w = (g.standard_normal((cout, cin)) / np.sqrt(cin)).astype(np.float32)
    bias = g.standard_normal(cout).astype(np.float32)
    res = g.standard_normal((B * T, cout)).astype(np.float32) if use_res else None
    ref = _ref(x, w, bias, res, act)
    y, ycf, ysp = ctx.gemm_tm(x, w, bias, res, act)
    e = rms(y - ref) / rms(ref)
    print(f"gemm_tm {B}x{T} {cin}->{cout}: rel err {e:.2e}")
    assert np.isfinite(y).all() and e < 2e-6
    assert np.array_equal(ycf, y.reshape(B, T, cout).transpose(0, 2, 1))          # the channel-first copy: same values
    if ysp is not None:
        assert np.abs(ysp - y).max() <= 2.0 ** -20 * np.abs(y).max() + 1e-30      # split form: 22 significant bits
    try:
        outs = []
        for tile in range(5):      # 4: gemm_bd (round 6: activations straight from global memory; taken when cout % 128 == 0)
            _lib.Context.conv_override(tile=200 + tile)
            outs.append(ctx.gemm_tm(x, w, bias, res, act)[0])
        for o in outs[1:]:
            assert np.array_equal(o, outs[0])                                     # k-order is tile independent
        assert np.array_equal(outs[0], y)
    finally:
        _lib.Context.conv_override()
    yf = ctx.gemm_tm(x, w, bias, res, act, exact_fp32=True)[0]
    ef = rms(yf - ref) / rms(ref)
    assert ef < 1e-6, ef


def test_gemm_tm_split_output_flags_values_beyond_fp16_range(ctx):
    """The producer of a split-form tensor range-checks what it writes (conv.h: kH3ActLimit): an output of 1e5 must make
    the call fall back (rvcx_fp32_reruns) and still return the fp32 answer."""
    g = np.random.Generator(np.random.PCG64(5))
    x = g.standard_normal((1, 64, 40)).astype(np.float32)
    w = (g.standard_normal((32, 64)) * 4e3).astype(np.float32)                    # outputs ~ 3e4 rms, tails > 6e4
    n0 = ctx.fp32_reruns()
    y, _, ysp = ctx.gemm_tm(x, w)
    ref = _ref(x, w, None, None, 0)
    assert np.abs(ref).max() > 6.5e4
    assert ctx.fp32_reruns() == n0 + 1
    assert rms(y - ref) / rms(ref) < 1e-6


@pytest.mark.parametrize("rows,C", [(1599, 768), (7, 128), (130, 512), (3, 36)])
def test_layernorm_tm_vs_torch(ctx, rows, C):
    g = np.random.Generator(np.random.PCG64(rows + C))
    x = (g.standard_normal((rows, C)) * 3 + 0.7).astype(np.float32)
    gamma, beta = g.standard_normal(C).astype(np.float32), g.standard_normal(C).astype(np.float32)
    ref = torch.nn.functional.layer_norm(torch.from_numpy(x).double(), (C,), torch.from_numpy(gamma).double(),
                                         torch.from_numpy(beta).double(), 1e-5).numpy()
    y, ysp = ctx.layernorm_tm(x, gamma, beta)
    e = rms(y - ref) / rms(ref)
    assert e < 1e-6, e
    if ysp is not None:
        assert np.abs(ysp - y).max() <= 2.0 ** -20 * np.abs(y).max()


@pytest.mark.parametrize("B,T,cin,cout,act,use_res", [(1, 1599, 3072, 768, 0, True), (2, 100, 2048, 256, 3, False),
                                                      (1, 40, 4096, 64, 0, True), (3, 700, 3072, 768, 0, True)])
def test_long_k_layers_sum_in_segments_split_or_not(ctx, B, T, cin, cout, act, use_res):
    """K >= 2048 (HuBERT fc2): the canonical order is four K segments, each summed from zero, added left to right.  One
    workgroup per tile with a second accumulator set (what a batch runs) and four workgroups per tile + the finish pass (what an
    underfilled single-utterance launch runs) give the same bits -- every output form -- and the torch result."""
    from polgen_rvc_amd import _lib
    g = np.random.Generator(np.random.PCG64(B * 100 + T + cin))
    x = g.standard_normal((B, cin, T)).astype(np.float32)
    w = (g.standard_normal((cout, cin)) / np.sqrt(cin)).astype(np.float32)
    bias = g.standard_normal(cout).astype(np.float32)
    res = g.standard_normal((B * T, cout)).astype(np.float32) if use_res else None
    ref = _ref(x, w, bias, res, act)
    try:
        _lib.Context.conv_override(splitk=1)
        one = ctx.gemm_tm(x, w, bias, res, act)
        _lib.Context.conv_override(splitk=2)
        four = ctx.gemm_tm(x, w, bias, res, act)
    finally:
        _lib.Context.conv_override()
    auto = ctx.gemm_tm(x, w, bias, res, act)
    e = rms(one[0] - ref) / rms(ref)
    print(f"gemm_tm long K {B}x{T} {cin}->{cout}: rel err {e:.2e}")
    assert np.isfinite(one[0]).all() and e < 2e-6
    try:
        _lib.Context.conv_override(tile=204)            # gemm_bd's long-K form (128 x 128, second accumulator set)
        bd = ctx.gemm_tm(x, w, bias, res, act)
    finally:
        _lib.Context.conv_override()
    for a_, b_, c_, d_ in zip(one, four, auto, bd):
        if a_ is not None:
            assert np.array_equal(a_, b_) and np.array_equal(a_, c_) and np.array_equal(a_, d_)
