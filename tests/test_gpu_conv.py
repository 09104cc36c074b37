"""Parity of the MFMA implicit-GEMM conv kernel family (polgen-rvc_amd/csrc/conv.hip) against
torch-CPU fp32 (floating-point kernel: tolerance 2e-5 relative RMS, fp32 accumulate order)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rms

pytestmark = pytest.mark.gpu
TOL = 2e-5


def _chk(got, ref, tag=""):
    ref = ref.numpy() if hasattr(ref, "numpy") else ref
    e = rms(got - ref) / max(rms(ref), 1e-12)
    assert np.isfinite(got).all(), tag
    assert e < TOL, f"{tag}: rel rms err {e:.3e}"


CASES_1D = [
    # B, Cin, Tin, Cout, K, stride, dil, groups
    (1, 32, 300, 32, 3, 1, 1, 1),
    (2, 64, 517, 64, 7, 1, 3, 1),
    (1, 128, 400, 128, 11, 1, 5, 1),
    (1, 256, 257, 256, 3, 1, 1, 1),
    (1, 192, 100, 768, 3, 1, 1, 1),
    (1, 768, 131, 192, 1, 1, 1, 1),
    (1, 1, 4000, 48, 10, 5, 1, 1),      # HuBERT conv0 shape (Cin=1, strided)
    (2, 48, 799, 48, 3, 2, 1, 1),
    (1, 48, 399, 48, 2, 2, 1, 1),
    (1, 512, 1001, 512, 3, 2, 1, 1),    # HuBERT conv1-4 shape: stride-2 tile family
    (2, 512, 300, 512, 2, 2, 1, 1),     # HuBERT conv5-6 shape
    (1, 64, 4097, 200, 3, 2, 1, 1),
    (1, 1, 6000, 40, 24, 12, 1, 1),     # noise conv (Cin=1, k=2*stride)
    (1, 128, 70, 128, 128, 1, 1, 16),   # HuBERT pos_conv (grouped, k=128); 8 channels per group: the generic kernel
    (2, 768, 203, 768, 128, 1, 1, 16),  # its real shape, 48 channels per group: split-fp16 tiles, blockIdx.z = (item, group)
    (1, 96, 333, 64, 5, 1, 1, 2),       # grouped with Cout_g != Cin_g
    (1, 24, 50, 288, 1, 1, 1, 1),       # odd channel counts -> padding guards
    (3, 5, 33, 7, 5, 1, 1, 1),
    (1, 32, 1, 1, 7, 1, 1, 1),          # conv_post shape, T=1
]


@pytest.mark.parametrize("case", CASES_1D)
def test_conv1d(ctx, case):
    B, Cin, Tin, Cout, K, s, d, g = case
    gen = torch.Generator().manual_seed(hash(case) % 2**31)
    x = torch.randn(B, Cin, Tin, generator=gen)
    w = torch.randn(Cout, Cin // g, K, generator=gen) / (Cin // g * K) ** 0.5
    b = torch.randn(Cout, generator=gen)
    pad = (K * d - d) // 2 if s == 1 else 0
    if g > 1:
        pad = K // 2
    ref = F.conv1d(x, w, b, stride=s, dilation=d, padding=pad, groups=g)
    got = ctx.conv1d(x.numpy(), w.numpy(), b.numpy(), stride=s, dil=d, pad_left=pad, groups=g, Tout=ref.shape[2])
    _chk(got, ref, str(case))


def test_conv1d_fused_epilogue(ctx):
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(2, 64, 333, generator=gen)
    w = torch.randn(64, 64, 7, generator=gen) / (64 * 7) ** 0.5
    b = torch.randn(64, generator=gen)
    r = torch.randn(2, 64, 333, generator=gen)
    # pre-activation lrelu(0.1) on the input, residual add after (ResBlock1 pattern, residuals.py:45-53)
    ref = F.conv1d(F.leaky_relu(x, 0.1), w, b, padding=9, dilation=3) + r
    got = ctx.conv1d(x.numpy(), w.numpy(), b.numpy(), res=r.numpy(), dil=3, pad_left=9, pre_lrelu=0.1)
    _chk(got, ref, "pre+res")
    for act, fn in ((1, lambda t: F.leaky_relu(t, 0.1)), (2, F.relu), (3, F.gelu), (4, torch.tanh),
                    (5, torch.sigmoid)):
        ref = fn(F.conv1d(x, w, b, padding=3)) + r
        got = ctx.conv1d(x.numpy(), w.numpy(), b.numpy(), res=r.numpy(), pad_left=3, act=act, act_slope=0.1)
        _chk(got, ref, f"act{act}")


def test_grouped_conv_with_the_pos_conv_epilogue_ragged(ctx):
    """x + gelu(pos_conv(x)) of a ragged batch (hubert.hip): grouped split-fp16 tiles with the GELU, the residual and the
    per-item zero padding; every item equals its single run bit for bit and torch within the tolerance."""
    gen = torch.Generator().manual_seed(9)
    lens = [150, 97, 31]
    x = torch.randn(3, 768, 150, generator=gen)
    for i, L in enumerate(lens):
        x[i, :, L:] = 0
    w = torch.randn(768, 48, 128, generator=gen) / (48 * 128) ** 0.5
    b = torch.randn(768, generator=gen)
    got = ctx.conv1d(x.numpy(), w.numpy(), b.numpy(), res=x.numpy(), pad_left=64, groups=16, act=3, Tout=150, lens_in=lens,
                     lens_out=lens)
    for i, L in enumerate(lens):
        xi = x[i:i + 1, :, :L]
        ref = (xi + F.gelu(F.conv1d(xi, w, b, padding=64, groups=16)[:, :, :L]))[0]
        _chk(got[i, :, :L], ref, f"item{i}")
        assert (got[i, :, L:] == 0).all()
        one = ctx.conv1d(xi.numpy(), w.numpy(), b.numpy(), res=xi.numpy(), pad_left=64, groups=16, act=3, Tout=L)
        assert np.array_equal(one[0], got[i, :, :L])


def test_conv1d_ragged_lengths(ctx):
    """(B,C,Tmax) batches with per-item lengths must equal running each item alone (zero padding at
    each item's own end -- the x_mask semantics of encoders.py:120-123)."""
    gen = torch.Generator().manual_seed(5)
    lens = [97, 64, 1]
    x = torch.randn(3, 48, 97, generator=gen)
    w = torch.randn(96, 48, 3, generator=gen) / 12
    b = torch.randn(96, generator=gen)
    got = ctx.conv1d(x.numpy(), w.numpy(), b.numpy(), pad_left=1, lens_in=lens, lens_out=lens)
    for i, L in enumerate(lens):
        ref = F.conv1d(x[i:i + 1, :, :L], w, b, padding=1)[0]
        _chk(got[i, :, :L], ref, f"item{i}")
        assert (got[i, :, L:] == 0).all()


@pytest.mark.parametrize("case", [(1, 64, 50, 32, 24, 12, 6), (2, 80, 37, 40, 8, 4, 2), (1, 40, 29, 20, 7, 3, 2),
                                  (1, 128, 100, 64, 16, 10, 3), (1, 64, 300, 32, 4, 2, 1),
                                  # k = 4, stride 2, C = 64 / 128: the streaming kernel of round 5 (csrc/convt_thin.hip)
                                  (2, 128, 1000, 64, 4, 2, 1), (1, 64, 33, 32, 4, 2, 1), (3, 128, 31, 64, 4, 2, 1),
                                  (1, 64, 70001, 32, 4, 2, 1), (1, 128, 9000, 64, 4, 2, 1)])
def test_convtranspose1d(ctx, case):
    B, Cin, Tin, Cout, K, s, p = case
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(B, Cin, Tin, generator=gen)
    w = torch.randn(Cin, Cout, K, generator=gen) / (Cin * K / s) ** 0.5
    b = torch.randn(Cout, generator=gen)
    ref = F.conv_transpose1d(F.leaky_relu(x, 0.1), w, b, stride=s, padding=p)
    got = ctx.convtranspose1d(x.numpy(), w.numpy(), b.numpy(), stride=s, pad=p, pre_lrelu=0.1)
    _chk(got, ref, str(case))


@pytest.mark.parametrize("case", [(1, 1, 40, 128, 16), (2, 16, 33, 64, 32), (1, 64, 12, 8, 128), (1, 256, 6, 4, 512),
                                  (1, 4, 64, 128, 3)])
def test_conv2d3x3(ctx, case):
    B, Cin, H, W, Cout = case
    gen = torch.Generator().manual_seed(13)
    x = torch.randn(B, Cin, H, W, generator=gen)
    w = torch.randn(Cout, Cin, 3, 3, generator=gen) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=gen)
    r = torch.randn(B, Cout, H, W, generator=gen)
    ref = F.relu(F.conv2d(x, w, b, padding=1)) + r          # ConvBlockRes tail, RMVPE.py:171-175
    got = ctx.conv2d3x3(x.numpy(), w.numpy(), b.numpy(), res=r.numpy(), act=2)
    _chk(got, ref, str(case))


@pytest.mark.parametrize("case", [(1, 32, 10, 4, 16), (2, 8, 7, 16, 4), (1, 512, 5, 4, 256), (1, 32, 40, 64, 16)])
def test_convtranspose2d(ctx, case):
    B, Cin, H, W, Cout = case
    gen = torch.Generator().manual_seed(17)
    x = torch.randn(B, Cin, H, W, generator=gen)
    w = torch.randn(Cin, Cout, 3, 3, generator=gen) / (Cin * 9 / 4) ** 0.5
    b = torch.randn(Cout, generator=gen)
    ref = F.relu(F.conv_transpose2d(x, w, b, stride=2, padding=1, output_padding=1))
    got = ctx.convtranspose2d(x.numpy(), w.numpy(), b.numpy(), act=2)
    _chk(got, ref, str(case))


@pytest.mark.parametrize("case", [(1, 32, 70000, 32, 3, 1), (2, 64, 40000, 64, 7, 3), (1, 128, 17000, 128, 11, 5),
                                  (1, 256, 9000, 256, 7, 1)])
def test_conv1d_h3_split_matches_fp64(ctx, case):
    """Long-sequence dense convs run on conv_h3_kernel (fp16 hi/lo split, three v_mfma_f32_32x32x16_f16 per
    product block): the result must be as close to the float64 convolution as an fp32 computation is
    (relative RMS <= 1e-6; torch's own fp32 conv sits at ~2e-7), with the fused prologue/epilogue."""
    B, C, T, Co, K, d = case
    gen = torch.Generator().manual_seed(K * 1000 + C)
    x = torch.randn(B, C, T, generator=gen) * 4
    w = torch.randn(Co, C, K, generator=gen) / (C * K) ** 0.5
    b = torch.randn(Co, generator=gen)
    r = torch.randn(B, Co, T, generator=gen)
    pad = (K * d - d) // 2
    ref = (F.conv1d(F.leaky_relu(x.double(), 0.1), w.double(), b.double(), dilation=d, padding=pad) + r.double()).numpy()
    got = ctx.conv1d(x.numpy(), w.numpy(), b.numpy(), res=r.numpy(), dil=d, pad_left=pad, Tout=T, pre_lrelu=0.1)
    e = rms(got - ref) / rms(ref)
    assert np.isfinite(got).all() and e < 1e-6, e


def test_conv1d_h3_falls_back_when_weights_overflow_fp16(ctx):
    """A layer whose weights would overflow fp16 at the split scale (|w| * 256 >= 60000) keeps the exact-fp32 MFMA
    kernel (ctx.hip: pack_h3 returns no image): results stay correct for weights of magnitude ~1e3."""
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(1, 64, 20000, generator=gen)
    w = torch.randn(64, 64, 3, generator=gen) * 400.0
    b = torch.randn(64, generator=gen)
    ref = F.conv1d(x.double(), w.double(), b.double(), padding=1).numpy()
    got = ctx.conv1d(x.numpy(), w.numpy(), b.numpy(), pad_left=1, Tout=20000)
    assert np.isfinite(got).all() and rms(got - ref) / rms(ref) < 1e-6


@pytest.mark.parametrize("tile", [103, 104, 109, 110])
def test_every_3x3_tile_stores_every_subtile(ctx, tile):
    """Batched (B = 3) 3x3 conv on a U-Net level-0 sized map with each tile of the conv_h3 3x3 family forced: every
    output element is written (the destination is NaN-filled by hipMemset in the op) and equals the B = 1 result
    bit for bit.  Regression for the 32x512 tile whose epilogue stored only two of its four 32x32 sub-tiles --
    a latent bug the batched path's tile choice exposed in round 2."""
    import torch
    g = torch.Generator().manual_seed(0)
    B, C, H, W = 3, 16, 808, 128
    x = torch.randn(B, C, H, W, generator=g).numpy()
    w = (torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5).numpy()
    res = torch.randn(B, C, H, W, generator=g).numpy()
    try:
        ctx.conv_override(109, -1, -1)
        ref = np.concatenate([ctx.conv2d3x3(x[b:b + 1], w, None, res=res[b:b + 1], act=2) for b in range(B)])
        # poison the arena region the next call will reuse for its output
        ctx.conv2d3x3(np.full_like(x, np.nan), w, None, act=0)
        ctx.conv_override(tile, -1, -1)
        got = ctx.conv2d3x3(x, w, None, res=res, act=2)
        assert np.isfinite(got).all()
        assert np.array_equal(got, ref)
    finally:
        ctx.conv_override(-1, -1, -1)


@pytest.mark.parametrize("C,K,d,T", [(32, 3, 1, 5000), (32, 11, 5, 777), (64, 7, 3, 4099), (64, 11, 5, 300),
                                     (128, 3, 1, 1000), (128, 7, 1, 2500), (128, 11, 5, 1531), (128, 11, 3, 100),
                                     (256, 3, 1, 700), (256, 3, 1, 1201)])
def test_fused_resblock_pair_equals_two_launches(ctx, C, K, d, T):
    """One ResBlock1 step (residuals.py:45-53) as one kernel (resblock.hip): against torch fp32 within fp32
    rounding, and bit-identical to the two conv_h3 launches it replaces -- ragged batch (per-item lengths,
    one of them shorter than a tile) included; every output element is written (NaN-filled destination)."""
    g = torch.Generator().manual_seed(C * 100 + K)
    B = 3
    x = torch.randn(B, C, T, generator=g)
    w1 = torch.randn(C, C, K, generator=g) / (C * K) ** 0.5
    w2 = torch.randn(C, C, K, generator=g) / (C * K) ** 0.5
    b1, b2 = torch.randn(C, generator=g), torch.randn(C, generator=g)
    lens = np.array([T, max(1, T // 2 + 3), min(T, 17)], np.int32)
    xm = x.clone()
    for b in range(B):
        xm[b, :, lens[b]:] = 0
    ref = torch.zeros_like(x)
    for b in range(B):       # each item alone at its own length: exactly the semantics of the batched kernels
        xb = xm[b:b + 1, :, :lens[b]]
        t = F.conv1d(F.leaky_relu(xb, 0.1), w1, b1, dilation=d, padding=(K * d - d) // 2)
        t = F.conv1d(F.leaky_relu(t, 0.1), w2, b2, padding=(K - 1) // 2)
        ref[b, :, :lens[b]] = t + xb
    fused = ctx.resblock_pair(xm.numpy(), w1.numpy(), b1.numpy(), w2.numpy(), b2.numpy(), dil=d, lens=lens)
    two = ctx.resblock_pair(xm.numpy(), w1.numpy(), b1.numpy(), w2.numpy(), b2.numpy(), dil=d, lens=lens, fused=False)
    assert np.isfinite(fused).all()
    e = rms(fused - ref.numpy()) / rms(ref.numpy())
    print(f"C={C} k={K} d={d} T={T}: fused vs torch rel err {e:.2e}; equal to two launches: {np.array_equal(fused, two)}")
    assert e < 2e-6
    assert np.array_equal(fused, two)


def test_activation_beyond_fp16_range_falls_back_to_exact_fp32(ctx):
    """the split-fp16 kernels hold activations as fp16 hi/lo halves -- |x| >= 65504 would
    become inf -> NaN.  The kernels flag such an input, the entry point repeats the call on the exact-fp32
    kernels and counts it: the result is the fp32 answer."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 64, 2000, generator=g)
    x[0, 3, 777] = 1.0e5
    x[0, 40, 12] = -3.0e6
    w = torch.randn(64, 64, 7, generator=g) / (64 * 7) ** 0.5
    b = torch.randn(64, generator=g)
    ref = F.conv1d(x, w, b, padding=3).numpy()
    n0 = ctx.fp32_reruns()
    got = ctx.conv1d(x.numpy(), w.numpy(), b.numpy(), pad_left=3)
    assert ctx.fp32_reruns() == n0 + 1
    assert np.isfinite(got).all()
    assert rms(got - ref) / rms(ref) < 2e-6
    # the fused ResBlock step has the same guard, on its input and on the tile it keeps in LDS
    w2 = torch.randn(64, 64, 7, generator=g) / (64 * 7) ** 0.5
    x2 = torch.randn(1, 64, 1500, generator=g)
    x2[0, 5, 100] = 2.0e5
    t = F.conv1d(F.leaky_relu(x2, 0.1), w, b, padding=3)
    ref2 = (F.conv1d(F.leaky_relu(t, 0.1), w2, b, padding=3) + x2).numpy()
    got2 = ctx.resblock_pair(x2.numpy(), w.numpy(), b.numpy(), w2.numpy(), b.numpy())
    assert ctx.fp32_reruns() == n0 + 2 and np.isfinite(got2).all()
    assert rms(got2 - ref2) / rms(ref2) < 2e-6
    # in-range data never takes the detour
    ctx.conv1d(torch.randn(1, 64, 500, generator=g).numpy(), w.numpy(), b.numpy(), pad_left=3)
    assert ctx.fp32_reruns() == n0 + 2


@pytest.mark.parametrize("shape", [(512, 512, 9, 606, 1), (256, 256, 9, 2020, 1), (192, 768, 3, 3198, 1), (512, 512, 3, 3199, 2),
                                   (768, 192, 1, 1599, 1)])
def test_split_k_shapes_in_a_batch_equal_their_single_runs(ctx, shape):
    """The split-K factor is decided for ONE batch item, so a batch of 8 splits exactly as the single run does and every
    item must equal its single run bit for bit -- on shapes where the single run really splits (deep U-Net levels,
    TextEncoder FFN, a stride-2 extractor layer, a k = 1 Linear) -- and torch within fp32 rounding.  (Round 4 also built a
    "virtual" split -- the ranges walked inside one workgroup, no slabs, no finish launch -- against this test: bit-identical,
    not faster, not kept.)"""
    import torch
    import torch.nn.functional as F
    cin, cout, k, T, stride = shape
    g = torch.Generator().manual_seed(3)
    x = torch.randn(8, cin, T, generator=g)
    w = torch.randn(cout, cin, k, generator=g) / (cin * k) ** 0.5
    b = torch.randn(cout, generator=g)
    pad = (k - 1) // 2 if stride == 1 else 0
    ref = F.conv1d(x, w, b, stride=stride, padding=pad).numpy()
    batch = ctx.conv1d(x.numpy(), w.numpy(), b.numpy(), stride=stride, pad_left=pad, Tout=ref.shape[2])
    assert rms(batch - ref) / rms(ref) < 2e-6
    for i in (0, 3, 7):
        alone = ctx.conv1d(x[i:i + 1].numpy(), w.numpy(), b.numpy(), stride=stride, pad_left=pad, Tout=ref.shape[2])
        assert np.array_equal(alone[0], batch[i]), (shape, i)


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(1, 16, 16, 404, 128), (2, 32, 32, 320, 64), (1, 32, 16, 300, 128), (3, 16, 16, 170, 128)])
def test_streaming_3x3_kernel_equals_the_tiled_one(ctx, B, Cin, Cout, H, W):
    """conv3_thin.hip (shallow U-Net levels: C = 16 / 32, > 20 000 positions) against torch fp32, and bit-identical to the
    conv_h3 tile it replaces (forced through the override): same k-order, same epilogue; the op NaN-checks the pad columns."""
    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    bias = torch.randn(Cout, generator=g)
    res = torch.randn(B, Cout, H, W, generator=g)
    ref = F.relu(F.conv2d(x, w, bias, padding=1)) + res
    ctx.conv_profile_begin()
    got = ctx.conv2d3x3(x.numpy(), w.numpy(), bias.numpy(), res=res.numpy(), act=2)
    names = [r["tile"] for r in ctx.conv_profile_end()]
    assert any(n.startswith("conv3_thin") for n in names), names
    try:
        ctx.conv_override(103, -1, -1)
        tiled = ctx.conv2d3x3(x.numpy(), w.numpy(), bias.numpy(), res=res.numpy(), act=2)
    finally:
        ctx.conv_override(-1, -1, -1)
    e = rms(got - ref.numpy()) / rms(ref.numpy())
    print(f"B={B} {Cin}->{Cout} {H}x{W}: vs torch {e:.2e}; equal to the tiled kernel: {np.array_equal(got, tiled)}")
    assert e < 2e-6
    assert np.array_equal(got, tiled)


@pytest.mark.parametrize("B,C,H,W,ragged", [(1, 16, 404, 128, False), (3, 16, 200, 128, True), (2, 32, 320, 64, True),
                                            (1, 32, 320, 64, False)])
def test_convblockres_through_the_models_block_path(ctx, B, C, H, W, ragged):
    """One ConvBlockRes (RMVPE.py:140-175) through the F0 model's own block path -- the first conv hands its output over in
    split fp16 form, the second adds the residual; per-item row counts as in a ragged micro-batch -- against torch fp32 per
    item at its own height, and bit-identical with the streaming kernel (default) and the tiled one (forced)."""
    g = torch.Generator().manual_seed(B * 77 + C)
    x = torch.randn(B, C, H, W, generator=g)
    w1 = torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5
    w2 = torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5
    b1, b2 = torch.randn(C, generator=g), torch.randn(C, generator=g)
    rows = np.array([H, H // 2 + 3, 5][:B], np.int32) if ragged else None
    ref = torch.zeros(B, C, H, W)
    for b in range(B):
        hb = int(rows[b]) if ragged else H
        xb = x[b:b + 1, :, :hb]
        t = F.relu(F.conv2d(xb, w1, b1, padding=1))
        ref[b, :, :hb] = F.relu(F.conv2d(t, w2, b2, padding=1)) + xb
    ctx.conv_profile_begin()
    got = ctx.convblock2d(x.numpy(), w1.numpy(), b1.numpy(), w2.numpy(), b2.numpy(), rows=rows)
    names = [r["tile"] for r in ctx.conv_profile_end()]
    assert sum(n.startswith("conv3_thin") for n in names) >= 1, names
    try:
        ctx.conv_override(103, -1, -1)
        tiled = ctx.convblock2d(x.numpy(), w1.numpy(), b1.numpy(), w2.numpy(), b2.numpy(), rows=rows)
    finally:
        ctx.conv_override(-1, -1, -1)
    assert np.isfinite(got).all()
    e = rms(got - ref.numpy()) / rms(ref.numpy())
    print(f"B={B} C={C} {H}x{W} ragged={ragged}: vs torch {e:.2e}; equal to the tiled kernels: {np.array_equal(got, tiled)}")
    assert e < 2e-6
    assert np.array_equal(got, tiled)


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(1, 32, 16, 404, 128), (2, 64, 32, 330, 64), (1, 1, 16, 300, 128)])
def test_convblockres_with_the_1x1_shortcut(ctx, B, Cin, Cout, H, W):
    """The first block of a U-Net level changes the channel count: ConvBlockRes adds ``shortcut(x)`` (a 1 x 1 conv,
    RMVPE.py:158-175) instead of x.  32 -> 16 is the decoder's level 0 (the streaming kernel's Cin = 32 / Cout = 16 form feeds
    the split hand-off), 1 -> 16 the encoder's first block (the Cin = 1 FIR kernel)."""
    g = torch.Generator().manual_seed(B * 31 + Cin)
    x = torch.randn(B, Cin, H, W, generator=g)
    w1 = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    w2 = torch.randn(Cout, Cout, 3, 3, generator=g) / (Cout * 9) ** 0.5
    wsc = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    b1, b2, bsc = (torch.randn(Cout, generator=g) for _ in range(3))
    t = F.relu(F.conv2d(x, w1, b1, padding=1))
    ref = F.relu(F.conv2d(t, w2, b2, padding=1)) + F.conv2d(x, wsc, bsc)
    got = ctx.convblock2d(x.numpy(), w1.numpy(), b1.numpy(), w2.numpy(), b2.numpy(), wsc=wsc.numpy().reshape(Cout, Cin),
                          bsc=bsc.numpy())
    # (no bit comparison with a forced tile here: the override would also take the 1 x 1 shortcut off its split-fp16 tile)
    e = rms(got - ref.numpy()) / rms(ref.numpy())
    print(f"B={B} {Cin}->{Cout} {H}x{W}: vs torch {e:.2e}")
    assert np.isfinite(got).all() and e < 2e-6
