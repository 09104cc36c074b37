"""Deterministic synthetic checkpoints and clips in the *real* checkpoint layouts.

There are no model weights in the build container and no network, so parity and
benchmark runs use weights produced by a counter-based generator: every tensor is
drawn from ``numpy.random.Philox(key = crc32(name) ^ seed)``, which makes any
single tensor reproducible on its own (no stream ordering between tensors) and
lets 0.85 GB of weights be regenerated on the GPU box instead of being committed.

Layouts follow SURVEY.md Appendix B (names/shapes dumped from the reference):
  * voice model  : rvc/infer/infer.py:78-105 (``cpt = {"weight", "config", "f0", "version"}``)
  * rmvpe.pt     : rvc/lib/predictors/RMVPE.py:449-456 (bare state_dict of E2E(4,1,(2,2)))
  * hubert_base  : fairseq 0.12.2 HubertModel naming (not vendored in the reference)
  * fcpe.pt      : rvc/lib/predictors/FCPE.py:708-736 (``{"config": {...}, "model": state_dict of FCPE}``)
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, List, Tuple

import numpy as np

# canonical RVC v2 configs (read from the checkpoint by the reference, infer.py:92-97)
SYNTH_CFG_48K = [1025, 32, 192, 192, 768, 2, 6, 3, 0, "1", [3, 7, 11],
                 [[1, 3, 5], [1, 3, 5], [1, 3, 5]], [12, 10, 2, 2], 512,
                 [24, 20, 4, 4], 109, 256, 48000]
SYNTH_CFG_40K = [1025, 32, 192, 192, 768, 2, 6, 3, 0, "1", [3, 7, 11],
                 [[1, 3, 5], [1, 3, 5], [1, 3, 5]], [10, 10, 2, 2], 512,
                 [16, 16, 4, 4], 109, 256, 40000]
# the 32 k geometry of RVC v2 (upstream configs/v2/32000.json): rates 10 x 8 x 2 x 2, kernels 20 / 16 / 4 / 4
SYNTH_CFG_32K = [513, 32, 192, 192, 768, 2, 6, 3, 0, "1", [3, 7, 11],
                 [[1, 3, 5], [1, 3, 5], [1, 3, 5]], [10, 8, 2, 2], 512,
                 [20, 16, 4, 4], 109, 256, 32000]
# reduced config for fast unit tests: channel counts that are NOT multiples of the
# 32-wide MFMA tile on purpose (exercises the padding / guard paths)
SYNTH_CFG_TINY = [1025, 32, 48, 48, 96, 2, 2, 3, 0, "1", [3, 7, 11],
                  [[1, 3, 5], [1, 3, 5], [1, 3, 5]], [4, 3, 2, 2], 80,
                  [8, 7, 4, 4], 5, 24, 4800]

RMVPE_CFG_FULL = dict(n_blocks=4, n_gru=1, en_de_layers=5, inter_layers=4,
                      in_channels=1, en_out_channels=16)
RMVPE_CFG_TINY = dict(n_blocks=1, n_gru=1, en_de_layers=2, inter_layers=1,
                      in_channels=1, en_out_channels=4)

# fcpe.pt's own config block (FCPE.py:715-733 reads these); heads = 8 x dim_head = 64 are SelfAttention defaults
FCPE_CFG_FULL = dict(n_layers=6, n_chans=512, input_channel=128, out_dims=360)
FCPE_CFG_TINY = dict(n_layers=2, n_chans=64, input_channel=128, out_dims=360)

HUBERT_CFG_BASE = dict(conv_dim=512, conv_kernels=[10, 3, 3, 3, 3, 2, 2],
                       conv_strides=[5, 2, 2, 2, 2, 2, 2], embed_dim=768, ffn_dim=3072,
                       heads=12, layers=12, pos_kernel=128, pos_groups=16, final_dim=256)
HUBERT_CFG_TINY = dict(conv_dim=48, conv_kernels=[10, 3, 3, 3, 3, 2, 2],
                       conv_strides=[5, 2, 2, 2, 2, 2, 2], embed_dim=128, ffn_dim=160,
                       heads=2, layers=2, pos_kernel=128, pos_groups=16, final_dim=32)


# ----------------------------------------------------------------------------
# counter-based per-tensor generator
# ----------------------------------------------------------------------------
def _rng(name: str, seed: int) -> np.random.Generator:
    key = (zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0xFFFFFFFFFFFFFFFF
    return np.random.Generator(np.random.Philox(key=key))


# shape-only mode: ranks that receive rank 0's folded weights by broadcast (dist.broadcast_weights) only need tensors of
# the right SHAPES to build the same region layout -- zeros, no random numbers drawn (bench.py, ranks != 0)
_SHAPES_ONLY = False


class shapes_only:
    """with S.shapes_only(): S.hubert_state(...) -> same keys and shapes, all zeros, at a fraction of the host time."""
    def __enter__(self):
        global _SHAPES_ONLY
        self._prev, _SHAPES_ONLY = _SHAPES_ONLY, True

    def __exit__(self, *exc):
        global _SHAPES_ONLY
        _SHAPES_ONLY = self._prev


def _normal(name: str, shape, std: float, seed: int, mean: float = 0.0) -> np.ndarray:
    if _SHAPES_ONLY:
        return np.zeros(tuple(shape), np.float32)
    x = _rng(name, seed).standard_normal(size=tuple(shape), dtype=np.float32)
    return (x * np.float32(std) + np.float32(mean)).astype(np.float32)


def _smooth_axis0(x: np.ndarray, width: float) -> np.ndarray:
    """Gaussian low-pass along axis 0 (keeps overall scale)."""
    n = x.shape[0]
    r = int(3 * width)
    k = np.exp(-0.5 * (np.arange(-r, r + 1) / width) ** 2)
    k /= np.sqrt((k ** 2).sum())
    pad = np.pad(x, ((r, r),) + ((0, 0),) * (x.ndim - 1), mode="wrap")
    out = np.zeros_like(x)
    for i, kv in enumerate(k):
        out += np.float32(kv) * pad[i:i + n]
    return out.astype(np.float32)


class _Table:
    def __init__(self, seed: int):
        self.seed = seed
        self.t: Dict[str, np.ndarray] = {}

    def normal(self, name, shape, std, mean=0.0):
        self.t[name] = _normal(name, shape, std, self.seed, mean)
        return self.t[name]

    def conv(self, prefix, shape, gain=1.0, bias=True, fan_in=None, bias_std=0.05, nbias=None):
        fi = fan_in if fan_in is not None else int(np.prod(shape[1:]))
        self.normal(prefix + ".weight", shape, gain / math.sqrt(max(fi, 1)))
        if bias:
            self.normal(prefix + ".bias", (nbias if nbias is not None else shape[0],), bias_std)

    def wn_conv(self, prefix, shape, gain=1.0, fan_in=None, bias_std=0.05, nbias=None):
        """weight-normalised conv in the parametrized layout (original0 = g, original1 = v)."""
        fi = fan_in if fan_in is not None else int(np.prod(shape[1:]))
        v = self.normal(prefix + ".parametrizations.weight.original1", shape, 1.0)
        vn = np.sqrt((v.reshape(shape[0], -1).astype(np.float64) ** 2).sum(1))
        target = gain / math.sqrt(max(fi, 1)) * math.sqrt(int(np.prod(shape[1:])))
        jitter = 1.0 + 0.1 * _rng(prefix + ".g", self.seed).standard_normal(shape[0])
        g = (target * jitter * vn / np.maximum(vn, 1e-12)).astype(np.float32)
        self.t[prefix + ".parametrizations.weight.original0"] = g.reshape(
            (shape[0],) + (1,) * (len(shape) - 1))
        self.normal(prefix + ".bias", (nbias if nbias is not None else shape[0],), bias_std)


# ----------------------------------------------------------------------------
# Synthesizer (enc_q removed, as after infer.py:99)
# ----------------------------------------------------------------------------
# Real voice models are not O(1) everywhere either: weight-norm g vectors of trained HiFi-GAN decoders spread over two
# orders of magnitude and single channels carry activations of a few hundred.  ``outliers=True`` plants that in the NSF
# decoder (78 % of the path's FLOPs, all on the split-fp16 kernels), function-preservingly -- leaky_relu is positively
# homogeneous, so a producer channel scaled by G is undone in its consumer's input column:
#   * dec.conv_pre / dec.cond rows DEC_OUTLIER_PRE x G, the matching weight-norm g entries of dec.ups.0 (per INPUT channel:
#     ConvTranspose1d, dim = 0) / G -> g of ups.0 spreads over 1 : G, its input carries channels of a few hundred;
#   * one ResBlock1 step per (stage, kernel): rows DEC_OUTLIER_UNITS of convs1[m] (g and bias) x G, the matching input
#     columns of convs2[m] / G (re-parametrised: v' = the new effective weight, g' = its row norms) -> the c1 -> c2 hand-off
#     of the fused step (an LDS tile of split halves) holds values of 50 ... 500, c2 rows mix weights 1 : G apart.
DEC_OUTLIER_GAIN, DEC_OUTLIER_UNITS, DEC_OUTLIER_PRE = 150.0, 3, 2      # not a power of two: the split halves change


def _wn_effective(t, prefix):
    """effective weight g v / ||v|| (norm over all dims but 0) of a parametrized conv, float64"""
    g = t[prefix + ".parametrizations.weight.original0"].astype(np.float64)
    v = t[prefix + ".parametrizations.weight.original1"].astype(np.float64)
    n = np.sqrt((v.reshape(v.shape[0], -1) ** 2).sum(1)).reshape(g.shape)
    return g * v / n


def _wn_set(t, prefix, w):
    """re-parametrise: v = w, g = ||w|| per dim-0 slice (so g v / ||v|| = w exactly up to fp32 rounding)"""
    w = np.asarray(w, np.float64)
    n = np.sqrt((w.reshape(w.shape[0], -1) ** 2).sum(1))
    t[prefix + ".parametrizations.weight.original1"] = w.astype(np.float32)
    t[prefix + ".parametrizations.weight.original0"] = n.reshape((w.shape[0],) + (1,) * (w.ndim - 1)).astype(np.float32)


def _plant_decoder_outliers(t: Dict[str, np.ndarray], cfg: List, seed: int) -> None:
    (_, _, inter, hidden, filt, n_heads, n_layers, ksz, _, _, rks, rds, ups, up_init, upks, spk, gin, _) = cfg
    G = DEC_OUTLIER_GAIN
    # conv_pre (+ cond) -> lrelu -> ups.0
    rows = _rng("outlier.dec.pre", seed).choice(up_init, min(DEC_OUTLIER_PRE, up_init), replace=False)
    for n in ("dec.conv_pre", "dec.cond"):
        w, b = np.array(t[n + ".weight"]), np.array(t[n + ".bias"])
        w[rows] *= np.float32(G)
        b[rows] *= np.float32(G)
        t[n + ".weight"], t[n + ".bias"] = w, b
    g0 = np.array(t["dec.ups.0.parametrizations.weight.original0"])
    g0[rows] /= np.float32(G)
    t["dec.ups.0.parametrizations.weight.original0"] = g0
    # one dilation step m of every ResBlock1: convs1[m] rows x G, convs2[m] columns / G
    for i in range(len(ups)):
        co = up_init // (2 ** (i + 1))
        for j in range(len(rks)):
            blk = f"dec.resblocks.{i * len(rks) + j}"
            m = (i + j) % len(rds[j])
            units = _rng(f"outlier.{blk}", seed).choice(co, min(DEC_OUTLIER_UNITS, co), replace=False)
            p1, p2 = f"{blk}.convs1.{m}", f"{blk}.convs2.{m}"
            g1 = np.array(t[p1 + ".parametrizations.weight.original0"])
            b1 = np.array(t[p1 + ".bias"])
            g1[units] *= np.float32(G)
            b1[units] *= np.float32(G)
            t[p1 + ".parametrizations.weight.original0"], t[p1 + ".bias"] = g1, b1
            w2 = _wn_effective(t, p2)
            w2[:, units] /= G
            _wn_set(t, p2, w2)


def synth_state(cfg: List, seed: int = 0, input_dim: int = 768, outliers: bool = False) -> Dict[str, np.ndarray]:
    (_, _, inter, hidden, filt, n_heads, n_layers, ksz, _, _, rks, rds, ups, up_init,
     upks, spk, gin, _) = cfg
    T = _Table(seed)
    kch = hidden // n_heads
    # enc_p (encoders.py:76-126)
    T.conv("enc_p.emb_phone", (hidden, input_dim), gain=0.1)
    T.t["enc_p.emb_pitch.weight"] = _smooth_axis0(
        _normal("enc_p.emb_pitch.weight", (256, hidden), 0.1, seed), 6.0)
    for i in range(n_layers):
        a = f"enc_p.encoder.attn_layers.{i}"
        T.normal(a + ".emb_rel_k", (1, 21, kch), kch ** -0.5)
        T.normal(a + ".emb_rel_v", (1, 21, kch), kch ** -0.5)
        for n in ("conv_q", "conv_k", "conv_v", "conv_o"):
            T.conv(f"{a}.{n}", (hidden, hidden, 1), gain=1.0)
        for j in (1, 2):
            T.normal(f"enc_p.encoder.norm_layers_{j}.{i}.gamma", (hidden,), 0.1, 1.0)
            T.normal(f"enc_p.encoder.norm_layers_{j}.{i}.beta", (hidden,), 0.05)
        T.conv(f"enc_p.encoder.ffn_layers.{i}.conv_1", (filt, hidden, ksz), gain=1.0)
        T.conv(f"enc_p.encoder.ffn_layers.{i}.conv_2", (hidden, filt, ksz), gain=1.0)
    T.conv("enc_p.proj", (2 * inter, hidden, 1), gain=0.5)
    # dec (nsf.py:43-118)
    T.t["dec.m_source.l_linear.weight"] = _normal("dec.m_source.l_linear.weight", (1, 1), 0.1, seed, 1.0)
    T.normal("dec.m_source.l_linear.bias", (1,), 0.01)
    T.conv("dec.conv_pre", (up_init, inter, 7), gain=1.0)
    T.conv("dec.cond", (up_init, gin, 1), gain=0.3)
    ch = up_init
    for i, (u, k) in enumerate(zip(ups, upks)):
        co = up_init // (2 ** (i + 1))
        # ConvTranspose1d weight is (Cin, Cout, k); g is per *input* channel (dim=0)
        T.wn_conv(f"dec.ups.{i}", (ch, co, k), gain=1.0, fan_in=max(1, (ch * k) // u), nbias=co)
        sf0 = int(np.prod(ups[i + 1:])) if i + 1 < len(ups) else 1
        nk = sf0 * 2 if sf0 > 1 else 1
        T.conv(f"dec.noise_convs.{i}", (co, 1, nk), gain=1.0)
        for j, (rk, rd) in enumerate(zip(rks, rds)):
            for m in range(len(rd)):
                T.wn_conv(f"dec.resblocks.{i * len(rks) + j}.convs1.{m}", (co, co, rk), gain=0.6)
                T.wn_conv(f"dec.resblocks.{i * len(rks) + j}.convs2.{m}", (co, co, rk), gain=0.6)
        ch = co
    T.conv("dec.conv_post", (1, ch, 7), gain=1.0, bias=False)
    # flow (residuals.py:109-232): 4 coupling layers at even indices
    half = inter // 2
    for f in (0, 2, 4, 6):
        p = f"flow.flows.{f}"
        T.conv(p + ".pre", (hidden, half, 1), gain=1.0)
        for i in range(3):
            T.wn_conv(f"{p}.enc.in_layers.{i}", (2 * hidden, hidden, 5), gain=1.0)
            rs = hidden if i == 2 else 2 * hidden
            T.wn_conv(f"{p}.enc.res_skip_layers.{i}", (rs, hidden, 1), gain=0.7)
        T.wn_conv(p + ".enc.cond_layer", (2 * hidden * 3, gin, 1), gain=0.5)
        T.conv(p + ".post", (half, hidden, 1), gain=0.3)
    T.normal("emb_g.weight", (spk, gin), 0.3)
    if outliers and not _SHAPES_ONLY:
        _plant_decoder_outliers(T.t, cfg, seed)
    return T.t


def synth_checkpoint(cfg: List, seed: int = 0, version: str = "v2") -> dict:
    """The dict ``torch.load(model.pth)`` returns for a voice model (numpy tensors)."""
    return {"weight": synth_state(cfg, seed), "config": list(cfg), "f0": 1,
            "version": version, "sr": f"{cfg[-1] // 1000}k", "info": "synthetic"}


# ----------------------------------------------------------------------------
# RMVPE E2E (RMVPE.py:340-376)
# ----------------------------------------------------------------------------
def _bn(T: _Table, prefix: str, c: int):
    T.normal(prefix + ".weight", (c,), 0.1, 1.0)
    T.normal(prefix + ".bias", (c,), 0.05)
    T.normal(prefix + ".running_mean", (c,), 0.05)
    T.t[prefix + ".running_var"] = (1.0 + 0.1 * np.abs(
        _normal(prefix + ".running_var", (c,), 1.0, T.seed))).astype(np.float32)
    T.t[prefix + ".num_batches_tracked"] = np.array(1000, dtype=np.int64)


def _conv_block_res(T: _Table, prefix: str, cin: int, cout: int):
    g = 1.0
    T.conv(prefix + ".conv.0", (cout, cin, 3, 3), gain=g, bias=False)
    _bn(T, prefix + ".conv.1", cout)
    T.conv(prefix + ".conv.3", (cout, cout, 3, 3), gain=g, bias=False)
    _bn(T, prefix + ".conv.4", cout)
    if cin != cout:
        T.conv(prefix + ".shortcut", (cout, cin, 1, 1), gain=0.7)


# U-Net of the F0 model: a BatchNorm scale of G on a few channels of a block's first conv (ReLU is positively homogeneous),
# undone in the input columns of the block's second conv -- the pre-split hand-off between the two 3 x 3 convs of a
# ConvBlockRes then carries values of a few hundred.  One block per encoder level, one intermediate, one decoder block.
UNET_OUTLIER_GAIN, UNET_OUTLIER_UNITS = 150.0, 2


def _plant_unet_outliers(t: Dict[str, np.ndarray], cfg: dict, seed: int) -> None:
    nb, nenc, nint = cfg["n_blocks"], cfg["en_de_layers"], cfg["inter_layers"]
    G = np.float32(UNET_OUTLIER_GAIN)
    blocks = [f"unet.encoder.layers.{l}.conv.{l % nb}" for l in range(nenc)]
    blocks += [f"unet.intermediate.layers.{nint - 1}.conv.{nb - 1}", f"unet.decoder.layers.{nenc - 1}.conv2.{nb - 1}",
               "unet.decoder.layers.0.conv2.0"]
    for blk in blocks:
        c = t[blk + ".conv.1.weight"].shape[0]
        units = _rng(f"outlier.{blk}", seed).choice(c, min(UNET_OUTLIER_UNITS, c), replace=False)
        gam, bet = np.array(t[blk + ".conv.1.weight"]), np.array(t[blk + ".conv.1.bias"])
        mean = np.array(t[blk + ".conv.1.running_mean"])
        gam[units] *= G
        bet[units] *= G           # y = gam (x - mean) / sd + bet: both terms x G
        t[blk + ".conv.1.weight"], t[blk + ".conv.1.bias"], t[blk + ".conv.1.running_mean"] = gam, bet, mean
        w = np.array(t[blk + ".conv.3.weight"])
        w[:, units] /= G
        t[blk + ".conv.3.weight"] = w


def rmvpe_state(cfg: dict = None, seed: int = 0, outliers: bool = False) -> Dict[str, np.ndarray]:
    cfg = cfg or RMVPE_CFG_FULL
    nb, nenc, nint, c0 = cfg["n_blocks"], cfg["en_de_layers"], cfg["inter_layers"], cfg["en_out_channels"]
    T = _Table(seed + 17)
    _bn(T, "unet.encoder.bn", cfg["in_channels"])
    cin, cout = cfg["in_channels"], c0
    for l in range(nenc):
        for b in range(nb):
            _conv_block_res(T, f"unet.encoder.layers.{l}.conv.{b}", cin if b == 0 else cout, cout)
        cin, cout = cout, cout * 2
    # intermediate: first layer cin -> cout (=2*cin)
    for l in range(nint):
        for b in range(nb):
            _conv_block_res(T, f"unet.intermediate.layers.{l}.conv.{b}",
                            cin if (l == 0 and b == 0) else cout, cout)
    dch = cout
    for l in range(nenc):
        oc = dch // 2
        p = f"unet.decoder.layers.{l}"
        # ConvTranspose2d weight (Cin, Cout, 3, 3); effective fan-in ~ Cin*9/4
        T.conv(p + ".conv1.0", (dch, oc, 3, 3), gain=1.4, bias=False, fan_in=max(1, dch * 9 // 4))
        _bn(T, p + ".conv1.1", oc)
        for b in range(nb):
            _conv_block_res(T, f"{p}.conv2.{b}", oc * 2 if b == 0 else oc, oc)
        dch = oc
    T.conv("cnn", (3, c0, 3, 3), gain=1.0)
    H = 256
    for sfx in ("", "_reverse"):
        T.normal(f"fc.0.gru.weight_ih_l0{sfx}", (3 * H, 384), 1.0 / math.sqrt(384))
        T.normal(f"fc.0.gru.weight_hh_l0{sfx}", (3 * H, H), 1.0 / math.sqrt(H))
        T.normal(f"fc.0.gru.bias_ih_l0{sfx}", (3 * H,), 0.05)
        T.normal(f"fc.0.gru.bias_hh_l0{sfx}", (3 * H,), 0.05)
    # smooth across the 360 pitch bins (correlation ~1.6 bins, like the label blur of a trained
    # model): peaks are narrower than the +-4-bin averaging window of to_local_average_cents, so an
    # argmax flip between adjacent bins barely moves the decoded f0
    w = _normal("fc.1.weight", (360, 2 * H), 1.0, seed + 17)
    T.t["fc.1.weight"] = (_smooth_axis0(w, 1.6) * np.float32(4.0 / math.sqrt(2 * H))).astype(np.float32)
    T.t["fc.1.bias"] = _smooth_axis0(_normal("fc.1.bias", (360,), 1.0, seed + 17), 9.0) * np.float32(0.3) - np.float32(10.5 if c0 >= 16 else 11.5)
    if outliers and not _SHAPES_ONLY:
        _plant_unet_outliers(T.t, cfg, seed)
    return T.t


# ----------------------------------------------------------------------------
# HuBERT base, fairseq 0.12.2 key names
# ----------------------------------------------------------------------------
# Real HuBERT / wav2vec2-base checkpoints are not O(1) everywhere: a handful of FFN units and attention value heads carry
# "massive activations" -- hundreds to a few thousand against a bulk of order one (the published outlier-dimension
# analyses of BERT-family and wav2vec2 / HuBERT encoders).  ``outliers=True`` plants that shape in the synthetic model,
# function-preservingly (each scaled producer row is undone in its consumer's column), so the bulk statistics and every
# downstream activation stay where the goldens expect them:
#   * layers OUTLIER_FFN_LAYERS: OUTLIER_FFN_UNITS units of fc1 (weight row + bias) x 200, the matching fc2 columns / 200
#     -> GELU outputs of several hundred to ~2000 enter fc2 (inside the split kernels' range, 6e4: must NOT be pinned);
#   * layers OUTLIER_V_LAYERS: the v_proj rows (+ bias) of head OUTLIER_V_HEAD x 400, the out_proj columns of that head
#     / 400 -> V values of 400 ... 1500, beyond the attention kernel's fp16 range for K / V (255): exactly these layers'
#     attention calls must be pinned to the exact-fp32 kernel by the range guard, once, and stay pinned.
OUTLIER_FFN_LAYERS, OUTLIER_FFN_UNITS, OUTLIER_FFN_GAIN = (2, 6, 10), 4, 200.0
OUTLIER_V_LAYERS, OUTLIER_V_HEAD, OUTLIER_V_GAIN = (4, 8), 3, 400.0


def _plant_outliers(t: Dict[str, np.ndarray], cfg: dict, seed: int) -> None:
    E, Fd, L, H = cfg["embed_dim"], cfg["ffn_dim"], cfg["layers"], cfg["heads"]
    d = E // H
    for l in OUTLIER_FFN_LAYERS:
        if l >= L:
            continue
        p = f"encoder.layers.{l}"
        units = _rng(f"outlier.ffn.{l}", seed).choice(Fd, OUTLIER_FFN_UNITS, replace=False)
        w1, b1, w2 = (np.array(t[p + ".fc1.weight"]), np.array(t[p + ".fc1.bias"]), np.array(t[p + ".fc2.weight"]))
        g = np.float32(OUTLIER_FFN_GAIN)
        w1[units] *= g
        # a positive pre-activation keeps the unit in GELU's linear range (gelu(g x) = g gelu(x) only holds there): the
        # planted unit then IS a large activation on every frame, as the published outlier units are
        b1[units] = np.abs(b1[units]) * g + g
        w2[:, units] /= g
        t[p + ".fc1.weight"], t[p + ".fc1.bias"], t[p + ".fc2.weight"] = w1, b1, w2
    for l in OUTLIER_V_LAYERS:
        if l >= L:
            continue
        p = f"encoder.layers.{l}.self_attn"
        rows = slice((OUTLIER_V_HEAD % H) * d, (OUTLIER_V_HEAD % H + 1) * d)
        wv, bv, wo = np.array(t[p + ".v_proj.weight"]), np.array(t[p + ".v_proj.bias"]), np.array(t[p + ".out_proj.weight"])
        g = np.float32(OUTLIER_V_GAIN)
        wv[rows] *= g
        bv[rows] *= g
        wo[:, rows] /= g
        t[p + ".v_proj.weight"], t[p + ".v_proj.bias"], t[p + ".out_proj.weight"] = wv, bv, wo


def hubert_state(cfg: dict = None, seed: int = 0, outliers: bool = False) -> Dict[str, np.ndarray]:
    cfg = cfg or HUBERT_CFG_BASE
    T = _Table(seed + 31)
    C, E, Fd = cfg["conv_dim"], cfg["embed_dim"], cfg["ffn_dim"]
    cin = 1
    for i, k in enumerate(cfg["conv_kernels"]):
        T.conv(f"feature_extractor.conv_layers.{i}.0", (C, cin, k), gain=1.6 if i else 1.0, bias=False)
        cin = C
    T.normal("feature_extractor.conv_layers.0.2.weight", (C,), 0.1, 1.0)   # GroupNorm(C, C)
    T.normal("feature_extractor.conv_layers.0.2.bias", (C,), 0.05)
    T.normal("layer_norm.weight", (C,), 0.1, 1.0)
    T.normal("layer_norm.bias", (C,), 0.05)
    T.conv("post_extract_proj", (E, C), gain=1.0)
    # pos_conv: weight_norm(dim=2) -> g has shape (1,1,K), v (E, E/groups, K)
    K, G = cfg["pos_kernel"], cfg["pos_groups"]
    v = T.normal("encoder.pos_conv.0.weight_v", (E, E // G, K), 1.0)
    vn = np.sqrt((v.astype(np.float64) ** 2).sum(axis=(0, 1), keepdims=True))
    target = 0.7 / math.sqrt(E // G * K) * math.sqrt(E * (E // G))
    jitter = 1.0 + 0.1 * _rng("encoder.pos_conv.0.g", seed + 31).standard_normal((1, 1, K))
    T.t["encoder.pos_conv.0.weight_g"] = (target * jitter * vn / np.maximum(vn, 1e-12)).astype(np.float32)
    T.normal("encoder.pos_conv.0.bias", (E,), 0.05)
    T.normal("encoder.layer_norm.weight", (E,), 0.1, 1.0)
    T.normal("encoder.layer_norm.bias", (E,), 0.05)
    for l in range(cfg["layers"]):
        p = f"encoder.layers.{l}"
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            T.conv(f"{p}.self_attn.{n}", (E, E), gain=1.0 if n != "out_proj" else 0.7)
        T.normal(p + ".self_attn_layer_norm.weight", (E,), 0.1, 1.0)
        T.normal(p + ".self_attn_layer_norm.bias", (E,), 0.05)
        T.conv(p + ".fc1", (Fd, E), gain=1.0)
        T.conv(p + ".fc2", (E, Fd), gain=0.7)
        T.normal(p + ".final_layer_norm.weight", (E,), 0.1, 1.0)
        T.normal(p + ".final_layer_norm.bias", (E,), 0.05)
    # present in real checkpoints, unused by the v2 path (pipeline.py:236 uses final_proj for v1 only)
    T.normal("mask_emb", (E,), 0.1)
    T.conv("final_proj", (cfg["final_dim"], E), gain=1.0)
    if outliers and not _SHAPES_ONLY:
        _plant_outliers(T.t, cfg, seed)
    return T.t


# ----------------------------------------------------------------------------
# FCPE (rvc/lib/predictors/FCPE.py:551-627), state_dict names of the reference module
# ----------------------------------------------------------------------------
def _gaussian_orthogonal(name: str, rows: int, cols: int, seed: int) -> np.ndarray:
    """A draw with the structure of gaussian_orthogonal_random_matrix(scaling=0) (FCPE.py:355-381): stacked
    orthogonal blocks, rows rescaled to the norms of Gaussian vectors.  In a checkpoint it is a stored buffer."""
    blocks, r = [], rows
    i = 0
    while r > 0:
        g = _rng(f"{name}.blk{i}", seed).standard_normal((cols, cols))
        q, _ = np.linalg.qr(g)
        blocks.append(q.T[:min(r, cols)])
        r -= cols
        i += 1
    mult = np.linalg.norm(_rng(name + ".mult", seed).standard_normal((rows, cols)), axis=1)
    return (mult[:, None] * np.concatenate(blocks)).astype(np.float32)


def fcpe_state(cfg: dict = None, seed: int = 0) -> Dict[str, np.ndarray]:
    cfg = cfg or FCPE_CFG_FULL
    C, L, cin, nout = cfg["n_chans"], cfg["n_layers"], cfg["input_channel"], cfg["out_dims"]
    inner, dh = 512, 64                                  # heads * dim_head, FCPE.py:445-446
    nfeat = int(dh * math.log(dh))                       # 266, FCPE.py:435
    T = _Table(seed + 43)
    T.conv("stack.0", (C, cin, 3), gain=0.25)            # log-mel inputs are O(5)
    T.normal("stack.1.weight", (C,), 0.1, mean=1.0)
    T.normal("stack.1.bias", (C,), 0.05)
    T.conv("stack.3", (C, C, 3), gain=1.0)
    for i in range(L):
        p = f"decoder._layers.{i}"
        T.normal(p + ".norm.weight", (C,), 0.1, mean=1.0)
        T.normal(p + ".norm.bias", (C,), 0.05)
        for nm in ("to_q", "to_k", "to_v"):
            T.conv(f"{p}.attn.{nm}", (inner, C), gain=1.0)
        T.conv(f"{p}.attn.to_out", (C, inner), gain=0.7)
        T.t[f"{p}.attn.fast_attention.projection_matrix"] = _gaussian_orthogonal(
            f"{p}.attn.fast_attention.projection_matrix", nfeat, dh, seed + 43)
        T.normal(p + ".conformer.net.0.weight", (C,), 0.1, mean=1.0)
        T.normal(p + ".conformer.net.0.bias", (C,), 0.05)
        T.conv(p + ".conformer.net.2", (4 * C, C, 1), gain=1.0)
        T.conv(p + ".conformer.net.4.conv", (2 * C, 1, 31), gain=1.5)
        T.conv(p + ".conformer.net.6", (C, 2 * C, 1), gain=1.0)
    T.normal("norm.weight", (C,), 0.1, mean=1.0)
    T.normal("norm.bias", (C,), 0.05)
    # dense_out = weight_norm(Linear(C, 360)): rows smooth across the pitch bins (as a trained salience head is),
    # gains and bias placed so that the per-frame maximum straddles the 0.03 confidence threshold of VC.get_f0
    v = _smooth_axis0(_normal("dense_out.v", (nout, C), 1.0, seed + 43), 1.6)
    T.t["dense_out.parametrizations.weight.original1"] = v
    T.t["dense_out.parametrizations.weight.original0"] = (
        4.0 * (1.0 + 0.1 * _rng("dense_out.g", seed + 43).standard_normal((nout, 1)))).astype(np.float32)
    T.t["dense_out.bias"] = (_smooth_axis0(_normal("dense_out.bias", (nout,), 1.0, seed + 43), 9.0)
                             * np.float32(0.3) - np.float32(FCPE_BIAS)).astype(np.float32)
    lo, hi = np.float32(1200.0) * np.log2(np.float32(32.70) / np.float32(10.0)), \
        np.float32(1200.0) * np.log2(np.float32(1975.5) / np.float32(10.0))
    T.t["cent_table"] = np.linspace(float(lo), float(hi), nout).astype(np.float32)
    return T.t


FCPE_BIAS = 14.2

# torchcrepe's two capacities (model.Crepe.__init__): filters per layer; kernels 512 then 5 x 64, strides 4 then 1
CREPE_FILTERS = {"full": [1024, 128, 128, 128, 256, 512], "tiny": [128, 16, 16, 16, 32, 64]}


def crepe_state(capacity: str = "full", seed: int = 0) -> Dict[str, np.ndarray]:
    """A state dict with torchcrepe's keys and shapes (conv{i}.weight (Cout, Cin, K, 1), conv{i}_BN.*, classifier.*).
    The classifier rows are smooth across the 360 pitch bins (as a trained salience head is), so the per-frame maxima
    move along the bins instead of jumping: Viterbi paths that mean something."""
    out = CREPE_FILTERS[capacity]
    cin = [1] + out[:-1]
    T = _Table(seed + 71)
    for i, (ci, co) in enumerate(zip(cin, out), 1):
        k = 512 if i == 1 else 64
        T.conv(f"conv{i}", (co, ci, k, 1), gain=1.6, bias_std=0.1)
        T.normal(f"conv{i}_BN.weight", (co,), 0.15, mean=1.0)
        T.normal(f"conv{i}_BN.bias", (co,), 0.1)
        T.normal(f"conv{i}_BN.running_mean", (co,), 0.1, mean=0.35)
        T.t[f"conv{i}_BN.running_var"] = (0.5 + np.abs(_normal(f"conv{i}_BN.running_var", (co,), 0.2, seed + 71))).astype(np.float32)
    nin = 4 * out[-1]
    w = _smooth_axis0(_normal("classifier.w", (360, nin), 1.0, seed + 71), 2.5)
    T.t["classifier.weight"] = (w * np.float32(3.0 / math.sqrt(nin))).astype(np.float32)
    T.t["classifier.bias"] = (_smooth_axis0(_normal("classifier.b", (360,), 1.0, seed + 71), 12.0) * np.float32(0.5)
                              - np.float32(2.0)).astype(np.float32)
    return T.t


def fcpe_checkpoint(cfg: dict = None, seed: int = 0) -> dict:
    """The container FCPEInfer.__init__ reads (FCPE.py:708-736)."""
    cfg = cfg or FCPE_CFG_FULL
    return {
        "config": {
            "model": {"input_channel": cfg["input_channel"], "out_dims": cfg["out_dims"], "n_layers": cfg["n_layers"],
                      "n_chans": cfg["n_chans"], "use_siren": False, "use_full": False, "f0_max": 1975.5,
                      "f0_min": 32.70, "confidence": False},
            "loss": {"loss_mse_scale": 10, "loss_l2_regularization": False, "loss_l2_regularization_scale": 1,
                     "loss_grad1_mse": False, "loss_grad1_mse_scale": 1},
            "mel": {"sampling_rate": 16000, "num_mels": 128, "n_fft": 1024, "win_size": 1024, "hop_size": 160,
                    "fmin": 0, "fmax": 8000},
        },
        "model": fcpe_state(cfg, seed),
    }


# ----------------------------------------------------------------------------
# synthetic clips
# ----------------------------------------------------------------------------
def make_clip(index: int, seconds: float, sr: int = 16000) -> np.ndarray:
    """Harmonic-plus-noise 16 kHz mono clip, seed 1000+index, float32 in [-1,1]."""
    rng = np.random.Generator(np.random.PCG64(1000 + index))
    n = int(round(seconds * sr))
    t = np.arange(n, dtype=np.float64) / sr
    phi = rng.uniform(0, 2 * np.pi)
    f0 = 220.0 * 2.0 ** (0.25 * np.sin(2 * np.pi * 0.5 * t + phi))
    phase = 2 * np.pi * np.cumsum(f0) / sr
    x = np.zeros(n)
    for h in range(1, 6):
        x += np.sin(h * phase) / h
    gate = ((t % 1.0) < 0.8).astype(np.float64)
    # 5 ms raised-cosine edges on the voiced gate
    edge = int(0.005 * sr)
    k = np.hanning(2 * edge + 1)
    gate = np.convolve(gate, k / k.sum(), mode="same")
    x = 0.3 * x * gate + 0.01 * rng.standard_normal(n)
    return x.astype(np.float32)


def make_index(n: int, dim: int = 768, seed: int = 0) -> np.ndarray:
    """Synthetic retrieval matrix (the ``big_npy`` of pipeline.py:323)."""
    return _normal(f"index.{n}.{dim}", (n, dim), 1.0, seed)


def make_index_from_feats(feats: np.ndarray, n_rows: int, seed: int = 0) -> np.ndarray:
    """Retrieval matrix with UNAMBIGUOUS neighbours for the queries ``feats`` (T, dim) ("queries
    perturbed copies so neighbours are unambiguous"): for every query 8 stored vectors at squared distances
    ~ dim * (0.005 k)^2, k = 1..8 (0.02 .. 1.2 at dim 768).  HuBERT frames of one clip lie as close as d ~ 2 to
    each other, so the planted copies must stay below that for the top-8 of a query to be its own 8 copies, in
    order, with gaps (>= 0.05) far above the float32 noise of the |q|^2 + |b|^2 - 2 q.b form (~1e-4).  The
    remaining rows are N(0,1) filler; rows are shuffled by a fixed permutation.  A real RVC index is built from
    the training set's HuBERT features, i.e. exactly "features plus small perturbations"."""
    feats = np.asarray(feats, np.float32)
    T, dim = feats.shape
    if 8 * T > n_rows:
        raise ValueError("index smaller than 8 rows per query")
    rows = _normal(f"index.filler.{n_rows}.{dim}", (n_rows, dim), 1.0, seed)
    eps = _normal(f"index.eps.{T}.{dim}", (8, T, dim), 1.0, seed)
    for k in range(8):
        rows[k * T:(k + 1) * T] = feats + np.float32(0.005 * (k + 1)) * eps[k]
    perm = _rng(f"index.perm.{n_rows}", seed).permutation(n_rows)
    return np.ascontiguousarray(rows[perm])


def to_torch(state: Dict[str, np.ndarray]):
    import torch
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in state.items()}
