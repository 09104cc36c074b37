"""Checkpoint dicts -> the config structs / tensor tables of the C ABI (include/rvcx.h)."""
from __future__ import annotations

from typing import Dict, List

import numpy as np

from . import _lib


def synth_cfg_struct(cfg: List, input_dim: int = 768) -> _lib.SynthCfg:
    """cfg is the 18-element ``cpt["config"]`` list (rvc/infer/infer.py:86-97)."""
    (_, _, inter, hidden, filt, n_heads, n_layers, ksz, _, resblock, rks, rds, ups, up_init, upks, spk,
     gin, sr) = cfg
    if str(resblock) != "1":
        raise ValueError("only resblock='1' (ResBlock1) checkpoints are supported (RVC v2)")
    s = _lib.SynthCfg()
    s.inter_channels, s.hidden_channels, s.filter_channels = inter, hidden, filt
    s.n_heads, s.n_layers, s.kernel_size = n_heads, n_layers, ksz
    s.n_resblocks = len(rks)
    for i, k in enumerate(rks):
        s.res_kernels[i] = k
        if len(rds[i]) != 3:
            raise ValueError("ResBlock1 needs 3 dilations")
        for j, d in enumerate(rds[i]):
            s.res_dilations[i][j] = d
    s.n_ups = len(ups)
    for i, (u, k) in enumerate(zip(ups, upks)):
        s.up_rates[i], s.up_kernels[i] = u, k
    s.up_initial_channel, s.spk_embed_dim, s.gin_channels, s.sr = up_init, spk, gin, sr
    s.input_dim = input_dim
    return s


def synth_dec_rf(cfg: List) -> int:
    """Frames of z an output sample of the NSF decoder can depend on, each side, + 2 -- the host-side twin of
    SynthModel::dec_rf_frames (csrc/synth.hip): conv_pre k = 7; per stage the ConvTranspose1d's (k - 1) // stride + 1 input
    samples and the widest ResBlock1's sum_d (k - 1) // 2 * (d + 1) samples at that stage's rate; conv_post k = 7
    (rvc/lib/algorithm/nsf.py:100-144, residuals.py:15-62).  The decoder window of the pipeline (SynthIO::dec_skip) leaves
    this many frames around the samples VC.pipeline keeps."""
    import math
    rks, rds, ups, upks = cfg[10], cfg[11], cfg[12], cfg[14]
    rf, rate = 3.0, 1.0
    for u, k in zip(ups, upks):
        rf += ((k - 1) // u + 1) / rate
        rate *= u
        rf += max(sum((rk - 1) // 2 * (d + 1) for d in ds) for rk, ds in zip(rks, rds)) / rate
    rf += 3.0 / rate
    return int(math.ceil(rf)) + 2


def rmvpe_cfg_struct(cfg: Dict) -> _lib.RmvpeCfg:
    s = _lib.RmvpeCfg()
    s.n_blocks, s.en_de_layers = cfg["n_blocks"], cfg["en_de_layers"]
    s.inter_layers, s.en_out_channels = cfg["inter_layers"], cfg["en_out_channels"]
    return s


def fcpe_cfg_struct(cfg: Dict) -> _lib.FcpeCfg:
    s = _lib.FcpeCfg()
    s.n_layers, s.n_chans = cfg["n_layers"], cfg["n_chans"]
    s.input_channel, s.out_dims = cfg.get("input_channel", 128), cfg.get("out_dims", 360)
    s.heads, s.dim_head = cfg.get("heads", 8), cfg.get("dim_head", 64)
    s.nb_features, s.dw_kernel = cfg.get("nb_features", 266), cfg.get("dw_kernel", 31)
    s.mel_fmin, s.mel_fmax = float(cfg.get("mel_fmin", 0.0)), float(cfg.get("mel_fmax", 8000.0))
    return s


def hubert_cfg_struct(cfg: Dict) -> _lib.HubertCfg:
    s = _lib.HubertCfg()
    s.conv_dim, s.n_conv = cfg["conv_dim"], len(cfg["conv_kernels"])
    for i, (k, st) in enumerate(zip(cfg["conv_kernels"], cfg["conv_strides"])):
        s.conv_kernels[i], s.conv_strides[i] = k, st
    s.embed_dim, s.ffn_dim, s.heads, s.layers = cfg["embed_dim"], cfg["ffn_dim"], cfg["heads"], cfg["layers"]
    s.pos_kernel, s.pos_groups = cfg["pos_kernel"], cfg["pos_groups"]
    return s


def strip_enc_q(state: Dict) -> Dict:
    """get_vc deletes enc_q before loading (rvc/infer/infer.py:99-100)."""
    return {k: v for k, v in state.items() if not k.startswith("enc_q.")}


def _shape(t):
    return tuple(int(v) for v in t.shape)


def hubert_cfg_from_state(state: Dict) -> Dict:
    """HuBERT geometry read off a fairseq-named state dict (custom HuBERTs, tabs/install/install_huberts.py:11-18,
    share the architecture but need not share the sizes).  Strides are not recoverable from weights: fairseq's
    default extractor "[(512,10,5)] + [(512,3,2)]*4 + [(512,2,2)]*2" is assumed."""
    convs = []
    while f"feature_extractor.conv_layers.{len(convs)}.0.weight" in state:
        convs.append(_shape(state[f"feature_extractor.conv_layers.{len(convs)}.0.weight"]))
    layers = 0
    while f"encoder.layers.{layers}.fc1.weight" in state:
        layers += 1
    embed = _shape(state["post_extract_proj.weight"])[0]
    wv = _shape(state["encoder.pos_conv.0.weight_v"])
    kernels = [c[2] for c in convs]
    if len(kernels) != 7:
        raise ValueError(f"unsupported HuBERT feature extractor ({len(kernels)} conv layers)")
    return dict(conv_dim=convs[0][0], conv_kernels=kernels, conv_strides=[5, 2, 2, 2, 2, 2, 2], embed_dim=embed,
                ffn_dim=_shape(state["encoder.layers.0.fc1.weight"])[0], heads=max(1, embed // 64), layers=layers,
                pos_kernel=wv[2], pos_groups=embed // wv[1],
                final_dim=_shape(state["final_proj.weight"])[0] if "final_proj.weight" in state else 256)


def fcpe_cfg_from_state(state: Dict, config: Dict = None) -> Dict:
    """FCPE geometry read off the "model" state dict of fcpe.pt (the reference builds the module from the
    checkpoint's "config" block, FCPE.py:715-733; the tensor shapes say the same and also give the module
    defaults the block does not hold: heads x dim_head, the number of random features, the depth-wise kernel).
    ``config``: that block, used for the mel band edges and to refuse a front end this library does not have."""
    layers = 0
    while f"decoder._layers.{layers}.norm.weight" in state:
        layers += 1
    n_chans, input_channel = _shape(state["stack.0.weight"])[:2]
    proj = _shape(state["decoder._layers.0.attn.fast_attention.projection_matrix"])
    inner = _shape(state["decoder._layers.0.attn.to_q.weight"])[0]
    key = "dense_out.parametrizations.weight.original0" if "dense_out.parametrizations.weight.original0" in state \
        else "dense_out.weight_g"
    cfg = dict(n_layers=layers, n_chans=n_chans, input_channel=input_channel, out_dims=_shape(state[key])[0],
               heads=inner // proj[1], dim_head=proj[1], nb_features=proj[0],
               dw_kernel=_shape(state["decoder._layers.0.conformer.net.4.conv.weight"])[2], mel_fmin=0.0, mel_fmax=8000.0)
    mel = dict((config or {}).get("mel", {}))
    if mel:
        want = dict(sampling_rate=16000, num_mels=128, n_fft=1024, win_size=1024, hop_size=160)
        bad = {k: mel.get(k) for k, v in want.items() if mel.get(k, v) != v}
        if bad:
            raise ValueError(f"fcpe checkpoint with an unsupported mel front end: {bad} (supported: {want})")
        cfg["mel_fmin"], cfg["mel_fmax"] = float(mel.get("fmin", 0.0)), float(mel.get("fmax", 8000.0))
    return cfg


def rmvpe_cfg_from_state(state: Dict) -> Dict:
    """E2E(n_blocks, n_gru, (2,2), en_de_layers, inter_layers, 1, en_out_channels) read off rmvpe.pt's keys."""
    def count(fmt):
        n = 0
        while any(k.startswith(fmt.format(n)) for k in state):
            n += 1
        return n
    return dict(n_blocks=count("unet.encoder.layers.0.conv.{}."), n_gru=1,
                en_de_layers=count("unet.encoder.layers.{}."), inter_layers=count("unet.intermediate.layers.{}."),
                in_channels=1, en_out_channels=_shape(state["unet.encoder.layers.0.conv.0.conv.0.weight"])[0])
