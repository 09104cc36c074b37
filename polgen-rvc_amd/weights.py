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


def rmvpe_cfg_struct(cfg: Dict) -> _lib.RmvpeCfg:
    s = _lib.RmvpeCfg()
    s.n_blocks, s.en_de_layers = cfg["n_blocks"], cfg["en_de_layers"]
    s.inter_layers, s.en_out_channels = cfg["inter_layers"], cfg["en_out_channels"]
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
