"""Oracle: HuBERT-base / ContentVec ``extract_features(source, padding_mask, output_layer)``.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED BY THE REFERENCE: the model
is fairseq==0.12.2 ``HubertModel`` (requirements.txt:3), loaded at rvc/infer/infer.py:67-74
and called at rvc/infer/pipeline.py:228-236; its source is not under /root/reference and
fairseq is not installable offline.  This file restates the published architecture
(hubert_base_ls960: "default" extractor mode = GroupNorm after conv 0, no conv bias,
layer_norm_first=False i.e. post-LN, pos_conv k=128 g=16 with weight_norm(dim=2) + SamePad
+ GELU) on fairseq's key names; tools/gen_golden.py cross-checks it against
``transformers.HubertModel`` (same architecture, HF names) with identical weights.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F


def pos_conv_weight(sd):
    g = sd["encoder.pos_conv.0.weight_g"].float()        # (1,1,K): weight_norm(dim=2)
    v = sd["encoder.pos_conv.0.weight_v"].float()        # (E, E/G, K)
    return v * (g / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt())


def feature_extractor(sd, cfg, wav):
    """fairseq ConvFeatureExtractionModel, mode='default', conv_bias=False.  wav (B,N) -> (B,C,T')."""
    x = wav.unsqueeze(1)
    for i, (k, s) in enumerate(zip(cfg["conv_kernels"], cfg["conv_strides"])):
        x = F.conv1d(x, sd[f"feature_extractor.conv_layers.{i}.0.weight"].float(), None, stride=s)
        if i == 0:
            C = x.shape[1]
            x = F.group_norm(x, C, sd["feature_extractor.conv_layers.0.2.weight"].float(),
                             sd["feature_extractor.conv_layers.0.2.bias"].float(), 1e-5)
        x = F.gelu(x)
    return x


def self_attention(sd, p, x, heads):
    # fairseq MultiheadAttention: q scaled by head_dim**-0.5 after projection
    B, T, E = x.shape
    hd = E // heads
    q = F.linear(x, sd[p + ".q_proj.weight"].float(), sd[p + ".q_proj.bias"].float()) * hd ** -0.5
    k = F.linear(x, sd[p + ".k_proj.weight"].float(), sd[p + ".k_proj.bias"].float())
    v = F.linear(x, sd[p + ".v_proj.weight"].float(), sd[p + ".v_proj.bias"].float())
    q = q.view(B, T, heads, hd).transpose(1, 2)
    k = k.view(B, T, heads, hd).transpose(1, 2)
    v = v.view(B, T, heads, hd).transpose(1, 2)
    a = F.softmax(torch.matmul(q, k.transpose(-1, -2)), dim=-1)
    o = torch.matmul(a, v).transpose(1, 2).reshape(B, T, E)
    return F.linear(o, sd[p + ".out_proj.weight"].float(), sd[p + ".out_proj.bias"].float())


@torch.no_grad()
def extract_features(sd: Dict[str, torch.Tensor], cfg: dict, wav: torch.Tensor,
                     output_layer: int = 12, return_parts=False) -> torch.Tensor:
    """wav (B,N) f32 (no waveform normalisation: pipeline.py:220-232) -> (B,T',E), the
    output of transformer layer ``output_layer`` (1-based), as fairseq
    HubertModel.extract_features(..., mask=False, output_layer=L)[0]."""
    feats = feature_extractor(sd, cfg, wav)                       # (B,C,T')
    x = feats.transpose(1, 2)
    x = F.layer_norm(x, (x.shape[-1],), sd["layer_norm.weight"].float(), sd["layer_norm.bias"].float())
    x = F.linear(x, sd["post_extract_proj.weight"].float(), sd["post_extract_proj.bias"].float())
    K, G = cfg["pos_kernel"], cfg["pos_groups"]
    pc = F.conv1d(x.transpose(1, 2), pos_conv_weight(sd), sd["encoder.pos_conv.0.bias"].float(),
                  padding=K // 2, groups=G)
    if K % 2 == 0:
        pc = pc[:, :, :-1]                                        # SamePad
    x = x + F.gelu(pc).transpose(1, 2)
    x = F.layer_norm(x, (x.shape[-1],), sd["encoder.layer_norm.weight"].float(),
                     sd["encoder.layer_norm.bias"].float())       # layer_norm_first=False
    parts = {"conv": feats, "pre": x}
    for l in range(min(output_layer, cfg["layers"])):
        p = f"encoder.layers.{l}"
        x = x + self_attention(sd, p + ".self_attn", x, cfg["heads"])
        x = F.layer_norm(x, (x.shape[-1],), sd[p + ".self_attn_layer_norm.weight"].float(),
                         sd[p + ".self_attn_layer_norm.bias"].float())
        h = F.gelu(F.linear(x, sd[p + ".fc1.weight"].float(), sd[p + ".fc1.bias"].float()))
        h = F.linear(h, sd[p + ".fc2.weight"].float(), sd[p + ".fc2.bias"].float())
        x = F.layer_norm(x + h, (x.shape[-1],), sd[p + ".final_layer_norm.weight"].float(),
                         sd[p + ".final_layer_norm.bias"].float())
    return (x, parts) if return_parts else x


def final_proj(sd, x):
    """v1 path only (pipeline.py:236)."""
    return F.linear(x, sd["final_proj.weight"].float(), sd["final_proj.bias"].float())
