"""Oracle: RVC v2 Synthesizer.infer (TextEncoder -> reverse flow -> NSF-HiFi-GAN).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Functional torch-CPU fp32 restatement of
  rvc/lib/algorithm/synthesizers.py:163-188  (Synthesizer.infer)
  rvc/lib/algorithm/encoders.py:61-126       (Encoder / TextEncoder)
  rvc/lib/algorithm/attentions.py:63-221     (rel-pos MultiHeadAttention, FFN)
  rvc/lib/algorithm/residuals.py:45-53,144-229 (ResBlock1, coupling layers, Flip)
  rvc/lib/algorithm/modules.py:58-84         (WaveNet)
  rvc/lib/algorithm/nsf.py:36-40,120-144     (source module, GeneratorNSF.forward)
  rvc/lib/algorithm/generators.py:117-156    (SineGen)
The state dict is the checkpoint's ``cpt["weight"]`` (enc_q keys ignored), weight-norm
given either as parametrizations.weight.original0/1 or legacy weight_g/weight_v.
Noise tensors are explicit inputs: ``z_noise`` (B,inter,T) replaces
randn_like at synthesizers.py:174 and ``src_noise`` (B,T*upp,1) the one at
generators.py:154.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.1
WINDOW = 10  # encoders.py:22 window_size=10


def eff_weight(sd: Dict[str, torch.Tensor], prefix: str, dim: int = 0) -> torch.Tensor:
    """Effective conv weight: plain, or weight-norm folded  w = g * v / ||v||  with the norm
    over all dims except ``dim`` (torch.nn.utils.parametrizations.weight_norm)."""
    if prefix + ".weight" in sd:
        return sd[prefix + ".weight"].float()
    if prefix + ".parametrizations.weight.original0" in sd:
        g = sd[prefix + ".parametrizations.weight.original0"].float()
        v = sd[prefix + ".parametrizations.weight.original1"].float()
    else:
        g = sd[prefix + ".weight_g"].float()
        v = sd[prefix + ".weight_v"].float()
    red = [d for d in range(v.dim()) if d != dim]
    return v * (g / v.pow(2).sum(dim=red, keepdim=True).sqrt())


def _b(sd, prefix):
    return sd[prefix + ".bias"].float() if prefix + ".bias" in sd else None


def cfg_fields(cfg):
    (spec, seg, inter, hidden, filt, n_heads, n_layers, ksz, pdrop, resblock, rks, rds,
     ups, up_init, upks, spk, gin, sr) = cfg
    return dict(inter=inter, hidden=hidden, filt=filt, n_heads=n_heads, n_layers=n_layers,
                ksz=ksz, rks=rks, rds=rds, ups=ups, up_init=up_init, upks=upks, spk=spk,
                gin=gin, sr=sr, upp=math.prod(ups))


# ---------------------------------------------------------------- TextEncoder
def layer_norm_c(x, gamma, beta, eps=1e-5):
    # normalization.py:13-16 -- LayerNorm over the channel dim of (B,C,T)
    return F.layer_norm(x.transpose(1, -1), (x.size(1),), gamma, beta, eps).transpose(1, -1)


def rel_attention(sd, p, x, n_heads, mask):
    """attentions.py:63-113 with window_size=10, written in banded form: the reference
    pads emb_rel to 2T-1 and reshapes; mathematically scores[i,j] += q_i . Ek[j-i+10] for
    |j-i| <= 10 and out_i += sum_r p[i,i+r-10] Ev[r]."""
    B, C, T = x.shape
    kc = C // n_heads
    q = F.conv1d(x, sd[p + ".conv_q.weight"].float(), _b(sd, p + ".conv_q"))
    k = F.conv1d(x, sd[p + ".conv_k.weight"].float(), _b(sd, p + ".conv_k"))
    v = F.conv1d(x, sd[p + ".conv_v.weight"].float(), _b(sd, p + ".conv_v"))
    q = q.view(B, n_heads, kc, T).transpose(2, 3)
    k = k.view(B, n_heads, kc, T).transpose(2, 3)
    v = v.view(B, n_heads, kc, T).transpose(2, 3)
    qs = q / math.sqrt(kc)                              # attentions.py:79
    scores = torch.matmul(qs, k.transpose(-2, -1))      # (B,H,T,T)
    ek = sd[p + ".emb_rel_k"].float()[0]                # (21,kc), heads_share=True
    ev = sd[p + ".emb_rel_v"].float()[0]
    rel = torch.matmul(qs, ek.t())                      # (B,H,T,21)
    for r in range(2 * WINDOW + 1):
        off = r - WINDOW
        if abs(off) >= T:
            continue
        lo, hi = max(0, -off), T - max(0, off)
        scores.diagonal(offset=off, dim1=-2, dim2=-1).add_(rel[..., lo:hi, r])
    if mask is not None:                                # attentions.py:94 (fill -1e4)
        am = mask.unsqueeze(2) * mask.unsqueeze(-1)
        scores = scores.masked_fill(am == 0, -1e4)
    pa = F.softmax(scores, dim=-1)
    out = torch.matmul(pa, v)                           # (B,H,T,kc)
    band = torch.zeros(B, n_heads, T, 2 * WINDOW + 1, dtype=pa.dtype)
    for r in range(2 * WINDOW + 1):
        off = r - WINDOW
        if abs(off) >= T:
            continue
        lo, hi = max(0, -off), T - max(0, off)
        band[..., lo:hi, r] = pa.diagonal(offset=off, dim1=-2, dim2=-1)
    out = out + torch.matmul(band, ev)                  # attentions.py:106-111
    out = out.transpose(2, 3).contiguous().view(B, C, T)
    return F.conv1d(out, sd[p + ".conv_o.weight"].float(), _b(sd, p + ".conv_o"))


def ffn(sd, p, x, mask, ksz):
    # attentions.py:195-203, "same" padding :214-221, activation=None -> relu
    pl, pr = (ksz - 1) // 2, ksz // 2
    h = F.conv1d(F.pad(x * mask, (pl, pr)), sd[p + ".conv_1.weight"].float(), _b(sd, p + ".conv_1"))
    h = torch.relu(h)
    h = F.conv1d(F.pad(h * mask, (pl, pr)), sd[p + ".conv_2.weight"].float(), _b(sd, p + ".conv_2"))
    return h * mask


def text_encoder(sd, cfg, phone, pitch, lengths):
    """encoders.py:111-126.  phone (B,T,D) f32, pitch (B,T) int64, lengths (B,)."""
    c = cfg_fields(cfg)
    x = F.linear(phone, sd["enc_p.emb_phone.weight"].float(), sd["enc_p.emb_phone.bias"].float())
    if pitch is not None:
        x = x + F.embedding(pitch, sd["enc_p.emb_pitch.weight"].float())
    x = x * math.sqrt(c["hidden"])
    x = F.leaky_relu(x, 0.1)
    x = x.transpose(1, -1)
    T = x.size(2)
    mask = (torch.arange(T)[None, :] < lengths[:, None]).unsqueeze(1).to(x.dtype)
    x = x * mask
    # Encoder.forward encoders.py:61-73
    x = x * mask
    for i in range(c["n_layers"]):
        y = rel_attention(sd, f"enc_p.encoder.attn_layers.{i}", x, c["n_heads"], mask)
        x = layer_norm_c(x + y, sd[f"enc_p.encoder.norm_layers_1.{i}.gamma"].float(),
                         sd[f"enc_p.encoder.norm_layers_1.{i}.beta"].float())
        y = ffn(sd, f"enc_p.encoder.ffn_layers.{i}", x, mask, c["ksz"])
        x = layer_norm_c(x + y, sd[f"enc_p.encoder.norm_layers_2.{i}.gamma"].float(),
                         sd[f"enc_p.encoder.norm_layers_2.{i}.beta"].float())
    x = x * mask
    stats = F.conv1d(x, sd["enc_p.proj.weight"].float(), _b(sd, "enc_p.proj")) * mask
    m, logs = torch.split(stats, c["inter"], dim=1)
    return m, logs, mask


# ---------------------------------------------------------------- flow (reverse)
def wavenet(sd, p, x, mask, g, hidden, n_layers=3, ksz=5):
    # modules.py:58-84, dilation_rate=1
    out = torch.zeros_like(x)
    gc = F.conv1d(g, eff_weight(sd, p + ".cond_layer"), _b(sd, p + ".cond_layer"))
    for i in range(n_layers):
        x_in = F.conv1d(x, eff_weight(sd, f"{p}.in_layers.{i}"), _b(sd, f"{p}.in_layers.{i}"),
                        padding=(ksz - 1) // 2)
        a = x_in + gc[:, i * 2 * hidden:(i + 1) * 2 * hidden]
        acts = torch.tanh(a[:, :hidden]) * torch.sigmoid(a[:, hidden:])   # commons.py:79-86
        rs = F.conv1d(acts, eff_weight(sd, f"{p}.res_skip_layers.{i}"), _b(sd, f"{p}.res_skip_layers.{i}"))
        if i < n_layers - 1:
            x = (x + rs[:, :hidden]) * mask
            out = out + rs[:, hidden:]
        else:
            out = out + rs
    return out * mask


def flow_reverse(sd, cfg, z, mask, g):
    # residuals.py:154-157: for flow in reversed([RCL0,Flip,RCL1,Flip,RCL2,Flip,RCL3,Flip])
    c = cfg_fields(cfg)
    half = c["inter"] // 2
    x = z
    for f in (6, 4, 2, 0):
        x = torch.flip(x, [1])
        p = f"flow.flows.{f}"
        x0, x1 = x[:, :half], x[:, half:]
        h = F.conv1d(x0, sd[p + ".pre.weight"].float(), _b(sd, p + ".pre")) * mask
        h = wavenet(sd, p + ".enc", h, mask, g, c["hidden"])
        m = F.conv1d(h, sd[p + ".post.weight"].float(), _b(sd, p + ".post")) * mask
        x1 = (x1 - m) * mask                    # mean_only: logs == 0  (residuals.py:218-227)
        x = torch.cat([x0, x1], 1)
    return x


# ---------------------------------------------------------------- NSF decoder
def sine_source(f0, upp, sr, src_noise, sine_amp=0.1, noise_std=0.003):
    """generators.py:117-156 with harmonic_num=0 (rand_ini zeroed at :128).
    f0 (B,T) -> sine (B,T*upp,1); src_noise (B,T*upp,1) replaces randn_like (:154)."""
    f0 = f0[:, None].transpose(1, 2)                        # (B,T,1)
    rad = (f0 / float(sr)) % 1
    tmp = torch.cumsum(rad, 1) * upp
    tmp = F.interpolate(tmp.transpose(2, 1), scale_factor=float(upp), mode="linear",
                        align_corners=True).transpose(2, 1)
    rad_up = F.interpolate(rad.transpose(2, 1), scale_factor=float(upp), mode="nearest").transpose(2, 1)
    tmp = tmp % 1
    wrap = (tmp[:, 1:, :] - tmp[:, :-1, :]) < 0
    shift = torch.zeros_like(rad_up)
    shift[:, 1:, :] = wrap * -1.0
    sine = torch.sin(torch.cumsum(rad_up + shift, dim=1) * 2 * torch.pi) * sine_amp
    uv = (f0 > 0).to(f0.dtype)
    uv = F.interpolate(uv.transpose(2, 1), scale_factor=float(upp), mode="nearest").transpose(2, 1)
    noise_amp = uv * noise_std + (1 - uv) * sine_amp / 3
    return sine * uv + noise_amp * src_noise


def resblock1(sd, p, x, ksz, dils):
    # residuals.py:45-53 (x_mask is None on this path)
    for m, d in enumerate(dils):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = F.conv1d(xt, eff_weight(sd, f"{p}.convs1.{m}"), _b(sd, f"{p}.convs1.{m}"),
                      dilation=d, padding=(ksz * d - d) // 2)
        xt = F.leaky_relu(xt, LRELU_SLOPE)
        xt = F.conv1d(xt, eff_weight(sd, f"{p}.convs2.{m}"), _b(sd, f"{p}.convs2.{m}"),
                      padding=(ksz - 1) // 2)
        x = xt + x
    return x


def nsf_decoder(sd, cfg, z, f0, g, src_noise, return_source=False):
    """nsf.py:120-144.  z (B,inter,T), f0 (B,T), g (B,gin,1) -> (B,1,T*upp)."""
    c = cfg_fields(cfg)
    sine = sine_source(f0, c["upp"], c["sr"], src_noise)
    har = torch.tanh(F.linear(sine, sd["dec.m_source.l_linear.weight"].float(),
                              sd["dec.m_source.l_linear.bias"].float())).transpose(1, 2)
    x = F.conv1d(z, sd["dec.conv_pre.weight"].float(), _b(sd, "dec.conv_pre"), padding=3)
    x = x + F.conv1d(g, sd["dec.cond.weight"].float(), _b(sd, "dec.cond"))
    nk = len(c["rks"])
    for i, (u, k) in enumerate(zip(c["ups"], c["upks"])):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, eff_weight(sd, f"dec.ups.{i}"), _b(sd, f"dec.ups.{i}"),
                               stride=u, padding=(k - u) // 2)
        sf0 = math.prod(c["ups"][i + 1:]) if i + 1 < len(c["ups"]) else 1
        x = x + F.conv1d(har, sd[f"dec.noise_convs.{i}.weight"].float(), _b(sd, f"dec.noise_convs.{i}"),
                         stride=sf0, padding=(sf0 // 2 if sf0 > 1 else 0))
        xs = None
        for j in range(nk):
            y = resblock1(sd, f"dec.resblocks.{i * nk + j}", x, c["rks"][j], c["rds"][j])
            xs = y if xs is None else xs + y
        x = xs / nk
    x = F.leaky_relu(x)                                     # nsf.py:142 default slope 0.01
    x = torch.tanh(F.conv1d(x, sd["dec.conv_post.weight"].float(), None, padding=3))
    return (x, har) if return_source else x


# ---------------------------------------------------------------- Synthesizer.infer
@torch.no_grad()
def synthesizer_infer(sd, cfg, phone, lengths, pitch, nsff0, sid, z_noise, src_noise,
                      return_parts=False):
    """synthesizers.py:163-188 (rate=None, use_f0=1)."""
    g = F.embedding(sid, sd["emb_g.weight"].float()).unsqueeze(-1)          # :172
    m_p, logs_p, mask = text_encoder(sd, cfg, phone, pitch, lengths)        # :173
    z_p = (m_p + torch.exp(logs_p) * z_noise * 0.66666) * mask              # :174
    z = flow_reverse(sd, cfg, z_p, mask, g)                                 # :183
    o = nsf_decoder(sd, cfg, z * mask, nsff0, g, src_noise)                 # :184
    if return_parts:
        return o, dict(m_p=m_p, logs_p=logs_p, z_p=z_p, z=z, g=g)
    return o
