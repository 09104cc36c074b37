// HuBERT-base / ContentVec content encoder: HubertModel.extract_features(source, padding_mask=False,
// output_layer=L)[0] of fairseq 0.12.2 (loaded at rvc/infer/infer.py:67-74, called at
// rvc/infer/pipeline.py:228-236).  fairseq is not vendored in the reference, so the arithmetic follows
// the published architecture (extractor mode "default": GroupNorm after conv 0, no conv bias; post-LN
// encoder; pos_conv k=128 g=16 weight_norm(dim=2) + SamePad + GELU).
// Activations stay channel-first (B, C, T): every Linear is a 1x1 MFMA conv, QKV is one fused GEMM.
#include <cmath>

#include "models.h"
#include "ops.h"

namespace rvcx {

std::unique_ptr<HubertModel> hubert_load(Ctx& c, const rvcx_hubert_cfg& cfg, const TensorTable& t) {
  auto M = std::make_unique<HubertModel>();
  RegionScope scope(c, *M->region);
  M->cfg = cfg;
  const int E = cfg.embed_dim;
  for (int i = 0; i < cfg.n_conv; ++i) {
    const std::string p = "feature_extractor.conv_layers." + std::to_string(i) + ".0.weight";
    auto w = t.f32(p);
    const auto shp = t.shape(p);
    RVCX_CHECK((int)shp[2] == cfg.conv_kernels[i], "hubert conv kernel mismatch");
    M->convs.push_back(make_conv(c, w.data(), nullptr, (int)shp[0], (int)shp[1], (int)shp[2], 1));
  }
  M->gn_g = c.slab.upload(t.f32("feature_extractor.conv_layers.0.2.weight"));
  M->gn_b = c.slab.upload(t.f32("feature_extractor.conv_layers.0.2.bias"));
  M->ln0_g = c.slab.upload(t.f32("layer_norm.weight"));
  M->ln0_b = c.slab.upload(t.f32("layer_norm.bias"));
  {
    auto w = t.f32("post_extract_proj.weight");
    auto b = t.f32("post_extract_proj.bias");
    M->proj = make_conv(c, w.data(), b.data(), E, cfg.conv_dim, 1, 1);
  }
  {
    std::vector<float> w = wn_weight(t, "encoder.pos_conv.0", 2);
    auto b = t.f32("encoder.pos_conv.0.bias");
    M->pos_conv = make_conv(c, w.data(), b.data(), E, E / cfg.pos_groups, cfg.pos_kernel, cfg.pos_groups);
  }
  M->eln_g = c.slab.upload(t.f32("encoder.layer_norm.weight"));
  M->eln_b = c.slab.upload(t.f32("encoder.layer_norm.bias"));
  for (int l = 0; l < cfg.layers; ++l) {
    HubertModel::Layer L;
    const std::string p = "encoder.layers." + std::to_string(l);
    std::vector<float> w, b;
    for (const char* n : {".self_attn.q_proj", ".self_attn.k_proj", ".self_attn.v_proj"}) {
      auto wi = t.f32(p + n + ".weight");
      auto bi = t.f32(p + n + ".bias");
      w.insert(w.end(), wi.begin(), wi.end());
      b.insert(b.end(), bi.begin(), bi.end());
    }
    L.qkv = make_conv(c, w.data(), b.data(), 3 * E, E, 1, 1);
    auto lin = [&](const std::string& n, int co, int ci) {
      auto wi = t.f32(p + n + ".weight");
      auto bi = t.f32(p + n + ".bias");
      return make_conv(c, wi.data(), bi.data(), co, ci, 1, 1);
    };
    L.att = make_att_flag(c);
    L.o = lin(".self_attn.out_proj", E, E);
    L.fc1 = lin(".fc1", cfg.ffn_dim, E);
    L.fc2 = lin(".fc2", E, cfg.ffn_dim);
    L.ln1_g = c.slab.upload(t.f32(p + ".self_attn_layer_norm.weight"));
    L.ln1_b = c.slab.upload(t.f32(p + ".self_attn_layer_norm.bias"));
    L.ln2_g = c.slab.upload(t.f32(p + ".final_layer_norm.weight"));
    L.ln2_b = c.slab.upload(t.f32(p + ".final_layer_norm.bias"));
    M->layers.push_back(L);
  }
  // present in fairseq checkpoints, used by RVC v1 voice models only (feats = final_proj(layer 9), pipeline.py:231-236)
  if (t.has("final_proj.weight") && t.has("final_proj.bias")) {
    auto w = t.f32("final_proj.weight");
    auto b = t.f32("final_proj.bias");
    const auto shp = t.shape("final_proj.weight");
    RVCX_CHECK((int)shp[1] == E, "hubert final_proj: input width != embed_dim");
    M->final_proj = make_conv(c, w.data(), b.data(), (int)shp[0], E, 1, 1);
    M->has_final_proj = true;
  }
  M->region->seal();
  return M;
}

int hubert_frames(const HubertModel& m, int64_t n) {
  int64_t t = n;
  for (int i = 0; i < m.cfg.n_conv; ++i) t = (t - m.cfg.conv_kernels[i]) / m.cfg.conv_strides[i] + 1;
  return (int)t;
}

size_t hubert_arena_bytes(const HubertModel& m, int B, int64_t n) {
  const int64_t t0 = (n - m.cfg.conv_kernels[0]) / m.cfg.conv_strides[0] + 1;
  const int T = hubert_frames(m, n);
  const size_t v1_extra = m.has_final_proj ? (size_t)B * m.cfg.embed_dim * T * sizeof(float) + 256 : 0;   // layer-9 map of the v1 path
  size_t conv = 2 * (size_t)m.cfg.conv_dim * t0;
  size_t enc = (size_t)T * (size_t)(11 * m.cfg.embed_dim + m.cfg.ffn_dim + 64 + (8 * 98 + 2 * 96 + 8) * m.cfg.heads);
  const size_t gn0 = hubert_conv0_part_doubles(B, m.cfg.conv_dim, (int)t0) * sizeof(double) + 256;   // fused conv0 + GroupNorm partial sums
  return (size_t)B * (conv + enc) * sizeof(float) + v1_extra + gn0 + ((size_t)64 << 20);
}

void hubert_forward(Ctx& c, const HubertModel& m, int B, const float* wav, int64_t n, int output_layer,
                    float* feats_ct, hipStream_t s, const std::function<void()>* after_extractor, long wav_bs,
                    const int* ns_host) {
  Arena& A = c.arena;
  const auto& cf = m.cfg;
  const int C = cf.conv_dim, E = cf.embed_dim;
  const int64_t t0 = (n - cf.conv_kernels[0]) / cf.conv_strides[0] + 1;
  RVCX_CHECK(t0 > 0, "hubert: input too short");
  // ---- ragged batch: item b holds ns_host[b] <= n samples (zeros behind them).  The extractor has no padding, so an
  // item's valid frames depend on its valid samples only; what needs the item's own length is the GroupNorm
  // statistics, the zero padding of pos_conv and the attention's key range.
  bool ragged = false;
  std::vector<int> t0b(B, (int)t0), Tb(B, hubert_frames(m, n));
  if (ns_host)
    for (int b = 0; b < B; ++b) {
      RVCX_CHECK(ns_host[b] <= n, "hubert: item longer than the batch geometry");
      t0b[b] = (int)((ns_host[b] - cf.conv_kernels[0]) / cf.conv_strides[0] + 1);
      Tb[b] = hubert_frames(m, ns_host[b]);
      RVCX_CHECK(Tb[b] > 0, "hubert: input too short");
      ragged |= ns_host[b] != n;
    }
  const int *d_t0 = nullptr, *d_T = nullptr;
  std::vector<const int*> d_tl(cf.n_conv, nullptr);     // valid frames behind extractor layer i
  if (ragged) {
    std::vector<std::vector<int>> all{t0b, Tb};
    std::vector<int> tl(t0b);
    for (int i = 1; i < cf.n_conv; ++i) {
      for (auto& t : tl) t = (t - cf.conv_kernels[i]) / cf.conv_strides[i] + 1;
      all.push_back(tl);
    }
    const std::vector<int*> d = dev_ints_many(A, all, s);
    d_t0 = d[0];
    d_T = d[1];
    d_tl[0] = d_t0;
    for (int i = 1; i < cf.n_conv; ++i) d_tl[i] = d[1 + i];
  }
  float* bufa = A.alloc<float>((size_t)B * C * t0);
  float* bufb = A.alloc<float>((size_t)B * C * t0);
  float* gn_stats = A.alloc<float>((size_t)B * C * 2);
  // ---- conv feature extractor
  int64_t Tin = n;
  const float* x = wav;
  float* y = bufa;
  // Layers 1 .. 4 (k = 3, stride 2, 512 -> 512: 143 of the extractor's 148 GFLOP per 30 s) hand their GELU outputs on
  // already split into the fp16 hi / lo pairs the next layer's tiles are made of (ConvArgs::y_split -> x_split, the
  // same bytes as fp32): a stride-2 layer stages two input positions per output, so converting them in the consumer --
  // once per 64-channel block row, eight times over -- cost as many VALU cycles as the tile's MFMAs.  Same values,
  // same fragments: the features are bit-identical to the fp32 hand-off (RVCX_NO_SPLIT=1).
  static const bool no_split = getenv("RVCX_NO_SPLIT") && atoi(getenv("RVCX_NO_SPLIT")) != 0;
  std::vector<ConvArgs> la(cf.n_conv);
  {
    int64_t ti = n;
    for (int i = 0; i < cf.n_conv; ++i) {
      const int64_t to = (ti - cf.conv_kernels[i]) / cf.conv_strides[i] + 1;
      RVCX_CHECK(to > 0, "hubert: input too short");
      la[i] = conv1d_args(m.convs[i], nullptr, nullptr, B, (int)ti, (int)to, cf.conv_strides[i], 1, 0);
      if (i > 0) la[i].act = ACT_GELU;
      ti = to;
    }
  }
  auto split_after = [&](int i) {        // does layer i (0: its GroupNorm + GELU) store split outputs for layer i + 1?
    if (no_split || i + 1 >= cf.n_conv || i > 3 || cf.conv_strides[i + 1] != 2 || !conv_h3_split_ok(la[i + 1])) return false;
    return i == 0 ? C % 16 == 0 : (cf.conv_kernels[i + 1] == cf.conv_kernels[i] && conv_h3_split_ok(la[i]));
  };
  for (int i = 0; i < cf.n_conv; ++i) {
    const int64_t Tout = la[i].Nout;
    ConvArgs a = la[i];
    a.x = x;
    a.y = y;
    if (i == 0 && wav_bs > 0) a.x_bs = wav_bs;      // the B signals are slices of longer rows
    a.lens_out = d_tl[i];       // frames behind an item's last one: zeros, and tiles that hold nothing else are skipped
    if (i > 0 && split_after(i - 1)) a.x_split = x;
    if (i > 0 && split_after(i)) {
      a.y_split = y;
      a.y = nullptr;
    }
    // Round 6: layer 0 (a Cin = 1 FIR) is recomputed inside its GroupNorm's two passes instead of being stored (ops.hip:
    // hubert_conv0_*): the (B, 512, T0) fp32 map -- 210 MB per 32 s clip, written once and read twice -- never exists.
    // RVCX_HUBERT_FUSE0=0: the three separate passes.
    static const bool fuse0_on = !getenv("RVCX_HUBERT_FUSE0") || atoi(getenv("RVCX_HUBERT_FUSE0")) != 0;
    // For micro-batches only (B >= 8: 256 blocks of 16 channels): the recomputed statistics pass is VALU-bound and as long as
    // the read it replaces (368 vs 363 us at B = 8); at B = 1 its 32 - 128 blocks are latency-bound (170 us against 45) and
    // the fusion loses 0.1 ms.  Both forms produce the same bits (same FIR, same order of the sums), so a batch member
    // still equals its single run.
    const bool fuse0 = i == 0 && fuse0_on && split_after(0) && m.convs[0].cin == 1 && m.convs[0].groups == 1 && !m.convs[0].bias &&
                       m.convs[0].w && cf.conv_kernels[0] <= 16 && (long)B * (C / 16) >= 256;
    if (fuse0) {
      float* z = (y == bufa) ? bufb : bufa;
      double* part = A.alloc<double>(hubert_conv0_part_doubles(B, C, (int)Tout));
      launch_hubert_conv0_gn_gelu_split(wav, wav_bs > 0 ? wav_bs : (long)n, m.convs[0].w, cf.conv_kernels[0], cf.conv_strides[0],
                                        m.convs[0].cin_gp * m.convs[0].cout_gp, m.gn_g, m.gn_b, part, gn_stats, z, B, C, (int)Tout, 1e-5f, s, d_t0,
                                        c.dev_err, m.convs[1].ovf_word, ++c.launch_seq);
      c.flops += 2.0 * cf.conv_kernels[0] * C * (double)Tout * B * 2;      // the FIR, evaluated twice
      RVCX_HIP(hipGetLastError());
      y = z;
      x = y;
      y = (y == bufa) ? bufb : bufa;
      Tin = Tout;
      continue;
    }
    c.conv_on(a, s);
    if (i == 0) {
      float* z = (y == bufa) ? bufb : bufa;
      if (split_after(0))
        launch_groupnorm_gelu_split(y, m.gn_g, m.gn_b, gn_stats, z, B, C, (int)Tout, 1e-5f, s, d_t0, c.dev_err,
                                    m.convs[1].ovf_word, ++c.launch_seq);
      else
        launch_groupnorm_gelu(y, m.gn_g, m.gn_b, z, B, C, (int)Tout, 1e-5f, s, d_t0);
      y = z;
    }
    x = y;
    y = (y == bufa) ? bufb : bufa;
    Tin = Tout;
  }
  const int T = (int)Tin;
  if (after_extractor) (*after_extractor)();
  // ---- LayerNorm(C) -> proj -> x + gelu(pos_conv(x)) -> LayerNorm(E)
  float* ln = y;
  launch_layernorm_c(x, m.ln0_g, m.ln0_b, ln, B, C, T, 1e-5f, nullptr, s);
  float* h = A.alloc<float>((size_t)B * E * T);
  float* h2 = A.alloc<float>((size_t)B * E * T);
  {
    ConvArgs a = conv1d_args(m.proj, ln, h, B, T, T);
    a.lens_out = d_T;
    c.conv_on(a, s);
    a = conv1d_args(m.pos_conv, h, h2, B, T, T, 1, 1, cf.pos_kernel / 2);   // SamePad: drop the last frame
    a.act = ACT_GELU;
    a.lens_in = a.lens_out = d_T;    // zero padding behind the item's last frame
    conv_set_res(a, h, E, T);
    c.conv_on(a, s);
  }
  const int nl = std::min(output_layer, cf.layers);
  const int hd = E / cf.heads;
  const float scale = 1.f / std::sqrt((float)hd);
  if (gemm_h3_enabled() && E % 16 == 0 && cf.ffn_dim % 16 == 0 && E <= 1024 && nl > 0) {
    // ---- post-LN transformer layers, TIME-MAJOR (gemm.hip): rows r = b T + t.  Between layers the activations
    // travel as fp32 rows (residuals, LayerNorm input) and in the split form the next GEMM stages.  A layer pinned to
    // fp32 (sticky range guard, conv.h) takes fp32 rows instead and runs gemm_f32_kernel.
    const long R = (long)B * T;
    const int F = cf.ffn_dim;
    float* hf = A.alloc<float>((size_t)R * E);
    float* hs = A.alloc<float>((size_t)R * E);          // split rows: same bytes as fp32
    float* tmp = A.alloc<float>((size_t)R * E);
    float* qkv = A.alloc<float>((size_t)R * 3 * E);     // channel-first (B, 3E, T): the attention kernels' layout
    float* att = A.alloc<float>((size_t)R * E);         // channel-first (B, E, T)
    float* attr = A.alloc<float>((size_t)R * E);        // attention output as rows (split or fp32)
    float* ff = A.alloc<float>((size_t)R * F);          // FFN intermediate rows (split or fp32)
    float* asplit = A.alloc<float>(attention_split_floats(B, cf.heads, T));
    const long ldE = E, ldEb = (long)E * 4, ldFb = (long)F * 4;
    auto first_use = [&](const ConvW& w) { return conv_h3_ok(w); };
    launch_cf_to_tm(h2, (long)E * T, tmp, ldE, nullptr, 0, B, E, T, nullptr, nullptr, 0, s);
    {
      const bool q3 = first_use(m.layers[0].qkv);
      launch_layernorm_tm(tmp, ldE, m.eln_g, m.eln_b, hf, ldE, q3 ? hs : nullptr, ldEb, R, E, 1e-5f, c.dev_err,
                          m.layers[0].qkv.ovf_word, ++c.launch_seq, s);
    }
    for (int l = 0; l < nl; ++l) {
      const auto& L = m.layers[l];
      const bool q3 = conv_h3_ok(L.qkv), o3 = conv_h3_ok(L.o), f13 = conv_h3_ok(L.fc1), f23 = conv_h3_ok(L.fc2);
      GemmArgs g = gemm_args(L.qkv, R, T);
      if (q3) g.xs = hs, g.ld_xs = ldEb;
      else g.x = hf, g.ld_x = ldE;
      g.y_cf = qkv;
      g.cf_bs = (long)3 * E * T;
      g.lens = d_T;                  // rows behind an item's last frame are stored as zeros
      c.gemm_on(g, s);
      launch_attention(qkv, qkv + (size_t)E * T, qkv + (size_t)2 * E * T, att, B, cf.heads, hd, T, T, (long)3 * E * T,
                       (long)E * T, scale, nullptr, nullptr, 0, d_T, nullptr, asplit, s, c.dev_err, L.att.word,
                       ++c.launch_seq, L.att.h3());
      c.flops += attention_flops(B, cf.heads, hd, T);
      launch_cf_to_tm(att, (long)E * T, o3 ? nullptr : attr, ldE, o3 ? attr : nullptr, ldEb, B, E, T, c.dev_err,
                      L.o.ovf_word, ++c.launch_seq, s);
      g = gemm_args(L.o, R, T);
      g.lens = d_T;
      if (o3) g.xs = attr, g.ld_xs = ldEb;
      else g.x = attr, g.ld_x = ldE;
      g.res = hf;
      g.ld_res = ldE;
      g.y = tmp;
      g.ld_y = ldE;
      c.gemm_on(g, s);
      launch_layernorm_tm(tmp, ldE, L.ln1_g, L.ln1_b, hf, ldE, f13 ? hs : nullptr, ldEb, R, E, 1e-5f, c.dev_err,
                          L.fc1.ovf_word, ++c.launch_seq, s);
      g = gemm_args(L.fc1, R, T);
      g.lens = d_T;
      if (f13) g.xs = hs, g.ld_xs = ldEb;
      else g.x = hf, g.ld_x = ldE;
      g.act = ACT_GELU;
      if (f23) g.ys = ff, g.ld_ys = ldFb, g.ovf_next = L.fc2.ovf_word;
      else g.y = ff, g.ld_y = F;
      c.gemm_on(g, s);
      g = gemm_args(L.fc2, R, T);
      g.lens = d_T;
      if (f23) g.xs = ff, g.ld_xs = ldFb;
      else g.x = ff, g.ld_x = F;
      g.res = hf;
      g.ld_res = ldE;
      g.y = tmp;
      g.ld_y = ldE;
      c.gemm_on(g, s);
      if (l == nl - 1) {
        launch_layernorm_tm(tmp, ldE, L.ln2_g, L.ln2_b, hf, ldE, nullptr, 0, R, E, 1e-5f, nullptr, nullptr, 0, s);
        launch_transpose(hf, feats_ct, B, T, E, s);      // rows (B, T, E) -> channel-first (B, E, T)
      } else {
        const ConvW& nq = m.layers[l + 1].qkv;
        launch_layernorm_tm(tmp, ldE, L.ln2_g, L.ln2_b, hf, ldE, conv_h3_ok(nq) ? hs : nullptr, ldEb, R, E, 1e-5f,
                            c.dev_err, nq.ovf_word, ++c.launch_seq, s);
      }
    }
    RVCX_HIP(hipGetLastError());
    return;
  }
  // ---- post-LN transformer layers, channel-first (RVCX_GEMM=0, the exact-fp32 mode, odd geometries)
  launch_layernorm_c(h2, m.eln_g, m.eln_b, h, B, E, T, 1e-5f, nullptr, s);
  float* qkv = A.alloc<float>((size_t)B * 3 * E * T);
  float* att = A.alloc<float>((size_t)B * E * T);
  float* ff = A.alloc<float>((size_t)B * cf.ffn_dim * T);
  float* asplit = A.alloc<float>(attention_split_floats(B, cf.heads, T));
  for (int l = 0; l < nl; ++l) {
    const auto& L = m.layers[l];
    ConvArgs a = conv1d_args(L.qkv, h, qkv, B, T, T);
    a.lens_out = d_T;
    c.conv_on(a, s);
    launch_attention(qkv, qkv + (size_t)E * T, qkv + (size_t)2 * E * T, att, B, cf.heads, hd, T, T,
                     (long)3 * E * T, (long)E * T, scale, nullptr, nullptr, 0, d_T, nullptr, asplit, s, c.dev_err, L.att.word,
                     ++c.launch_seq, L.att.h3());
    c.flops += attention_flops(B, cf.heads, hd, T);
    a = conv1d_args(L.o, att, h2, B, T, T);
    conv_set_res(a, h, E, T);
    c.conv_on(a, s);
    launch_layernorm_c(h2, L.ln1_g, L.ln1_b, h, B, E, T, 1e-5f, nullptr, s);
    a = conv1d_args(L.fc1, h, ff, B, T, T);
    a.act = ACT_GELU;
    c.conv_on(a, s);
    a = conv1d_args(L.fc2, ff, h2, B, T, T);
    conv_set_res(a, h, E, T);
    c.conv_on(a, s);
    launch_layernorm_c(h2, L.ln2_g, L.ln2_b, (l == nl - 1) ? feats_ct : h, B, E, T, 1e-5f, nullptr, s);
  }
  if (nl == 0) RVCX_HIP(hipMemcpyAsync(feats_ct, h, (size_t)B * E * T * sizeof(float), hipMemcpyDeviceToDevice, s));
  RVCX_HIP(hipGetLastError());
}

void hubert_features_for(Ctx& c, const HubertModel& m, int out_dim, int B, const float* wav, int64_t n, float* feats_ct,
                         hipStream_t s, long wav_bs, const int* ns_host) {
  if (out_dim == m.cfg.embed_dim) {
    hubert_forward(c, m, B, wav, n, 12, feats_ct, s, nullptr, wav_bs, ns_host);
    return;
  }
  RVCX_CHECK(m.has_final_proj && m.final_proj.cout == out_dim,
             "the voice model's input width is neither the HuBERT's embed_dim (RVC v2) nor its final_proj width (RVC v1)");
  const int T = hubert_frames(m, n);
  float* l9 = c.arena.alloc<float>((size_t)B * m.cfg.embed_dim * T);
  hubert_forward(c, m, B, wav, n, 9, l9, s, nullptr, wav_bs, ns_host);
  ConvArgs a = conv1d_args(m.final_proj, l9, feats_ct, B, T, T);
  c.conv_on(a, s);
  RVCX_HIP(hipGetLastError());
}

}  // namespace rvcx
