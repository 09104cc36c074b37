// Synthesizer.infer (rvc/lib/algorithm/synthesizers.py:163-188): TextEncoder -> z sampling ->
// reverse ResidualCouplingBlock -> GeneratorNSF, as a sequence of MFMA conv launches with fused
// epilogues plus a few memory-bound glue kernels.  Layout: (B, C, T) channel-first, T contiguous.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "models.h"
#include "ops.h"

namespace rvcx {

std::vector<float> wn_weight(const TensorTable& t, const std::string& p, int dim) {
  if (t.has(p + ".weight")) return t.f32(p + ".weight");
  std::string gn = p + ".parametrizations.weight.original0", vn = p + ".parametrizations.weight.original1";
  if (!t.has(gn)) {
    gn = p + ".weight_g";
    vn = p + ".weight_v";
  }
  std::vector<float> g = t.f32(gn), v = t.f32(vn);
  const auto shp = t.shape(vn);
  RVCX_CHECK(shp.size() == 3, "weight-norm tensor must be 3-D: " + p);
  const int64_t d0 = shp[0], d1 = shp[1], d2 = shp[2];
  std::vector<float> w(v.size());
  if (dim == 0) {
    RVCX_CHECK((int64_t)g.size() == d0, "weight_g shape: " + p);
    for (int64_t a = 0; a < d0; ++a) {
      double ss = 0.0;
      for (int64_t i = 0; i < d1 * d2; ++i) ss += (double)v[a * d1 * d2 + i] * v[a * d1 * d2 + i];
      const float sc = g[a] / (float)std::sqrt(ss);
      for (int64_t i = 0; i < d1 * d2; ++i) w[a * d1 * d2 + i] = v[a * d1 * d2 + i] * sc;
    }
  } else {  // dim == 2 (HuBERT pos_conv): norm over dims (0,1) per k
    RVCX_CHECK((int64_t)g.size() == d2, "weight_g shape: " + p);
    for (int64_t k = 0; k < d2; ++k) {
      double ss = 0.0;
      for (int64_t i = 0; i < d0 * d1; ++i) ss += (double)v[i * d2 + k] * v[i * d2 + k];
      const float sc = g[k] / (float)std::sqrt(ss);
      for (int64_t i = 0; i < d0 * d1; ++i) w[i * d2 + k] = v[i * d2 + k] * sc;
    }
  }
  return w;
}

static ConvW load_conv(Ctx& c, const TensorTable& t, const std::string& p, bool wn = false, bool bias = true,
                       bool h3 = true) {
  std::vector<float> w = wn ? wn_weight(t, p) : t.f32(p + ".weight");
  const auto shp = wn ? t.shape(t.has(p + ".weight") ? p + ".weight"
                                : (t.has(p + ".weight_v") ? p + ".weight_v"
                                                          : p + ".parametrizations.weight.original1"))
                      : t.shape(p + ".weight");
  const int cout = (int)shp[0], cin = (int)shp[1], k = shp.size() > 2 ? (int)shp[2] : 1;
  std::vector<float> b;
  if (bias && t.has(p + ".bias")) b = t.f32(p + ".bias");
  return make_conv(c, w.data(), b.empty() ? nullptr : b.data(), cout, cin, k, 1, h3);
}

std::unique_ptr<SynthModel> synth_load(Ctx& c, const rvcx_synth_cfg& cfg, const TensorTable& t) {
  auto M = std::make_unique<SynthModel>();
  RegionScope scope(c, *M->region);
  M->cfg = cfg;
  M->upp = 1;
  for (int i = 0; i < cfg.n_ups; ++i) M->upp *= cfg.up_rates[i];
  const int hid = cfg.hidden_channels;
  RVCX_CHECK(cfg.n_resblocks <= 4 && cfg.n_ups <= 6, "synth cfg out of range");
  // ---- enc_p
  M->emb_phone = load_conv(c, t, "enc_p.emb_phone");
  M->emb_pitch = c.slab.upload(t.f32("enc_p.emb_pitch.weight"));
  for (int i = 0; i < cfg.n_layers; ++i) {
    SynthModel::EncLayer L;
    const std::string a = "enc_p.encoder.attn_layers." + std::to_string(i);
    // fused q/k/v projection: one (3*hid, hid) 1x1 conv
    std::vector<float> w, b;
    for (const char* n : {".conv_q", ".conv_k", ".conv_v"}) {
      auto wi = t.f32(a + n + ".weight");
      auto bi = t.f32(a + n + ".bias");
      w.insert(w.end(), wi.begin(), wi.end());
      b.insert(b.end(), bi.begin(), bi.end());
    }
    L.qkv = make_conv(c, w.data(), b.data(), 3 * hid, hid, 1, 1);
    L.att = make_att_flag(c);
    L.o = load_conv(c, t, a + ".conv_o");
    L.rel_k = c.slab.upload(t.f32(a + ".emb_rel_k"));
    L.rel_v = c.slab.upload(t.f32(a + ".emb_rel_v"));
    RVCX_CHECK(t.shape(a + ".emb_rel_k")[1] == 21, "rel-pos window must be 10");
    const std::string is = std::to_string(i);
    L.g1 = c.slab.upload(t.f32("enc_p.encoder.norm_layers_1." + is + ".gamma"));
    L.b1 = c.slab.upload(t.f32("enc_p.encoder.norm_layers_1." + is + ".beta"));
    L.g2 = c.slab.upload(t.f32("enc_p.encoder.norm_layers_2." + is + ".gamma"));
    L.b2 = c.slab.upload(t.f32("enc_p.encoder.norm_layers_2." + is + ".beta"));
    L.ffn1 = load_conv(c, t, "enc_p.encoder.ffn_layers." + is + ".conv_1");
    L.ffn2 = load_conv(c, t, "enc_p.encoder.ffn_layers." + is + ".conv_2");
    M->enc.push_back(L);
  }
  M->proj = load_conv(c, t, "enc_p.proj");
  // ---- flow
  for (int f = 0; f < 4; ++f) {
    const std::string p = "flow.flows." + std::to_string(2 * f);
    auto& F = M->flows[f];
    F.pre = load_conv(c, t, p + ".pre");
    F.post = load_conv(c, t, p + ".post");
    F.cond = load_conv(c, t, p + ".enc.cond_layer", true);
    for (int i = 0; i < 3; ++i) {
      F.in_l[i] = load_conv(c, t, p + ".enc.in_layers." + std::to_string(i), true);
      F.rs_l[i] = load_conv(c, t, p + ".enc.res_skip_layers." + std::to_string(i), true);
    }
  }
  // ---- dec
  {
    auto lw = t.f32("dec.m_source.l_linear.weight");
    auto lb = t.f32("dec.m_source.l_linear.bias");
    std::vector<float> wb = {lw[0], lb[0]};
    M->lin_wb = c.slab.upload(wb);
  }
  M->conv_pre = load_conv(c, t, "dec.conv_pre", false, true, true);
  M->cond = load_conv(c, t, "dec.cond");
  M->conv_post = load_conv(c, t, "dec.conv_post", false, false, true);
  int ch = cfg.up_initial_channel;
  for (int i = 0; i < cfg.n_ups; ++i) {
    SynthModel::Stage S;
    const int u = cfg.up_rates[i], k = cfg.up_kernels[i], co = cfg.up_initial_channel >> (i + 1);
    const std::string p = "dec.ups." + std::to_string(i);
    // ConvTranspose1d weight (Cin, Cout, K), weight-norm over dim 0 (= Cin)
    std::vector<float> w = wn_weight(t, p, 0);
    std::vector<float> b = t.f32(p + ".bias");
    S.up = make_convT1d(c, w.data(), b.data(), ch, co, k, u, (k - u) / 2);
    int sf0 = 1;
    for (int j = i + 1; j < cfg.n_ups; ++j) sf0 *= cfg.up_rates[j];
    S.noise = load_conv(c, t, "dec.noise_convs." + std::to_string(i));
    S.noise_stride = sf0;
    S.noise_pad = sf0 > 1 ? sf0 / 2 : 0;
    if (sf0 >= 16 && S.noise.k == 2 * sf0 && S.noise.cin == 1) {
      // y[co][q] = sum_j w[co][j] har[sf0 q + j - sf0/2]: with j - sf0/2 = sf0 (m - 1) + c (c in [0, sf0), m in {0, 1, 2}) this
      // is a dense conv over q with three taps and sf0 input channels -- on the matrix cores instead of an 80-tap FIR whose
      // loads are 40 samples apart between neighbouring lanes (224 us per clip, 3 ms per 16-item micro-batch)
      const std::vector<float> w1 = t.f32("dec.noise_convs." + std::to_string(i) + ".weight");     // (co, 1, 2 sf0)
      const std::vector<float> b1 = t.f32("dec.noise_convs." + std::to_string(i) + ".bias");
      const int sp = (sf0 + 15) / 16 * 16, kn = 2 * sf0, half = sf0 / 2;
      std::vector<float> w2((size_t)co * sp * 3, 0.f);
      for (int o = 0; o < co; ++o)
        for (int c2 = 0; c2 < sf0; ++c2)
          for (int m2 = 0; m2 < 3; ++m2) {
            const int j = sf0 * (m2 - 1) + c2 + half;
            if (j >= 0 && j < kn) w2[((size_t)o * sp + c2) * 3 + m2] = w1[(size_t)o * kn + j];
          }
      S.noise_dense = make_conv(c, w2.data(), b1.data(), co, sp, 3, 1, true);
      S.noise_dense_cin = sp;
    }
    S.ch = co;
    for (int j = 0; j < cfg.n_resblocks; ++j)
      for (int m = 0; m < 3; ++m) {
        const std::string rp = "dec.resblocks." + std::to_string(i * cfg.n_resblocks + j);
        S.c1[j][m] = load_conv(c, t, rp + ".convs1." + std::to_string(m), true, true, true);
        S.c2[j][m] = load_conv(c, t, rp + ".convs2." + std::to_string(m), true, true, true);
      }
    M->stages.push_back(S);
    ch = co;
  }
  M->emb_g = c.slab.upload(t.f32("emb_g.weight"));
  {
    // Receptive field of the decoder in frames of z, each side (nsf.py:100-144, residuals.py:15-62): conv_pre k = 7 -> 3
    // frames; stage i at R_i samples per frame: the ConvTranspose1d (k_i taps, stride s_i) reaches (k_i - 1) / s_i + 1 input
    // samples (rate R_{i-1}), a ResBlock1 of kernel k and dilations d reaches sum_d (k - 1) / 2 (d + 1) samples; conv_post
    // k = 7 -> 3 samples.  (The harmonic source and its noise convs are evaluated whole, they add nothing.)  48 k: 11.1.
    double rf = 3.0, rate = 1.0;
    for (int i = 0; i < cfg.n_ups; ++i) {
      rf += ((cfg.up_kernels[i] - 1) / cfg.up_rates[i] + 1) / rate;
      rate *= cfg.up_rates[i];
      int res = 0;
      for (int j = 0; j < cfg.n_resblocks; ++j) {
        int r = 0;
        for (int d = 0; d < 3; ++d) r += (cfg.res_kernels[j] - 1) / 2 * (cfg.res_dilations[j][d] + 1);
        res = std::max(res, r);
      }
      rf += res / rate;
    }
    rf += 3.0 / rate;
    M->dec_rf_frames = (int)std::ceil(rf) + 2;
  }
  M->region->seal();
  return M;
}

size_t synth_arena_bytes(const SynthModel& m, int B, int T) {
  const auto& cf = m.cfg;
  size_t enc = (size_t)B * T * (size_t)(cf.input_dim + 8 * cf.hidden_channels + 2 * cf.filter_channels +
                                        4 * cf.inter_channels + 64 + (8 * 98 + 2 * 96 + 8) * cf.n_heads);
  size_t dec = (size_t)B * T * m.upp * 5;  // har, noise, out, the hop-major copy of har for the dense noise conv (1.2x)
  size_t mx = 0, sum = (size_t)B * T * cf.up_initial_channel;
  long tt = T;
  for (size_t i = 0; i < m.stages.size(); ++i) {
    tt *= cf.up_rates[i];
    const size_t n = (size_t)B * tt * m.stages[i].ch;
    mx = std::max(mx, n);
    sum += n + 1024;
  }
  return (enc + dec + 2 * sum + 9 * mx) * sizeof(float) + ((size_t)64 << 20);   // sum twice: stage outputs + noise-conv outputs
}

namespace {

struct Timer3 {      // records the caller's 4 stage events on the main stream; never synchronises
  Ctx& c;
  hipEvent_t* e;
  Timer3(Ctx& cc, hipEvent_t* ev) : c(cc), e(ev) {}
  void mark(int i) {
    if (e) RVCX_HIP(hipEventRecord(e[i], c.stream));
  }
};

int* upload_ints(Ctx& c, const std::vector<int>& v) {
  int* d = c.arena.alloc<int>(v.size());
  RVCX_HIP(hipMemcpyAsync(d, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice, c.stream));
  return d;
}

}  // namespace

void synth_forward(Ctx& c, const SynthModel& m, const SynthIO& io, hipEvent_t* stage_ev) {
  const auto& cf = m.cfg;
  const int B = io.B, T = io.T, hid = cf.hidden_channels, inter = cf.inter_channels, filt = cf.filter_channels;
  const int heads = cf.n_heads, kc = hid / heads, half = inter / 2, gin = cf.gin_channels;
  hipStream_t s = c.stream;
  Arena& A = c.arena;
  Timer3 tm(c, stage_ev);
  tm.mark(0);

  // per-stage valid lengths (ragged batches); null when every item spans the full T
  bool ragged = false;
  if (io.lens_host)
    for (int b = 0; b < B; ++b) ragged |= io.lens_host[b] != T;
  std::vector<const int*> lens_stage(m.stages.size() + 2, nullptr);
  const int* lens = nullptr;
  if (ragged) {
    std::vector<int> l(io.lens_host, io.lens_host + B);
    lens = upload_ints(c, l);
    lens_stage[0] = lens;
    for (size_t i = 0; i < m.stages.size(); ++i) {
      for (auto& x : l) x *= cf.up_rates[i];
      lens_stage[i + 1] = upload_ints(c, l);
    }
  }
  std::vector<int> sidv(io.sid_host, io.sid_host + B);

  // ================================================================ NSF source branch (nsf.py:116-129)
  // The harmonic source and the four noise convs depend on f0 only: they run on a side stream beside the
  // (latency-bound, 192-channel) TextEncoder and flow, and each stage's ConvTranspose1d adds its noise-conv
  // output as the epilogue residual -- the same (up + bias) + (noise + bias) sum the reference forms, without
  // a separate read-modify-write pass over the upsampled activations.
  const long Tupp = (long)T * m.upp;
  float* har = A.alloc<float>((size_t)B * Tupp);
  std::vector<float*> nz(m.stages.size(), nullptr);
  // Stages whose ConvTranspose1d runs on the streaming kernel (convt_thin.hip: k = 4, stride 2, C = 128 / 64 -- the two
  // last stages of every RVC v2 decoder) evaluate their noise conv inside that kernel's epilogue from `har`: its output is
  // never written to HBM (196 MB per stage and 30 s clip, written here and read back there).
  std::vector<char> nz_fused(m.stages.size(), 0);
  for (size_t i = 0; i < m.stages.size(); ++i) {
    const auto& S = m.stages[i];
    ConvArgs probe = convT1d_args(S.up, nullptr, nullptr, 1, 1, cf.up_rates[i]);
    probe.pre_act = ACT_LRELU;
    probe.pre_slope = 0.1f;
    static const bool fuse_on = !getenv("RVCX_NOISE_FUSE") || atoi(getenv("RVCX_NOISE_FUSE")) != 0;
    // k <= 4: the two last stages, on the streaming kernel.  The shuffle-store epilogues of the tiled kernels implement the
    // fused taps too (RVCX_NOISE_FUSE_K=8 sends stage 1's 8-tap noise conv there), but measured it is a loss: 436 -> 1132 us
    // for that launch -- eight dependent source loads per output element in an epilogue that was already store-bound
    static const int fuse_max_k = getenv("RVCX_NOISE_FUSE_K") ? atoi(getenv("RVCX_NOISE_FUSE_K")) : 4;
    nz_fused[i] = fuse_on && S.noise.cin == 1 && S.noise.groups == 1 && S.noise.k >= 1 && S.noise.cout == S.ch &&
                  S.noise.k <= fuse_max_k;       // every ConvTranspose1d path implements the fused taps (conv_device.h, convt_thin.hip)
    (void)probe;
  }
  {
    hipStream_t sn = c.serial ? s : c.aux[0];
    if (sn != s) {
      RVCX_HIP(hipEventRecord(c.ev_src[0], s));
      RVCX_HIP(hipStreamWaitEvent(sn, c.ev_src[0], 0));
    }
    double* sc = A.alloc<double>((size_t)B * T * 2);
    launch_sine_source(io.pitchf, io.src_noise, har, B, T, m.upp, (float)cf.sr, m.lin_wb, lens, sc, sn);
    long tt = T;
    for (size_t i = 0; i < m.stages.size(); ++i) {
      const auto& S = m.stages[i];
      tt *= cf.up_rates[i];
      if (nz_fused[i]) continue;
      nz[i] = A.alloc<float>((size_t)B * S.ch * tt);
      static const bool dense_on = !getenv("RVCX_NOISE_DENSE") || atoi(getenv("RVCX_NOISE_DENSE")) != 0;
      if (dense_on && S.noise_dense_cin > 0 && Tupp == tt * S.noise_stride) {
        // hop-major transpose of the source (zero beyond an item's length already: sine_source_kernel), then the dense conv
        const int sp = S.noise_dense_cin, st = S.noise_stride;
        float* x2 = A.alloc<float>((size_t)B * sp * tt);
        RVCX_HIP(hipMemsetAsync(x2, 0, (size_t)B * sp * tt * sizeof(float), sn));
        for (int b = 0; b < B; ++b) launch_transpose(har + (size_t)b * Tupp, x2 + (size_t)b * sp * tt, 1, (int)tt, st, sn);
        ConvArgs a = conv1d_args(S.noise_dense, x2, nz[i], B, (int)tt, (int)tt, 1, 1, 1);
        a.lens_out = lens_stage[i + 1];
        c.conv_on(a, sn);
        continue;
      }
      ConvArgs a = conv1d_args(S.noise, har, nz[i], B, (int)Tupp, (int)tt, S.noise_stride, 1, S.noise_pad);
      a.lens_in = lens_stage[m.stages.size()];
      a.lens_out = lens_stage[i + 1];
      c.conv_on(a, sn);
    }
    if (sn != s) RVCX_HIP(hipEventRecord(c.ev_src[1], sn));
  }

  // ================================================================ TextEncoder
  float* x = A.alloc<float>((size_t)B * hid * T);
  {
    ConvArgs a = conv1d_args(m.emb_phone, io.phone_ct, x, B, T, T);
    a.lens_out = lens;      // (everything behind an item's last frame is masked below anyway: its tiles are skipped)
    c.conv(a);
    launch_embed_pitch(x, m.emb_pitch, io.pitch, B, hid, T, std::sqrt((float)hid), 0.1f, lens, s);
  }
  float* qkv = A.alloc<float>((size_t)B * 3 * hid * T);
  float* att = A.alloc<float>((size_t)B * hid * T);
  float* tmp = A.alloc<float>((size_t)B * hid * T);
  float* hbuf = A.alloc<float>((size_t)B * filt * T);
  float* scratch = A.alloc<float>(attention_scratch_floats(B, heads, T, 10));
  float* asplit = A.alloc<float>(attention_split_floats(B, heads, T));
  const float scale = 1.f / std::sqrt((float)kc);
  for (const auto& L : m.enc) {
    ConvArgs a = conv1d_args(L.qkv, x, qkv, B, T, T);
    a.lens_out = lens;
    c.conv(a);
    launch_attention(qkv, qkv + (size_t)hid * T, qkv + (size_t)2 * hid * T, att, B, heads, kc, T, T,
                     (long)3 * hid * T, (long)hid * T, scale, L.rel_k, L.rel_v, 10, lens, scratch, asplit, s, c.dev_err, L.att.word,
                     ++c.launch_seq, L.att.h3());
    c.flops += attention_flops(B, heads, kc, T);
    a = conv1d_args(L.o, att, tmp, B, T, T);
    a.lens_out = lens;
    conv_set_res(a, x, hid, T);
    c.conv(a);
    launch_layernorm_c(tmp, L.g1, L.b1, x, B, hid, T, 1e-5f, nullptr, s);
    // FFN (attentions.py:195-203): conv_1(pad(x*mask)) -> relu -> conv_2(pad(h*mask)) * mask
    const int k = L.ffn1.k, pl = (k - 1) / 2;
    a = conv1d_args(L.ffn1, x, hbuf, B, T, T, 1, 1, pl);
    a.act = ACT_RELU;
    a.lens_in = lens;
    a.lens_out = lens;
    c.conv(a);
    a = conv1d_args(L.ffn2, hbuf, tmp, B, T, T, 1, 1, pl);
    a.lens_in = lens;
    a.lens_out = lens;
    conv_set_res(a, x, hid, T);      // x + y; positions beyond len are never read by valid frames
    c.conv(a);
    launch_layernorm_c(tmp, L.g2, L.b2, x, B, hid, T, 1e-5f, nullptr, s);
  }
  float* stats = io.stats_out ? io.stats_out : A.alloc<float>((size_t)B * 2 * inter * T);
  {
    launch_mask(x, B, hid, T, lens, s);                       // x = x * x_mask (encoders.py:72)
    ConvArgs a = conv1d_args(m.proj, x, stats, B, T, T);
    a.lens_out = lens;
    c.conv(a);
  }
  tm.mark(1);

  // ================================================================ z_p and the reverse flow
  float* z = A.alloc<float>((size_t)B * inter * T);
  float* z2 = A.alloc<float>((size_t)B * inter * T);
  launch_sample_z(stats, io.z_noise, z, B, inter, T, lens, s);
  // speaker embedding g (B, gin)
  float* g = A.alloc<float>((size_t)B * gin);
  for (int b = 0; b < B; ++b)
    RVCX_HIP(hipMemcpyAsync(g + (size_t)b * gin, m.emb_g + (size_t)sidv[b] * gin, gin * sizeof(float),
                            hipMemcpyDeviceToDevice, s));
  {
    float* h = att;       // reuse (B,hid,T)
    float* xin = qkv;     // (B,2*hid,T) fits in the 3*hid buffer
    float* acts = tmp;
    float* rs = hbuf;     // needs 2*hid*T <= filt*T
    RVCX_CHECK(filt >= 2 * hid, "buffer reuse assumes filter_channels >= 2*hidden");
    float* wout = A.alloc<float>((size_t)B * hid * T);
    float* mbuf = A.alloc<float>((size_t)B * half * T);
    float* gc = A.alloc<float>((size_t)B * 6 * hid);
    for (int f = 3; f >= 0; --f) {
      const auto& F = m.flows[f];
      launch_flip_channels(z, z2, B, inter, T, s);
      std::swap(z, z2);
      ConvArgs a = conv1d_args(F.pre, z, h, B, T, T);
      a.x_bs = (long)inter * T;          // reads only the first `half` channels of z
      a.lens_out = lens;
      c.conv(a);
      a = conv1d_args(F.cond, g, gc, B, 1, 1);
      c.conv(a);
      for (int i = 0; i < 3; ++i) {
        a = conv1d_args(F.in_l[i], h, xin, B, T, T, 1, 1, (F.in_l[i].k - 1) / 2);
        a.lens_in = lens;
        a.lens_out = lens;
        c.conv(a);
        launch_wn_gate(xin, gc, i * 2 * hid, 6 * hid, acts, B, hid, T, s);
        a = conv1d_args(F.rs_l[i], acts, rs, B, T, T);
        a.lens_out = lens;
        c.conv(a);
        launch_wn_res_skip(h, wout, rs, B, hid, T, i == 2, i == 0, lens, s);
      }
      a = conv1d_args(F.post, wout, mbuf, B, T, T);
      a.lens_out = lens;
      c.conv(a);
      launch_coupling_sub(z, mbuf, B, half, T, lens, s);
    }
  }
  if (io.z_out)
    RVCX_HIP(hipMemcpyAsync(io.z_out, z, (size_t)B * inter * T * sizeof(float), hipMemcpyDeviceToDevice, s));
  tm.mark(2);
  if (io.ev_decoder) RVCX_HIP(hipEventRecord(io.ev_decoder, s));

  // ================================================================ NSF-HiFi-GAN decoder
  // The decoder runs `db` utterances at a time (default ONE) even inside a batch: its activations are ~200 MB
  // per tensor and utterance, and at one utterance the producer -> consumer hand-off between consecutive layers is
  // largely served by the 256 MB Infinity Cache.  Measured with B = 8 in one launch sequence: 22.4 ms per clip
  // against 16.5 ms one at a time (every layer then streams from HBM) -- RVCX_DEC_BATCH overrides.
  if (!c.serial) RVCX_HIP(hipStreamWaitEvent(s, c.ev_src[1], 0));
  // Round 6: utterances of EQUAL length go through the decoder `RVCX_DEC_BATCH` at a time (default 8): the persistent fused
  // steps walk ceil(tiles / 256) rounds of tiles, 8.33 -> 9 for one 30 s utterance at C = 128, 33.3 -> 34 for four (C3: 1408 -
  // 1414x at 1, 1419 at 2, 1428 at 3, 1439 at 4 on one box, 1455 - 1459 at 4 / 8 and 1464 at 16 on another; the round-2 finding
  // above predates the fused steps).  Ragged members keep
  // one utterance at a time at its own length (no work on padding).  Results do not depend on the grouping (the fused steps
  // are per-position, every split decision is per item): batch == single holds bit for bit (tests/test_gpu_fullsize_batch.py).
  static const int dec_batch_env = getenv("RVCX_DEC_BATCH") ? std::max(1, atoi(getenv("RVCX_DEC_BATCH"))) : 8;
  bool equal_lens = true;
  if (io.lens_host)
    for (int b = 1; b < B; ++b) equal_lens &= io.lens_host[b] == io.lens_host[0];
  const int dec_batch = equal_lens ? dec_batch_env : 1;
  const int C0 = cf.up_initial_channel;
  const size_t dec_mark = A.mark();
  for (int b0 = 0; b0 < B; b0 += dec_batch) {
    const int db = std::min(dec_batch, B - b0);
    A.reset(dec_mark);
    // One utterance at a time -- or a group of EQUAL length -- runs at ITS OWN length Td (rows of the batched tensors stay T
    // apart): no masks, no work on padding, and every launch decision (tile, split-K) is the one the utterance's single run takes.
    const bool own = io.lens_host != nullptr && (db == 1 || equal_lens);
    const int Tfull = own ? io.lens_host[b0] : T;
    RVCX_CHECK(Tfull > 0 && Tfull <= T, "synth: item length outside (0, T]");
    // the window of frames the decoder evaluates (SynthIO::dec_skip): [skip, Tfull - skip) of every member of the group
    const int skip = (io.dec_skip > 0 && (own || !lens) && Tfull - 2 * io.dec_skip > 2 * m.dec_rf_frames) ? io.dec_skip : 0;
    const int Td = Tfull - 2 * skip;
    float* cur = A.alloc<float>((size_t)db * C0 * Td);
    {
      ConvArgs a = conv1d_args(m.conv_pre, z + (size_t)b0 * inter * T + skip, cur, db, Td, Td, 1, 1, 3);
      a.x_bs = (long)inter * T;
      a.x_cs = T;
      a.lens_in = (lens && !own) ? lens + b0 : nullptr;
      c.conv(a);
      float* gcond = A.alloc<float>((size_t)db * C0);
      a = conv1d_args(m.cond, g + (size_t)b0 * gin, gcond, db, 1, 1);
      c.conv(a);
      launch_add_channel_bias(cur, gcond, db, C0, Td, s);
    }
    long Tin = Td, Tin_c = T, off = skip;      // off: first sample of the window at this stage's rate
    for (size_t i = 0; i < m.stages.size(); ++i) {
      const auto& S = m.stages[i];
      const long Tout = Tin * cf.up_rates[i], Tout_c = Tin_c * cf.up_rates[i];
      off *= cf.up_rates[i];
      const size_t n = (size_t)db * S.ch * Tout;
      float* xs = A.alloc<float>(n);      // survives the stage (next stage's input)
      const size_t stage_mark = A.mark();
      float* xu = A.alloc<float>(n);
      const int nk = cf.n_resblocks;
      float* xcb[4];
      float* xtb[4];
      for (int j = 0; j < nk; ++j) {
        xcb[j] = A.alloc<float>(n);
        xtb[j] = A.alloc<float>(n);
      }
      const int* lin = (lens_stage[i] && !own) ? lens_stage[i] + b0 : nullptr;
      const int* lout = (lens_stage[i + 1] && !own) ? lens_stage[i + 1] + b0 : nullptr;
      ConvArgs a = convT1d_args(S.up, cur, xu, db, (int)Tin, (int)Tout);
      a.pre_act = ACT_LRELU;
      a.pre_slope = 0.1f;
      a.lens_in = lin;
      a.lens_out = lout;
      if (nz_fused[i]) {               // x = up(x) + noise_conv(har_source), the noise conv inside the epilogue
        a.nz_har = har + (size_t)b0 * Tupp + (size_t)skip * m.upp;
        a.nz_bs = Tupp;
        a.nz_w = S.noise.w;
        a.nz_wstride = S.noise.cin_gp * S.noise.cout_gp;
        a.nz_b = S.noise.bias;
        a.nz_k = S.noise.k;
        a.nz_stride = S.noise_stride;
        a.nz_pad = S.noise_pad;
        a.nz_len = (own || skip) ? (int)((long)Td * m.upp) : (int)Tupp;
        a.nz_lens = (lens_stage[m.stages.size()] && !own) ? lens_stage[m.stages.size()] + b0 : nullptr;
      } else {
        conv_set_res(a, nz[i] + (size_t)b0 * S.ch * Tout_c + off, S.ch, (int)Tout_c);   // x = up(x) + noise_conv(har_source)   (nsf.py:129)
      }
      c.conv(a);
      // xs = mean_j ResBlock1_j(x)   (nsf.py:131-139, residuals.py:45-53).  The nk blocks only share their
      // input, so block j runs on its own stream (main, aux0, aux1): their MFMA, staging and store phases
      // interleave on the CUs instead of marching in lockstep.  Only the running sum xs is ordered
      // (SET -> ADD -> ADD_DIV) through events.
      const int side = c.serial ? 0 : c.resblock_streams;   // 1: three streams; 2: main + aux[0]; 0: main only
      hipStream_t rs[4] = {s, side ? c.aux[0] : s, side == 1 ? c.aux[1] : s, s};
      RVCX_HIP(hipEventRecord(c.ev_aux[0], s));
      for (int j = 1; j < nk && j < 3; ++j) RVCX_HIP(hipStreamWaitEvent(rs[j], c.ev_aux[0], 0));
      for (int j = 0; j < nk; ++j) {
        hipStream_t sj = rs[j];
        const int k = cf.res_kernels[j];
        const float* xin = xu;
        float* xc = xcb[j];
        float* xt = xtb[j];
        // ---- the whole block in one kernel (resblock3.hip): kernel size 3 at C = 32 / 64 -- read x once, write the block's output once
        if (k == 3) {
          Block3Args ba;
          ba.x = xu;
          ba.y = nullptr;
          ba.y2 = xs;
          bool ok3 = true;
          for (int mi = 0; mi < 3; ++mi) {
            const auto &L1 = S.c1[j][mi], &L2 = S.c2[j][mi];
            ba.w1[mi] = (L1.w_h3 && L1.h3_ok && *L1.h3_ok) ? L1.w_h3 : nullptr;
            ba.w2[mi] = (L2.w_h3 && L2.h3_ok && *L2.h3_ok) ? L2.w_h3 : nullptr;
            ba.b1[mi] = L1.bias;
            ba.b2[mi] = L2.bias;
            ba.dil[mi] = cf.res_dilations[j][mi];
            ok3 = ok3 && L1.cin == S.ch && L1.cout == S.ch && L1.k == 3 && L2.cin == S.ch && L2.cout == S.ch && L2.k == 3;
          }
          ba.ovf_layer = S.c1[j][0].ovf_word;
          ba.lens = lout;
          ba.B = db;
          ba.C = S.ch;
          ba.T = (int)Tout;
          ba.bs = (long)S.ch * Tout;
          ba.cs = (int)Tout;
          ba.slope = 0.1f;
          ba.acc2_mode = j == 0 ? ACC2_SET : (j == nk - 1 ? ACC2_ADD_DIV : ACC2_ADD);
          ba.acc2_div = (float)nk;
          if (nk == 1) ba.acc2_mode = ACC2_SET;
          if (ok3 && resblock3_ok(ba)) {
            if (j > 0) RVCX_HIP(hipStreamWaitEvent(sj, c.ev_aux[j], 0));   // xs of block j-1 is complete
            c.block3_on(ba, sj);
            RVCX_HIP(hipEventRecord(c.ev_aux[j + 1], sj));
            continue;
          }
        }
        for (int mi = 0; mi < 3; ++mi) {
          const int d = cf.res_dilations[j][mi];
          // ---- fused step (resblock.hip): c1 -> c2 -> + x in one kernel, xt never leaves the CU
          {
            // x and y must be different buffers: a workgroup reads a halo of x that its neighbours own as output
            float* yout = (xin == xc) ? xt : xc;
            PairArgs pa;
            pa.x = xin;
            pa.y = yout;
            pa.w1 = (S.c1[j][mi].w_h3 && S.c1[j][mi].h3_ok && *S.c1[j][mi].h3_ok) ? S.c1[j][mi].w_h3 : nullptr;
            pa.w2 = (S.c2[j][mi].w_h3 && S.c2[j][mi].h3_ok && *S.c2[j][mi].h3_ok) ? S.c2[j][mi].w_h3 : nullptr;
            pa.b1 = S.c1[j][mi].bias;
            pa.b2 = S.c2[j][mi].bias;
            pa.ovf_layer = S.c1[j][mi].ovf_word;
            pa.lens = lout;
            pa.B = db;
            pa.C = S.ch;
            pa.T = (int)Tout;
            pa.bs = (long)S.ch * Tout;
            pa.cs = (int)Tout;
            pa.k = k;
            pa.dil = d;
            pa.slope = 0.1f;
            if (S.c1[j][mi].cin == S.ch && S.c1[j][mi].cout == S.ch && S.c2[j][mi].k == k && resblock_pair_ok(pa)) {
              if (mi == 2) {
                pa.y = nullptr;
                pa.y2 = xs;
                pa.acc2_mode = j == 0 ? ACC2_SET : (j == nk - 1 ? ACC2_ADD_DIV : ACC2_ADD);
                pa.acc2_div = (float)nk;
                if (nk == 1) pa.acc2_mode = ACC2_SET;
                if (j > 0) RVCX_HIP(hipStreamWaitEvent(sj, c.ev_aux[j], 0));   // xs of block j-1 is complete
              }
              c.pair_on(pa, sj);
              if (mi == 2) RVCX_HIP(hipEventRecord(c.ev_aux[j + 1], sj));
              xin = yout;
              continue;
            }
          }
          float* XT_ = (xin == xt) ? xc : xt;     // unfused fallback: c1 output / c2 output (c2 may run in place on xin)
          float* XC_ = (xin == xt) ? xt : xc;
          a = conv1d_args(S.c1[j][mi], xin, XT_, db, (int)Tout, (int)Tout, 1, d, (k * d - d) / 2);
          a.pre_act = ACT_LRELU;
          a.pre_slope = 0.1f;
          a.act = ACT_LRELU;       // the leaky_relu that precedes c2, applied once here
          a.act_slope = 0.1f;
          a.lens_in = lout;
          a.lens_out = lout;
          // c1 -> c2 hand-off: when both run on the split-fp16 kernels, XT_ travels already split (hi / scaled lo
          // halves in the consumer's LDS element order): c2 stages it with 16-byte loads and no conversion
          ConvArgs a2 = conv1d_args(S.c2[j][mi], XT_, XC_, db, (int)Tout, (int)Tout, 1, 1, (k - 1) / 2);
          const bool split = conv_h3_split_ok(a) && conv_h3_split_ok(a2) && !getenv("RVCX_NO_SPLIT");
          if (split) {
            a.y_split = XT_;
            a.y = nullptr;
          }
          c.conv_on(a, sj);
          a = a2;
          if (split) a.x_split = XT_;
          conv_set_res(a, xin, S.ch, (int)Tout);
          a.lens_in = lout;
          a.lens_out = lout;
          if (mi == 2) {           // last conv of the block: accumulate into xs, skip the plain store
            a.y = nullptr;
            a.y2 = xs;
            a.y2_bs = (long)S.ch * Tout;
            a.y2_cs = (int)Tout;
            a.acc2_mode = j == 0 ? ACC2_SET : (j == nk - 1 ? ACC2_ADD_DIV : ACC2_ADD);
            a.acc2_div = (float)nk;
            if (nk == 1) a.acc2_mode = ACC2_SET;
            if (j > 0) RVCX_HIP(hipStreamWaitEvent(sj, c.ev_aux[j], 0));   // xs of block j-1 is complete
          }
          c.conv_on(a, sj);
          if (mi == 2) RVCX_HIP(hipEventRecord(c.ev_aux[j + 1], sj));
          xin = XC_;
        }
      }
      if (rs[nk - 1] != s) RVCX_HIP(hipStreamWaitEvent(s, c.ev_aux[nk], 0));   // join before the next stage
      cur = xs;
      Tin = Tout;
      Tin_c = Tout_c;
      A.reset(stage_mark);
    }
    {
      // x = leaky_relu(x) [default slope 0.01, nsf.py:142]; tanh(conv_post(x))
      ConvArgs a = conv1d_args(m.conv_post, cur, io.out + (size_t)b0 * Tupp + (size_t)skip * m.upp, db, (int)Tin, (int)Tin, 1, 1, 3);
      a.pre_act = ACT_LRELU;
      a.pre_slope = 0.01f;
      a.act = ACT_TANH;
      a.lens_in = (lens_stage[m.stages.size()] && !own) ? lens_stage[m.stages.size()] + b0 : nullptr;
      a.lens_out = a.lens_in;
      a.y_bs = (long)Tupp;             // output rows are T * upp apart whatever the group's own length
      c.conv(a);
      if (skip + Td < T)               // the rest of each item's output row reads as silence
        for (int q = 0; q < db; ++q)
          RVCX_HIP(hipMemsetAsync(io.out + (size_t)(b0 + q) * Tupp + (size_t)(skip + Td) * m.upp, 0,
                                  (size_t)(T - skip - Td) * m.upp * sizeof(float), s));
      if (skip)                        // and so does what lies in front of the window
        for (int q = 0; q < db; ++q)
          RVCX_HIP(hipMemsetAsync(io.out + (size_t)(b0 + q) * Tupp, 0, (size_t)skip * m.upp * sizeof(float), s));
    }
  }
  tm.mark(3);
}

}  // namespace rvcx
