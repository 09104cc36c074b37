// RMVPE F0 estimator (rvc/lib/predictors/RMVPE.py): conv-STFT -> mel -> Deep U-Net -> BiGRU ->
// 360-bin salience -> cents -> Hz.  Every conv (STFT basis, mel filterbank, 3x3 U-Net convs with
// folded eval-mode BatchNorm, polyphase transposed convs, GRU input projection, classifier) is an
// MFMA implicit-GEMM launch; 2-D maps are (B, C, T, 128+2) row-padded so 3x3 convs are flat 1-D convs.
#include <cmath>

#include "models.h"
#include "ops.h"

namespace rvcx {

namespace {

constexpr int N_FFT = 1024, HOP = 160, N_MELS = 128;
constexpr double SR = 16000.0, FMIN = 30.0, FMAX = 8000.0;

// RMVPE.py:46-66 forward_basis: [Re F[0..512] ; Im F[0..512]] * hann, float32
std::vector<float> stft_basis() {
  const int nb = N_FFT / 2 + 1;
  std::vector<float> w((size_t)2 * nb * N_FFT);
  for (int k = 0; k < nb; ++k)
    for (int n = 0; n < N_FFT; ++n) {
      const double ang = 2.0 * M_PI * (double)(k * n) / (double)N_FFT;
      const float win = (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * (double)n / (double)N_FFT));
      w[(size_t)k * N_FFT + n] = (float)std::cos(ang) * win;
      w[(size_t)(nb + k) * N_FFT + n] = (float)(-std::sin(ang)) * win;
    }
  return w;
}

// librosa.filters.mel(sr=16000, n_fft=1024, n_mels=128, fmin=30, fmax=8000, htk=True) (RMVPE.py:395-402)
std::vector<float> mel_filterbank() {
  const int nb = N_FFT / 2 + 1;
  auto hz2mel = [](double f) { return 2595.0 * std::log10(1.0 + f / 700.0); };
  auto mel2hz = [](double m) { return 700.0 * (std::pow(10.0, m / 2595.0) - 1.0); };
  std::vector<double> melf(N_MELS + 2);
  const double m0 = hz2mel(FMIN), m1 = hz2mel(FMAX);
  for (int i = 0; i < N_MELS + 2; ++i) melf[i] = mel2hz(m0 + (m1 - m0) * (double)i / (double)(N_MELS + 1));
  std::vector<float> w((size_t)N_MELS * nb, 0.f);
  for (int i = 0; i < N_MELS; ++i) {
    const double enorm = 2.0 / (melf[i + 2] - melf[i]);
    for (int j = 0; j < nb; ++j) {
      const double fj = (SR / 2.0) * (double)j / (double)(nb - 1);
      const double lower = (fj - melf[i]) / (melf[i + 1] - melf[i]);
      const double upper = (melf[i + 2] - fj) / (melf[i + 2] - melf[i + 1]);
      const double v = std::max(0.0, std::min(lower, upper));
      w[(size_t)i * nb + j] = (float)(v * enorm);
    }
  }
  return w;
}

struct BnFold {
  std::vector<float> scale, shift;
};
BnFold bn_fold(const TensorTable& t, const std::string& p) {
  auto g = t.f32(p + ".weight"), b = t.f32(p + ".bias"), mu = t.f32(p + ".running_mean"),
       var = t.f32(p + ".running_var");
  BnFold f;
  f.scale.resize(g.size());
  f.shift.resize(g.size());
  for (size_t i = 0; i < g.size(); ++i) {
    const float s = g[i] / std::sqrt(var[i] + 1e-5f);
    f.scale[i] = s;
    f.shift[i] = b[i] - mu[i] * s;
  }
  return f;
}

// Conv2d 3x3 (no bias) followed by eval-mode BN -> one packed conv with bias
ConvW conv_bn(Ctx& c, const TensorTable& t, const std::string& conv, const std::string& bn) {
  auto w = t.f32(conv + ".weight");
  const auto shp = t.shape(conv + ".weight");
  const int cout = (int)shp[0], cin = (int)shp[1], kk = (int)(shp[2] * shp[3]);
  BnFold f = bn_fold(t, bn);
  for (int co = 0; co < cout; ++co)
    for (int i = 0; i < cin * kk; ++i) w[(size_t)co * cin * kk + i] *= f.scale[co];
  return make_conv(c, w.data(), f.shift.data(), cout, cin, kk, 1);
}

RmvpeModel::Block load_block(Ctx& c, const TensorTable& t, const std::string& p) {
  RmvpeModel::Block b;
  b.c1 = conv_bn(c, t, p + ".conv.0", p + ".conv.1");
  b.c2 = conv_bn(c, t, p + ".conv.3", p + ".conv.4");
  b.cin = b.c1.cin;
  b.cout = b.c1.cout;
  if (t.has(p + ".shortcut.weight")) {
    auto w = t.f32(p + ".shortcut.weight");
    auto bias = t.f32(p + ".shortcut.bias");
    b.sc = make_conv(c, w.data(), bias.data(), b.cout, b.cin, 1, 1);
    b.has_sc = true;
  }
  return b;
}

}  // namespace

// The strided STFT conv (Cin=1, k=1024, stride 160) is re-indexed k = 160*a + r so that it
// becomes a dense stride-1 conv with Cin=160, k=7 over the hop-major transposed signal
// X2[r][m] = x[160*m + r]: no wasted MFMA k-slots and a small LDS tile.  Shared with FCPE (same n_fft / hop / Hann).
ConvW make_stft_conv(Ctx& c) {
  auto b = stft_basis();
  const int KA = (N_FFT + HOP - 1) / HOP;  // 7
  std::vector<float> w2((size_t)(N_FFT + 2) * HOP * KA, 0.f);
  for (int co = 0; co < N_FFT + 2; ++co)
    for (int r = 0; r < HOP; ++r)
      for (int a = 0; a < KA; ++a)
        if (HOP * a + r < N_FFT) w2[((size_t)co * HOP + r) * KA + a] = b[(size_t)co * N_FFT + HOP * a + r];
  return make_conv(c, w2.data(), nullptr, N_FFT + 2, HOP, KA, 1);
}

std::unique_ptr<RmvpeModel> rmvpe_load(Ctx& c, const rvcx_rmvpe_cfg& cfg, const TensorTable& t) {
  auto M = std::make_unique<RmvpeModel>();
  RegionScope scope(c, *M->region);
  M->cfg = cfg;
  {
    M->stft = make_stft_conv(c);
    auto m = mel_filterbank();
    M->melfb = make_conv(c, m.data(), nullptr, N_MELS, N_FFT / 2 + 1, 1, 1);
  }
  {
    BnFold f = bn_fold(t, "unet.encoder.bn");
    std::vector<float> ss = {f.scale[0], f.shift[0]};
    M->bn0 = c.slab.upload(ss);
  }
  for (int l = 0; l < cfg.en_de_layers; ++l) {
    std::vector<RmvpeModel::Block> blocks;
    for (int b = 0; b < cfg.n_blocks; ++b)
      blocks.push_back(load_block(c, t, "unet.encoder.layers." + std::to_string(l) + ".conv." + std::to_string(b)));
    M->enc.push_back(std::move(blocks));
  }
  for (int l = 0; l < cfg.inter_layers; ++l) {
    std::vector<RmvpeModel::Block> blocks;
    for (int b = 0; b < cfg.n_blocks; ++b)
      blocks.push_back(
          load_block(c, t, "unet.intermediate.layers." + std::to_string(l) + ".conv." + std::to_string(b)));
    M->inter.push_back(std::move(blocks));
  }
  for (int l = 0; l < cfg.en_de_layers; ++l) {
    RmvpeModel::Dec D;
    const std::string p = "unet.decoder.layers." + std::to_string(l);
    auto w = t.f32(p + ".conv1.0.weight");
    const auto shp = t.shape(p + ".conv1.0.weight");  // (Cin, Cout, 3, 3)
    BnFold f = bn_fold(t, p + ".conv1.1");
    D.cout = (int)shp[1];
    D.up = make_convT2d(c, w.data(), f.scale.data(), f.shift.data(), (int)shp[0], (int)shp[1]);
    for (int b = 0; b < cfg.n_blocks; ++b) D.blocks.push_back(load_block(c, t, p + ".conv2." + std::to_string(b)));
    M->dec.push_back(std::move(D));
  }
  {
    auto w = t.f32("cnn.weight");
    auto b = t.f32("cnn.bias");
    const auto shp = t.shape("cnn.weight");
    M->cnn = make_conv(c, w.data(), b.data(), (int)shp[0], (int)shp[1], 9, 1);
  }
  {
    const std::string g = "fc.0.gru.";
    const auto shp = t.shape(g + "weight_ih_l0");  // (3H, I)
    const int H3 = (int)shp[0], I = (int)shp[1], H = H3 / 3;
    std::vector<float> wih, bih, whh_t((size_t)2 * H * H3), bhh;
    int d = 0;
    for (const char* sfx : {"", "_reverse"}) {
      auto w = t.f32(g + "weight_ih_l0" + sfx);
      auto b = t.f32(g + "bias_ih_l0" + sfx);
      wih.insert(wih.end(), w.begin(), w.end());
      bih.insert(bih.end(), b.begin(), b.end());
      auto wh = t.f32(g + "weight_hh_l0" + sfx);  // (3H, H)
      for (int j = 0; j < H3; ++j)
        for (int k = 0; k < H; ++k) whh_t[((size_t)d * H + k) * H3 + j] = wh[(size_t)j * H + k];
      auto bh = t.f32(g + "bias_hh_l0" + sfx);
      bhh.insert(bhh.end(), bh.begin(), bh.end());
      ++d;
    }
    M->gru_ih = make_conv(c, wih.data(), bih.data(), 2 * H3, I, 1, 1);
    M->whh_t = c.slab.upload(whh_t);
    M->bhh = c.slab.upload(bhh);
  }
  {
    auto w = t.f32("fc.1.weight");
    auto b = t.f32("fc.1.bias");
    const auto shp = t.shape("fc.1.weight");
    M->fc = make_conv(c, w.data(), b.data(), (int)shp[0], (int)shp[1], 1, 1);
  }
  M->region->seal();
  return M;
}

static int padded_frames(int F) {
  const int pad = std::min(32 * ((F - 1) / 32 + 1) - F, F);   // RMVPE.py:464
  return F + pad;
}

size_t rmvpe_arena_bytes(const RmvpeModel& m, int B, int64_t n) {
  const int F = (int)(1 + n / HOP), Tp = padded_frames(F);
  const int c0 = m.cfg.en_out_channels;
  size_t per = (size_t)Tp * 130;
  size_t tot = 2 * (size_t)(n + 1024 + 160) + (size_t)(1026 + 513 + 128) * F + per;
  // level-0 maps dominate: cat buffer (2*c0) + 3 work maps (c0 each) + halves at deeper levels (x2 bound)
  tot += 2 * (size_t)(2 * c0 + 4 * c0) * per;
  tot += (size_t)Tp * (384 + 1536 + 512 + 360 + 3 * 130);
  return (size_t)B * tot * sizeof(float) + ((size_t)64 << 20);
}

namespace {
struct Map2 {
  float* p = nullptr;
  long bs = 0;  // batch stride (floats)
  int C = 0, H = 0, Wp = 0;
};

// ConvBlockRes (RMVPE.py:140-175): relu(bn(conv2(relu(bn(conv1(x)))))) + shortcut(x)
// lens (device, B ints or null): valid flat positions (rows * Wp) per item on this level -- reads beyond see zeros, stores
// beyond are zeros: the zero padding a shorter item's single run sees below its last row
void run_block(Ctx& c, const RmvpeModel::Block& b, const Map2& x, const Map2& y, float* tmp1, float* tmp2, int B,
               hipStream_t s, const int* lens) {
  const int H = x.H, Wp = x.Wp;
  ConvArgs a = conv2d_args(b.c1, x.p, tmp1, B, H, Wp);
  a.x_bs = x.bs;
  a.act = ACT_RELU;
  a.lens_in = a.lens_out = lens;
  // conv1 -> conv2 hand-off in split fp16 form on the large (shallow) levels; the deep levels keep fp32 because
  // their launches rely on split-K, which the split store does not go through
  ConvArgs a2 = conv2d_args(b.c2, tmp1, y.p, B, H, Wp);
  a2.lens_in = a2.lens_out = lens;
  // (round 6: a block both of whose convs the weight-stationary tile takes -- 64 channels and up -- stays on fp32 maps)
  const bool ws = conv_deep_ok(a) && conv_deep_ok(a2);
  const bool split = !ws && (long)H * Wp >= 20000 && conv_h3_split_ok(a) && conv_h3_split_ok(a2) && !getenv("RVCX_NO_SPLIT");
  if (split) {
    a.y_split = tmp1;
    a.y = nullptr;
  }
  c.conv_on(a, s);
  const float* res = x.p;
  long res_bs = x.bs;
  if (b.has_sc) {
    a = conv2d_args(b.sc, x.p, tmp2, B, H, Wp);
    a.x_bs = x.bs;
    a.lens_in = a.lens_out = lens;
    c.conv_on(a, s);
    res = tmp2;
    res_bs = (long)b.cout * H * Wp;
  }
  a = a2;
  if (split) a.x_split = tmp1;
  a.act = ACT_RELU;
  a.res = res;
  a.res_bs = res_bs;
  a.res_cs = H * Wp;
  a.y_bs = y.bs;
  c.conv_on(a, s);
}
}  // namespace

// One ConvBlockRes on device maps (parity tests of the block path: the split hand-off, per-item row counts): x / y are
// (B, C, H, Wp) row-padded maps, rows[b] (host, or null) the valid rows of item b, tmp1 / tmp2 two maps of y's size
void rmvpe_block_op(Ctx& c, const ConvW& c1, const ConvW& c2, const ConvW* sc, const float* x, float* y, float* tmp1,
                    float* tmp2, int B, int H, int Wp, const int* rows, hipStream_t s) {
  RmvpeModel::Block blk;
  blk.c1 = c1;
  blk.c2 = c2;
  blk.cin = c1.cin;
  blk.cout = c1.cout;
  if (sc) {
    blk.sc = *sc;
    blk.has_sc = true;
  }
  const int* d_len = nullptr;
  if (rows) {
    std::vector<int> v(B);
    for (int b = 0; b < B; ++b) v[b] = rows[b] * Wp;
    d_len = dev_ints_many(c.arena, {v}, s)[0];
  }
  Map2 xm{const_cast<float*>(x), (long)blk.cin * H * Wp, blk.cin, H, Wp};
  Map2 ym{y, (long)blk.cout * H * Wp, blk.cout, H, Wp};
  run_block(c, blk, xm, ym, tmp1, tmp2, B, s, d_len);
}

void rmvpe_forward(Ctx& c, const RmvpeModel& m, int B, const float* audio, int64_t n, float thred, float f0_min,
                   float f0_max, float* f0, float* hidden, hipStream_t s, float* mel_out,
                   const std::function<void()>* after_shallow, const int* ns_host) {
  Arena& A = c.arena;
  const int F = (int)(1 + n / HOP), Tp = padded_frames(F);
  const int nenc = m.cfg.en_de_layers;
  // where the caller's hook (pipeline.hip: "now enqueue HuBERT") runs: behind encoder level 0 .. nenc-1, behind the
  // intermediate layers (nenc), or behind the whole U-Net, in front of the recurrence (nenc + 1 and above; the default)
  static const int hook_env = getenv("RVCX_HUBERT_AFTER") ? atoi(getenv("RVCX_HUBERT_AFTER")) : 99;
  // RVCX_HUBERT_AFTER_DEC = d (round 6 probe): behind decoder level d of the U-Net instead (0 = the deepest; nenc - 1 = behind
  // the whole U-Net, the default's place)
  static const int hook_dec = getenv("RVCX_HUBERT_AFTER_DEC") ? atoi(getenv("RVCX_HUBERT_AFTER_DEC")) : -1;
  const bool dec_hook = hook_dec >= 0 && hook_dec < nenc - 1;
  const int hook_at = dec_hook ? 1000 : std::min(hook_env, nenc + 1);
  RVCX_CHECK(Tp % (1 << nenc) == 0, "rmvpe: clip too short for the U-Net depth");
  const int nb = N_FFT / 2 + 1;
  // ---- ragged batch: item b holds ns_host[b] <= n samples.  The launch geometry is that of n for every item; the
  // arithmetic of item b is that of a clip of ns_host[b] samples (its own frame count, reflect padding, zero rows
  // below its last U-Net row, recurrence length) -- per-item length arrays do the rest.
  bool ragged = false;
  std::vector<int> Fb(B, F), Tb(B, Tp);
  if (ns_host)
    for (int b = 0; b < B; ++b) {
      RVCX_CHECK(ns_host[b] > N_FFT / 2 && ns_host[b] <= n, "rmvpe: item length outside (512, n]");
      Fb[b] = 1 + ns_host[b] / HOP;
      Tb[b] = padded_frames(Fb[b]);
      RVCX_CHECK(Tb[b] <= Tp && Tb[b] % (1 << nenc) == 0, "rmvpe: item frames exceed the batch geometry");
      ragged |= ns_host[b] != n;
    }
  const int *d_ns = nullptr, *d_F = nullptr, *d_T = nullptr;
  std::vector<const int*> d_lv(nenc + 1, nullptr);     // valid flat positions per U-Net level
  if (ragged) {
    std::vector<std::vector<int>> all{std::vector<int>(ns_host, ns_host + B), Fb, Tb};
    for (int l = 0; l <= nenc; ++l) {
      std::vector<int> v(B);
      for (int b = 0; b < B; ++b) v[b] = (Tb[b] >> l) * ((N_MELS >> l) + 2);
      all.push_back(v);
    }
    const std::vector<int*> d = dev_ints_many(A, all, s);     // one launch
    d_ns = d[0];
    d_F = d[1];
    d_T = d[2];
    for (int l = 0; l <= nenc; ++l) d_lv[l] = d[3 + l];
  }
  // ---- mel
  const int Mh = cdiv((int)n + N_FFT, HOP);     // hops covering the reflect-padded signal
  float* apad = A.alloc<float>((size_t)B * Mh * HOP);
  RVCX_HIP(hipMemsetAsync(apad, 0, (size_t)B * Mh * HOP * sizeof(float), s));
  launch_reflect_pad(audio, apad, B, (int)n, N_FFT / 2, (long)Mh * HOP, s, d_ns);
  float* x2 = A.alloc<float>((size_t)B * Mh * HOP);
  launch_transpose(apad, x2, B, Mh, HOP, s);    // (B, Mh, 160) -> (B, 160, Mh)
  float* ft = A.alloc<float>((size_t)B * 2 * nb * F);
  {
    ConvArgs a = conv1d_args(m.stft, x2, ft, B, Mh, F, 1, 1, 0);
    c.conv_on(a, s);
  }
  float* mag = A.alloc<float>((size_t)B * nb * F);
  launch_magnitude(ft, mag, B, nb, F, s);
  float* mel = A.alloc<float>((size_t)B * N_MELS * F);
  {
    ConvArgs a = conv1d_args(m.melfb, mag, mel, B, F, F);
    c.conv_on(a, s);
  }
  // ---- U-Net
  std::vector<int> Hs(nenc + 1), Ws(nenc + 1), Cs(nenc + 1);
  for (int l = 0; l <= nenc; ++l) {
    Hs[l] = Tp >> l;
    Ws[l] = (N_MELS >> l) + 2;
    Cs[l] = m.cfg.en_out_channels << l;
  }
  // concat buffers of the decoder, one per level: [upsampled | skip]
  std::vector<Map2> cat(nenc);
  for (int l = 0; l < nenc; ++l) {
    cat[l].C = 2 * Cs[l];
    cat[l].H = Hs[l];
    cat[l].Wp = Ws[l];
    cat[l].bs = (long)2 * Cs[l] * Hs[l] * Ws[l];
    cat[l].p = A.alloc<float>((size_t)B * cat[l].bs);
  }
  const size_t big = (size_t)B * std::max(Cs[0], 2) * Hs[0] * Ws[0];
  float* w0 = A.alloc<float>(big);
  float* w1 = A.alloc<float>(big);
  float* w2 = A.alloc<float>(big);
  float* w3 = A.alloc<float>(big);
  Map2 x{w0, (long)Hs[0] * Ws[0], 1, Hs[0], Ws[0]};
  if (mel_out) launch_log_clamp(mel, mel_out, (long)B * N_MELS * F, 1e-5f, s);   // log(clamp(mel, 1e-5)), RMVPE.py:438
  launch_mel_post(mel, x.p, B, N_MELS, F, Tp, m.bn0, s, d_F, d_T);
  for (int l = 0; l < nenc; ++l) {
    const int nblk = (int)m.enc[l].size();
    for (int b = 0; b < nblk; ++b) {
      Map2 y;
      y.C = Cs[l];
      y.H = Hs[l];
      y.Wp = Ws[l];
      if (b == nblk - 1) {  // skip connection lands directly in the decoder's concat buffer
        y.p = cat[l].p + (size_t)Cs[l] * Hs[l] * Ws[l];
        y.bs = cat[l].bs;
      } else {
        y.p = (x.p == w0) ? w3 : w0;
        y.bs = (long)Cs[l] * Hs[l] * Ws[l];
      }
      run_block(c, m.enc[l][b], x, y, w1, w2, B, s, d_lv[l]);
      x = y;
    }
    // 2x2 average pool (RMVPE.py:195)
    Map2 y{(x.p == w0) ? w3 : w0, (long)Cs[l] * Hs[l + 1] * Ws[l + 1], Cs[l], Hs[l + 1], Ws[l + 1]};
    if (B == 1 || x.bs == (long)Cs[l] * Hs[l] * Ws[l]) {
      launch_avgpool2(x.p, y.p, B * Cs[l], Hs[l], Ws[l], (long)Hs[l] * Ws[l], (long)Hs[l + 1] * Ws[l + 1], s);
    } else {
      for (int b = 0; b < B; ++b)
        launch_avgpool2(x.p + (size_t)b * x.bs, y.p + (size_t)b * y.bs, Cs[l], Hs[l], Ws[l], (long)Hs[l] * Ws[l],
                        (long)Hs[l + 1] * Ws[l + 1], s);
    }
    x = y;
    if (after_shallow && hook_at < nenc && l == std::min(hook_at, nenc - 1)) (*after_shallow)();
  }
  for (const auto& layer : m.inter)
    for (const auto& blk : layer) {
      Map2 y{(x.p == w0) ? w3 : w0, (long)blk.cout * Hs[nenc] * Ws[nenc], blk.cout, Hs[nenc], Ws[nenc]};
      run_block(c, blk, x, y, w1, w2, B, s, d_lv[nenc]);
      x = y;
    }
  if (after_shallow && hook_at == nenc) (*after_shallow)();
  for (int l = 0; l < nenc; ++l) {
    const int lv = nenc - 1 - l;
    const auto& D = m.dec[l];
    ConvArgs a = convT2d_args(D.up, x.p, cat[lv].p, B, x.H, x.Wp);
    a.x_bs = x.bs;
    a.y_bs = cat[lv].bs;
    a.act = ACT_RELU;
    a.lens_in = d_lv[lv + 1];        // the polyphase conv runs on the low-resolution grid: an item's rows there
    a.lens_out = d_lv[lv + 1];
    c.conv_on(a, s);
    x = cat[lv];
    for (const auto& blk : D.blocks) {
      Map2 y{(x.p == w0) ? w3 : w0, (long)blk.cout * Hs[lv] * Ws[lv], blk.cout, Hs[lv], Ws[lv]};
      run_block(c, blk, x, y, w1, w2, B, s, d_lv[lv]);
      x = y;
    }
    if (after_shallow && dec_hook && l == hook_dec) (*after_shallow)();
  }
  if (after_shallow && hook_at == nenc + 1) (*after_shallow)();
  // ---- cnn -> BiGRU -> Linear -> sigmoid   (RMVPE.py:373-376)
  float* cnn = w1;
  {
    ConvArgs a = conv2d_args(m.cnn, x.p, cnn, B, Hs[0], Ws[0]);
    a.x_bs = x.bs;
    a.lens_in = a.lens_out = d_lv[0];
    c.conv_on(a, s);
  }
  const int I = 3 * N_MELS, H = 256;
  float* gin = A.alloc<float>((size_t)B * I * Tp);
  launch_gru_input(cnn, gin, B, 3, Tp, Ws[0], s);
  float* gi = A.alloc<float>((size_t)B * Tp * 6 * H);
  {
    ConvArgs a = conv1d_args(m.gru_ih, gin, gi, B, Tp, Tp);
    a.out_mode = OUT_TRANSPOSED;
    a.y_bs = (long)Tp * 6 * H;
    a.y_cs = 6 * H;
    c.conv_on(a, s);
  }
  float* gy = A.alloc<float>((size_t)B * 2 * H * Tp);
  void* gscr = A.alloc<unsigned long long>(bigru_scratch_bytes(B) / 8);
  launch_bigru(gi, m.whh_t, m.bhh, gy, B, Tp, H, gscr, c.dev_err, s, d_T);
  float* sal = A.alloc<float>((size_t)B * Tp * 360);
  {
    ConvArgs a = conv1d_args(m.fc, gy, sal, B, Tp, Tp);
    a.lens_in = d_T;                 // the recurrence left frames beyond an item's length unwritten
    a.act = ACT_SIGMOID;
    a.out_mode = OUT_TRANSPOSED;
    a.y_bs = (long)Tp * 360;
    a.y_cs = 360;
    c.conv_on(a, s);
  }
  // hidden[:, :F] -> decode
  if (hidden) launch_copy_strided(sal, hidden, B, (long)F * 360, (long)Tp * 360, (long)F * 360, s);
  if (B == 1) {
    launch_decode_f0(sal, f0, 1, Fb[0], 360, thred, f0_min, f0_max, s);
  } else {
    for (int b = 0; b < B; ++b)
      launch_decode_f0(sal + (size_t)b * Tp * 360, f0 + (size_t)b * F, 1, Fb[b], 360, thred, f0_min, f0_max, s);
  }
  RVCX_HIP(hipGetLastError());
}

}  // namespace rvcx
