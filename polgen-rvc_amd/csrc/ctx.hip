// Context lifecycle, checkpoint-tensor helpers, packed-layer constructors.
#include <atomic>
#include <cmath>
#include <thread>
#include <cstring>

#include "ctx.h"
#include "layers.h"
#include "models.h"

namespace rvcx {

float half_to_float(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000) << 16;
  uint32_t exp = (h >> 10) & 0x1f, man = h & 0x3ff;
  uint32_t f;
  if (exp == 0) {
    if (man == 0) {
      f = sign;
    } else {
      exp = 127 - 15 + 1;
      while (!(man & 0x400)) {
        man <<= 1;
        --exp;
      }
      man &= 0x3ff;
      f = sign | (exp << 23) | (man << 13);
    }
  } else if (exp == 31) {
    f = sign | 0x7f800000u | (man << 13);
  } else {
    f = sign | ((exp + 127 - 15) << 23) | (man << 13);
  }
  float out;
  std::memcpy(&out, &f, 4);
  return out;
}

std::vector<float> TensorTable::f32(const std::string& name) const {
  const HostTensor& t = raw(name);
  const int64_t n = t.numel();
  std::vector<float> out((size_t)n);
  if (t.dtype == 0) {
    std::memcpy(out.data(), t.data, (size_t)n * 4);
  } else if (t.dtype == 1) {
    const uint16_t* p = static_cast<const uint16_t*>(t.data);
    for (int64_t i = 0; i < n; ++i) out[(size_t)i] = half_to_float(p[i]);
  } else if (t.dtype == 2) {
    const int64_t* p = static_cast<const int64_t*>(t.data);
    for (int64_t i = 0; i < n; ++i) out[(size_t)i] = (float)p[i];
  } else {
    fail("unsupported dtype for " + name);
  }
  return out;
}

Ctx::Ctx() {}

void Ctx::snapshot_dev_err(hipStream_t s) {
  if (!dev_err || !err_host) return;
  RVCX_HIP(hipMemcpyAsync(err_host, dev_err, sizeof(int), hipMemcpyDeviceToHost, s));
  err_snapshot = true;
}

void Ctx::check_dev_err() {
  if (!dev_err) return;
  int v = 0;
  if (err_snapshot) v = *err_host;
  else RVCX_HIP(hipMemcpy(&v, dev_err, sizeof(int), hipMemcpyDeviceToHost));
  if (inject_gru_timeout) {       // test hook (rvcx_debug_inject): behave as if the cluster kernel had timed out
    inject_gru_timeout = false;
    v |= kErrGruTimeout;
  }
  if (v & kErrGruTimeout) {
    v &= ~kErrGruTimeout;
    RVCX_HIP(hipMemcpy(dev_err, &v, sizeof(int), hipMemcpyHostToDevice));
    if (err_snapshot) *err_host = v;
    throw GruTimeout("device-side timeout: a GRU cluster workgroup lost its partner");
  }
}

bool Ctx::take_overflow() {
  if (!dev_err) return false;
  int v = 0;
  if (err_snapshot) v = *err_host;
  else RVCX_HIP(hipMemcpy(&v, dev_err, sizeof(int), hipMemcpyDeviceToHost));
  err_snapshot = false;            // the copy was this call's: the next reader asks the device
  if (!(v & kErrH3Overflow)) return false;
  v &= ~kErrH3Overflow;
  RVCX_HIP(hipMemcpy(dev_err, &v, sizeof(int), hipMemcpyHostToDevice));
  return true;
}

Ctx::~Ctx() {
  hubert.reset();
  rmvpe.reset();
  fcpe.reset();
  synths.clear();
  index.reset();
  if (timer.made)
    for (auto& e : timer.ev) (void)hipEventDestroy(e);
  if (dev_err) (void)hipFree(dev_err);
  if (err_host) (void)hipHostFree(err_host);
  for (auto* p : splitk_buf)
    if (p) (void)hipFree(p);
  if (ev_fork) (void)hipEventDestroy(ev_fork);
  for (auto e : ev_src)
    if (e) (void)hipEventDestroy(e);
  if (ev_join) (void)hipEventDestroy(ev_join);
  if (ev_io) (void)hipEventDestroy(ev_io);
  for (auto e : ev_front)
    if (e) (void)hipEventDestroy(e);
  for (auto e : ev_done)
    if (e) (void)hipEventDestroy(e);
  if (stream_io) (void)hipStreamDestroy(stream_io);
  for (auto e : ev_aux)
    if (e) (void)hipEventDestroy(e);
  for (auto s : aux)
    if (s) (void)hipStreamDestroy(s);
  if (stream2) (void)hipStreamDestroy(stream2);
  if (ev_hub) (void)hipEventDestroy(ev_hub);
  for (auto e : ev_hubdone)
    if (e) (void)hipEventDestroy(e);
  for (auto e : ev_syn)
    if (e) (void)hipEventDestroy(e);
  if (stream) (void)hipStreamDestroy(stream);
}

// fp16 hi/lo split of the packed weights for conv_h3_kernel: H3[kk][chunk of 16 ci][op (2)][h][co_pad][8 halves],
// op 0 = S*wh, 1 = S*wl  (S = 256; wh = fp16(w), wl = fp16((w - wh) * S) / S; the kernel re-derives wh = op0 / S).  Returns an empty vector
// when a weight would overflow fp16 at scale S (the layer then stays on the fp32 MFMA path).
static std::vector<float> pack_h3(const std::vector<float>& wp, int k, int cin_gp, int cout_gp) {
  const float S = 256.f;
  const int nchunk = cin_gp / 16;
  std::vector<float> out((size_t)k * nchunk * 4 * cout_gp * 4);          // 8 halves = 4 floats per element
  _Float16* h = reinterpret_cast<_Float16*>(out.data());
  const int slabs = k * nchunk;                                          // independent (tap, chunk) slabs
  std::atomic<bool> overflow{false};
  auto work = [&](int s0, int s1) {
    for (int sl = s0; sl < s1 && !overflow.load(std::memory_order_relaxed); ++sl) {
      const int kk = sl / nchunk, ch = sl % nchunk;
      _Float16* hs = h + (size_t)sl * 4 * cout_gp * 8;
      for (int hh = 0; hh < 2; ++hh)
        for (int q = 0; q < 8; ++q) {
          const float* src = wp.data() + ((size_t)kk * cin_gp + ch * 16 + hh * 8 + q) * cout_gp;
          for (int co = 0; co < cout_gp; ++co) {
            const float w = src[co];
            if (!(std::fabs(w) * S < 60000.f)) {
              overflow.store(true, std::memory_order_relaxed);
              return;
            }
            const _Float16 wh = (_Float16)w;
            hs[((size_t)(0 * 2 + hh) * cout_gp + co) * 8 + q] = (_Float16)((float)wh * S);
            hs[((size_t)(1 * 2 + hh) * cout_gp + co) * 8 + q] = (_Float16)((w - (float)wh) * S);
          }
        }
    }
  };
  const size_t total = (size_t)k * cin_gp * cout_gp;
  const int nthreads = total > (1u << 20) ? std::min<int>(8, std::max(1u, std::thread::hardware_concurrency())) : 1;
  if (nthreads <= 1) {
    work(0, slabs);
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t)
      th.emplace_back(work, (int)((long)slabs * t / nthreads), (int)((long)slabs * (t + 1) / nthreads));
    for (auto& t : th) t.join();
  }
  if (overflow.load()) return {};
  return out;
}

// The image's space is reserved from the layer's shape alone (same bytes as the fp32 weights); whether it may be
// used is a region flag, so ranks that load placeholder values build the same layout and learn the flag from
// the broadcast (zero-filled weight-norm tensors fold to NaN and used to shrink the layout).
static void upload_h3(Ctx& c, ConvW& L, const std::vector<float>& wp, int k) {
  // grouped layers: one image per group, back to back (pack_conv_weight lays the groups out as dense layers)
  const size_t per_w = (size_t)k * L.cin_gp * L.cout_gp, per_h = (size_t)k * (L.cin_gp / 16) * 4 * L.cout_gp * 4;
  std::vector<float> hp;
  if (L.groups == 1) {
    hp = pack_h3(wp, k, L.cin_gp, L.cout_gp);
  } else {
    for (int g = 0; g < L.groups; ++g) {
      const std::vector<float> one =
          pack_h3(std::vector<float>(wp.begin() + g * per_w, wp.begin() + (g + 1) * per_w), k, L.cin_gp, L.cout_gp);
      if (one.empty()) {
        hp.clear();
        break;
      }
      hp.insert(hp.end(), one.begin(), one.end());
    }
  }
  const size_t n = per_h * L.groups;
  L.w_h3 = hp.empty() ? c.slab.cur().reserve(n * sizeof(float)) : c.slab.upload(hp);
  L.h3_ok = c.slab.new_flag(!hp.empty());
  L.ovf_word = c.slab.cur().ovf_word(L.h3_ok);
  c.slab.cur().describe_flag(L.h3_ok, "conv " + std::to_string(L.cout) + " <- " + std::to_string(L.cin) + " x " + std::to_string(k) +
                                          (L.groups > 1 ? " groups " + std::to_string(L.groups) : std::string()) +
                                          (hp.empty() ? ", weights beyond fp16 range at load" : ""));
}

ConvW make_conv(Ctx& c, const float* w, const float* bias, int cout, int cin_g, int k, int groups, bool h3) {
  ConvW L;
  L.cin = cin_g * groups;
  L.cout = cout;
  L.k = k;
  L.groups = groups;
  L.cin_gp = conv_cin_pad(cin_g);
  L.cout_gp = conv_cout_pad(cout / groups);
  const std::vector<float> wp = pack_conv_weight(w, cout, cin_g, k, groups);
  L.w = c.slab.upload(wp);
  if (h3 && L.cin_gp % 16 == 0 && (groups == 1 || k > 1) && conv_h3_configured()) upload_h3(c, L, wp, k);
  L.bias = bias ? c.slab.upload(bias, (size_t)cout) : nullptr;
  return L;
}

ConvT1dW make_convT1d(Ctx& c, const float* w, const float* bias, int cin, int cout, int k, int s, int p) {
  PolyPhase1d pp = pack_convtranspose1d(w, bias, cin, cout, k, s, p);
  ConvT1dW L;
  L.w.cin = cin;
  L.w.cout = pp.cout_total;
  L.w.k = pp.taps;
  L.w.groups = 1;
  L.w.cin_gp = conv_cin_pad(cin);
  L.w.cout_gp = conv_cout_pad(pp.cout_total);
  L.w.w = c.slab.upload(pp.w);
  if (L.w.cin_gp % 16 == 0 && conv_h3_configured()) upload_h3(c, L.w, pp.w, pp.taps);   // polyphase taps: a dense stride-1 conv
  L.w.bias = c.slab.upload(pp.bias);
  L.stride = s;
  L.pad_t = p;
  L.cout_real = cout;
  L.k_orig = k;
  L.taps_pad = pp.pad;
  return L;
}

ConvT2dW make_convT2d(Ctx& c, const float* w, const float* scale, const float* shift, int cin, int cout) {
  // y[co][2i+a][2j+b] = sum_ci sum_(dy,dx) Wc[(a,b,co)][ci][(dy,dx)] x[ci][i+dy][j+dx]
  //   a=0: dy=0 <-> ky=1          a=1: dy=0 <-> ky=2, dy=1 <-> ky=0      (same for b/kx)
  // (oy = 2*iy - 1 + ky, RMVPE.py:263-271 stride 2, padding 1, output_padding 1)
  std::vector<float> wc((size_t)4 * cout * cin * 4, 0.f);
  auto kmap = [](int phase, int d) -> int {
    if (phase == 0) return d == 0 ? 1 : -1;
    return d == 0 ? 2 : 0;
  };
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b)
      for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
          for (int dy = 0; dy < 2; ++dy)
            for (int dx = 0; dx < 2; ++dx) {
              int ky = kmap(a, dy), kx = kmap(b, dx);
              if (ky < 0 || kx < 0) continue;
              float v = w[(((size_t)ci * cout + co) * 3 + ky) * 3 + kx];
              if (scale) v *= scale[co];
              wc[((((size_t)(a * 2 + b) * cout + co) * cin + ci) * 4) + dy * 2 + dx] = v;
            }
  std::vector<float> bias((size_t)4 * cout, 0.f);
  if (shift)
    for (int ph = 0; ph < 4; ++ph)
      for (int co = 0; co < cout; ++co) bias[(size_t)ph * cout + co] = shift[co];
  ConvT2dW L;
  L.w = make_conv(c, wc.data(), bias.data(), 4 * cout, cin, 4, 1);
  L.cout_real = cout;
  return L;
}

}  // namespace rvcx
