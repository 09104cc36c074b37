// The `mangio-crepe` F0 back-end of VC.get_f0 (rvc/infer/pipeline.py:86-117, 151-152):
//     x /= quantile(|x|, 0.999);  pitch = torchcrepe.predict(x, 16000, hop, f0_min, f0_max, "full",
//                                                             batch_size = 2 hop, pad = True)       # Viterbi decoder
//     pitch[pitch < 0.001] = nan;  f0 = nan_to_num(np.interp(arange(0, L p_len, L) / p_len, arange(L), pitch))
// torchcrepe (0.0.23, requirements.txt:9) is a third-party package that is not vendored with the reference and not
// installed here: what follows restates its published algorithm (core.predict / preprocess / postprocess, model.Crepe,
// decode.viterbi, convert.*) and librosa.sequence.viterbi; oracle/crepe.py is the CPU twin the tests compare with.
// PARITY UNPINNED (no torchcrepe, no weights offline).
//
//   frames      1024-sample windows every `hop` samples of the signal zero-padded by 512, zero mean / unit (unbiased) std
//   network     6 x {zero pad, conv (512 taps stride 4, then 64 taps), ReLU, eval BatchNorm, max-pool 2}, Linear, sigmoid
//               -- the convs run on conv_h3 / conv_fast (the stride-4 first layer re-indexed k = 4 a + r into a dense
//               stride-1 conv over 4 phase channels, as the RMVPE STFT is), batch item = frame
//   decoding    per batch of 2 hop frames: bins outside [fmin, fmax) masked, softmax over the sigmoid outputs, Viterbi with
//               the triangular transition band max(12 - |i - j|, 0) (float64, argmax = first maximum, like numpy),
//               cents = 20 bin + 1997.379 + dither, f = 10 * 2^(cents / 1200).  The +-20 cent triangular dither torchcrepe
//               draws from scipy's global RNG is an INPUT here (rvcx_utt_extra::crepe_dither), Philox-drawn when absent.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "common.h"
#include "ctx.h"
#include "layers.h"
#include "models.h"
#include "ops.h"

namespace rvcx {

namespace {

constexpr int CR_WIN = 1024, CR_BINS = 360, CR_PAD1 = 254, CR_PH = 4, CR_TP = 384, CR_T1 = 256;
constexpr float CR_CENTS_OFFSET = 1997.3794084376191f;
constexpr int CR_BAND = 12;

// one workgroup per frame: window -> (x - mean) / max(1e-10, std) -> the 4 phase rows of the padded window
__global__ __launch_bounds__(256) void crepe_frames_kernel(const float* x, long n, float inv_scale_div, int hop, int f0,
                                                           float* out) {
  __shared__ float red[8];
  const int f = blockIdx.x, tid = threadIdx.x;
  const long base = (long)(f0 + f) * hop - CR_WIN / 2;
  float v[4];
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const long i = base + tid + 256 * q;
    v[q] = (i >= 0 && i < n) ? x[i] / inv_scale_div : 0.f;     // x /= quantile (float32 division, like numpy's in-place op)
    s += v[q];
  }
  auto block_sum = [&](float t) {
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = t;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
  };
  const float mean = block_sum(s) / (float)CR_WIN;
  float d = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    v[q] -= mean;
    d += v[q] * v[q];
  }
  const float sd = sqrtf(block_sum(d) / (float)(CR_WIN - 1));
  const float den = fmaxf(1e-10f, sd);
  float* o = out + (long)f * CR_PH * CR_TP;
  // zero the pad columns: padded index j = s + 254, phase r = j & 3, column p = j >> 2 (0 .. 382), column 383 unused
  for (int e = tid; e < CR_PH * CR_TP; e += 256) {
    const int r = e / CR_TP, p = e - r * CR_TP;
    const int j = 4 * p + r;
    if (j < CR_PAD1 || j >= CR_PAD1 + CR_WIN) o[e] = 0.f;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int j = tid + 256 * q + CR_PAD1;
    o[(j & 3) * CR_TP + (j >> 2)] = v[q] / den;
  }
}

// eval BatchNorm (after the ReLU the conv epilogue applied) + max-pool 2 along time.
// feat == 0: (F, C, T) -> (F, C, T / 2);  feat != 0 (last layer): -> (T / 2 * C, F_total) at column f0 + f, row h * C + c
// (the order of x.permute(0, 2, 1, 3).reshape(-1, in_features) in model.Crepe.forward)
__global__ void crepe_bn_pool_kernel(const float* x, const float* sc, const float* sh, float* y, int C, int T, long total,
                                     int feat, long f_total, int f0) {
  const int T2 = T / 2;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int t = (int)(idx % T2);
    const long fc = idx / T2;
    const int c = (int)(fc % C);
    const long f = fc / C;
    const float a = x[fc * T + 2 * t] * sc[c] + sh[c], b = x[fc * T + 2 * t + 1] * sc[c] + sh[c];
    const float m = fmaxf(a, b);
    if (feat) y[((long)t * C + c) * f_total + f0 + f] = m;
    else y[idx] = m;
  }
}

// core.postprocess + decode.viterbi's softmax + librosa's log: one wave per frame.
// prob (360, F) sigmoid outputs -> logp (F, 360) = log(softmax(masked) + tiny) in float32
__global__ __launch_bounds__(256) void crepe_logprob_kernel(const float* prob, long F, int minidx, int maxidx, float* logp) {
  const long f = blockIdx.x * 4L + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (f >= F) return;
  float v[6];
  float mx = -INFINITY;
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const int j = lane + 64 * q;
    v[q] = (j < CR_BINS && j >= minidx && j < maxidx) ? prob[(long)j * F + f] : -INFINITY;
    mx = fmaxf(mx, v[q]);
  }
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    v[q] = expf(v[q] - mx);        // exp(-inf) = 0 for the masked bins
    s += v[q];
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const int j = lane + 64 * q;
    if (j < CR_BINS) logp[f * CR_BINS + j] = logf(v[q] / s + 1.1754943508222875e-38f);
  }
}

// librosa.sequence.viterbi for one batch of frames per workgroup (values float64, ptr = first argmax over the previous
// state, uniform initial distribution).  band[i][d + 11] = log(transition[i][i + d] + tiny) for |d| <= 11, log_far the
// same for the zero entries of the matrix.  ptr: (F, 360) uint16 scratch.
__global__ __launch_bounds__(384) void crepe_viterbi_kernel(const float* logp, long F, int batch, const double* band,
                                                            double log_far, double log_init, unsigned short* ptr,
                                                            int* bins) {
  __shared__ double val[2][CR_BINS];
  __shared__ double bnd[CR_BINS][2 * CR_BAND - 1];
  __shared__ int last;
  const long t0 = (long)blockIdx.x * batch;
  const int nt = (int)std::min<long>(batch, F - t0);
  const int j = threadIdx.x;
  for (int e = j; e < CR_BINS * (2 * CR_BAND - 1); e += 384) bnd[e / (2 * CR_BAND - 1)][e % (2 * CR_BAND - 1)] = band[e];
  if (j < CR_BINS) val[0][j] = (double)logp[t0 * CR_BINS + j] + log_init;
  __syncthreads();
  for (int t = 1; t < nt; ++t) {
    const double* vp = val[(t - 1) & 1];
    if (j < CR_BINS) {
      double best = -INFINITY;
      int arg = 0;
      for (int i = 0; i < CR_BINS; ++i) {
        const int d = j - i;
        const double c = vp[i] + ((d > -CR_BAND && d < CR_BAND) ? bnd[i][d + CR_BAND - 1] : log_far);
        if (c > best) {                                   // strict: the FIRST maximum, as np.argmax
          best = c;
          arg = i;
        }
      }
      ptr[(t0 + t) * CR_BINS + j] = (unsigned short)arg;
      val[t & 1][j] = (double)logp[(t0 + t) * CR_BINS + j] + best;
    }
    __syncthreads();
  }
  if (j == 0) {
    const double* vl = val[(nt - 1) & 1];
    int arg = 0;
    for (int i = 1; i < CR_BINS; ++i)
      if (vl[i] > vl[arg]) arg = i;
    bins[t0 + nt - 1] = arg;
    for (int t = nt - 2; t >= 0; --t) {
      arg = ptr[(t0 + t + 1) * CR_BINS + arg];
      bins[t0 + t] = arg;
    }
  }
  (void)last;
}

// convert.bins_to_frequency with the dither passed in (float32 steps as torch takes them), then get_f0_crepe's
// "pitch < 0.001 -> nan"
__global__ void crepe_pitch_kernel(const int* bins, const float* dither, long F, float* pitch) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < F; i += (long)gridDim.x * 256) {
    const float cents = ((float)(20 * bins[i]) + CR_CENTS_OFFSET) + dither[i];
    pitch[i] = 10.f * exp2f(cents / 1200.f);
  }
}

// +-20 cent triangular noise (scipy.stats.triang(c = 0.5, loc = -20, scale = 40)) from the Philox stream of the call
__global__ void crepe_dither_kernel(float* out, long n, uint64_t seed, uint64_t offset) {
  for (long q = blockIdx.x * 256L + threadIdx.x; q < n; q += (long)gridDim.x * 256) {
    uint64_t z = seed ^ (0x9E3779B97F4A7C15ull * (uint64_t)(offset + q + 1));   // splitmix64: two uniforms per frame
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const float u0 = ((float)(uint32_t)z + 0.5f) * 2.3283064365386963e-10f;
    const float u1 = ((float)(uint32_t)(z >> 32) + 0.5f) * 2.3283064365386963e-10f;
    out[q] = 20.f * (u0 + u1 - 1.f);
  }
}

// np.interp(arange(0, L p_len, L) / p_len, arange(L), pitch-with-nan) + nan_to_num (pipeline.py:108-116), numpy's
// arr_interp rules for NaN neighbours; float64 like numpy, stored as float32 (what the synthesizer is fed)
__global__ void crepe_resize_kernel(const float* pitch, long L, long p_len, float* f0) {
  for (long k = blockIdx.x * 256L + threadIdx.x; k < p_len; k += (long)gridDim.x * 256) {
    const double x = (double)(k * L) / (double)p_len;
    auto fp = [&](long i) {
      const float v = pitch[i];
      return v < 0.001f ? (double)NAN : (double)v;
    };
    const long j = (long)x;
    double r;
    if (j >= L - 1) {
      r = fp(L - 1);
    } else if ((double)j == x) {
      r = fp(j);
    } else {
      const double a = fp(j), b = fp(j + 1);
      const double slope = (b - a) / 1.0;
      r = slope * (x - (double)j) + a;
      if (isnan(r)) {
        r = slope * (x - (double)(j + 1)) + b;
        if (isnan(r) && a == b) r = a;
      }
    }
    f0[k] = isnan(r) ? 0.f : (float)r;
  }
}

}  // namespace

std::unique_ptr<CrepeModel> crepe_load(Ctx& c, const TensorTable& t) {
  auto M = std::make_unique<CrepeModel>();
  RegionScope scope(c, *M->region);
  int cin = 1;
  for (int i = 0; i < 6; ++i) {
    const std::string p = "conv" + std::to_string(i + 1);
    const auto shp = t.shape(p + ".weight");
    RVCX_CHECK(shp.size() == 4 && (int)shp[1] == cin && (int)shp[3] == 1 && (int)shp[2] == (i == 0 ? 512 : 64),
               "crepe: conv weight shape (Cout, Cin, K, 1) with K = 512 / 64");
    const int co = (int)shp[0];
    RVCX_CHECK(co % 16 == 0, "crepe: filter counts are multiples of 16");
    auto w = t.f32(p + ".weight");
    auto b = t.f32(p + ".bias");
    if (i == 0) {
      // stride 4, 512 taps, 1 input channel  ->  stride 1, 128 taps, 4 phase channels: w'[o][r][a] = w[o][4 a + r]
      std::vector<float> w2((size_t)co * CR_PH * 128);
      for (int o = 0; o < co; ++o)
        for (int r = 0; r < CR_PH; ++r)
          for (int a = 0; a < 128; ++a) w2[((size_t)o * CR_PH + r) * 128 + a] = w[(size_t)o * 512 + 4 * a + r];
      M->conv[i] = make_conv(c, w2.data(), b.data(), co, CR_PH, 128, 1);
    } else {
      M->conv[i] = make_conv(c, w.data(), b.data(), co, cin, 64, 1);
    }
    auto g = t.f32(p + "_BN.weight"), be = t.f32(p + "_BN.bias"), mu = t.f32(p + "_BN.running_mean"),
         var = t.f32(p + "_BN.running_var");
    RVCX_CHECK((int)g.size() == co && (int)be.size() == co && (int)mu.size() == co && (int)var.size() == co, "crepe: BN shape");
    std::vector<float> sc(co), sh(co);
    for (int k = 0; k < co; ++k) {
      sc[k] = g[k] / std::sqrt(var[k] + 0.0010000000474974513f);     // torchcrepe's eps
      sh[k] = be[k] - mu[k] * sc[k];
    }
    M->bn_scale[i] = c.slab.upload(sc);
    M->bn_shift[i] = c.slab.upload(sh);
    M->filters[i] = co;
    cin = co;
  }
  M->in_features = 4 * cin;
  {
    const auto shp = t.shape("classifier.weight");
    RVCX_CHECK(shp.size() == 2 && (int)shp[0] == CR_BINS && (int)shp[1] == M->in_features, "crepe: classifier shape");
    auto w = t.f32("classifier.weight");
    auto b = t.f32("classifier.bias");
    M->classifier = make_conv(c, w.data(), b.data(), CR_BINS, M->in_features, 1, 1);
  }
  {
    // decode.viterbi's transition matrix, row-normalised, + librosa's epsilon (tiny of the float32 probabilities), log
    std::vector<double> band((size_t)CR_BINS * (2 * CR_BAND - 1), 0.0);
    const double eps = 1.1754943508222875e-38;
    for (int i = 0; i < CR_BINS; ++i) {
      double sum = 0.0;
      for (int j = 0; j < CR_BINS; ++j) sum += (double)std::max(CR_BAND - std::abs(i - j), 0);
      for (int d = -(CR_BAND - 1); d < CR_BAND; ++d) {
        const int j = i + d;
        const double tr = (j >= 0 && j < CR_BINS) ? (double)(CR_BAND - std::abs(d)) / sum : 0.0;
        band[(size_t)i * (2 * CR_BAND - 1) + d + CR_BAND - 1] = std::log(tr + eps);
      }
    }
    M->band = reinterpret_cast<const double*>(c.slab.upload(reinterpret_cast<const float*>(band.data()), band.size() * 2));
    M->log_far = std::log(eps);
    M->log_init = std::log(1.0 / CR_BINS + eps);
  }
  M->region->seal();
  return M;
}

int crepe_frames(int64_t n, int hop) { return (int)(1 + n / hop); }

int crepe_frequency_to_bin(float f, bool ceil) {
  // convert.frequency_to_bins on a float32 scalar: cents = 1200 log2(f / 10), bins = (cents - offset) / 20
  const float cents = 1200.f * std::log2(f / 10.f);
  const float bins = (cents - CR_CENTS_OFFSET) / 20.f;
  return (int)(ceil ? std::ceil(bins) : std::floor(bins));
}

// np.quantile(|x|, 0.999) of a float32 array the way numpy 1.23 computes it (method "linear": the two order statistics
// are float32, the interpolation weight float64, _lerp's two branches); the caller divides by (float) of it
double crepe_quantile999(std::vector<float>& a) {
  RVCX_CHECK(!a.empty(), "crepe: empty signal");
  for (auto& v : a) v = std::fabs(v);
  const double vi = 0.999 * (double)(a.size() - 1);
  const size_t lo = (size_t)std::floor(vi), hi = std::min(lo + 1, a.size() - 1);
  std::nth_element(a.begin(), a.begin() + lo, a.end());
  const float va = a[lo];
  const float vb = hi == lo ? va : *std::min_element(a.begin() + lo + 1, a.end());
  const double t = vi - (double)lo;
  const float diff = vb - va;
  double r = (double)va + (double)diff * t;
  if (t >= 0.5) r = (double)vb - (double)diff * (1.0 - t);
  return r;
}

size_t crepe_arena_bytes(const CrepeModel& m, int64_t n, int hop) {
  // exactly what crepe_forward allocates: per batch of 2 * hop frames the framed input and EVERY layer's conv and pooled
  // buffers (they stay live until the batch's reset), each rounded up to the arena's 256-byte granule
  auto al = [](size_t floats) { return (floats * sizeof(float) + 255) & ~size_t(255); };
  const size_t F = (size_t)crepe_frames(n, hop), fb = (size_t)std::min<size_t>(2 * (size_t)hop, F);
  size_t batch = al(fb * CR_PH * CR_TP);
  int T = CR_T1;
  for (int i = 0; i < 6; ++i) {
    batch += al(fb * (size_t)m.filters[i] * T);
    if (i < 5) batch += al(fb * (size_t)m.filters[i] * (T / 2));
    T /= 2;
  }
  return batch + al((size_t)m.in_features * F) + 2 * al((size_t)CR_BINS * F) + 3 * al(F) + al((size_t)n) +
         ((F * CR_BINS * sizeof(unsigned short) + 255) & ~size_t(255)) + ((size_t)16 << 20);
}

// x: device, n samples of the (already normalised-by-`scale`) signal; writes `pitch` (F frames, Hz) and optionally the
// sigmoid outputs (360, F) and the Viterbi bins
void crepe_forward(Ctx& c, const CrepeModel& m, const float* x, int64_t n, float scale, int hop, float fmin, float fmax,
                   const float* dither, float* pitch, float* probs_out, int* bins_out, hipStream_t s) {
  Arena& A = c.arena;
  RVCX_CHECK(hop >= 1 && hop <= 4096, "crepe: hop_length out of range");
  const long F = crepe_frames(n, hop);
  const int batch = 2 * hop;                               // batch_size = hop_length * 2 (pipeline.py:103)
  float* feat = A.alloc<float>((size_t)m.in_features * F);
  float* probs = A.alloc<float>((size_t)CR_BINS * F);
  for (long f0 = 0; f0 < F; f0 += batch) {
    const int fb = (int)std::min<long>(batch, F - f0);
    const size_t mark = A.mark();
    float* xp = A.alloc<float>((size_t)fb * CR_PH * CR_TP);
    hipLaunchKernelGGL(crepe_frames_kernel, dim3(fb), dim3(256), 0, s, x, (long)n, scale, hop, (int)f0, xp);
    int T = CR_T1, cin_t = CR_TP;
    const float* cur = xp;
    for (int i = 0; i < 6; ++i) {
      const int co = m.filters[i];
      float* y = A.alloc<float>((size_t)fb * co * T);
      ConvArgs a = conv1d_args(m.conv[i], cur, y, fb, cin_t, T, 1, 1, i == 0 ? 0 : 31);
      a.act = ACT_RELU;
      c.conv_on(a, s);
      const long tot = (long)fb * co * (T / 2);
      const bool last = i == 5;
      float* pl = last ? feat : A.alloc<float>((size_t)tot);
      hipLaunchKernelGGL(crepe_bn_pool_kernel, dim3((unsigned)std::min<long>(cdiv64(tot, 256), 65535)), dim3(256), 0, s, y,
                         m.bn_scale[i], m.bn_shift[i], pl, co, T, tot, last ? 1 : 0, F, (int)f0);
      cur = pl;
      T /= 2;
      cin_t = T;
    }
    A.reset(mark);
  }
  {
    ConvArgs a = conv1d_args(m.classifier, feat, probs, 1, (int)F, (int)F);
    a.act = ACT_SIGMOID;
    c.conv_on(a, s);
  }
  if (probs_out) RVCX_HIP(hipMemcpyAsync(probs_out, probs, (size_t)CR_BINS * F * sizeof(float), hipMemcpyDeviceToDevice, s));
  crepe_decode(c, m, probs, F, batch, fmin, fmax, dither, pitch, bins_out, s);
}

void crepe_decode(Ctx& c, const CrepeModel& m, const float* probs, long F, int batch, float fmin, float fmax,
                  const float* dither, float* pitch, int* bins_out, hipStream_t s) {
  Arena& A = c.arena;
  float* logp = A.alloc<float>((size_t)F * CR_BINS);
  unsigned short* ptr = A.alloc<unsigned short>((size_t)F * CR_BINS);
  int* bins = A.alloc<int>((size_t)F);
  const int minidx = std::max(0, crepe_frequency_to_bin(fmin, false));
  const int maxidx = std::min(CR_BINS, crepe_frequency_to_bin(fmax, true));
  RVCX_CHECK(minidx < maxidx, "crepe: empty pitch range");
  hipLaunchKernelGGL(crepe_logprob_kernel, dim3((unsigned)cdiv64(F, 4)), dim3(256), 0, s, probs, F, minidx, maxidx, logp);
  hipLaunchKernelGGL(crepe_viterbi_kernel, dim3((unsigned)cdiv64(F, batch)), dim3(384), 0, s, logp, F, batch, m.band,
                     m.log_far, m.log_init, ptr, bins);
  hipLaunchKernelGGL(crepe_pitch_kernel, dim3((unsigned)std::min<long>(cdiv64(F, 256), 1024)), dim3(256), 0, s, bins, dither,
                     F, pitch);
  if (bins_out) RVCX_HIP(hipMemcpyAsync(bins_out, bins, (size_t)F * sizeof(int), hipMemcpyDeviceToDevice, s));
  RVCX_HIP(hipGetLastError());
}

void launch_crepe_dither(float* out, long n, uint64_t seed, uint64_t offset, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(crepe_dither_kernel, dim3((unsigned)std::min<long>(cdiv64(n, 256), 1024)), dim3(256), 0, s, out, n, seed,
                     offset);
}

void launch_crepe_resize(const float* pitch, long L, long p_len, float* f0, hipStream_t s) {
  hipLaunchKernelGGL(crepe_resize_kernel, dim3((unsigned)std::min<long>(cdiv64(p_len, 256), 1024)), dim3(256), 0, s, pitch, L,
                     p_len, f0);
}

}  // namespace rvcx
