// FCPE F0 estimator (rvc/lib/predictors/FCPE.py): conv-STFT -> slaney mel -> conv stack (GroupNorm(4)) ->
// n_layers x [LayerNorm -> Performer (FAVOR+) self-attention -> + ; Conformer conv module -> +] -> LayerNorm ->
// weight-normed Linear -> sigmoid -> local-argmax cents -> Hz, then FCPEF0Predictor's post-processing.
//
// Layout: every activation is channel-first (B, C, T) so that each Linear / 1x1 / k=3 conv is a launch of the MFMA
// implicit-GEMM family (conv.h).  The Performer attention never materialises the (T x m) random-feature maps in
// HBM: one kernel builds k' for 64 frames in LDS and reduces it into a partial (m x 64) context, a second sums
// the partials in a fixed order, a third builds q' in LDS and applies context and normaliser -- linear in T,
// deterministic, and independent of the batch size (one item's frames never share a workgroup with another's).
#include <climits>
#include <cmath>
#include <mutex>

#include "models.h"
#include "ops.h"

namespace rvcx {

namespace {

constexpr int N_FFT = 1024, HOP = 160, N_MELS = 128, PADL = (N_FFT - HOP) / 2;   // 432, FCPE.py:124
constexpr double SR = 16000.0;
constexpr int PF_T = 64;      // frames per workgroup of the Performer kernels
constexpr int PF_D = 64;      // dim_head (SelfAttention default, FCPE.py:446)
constexpr int PF_MAXM = 320;  // random features held in LDS (int(64 ln 64) = 266)

// librosa.filters.mel(sr=16000, n_fft=1024, n_mels=128, fmin, fmax) with librosa's defaults htk=False,
// norm="slaney" (FCPE.py:115-117): linear below 1 kHz, logarithmic above
std::vector<float> mel_filterbank_slaney(double fmin, double fmax) {
  const int nb = N_FFT / 2 + 1;
  const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
  auto hz2mel = [&](double f) { return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp; };
  auto mel2hz = [&](double m) { return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m; };
  std::vector<double> melf(N_MELS + 2);
  const double m0 = hz2mel(fmin), m1 = hz2mel(fmax);
  for (int i = 0; i < N_MELS + 2; ++i) melf[i] = mel2hz(m0 + (m1 - m0) * (double)i / (double)(N_MELS + 1));
  std::vector<float> w((size_t)N_MELS * nb, 0.f);
  for (int i = 0; i < N_MELS; ++i) {
    const double enorm = 2.0 / (melf[i + 2] - melf[i]);
    for (int j = 0; j < nb; ++j) {
      const double fj = (SR / 2.0) * (double)j / (double)(nb - 1);
      const double lower = (fj - melf[i]) / (melf[i + 1] - melf[i]);
      const double upper = (melf[i + 2] - fj) / (melf[i + 2] - melf[i + 1]);
      w[(size_t)i * nb + j] = (float)(std::max(0.0, std::min(lower, upper)) * enorm);
    }
  }
  return w;
}

// ------------------------------------------------------------------ mel post: log(clamp(., 1e-5)), last frame repeated
// Wav2Mel.extract_mel (FCPE.py:768-785): the STFT yields n//160 frames, the model is fed n//160 + 1
__global__ void fcpe_mel_post_kernel(const float* __restrict__ mel, float* __restrict__ out, int F0, int F, long total) {
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int t = idx % F;
    const long bc = idx / F;
    out[idx] = logf(fmaxf(mel[bc * F0 + min(t, F0 - 1)], 1e-5f));
  }
}

// ------------------------------------------------------------------ GroupNorm(G, C) + LeakyReLU   (FCPE.py:605-607)
// stats: one block per (channel, item) -> fp64 {sum, sum of squares}; apply: every block re-adds its group's
// channel sums in channel order (deterministic), then normalises its own row.
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, double* __restrict__ st, int C, int T) {
  __shared__ double red[2][4];
  const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const float* xr = x + ((long)b * C + c) * T;
  double s = 0.0, q = 0.0;
  for (int t = tid; t < T; t += 256) {
    const double v = xr[t];
    s += v;
    q += v * v;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    s += __shfl_xor(s, d, 64);
    q += __shfl_xor(q, d, 64);
  }
  if ((tid & 63) == 0) {
    red[0][tid >> 6] = s;
    red[1][tid >> 6] = q;
  }
  __syncthreads();
  if (tid == 0) {
    st[((long)b * C + c) * 2] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    st[((long)b * C + c) * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

__global__ __launch_bounds__(256) void gn_apply_lrelu_kernel(const float* __restrict__ x, const double* __restrict__ st,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ y,
                                                             int C, int T, int G, float eps, float slope) {
  __shared__ float mr[2];
  const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int cg = C / G, g0 = (c / cg) * cg;
  if (tid == 0) {
    double s = 0.0, q = 0.0;
    for (int i = 0; i < cg; ++i) {
      s += st[((long)b * C + g0 + i) * 2];
      q += st[((long)b * C + g0 + i) * 2 + 1];
    }
    const double n = (double)cg * T, mean = s / n, var = fmax(q / n - mean * mean, 0.0);
    mr[0] = (float)mean;
    mr[1] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  const float mean = mr[0], rstd = mr[1], ga = gamma[c], be = beta[c];
  const float* xr = x + ((long)b * C + c) * T;
  float* yr = y + ((long)b * C + c) * T;
  for (int t = tid; t < T; t += 256) {
    const float v = (xr[t] - mean) * rstd * ga + be;
    yr[t] = v > 0.f ? v : v * slope;
  }
}

// ------------------------------------------------------------------ Performer attention (FAVOR+, softmax kernel)
// softmax_kernel (FCPE.py:170-197) for keys, linear_attention's k_cumsum and context (FCPE.py:344-350), 64 frames
// per workgroup: part[(b,h)][tile][j][0..63] = sum_t k'[t][j] v[t][e], part[..][M*64 + j] = sum_t k'[t][j].
__global__ __launch_bounds__(256) void performer_kv_kernel(const float* __restrict__ k, const float* __restrict__ v,
                                                           long bs, int T, const float* __restrict__ P, int M,
                                                           float* __restrict__ part, int ntile, float dn, float ratio,
                                                           float eps) {
  extern __shared__ float lds[];
  float* tile = lds;                      // [64][65] staging of the k, then the v tile; dead once both sit in registers
  float* kp = lds;                        // [M][64], overlays the staging tile (2 workgroups per CU fit in 160 KB)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tl = blockIdx.x, h = blockIdx.y, b = blockIdx.z, H = gridDim.y;
  const int t0 = tl * PF_T;
  const float* kb = k + (long)b * bs + (long)h * PF_D * T;
  const float* vb = v + (long)b * bs + (long)h * PF_D * T;
  for (int i = tid; i < PF_D * PF_T; i += 256) {
    const int d = i >> 6, t = i & 63;
    tile[d * (PF_T + 1) + t] = (t0 + t < T) ? kb[(long)d * T + t0 + t] : 0.f;
  }
  __syncthreads();
  float xr[PF_D], vr[PF_T];
  float diag = 0.f;
#pragma unroll
  for (int d = 0; d < PF_D; ++d) {
    const float x = tile[d * (PF_T + 1) + lane];
    diag += x * x;
    xr[d] = dn * x;
  }
  diag = diag / 2.0f * (dn * dn);
  __syncthreads();
  for (int i = tid; i < PF_D * PF_T; i += 256) {
    const int d = i >> 6, t = i & 63;
    tile[d * (PF_T + 1) + t] = (t0 + t < T) ? vb[(long)d * T + t0 + t] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < PF_T; ++t) vr[t] = tile[lane * (PF_T + 1) + t];   // v[e = lane][t]
  __syncthreads();
  const bool valid = t0 + lane < T;
  for (int j = wave; j < M; j += 4) {
    const float* pj = P + (long)j * PF_D;
    float dot = 0.f;
#pragma unroll
    for (int d = 0; d < PF_D; ++d) dot += xr[d] * pj[d];
    kp[j * PF_T + lane] = valid ? ratio * expf(dot - diag + eps) : 0.f;
  }
  __syncthreads();
  float* out = part + (((long)b * H + h) * ntile + tl) * ((long)M * (PF_D + 1));
  for (int j = wave; j < M; j += 4) {
    const float* kj = kp + j * PF_T;
    float acc = 0.f, ks = 0.f;
#pragma unroll
    for (int t = 0; t < PF_T; ++t) {
      acc += kj[t] * vr[t];
      ks += kj[t];
    }
    out[(long)j * PF_D + lane] = acc;
    if (lane == 0) out[(long)M * PF_D + j] = ks;
  }
}

// ctx[(b,h)][i] = sum over tiles (ascending) of part[(b,h)][tile][i],  i < M*65
__global__ void performer_reduce_kernel(const float* __restrict__ part, float* __restrict__ ctx, int ntile, int len) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= len) return;
  const float* p = part + (long)blockIdx.y * ntile * len + i;
  float s = 0.f;
  for (int t = 0; t < ntile; ++t) s += p[(long)t * len];
  ctx[(long)blockIdx.y * len + i] = s;
}

// softmax_kernel for queries + out = (q' ctx) / (q' . k_cumsum + 1e-8)    (FCPE.py:186-193, 345-351)
__global__ __launch_bounds__(256) void performer_q_kernel(const float* __restrict__ q, long bs, int T,
                                                          const float* __restrict__ P, int M,
                                                          const float* __restrict__ ctx, float* __restrict__ out,
                                                          long out_bs, float dn, float ratio, float eps) {
  extern __shared__ float lds[];
  float* tile = lds;                      // [64][65] staging of the q tile, dead once it sits in registers
  float* qp = lds;                        // [M][64], overlays the staging tile
  float* red = qp + (long)max(M, PF_D + 2) * PF_T;   // [4][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tl = blockIdx.x, h = blockIdx.y, b = blockIdx.z, H = gridDim.y;
  const int t0 = tl * PF_T;
  const float* qb = q + (long)b * bs + (long)h * PF_D * T;
  for (int i = tid; i < PF_D * PF_T; i += 256) {
    const int d = i >> 6, t = i & 63;
    tile[d * (PF_T + 1) + t] = (t0 + t < T) ? qb[(long)d * T + t0 + t] : 0.f;
  }
  __syncthreads();
  float xr[PF_D];
  float diag = 0.f;
#pragma unroll
  for (int d = 0; d < PF_D; ++d) {
    const float x = tile[d * (PF_T + 1) + lane];
    diag += x * x;
    xr[d] = dn * x;
  }
  diag = diag / 2.0f * (dn * dn);
  __syncthreads();
  float mx = -INFINITY;
  for (int j = wave; j < M; j += 4) {
    const float* pj = P + (long)j * PF_D;
    float dot = 0.f;
#pragma unroll
    for (int d = 0; d < PF_D; ++d) dot += xr[d] * pj[d];
    qp[j * PF_T + lane] = dot;
    mx = fmaxf(mx, dot);
  }
  red[wave * PF_T + lane] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[lane], red[PF_T + lane]), fmaxf(red[2 * PF_T + lane], red[3 * PF_T + lane]));
  for (int j = wave; j < M; j += 4) qp[j * PF_T + lane] = ratio * (expf(qp[j * PF_T + lane] - diag - mx) + eps);
  __syncthreads();
  // thread (t = lane, e = wave*16 .. +15)
  const float* cb = ctx + ((long)b * H + h) * ((long)M * (PF_D + 1));
  const float* ksum = cb + (long)M * PF_D;
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float dsum = 0.f;
  for (int j = 0; j < M; ++j) {
    const float qv = qp[j * PF_T + lane];
    const float* cj = cb + (long)j * PF_D + wave * 16;
    dsum += qv * ksum[j];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] += qv * cj[i];
  }
  const float dinv = 1.0f / (dsum + 1e-8f);
  if (t0 + lane < T) {
    float* ob = out + (long)b * out_bs + ((long)h * PF_D + wave * 16) * T + t0 + lane;
#pragma unroll
    for (int i = 0; i < 16; ++i) ob[(long)i * T] = acc[i] * dinv;
  }
}

// ------------------------------------------------------------------ GLU -> depth-wise conv -> Swish  (FCPE.py:329-333)
// x (B, 2*Ci, T): a = x[c], gate = x[Ci + c];  y[c][t] = swish(bias[c] + sum_k w[c][k] * (a * sigmoid(gate))[t + k - padl])
constexpr int DW_TT = 256;
__global__ __launch_bounds__(DW_TT) void glu_dwconv_swish_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ bias, float* __restrict__ y,
                                                                int Ci, int T, int K, int padl) {
  extern __shared__ float lds[];
  float* g = lds;            // [DW_TT + K - 1]
  float* wk = g + DW_TT + K; // [K]
  const int c = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const int t0 = blockIdx.x * DW_TT;
  const float* xa = x + ((long)b * 2 * Ci + c) * T;
  const float* xg = xa + (long)Ci * T;
  for (int i = tid; i < DW_TT + K - 1; i += DW_TT) {
    const int t = t0 - padl + i;
    float v = 0.f;
    if (t >= 0 && t < T) v = xa[t] * (1.0f / (1.0f + expf(-xg[t])));
    g[i] = v;
  }
  if (tid < K) wk[tid] = w[(long)c * K + tid];
  __syncthreads();
  const int t = t0 + tid;
  if (t >= T) return;
  float acc = 0.f;
  for (int kk = 0; kk < K; ++kk) acc += wk[kk] * g[tid + kk];
  acc += bias[c];
  y[((long)b * Ci + c) * T + t] = acc * (1.0f / (1.0f + expf(-acc)));
}

// ------------------------------------------------------------------ cents_local_decoder + cent_to_f0 (FCPE.py:673-693)
// one wave per frame: arg-max over the bins (first maximum, as torch.max), 9 bins around it (indices clamped, so the
// edge bins repeat exactly as torch.gather sees them), weighted cents -> Hz; unconfident frames -> 0 (x * -inf -> 2^-inf)
__global__ __launch_bounds__(256) void fcpe_decode_kernel(const float* __restrict__ sal, const float* __restrict__ cent,
                                                          float* __restrict__ f0, long frames, int nbins, float threshold) {
  const long fr = blockIdx.x * 4L + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (fr >= frames) return;
  const float* s = sal + fr * nbins;
  float mx = -INFINITY;
  int arg = 0x7fffffff;
  for (int i = lane; i < nbins; i += 64) {
    const float v = s[i];
    if (v > mx) {
      mx = v;
      arg = i;
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const float om = __shfl_xor(mx, d, 64);
    const int oa = __shfl_xor(arg, d, 64);
    if (om > mx || (om == mx && oa < arg)) {
      mx = om;
      arg = oa;
    }
  }
  if (lane == 0) {
    float num = 0.f, den = 0.f;
    for (int k = 0; k < 9; ++k) {
      const int idx = min(max(arg - 4 + k, 0), nbins - 1);
      num += cent[idx] * s[idx];
      den += s[idx];
    }
    const float c = num / den;
    f0[fr] = (mx <= threshold) ? 0.f : 10.0f * exp2f(c / 1200.0f);
  }
}

// ------------------------------------------------------------------ FCPEF0Predictor.post_process + VC.get_f0 tail
// (FCPE.py:841-867, pipeline.py:183-201).  One workgroup per item.
//  1. F.interpolate(mode="nearest") from F_in to p_len frames: src = min(int(floorf(i * (float)F_in / p_len)), F_in - 1)
//  2. the unvoiced (== 0) frames are bridged by np.interp between their voiced neighbours in float64, on the time
//     axes the reference builds: x_i = (i * 512) / 16000 and xp_j = (512 / 16000) * j (the two roundings differ, so
//     the interval search is restated literally: the largest voiced j with xp_j <= x_i)
//  3. f0 *= 2^(pitch/12); coarse = rint of the mel-scaled value clipped to 1..255; f0 leaves as float32
__device__ inline int block_scan_max(int v, int* sh, int tid) {   // inclusive max-scan over 1024 threads
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(v, d, 64);
    if (lane >= d) v = max(v, o);
  }
  if (lane == 63) sh[wave] = v;
  __syncthreads();
  int carry = INT_MIN;
  for (int w = 0; w < wave; ++w) carry = max(carry, sh[w]);
  __syncthreads();
  return max(v, carry);
}

__global__ __launch_bounds__(1024) void fcpe_post_kernel(const float* __restrict__ f0raw, long raw_stride, int F_in,
                                                         int p_len, float* __restrict__ tmp, int* __restrict__ prevv,
                                                         int* __restrict__ nextv, long tmp_stride,
                                                         float* __restrict__ f0_out, int* __restrict__ coarse,
                                                         long out_stride, double shift, double mel_min, double mel_max) {
  __shared__ int sh[16];
  __shared__ int carry_s;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* src = f0raw + (long)b * raw_stride;
  float* r = tmp + (long)b * tmp_stride;
  int* pv = prevv + (long)b * tmp_stride;
  int* nx = nextv + (long)b * tmp_stride;
  const float scale = (float)F_in / (float)p_len;
  for (int i = tid; i < p_len; i += 1024) {
    const int si = (F_in == p_len) ? i : min((int)floorf((float)i * scale), F_in - 1);
    r[i] = src[si];
  }
  __syncthreads();
  // prev[i]: largest voiced index <= i (or -1)
  if (tid == 0) carry_s = -1;
  __syncthreads();
  for (int base = 0; base < p_len; base += 1024) {
    const int i = base + tid;
    int v = (i < p_len && r[i] != 0.f) ? i : -1;
    v = block_scan_max(v, sh, tid);
    v = max(v, carry_s);
    if (i < p_len) pv[i] = v;
    __syncthreads();
    if (tid == 1023) carry_s = v;
    __syncthreads();
  }
  // next[i]: smallest voiced index >= i (or p_len): the same scan on the reversed axis with negated indices
  if (tid == 0) carry_s = -p_len;
  __syncthreads();
  for (int base = 0; base < p_len; base += 1024) {
    const int i = p_len - 1 - (base + tid);
    int v = (i >= 0 && r[i] != 0.f) ? -i : -p_len;
    v = block_scan_max(v, sh, tid);
    v = max(v, carry_s);
    if (i >= 0) nx[i] = -v;
    __syncthreads();
    if (tid == 1023) carry_s = v;
    __syncthreads();
  }
  __syncthreads();
  const int first = nx[0];                        // p_len when no frame is voiced
  const double step = 512.0 / 16000.0;            // hop_length / sample_rate as the reference's Python float
  for (int i = tid; i < p_len; i += 1024) {
    double f = 0.0;
    if (first < p_len) {
      const double x = (double)((long)i * 512) / 16000.0;
      int j = pv[i];
      if (j == i && !(__dmul_rn(step, (double)i) <= x)) j = (i > 0) ? pv[i - 1] : -1;
      if (j < 0) {
        f = (double)r[first];                                          // left
      } else {
        const int jn = (j + 1 < p_len) ? nx[j + 1] : p_len;
        const double xj = __dmul_rn(step, (double)j), fj = (double)r[j];
        if (jn >= p_len || xj == x) {
          f = fj;                                                      // last point / right / exact hit
        } else {
          const double slope = __ddiv_rn(__dsub_rn((double)r[jn], fj), __dsub_rn(__dmul_rn(step, (double)jn), xj));
          f = __dadd_rn(__dmul_rn(slope, __dsub_rn(x, xj)), fj);
        }
      }
    }
    f *= shift;
    double m = 1127.0 * log(1.0 + f / 700.0);
    if (m > 0.0) m = (m - mel_min) * 254.0 / (mel_max - mel_min) + 1.0;
    if (m <= 1.0) m = 1.0;
    if (m > 255.0) m = 255.0;
    coarse[(long)b * out_stride + i] = (int)rint(m);
    f0_out[(long)b * out_stride + i] = (float)f;
  }
}

ConvW lin(Ctx& c, const TensorTable& t, const std::string& p) {
  auto w = t.f32(p + ".weight");
  auto b = t.f32(p + ".bias");
  const auto shp = t.shape(p + ".weight");
  const int k = shp.size() > 2 ? (int)shp[2] : 1;
  return make_conv(c, w.data(), b.data(), (int)shp[0], (int)shp[1], k, 1);
}

}  // namespace

std::unique_ptr<FcpeModel> fcpe_load(Ctx& c, const rvcx_fcpe_cfg& cfg, const TensorTable& t) {
  auto M = std::make_unique<FcpeModel>();
  RegionScope scope(c, *M->region);
  M->cfg = cfg;
  RVCX_CHECK(cfg.dim_head == PF_D, "fcpe: dim_head must be 64");
  RVCX_CHECK(cfg.nb_features > 0 && cfg.nb_features <= PF_MAXM, "fcpe: too many random features");
  RVCX_CHECK(cfg.input_channel == N_MELS, "fcpe: the mel front end has 128 bands");
  RVCX_CHECK(cfg.n_chans % 4 == 0 && cfg.n_chans <= 1024, "fcpe: n_chans");
  const int C = cfg.n_chans, inner = cfg.heads * cfg.dim_head;
  M->stft = make_stft_conv(c);
  {
    auto m = mel_filterbank_slaney(cfg.mel_fmin, cfg.mel_fmax);
    M->melfb = make_conv(c, m.data(), nullptr, N_MELS, N_FFT / 2 + 1, 1, 1);
  }
  M->stack0 = lin(c, t, "stack.0");
  M->gn_g = c.slab.upload(t.f32("stack.1.weight"));
  M->gn_b = c.slab.upload(t.f32("stack.1.bias"));
  M->stack3 = lin(c, t, "stack.3");
  for (int l = 0; l < cfg.n_layers; ++l) {
    const std::string p = "decoder._layers." + std::to_string(l);
    FcpeModel::Layer L;
    L.ln_g = c.slab.upload(t.f32(p + ".norm.weight"));
    L.ln_b = c.slab.upload(t.f32(p + ".norm.bias"));
    {
      std::vector<float> w, b;
      for (const char* nm : {".attn.to_q", ".attn.to_k", ".attn.to_v"}) {
        auto wi = t.f32(p + nm + ".weight");
        auto bi = t.f32(p + nm + ".bias");
        RVCX_CHECK((int)wi.size() == inner * C, "fcpe: to_q/k/v shape");
        w.insert(w.end(), wi.begin(), wi.end());
        b.insert(b.end(), bi.begin(), bi.end());
      }
      L.qkv = make_conv(c, w.data(), b.data(), 3 * inner, C, 1, 1);
    }
    L.out = lin(c, t, p + ".attn.to_out");
    {
      const auto shp = t.shape(p + ".attn.fast_attention.projection_matrix");
      RVCX_CHECK((int)shp[0] == cfg.nb_features && (int)shp[1] == cfg.dim_head, "fcpe: projection_matrix shape");
      L.proj = c.slab.upload(t.f32(p + ".attn.fast_attention.projection_matrix"));
    }
    L.cln_g = c.slab.upload(t.f32(p + ".conformer.net.0.weight"));
    L.cln_b = c.slab.upload(t.f32(p + ".conformer.net.0.bias"));
    L.pw1 = lin(c, t, p + ".conformer.net.2");
    {
      const auto shp = t.shape(p + ".conformer.net.4.conv.weight");   // (inner_dim, 1, K)
      RVCX_CHECK((int)shp[0] * 2 == L.pw1.cout && (int)shp[1] == 1 && (int)shp[2] == cfg.dw_kernel, "fcpe: depth-wise conv shape");
      L.dw_w = c.slab.upload(t.f32(p + ".conformer.net.4.conv.weight"));
      L.dw_b = c.slab.upload(t.f32(p + ".conformer.net.4.conv.bias"));
    }
    L.pw2 = lin(c, t, p + ".conformer.net.6");
    M->layers.push_back(L);
  }
  M->norm_g = c.slab.upload(t.f32("norm.weight"));
  M->norm_b = c.slab.upload(t.f32("norm.bias"));
  {
    // dense_out = weight_norm(nn.Linear): w = g * v / ||v|| per output row, either container layout (FCPE.py:627)
    const bool par = t.has("dense_out.parametrizations.weight.original0");
    auto g = t.f32(par ? "dense_out.parametrizations.weight.original0" : "dense_out.weight_g");
    auto v = t.f32(par ? "dense_out.parametrizations.weight.original1" : "dense_out.weight_v");
    auto b = t.f32("dense_out.bias");
    const int nout = (int)g.size(), cin = (int)(v.size() / g.size());
    RVCX_CHECK(nout == cfg.out_dims && cin == C, "fcpe: dense_out shape");
    for (int o = 0; o < nout; ++o) {
      double nn = 0.0;
      for (int i = 0; i < cin; ++i) nn += (double)v[(size_t)o * cin + i] * v[(size_t)o * cin + i];
      const float sc = g[o] / (float)std::sqrt(nn);
      for (int i = 0; i < cin; ++i) v[(size_t)o * cin + i] *= sc;
    }
    M->dense = make_conv(c, v.data(), b.data(), nout, cin, 1, 1);
  }
  {
    // cent_table is a persistent buffer: the checkpoint's copy wins (FCPE.py:594-602)
    std::vector<float> ct;
    if (t.has("cent_table")) {
      ct = t.f32("cent_table");
    } else {
      const double lo = (double)(1200.0f * std::log2(32.70f / 10.0f)), hi = (double)(1200.0f * std::log2(1975.5f / 10.0f));
      ct.resize(cfg.out_dims);
      for (int i = 0; i < cfg.out_dims; ++i) ct[i] = (float)(lo + (hi - lo) * (double)i / (double)(cfg.out_dims - 1));
    }
    RVCX_CHECK((int)ct.size() == cfg.out_dims, "fcpe: cent_table size");
    M->cent_table = c.slab.upload(ct);
  }
  M->region->seal();
  return M;
}

static int fcpe_tiles(int F) { return cdiv(F, PF_T); }

size_t fcpe_arena_bytes(const FcpeModel& m, int B, int64_t n) {
  const size_t F = (size_t)(n / HOP + 1), C = m.cfg.n_chans, inner = (size_t)m.cfg.heads * m.cfg.dim_head;
  size_t tot = 2 * (size_t)(n + 2 * N_FFT) + (size_t)(N_FFT + 2 + N_FFT / 2 + 1 + 2 * N_MELS) * F;   // mel front end
  tot += (size_t)(4 * C + 3 * inner + inner + 4 * C + 2 * C + m.cfg.out_dims + 8) * F;                // encoder buffers
  tot += ((size_t)fcpe_tiles((int)F) + 1) * m.cfg.heads * m.cfg.nb_features * (PF_D + 1);             // context partials
  return (size_t)B * tot * sizeof(float) + (size_t)B * C * 16 + ((size_t)64 << 20);
}

void fcpe_forward(Ctx& c, const FcpeModel& m, int B, const float* audio, int64_t n, float threshold, float* f0,
                  float* sal_out, float* mel_out, hipStream_t s, const std::function<void()>* after_stack) {
  Arena& A = c.arena;
  const auto& cf = m.cfg;
  RVCX_CHECK(n > N_FFT, "fcpe: clip shorter than one analysis window");   // the reference would zero-pad instead (FCPE.py:125-129)
  const int F0 = (int)(n / HOP), F = F0 + 1, nb = N_FFT / 2 + 1;
  const int C = cf.n_chans, H = cf.heads, inner = H * cf.dim_head, Mf = cf.nb_features;
  // ---- mel (STFT.get_mel + Wav2Mel.extract_mel, FCPE.py:96-160, 768-785)
  const int Mh = cdiv((int)n + 2 * PADL, HOP) + 1;
  float* apad = A.alloc<float>((size_t)B * Mh * HOP);
  RVCX_HIP(hipMemsetAsync(apad, 0, (size_t)B * Mh * HOP * sizeof(float), s));
  launch_reflect_pad(audio, apad, B, (int)n, PADL, (long)Mh * HOP, s);
  float* x2 = A.alloc<float>((size_t)B * Mh * HOP);
  launch_transpose(apad, x2, B, Mh, HOP, s);    // (B, Mh, 160) -> (B, 160, Mh)
  float* ft = A.alloc<float>((size_t)B * 2 * nb * F0);
  {
    ConvArgs a = conv1d_args(m.stft, x2, ft, B, Mh, F0, 1, 1, 0);
    c.conv_on(a, s);
  }
  float* mag = A.alloc<float>((size_t)B * nb * F0);
  launch_magnitude(ft, mag, B, nb, F0, s, 1e-9f);
  float* mel0 = A.alloc<float>((size_t)B * N_MELS * F0);
  {
    ConvArgs a = conv1d_args(m.melfb, mag, mel0, B, F0, F0);
    c.conv_on(a, s);
  }
  float* mel = A.alloc<float>((size_t)B * N_MELS * F);
  {
    const long tot = (long)B * N_MELS * F;
    hipLaunchKernelGGL(fcpe_mel_post_kernel, dim3((unsigned)cdiv64(tot, 256)), dim3(256), 0, s, mel0, mel, F0, F, tot);
  }
  if (mel_out) RVCX_HIP(hipMemcpyAsync(mel_out, mel, (size_t)B * N_MELS * F * sizeof(float), hipMemcpyDeviceToDevice, s));
  // ---- stack: conv3 -> GroupNorm(4) -> LeakyReLU(0.01) -> conv3   (FCPE.py:604-609)
  float* x = A.alloc<float>((size_t)B * C * F);
  float* y = A.alloc<float>((size_t)B * C * F);
  float* ln = A.alloc<float>((size_t)B * C * F);
  {
    ConvArgs a = conv1d_args(m.stack0, mel, y, B, F, F, 1, 1, 1);
    c.conv_on(a, s);
    double* st = A.alloc<double>((size_t)B * C * 2);
    hipLaunchKernelGGL(gn_stats_kernel, dim3(C, B), dim3(256), 0, s, y, st, C, F);
    hipLaunchKernelGGL(gn_apply_lrelu_kernel, dim3(C, B), dim3(256), 0, s, y, st, m.gn_g, m.gn_b, ln, C, F, 4, 1e-5f, 0.01f);
    a = conv1d_args(m.stack3, ln, x, B, F, F, 1, 1, 1);
    c.conv_on(a, s);
  }
  if (after_stack) (*after_stack)();
  // ---- PCmer encoder   (FCPE.py:227-268)
  const int ntile = fcpe_tiles(F);
  const long plen = (long)Mf * (PF_D + 1);
  float* qkv = A.alloc<float>((size_t)B * 3 * inner * F);
  float* att = A.alloc<float>((size_t)B * inner * F);
  float* part = A.alloc<float>((size_t)B * H * ntile * plen);
  float* ctx = A.alloc<float>((size_t)B * H * plen);
  const int ci = m.layers.empty() ? 0 : m.layers[0].pw1.cout / 2;
  float* glu = A.alloc<float>((size_t)B * 2 * ci * F);
  float* dw = A.alloc<float>((size_t)B * ci * F);
  const float dn = 1.0f / std::sqrt(std::sqrt((float)cf.dim_head)), ratio = 1.0f / std::sqrt((float)Mf);
  const size_t lds_kv = (size_t)std::max(PF_D * (PF_T + 1), Mf * PF_T) * sizeof(float);
  const size_t lds_q = (size_t)(std::max(PF_D + 2, Mf) + 4) * PF_T * sizeof(float);
  static std::once_flag attr_once;
  std::call_once(attr_once, [] {
    RVCX_HIP(hipFuncSetAttribute((const void*)performer_kv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)(PF_MAXM * PF_T * sizeof(float))));
    RVCX_HIP(hipFuncSetAttribute((const void*)performer_q_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)((PF_MAXM + 4) * PF_T * sizeof(float))));
  });
  for (const auto& L : m.layers) {
    // phone = phone + attn(norm(phone))
    launch_layernorm_c(x, L.ln_g, L.ln_b, ln, B, C, F, 1e-5f, nullptr, s);
    ConvArgs a = conv1d_args(L.qkv, ln, qkv, B, F, F);
    c.conv_on(a, s);
    const long qbs = (long)3 * inner * F;
    hipLaunchKernelGGL(performer_kv_kernel, dim3(ntile, H, B), dim3(256), lds_kv, s, qkv + (size_t)inner * F,
                       qkv + (size_t)2 * inner * F, qbs, F, L.proj, Mf, part, ntile, dn, ratio, 1e-4f);
    hipLaunchKernelGGL(performer_reduce_kernel, dim3(cdiv((int)plen, 256), B * H), dim3(256), 0, s, part, ctx, ntile, (int)plen);
    hipLaunchKernelGGL(performer_q_kernel, dim3(ntile, H, B), dim3(256), lds_q, s, qkv, qbs, F, L.proj, Mf, ctx, att,
                       (long)inner * F, dn, ratio, 1e-4f);
    c.flops += 2.0 * B * H * (double)F * Mf * PF_D * 4.0;
    a = conv1d_args(L.out, att, y, B, F, F);
    conv_set_res(a, x, C, F);
    c.conv_on(a, s);
    // phone = phone + conformer(phone)
    launch_layernorm_c(y, L.cln_g, L.cln_b, ln, B, C, F, 1e-5f, nullptr, s);
    a = conv1d_args(L.pw1, ln, glu, B, F, F);
    c.conv_on(a, s);
    {
      const int K = cf.dw_kernel, padl = K / 2;      // calc_same_padding: (K/2, K/2 - (K+1)%2), FCPE.py:271-273
      hipLaunchKernelGGL(glu_dwconv_swish_kernel, dim3(cdiv(F, DW_TT), ci, B), dim3(DW_TT),
                         (size_t)(DW_TT + 2 * K + 1) * sizeof(float), s, glu, L.dw_w, L.dw_b, dw, ci, F, K, padl);
    }
    a = conv1d_args(L.pw2, dw, x, B, F, F);
    conv_set_res(a, y, C, F);
    c.conv_on(a, s);
  }
  // ---- norm -> dense_out -> sigmoid -> local arg-max decoder   (FCPE.py:643-654)
  launch_layernorm_c(x, m.norm_g, m.norm_b, ln, B, C, F, 1e-5f, nullptr, s);
  float* sal = A.alloc<float>((size_t)B * F * cf.out_dims);
  {
    ConvArgs a = conv1d_args(m.dense, ln, sal, B, F, F);
    a.act = ACT_SIGMOID;
    a.out_mode = OUT_TRANSPOSED;
    a.y_bs = (long)F * cf.out_dims;
    a.y_cs = cf.out_dims;
    c.conv_on(a, s);
  }
  if (sal_out) RVCX_HIP(hipMemcpyAsync(sal_out, sal, (size_t)B * F * cf.out_dims * sizeof(float), hipMemcpyDeviceToDevice, s));
  hipLaunchKernelGGL(fcpe_decode_kernel, dim3((unsigned)cdiv64((long)B * F, 4)), dim3(256), 0, s, sal, m.cent_table, f0,
                     (long)B * F, cf.out_dims, threshold);
  RVCX_HIP(hipGetLastError());
}

void fcpe_post_coarse(Ctx& c, const float* f0raw, int B, int F_in, int p_len, float* f0_out, int* coarse, long out_stride,
                      double pitch, double f0_min, double f0_max, hipStream_t s) {
  RVCX_CHECK(p_len > 0 && F_in > 0, "fcpe: empty f0 track");
  float* tmp = c.arena.alloc<float>((size_t)B * p_len);
  int* pv = c.arena.alloc<int>((size_t)B * p_len);
  int* nx = c.arena.alloc<int>((size_t)B * p_len);
  const double mel_min = 1127.0 * std::log(1.0 + f0_min / 700.0), mel_max = 1127.0 * std::log(1.0 + f0_max / 700.0);
  hipLaunchKernelGGL(fcpe_post_kernel, dim3(B), dim3(1024), 0, s, f0raw, (long)F_in, F_in, p_len, tmp, pv, nx, (long)p_len,
                     f0_out, coarse, out_stride, std::pow(2.0, pitch / 12.0), mel_min, mel_max);
  RVCX_HIP(hipGetLastError());
}

}  // namespace rvcx
