// Non-GEMM kernels.  All memory-bound: coalesced along the contiguous (time) axis, one pass
// where the math allows, wavefront/LDS reductions for the normalisations.
#include <algorithm>
#include <cmath>
#include <vector>

#include "conv.h"
#include "h3_device.h"
#include "ops.h"

namespace rvcx {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ __forceinline__ float gelu_erf(float v) {
  return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
}

// ------------------------------------------------------------------ LayerNorm over channels
// block = 16 time steps x 16 channel slices (T is only a few thousand frames: small column groups keep
// every CU busy); each thread caches its <= LN_MAXC/16 channel values in registers: one global read.
// Reductions over the channel axis: a wavefront holds 4 channel slices of the 16 time steps (lane = tx + 16 p), so
// the partial sums of a time step are combined with two wavefront shuffles (xor 16, xor 32); the four waves of the
// block then exchange one value per time step through a 4 x 16 LDS array.
constexpr int LN_TX = 16, LN_PARTS = 16, LN_MAXC = 1024, LN_RPT = LN_MAXC / LN_PARTS;
__global__ __launch_bounds__(256) void layernorm_c_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ y,
                                                          int C, int T, float eps, const int* lens) {
  __shared__ float red[2][4][LN_TX];
  const int tx = threadIdx.x % LN_TX, part = threadIdx.x / LN_TX;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  auto block_sum = [&](float v, int slot) {     // sum over the 16 channel slices of time step tx
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    if (lane < LN_TX) red[slot][wave][tx] = v;
    __syncthreads();
    return (red[slot][0][tx] + red[slot][1][tx]) + (red[slot][2][tx] + red[slot][3][tx]);
  };
  const int b = blockIdx.y;
  const int t = blockIdx.x * LN_TX + tx;
  const bool valid = t < T;
  const float* xb = x + (long)b * C * T;
  float* yb = y + (long)b * C * T;
  float v[LN_RPT];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < LN_RPT; ++k) {
    const int c = part + k * LN_PARTS;
    v[k] = xb[(long)min(c, C - 1) * T + min(t, T - 1)];   // always in range: 64 independent loads in flight
  }
#pragma unroll
  for (int k = 0; k < LN_RPT; ++k) {
    const int c = part + k * LN_PARTS;
    v[k] = (valid && c < C) ? v[k] : 0.f;
    s += v[k];
  }
  const float mean = block_sum(s, 0) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < LN_RPT; ++k) {
    const int c = part + k * LN_PARTS;
    const float d = (c < C) ? v[k] - mean : 0.f;
    q += d * d;
  }
  const float rstd = 1.f / sqrtf(block_sum(q, 1) / (float)C + eps);
  if (!valid) return;
  const bool live = !lens || t < lens[b];
#pragma unroll
  for (int k = 0; k < LN_RPT; ++k) {
    const int c = part + k * LN_PARTS;
    if (c < C) {
      const float o = (v[k] - mean) * rstd * gamma[c] + beta[c];
      yb[(long)c * T + t] = live ? o : 0.f;
    }
  }
}

void launch_layernorm_c(const float* x, const float* gamma, const float* beta, float* y, int B, int C, int T,
                        float eps, const int* lens, hipStream_t s) {
  RVCX_CHECK(C <= LN_MAXC, "layernorm: too many channels");
  hipLaunchKernelGGL(layernorm_c_kernel, dim3(cdiv(T, LN_TX), B), dim3(256), 0, s, x, gamma, beta, y, C, T, eps, lens);
}

// ------------------------------------------------------------------ time-major section (gemm.hip)
// 4 channels c .. c+3 of row `row` into the fp16 hi/lo split form the GEMM stages (gemm.h: 64 bytes per 16 channels)
__device__ __forceinline__ bool store_xs4(void* ys, long ld_ys, long row, int c, const float (&v)[4]) {
  typedef _Float16 half4 __attribute__((ext_vector_type(4)));
  half4 hi, lo;
  bool ovf = false;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    ovf |= !(fabsf(v[q]) < kH3ActLimit);
    const _Float16 vh = (_Float16)v[q];
    hi[q] = vh;
    lo[q] = (_Float16)((v[q] - (float)vh) * 256.f);
  }
  char* e = static_cast<char*>(ys) + row * ld_ys + (c >> 4) * 64 + ((c >> 3) & 1) * 16 + (c & 7) * 2;
  *reinterpret_cast<half4*>(e) = hi;
  *reinterpret_cast<half4*>(e + 32) = lo;
  return ovf;
}

// LayerNorm over the channels of time-major rows: ONE WAVEFRONT PER ROW, the row lives in registers (float4 per lane
// and pass), both reductions are wavefront shuffles.  Writes fp32 rows and / or the split form for the next GEMM.
constexpr int LNT_MAXC = 1024;
__global__ __launch_bounds__(256) void layernorm_tm_kernel(const float* __restrict__ x, long ld_x,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ y,
                                                           long ld_y, void* __restrict__ ys, long ld_ys, long rows, int C,
                                                           float eps, int* ovf, int* ovf_next, int seq) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  constexpr int NP = LNT_MAXC / 256;
  float4 v[NP];
  const float4* xr = reinterpret_cast<const float4*>(x + row * ld_x);
  const int n4 = C >> 2;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int f = lane + 64 * k;
    v[k] = f < n4 ? xr[f] : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    if (lane + 64 * k < n4) {
      const float a = v[k].x - mean, b = v[k].y - mean, c = v[k].z - mean, d = v[k].w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float rstd = 1.f / sqrtf(wave_sum(q) / (float)C + eps);
  bool bad = false;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int f = lane + 64 * k;
    if (f < n4) {
      const float4 g = reinterpret_cast<const float4*>(gamma)[f], bb = reinterpret_cast<const float4*>(beta)[f];
      const float o[4] = {(v[k].x - mean) * rstd * g.x + bb.x, (v[k].y - mean) * rstd * g.y + bb.y,
                          (v[k].z - mean) * rstd * g.z + bb.z, (v[k].w - mean) * rstd * g.w + bb.w};
      if (y) reinterpret_cast<float4*>(y + row * ld_y)[f] = make_float4(o[0], o[1], o[2], o[3]);
      if (ys) bad |= store_xs4(ys, ld_ys, row, 4 * f, o);
    }
  }
  if (bad) {
    if (ovf) atomicOr(ovf, kErrH3Overflow);
    if (ovf_next) atomicMax(ovf_next, 0x7fffffff - seq);
  }
}

void launch_layernorm_tm(const float* x, long ld_x, const float* gamma, const float* beta, float* y, long ld_y, void* ys,
                         long ld_ys, long rows, int C, float eps, int* ovf, int* ovf_next, int seq, hipStream_t s) {
  RVCX_CHECK(C % 4 == 0 && C <= LNT_MAXC, "layernorm_tm: channels must be a multiple of 4 and <= 1024");
  hipLaunchKernelGGL(layernorm_tm_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, s, x, ld_x, gamma, beta, y, ld_y,
                     ys, ld_ys, rows, C, eps, ovf, ovf_next, seq);
}

// channel-first (B, C, T) fp32 -> time-major rows (r = b T + t): fp32 and / or split form.  64 x 64 tiles through LDS:
// reads coalesced along t, writes 16 bytes (4 channels) per lane along c.
__global__ __launch_bounds__(256) void cf_to_tm_kernel(const float* __restrict__ x, long x_bs, float* __restrict__ y,
                                                       long ld_y, void* __restrict__ ys, long ld_ys, int C, int T,
                                                       int* ovf, int* ovf_next, int seq) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z, c0 = blockIdx.y * 64, t0 = blockIdx.x * 64;
  const float* xb = x + (long)b * x_bs;
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int c = e >> 6, t = e & 63;
    tile[c][t] = (c0 + c < C && t0 + t < T) ? xb[(long)(c0 + c) * T + t0 + t] : 0.f;
  }
  __syncthreads();
  bool bad = false;
  for (int e = threadIdx.x; e < 64 * 16; e += 256) {
    const int t = e >> 4, c4 = (e & 15) * 4;
    if (t0 + t < T && c0 + c4 < C) {
      const float o[4] = {tile[c4][t], tile[c4 + 1][t], tile[c4 + 2][t], tile[c4 + 3][t]};
      const long row = (long)b * T + t0 + t;
      if (y) *reinterpret_cast<float4*>(y + row * ld_y + c0 + c4) = make_float4(o[0], o[1], o[2], o[3]);
      if (ys) bad |= store_xs4(ys, ld_ys, row, c0 + c4, o);
    }
  }
  if (bad) {
    if (ovf) atomicOr(ovf, kErrH3Overflow);
    if (ovf_next) atomicMax(ovf_next, 0x7fffffff - seq);
  }
}

void launch_cf_to_tm(const float* x, long x_bs, float* y, long ld_y, void* ys, long ld_ys, int B, int C, int T, int* ovf,
                     int* ovf_next, int seq, hipStream_t s) {
  RVCX_CHECK(C % 4 == 0, "cf_to_tm: channels must be a multiple of 4");
  hipLaunchKernelGGL(cf_to_tm_kernel, dim3(cdiv(T, 64), cdiv(C, 64), B), dim3(256), 0, s, x, x_bs, y, ld_y, ys, ld_ys, C, T,
                     ovf, ovf_next, seq);
}

// ------------------------------------------------------------------ GroupNorm(C,C) + GELU
// One block per (channel, item); two sweeps over the row (the second hits L2): fp64 sum / sum of squares in
// one sweep (exact enough that var = E[x^2] - mean^2 matches the two-pass fp32 result to rounding), then apply.
template <bool STATS_ONLY>
__global__ __launch_bounds__(512) void groupnorm_gelu_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ y,
                                                             int C, int Trow, float eps, const int* __restrict__ lens) {
  __shared__ double red[2][8];
  const int c = blockIdx.x, b = blockIdx.y;
  const float* xr = x + ((long)b * C + c) * Trow;
  float* yr = y + ((long)b * C + c) * Trow;
  const int T = lens ? lens[b] : Trow;        // the item's own frame count: sums, and their order, are those of its single run
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double s = 0.0, q = 0.0;
  int t = tid;
  for (; t + 3 * 512 < T; t += 4 * 512) {
    const float v0 = xr[t], v1 = xr[t + 512], v2 = xr[t + 1024], v3 = xr[t + 1536];
    s += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
    q += ((double)v0 * v0 + (double)v1 * v1) + ((double)v2 * v2 + (double)v3 * v3);
  }
  for (; t < T; t += 512) {
    const float v = xr[t];
    s += v;
    q += (double)v * v;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    s += __shfl_xor(s, d, 64);
    q += __shfl_xor(q, d, 64);
  }
  if (lane == 0) {
    red[0][wave] = s;
    red[1][wave] = q;
  }
  __syncthreads();
  double ts = 0.0, tq = 0.0;
#pragma unroll
  for (int w = 0; w < 8; ++w) {
    ts += red[0][w];
    tq += red[1][w];
  }
  const double meand = ts / T;
  const float mean = (float)meand;
  const float var = (float)fmax(tq / T - meand * meand, 0.0);
  const float rstd = 1.f / sqrtf(var + eps);
  if (STATS_ONLY) {        // y = (B, C, 2): the apply pass is groupnorm_gelu_split_kernel
    if (tid == 0) {
      y[((long)b * C + c) * 2] = mean;
      y[((long)b * C + c) * 2 + 1] = rstd;
    }
    return;
  }
  const float g = gamma[c], bb = beta[c];
  t = tid;
  for (; t + 3 * 512 < T; t += 4 * 512) {
    const float v0 = xr[t], v1 = xr[t + 512], v2 = xr[t + 1024], v3 = xr[t + 1536];
    yr[t] = gelu_erf((v0 - mean) * rstd * g + bb);
    yr[t + 512] = gelu_erf((v1 - mean) * rstd * g + bb);
    yr[t + 1024] = gelu_erf((v2 - mean) * rstd * g + bb);
    yr[t + 1536] = gelu_erf((v3 - mean) * rstd * g + bb);
  }
  for (; t < T; t += 512) yr[t] = gelu_erf((xr[t] - mean) * rstd * g + bb);
  for (t = T + tid; t < Trow; t += 512) yr[t] = 0.f;
}

// The apply pass when the consumer is a split-fp16 conv tile reading pre-split input (ConvArgs::x_split): a thread owns 8
// channels of one frame and stores them as one 16-byte element of fp16 hi parts and one of scaled lo parts,
// XS[c/16][op][(c%16)/8][t][8] -- the layout and the arithmetic of store_tile_split (h3_device.h).
__global__ __launch_bounds__(256) void groupnorm_gelu_split_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                                   const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, void* __restrict__ ys, int C,
                                                                   int Trow, const int* __restrict__ lens, int* ovf,
                                                                   int* ovf_layer, int seq) {
  const int t = blockIdx.x * 256 + threadIdx.x, cg = blockIdx.y, b = blockIdx.z;
  if (t >= Trow) return;
  const int T = lens ? lens[b] : Trow;
  const float* xr = x + ((long)b * C + cg * 8) * Trow + t;
  half8 hi, lo;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int c = cg * 8 + q;
    const float mean = stats[((long)b * C + c) * 2], rstd = stats[((long)b * C + c) * 2 + 1];
    float v = gelu_erf((xr[(long)q * Trow] - mean) * rstd * gamma[c] + beta[c]);
    v = t < T ? v : 0.f;
    if (!(fabsf(v) < kH3ActLimit)) {
      if (ovf) atomicOr(ovf, kErrH3Overflow);
      if (ovf_layer) atomicMax(ovf_layer, 0x7fffffff - seq);
    }
    const _Float16 vh = (_Float16)v;
    hi[q] = vh;
    lo[q] = (_Float16)((v - (float)vh) * kH3Scale);
  }
  uint4* base = reinterpret_cast<uint4*>(static_cast<char*>(ys) + (long)b * C * Trow * 4);
  const int chunk = cg >> 1, hh = cg & 1;
  base[(long)((chunk * 2 + 0) * 2 + hh) * Trow + t] = __builtin_bit_cast(uint4, hi);
  base[(long)((chunk * 2 + 1) * 2 + hh) * Trow + t] = __builtin_bit_cast(uint4, lo);
}

// ------------------------------------------------------------------ HuBERT conv0 + GroupNorm + GELU without the fp32 map
// Round 6.  The extractor's first layer (Cin = 1, k = 10, stride 5: ten FMAs per output) wrote a (B, 512, T0) fp32 map
// (210 MB per 32 s clip) that GroupNorm then read twice -- statistics, then normalise + GELU + split store -- 840 MB of HBM
// traffic for 0.5 GFLOP.  Here the map is never stored: the statistics pass and the apply pass both recompute the ten-tap
// FIR from the waveform (2 MB, cache-resident), with conv_cin1_kernel's arithmetic (fmaf over the taps in order), so
// what reaches HBM is the split image the next conv reads and nothing else.
//   stats: block = 16 channels of one item, 512 threads; per channel the sums are formed in EXACTLY the order of
//          groupnorm_gelu_kernel<true> (thread t: frames t, t + 512, ... in groups of four, fp64; xor-shuffle; waves 0 .. 7),
//          so {mean, rstd} -- and with them every feature -- are bit-identical to the three-pass form (RVCX_HUBERT_FUSE0=0).
//          (A first version summed 1024-frame slices: equal to 1 ulp of rstd, enough to flip a retrieval neighbour of the C3
//          golden's item 0 and move its waveform by 3.6e-5.)
//   apply: groupnorm_gelu_split_kernel with the FIR in place of the load.
// CB channels per block (the per-channel order of the sums does not depend on it): 16 for micro-batches; hubert.hip keeps the
// three-pass form for fewer than 256 such blocks (a single clip: 4.40 -> 4.62 ms with 32 blocks of 16, 4.48 with 128 of 4)
template <int CB>
__global__ __launch_bounds__(512) void hubert_conv0_stats_kernel(const float* __restrict__ wav, long wav_bs,
                                                                 const float* __restrict__ w, int K, int stride, int Cp, int C,
                                                                 int Trow, const int* __restrict__ lens, float* __restrict__ stats,
                                                                 float eps) {
  __shared__ float ws[16 * CB];             // [kk][CB channels]
  __shared__ double red[2][8][CB];
  const int c0 = blockIdx.x * CB, b = blockIdx.y;
  const int T = lens ? lens[b] : Trow;
  for (int idx = threadIdx.x; idx < K * CB; idx += 512) ws[idx] = w[(long)(idx / CB) * Cp + c0 + (idx % CB)];
  __syncthreads();
  const float* xb = wav + (long)b * wav_bs;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double s[CB], q[CB];
#pragma unroll
  for (int j = 0; j < CB; ++j) s[j] = q[j] = 0.0;
  auto fir = [&](int t, float (&acc)[CB]) {
#pragma unroll
    for (int j = 0; j < CB; ++j) acc[j] = 0.f;
    const float* xp = xb + (long)t * stride;
    for (int kk = 0; kk < K; ++kk) {
      const float xv = xp[kk];
#pragma unroll
      for (int j = 0; j < CB; ++j) acc[j] = fmaf(ws[kk * CB + j], xv, acc[j]);
    }
  };
  int t = tid;
  for (; t + 3 * 512 < T; t += 4 * 512) {
    float v0[CB], v1[CB], v2[CB], v3[CB];
    fir(t, v0);
    fir(t + 512, v1);
    fir(t + 1024, v2);
    fir(t + 1536, v3);
#pragma unroll
    for (int j = 0; j < CB; ++j) {
      s[j] += ((double)v0[j] + (double)v1[j]) + ((double)v2[j] + (double)v3[j]);
      q[j] += ((double)v0[j] * v0[j] + (double)v1[j] * v1[j]) + ((double)v2[j] * v2[j] + (double)v3[j] * v3[j]);
    }
  }
  for (; t < T; t += 512) {
    float v[CB];
    fir(t, v);
#pragma unroll
    for (int j = 0; j < CB; ++j) {
      s[j] += v[j];
      q[j] += (double)v[j] * v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < CB; ++j) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      s[j] += __shfl_xor(s[j], d, 64);
      q[j] += __shfl_xor(q[j], d, 64);
    }
    if (lane == 0) {
      red[0][wave][j] = s[j];
      red[1][wave][j] = q[j];
    }
  }
  __syncthreads();
  if (tid < CB && c0 + tid < C) {
    double ts = 0.0, tq = 0.0;
#pragma unroll
    for (int wv = 0; wv < 8; ++wv) {
      ts += red[0][wv][tid];
      tq += red[1][wv][tid];
    }
    const double meand = ts / T;
    const float mean = (float)meand;
    const float var = (float)fmax(tq / T - meand * meand, 0.0);
    const float rstd = 1.f / sqrtf(var + eps);
    stats[((long)b * C + c0 + tid) * 2] = mean;
    stats[((long)b * C + c0 + tid) * 2 + 1] = rstd;
  }
}

__global__ __launch_bounds__(256) void hubert_conv0_apply_kernel(const float* __restrict__ wav, long wav_bs,
                                                                 const float* __restrict__ w, int K, int stride, int Cp,
                                                                 const float* __restrict__ stats, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, void* __restrict__ ys, int C,
                                                                 int Trow, const int* __restrict__ lens, int* ovf, int* ovf_layer,
                                                                 int seq) {
  __shared__ float ws[16 * 8];               // [kk][8 channels]
  const int cg = blockIdx.y, b = blockIdx.z;
  for (int idx = threadIdx.x; idx < K * 8; idx += 256) ws[idx] = w[(long)(idx >> 3) * Cp + cg * 8 + (idx & 7)];
  __syncthreads();
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= Trow) return;
  const int T = lens ? lens[b] : Trow;
  float acc[8];
#pragma unroll
  for (int qq = 0; qq < 8; ++qq) acc[qq] = 0.f;
  if (t < T) {
    const float* xb = wav + (long)b * wav_bs + (long)t * stride;
    for (int kk = 0; kk < K; ++kk) {
      const float xv = xb[kk];
#pragma unroll
      for (int qq = 0; qq < 8; ++qq) acc[qq] = fmaf(ws[kk * 8 + qq], xv, acc[qq]);
    }
  }
  half8 hi, lo;
#pragma unroll
  for (int qq = 0; qq < 8; ++qq) {
    const int c = cg * 8 + qq;
    const float mean = stats[((long)b * C + c) * 2], rstd = stats[((long)b * C + c) * 2 + 1];
    float v = gelu_erf((acc[qq] - mean) * rstd * gamma[c] + beta[c]);
    v = t < T ? v : 0.f;
    if (!(fabsf(v) < kH3ActLimit)) {
      if (ovf) atomicOr(ovf, kErrH3Overflow);
      if (ovf_layer) atomicMax(ovf_layer, 0x7fffffff - seq);
    }
    const _Float16 vh = (_Float16)v;
    hi[qq] = vh;
    lo[qq] = (_Float16)((v - (float)vh) * kH3Scale);
  }
  uint4* base = reinterpret_cast<uint4*>(static_cast<char*>(ys) + (long)b * C * Trow * 4);
  const int chunk = cg >> 1, hh = cg & 1;
  base[(long)((chunk * 2 + 0) * 2 + hh) * Trow + t] = __builtin_bit_cast(uint4, hi);
  base[(long)((chunk * 2 + 1) * 2 + hh) * Trow + t] = __builtin_bit_cast(uint4, lo);
}

size_t hubert_conv0_part_doubles(int, int, int) { return 0; }      // (the statistics pass needs no scratch any more)

void launch_hubert_conv0_gn_gelu_split(const float* wav, long wav_bs, const float* w, int K, int stride, int Cp, const float* gamma,
                                       const float* beta, double* part, float* stats, void* y_split, int B, int C, int T,
                                       float eps, hipStream_t s, const int* lens, int* ovf, int* ovf_layer, int seq) {
  RVCX_CHECK(C % 16 == 0 && K <= 16, "hubert conv0 fusion: channels must be a multiple of 16, at most 16 taps");
  (void)part;
  if ((long)B * (C / 16) >= 256)
    hipLaunchKernelGGL(hubert_conv0_stats_kernel<16>, dim3(C / 16, B), dim3(512), 0, s, wav, wav_bs, w, K, stride, Cp, C, T, lens,
                       stats, eps);
  else
    hipLaunchKernelGGL(hubert_conv0_stats_kernel<4>, dim3(C / 4, B), dim3(512), 0, s, wav, wav_bs, w, K, stride, Cp, C, T, lens,
                       stats, eps);
  hipLaunchKernelGGL(hubert_conv0_apply_kernel, dim3(cdiv(T, 256), C / 8, B), dim3(256), 0, s, wav, wav_bs, w, K, stride, Cp,
                     stats, gamma, beta, y_split, C, T, lens, ovf, ovf_layer, seq);
}

void launch_groupnorm_gelu(const float* x, const float* gamma, const float* beta, float* y, int B, int C, int T,
                           float eps, hipStream_t s, const int* lens) {
  hipLaunchKernelGGL(groupnorm_gelu_kernel<false>, dim3(C, B), dim3(512), 0, s, x, gamma, beta, y, C, T, eps, lens);
}

void launch_groupnorm_gelu_split(const float* x, const float* gamma, const float* beta, float* stats, void* y_split, int B,
                                 int C, int T, float eps, hipStream_t s, const int* lens, int* ovf, int* ovf_layer, int seq) {
  RVCX_CHECK(C % 16 == 0, "groupnorm_gelu_split: channels must be a multiple of 16");
  hipLaunchKernelGGL(groupnorm_gelu_kernel<true>, dim3(C, B), dim3(512), 0, s, x, gamma, beta, stats, C, T, eps, lens);
  hipLaunchKernelGGL(groupnorm_gelu_split_kernel, dim3(cdiv(T, 256), C / 8, B), dim3(256), 0, s, x, stats, gamma, beta,
                     y_split, C, T, lens, ovf, ovf_layer, seq);
}

// ------------------------------------------------------------------ batched transpose
__global__ void transpose_kernel(const float* __restrict__ x, float* __restrict__ y, int R, int Cc) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const float* xb = x + (long)b * R * Cc;
  float* yb = y + (long)b * R * Cc;
  int c = blockIdx.x * 32 + threadIdx.x;
  for (int j = threadIdx.y; j < 32; j += 8) {
    int r = blockIdx.y * 32 + j;
    if (r < R && c < Cc) tile[j][threadIdx.x] = xb[(long)r * Cc + c];
  }
  __syncthreads();
  int r2 = blockIdx.y * 32 + threadIdx.x;
  for (int j = threadIdx.y; j < 32; j += 8) {
    int c2 = blockIdx.x * 32 + j;
    if (r2 < R && c2 < Cc) yb[(long)c2 * R + r2] = tile[threadIdx.x][j];
  }
}

void launch_transpose(const float* x, float* y, int B, int R, int Cc, hipStream_t s) {
  hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(Cc, 32), cdiv(R, 32), B), dim3(32, 8), 0, s, x, y, R, Cc);
}

// ------------------------------------------------------------------ element-wise glue
#define EW_GRID(n) dim3((unsigned)std::min<long>(cdiv64((n), 256), 1 << 20)), dim3(256)

__global__ void embed_pitch_kernel(float* x, const float* emb, const int* pitch, int C, int T, float scale,
                                   float slope, const int* lens, long total) {
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int t = idx % T;
    const long bc = idx / T;
    const int c = bc % C, b = bc / C;
    float v = (x[idx] + emb[(long)pitch[(long)b * T + t] * C + c]) * scale;
    v = v > 0.f ? v : v * slope;
    if (lens && t >= lens[b]) v = 0.f;
    x[idx] = v;
  }
}
void launch_embed_pitch(float* x, const float* emb, const int* pitch, int B, int C, int T, float scale,
                        float slope, const int* lens, hipStream_t s) {
  long n = (long)B * C * T;
  hipLaunchKernelGGL(embed_pitch_kernel, EW_GRID(n), 0, s, x, emb, pitch, C, T, scale, slope, lens, n);
}

__global__ void sample_z_kernel(const float* stats, const float* noise, float* z, int C, int T, const int* lens,
                                long total) {
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int t = idx % T;
    const long bc = idx / T;
    const int c = bc % C, b = bc / C;
    const float m = stats[((long)b * 2 * C + c) * T + t];
    const float logs = stats[((long)b * 2 * C + C + c) * T + t];
    float v = m + expf(logs) * noise[idx] * 0.66666f;
    if (lens && t >= lens[b]) v = 0.f;
    z[idx] = v;
  }
}
void launch_sample_z(const float* stats, const float* noise, float* z, int B, int C, int T, const int* lens,
                     hipStream_t s) {
  long n = (long)B * C * T;
  hipLaunchKernelGGL(sample_z_kernel, EW_GRID(n), 0, s, stats, noise, z, C, T, lens, n);
}

__global__ void flip_channels_kernel(const float* x, float* y, int C, int T, long total) {
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int t = idx % T;
    const long bc = idx / T;
    const int c = bc % C;
    const long b = bc / C;
    y[idx] = x[(b * C + (C - 1 - c)) * T + t];
  }
}
void launch_flip_channels(const float* x, float* y, int B, int C, int T, hipStream_t s) {
  long n = (long)B * C * T;
  hipLaunchKernelGGL(flip_channels_kernel, EW_GRID(n), 0, s, x, y, C, T, n);
}

__global__ void wn_gate_kernel(const float* a, const float* g, int goff, int gstride, float* acts, int H, int T,
                               long total) {
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int t = idx % T;
    const long bc = idx / T;
    const int c = bc % H;
    const long b = bc / H;
    const float ta = a[(b * 2 * H + c) * T + t] + g[b * gstride + goff + c];
    const float sa = a[(b * 2 * H + H + c) * T + t] + g[b * gstride + goff + H + c];
    acts[idx] = tanhf(ta) * (1.f / (1.f + expf(-sa)));
  }
}
void launch_wn_gate(const float* a, const float* g, int goff, int gstride, float* acts, int B, int H, int T,
                    hipStream_t s) {
  long n = (long)B * H * T;
  hipLaunchKernelGGL(wn_gate_kernel, EW_GRID(n), 0, s, a, g, goff, gstride, acts, H, T, n);
}

__global__ void wn_res_skip_kernel(float* x, float* out, const float* rs, int H, int T, int last, int first,
                                   const int* lens, long total) {
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int t = idx % T;
    const long bc = idx / T;
    const int c = bc % H;
    const long b = bc / H;
    const bool live = !lens || t < lens[b];
    if (!last) {
      const float r = rs[(b * 2 * H + c) * T + t];
      const float sk = rs[(b * 2 * H + H + c) * T + t];
      x[idx] = live ? (x[idx] + r) : 0.f;
      out[idx] = first ? sk : out[idx] + sk;
    } else {
      const float sk = rs[(b * H + c) * T + t];
      const float o = first ? sk : out[idx] + sk;
      out[idx] = live ? o : 0.f;   // WaveNet returns output * x_mask
    }
  }
}
void launch_wn_res_skip(float* x, float* out, const float* rs, int B, int H, int T, int last, int first,
                        const int* lens, hipStream_t s) {
  long n = (long)B * H * T;
  hipLaunchKernelGGL(wn_res_skip_kernel, EW_GRID(n), 0, s, x, out, rs, H, T, last, first, lens, n);
}

__global__ void coupling_sub_kernel(float* x, const float* m, int half, int T, const int* lens, long total) {
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int t = idx % T;
    const long bc = idx / T;
    const int c = bc % half;
    const long b = bc / half;
    const long xi = (b * 2 * half + half + c) * T + t;
    float v = x[xi] - m[idx];
    if (lens && t >= lens[b]) v = 0.f;
    x[xi] = v;
  }
}
void launch_coupling_sub(float* x, const float* m, int B, int half, int T, const int* lens, hipStream_t s) {
  long n = (long)B * half * T;
  hipLaunchKernelGGL(coupling_sub_kernel, EW_GRID(n), 0, s, x, m, half, T, lens, n);
}

__global__ void mask_kernel(float* x, int C, int T, const int* lens, long total) {
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int t = idx % T;
    const long b = idx / ((long)C * T);
    if (t >= lens[b]) x[idx] = 0.f;
  }
}
void launch_mask(float* x, int B, int C, int T, const int* lens, hipStream_t s) {
  if (!lens) return;
  long n = (long)B * C * T;
  hipLaunchKernelGGL(mask_kernel, EW_GRID(n), 0, s, x, C, T, lens, n);
}

__global__ void add_channel_bias_kernel(float* x, const float* g, int T, long total) {
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256)
    x[idx] += g[idx / T];
}
void launch_add_channel_bias(float* x, const float* g, int B, int C, int T, hipStream_t s) {
  long n = (long)B * C * T;
  hipLaunchKernelGGL(add_channel_bias_kernel, EW_GRID(n), 0, s, x, g, T, n);
}

// ------------------------------------------------------------------ NSF harmonic source
// phase prefix per frame (float64): P[t] = frac(sum_{t'<t} upp * rad[t']), rad = (f0/sr) % 1 in fp32.
// sin(2*pi*cumsum) is invariant to the integer "cumsum_shift" of generators.py:140-146, so the
// per-sample phase is P[t] + (jj+1)*rad[t] without ever materialising the 1.5M-sample cumsum.
// One wave per item: lane l owns a contiguous segment of frames (local wrapped sum, wave-level exclusive
// scan of the 64 segment totals, second pass writes the prefixes).  Subtracting floor() only removes
// integers, so the wrapped running sum equals the sequential one up to fp64 rounding (1e-16 of a cycle).
__global__ __launch_bounds__(64) void sine_prefix_kernel(const float* f0, double* P, float* rad_out, int T, int upp,
                                                         float sr) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int seg = (T + 63) / 64;
  const int t0 = min(lane * seg, T), t1 = min(t0 + seg, T);
  const float* f = f0 + (long)b * T;
  double acc = 0.0;
  for (int t = t0; t < t1; ++t) {
    acc += (double)fmodf(f[t] / sr, 1.0f) * upp;
    acc -= floor(acc);
  }
  // inclusive scan over lanes, then shift to exclusive
  double incl = acc;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double o = __shfl_up(incl, d, 64);
    if (lane >= d) incl += o;
    incl -= floor(incl);
  }
  double run = incl - acc;
  run -= floor(run);
  for (int t = t0; t < t1; ++t) {
    const float rad = fmodf(f[t] / sr, 1.0f);
    P[(long)b * T + t] = run;
    rad_out[(long)b * T + t] = rad;
    run += (double)rad * upp;
    run -= floor(run);
  }
}

__global__ void sine_source_kernel(const float* f0, const double* P, const float* rad, const float* noise,
                                   float* har, int T, int upp, const float* lin_wb, const int* lens,
                                   long total) {
  const long per = (long)T * upp;
  const float lin_w = lin_wb[0], lin_b = lin_wb[1];
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long b = idx / per;
    const long j = idx - b * per;
    const int t = j / upp, jj = j - (long)t * upp;
    const float f = f0[b * T + t];
    double ph = P[b * T + t] + (double)(jj + 1) * (double)rad[b * T + t];
    ph -= floor(ph);
    const float sine = (float)sin(ph * 6.283185307179586476925) * 0.1f;
    const float uv = f > 0.f ? 1.f : 0.f;
    const float namp = uv * 0.003f + (1.f - uv) * 0.1f / 3.f;
    float v = sine * uv + namp * noise[idx];
    v = tanhf(lin_w * v + lin_b);
    if (lens && t >= lens[b]) v = 0.f;
    har[idx] = v;
  }
}

void launch_sine_source(const float* f0, const float* noise, float* har, int B, int T, int upp, float sr,
                        const float* lin_wb, const int* lens, double* scratch, hipStream_t s) {
  double* P = scratch;
  float* rad = reinterpret_cast<float*>(scratch + (size_t)B * T);
  hipLaunchKernelGGL(sine_prefix_kernel, dim3(B), dim3(64), 0, s, f0, P, rad, T, upp, sr);
  long n = (long)B * T * upp;
  hipLaunchKernelGGL(sine_source_kernel, EW_GRID(n), 0, s, f0, P, rad, noise, har, T, upp, lin_wb, lens, n);
}

// ------------------------------------------------------------------ Philox4x32-10 normal noise
__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0,
                                             uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
  c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

__global__ void randn_kernel(float* out, size_t n, uint64_t seed, uint64_t offset) {
  const size_t nq = (n + 3) / 4;
  for (size_t q = blockIdx.x * 256UL + threadIdx.x; q < nq; q += (size_t)gridDim.x * 256) {
    const uint64_t ctr = offset + q;
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0x52564358u, c3 = 0u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      philox_round(c0, c1, c2, c3, k0, k1);
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
    const float u0 = ((float)c0 + 0.5f) * 2.3283064365386963e-10f, u1 = ((float)c1 + 0.5f) * 2.3283064365386963e-10f;
    const float u2 = ((float)c2 + 0.5f) * 2.3283064365386963e-10f, u3 = ((float)c3 + 0.5f) * 2.3283064365386963e-10f;
    const float r0 = sqrtf(-2.f * logf(u0)), r1 = sqrtf(-2.f * logf(u2));
    float v[4] = {r0 * cosf(6.2831853f * u1), r0 * sinf(6.2831853f * u1), r1 * cosf(6.2831853f * u3),
                  r1 * sinf(6.2831853f * u3)};
    for (int e = 0; e < 4; ++e)
      if (q * 4 + e < n) out[q * 4 + e] = v[e];
  }
}
void launch_randn(float* out, size_t n, uint64_t seed, uint64_t offset, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(randn_kernel, EW_GRID((long)((n + 3) / 4)), 0, s, out, n, seed, offset);
}

// ------------------------------------------------------------------ RMVPE helpers
__global__ void reflect_pad_kernel(const float* x, float* y, int n, int p, long y_bs, long total, const int* ns,
                                   long x_bs) {
  const int np2 = n + 2 * p;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long b = idx / np2;
    const long o = idx - b * np2;
    const int nb = ns ? ns[b] : n;
    // numpy's "reflect" for ANY pad width (np.pad reflects repeatedly when p >= n; VC.pipeline pads a 0.4 s clip by 1 s,
    // pipeline.py:348): the source index is the triangle wave of period 2 (n - 1)
    int j = (int)o - p;
    if (j < 0) j = -j;
    if (j >= nb) {
      const int per = nb > 1 ? 2 * (nb - 1) : 1;
      j %= per;
      if (j >= nb) j = per - j;
    }
    y[b * y_bs + o] = o < nb + 2 * p ? x[b * x_bs + j] : 0.f;
  }
}
void launch_reflect_pad(const float* x, float* y, int B, int n, int p, long y_bs, hipStream_t s, const int* ns, long x_bs) {
  long tot = (long)B * (n + 2 * p);
  hipLaunchKernelGGL(reflect_pad_kernel, EW_GRID(tot), 0, s, x, y, n, p, y_bs, tot, ns, x_bs > 0 ? x_bs : (long)n);
}

namespace {
struct IntPack {
  int v[32];
};
__global__ void set_ints_kernel(int* dst, IntPack pk, int n) {
  if ((int)threadIdx.x < n) dst[threadIdx.x] = pk.v[threadIdx.x];
}
}  // namespace
void set_dev_ints(int* d, const int* v, int n, hipStream_t s) {
  for (int o = 0; o < n; o += 32) {
    IntPack pk;
    const int m = std::min(32, n - o);
    for (int i = 0; i < 32; ++i) pk.v[i] = i < m ? v[o + i] : 0;
    hipLaunchKernelGGL(set_ints_kernel, dim3(1), dim3(32), 0, s, d + o, pk, m);
  }
}
int* dev_ints(Arena& A, const int* v, int n, hipStream_t s) {
  int* d = A.alloc<int>((size_t)std::max(n, 1));
  set_dev_ints(d, v, n, s);
  return d;
}

namespace {
struct IntPack8 {
  int v[256];
};
__global__ void set_ints8_kernel(int* dst, IntPack8 pk, int n) {
  if ((int)threadIdx.x < n) dst[threadIdx.x] = pk.v[threadIdx.x];
}
}  // namespace
// several arrays behind one launch (a model's per-level length arrays: HuBERT needs 8, the F0 model 9 -- one tiny launch
// each was 40-55 us of serial path at the head of either model)
std::vector<int*> dev_ints_many(Arena& A, const std::vector<std::vector<int>>& vs, hipStream_t s) {
  size_t total = 0;
  for (const auto& v : vs) total += std::max<size_t>(v.size(), 1);
  int* d = A.alloc<int>(total);
  std::vector<int> flat(total, 0);
  std::vector<int*> out;
  size_t o = 0;
  for (const auto& v : vs) {
    out.push_back(d + o);
    std::copy(v.begin(), v.end(), flat.begin() + o);
    o += std::max<size_t>(v.size(), 1);
  }
  for (size_t c = 0; c < total; c += 256) {
    IntPack8 pk;
    const int m = (int)std::min<size_t>(256, total - c);
    for (int i = 0; i < 256; ++i) pk.v[i] = i < m ? flat[c + i] : 0;
    hipLaunchKernelGGL(set_ints8_kernel, dim3(1), dim3(256), 0, s, d + c, pk, m);
  }
  return out;
}

__global__ void magnitude_kernel(const float* ft, float* mag, int nb, int F, long total, float eps) {
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long b = idx / ((long)nb * F);
    const long r = idx - b * nb * F;
    const float re = ft[b * 2 * nb * F + r], im = ft[b * 2 * nb * F + (long)nb * F + r];
    mag[idx] = sqrtf(re * re + im * im + eps);
  }
}
void launch_magnitude(const float* ft, float* mag, int B, int nb, int F, hipStream_t s, float eps) {
  long tot = (long)B * nb * F;
  hipLaunchKernelGGL(magnitude_kernel, EW_GRID(tot), 0, s, ft, mag, nb, F, tot, eps);
}

__global__ void mel_post_kernel(const float* mel, float* out, int nmel, int F, int Tp, const float* bn,
                                long total, const int* fs, const int* tps) {
  const int Wp = nmel + 2;
  const float sc = bn[0], sh = bn[1];
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int col = idx % Wp;
    const long bt = idx / Wp;
    const int t = bt % Tp;
    const long b = bt / Tp;
    const int Fb = fs ? fs[b] : F, Tb = tps ? tps[b] : Tp;
    float v = 0.f;
    if (col > 0 && col < Wp - 1 && t < Tb) {
      const int src = t < Fb ? t : 2 * Fb - 2 - t;   // F.pad(mel, (0,pad), "reflect"), RMVPE.py:465
      const float mv = mel[(b * nmel + (col - 1)) * F + src];
      v = logf(fmaxf(mv, 1e-5f)) * sc + sh;
    }
    out[idx] = v;
  }
}
void launch_mel_post(const float* mel, float* out, int B, int nmel, int F, int Tp, const float* bn,
                     hipStream_t s, const int* fs, const int* tps) {
  long tot = (long)B * Tp * (nmel + 2);
  hipLaunchKernelGGL(mel_post_kernel, EW_GRID(tot), 0, s, mel, out, nmel, F, Tp, bn, tot, fs, tps);
}

__global__ void log_clamp_kernel(const float* x, float* y, long n, float floor) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = logf(fmaxf(x[i], floor));
}
void launch_log_clamp(const float* x, float* y, long n, float floor, hipStream_t s) {
  hipLaunchKernelGGL(log_clamp_kernel, EW_GRID(n), 0, s, x, y, n, floor);
}

__global__ void avgpool2_kernel(const float* x, float* y, int H, int Wp, long x_ps, long y_ps, long total) {
  const int W2 = (Wp - 2) / 2, Wpo = W2 + 2, H2 = H / 2;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int col = idx % Wpo;
    const long pr = idx / Wpo;
    const int i = pr % H2;
    const long p = pr / H2;
    float v = 0.f;
    if (col > 0 && col < Wpo - 1) {
      const float* r0 = x + p * x_ps + (long)(2 * i) * Wp + 1 + 2 * (col - 1);
      const float* r1 = r0 + Wp;
      v = (((r0[0] + r0[1]) + r1[0]) + r1[1]) * 0.25f;
    }
    y[p * y_ps + (long)i * Wpo + col] = v;
  }
}
void launch_avgpool2(const float* x, float* y, int planes, int H, int Wp, long x_ps, long y_ps, hipStream_t s) {
  long tot = (long)planes * (H / 2) * ((Wp - 2) / 2 + 2);
  hipLaunchKernelGGL(avgpool2_kernel, EW_GRID(tot), 0, s, x, y, H, Wp, x_ps, y_ps, tot);
}

__global__ void copy_strided_kernel(const float* x, float* y, long n, long x_bs, long y_bs, long total) {
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long b = idx / n, r = idx - b * n;
    y[b * y_bs + r] = x[b * x_bs + r];
  }
}
void launch_copy_strided(const float* x, float* y, int B, long n, long x_bs, long y_bs, hipStream_t s) {
  long tot = (long)B * n;
  hipLaunchKernelGGL(copy_strided_kernel, EW_GRID(tot), 0, s, x, y, n, x_bs, y_bs, tot);
}

__global__ void gru_input_kernel(const float* x, float* y, int Cc, int T, int Wp, long total) {
  const int W = Wp - 2;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int t = idx % T;
    const long bk = idx / T;
    const int kf = bk % (Cc * W);
    const long b = bk / (Cc * W);
    const int c = kf / W, f = kf % W;
    y[idx] = x[((b * Cc + c) * T + t) * Wp + 1 + f];
  }
}
void launch_gru_input(const float* x, float* y, int B, int Cc, int T, int Wp, hipStream_t s) {
  long tot = (long)B * Cc * (Wp - 2) * T;
  hipLaunchKernelGGL(gru_input_kernel, EW_GRID(tot), 0, s, x, y, Cc, T, Wp, tot);
}

// to_local_average_cents + decode + range gate.  numpy semantics reproduced: first-max argmax,
// float32 salience x float64 cents mapping, float32 weight sum in numpy's 8-lane pairwise order.
// Round 6: one WAVE per frame (a thread per frame read its 360 bins with a stride of `ld` floats between neighbouring
// lanes: 32 us for 3201 frames, on the single clip's critical path and once per utterance in a batch).  Lane l scans bins
// l, l + 64, ... (its first maximum), the wave reduces (value, index) with ties to the lower index -- the same first-max
// argmax -- and lane 0 forms the nine-bin average exactly as before: same bits.
__global__ __launch_bounds__(256) void decode_f0_kernel(const float* sal, float* f0, int T, int ld, float thred, float f0_min,
                                                        float f0_max, long total) {
  const int lane = threadIdx.x & 63;
  for (long idx = blockIdx.x * 4L + (threadIdx.x >> 6); idx < total; idx += (long)gridDim.x * 4) {
    const float* s = sal + idx * ld;
    int center = lane;
    float mx = lane < 360 ? s[lane] : -INFINITY;
    for (int i = lane + 64; i < 360; i += 64) {
      const float v = s[i];
      if (v > mx) {
        mx = v;
        center = i;
      }
    }
    for (int o = 32; o > 0; o >>= 1) {
      const float v2 = __shfl_xor(mx, o);
      const int c2 = __shfl_xor(center, o);
      if (v2 > mx || (v2 == mx && c2 < center)) {
        mx = v2;
        center = c2;
      }
    }
    if (lane != 0) continue;
    float w[9];
    double pw[9];
    for (int k = 0; k < 9; ++k) {
      const int bin = center - 4 + k;
      const bool in = bin >= 0 && bin < 360;
      w[k] = in ? s[bin] : 0.f;
      const double cm = in ? (20.0 * bin + 1997.3794084376191) : 0.0;
      pw[k] = (double)w[k] * cm;
    }
    const double psum = (((pw[0] + pw[1]) + (pw[2] + pw[3])) + ((pw[4] + pw[5]) + (pw[6] + pw[7]))) + pw[8];
    const float wsum = (((w[0] + w[1]) + (w[2] + w[3])) + ((w[4] + w[5]) + (w[6] + w[7]))) + w[8];
    double cents = psum / (double)wsum;
    if (mx <= thred) cents = 0.0;
    double f = 10.0 * pow(2.0, cents / 1200.0);
    if (f == 10.0) f = 0.0;
    if (f < (double)f0_min || f > (double)f0_max) f = 0.0;
    f0[idx] = (float)f;
  }
}
void launch_decode_f0(const float* sal, float* f0, int B, int T, int ld, float thred, float f0_min, float f0_max,
                      hipStream_t s) {
  long tot = (long)B * T;
  hipLaunchKernelGGL(decode_f0_kernel, dim3((unsigned)std::min<long>((tot + 3) / 4, 65535)), dim3(256), 0, s, sal, f0, T, ld, thred,
                     f0_min, f0_max, tot);
}

__global__ void f0_coarse_kernel(const float* f0_in, float* f0_out, int* coarse, int n, double shift, double mel_min,
                                 double mel_max) {
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < n; idx += gridDim.x * 256) {
    const double f = (double)f0_in[idx] * shift;
    double m = 1127.0 * log(1.0 + f / 700.0);
    if (m > 0.0) m = (m - mel_min) * 254.0 / (mel_max - mel_min) + 1.0;
    if (m <= 1.0) m = 1.0;
    if (m > 255.0) m = 255.0;
    coarse[idx] = (int)rint(m);
    f0_out[idx] = (float)f;
  }
}
void launch_f0_coarse(const float* f0_in, float* f0_out, int* coarse, int n, double pitch, double f0_min,
                      double f0_max, hipStream_t s) {
  const double mel_min = 1127.0 * std::log(1.0 + f0_min / 700.0), mel_max = 1127.0 * std::log(1.0 + f0_max / 700.0);
  hipLaunchKernelGGL(f0_coarse_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, f0_in, f0_out, coarse, n,
                     std::pow(2.0, pitch / 12.0), mel_min, mel_max);
}

// f0 file (pipeline.py:185-191): frames [start, start + count) take the track read from the file -- after the pitch
// shift, like the reference -- and are quantised with the same float64 formula
__global__ void f0_override_kernel(const double* rep, int count, int start, float* f0, int* coarse, int n, double mel_min,
                                   double mel_max) {
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < count; idx += gridDim.x * 256) {
    const int t = start + idx;
    if (t >= n) return;
    const double f = rep[idx];
    double m = 1127.0 * log(1.0 + f / 700.0);
    if (m > 0.0) m = (m - mel_min) * 254.0 / (mel_max - mel_min) + 1.0;
    if (m <= 1.0) m = 1.0;
    if (m > 255.0) m = 255.0;
    coarse[t] = (int)rint(m);
    f0[t] = (float)f;
  }
}
void launch_f0_override(const double* rep, int count, int start, float* f0, int* coarse, int n, double f0_min,
                        double f0_max, hipStream_t s) {
  if (count <= 0 || start >= n) return;
  const double mel_min = 1127.0 * std::log(1.0 + f0_min / 700.0), mel_max = 1127.0 * std::log(1.0 + f0_max / 700.0);
  hipLaunchKernelGGL(f0_override_kernel, dim3(cdiv(count, 256)), dim3(256), 0, s, rep, count, start, f0, coarse, n, mel_min,
                     mel_max);
}

// The track VC.get_f0 builds from an f0 file's (time [s], f0 [Hz]) rows (pipeline.py:186-189), float32 table in, float64
// out: delta_t = int16(round((t.max() - t.min()) * 100 + 1)) in float32 arithmetic, then np.interp(range(delta_t),
// t * 100 [float32], f0) with numpy's own rules (flat extrapolation, exact hits return the sample, last of equal xp).
std::vector<double> f0_file_track(const float* tbl, int rows) {
  if (!tbl || rows <= 0) return {};
  float tmin = tbl[0], tmax = tbl[0];
  for (int i = 1; i < rows; ++i) {
    tmin = std::fmin(tmin, tbl[2 * i]);
    tmax = std::fmax(tmax, tbl[2 * i]);
  }
  const float span = (tmax - tmin) * 100.0f + 1.0f;
  const int delta_t = (int)(short)std::nearbyint(span);      // np.round (half to even) -> astype("int16")
  if (delta_t <= 0) return {};
  std::vector<double> xp((size_t)rows), fp((size_t)rows);
  for (int i = 0; i < rows; ++i) {
    xp[i] = (double)(tbl[2 * i] * 100.0f);
    fp[i] = (double)tbl[2 * i + 1];
  }
  std::vector<double> out((size_t)delta_t);
  for (int q = 0; q < delta_t; ++q) {
    const double x = (double)q;
    double r;
    if (x < xp[0]) {
      r = fp[0];
    } else if (x > xp[rows - 1]) {
      r = fp[rows - 1];
    } else {
      const int j = (int)(std::upper_bound(xp.begin(), xp.end(), x) - xp.begin()) - 1;
      if (j >= rows - 1 || xp[j] == x) {
        r = fp[std::min(j, rows - 1)];
      } else {
        const double slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
        r = slope * (x - xp[j]) + fp[j];
        if (std::isnan(r)) {
          r = slope * (x - xp[j + 1]) + fp[j + 1];
          if (std::isnan(r) && fp[j] == fp[j + 1]) r = fp[j];
        }
      }
    }
    out[q] = r;
  }
  return out;
}

__global__ void upsample_protect_kernel(const float* feats, const float* feats0, const float* pitchf, float* out,
                                        int Th, int p_len, float protect, int use_protect, long total, int ld_out) {
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int t = idx % p_len;
    const long c = idx / p_len;
    const int src = t >> 1;
    float v = feats[c * Th + src];
    if (use_protect) {
      const float ff = pitchf[t] < 1.f ? protect : 1.f;
      v = v * ff + feats0[c * Th + src] * (1.f - ff);
    }
    out[c * ld_out + t] = v;
  }
}
void launch_upsample_protect(const float* feats, const float* feats0, const float* pitchf, float* out, int C,
                             int Th, int p_len, float protect, int use_protect, hipStream_t s, int ld_in, int ld_out) {
  long tot = (long)C * p_len;
  hipLaunchKernelGGL(upsample_protect_kernel, EW_GRID(tot), 0, s, feats, feats0, pitchf, out, ld_in > 0 ? ld_in : Th, p_len,
                     protect, use_protect, tot, ld_out > 0 ? ld_out : p_len);
}

}  // namespace rvcx
