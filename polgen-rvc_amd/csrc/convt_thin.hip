// ConvTranspose1d(C -> C/2, k = 4, stride 2, padding 1) of the two last NSF stages (rvc/lib/algorithm/nsf.py:96-101,
// 126-129: x = ups[i](leaky_relu(x, 0.1)); x = x + noise_convs[i](har_source)), C = 128 and 64 over 384 k / 768 k input
// positions per 30 s clip -- as a STREAMING kernel.
//
// As a polyphase conv the layer has K = 2 taps x C input channels and C packed output channels (channel c x phase): 32
// FLOP per byte of compulsory traffic, below the ridge of the split-fp16 arithmetic (~55) -- it is bound by reading its
// input and writing its output once (393 MB each way at C = 64: ~80 us at HBM speed).  On the general conv tile
// (conv_h3<64,64>) it ran at 1.4 TB/s (218 / 297 us): a workgroup staged a 64-position tile through LDS behind two
// barriers for eight MFMA k-steps, and the epilogue scattered 4-byte stores at stride 2 and read the noise conv's output
// (another 196 MB written and read back per stage) as its residual.  Here:
//   * the weights (hi/lo fp16 image, 32 KB / 128 KB) sit in LDS for the whole launch;
//   * every wave owns 32 positions at a time and feeds the MFMA B operand STRAIGHT from global memory: lane (i, h) loads
//     x[16 chunk + 8 h + 0..7][q0 + i + tap - 1] -- eight dword loads, each a coalesced 128-byte row segment per half wave --
//     converts in registers and issues the chunk's MFMAs; loads run two chunks (and across tiles) ahead; no barriers;
//   * accumulator rows are (c, phase 0), (c, phase 1), (c + 1, phase 0), (c + 1, phase 1): a lane holds both phases of a
//     channel = two CONSECUTIVE output samples, stored as one 8-byte word; consecutive lanes are consecutive words;
//   * the stage's noise conv (Cin = 1, k = 4 stride 2 / k = 1) is evaluated in the epilogue from the harmonic source
//     (6 MB, cache resident) instead of being written to HBM by one kernel and read back by this one.
// Arithmetic: the split-fp16 products of conv_h3 (same split, same three MFMAs per block; chunks ascending, taps
// ascending), fp32 FMA chain for the noise taps (ascending), then (conv + bias) + (noise + bias) as the reference adds them.
#include <cstdlib>
#include <mutex>
#include <type_traits>

#include "conv.h"
#include "conv_device.h"
#include "h3_device.h"

namespace rvcx {

namespace {

// C = 128: one workgroup of 8 waves per CU (130 KB of LDS, 228 registers: two waves per SIMD).  C = 64: three workgroups of 4
// waves (33 KB, 132 registers: three waves per SIMD).
template <int CIN, int kThreads>
__global__ __launch_bounds__(kThreads, kThreads == 256 ? 3 : 2) void convt_thin_kernel(const ConvArgs a) {
  constexpr int NCH = CIN / 16, MB = CIN / 32;          // packed output channels = CIN (2 phases x CIN / 2)
  constexpr int W_ELEMS = 2 * NCH * 4 * CIN;            // [tap][chunk][op][h][co] 16-byte elements
  extern __shared__ uint4 lds[];
  uint4* Ws = lds;
  float* nzs = reinterpret_cast<float*>(Ws + W_ELEMS);  // noise taps [j][c] (4 x CIN / 2), noise bias [c], conv bias [cg]
  const int tid = threadIdx.x, wave = tid >> 6;
  int lane = tid & 63, i = lane & 31, h = lane >> 5;      // re-derived per tile from an opaque copy (see the tile loop)
  const int b = blockIdx.y;
  constexpr int CR = CIN / 2;                           // real output channels
  {
    const uint4* src = static_cast<const uint4*>(a.w_h3);
    for (int e = tid; e < W_ELEMS; e += kThreads) Ws[e] = src[e];
    for (int e = tid; e < 4 * CR; e += kThreads) {
      const int j = e / CR, c = e - j * CR;
      nzs[e] = (a.nz_har && j < a.nz_k) ? a.nz_w[(long)j * a.nz_wstride + c] : 0.f;
    }
    for (int c = tid; c < CR; c += kThreads) nzs[4 * CR + c] = (a.nz_har && a.nz_b) ? a.nz_b[c] : 0.f;
    for (int c = tid; c < CIN; c += kThreads) nzs[5 * CR + c] = a.bias ? a.bias[c] : 0.f;
  }
  __syncthreads();
  const int len_in = a.lens_in ? a.lens_in[b] : a.Tin;
  const int len_out = a.lens_out ? a.lens_out[b] : 0x7fffffff;
  const int nz_len = a.nz_lens ? a.nz_lens[b] : a.nz_len;
  const H3Rsrc xr = h3_rsrc(a.x + (long)b * a.x_bs, CIN * a.x_cs * 4);
  const int xrow = a.x_cs * 4;
  const float slope = a.pre_act == ACT_LRELU ? a.pre_slope : 1.f;
  constexpr float inv = 1.f / kH3Scale;
  const int ntiles = (a.Nout + 30) / 31;                // 31 new positions per tile (column 0 overlaps the previous tile)
  const int nwaves = gridDim.x * (kThreads / 64);
  bool ovf = false;

  // Raw input of one (tile, chunk) step: the 8 channels of this lane's k-slice at its own position q (tap 1).  Tap 0 reads
  // x[q - 1] -- the SAME values one lane to the left: it is formed from the converted halves by a whole-wave DPP shift
  // (wave_shr:1).  The first lane of each half wave (i = 0) has no left neighbour: tiles therefore advance by 31
  // positions and column 0 of every tile is the overlap column -- loaded for its neighbour's sake, its own outputs
  // discarded (3 % of the MFMA columns).  Half the loads and half the conversions of loading both taps.
  // Requests: G = 4 chunks (32 dword loads, 8 KB per wave) at a time into one of TWO buffers, one request ahead of the
  // arithmetic.  This is the pattern the compiler's wait-count insertion gets right across the loop's back edge: "wait for
  // everything but the request just issued" (vmcnt(32)).  A ring of single-chunk requests seven steps deep -- the first
  // design -- compiled to exactly that wait too, i.e. to ONE chunk in flight: 2.2 - 3.1 TB/s.  Loads the epilogue needs
  // (harmonic source) are issued BEFORE the tile's request for the same reason: loads return in order.
  constexpr int G = 4;
  float raw[2][G][8];
  auto request = [&](auto buf_tag, int tile, int chunk0) {
    constexpr int buf = decltype(buf_tag)::value;
    const int q = tile * 31 + i - 1;
    const bool ok = tile < ntiles && q >= 0 && q < len_in;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int base = ((chunk0 + g) * 16 + 8 * h) * xrow + q * 4;
#pragma unroll
      for (int j = 0; j < 8; ++j) raw[buf][g][j] = h3_load1(xr, ok ? base + j * xrow : kH3Oob);
    }
  };
  auto split8 = [&](const float* v, half8& hi, half8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float t = v[j];
      t = fmaxf(t, t * slope);                                   // leaky_relu ahead of the layer (0 <= slope <= 1)
      ovf |= !(fabsf(t) < kH3ActLimit);
      const _Float16 th = (_Float16)t;
      hi[j] = th;
      lo[j] = (_Float16)((t - (float)th) * kH3Scale);
    }
  };
  // lane l <- lane l - 1 over the whole wave
  auto shift_right = [&](const half8& v) -> half8 {
    typedef int int4v __attribute__((ext_vector_type(4)));
    int4v a4 = __builtin_bit_cast(int4v, v), r4;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
#if defined(__HIP_DEVICE_COMPILE__)
      r4[d] = __builtin_amdgcn_update_dpp(0, a4[d], 0x138, 0xf, 0xf, false);
#else
      r4[d] = a4[d];
#endif
    }
    return __builtin_bit_cast(half8, r4);
  };
  f32x16 acc[MB];
  int tile = blockIdx.x * (kThreads / 64) + wave;
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  const uint4* Wt = Ws;
  const float* nzt = nzs;
  float hv[2][4];
  // per tile: opaque copies (see below), zero accumulators, the harmonic-source samples of the tile's two output phases
  auto head = [&](int tile_) {
    // The LDS images never change and a lane's addresses never change, so every weight / bias read and every address term
    // is loop-invariant: left alone the compiler hoists all 2 NCH MB fragment pairs and ~100 address registers out of the
    // tile loop (1119 registers spilled at C = 128 in the first build).  An opaque zero offset and an opaque copy of the
    // lane id per tile keep reads and address arithmetic inside the loop (LDS stays LDS: offsets, not pointers).
    int woff = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(woff), "+v"(lane));
#endif
    i = lane & 31;
    h = lane >> 5;
    Wt = Ws + woff;
    nzt = nzs + woff;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    const int q = tile_ * 31 + i - 1;
    const int t0 = 2 * q - a.sh_pad;
    const bool okq = i >= 1 && q < a.Nout;
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long p = (long)(t0 + ph) * a.nz_stride + j - a.nz_pad;
        const bool okh = a.nz_har && j < a.nz_k && p >= 0 && p < nz_len && okq && t0 + ph >= 0 && t0 + ph < a.sh_tout;
        hv[ph][j] = okh ? a.nz_har[(long)b * a.nz_bs + p] : 0.f;
      }
  };
  auto compute = [&](auto buf_tag, auto chunk0_tag) {
    constexpr int buf = decltype(buf_tag)::value, chunk0 = decltype(chunk0_tag)::value;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      constexpr int dummy = 0;
      (void)dummy;
      const int c = chunk0 + g;
      half8 xh[2], xl[2];
      split8(raw[buf][g], xh[1], xl[1]);
      xh[0] = shift_right(xh[1]);                                 // garbage in the lanes with i == 0: the overlap column
      xl[0] = shift_right(xl[1]);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        // fence: the scheduler may not pull the next block's fragment reads (or the next chunk's conversion) above this
        // block's MFMAs -- unfenced it front-loads every LDS read of the tile and spills hundreds of registers
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MB; ++m) {
          const half8 whs = __builtin_bit_cast(half8, Wt[(((kk * NCH + c) * 2 + 0) * 2 + h) * CIN + m * 32 + i]);
          const half8 wls = __builtin_bit_cast(half8, Wt[(((kk * NCH + c) * 2 + 1) * 2 + h) * CIN + m * 32 + i]);
          const half8 wh = whs * (_Float16)inv;
          acc[m] = h3_mfma(whs, xh[kk], acc[m]);          // (S wh) xh
          acc[m] = h3_mfma(wh, xl[kk], acc[m]);           // wh (S xl)
          acc[m] = h3_mfma(wls, xh[kk], acc[m]);          // (S wl) xh
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // lane (i, h) holds packed rows 8 g + 4 h + {0, 1, 2, 3} of every 32-row block = channels c0, c0 + 1, both phases
  auto epilogue = [&](int tile_) {
    const int q = tile_ * 31 + i - 1;
    const int t0 = 2 * q - a.sh_pad;                      // output position of phase 0; phase 1 is t0 + 1
    const bool okq = i >= 1 && q < a.Nout;
    const bool ok0 = okq && t0 >= 0 && t0 < a.sh_tout, ok1 = okq && t0 + 1 >= 0 && t0 + 1 < a.sh_tout;
    float* yb = a.y + (long)b * a.y_bs;
    const float* rb = a.res ? a.res + (long)b * a.res_bs : nullptr;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          const int cg = m * 32 + 8 * g + 4 * h + 2 * cc;       // packed row of phase 0; channel = cg / 2
          const int c = cg >> 1;
          float o[2];
#pragma unroll
          for (int ph = 0; ph < 2; ++ph) {
            float v = acc[m][4 * g + 2 * cc + ph] * inv + nzt[5 * CR + cg + ph];
            if (a.nz_har) {
              float n = 0.f;
#pragma unroll
              for (int j = 0; j < 4; ++j) n = fmaf(nzt[j * CR + c], hv[ph][j], n);     // taps beyond nz_k carry zero weights
              v += n + nzt[4 * CR + c];
            }
            o[ph] = v;
          }
          float* yp = yb + (long)c * a.y_cs + t0;
          if (rb) {
            const float* rp = rb + (long)c * a.res_cs + t0;
            if (ok0) o[0] += rp[0];
            if (ok1) o[1] += rp[1];
          }
          if (t0 >= len_out) o[0] = 0.f;
          if (t0 + 1 >= len_out) o[1] = 0.f;
          if (ok0 && ok1) {
            *reinterpret_cast<float2*>(yp) = make_float2(o[0], o[1]);       // 4-byte aligned 8-byte store (t0 is odd)
          } else {
            if (ok0) yp[0] = o[0];
            if (ok1) yp[1] = o[1];
          }
        }
  };
  static_assert(NCH == G || NCH == 2 * G, "a tile is one or two requests");
  using C0 = std::integral_constant<int, 0>;
  using C4 = std::integral_constant<int, G>;
  if (tile < ntiles) request(B0{}, tile, 0);
  if constexpr (NCH == G) {
    // one request per tile: the buffers alternate by tile, so the loop body is written out for both parities
    while (tile < ntiles) {
      head(tile);
      request(B1{}, tile + nwaves, 0);
      compute(B0{}, C0{});
      epilogue(tile);
      tile += nwaves;
      if (tile >= ntiles) break;
      head(tile);
      request(B0{}, tile + nwaves, 0);
      compute(B1{}, C0{});
      epilogue(tile);
      tile += nwaves;
    }
  } else {
    for (; tile < ntiles; tile += nwaves) {
      head(tile);
      request(B1{}, tile, G);
      compute(B0{}, C0{});
      request(B0{}, tile + nwaves, 0);
      compute(B1{}, C4{});
      epilogue(tile);
    }
  }
  if (ovf) report_h3_overflow(a.ovf, a.ovf_layer, a.seq);
}

size_t lds_bytes(int cin) { return (size_t)2 * (cin / 16) * 4 * cin * 16 + (size_t)(5 * (cin / 2) + cin) * 4; }

}  // namespace

bool convt_thin_ok(const ConvArgs& a) {
  static const bool on = !getenv("RVCX_CONVT_THIN") || atoi(getenv("RVCX_CONVT_THIN")) != 0;
  if (!on || !a.w_h3 || a.out_mode != OUT_SHUF1D || a.sh_s != 2 || a.sh_pad != 1 || a.ksize != 2 || a.kw != 2 || a.pad != 1 ||
      a.stride != 1 || a.dil != 1 || a.groups != 1)
    return false;
  if (!(a.Cin_g == 64 || a.Cin_g == 128) || a.Cin_gp != a.Cin_g || a.Cout_g != a.Cin_g || a.Cout_gp != a.Cin_g) return false;
  if (a.x_split || a.y_split || a.acc2_mode != ACC2_NONE || a.act != ACT_NONE || a.zero_wp != 0) return false;
  if (!(a.pre_act == ACT_NONE || (a.pre_act == ACT_LRELU && a.pre_slope >= 0.f && a.pre_slope <= 1.f))) return false;
  if (a.nz_har && (a.nz_k < 1 || a.nz_k > 4)) return false;
  if ((long)a.Cin_g * a.x_cs * 4 >= kH3Oob) return false;
  return conv_h3_enabled();
}

void convt_thin_init() {
  static std::mutex mu;
  static uint64_t done = 0;
  int dev = 0;
  RVCX_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> g(mu);
  if ((done >> (dev & 63)) & 1) return;
  RVCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(convt_thin_kernel<64, 256>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds_bytes(64)));
  RVCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(convt_thin_kernel<128, 512>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds_bytes(128)));
  done |= 1ull << (dev & 63);
}

void launch_convt_thin(const ConvArgs& a, hipStream_t stream) {
  convt_thin_init();
  static const int ncu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return std::max(1, n);
  }();
  const int ntiles = cdiv(a.Nout, 31);
  const int per_cu = a.Cin_g == 64 ? 3 : 1, threads = a.Cin_g == 64 ? 256 : 512;
  const int wgs = std::max(1, std::min(ncu * per_cu, cdiv(ntiles, threads / 64)));
  dim3 grid(wgs, a.B);
  if (a.Cin_g == 64) hipLaunchKernelGGL((convt_thin_kernel<64, 256>), grid, dim3(256), lds_bytes(64), stream, a);
  else hipLaunchKernelGGL((convt_thin_kernel<128, 512>), grid, dim3(512), lds_bytes(128), stream, a);
  RVCX_HIP(hipGetLastError());
}

}  // namespace rvcx
