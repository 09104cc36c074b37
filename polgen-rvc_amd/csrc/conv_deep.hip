// Convs with a long K and few output positions -- the deep levels of the RMVPE U-Net (rvc/lib/predictors/RMVPE.py:140-320:
// 3 x 3 convs with 128 / 256 / 512 channels on 7488 / 2080 / 624 positions, ~60 launches per clip) and their 2 x 2
// polyphase ConvTranspose2d -- with the split-K INSIDE the workgroup.
//
// On the general tile (conv_h3<64,64>) every one of these launches cost 24 - 41 us whatever its shape and split-K factor
// (round 4: "the floor"): a workgroup is a serial chain of load -> LDS -> barrier -> MFMA stages, the K = 1152 ... 4608
// reduction is spread over 8 workgroups per output tile that meet again in a second launch (conv_splitk_finish_kernel:
// 70 launches of 8 us per clip), and at B = 1 there are too few positions to hide any of it.  Here a workgroup of W = 8 or 16
// waves owns a 64 x 32 output tile and the WAVES split K: wave w walks its own contiguous range of (chunk, tap) k-steps,
// loading both MFMA operands straight from global memory into registers (weights: the fragment-ordered hi/lo image, 16
// bytes per lane; input: eight coalesced dword loads per lane, converted in registers) -- no LDS staging, no barrier in the
// k-loop, requests one group of k-steps ahead (the pattern the wait-count insertion handles: convt_thin.hip).  The W
// partial tiles meet in LDS, are summed in a fixed order (wave 0, 1, ...: deterministic, and the same for every batch size:
// W depends on the layer's shape only) and leave through the common epilogue (bias, activation, residual, length and
// pad-column masks, shuffle stores).  One launch per conv, no partial-sum slabs in HBM.
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <type_traits>

#include "conv.h"
#include "conv_device.h"
#include "h3_device.h"

namespace rvcx {

namespace {

// W waves; kG k-steps per request; MB 32-row blocks of output channels per workgroup (tile 32 MB x 32)
// DB: double-buffered requests (one group in flight while one is computed); false: request a group of kG steps, wait, compute it
// -- a wave then pays one operand round trip per kG steps, and the other waves of the CU fill its wait
template <int W, int kG, int MB, bool DB = true>
__global__ __launch_bounds__(64 * W) void conv_deep_kernel(const ConvArgs a) {
  extern __shared__ float red[];                  // [W][MB][16][64] partial accumulators
  const int tid = threadIdx.x, lane = tid & 63;
#if defined(__HIP_DEVICE_COMPILE__)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: the k-step bookkeeping stays on the scalar unit
#else
  const int wave = tid >> 6;
#endif
  const int i = lane & 31, h = lane >> 5;
  const int b = blockIdx.z;
  const int n0 = blockIdx.x * 32, co0 = blockIdx.y * (32 * MB);
  const int len_in = a.lens_in ? a.lens_in[b] : a.Tin;
  const int len_out = a.lens_out ? a.lens_out[b] : 0x7fffffff;
  const bool dead = n0 >= len_out && a.out_mode == OUT_NORMAL;     // every output of the tile is masked: nothing to compute
  const int nchunk = a.Cin_gp / 16;
  const int KS = a.ksize * nchunk;                // k-steps of the layer, step s = chunk * ksize + tap
  const int per = (KS + W - 1) / W;
  const int s_begin = wave * per, s_end = min(KS, s_begin + per);
  const H3Rsrc xr = h3_rsrc(a.x + (long)b * a.x_bs, a.Cin_gp * a.x_cs * 4);
  const H3Rsrc wr = h3_rsrc(a.w_h3, a.ksize * nchunk * 4 * a.Cout_gp * 16);
  const int xrow = a.x_cs * 4;
  const int slab = 4 * a.Cout_gp * 16;            // bytes of one (tap, chunk) weight slab: [op][h][co] 16-byte elements
  constexpr float inv = 1.f / kH3Scale;
  bool ovf = false;

  float rawb[DB ? 2 : 1][kG][8];
  uint4 rawa[DB ? 2 : 1][kG][MB][2];              // [buf][step][m][op]
  // requests are issued for consecutive steps: (chunk, tap row, tap column) of the next step to request are counters
  int rs = s_begin, rchunk = s_begin / a.ksize, rkk = s_begin - rchunk * a.ksize, rky = rkk / a.kw, rkx = rkk - rky * a.kw;
  auto request = [&](auto buf_tag) {
    constexpr int buf = decltype(buf_tag)::value;
#pragma unroll
    for (int g = 0; g < kG; ++g) {
      const int s = rs;
      const bool live = s < s_end && !dead;
      const int chunk = rchunk, kk = rkk;
      const int pos = n0 + i + rky * a.rowpitch + rkx * a.dil - a.pad;
      ++rs;
      ++rkk;
      if (++rkx == a.kw) {
        rkx = 0;
        ++rky;
      }
      if (rkk == a.ksize) {
        rkk = rky = rkx = 0;
        ++rchunk;
      }
      const bool okx = live && pos >= 0 && pos < len_in;
      const int xb = (chunk * 16 + 8 * h) * xrow + pos * 4;
#pragma unroll
      for (int j = 0; j < 8; ++j) rawb[buf][g][j] = h3_load1(xr, okx ? xb + j * xrow : kH3Oob);
      const int wb = (kk * nchunk + chunk) * slab + (h * a.Cout_gp + co0 + i) * 16;
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int op = 0; op < 2; ++op)
          rawa[buf][g][m][op] = h3_load4(wr, live ? wb + (op * 2 * a.Cout_gp + m * 32) * 16 : kH3Oob);
    }
  };
  f32x16 acc[MB];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  auto compute = [&](auto buf_tag) {
    constexpr int buf = decltype(buf_tag)::value;
#pragma unroll
    for (int g = 0; g < kG; ++g) {
      half8 xh, xl;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = rawb[buf][g][j];
        ovf |= !(fabsf(v) < kH3ActLimit);
        const _Float16 vh = (_Float16)v;
        xh[j] = vh;
        xl[j] = (_Float16)((v - (float)vh) * kH3Scale);
      }
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        const half8 whs = __builtin_bit_cast(half8, rawa[buf][g][m][0]);
        const half8 wls = __builtin_bit_cast(half8, rawa[buf][g][m][1]);
        const half8 wh = whs * (_Float16)inv;
        acc[m] = h3_mfma(whs, xh, acc[m]);        // (S wh) xh
        acc[m] = h3_mfma(wh, xl, acc[m]);         // wh (S xl)
        acc[m] = h3_mfma(wls, xh, acc[m]);        // (S wl) xh
      }
    }
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  // groups of kG steps, double-buffered: request the next group, compute the current one ("everything but the request
  // just issued": the wait the compiler can count); steps past the wave's range load nothing and add zeros
  if constexpr (DB) {
    request(B0{});
    for (int s = s_begin; s < s_end; s += 2 * kG) {
      request(B1{});
      compute(B0{});
      request(B0{});
      compute(B1{});
    }
  } else {
    for (int s = s_begin; s < s_end; s += kG) {
      request(B0{});
      compute(B0{});
    }
  }
  if (ovf) report_h3_overflow(a.ovf, a.ovf_layer, a.seq);
  // ---- the W partial tiles meet in LDS
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((wave * MB + m) * 16 + r) * 64 + lane] = acc[m][r];
  __syncthreads();
  // thread t sums element t (and t + 64 W, ...) of the MB x 16 x 64 tile over the waves in order, then the common epilogue
  for (int e = tid; e < MB * 16 * 64; e += 64 * W) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < W; ++w) v += red[w * (MB * 16 * 64) + e];
    const int ln = e & 63, r = (e >> 6) & 15, m = e >> 10;
    const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
    const int nn = n0 + (ln & 31);
    if (co < a.Cout_g && nn < a.Nout) store_elem(a, b, co, nn, v * inv, len_out);
  }
}

}  // namespace

bool conv_deep_ok(const ConvArgs& a) {
  // OFF by default: measured (round 5, tools/bench_deep.py and the C2 bench, one box) it does not break the floor it was
  // built against -- 512 -> 512 on 624 positions 41 -> 48 us, 256 -> 256 on 2080 31 -> 24, 128 -> 128 on 7488 33 -> 31; F0 stage
  // 8.46 -> 8.61 ms.  Cutting the dependent round trips per wave from 18 to 3 (forms 2 / 3: six or four k-steps per request)
  // made it SLOWER (59 / 37 / 30 us), which names the real bound: L2 -> CU operand traffic.  With N = 624 positions no tiling
  // has both enough workgroups and enough reuse -- a 64 x 32 tile without LDS sharing re-reads 2 x 590 KB of operands per
  // workgroup (380 MB per conv), the 64 x 64 LDS tile 190 MB: 20 - 40 us at the ~10 TB/s the L2s deliver, whatever the schedule.
  // RVCX_CONV_DEEP=1 turns it on (tests/test_gpu_modes.py keeps it correct).
  static const bool on = getenv("RVCX_CONV_DEEP") && atoi(getenv("RVCX_CONV_DEEP")) != 0;
  static const int max_n = getenv("RVCX_CONV_DEEP_N") ? atoi(getenv("RVCX_CONV_DEEP_N")) : 8192;
  if (!on || !a.w_h3 || !conv_h3_enabled()) return false;
  if (a.groups != 1 || a.stride != 1 || a.Cin_gp % 16 != 0 || a.Cout_gp % 64 != 0 || a.Cin_g != a.Cin_gp) return false;
  if (a.x_split || a.y_split || a.pre_act != ACT_NONE || a.acc2_mode != ACC2_NONE || a.nz_har) return false;
  if (!(a.out_mode == OUT_NORMAL || a.out_mode == OUT_SHUF2D)) return false;
  // a long reduction over few positions PER ITEM (the decision must not depend on the batch size: the summation order is
  // part of the result) -- 2-D maps only (kw < ksize): the 1-D layers of this size are served by the time-major GEMM path
  if (a.kw >= a.ksize || a.Nout > max_n || a.ksize * (a.Cin_gp / 16) < 64) return false;
  if ((long)a.Cin_gp * a.x_cs * 4 >= kH3Oob || (long)a.ksize * a.Cin_gp * a.Cout_gp * 4 >= kH3Oob) return false;
  return true;
}

void launch_conv_deep(const ConvArgs& a, hipStream_t stream) {
  static std::mutex mu;
  static uint64_t done = 0;
  int dev = 0;
  RVCX_HIP(hipGetDevice(&dev));
  {
    std::lock_guard<std::mutex> g(mu);
    if (!((done >> (dev & 63)) & 1)) {
      RVCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_deep_kernel<16, 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   16 * 2 * 16 * 64 * 4));
      RVCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_deep_kernel<16, 2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   16 * 1 * 16 * 64 * 4));
      RVCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_deep_kernel<8, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   8 * 2 * 16 * 64 * 4));
      RVCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_deep_kernel<16, 6, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   16 * 1 * 16 * 64 * 4));
      RVCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_deep_kernel<16, 4, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   16 * 1 * 16 * 64 * 4));
      RVCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_deep_kernel<8, 5, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   8 * 1 * 16 * 64 * 4));
      done |= 1ull << (dev & 63);
    }
  }
  const int KS = a.ksize * (a.Cin_gp / 16);
  dim3 grid(cdiv(a.Nout, 32), a.Cout_gp / 64, a.B);
  static const int form = getenv("RVCX_CONV_DEEP_FORM") ? atoi(getenv("RVCX_CONV_DEEP_FORM")) : 0;
  // 16 waves when every wave still gets >= 8 k-steps, else 8 (a function of the layer's shape only)
  if (form == 2 || form == 3) {             // batch requests: 6 (or 4) steps per round trip, 32 x 32 tiles
    grid.y = a.Cout_gp / 32;
    if (KS < 128) hipLaunchKernelGGL((conv_deep_kernel<8, 5, 1, false>), grid, dim3(512), 8 * 1 * 16 * 64 * 4, stream, a);
    else if (form == 2) hipLaunchKernelGGL((conv_deep_kernel<16, 6, 1, false>), grid, dim3(1024), 16 * 1 * 16 * 64 * 4, stream, a);
    else hipLaunchKernelGGL((conv_deep_kernel<16, 4, 1, false>), grid, dim3(1024), 16 * 1 * 16 * 64 * 4, stream, a);
  } else if (KS >= 128 && form == 1) {          // 32 x 32 tiles: twice the workgroups (the 512-channel level: 320 instead of 160), two steps per request
    grid.y = a.Cout_gp / 32;
    hipLaunchKernelGGL((conv_deep_kernel<16, 2, 1>), grid, dim3(1024), 16 * 1 * 16 * 64 * 4, stream, a);
  } else if (KS >= 128) {
    hipLaunchKernelGGL((conv_deep_kernel<16, 1, 2>), grid, dim3(1024), 16 * 2 * 16 * 64 * 4, stream, a);
  } else {
    hipLaunchKernelGGL((conv_deep_kernel<8, 2, 2>), grid, dim3(512), 8 * 2 * 16 * 64 * 4, stream, a);
  }
  RVCX_HIP(hipGetLastError());
}

}  // namespace rvcx
