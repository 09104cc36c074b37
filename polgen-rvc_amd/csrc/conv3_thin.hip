// 3x3 convs of the shallow U-Net levels (RMVPE.py:140-175, C = 16 / 32 on 420 160 / 106 656 positions) as a STREAMING
// kernel on the split-fp16 matrix cores.
//
// On the tiled kernel (conv_h3<32,128,halo320>) these 28 launches per clip take 41 / 23 us each -- 49 / 87 TFLOP/s --
// because a workgroup's life is a latency chain: stage a (128 + 2 Wp + 2)-column input tile through registers into LDS,
// two barriers per tap group, 27 MFMAs per wave, epilogue.  The layers are HBM-bound (54-81 MB per launch: 13-20 us at
// 4 TB/s) and K is only 9 x 16, so nothing needs to be shared between waves except the 18 / 36 KB of weights:
//   * every WAVE owns tiles of 30 output positions and feeds the MFMA B operand straight from global memory: lane (i, h)
//     loads the 8 channels 8 h .. 8 h + 7 of position q0 + i for each of the three map rows (q0 = p0 - 1 + (dy - 1) Wp) --
//     two 16-byte loads per row when the producer stored the split form (ConvArgs::x_split: the element IS the B
//     fragment), eight dwords + the conversion otherwise;
//   * the taps dx = 1, 2 of a row are the same registers shifted by one / two lanes (whole-wave DPP shift, wave_shl:1),
//     so columns 30 and 31 of a tile are overlap columns and tiles advance by 30 positions;
//   * the weights sit in LDS in fragment order for the whole (persistent) workgroup; no barrier after the prologue;
//   * the k-order is the tiled kernel's (chunks ascending, taps ascending, {S wh xh, wh S xl, S wl xh}) and the epilogue IS
//     the tiled kernel's (conv_device.h / conv_h3.hip: bias, ReLU, residual, length mask, pad-column zeroing, split store),
//     so the results are bit-identical to conv_h3 (tests/test_gpu_conv.py).
#include <algorithm>
#include <cstdlib>
#include <mutex>

#include "conv.h"
#include "conv_device.h"
#include "h3_device.h"

namespace rvcx {

namespace {

constexpr int kTileN = 30;     // useful output positions of a 32-lane tile

__device__ __forceinline__ half8 shl1(half8 v) {        // lane j <- lane j + 1 (whole wave)
  uint4 u = __builtin_bit_cast(uint4, v);
#if defined(__HIP_DEVICE_COMPILE__)
  u.x = (unsigned)__builtin_amdgcn_update_dpp(0, (int)u.x, 0x130, 0xf, 0xf, false);
  u.y = (unsigned)__builtin_amdgcn_update_dpp(0, (int)u.y, 0x130, 0xf, 0xf, false);
  u.z = (unsigned)__builtin_amdgcn_update_dpp(0, (int)u.z, 0x130, 0xf, 0xf, false);
  u.w = (unsigned)__builtin_amdgcn_update_dpp(0, (int)u.w, 0x130, 0xf, 0xf, false);
#endif
  return __builtin_bit_cast(half8, u);
}

// CIN: input channels (16 / 32; Cout <= 32: one 32-row block).  XS: the input arrives split (ConvArgs::x_split).
template <int CIN, bool XS>
__global__ __launch_bounds__(256, 4) void conv3_thin_kernel(const ConvArgs a) {
  constexpr int NCH = CIN / 16;
  extern __shared__ uint4 As[];                      // [(chunk * 9 + kk) * 2 + op][lane]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  // ---- weights: fragment order, rows >= Cout_gp are zero
  {
    const H3Rsrc wres = h3_rsrc(a.w_h3, 9 * NCH * 4 * a.Cout_gp * 16);
    const int slab = 4 * a.Cout_gp * 16;
    for (int e = tid; e < NCH * 9 * 2 * 64; e += 256) {
      const int l = e & 63, f = e >> 6;              // f = (chunk * 9 + kk) * 2 + op
      const int op = f & 1, ck = f >> 1, chunk = ck / 9, kk = ck - chunk * 9;
      const int ii = l & 31, hh = l >> 5;
      As[e] = h3_load4(wres, ii < a.Cout_gp ? (kk * NCH + chunk) * slab + ((op * 2 + hh) * a.Cout_gp + ii) * 16 : kH3Oob);
    }
  }
  __syncthreads();

  const int len_in = a.lens_in ? a.lens_in[b] : a.Tin;
  const int len_out = a.lens_out ? a.lens_out[b] : 0x7fffffff;
  const int Wp = a.rowpitch;
  const H3Rsrc xr = XS ? h3_rsrc(static_cast<const char*>(a.x_split) + (long)b * a.x_bs * 4, CIN * a.x_cs * 4)
                       : h3_rsrc(a.x + (long)b * a.x_bs, CIN * a.x_cs * 4);
  const int ntiles = (a.Nout + kTileN - 1) / kTileN;
  // tiles in XCD-contiguous ranges: workgroup w runs on XCD w % 8, so the three rows a tile reads were (or will be) read
  // by tiles of the same L2
  const int xcd = blockIdx.x & 7, wg_x = blockIdx.x >> 3, nwg_x = gridDim.x >> 3;
  const int per = (ntiles + 7) >> 3;
  const int t_begin = xcd * per, t_end = min(ntiles, t_begin + per);
  constexpr float inv = 1.f / kH3Scale;
  bool ovf = false;

  // One tile at a time per wave, 16 waves per CU.  (Measured and dropped: a software pipeline inside the wave -- the raw input of
  // tile t + 1 and the residual of tile t requested ahead of the MFMAs of tile t -- needs 125 - 237 registers: at 12 / 8
  // waves per CU the launches take 34.7 / 26.4 us instead of 28.9 / 21.4, and 416 / 262 instead of 359 / 220 at B = 16.)
  for (int tile = t_begin + wg_x * 4 + wave; tile < t_end; tile += nwg_x * 4) {
    const int p0 = tile * kTileN;
    // the lane's coordinates are re-derived from an opaque copy per tile: everything that depends on them only (the weight
    // fragment reads, the epilogue's per-channel addresses and bias values) is loop-invariant and would be hoisted into
    // ~250 registers otherwise (the persistent-loop lesson of resblock.hip)
    int lz = lane;
    asm volatile("" : "+v"(lz));
    const int i = lz & 31, h = lz >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (p0 < len_out) {
      half8 bq[3][NCH][2];
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int q = p0 - 1 + i + (dy - 1) * Wp;
        const bool ok = q >= 0 && q < len_in;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          if constexpr (XS) {
#pragma unroll
            for (int op = 0; op < 2; ++op)
              bq[dy][c][op] = __builtin_bit_cast(
                  half8, h3_load4(xr, ok ? (((c * 2 + op) * 2 + h) * a.x_cs + q) * 16 : kH3Oob));
          } else {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = h3_load1(xr, ok ? ((c * 16 + 8 * h + k) * a.x_cs + q) * 4 : kH3Oob);
            half8 hi, lo;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              ovf |= !(fabsf(v[k]) < kH3ActLimit);
              const _Float16 vh = (_Float16)v[k];
              hi[k] = vh;
              lo[k] = (_Float16)((v[k] - (float)vh) * kH3Scale);
            }
            bq[dy][c][0] = hi;
            bq[dy][c][1] = lo;
          }
        }
      }
      // weight fragments one tap ahead of their MFMAs (the 54 / 27 MFMAs of a tile are one dependent chain on `acc`: what
      // hides their latency is the other waves of the SIMD, what must not be exposed on top is the LDS read of every tap)
      half8 swh = __builtin_bit_cast(half8, As[0 * 64 + lz]), swl = __builtin_bit_cast(half8, As[1 * 64 + lz]);
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          half8 xh = bq[dy][c][0], xl = bq[dy][c][1];
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const int f = c * 9 + dy * 3 + dx;
            half8 nwh = swh, nwl = swl;
            if (f + 1 < NCH * 9) {
              nwh = __builtin_bit_cast(half8, As[((f + 1) * 2 + 0) * 64 + lz]);
              nwl = __builtin_bit_cast(half8, As[((f + 1) * 2 + 1) * 64 + lz]);
            }
            const half8 wh = swh * (_Float16)(1.f / kH3Scale);
            acc = h3_mfma(swh, xh, acc);
            acc = h3_mfma(wh, xl, acc);
            acc = h3_mfma(swl, xh, acc);
            if (dx < 2) {
              xh = shl1(xh);
              xl = shl1(xl);
            }
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);       // no front-loading of all 18 / 36 fragment reads (registers)
#endif
            swh = nwh;
            swl = nwl;
          }
        }
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] *= inv;
    }
    if (i < kTileN) {
      const int nn = p0 + i;
      if (a.y_split) store_tile_split(a, b, 0, nn, h, acc, len_out);
      else store_tile_fast(a, b, 4 * h, nn, acc, len_out);
    }
  }
  if (ovf) report_h3_overflow(a.ovf, a.ovf_layer, a.seq);
}

size_t lds_bytes(int cin) { return (size_t)(cin / 16) * 9 * 2 * 64 * 16; }

template <int CIN, bool XS>
void set_attr() {
  RVCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_thin_kernel<CIN, XS>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(CIN)));
}

template <int CIN, bool XS>
void launch(const ConvArgs& a, dim3 grid, hipStream_t stream) {
  hipLaunchKernelGGL((conv3_thin_kernel<CIN, XS>), grid, dim3(256), lds_bytes(CIN), stream, a);
}

}  // namespace

bool conv3_thin_ok(const ConvArgs& a) {
  static const bool on = !getenv("RVCX_CONV3_THIN") || atoi(getenv("RVCX_CONV3_THIN")) != 0;
  static const int min_n = getenv("RVCX_CONV3_THIN_MIN") ? atoi(getenv("RVCX_CONV3_THIN_MIN")) : 20000;
  if (!on || !a.w_h3 || !conv_h3_enabled() || g_conv_override.tile >= 0 || g_conv_override.splitk > 0) return false;
  if (a.groups != 1 || a.stride != 1 || a.dil != 1 || a.ksize != 9 || a.kw != 3 || a.rowpitch < 3 || a.pad != a.rowpitch + 1)
    return false;
  if (!(a.Cin_g == 16 || a.Cin_g == 32) || a.Cin_gp != a.Cin_g || a.Cout_g > 32 || a.Cout_gp > 32 || a.Cout_g < 1) return false;
  if (a.out_mode != OUT_NORMAL || a.act > ACT_RELU || a.pre_act != ACT_NONE || a.nz_har) return false;
  if (a.y_split && (a.Cout_g % 16 != 0 || a.acc2_mode != ACC2_NONE || a.res)) return false;
  if (!a.y_split && !a.y && a.acc2_mode == ACC2_NONE) return false;
  if ((long)a.Nout < min_n) return false;
  if ((long)a.Cin_g * a.x_cs * 4 >= kH3Oob || (long)a.Cout_g * a.y_cs * 4 >= kH3Oob) return false;
  return true;
}

void launch_conv3_thin(const ConvArgs& a, hipStream_t stream) {
  static std::mutex mu;
  static uint64_t done = 0;
  int dev = 0;
  RVCX_HIP(hipGetDevice(&dev));
  {
    std::lock_guard<std::mutex> g(mu);
    if (!((done >> (dev & 63)) & 1)) {
      set_attr<16, false>();
      set_attr<16, true>();
      set_attr<32, false>();
      set_attr<32, true>();
      done |= 1ull << (dev & 63);
    }
  }
  static const int ncu = [] {
    int d = 0, n = 256;
    if (hipGetDevice(&d) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d);
    return std::max(8, n);
  }();
  static const int per_cu_env = getenv("RVCX_CONV3_THIN_WGS") ? atoi(getenv("RVCX_CONV3_THIN_WGS")) : 0;
  const int per_cu = per_cu_env > 0 ? per_cu_env : (a.Cin_g == 16 ? 6 : 4);   // 24 / 16 waves per CU (79 / 105 VGPRs, 18 / 36 KB of LDS per workgroup)
  const int ntiles = cdiv(a.Nout, kTileN);
  // Every workgroup is resident from the start, and every wave of an XCD's range gets the SAME number of tiles: with a full
  // grid the 106 656-position level gave 483 of 3072 waves a second tile and the launch lasted two tile lives instead of one
  const int per_xcd = cdiv(ntiles, 8), max_wg_x = std::max(1, ncu * per_cu / 8);
  const int rounds = cdiv(per_xcd, 4 * max_wg_x);
  const int wgs = 8 * std::max(1, cdiv(per_xcd, 4 * rounds));
  dim3 grid(wgs, a.B);
  const bool xs = a.x_split != nullptr;
  if (a.Cin_g == 16) {
    if (xs) launch<16, true>(a, grid, stream);
    else launch<16, false>(a, grid, stream);
  } else {
    if (xs) launch<32, true>(a, grid, stream);
    else launch<32, false>(a, grid, stream);
  }
  RVCX_HIP(hipGetLastError());
}

}  // namespace rvcx
