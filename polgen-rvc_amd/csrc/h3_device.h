// Device-side primitives shared by the fp16 hi/lo split ("h3") kernels: conv_h3.hip and resblock.hip.
#pragma once
#include "conv.h"
#include "conv_device.h"

namespace rvcx {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
constexpr float kH3Scale = 256.f;

struct H3Rsrc {
#if defined(__HIP_DEVICE_COMPILE__)
  __amdgpu_buffer_rsrc_t r;
#endif
};
__device__ __forceinline__ H3Rsrc h3_rsrc(const void* base, int bytes) {
  H3Rsrc b;
#if defined(__HIP_DEVICE_COMPILE__)
  b.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
#endif
  return b;
}
__device__ __forceinline__ float h3_load1(const H3Rsrc& b, int off) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b.r, off, 0, 0));
#else
  return 0.f;
#endif
}
__device__ __forceinline__ uint4 h3_load4(const H3Rsrc& b, int off) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(b.r, off, 0, 0));
#else
  return make_uint4(0, 0, 0, 0);
#endif
}
// per-lane offset + wave-uniform offset (an SGPR: no vector address arithmetic per load)
__device__ __forceinline__ uint4 h3_load4s(const H3Rsrc& b, int voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(b.r, voff, soff, 0));
#else
  return make_uint4(0, 0, 0, 0);
#endif
}
__device__ __forceinline__ f32x16 h3_mfma(half8 a, half8 b, f32x16 c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#else
  return c;
#endif
}
constexpr int kH3Oob = 0x7ffffff0;

// c1 -> c2 hand-off of a ResBlock: the activated outputs are stored already split (see ConvArgs::y_split).  A lane owns
// rows (r&3) + 8(r>>2) + 4 hl of a 32x32 tile: register group g = r>>2 is the 4 channels q = 4 hl .. 4 hl + 3 of
// element (chunk = (c0 + 8 g)/16, h = g & 1) -- one 8-byte store for the hi halves and one for the scaled lo halves.
__device__ __forceinline__ void store_tile_split(const ConvArgs& a, int b, int c0, int nn, int hl, const f32x16& t,
                                                 int len_out) {
  if (nn >= a.Nout) return;
  bool live = nn < len_out;
  if (a.zero_wp > 0) {                               // pad columns of a row-padded 2-D map stay zero
    const int col = nn % a.zero_wp;
    live = live && col != 0 && col != a.zero_wp - 1;
  }
  char* base = static_cast<char*>(a.y_split) + (long)b * a.y_bs * 4;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int cg = c0 + 8 * g;                       // first channel of the 8-channel element this group belongs to
    if (cg + 4 * hl < a.Cout_g) {
      typedef _Float16 half4 __attribute__((ext_vector_type(4)));
      half4 hi, lo;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v = t[4 * g + q] + (a.bias ? a.bias[cg + 4 * hl + q] : 0.f);
        v = apply_act(v, a.act, a.act_slope);
        v = live ? v : 0.f;
        if (!(fabsf(v) < kH3ActLimit)) report_h3_overflow(a.ovf, a.ovf_layer, a.seq);   // never taken on sane data
        const _Float16 vh = (_Float16)v;
        hi[q] = vh;
        lo[q] = (_Float16)((v - (float)vh) * kH3Scale);
      }
      const long e_hi = ((long)((cg >> 4) * 2 + 0) * 2 + (g & 1)) * a.y_cs + nn;
      const long e_lo = ((long)((cg >> 4) * 2 + 1) * 2 + (g & 1)) * a.y_cs + nn;
      *reinterpret_cast<half4*>(base + e_hi * 16 + 8 * hl) = hi;
      *reinterpret_cast<half4*>(base + e_lo * 16 + 8 * hl) = lo;
    }
  }
}

}  // namespace rvcx
