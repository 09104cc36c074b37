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

}  // namespace rvcx
