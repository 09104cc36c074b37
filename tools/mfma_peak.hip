// Practical fp32 MFMA ceiling of the box: register-only v_mfma_f32_32x32x2_f32 loop, no memory traffic.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o build/mfma_peak ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void spin(float* out, int iters, float a0, float b0) {
  f16v acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, int iters) {
  float* out;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  spin<NACC><<<blocks, 256>>>(out, 10, 1.f, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  spin<NACC><<<blocks, 256>>>(out, iters, 1.f, 1.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * iters * NACC * 2.0 * 32 * 32 * 2;
  printf("nacc %d blocks %d iters %d: %.3f ms  %.1f TFLOP/s\n", NACC, blocks, iters, ms, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  // waves per SIMD = blocks / 256 CUs (256-thread blocks = 1 wave per SIMD each)
  for (int wps : {1, 2, 3, 4, 5}) {
    run<1>(256 * wps, 40000);
    run<2>(256 * wps, 20000);
    run<4>(256 * wps, 10000);
  }
  return 0;
}
