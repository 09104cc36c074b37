// Back-to-back small kernels on one stream (and on two streams): per-kernel wall time unprofiled vs under rocprofv3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void work(float* p, int iters) {
  float v = p[threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  p[blockIdx.x * 256 + threadIdx.x] = v;
}
int main() {
  float* d;
  hipMalloc(&d, 1 << 24);
  hipStream_t s1, s2;
  hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  for (int iters : {200, 2000, 8000}) {
    for (int two = 0; two < 2; ++two) {
      const int N = 1000;
      work<<<256, 256, 0, s1>>>(d, iters);
      hipDeviceSynchronize();
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < N; ++i) {
        work<<<256, 256, 0, s1>>>(d, iters);
        if (two) work<<<256, 256, 0, s2>>>(d + (1 << 20), iters);
      }
      hipDeviceSynchronize();
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      printf("iters %5d streams %d: %.2f us per kernel slot\n", iters, two + 1, us / N);
    }
  }
  return 0;
}
