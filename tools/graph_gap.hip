// Does a hipGraph shorten the device-side gap between DEPENDENT small kernels on this pool?  A chain of N launches of a tiny
// kernel (each reads what the previous one wrote): (a) stream launches with the host far ahead, (b) the same chain captured
// once into a graph and replayed.  Wall time per kernel from HIP events around the whole chain.
// build: hipcc --offload-arch=gfx950 -O3 tools/graph_gap.hip -o /tmp/graph_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void link(const float* in, float* out, int iters) {
  float v = in[blockIdx.x * 256 + threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  out[blockIdx.x * 256 + threadIdx.x] = v;
}
int main() {
  float *a, *b;
  CK(hipMalloc(&a, 1 << 22));
  CK(hipMalloc(&b, 1 << 22));
  CK(hipMemset(a, 0, 1 << 22));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int N = 400;
  for (int wgs : {64, 1024}) {
    for (int iters : {50, 2000}) {
      auto chain = [&]() {
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(link, dim3(wgs), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, iters);
      };
      chain();
      CK(hipStreamSynchronize(s));
      float ms_stream = 1e9f, ms_graph = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, s));
        chain();
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < ms_stream) ms_stream = ms;
      }
      hipGraph_t g;
      hipGraphExec_t ge;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      chain();
      CK(hipStreamEndCapture(s, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ge, s));
      CK(hipStreamSynchronize(s));
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, s));
        CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < ms_graph) ms_graph = ms;
      }
      printf("wgs %4d iters %4d: stream %.2f us per kernel, graph %.2f us per kernel\n", wgs, iters, ms_stream * 1e3 / N,
             ms_graph * 1e3 / N);
      CK(hipGraphExecDestroy(ge));
      CK(hipGraphDestroy(g));
    }
  }
  return 0;
}
