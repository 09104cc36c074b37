// Which HIP streams share a hardware queue?  Streams are multiplexed onto a few HSA queues; two streams on one queue run
// their kernels in submission order however independent they are (rvcx's HuBERT stream, created fifth, shared its queue
// with the F0 model's stream for three rounds: the two "concurrent" branches of the front end excluded each other).
// Probe: streams are created in a fixed order (plain x N, one high-priority stream, one CU-masked stream at a chosen
// position) and used once in that order; then for every pair (i, j) a 2 ms spin kernel goes to stream i and an empty
// kernel to stream j: if the empty kernel finishes only after the spin, the two share a queue.
// Build: hipcc --offload-arch=gfx950 -O3 tools/stream_queue_probe.hip -o /tmp/stream_queue_probe ; usage: [order, e.g. pPppMp]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x)                                                   \
  do {                                                          \
    hipError_t e_ = (x);                                        \
    if (e_ != hipSuccess) {                                     \
      fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); \
      exit(1);                                                  \
    }                                                           \
  } while (0)

__global__ void spin(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {
  }
}
__global__ void nop() {}

int main(int argc, char** argv) {
  // creation order as a string: p = plain, P = high priority, M = CU mask (216 CUs); default = rvcx's order until round 4
  const std::string order = argc > 1 ? argv[1] : "pPppMp";
  std::vector<hipStream_t> st;
  std::vector<std::string> name;
  int lo = 0, hi = 0;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  for (size_t i = 0; i < order.size(); ++i) {
    hipStream_t s;
    if (order[i] == 'P') {
      CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi));
      name.push_back("prio" + std::to_string(i));
    } else if (order[i] == 'M') {
      uint32_t mask[8] = {0};
      for (int b = 0; b < 216; ++b) mask[b >> 5] |= 1u << (b & 31);
      CK(hipExtStreamCreateWithCUMask(&s, 8, mask));
      name.push_back("mask" + std::to_string(i));
    } else {
      CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
      name.push_back("s" + std::to_string(i));
    }
    st.push_back(s);
  }
  for (auto s : st) nop<<<1, 64, 0, s>>>();      // first use in creation order
  CK(hipDeviceSynchronize());
  const int n = (int)st.size();
  printf("streams in creation order:");
  for (auto& nm : name) printf(" %s", nm.c_str());
  printf("\nrow i = spinning stream, column j = stream of the empty kernel; X = j had to wait for i (one queue)\n      ");
  for (int j = 0; j < n; ++j) printf("%6s", name[j].c_str());
  printf("\n");
  for (int i = 0; i < n; ++i) {
    printf("%5s ", name[i].c_str());
    for (int j = 0; j < n; ++j) {
      if (i == j) {
        printf("     .");
        continue;
      }
      CK(hipDeviceSynchronize());
      spin<<<1, 64, 0, st[i]>>>(200000);          // 2 ms at 100 MHz
      const auto t0 = std::chrono::steady_clock::now();
      nop<<<1, 64, 0, st[j]>>>();
      CK(hipStreamSynchronize(st[j]));
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      printf("%6s", ms > 1.0 ? "X" : "-");
    }
    printf("\n");
  }
  CK(hipDeviceSynchronize());
  return 0;
}
