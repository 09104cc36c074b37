// Which CUs does a stream created with hipExtStreamCreateWithCUMask really get?  A kernel of many one-wave workgroups records
// (HW_REG_XCC_ID, HW_REG_HW_ID {se, sh, cu}) and spins ~20 us so that the grid spreads; the host prints, per mask, the number
// of distinct CUs used in every XCD and the workgroups each XCD received.  Workgroups are dealt round-robin to the XCDs
// whatever the mask says, so a mask that leaves the XCDs unequal numbers of CUs makes the narrowest XCD the straggler of
// every launch on that stream (rvcx ran HuBERT on "the first 216 bits" for three rounds).
// Build: hipcc --offload-arch=gfx950 -O3 tools/cu_mask_probe.hip -o /tmp/cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <vector>

#define CK(x)                                                   \
  do {                                                          \
    hipError_t e_ = (x);                                        \
    if (e_ != hipSuccess) {                                     \
      fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); \
      exit(1);                                                  \
    }                                                           \
  } while (0)

__global__ void probe(unsigned* out, int spin) {
  if (threadIdx.x == 0) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    out[blockIdx.x] = ((xcc & 15) << 16) | (hw & 0xff00);     // cu_id 8..11, sh_id 12, se_id 13..15
  }
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {
  }
}

static void run(const char* name, const std::vector<uint32_t>& mask) {
  hipStream_t s;
  if (mask.empty()) {
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  } else if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
    printf("%-44s hipExtStreamCreateWithCUMask failed\n", name);
    (void)hipGetLastError();
    return;
  }
  const int nblk = 8192;
  unsigned* d;
  CK(hipMalloc(&d, nblk * 4));
  CK(hipMemset(d, 0xff, nblk * 4));
  probe<<<nblk, 64, 0, s>>>(d, 2000);     // 20 us per workgroup at 100 MHz
  CK(hipStreamSynchronize(s));
  std::vector<unsigned> h(nblk);
  CK(hipMemcpy(h.data(), d, nblk * 4, hipMemcpyDeviceToHost));
  std::map<int, std::set<unsigned>> cus;
  std::map<int, int> wgs;
  for (unsigned v : h) {
    cus[v >> 16].insert(v & 0xffff);
    wgs[v >> 16]++;
  }
  int total = 0;
  printf("%-44s CUs per XCD:", name);
  for (int x = 0; x < 8; ++x) {
    printf(" %2zu", cus[x].size());
    total += (int)cus[x].size();
  }
  printf("  = %3d   workgroups per XCD:", total);
  for (int x = 0; x < 8; ++x) printf(" %4d", wgs[x]);
  printf("\n");
  CK(hipFree(d));
  CK(hipStreamDestroy(s));
}

static std::vector<uint32_t> first_bits(int n) {
  std::vector<uint32_t> m(8, 0);
  for (int i = 0; i < n; ++i) m[i >> 5] |= 1u << (i & 31);
  return m;
}

int main() {
  run("no mask", {});
  for (int n : {8, 32, 64, 128, 192, 200, 216, 224, 232, 248}) {
    char name[64];
    snprintf(name, sizeof name, "first %d bits", n);
    run(name, first_bits(n));
  }
  {   // 27 of every 32 bits
    std::vector<uint32_t> m(8, 0x07ffffffu);
    run("bits 0..26 of every 32-bit word", m);
  }
  {   // every bit except i % 8 == 7 for i >= 192 ... : 27 per XCD if bit i -> XCD i % 8
    std::vector<uint32_t> m(8, 0);
    for (int i = 0; i < 256; ++i)
      if (i / 8 < 27) m[i >> 5] |= 1u << (i & 31);
    run("bits i with i / 8 < 27 (= first 216)", m);
  }
  return 0;
}
