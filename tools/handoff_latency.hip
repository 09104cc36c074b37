// Hop latency of an 8-byte {value, tag} granule between two workgroups, by store / load flavour and placement -- the
// hand-off the BiGRU cluster step waits for (gru.hip; ~900 of a step's ~2500 cycles with sc1 stores + sc1 loads).
//   stores: plain | sc0 | sc1 | sc0 sc1 | nt          loads: sc1 | sc0 sc1 | nt | atomic-or-0 (executes at L2)
//   placement: blocks {0, 8} (same XCD under the observed round-robin), {0, 1} (neighbouring XCDs)
// Ping-pong: A stores tag i, B polls until it sees tag i and answers on a second granule, A polls for the answer;
// cycles per hop = round trip / 2.  A bounded poll (never seen -> "STALE") keeps a wrong flavour from hanging the GPU.
// Each workgroup reports its XCC id (s_getreg HW_REG_XCC_ID) so that "same XCD" is observed, not assumed.
// Build: hipcc --offload-arch=gfx950 -O3 tools/handoff_latency.hip -o /tmp/handoff ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>

#define CK(x)                                                   \
  do {                                                          \
    hipError_t e_ = (x);                                        \
    if (e_ != hipSuccess) {                                     \
      fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); \
      exit(1);                                                  \
    }                                                           \
  } while (0)

typedef unsigned long long u64;

template <int S>
__device__ __forceinline__ void st(u64* p, u64 v) {
  if (S == 0) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
  if (S == 1) asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
  if (S == 2) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  if (S == 3) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  if (S == 4) asm volatile("global_store_dwordx2 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
}
template <int L>
__device__ __forceinline__ u64 ld(u64* p) {
  u64 v;
  if (L == 0) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (L == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (L == 2) asm volatile("global_load_dwordx2 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (L == 3) {
    u64 z = 0;
    asm volatile("global_atomic_or_x2 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(z) : "memory");
  }
  return v;
}

struct Out {
  long long cycles;
  int stale, xcc_a, xcc_b;
};

template <int S, int L>
__global__ void pingpong(u64* box, Out* out, int peer, int rounds) {
  if (threadIdx.x != 0) return;
  const bool a = blockIdx.x == 0, b = (int)blockIdx.x == peer;
  if (!a && !b) return;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (a) out->xcc_a = xcc & 7;
  else out->xcc_b = xcc & 7;
  u64* ping = box;        // A -> B   (separate 128-byte lines)
  u64* pong = box + 16;   // B -> A
  const long long t0 = wall_clock64();
  const long long c0 = __builtin_readcyclecounter();
  int stale = 0;
  for (int i = 1; i <= rounds && !stale; ++i) {
    if (a) st<S>(ping, (u64)i);
    u64* src = a ? pong : ping;
    int spins = 0;
    while (ld<L>(src) != (u64)i)
      if (++spins > 200000) {
        stale = 1;
        break;
      }
    if (b) st<S>(pong, (u64)i);
  }
  if (a) {
    out->cycles = __builtin_readcyclecounter() - c0;
    out->stale = stale;
    out[1].cycles = wall_clock64() - t0;     // 100 MHz
  }
}

template <int S, int L>
void run(const char* sname, const char* lname, int peer) {
  u64* box;
  Out* out;
  CK(hipMalloc(&box, 4096));
  CK(hipMalloc(&out, 2 * sizeof(Out)));
  const int rounds = 20000;
  double best = 1e30, best_ns = 0;
  Out h[2];
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(box, 0, 4096));
    CK(hipMemset(out, 0, 2 * sizeof(Out)));
    pingpong<S, L><<<peer + 1, 64>>>(box, out, peer, rounds);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost));
    if (h[0].stale) break;
    const double c = (double)h[0].cycles / rounds / 2;
    if (c < best) best = c, best_ns = (double)h[1].cycles * 10.0 / rounds / 2;
  }
  if (h[0].stale) printf("  store %-8s load %-10s blocks {0,%d} xcc {%d,%d}: STALE (never seen)\n", sname, lname, peer, h[0].xcc_a, h[0].xcc_b);
  else printf("  store %-8s load %-10s blocks {0,%d} xcc {%d,%d}: %7.0f cycles  %6.0f ns per hop\n", sname, lname, peer, h[0].xcc_a, h[0].xcc_b, best, best_ns);
  CK(hipFree(box));
  CK(hipFree(out));
}

template <int L>
void loads(const char* lname, int peer) {
  run<0, L>("plain", lname, peer);
  run<1, L>("sc0", lname, peer);
  run<2, L>("sc1", lname, peer);
  run<3, L>("sc0 sc1", lname, peer);
  run<4, L>("nt", lname, peer);
}

// ---- the BiGRU exchange without the arithmetic: 4 workgroups on one XCD (blocks 0, 8, 16, 24), 256 threads each; every step
// wave 0 publishes the workgroup's 64 granules (one plain 8-byte store per lane), waves 1..3 poll the 192 foreign ones
// (sc1), write them to LDS, barrier.  MODE bit 0: two polls in flight per lane (as gru.hip) instead of one; bit 1: a first
// barrier + 12 LDS reads before the publish (the partial-sum phase); bit 2: s_sleep 1 between polls; bit 3: the pollers are
// 96 lanes reading 16 bytes (two granules) each.
template <int MODE>
__global__ __launch_bounds__(256) void exchange(u64* xbuf, long long* out, int steps) {
  __shared__ float hs[256];
  __shared__ float part[4][3][64];
  __shared__ int sfail;
  if (blockIdx.x & 7) return;
  const int c = blockIdx.x >> 3, tid = threadIdx.x;
  if (tid == 0) sfail = 0;
  hs[tid] = 0.f;
  __syncthreads();
  int kf = tid - 64;
  if (kf >= c * 64) kf += 64;
  const long long c0 = __builtin_readcyclecounter();
  float keep = 0.f;
  for (int step = 0; step < steps; ++step) {
    if (MODE & 2) {
      part[tid >> 6][0][tid & 63] = hs[tid];
      part[tid >> 6][1][tid & 63] = hs[tid] + 1.f;
      part[tid >> 6][2][tid & 63] = hs[tid] + 2.f;
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    u64* buf = xbuf + (step & 1) * 256;
    if (tid < 64) {
      float v = 1.f;
      if (MODE & 2)
        for (int p = 0; p < 4; ++p) v += part[p][0][tid] + part[p][1][tid] + part[p][2][tid];
      const u64 g = ((u64)(unsigned)(step + 1) << 32) | __float_as_uint(v * 1e-9f);
      st<0>(buf + c * 64 + tid, g);
      hs[c * 64 + tid] = v * 1e-9f;
    } else if ((MODE & 8) ? tid < 64 + 96 : true) {
      unsigned spins = 0;
      if (MODE & 8) {
        const int k2 = (tid - 64) * 2, kk = k2 >= c * 64 ? k2 + 64 : k2;
        uint4 g;
        do {
          asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(g) : "v"(buf + kk) : "memory");
          if (MODE & 4) __builtin_amdgcn_s_sleep(1);
        } while ((g.y != (unsigned)(step + 1) || g.w != (unsigned)(step + 1)) && ++spins < 400000);
        hs[kk] = __uint_as_float(g.x);
        hs[kk + 1] = __uint_as_float(g.z);
      } else if (MODE & 1) {
        u64 ga, gb;
        asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(ga) : "v"(buf + kf) : "memory");
        __builtin_amdgcn_s_sleep(3);
        asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(gb) : "v"(buf + kf) : "memory");
        u64 v;
        while (true) {
          asm volatile("s_waitcnt vmcnt(1)" : "+v"(ga)::"memory");
          if ((unsigned)(ga >> 32) == (unsigned)(step + 1)) { v = ga; break; }
          asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(ga) : "v"(buf + kf) : "memory");
          asm volatile("s_waitcnt vmcnt(1)" : "+v"(gb)::"memory");
          if ((unsigned)(gb >> 32) == (unsigned)(step + 1)) { v = gb; break; }
          asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(gb) : "v"(buf + kf) : "memory");
          if (++spins > 400000) { sfail = 1; v = 0; break; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        hs[kf] = __uint_as_float((unsigned)v);
      } else {
        u64 g;
        do {
          g = ld<0>(buf + kf);
          if (MODE & 4) __builtin_amdgcn_s_sleep(1);
        } while ((unsigned)(g >> 32) != (unsigned)(step + 1) && ++spins < 400000);
        hs[kf] = __uint_as_float((unsigned)g);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (sfail) break;
    keep += hs[(tid * 7) & 255];
  }
  if (tid == 0 && c == 0) out[0] = __builtin_readcyclecounter() - c0;
  if (keep == 123.f) out[1] = 1;
}

template <int MODE>
void run_exchange(const char* name) {
  u64* xbuf;
  long long* out;
  CK(hipMalloc(&xbuf, 2 * 256 * 8));
  CK(hipMalloc(&out, 16));
  const int steps = 20000;
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(xbuf, 0, 2 * 256 * 8));
    exchange<MODE><<<32, 256>>>(xbuf, out, steps);
    CK(hipDeviceSynchronize());
    long long h;
    CK(hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost));
    best = std::min(best, (double)h / steps);
  }
  printf("  exchange, %-72s %6.0f cycles per step\n", name, best);
  CK(hipFree(xbuf));
  CK(hipFree(out));
}

int main(int argc, char** argv) {
  if (argc > 1) {
    printf("4 workgroups on one XCD, 64 granules published / 192 polled per workgroup and step:\n");
    run_exchange<0>("one poll in flight");
    run_exchange<1>("two polls in flight");
    run_exchange<4>("one poll in flight, s_sleep 1 between polls");
    run_exchange<8>("96 lanes poll 16 bytes");
    run_exchange<12>("96 lanes poll 16 bytes, s_sleep 1 between polls");
    run_exchange<2>("one poll in flight + partial-sum phase (barrier, 12 LDS reads)");
    run_exchange<3>("two polls in flight + partial-sum phase");
    return 0;
  }

  for (int peer : {8, 1}) {
    printf("%s\n", peer == 8 ? "same XCD (blocks 0 and 8):" : "neighbouring XCDs (blocks 0 and 1):");
    loads<0>("sc1", peer);
    loads<1>("sc0 sc1", peer);
    loads<2>("nt", peer);
    loads<3>("atomic-or", peer);
  }
  return 0;
}
