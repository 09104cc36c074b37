// Stand-alone reproducer for the round-3 BiGRU anomaly (DESIGN.md "A 16-byte LDS store that lost a dword"; VERDICT r3
// item 3): the step pattern of bigru_cluster_kernel -- every thread stores its four partial sums to LDS, a raw
// `s_waitcnt lgkmcnt(0); s_barrier`, 64 gate lanes read the sums of all column slices back, a second barrier -- run for
// many steps by a few VICTIM workgroups while an AGGRESSOR kernel (LDS-heavy: 16-byte stores and loads, the shape of the
// time-major GEMM's staging) shares the CUs from a second stream.  Every stored value encodes (step, slice, row): a
// gate lane that reads anything but this step's value is counted, with the dword position inside the 16-byte store.
//
//   MODE 0: one float4 store per thread, natural layout part[cs][4 rg + q]            (the round-3 form that failed)
//   MODE 1: four 4-byte stores, layout part[cs][q NRG + rg]                           (the form that ships)
//   MODE 2: MODE 0 with __attribute__((aligned(16))) on the array
//   MODE 3: MODE 0 with __syncthreads() instead of the raw barrier
//   MODE 4: MODE 0 with the array deliberately placed 8 bytes off a 16-byte boundary  (what a misaligned b128 does)
//
// Build: hipcc --offload-arch=gfx950 -O3 tools/lds_b128_under_load.hip -o /tmp/lds_b128 ; run on the GPU box.
// `llvm-objdump -d` of the code object shows which store each mode really got (ds_write_b128 / ds_write2_b64 / b32).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                   \
  do {                                                          \
    hipError_t e_ = (x);                                        \
    if (e_ != hipSuccess) {                                     \
      fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); \
      exit(1);                                                  \
    }                                                           \
  } while (0)

constexpr int NRG = 48, NCS = 8, ROWS = 192, U = 64, THREADS = NRG * NCS;   // the <4, 32> cluster geometry of gru.hip

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float enc(int step, int cs, int row) { return (float)((step & 0x3fff) * 2048 + cs * 256 + row); }

struct Report {
  unsigned long long errors, stale_prev, other;
  unsigned long long by_q[4];
  int first_step, first_cs, first_row;
  float first_got, first_want;
};

template <int MODE>
__global__ __launch_bounds__(THREADS) void victim(Report* rep, int steps, float* sink) {
  __shared__ __attribute__((aligned(16))) float hs[256];
  // MODE 4: an 8-byte shim ahead of the array (LDS variables are laid out in declaration order of use)
  __shared__ __attribute__((aligned(16))) float shim[MODE == 4 ? 2 : 4];
  __shared__ float part_plain[NCS * ROWS + 4];
  __shared__ __attribute__((aligned(16))) float part_al[NCS * ROWS];
  float* part = MODE == 2 ? part_al : (MODE == 4 ? part_plain + 2 : part_plain);
  const int tid = threadIdx.x, rg = tid % NRG, cs = tid / NRG;
  for (int k = tid; k < 256; k += THREADS) hs[k] = 1.f + 0.001f * k;
  if (tid < 2) shim[tid] = 0.f;
  __syncthreads();
  float w[4][32];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int k = 0; k < 32; ++k) w[q][k] = 1e-3f * ((tid * 131 + q * 17 + k) % 97);
  float keep = 0.f;
  for (int step = 0; step < steps; ++step) {
    // the dot-product phase of the real kernel: broadcast 16-byte reads of h, 128 FMAs
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float4* h4 = reinterpret_cast<const float4*>(hs + cs * 32);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float4 hv = h4[k];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc[q] = fmaf(w[q][4 * k + 0], hv.x, acc[q]);
        acc[q] = fmaf(w[q][4 * k + 1], hv.y, acc[q]);
        acc[q] = fmaf(w[q][4 * k + 2], hv.z, acc[q]);
        acc[q] = fmaf(w[q][4 * k + 3], hv.w, acc[q]);
      }
    }
    keep += acc[0] + acc[1] + acc[2] + acc[3];
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = enc(step, cs, rg * 4 + q) + (acc[q] > 1e30f ? 1.f : 0.f);
    if (MODE == 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) part[cs * ROWS + q * NRG + rg] = v[q];
    } else {
      *reinterpret_cast<float4*>(part + cs * ROWS + rg * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
    if (MODE == 3) __syncthreads();
    else lds_barrier();
    if (tid < U) {
      // all 24 sums first (the compiler pairs them into ds_read2st64_b32, as in the kernel that failed), then the checks
      float gotv[NCS][3];
#pragma unroll
      for (int p = 0; p < NCS; ++p)
#pragma unroll
        for (int gate = 0; gate < 3; ++gate) {
          const int row = gate * U + tid;
          gotv[p][gate] = MODE == 1 ? part[p * ROWS + (row & 3) * NRG + (row >> 2)] : part[p * ROWS + row];
        }
#pragma unroll
      for (int p = 0; p < NCS; ++p)
#pragma unroll
        for (int gate = 0; gate < 3; ++gate) {
          const int row = gate * U + tid;
          const float got = gotv[p][gate];
          const float want = enc(step, p, row);
          if (got != want) {
            const unsigned long long n = atomicAdd(&rep->errors, 1ull);
            if (got == enc(step - 1, p, row)) atomicAdd(&rep->stale_prev, 1ull);
            else atomicAdd(&rep->other, 1ull);
            atomicAdd(&rep->by_q[row & 3], 1ull);
            if (n == 0) {
              rep->first_step = step;
              rep->first_cs = p;
              rep->first_row = row;
              rep->first_got = got;
              rep->first_want = want;
            }
          }
        }
      hs[tid] = 1.f + 1e-6f * (step & 255);       // the gate lanes update their slice of h
    }
    if (MODE == 3) __syncthreads();
    else lds_barrier();
  }
  if (keep == 12345.f) sink[0] = keep;
}

// LDS-heavy neighbour: 16-byte stores and loads over 64 KB, two barriers per round (the staging pattern of gemm_h3)
__global__ __launch_bounds__(256) void aggressor(float* sink, int rounds) {
  extern __shared__ uint4 buf[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint4 v = make_uint4(tid, tid * 3u, tid * 7u, tid * 11u);
  unsigned acc = 0;
  // gemm_h3's staging: A elements linear, B elements at an odd pitch of 129 (16-byte units); fragment reads of 16 bytes
  constexpr int BNP = 129;
  for (int r = 0; r < rounds; ++r) {
    const int bufsel = (r & 1) * 2048;
#pragma unroll
    for (int j = 0; j < 4; ++j) buf[bufsel + tid + 256 * j] = v;                                   // A: 4 x 256 elements
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = tid + 256 * j, l8 = e & 7, p = e >> 3;
      buf[bufsel + 1024 + (l8 * BNP + p) % 1024] = v;                                              // B: odd pitch
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const uint4 t = buf[bufsel + ((j & 3) * 256 + (wave >> 1) * 64 + (lane & 31) + (lane >> 5) * 128 + (j >> 2) * 1024) % 2048];
      acc += t.x ^ t.y ^ t.z ^ t.w;
    }
    v.x += acc;
  }
  if (acc == 0x12345678u) sink[1] = (float)acc;
}

template <int MODE>
void run(const char* name, bool load, int steps, int reps) {
  Report* rep;
  float* sink;
  CK(hipMalloc(&rep, sizeof(Report)));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(rep, 0, sizeof(Report)));
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(aggressor), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  for (int r = 0; r < reps; ++r) {
    if (load) aggressor<<<512, 256, 65536, s2>>>(sink, 12000);          // two 64 KB workgroups per CU, ~tens of ms
    victim<MODE><<<8, THREADS, 0, s1>>>(rep, steps, sink);             // 8 workgroups = one clip's two directions x 4
    CK(hipStreamSynchronize(s1));
    CK(hipStreamSynchronize(s2));
  }
  Report h;
  CK(hipMemcpy(&h, rep, sizeof(Report), hipMemcpyDeviceToHost));
  printf("%-44s load %d: %llu wrong reads in %d x %d steps (previous step's value: %llu, other: %llu; by dword of the 16 B: %llu %llu %llu %llu)",
         name, (int)load, h.errors, reps, steps, h.stale_prev, h.other, h.by_q[0], h.by_q[1], h.by_q[2], h.by_q[3]);
  if (h.errors) printf("  first: step %d slice %d row %d got %.0f want %.0f", h.first_step, h.first_cs, h.first_row, h.first_got, h.first_want);
  printf("\n");
  CK(hipFree(rep));
  CK(hipFree(sink));
  CK(hipStreamDestroy(s1));
  CK(hipStreamDestroy(s2));
}

int main(int argc, char** argv) {
  const int steps = argc > 1 ? atoi(argv[1]) : 3232, reps = argc > 2 ? atoi(argv[2]) : 40;
  for (int load = 0; load < 2; ++load) {
    run<0>("b128 store, natural layout", load, steps, reps);
    run<1>("4 x b32 stores, scattered layout (shipping)", load, steps, reps);
    run<2>("b128 store, aligned(16) array", load, steps, reps);
    run<3>("b128 store, __syncthreads()", load, steps, reps);
    run<4>("b128 store, array 8 bytes off alignment", load, steps, reps);
  }
  return 0;
}
