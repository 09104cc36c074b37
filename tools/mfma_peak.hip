// Practical matrix-core ceilings of the box: what the MFMA pipes SUSTAIN on real data.
//
//   f32   : v_mfma_f32_32x32x2_f32, register-only (the exact-fp32 mode's ceiling; round 2's measurement)
//   f16 R : v_mfma_f32_32x32x16_f16 on RANDOM fp16 operands held in registers -- four A and four B fragments rotate, so
//           consecutive MFMAs see different operands (data toggling = the power a real GEMM draws)
//   f16 L : the same MFMAs fed from LDS, one ds_read_b128 per MFMA (the diet of the fused ResBlock step: six fragment
//           reads for six MFMAs, resblock.hip)
//   f16 L2: both operands from LDS, two ds_read_b128 per MFMA (is the LDS 128 or 256 B/clk/CU for 16-byte reads?)
//   h3    : the k-step of the split-fp16 kernels (two accumulators x three dependent MFMAs, six LDS reads per six MFMAs),
//           the dependent MFMAs back to back or interleaved
//   f16 Z : all-zero operands (no toggling: the power floor; what the all-zero-input experiment of round 2 measured)
// at 1, 2 and 3 waves per SIMD.  Per launch: wall time (HIP events) -> TFLOP/s, and the shader clock the chip held
// (s_memtime cycles / s_memrealtime 100 MHz ticks, read by wave 0 of every workgroup).  The split-fp16 kernels form
// one fp32-grade product block from THREE of these MFMAs: their ceiling is a third of the f16 figure.
//
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o build/mfma_peak
// Run on the GPU box (also under `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- build/mfma_peak`);
// prints a table and one JSON line (profiles/mfma_peak_r04.json).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_));               \
      exit(1);                                                                \
    }                                                                         \
  } while (0)

__device__ __forceinline__ unsigned lcg(unsigned& s) {
  s = s * 1664525u + 1013904223u;
  return s;
}
// fp16 value in about [-2, 2) with a random 10-bit mantissa (normal numbers: every operand bit toggles)
__device__ __forceinline__ _Float16 rnd_half(unsigned& s) {
  const unsigned r = lcg(s) >> 8;
  return (_Float16)(((float)(r & 0xffff) - 32768.f) * (1.f / 16384.f));
}

struct Clocks {
  unsigned long long cyc, ticks;
};

// MODE 0: registers, random; 1: LDS-fed (one read per MFMA), random; 2: registers, zeros; 3: LDS-fed, TWO reads per MFMA
template <int MODE, int NACC>
__global__ __launch_bounds__(256) void spin_f16(float* out, Clocks* clk, int iters) {
  __shared__ half8 lds[4 * 2 * 64 * 4];            // [wave][a|b][lane][4 fragments]: 16 B elements, conflict-free per wave
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned seed = 1234567u + 977u * (blockIdx.x * 256 + tid);
  half8 a[4], b[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      a[q][e] = MODE == 2 ? (_Float16)0.f : rnd_half(seed);
      b[q][e] = MODE == 2 ? (_Float16)0.f : rnd_half(seed);
    }
  if (MODE == 1 || MODE == 3) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      lds[((wave * 2 + 0) * 4 + q) * 64 + lane] = a[q];
      lds[((wave * 2 + 1) * 4 + q) * 64 + lane] = b[q];
    }
    __syncthreads();
  }
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int it0 = 0; it0 < iters; it0 += 4) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {               // unrolled: every operand index below is a compile-time constant
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        half8 av, bv;
        if (MODE == 3) {                            // both operands from LDS: what the LDS can feed at most
          av = lds[((wave * 2 + 0) * 4 + ((i + it) & 3)) * 64 + lane];
          bv = lds[((wave * 2 + 1) * 4 + ((i >> 1) & 3)) * 64 + lane];
          asm volatile("" ::: "memory");
        } else if (MODE == 1) {
          // one 16-byte LDS read per MFMA: A and B fragments alternate (the other operand stays in its register)
          if (i & 1) {
            av = lds[((wave * 2 + 0) * 4 + ((i >> 1) & 3)) * 64 + lane];
            bv = b[(i + it) & 3];
          } else {
            av = a[(i + it) & 3];
            bv = lds[((wave * 2 + 1) * 4 + ((i >> 1) & 3)) * 64 + lane];
          }
          asm volatile("" ::: "memory");            // keep the LDS reads inside the loop, one per MFMA
        } else {
          av = a[(i + it) & 3];
          bv = b[((i >> 2) + it) & 3];
        }
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[i], 0, 0, 0);
      }
    }
  }
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) {
    clk[blockIdx.x].cyc = c1 - c0;
    clk[blockIdx.x].ticks = w1 - w0;
  }
}

// The accumulation pattern of the split-fp16 kernels: TWO accumulators per wave, THREE dependent MFMAs on each per k-step
// (hh, hl, lh), fed from LDS with 6 reads per 6 MFMAs.  DEP 1: the three MFMAs of an accumulator back to back (what the
// compiler emits); DEP 0: the two accumulators interleaved (a0 a1 a0 a1 a0 a1).
template <int DEP>
__global__ __launch_bounds__(256) void spin_h3(float* out, Clocks* clk, int iters) {
  __shared__ half8 lds[4 * 6 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned seed = 99u + 977u * (blockIdx.x * 256 + tid);
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    half8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = rnd_half(seed);
    lds[(wave * 6 + q) * 64 + lane] = v;
  }
  __syncthreads();
  f32x16 acc0, acc1;
#pragma unroll
  for (int j = 0; j < 16; ++j) acc0[j] = acc1[j] = 0.f;
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    const half8 a0 = lds[(wave * 6 + 0) * 64 + lane], a1 = lds[(wave * 6 + 1) * 64 + lane];
    const half8 b00 = lds[(wave * 6 + 2) * 64 + lane], b01 = lds[(wave * 6 + 3) * 64 + lane];
    const half8 b10 = lds[(wave * 6 + 4) * 64 + lane], b11 = lds[(wave * 6 + 5) * 64 + lane];
    asm volatile("" ::: "memory");
    const half8 wh = a0 * (_Float16)(1.f / 256.f);
    if (DEP) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b00, acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, b10, acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b00, acc0, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b01, acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, b11, acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b01, acc1, 0, 0, 0);
    } else {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b00, acc0, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b01, acc1, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, b10, acc0, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, b11, acc1, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b00, acc0, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b01, acc1, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) s += acc0[j] + acc1[j];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) {
    clk[blockIdx.x].cyc = c1 - c0;
    clk[blockIdx.x].ticks = w1 - w0;
  }
}

template <int NACC>
__global__ __launch_bounds__(256) void spin_f32(float* out, Clocks* clk, int iters) {
  unsigned seed = 7654321u + 31u * (blockIdx.x * 256 + threadIdx.x);
  float a[4], b[4];
  for (int q = 0; q < 4; ++q) {
    a[q] = (float)rnd_half(seed) + 1e-3f * (float)rnd_half(seed);
    b[q] = (float)rnd_half(seed) + 1e-3f * (float)rnd_half(seed);
  }
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int it0 = 0; it0 < iters; it0 += 4) {
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
      for (int i = 0; i < NACC; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(i + it) & 3], b[((i >> 2) + it) & 3], acc[i], 0, 0, 0);
  }
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    clk[blockIdx.x].cyc = c1 - c0;
    clk[blockIdx.x].ticks = w1 - w0;
  }
}

struct Result {
  std::string name;
  int wps;
  double ms, tflops, ghz;
};
std::vector<Result> g_results;

template <typename K>
void run(const char* name, K kern, int wps, int iters, int nacc, double flops_per_mfma) {
  const int blocks = 256 * wps;                   // a 256-thread workgroup = one wave per SIMD of its CU
  float* out;
  Clocks* clk;
  CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  CK(hipMalloc(&clk, (size_t)blocks * sizeof(Clocks)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  kern<<<blocks, 256>>>(out, clk, iters / 20 + 1);         // warm-up (clock ramp)
  CK(hipDeviceSynchronize());
  // ~40 ms of sustained work: long enough for the power management to settle on the kernel's steady clock
  CK(hipEventRecord(e0));
  kern<<<blocks, 256>>>(out, clk, iters);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<Clocks> h(blocks);
  CK(hipMemcpy(h.data(), clk, (size_t)blocks * sizeof(Clocks), hipMemcpyDeviceToHost));
  double cyc = 0, ticks = 0;
  for (auto& c : h) {
    cyc += (double)c.cyc;
    ticks += (double)c.ticks;
  }
  const double ghz = cyc / (ticks * 10.0);        // wall_clock64: 100 MHz
  const double flops = (double)blocks * 4 * (double)iters * nacc * flops_per_mfma;
  const double tf = flops / (ms * 1e-3) / 1e12;
  printf("%-26s waves/SIMD %d  %8.3f ms  %8.1f TFLOP/s  shader clock %.2f GHz\n", name, wps, ms, tf, ghz);
  g_results.push_back({name, wps, (double)ms, tf, ghz});
  CK(hipFree(out));
  CK(hipFree(clk));
}

int main() {
  const double F16 = 2.0 * 32 * 32 * 16, F32 = 2.0 * 32 * 32 * 2;
  for (int wps : {1, 2, 3}) {
    const int it16 = 600000 / wps, it32 = 300000 / wps;
    run("f16 R (registers, random)", spin_f16<0, 8>, wps, it16, 8, F16);
    run("f16 L (1 ds_read_b128/MFMA)", spin_f16<1, 8>, wps, it16, 8, F16);
    run("f16 L2 (2 ds_read_b128/MFMA)", spin_f16<3, 8>, wps, it16 / 2, 8, F16);
    run("f16 Z (registers, zeros)", spin_f16<2, 8>, wps, it16, 8, F16);
    run("f32   (registers, random)", spin_f32<8>, wps, it32, 8, F32);
    run("h3 step, 3 dependent MFMAs back to back", spin_h3<1>, wps, it16, 6, F16);
    run("h3 step, two accumulators interleaved", spin_h3<0>, wps, it16, 6, F16);
  }
  printf("{\"tool\": \"tools/mfma_peak.hip\", \"results\": [");
  for (size_t i = 0; i < g_results.size(); ++i) {
    const auto& r = g_results[i];
    printf("%s{\"kernel\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"tflops\": %.1f, \"shader_ghz\": %.3f}", i ? ", " : "",
           r.name.c_str(), r.wps, r.ms, r.tflops, r.ghz);
  }
  printf("]}\n");
  return 0;
}
