// Bidirectional GRU recurrence (RMVPE BiGRU(384 -> 2x256), rvc/lib/predictors/RMVPE.py:125-137).
// The input projections W_ih x + b_ih for every frame are one MFMA GEMM done by the conv kernel
// (transposed store, (B,T,6H)); what is left is the serial part h_t = f(gi_t, W_hh h_{t-1}), T = 3232
// dependent steps for a 30 s clip.  Gate order r, z, n (torch.nn.GRU).
//
// Two kernels:
//  * bigru_cluster_kernel: NC workgroups (= CUs) per (direction, batch item).  W_hh (786 KB fp32) does
//    not fit one CU, so each workgroup keeps the 3*H/NC gate rows of its H/NC hidden units in REGISTERS
//    for the whole sequence and the workgroups exchange their slice of h_t every step through
//    data-tagged 8-byte granules ({value, step} written by one write-through store, polled with
//    relaxed agent-scope loads -- no fences, placement independent; guide G16 "R2").  Per-step cost is
//    one 128-FMA register dot product + one inter-CU hand-off (~1.5 us) instead of streaming W_hh from
//    L2 (~7 us).
//  * bigru_kernel: one workgroup per (direction, item), W_hh streamed from L2 every step.  Used when
//    the cluster grid would not be co-resident (large batches) -- the cluster kernel spins.
#include "conv.h"
#include "ops.h"

namespace rvcx {

template <int H>
__global__ __launch_bounds__(3 * H) void bigru_kernel(const float* __restrict__ gi,
                                                      const float* __restrict__ whh_t,  // (2, H, 3H)
                                                      const float* __restrict__ bhh,    // (2, 3H)
                                                      float* __restrict__ y, int T,
                                                      const int* __restrict__ lens) {
  __shared__ float hs[H];
  __shared__ float gh[3 * H];
  const int dir = blockIdx.x, b = blockIdx.y;
  const int j = threadIdx.x;
  const float* W = whh_t + (long)dir * H * 3 * H;
  const float bj = bhh[dir * 3 * H + j];
  const float* gib = gi + (long)b * T * 6 * H + dir * 3 * H;
  float* yb = y + ((long)b * 2 + dir) * H * T;
  const int Tn = lens ? lens[b] : T;      // this item's frames (rows stay T apart)
  if (j < H) hs[j] = 0.f;
  __syncthreads();
  for (int step = 0; step < Tn; ++step) {
    const int t = dir == 0 ? step : Tn - 1 - step;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
#pragma unroll 8
    for (int k = 0; k < H; k += 4) {
      acc0 = fmaf(W[(long)(k + 0) * 3 * H + j], hs[k + 0], acc0);
      acc1 = fmaf(W[(long)(k + 1) * 3 * H + j], hs[k + 1], acc1);
      acc2 = fmaf(W[(long)(k + 2) * 3 * H + j], hs[k + 2], acc2);
      acc3 = fmaf(W[(long)(k + 3) * 3 * H + j], hs[k + 3], acc3);
    }
    gh[j] = ((acc0 + acc1) + (acc2 + acc3)) + bj;
    __syncthreads();
    if (j < H) {
      const float* g = gib + (long)t * 6 * H;
      const float r = 1.f / (1.f + expf(-(g[j] + gh[j])));
      const float z = 1.f / (1.f + expf(-(g[H + j] + gh[H + j])));
      const float n = tanhf(g[2 * H + j] + r * gh[2 * H + j]);
      const float hn = (1.f - z) * n + z * hs[j];
      hs[j] = hn;
      yb[(long)j * T + t] = hn;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------
constexpr int GRU_H = 256;
// Cluster geometry <NC, CPT>: NC workgroups per (direction, item), each owning U = H/NC hidden units = 3U gate rows; a
// thread owns RPT = 4 gate rows x CPT columns of W_hh in registers (h arrives as CPT/4 broadcast 16-byte LDS reads).
//   <4, 32>: 192 rows x 8 column slices = 384 threads, 128 weights each   (default)
//   <8, 16>:  96 rows x 16 column slices = 384 threads, 64 weights each   (RVCX_GRU_NC=8: half the dot-product length)
template <int NC, int CPT>
struct GruGeom {
  static constexpr int U = GRU_H / NC, ROWS = 3 * U, RPT = 4, NCS = GRU_H / CPT, NRG = ROWS / RPT, THREADS = NRG * NCS;
  static_assert(THREADS - U >= GRU_H - U, "one thread per foreign unit in the gather phase");
};
constexpr int GRU_NC = 4;                 // default cluster size (scratch / co-residency bounds use the maximum, 8)
constexpr unsigned GRU_SPIN_LIMIT = 1u << 22;   // ~seconds: a lost partner ends the kernel instead of hanging the GPU

__device__ __forceinline__ float fast_sigmoid(float x) { return __fdividef(1.f, 1.f + __expf(-x)); }
// tanh(x) = sign(x) (1 - e) / (1 + e), e = exp(-2|x|): no cancellation near 0, saturates cleanly
__device__ __forceinline__ float fast_tanh(float x) {
  const float e = __expf(-2.f * fabsf(x));
  return copysignf(__fdividef(1.f - e, 1.f + e), x);
}

// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() also waits vmcnt(0): here that would put
// the write-acknowledge of the h_t / y stores and the next step's prefetched input-projection loads (global round trips
// of ~0.5-1 us) on the serial path of every step.  Global visibility between workgroups is carried by the tagged granules.
__device__ __forceinline__ void lds_barrier() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// Partial sums in LDS: row r (= 4 rg + q) of one column slice lives at [q * NRG + rg], so a thread stores its four rows as
// four ds_write_b32 and the 64 gate lanes read 64 different banks.  The natural layout ([r], one ds_write_b128 per
// thread) is NOT used: with another LDS-heavy kernel (the time-major GEMM) resident on the same CU, gate lanes were seen
// to read the PREVIOUS step's value in one dword q of the 16-byte stores of 16 consecutive lanes -- after s_waitcnt
// lgkmcnt(0) + s_barrier, a few times per ten thousand steps, never without the co-resident kernel (round 3;
// tools/check_gru_under_load.py reproduces it: 4-7 distinct results of 24 with b128, 1 of 24 with b32).
// -DRVCX_GRU_B128=1 rebuilds the round-3 form (natural layout, the four stores merged into one ds_write_b128) so that the
// reproducer stays alive (tools/check_gru_under_load.py); =2: the same with the array declared aligned(16); 3-6: the
// round-4 experiments that narrowed it down.
#ifndef RVCX_GRU_B128
#define RVCX_GRU_B128 0
#endif
template <int NRG>
__device__ __forceinline__ int part_at(int row) { return RVCX_GRU_B128 ? row : (row & 3) * NRG + (row >> 2); }

union Granule {
  unsigned long long u;
  struct {
    float v;
    int tag;
  } s;
};

template <int NC, int CPT>
__global__ __launch_bounds__(384) void bigru_cluster_kernel(const float* __restrict__ gi,
                                                                    const float* __restrict__ whh_t,  // (2,H,3H)
                                                                    const float* __restrict__ bhh,
                                                                    float* __restrict__ y,
                                                                    unsigned long long* xbuf,  // (B,2,2,H) granules
                                                                    int* err, int T, int nq, int colocate,
                                                                    const int* __restrict__ lens) {
  using G = GruGeom<NC, CPT>;
  static_assert(G::THREADS == 384, "launch bounds");
  constexpr int GATHER0 = (G::U + 63) / 64 * 64;     // first polling thread: the wave after the publishing lanes
  static_assert(G::THREADS - GATHER0 >= GRU_H - G::U, "one polling thread per foreign unit");
  constexpr int H = GRU_H, GRU_U = G::U, GRU_ROWS = G::ROWS, GRU_RPT = G::RPT, GRU_CPT = CPT, GRU_NCS = G::NCS,
                GRU_NRG = G::NRG, GRU_THREADS = G::THREADS, GRU_NC = NC;
  __shared__ __attribute__((aligned(16))) float hs[H];
#if RVCX_GRU_B128 == 2
  __shared__ __attribute__((aligned(16))) float part[GRU_NCS][GRU_ROWS];
#else
  __shared__ float part[GRU_NCS][GRU_ROWS];
#endif
  __shared__ int sfail;
  // workgroup -> (cluster q = dir + 2 b, member c).  Workgroups are dealt round-robin to the 8 XCDs, so with
  // `colocate` the NC members of a cluster are the ids congruent mod 8: they share one XCD and its L2.
  int c, q;
  if (colocate) {
    const int g = blockIdx.x, rest = g >> 3;
    c = rest % GRU_NC;
    q = (rest / GRU_NC) * 8 + (g & 7);
  } else {
    c = blockIdx.x % GRU_NC;
    q = blockIdx.x / GRU_NC;
  }
  if (q >= nq) return;
  __builtin_amdgcn_s_setprio(3);   // latency-bound serial chain: issue ahead of co-resident conv waves
  const int dir = q & 1, b = q >> 1;
  const int Tn = lens ? lens[b] : T;      // this item's frames: every workgroup of the cluster takes the same count
  const int tid = threadIdx.x;
  const int rg = tid % GRU_NRG, cs = tid / GRU_NRG;
  const float* W = whh_t + (long)dir * H * 3 * H;
  // this thread's 4 x 32 weights: W_hh[grow(rg*4+q)][cs*32 .. +32)  (whh_t is (H, 3H): column-major rows)
  static_assert(GRU_RPT == 4, "part_at() places four rows per thread");
  float w[GRU_RPT][GRU_CPT];
#pragma unroll
  for (int q = 0; q < GRU_RPT; ++q) {
    const int row = rg * GRU_RPT + q;
    const int gate = row / GRU_U, ul = row % GRU_U;
    const int grow = gate * H + c * GRU_U + ul;          // row of W_hh (3H x H)
#pragma unroll
    for (int k = 0; k < GRU_CPT; ++k) w[q][k] = W[(long)(cs * GRU_CPT + k) * 3 * H + grow];
  }
  const float* gib = gi + (long)b * T * 6 * H + dir * 3 * H;
  float* yb = y + ((long)b * 2 + dir) * H * T;
  unsigned long long* xb = xbuf + ((long)b * 2 + dir) * 2 * H;
  const int ju = c * GRU_U + tid;                      // unit handled in the gate phase (tid < GRU_U)
  float bh_r = 0.f, bh_z = 0.f, bh_n = 0.f;
  if (tid < GRU_U) {
    bh_r = bhh[dir * 3 * H + ju];
    bh_z = bhh[dir * 3 * H + H + ju];
    bh_n = bhh[dir * 3 * H + 2 * H + ju];
  }
  for (int k = tid; k < H; k += GRU_THREADS) hs[k] = 0.f;
  if (tid == 0) sfail = 0;
  __syncthreads();
  // input-projection terms of the next step are fetched one step ahead (they do not depend on h)
  float gr_n = 0.f, gz_n = 0.f, gn_n = 0.f;
  if (tid < GRU_U && Tn > 0) {
    const float* g = gib + (long)(dir == 0 ? 0 : Tn - 1) * 6 * H;
    gr_n = g[ju];
    gz_n = g[H + ju];
    gn_n = g[2 * H + ju];
  }
  for (int step = 0; step < Tn; ++step) {
    const int t = dir == 0 ? step : Tn - 1 - step;
    const float g_r = gr_n, g_z = gz_n, g_n = gn_n;
    if (tid < GRU_U && step + 1 < Tn) {
      const float* g = gib + (long)(dir == 0 ? t + 1 : t - 1) * 6 * H;
      gr_n = g[ju];
      gz_n = g[H + ju];
      gn_n = g[2 * H + ju];
    }
    // ---- partial dot products of this thread's 4 rows over its 32 columns (h broadcast from LDS, 16 B at a time)
    float acc[GRU_RPT] = {0.f, 0.f, 0.f, 0.f};
    const float4* h4 = reinterpret_cast<const float4*>(hs + cs * GRU_CPT);
#pragma unroll
    for (int k = 0; k < GRU_CPT / 4; ++k) {
      const float4 hv = h4[k];
#pragma unroll
      for (int q = 0; q < GRU_RPT; ++q) {
        acc[q] = fmaf(w[q][4 * k + 0], hv.x, acc[q]);
        acc[q] = fmaf(w[q][4 * k + 1], hv.y, acc[q]);
        acc[q] = fmaf(w[q][4 * k + 2], hv.z, acc[q]);
        acc[q] = fmaf(w[q][4 * k + 3], hv.w, acc[q]);
      }
    }
#if RVCX_GRU_B128 == 3          // experiment: natural layout, but four separate ds_write_b32 (the merge into b128 is prevented)
#pragma unroll
    for (int q = 0; q < GRU_RPT; ++q)
      asm volatile("ds_write_b32 %0, %1" ::"v"((unsigned)(size_t)&part[cs][rg * GRU_RPT + q]), "v"(acc[q]) : "memory");
#else
#pragma unroll
    for (int q = 0; q < GRU_RPT; ++q) part[cs][part_at<GRU_NRG>(rg * GRU_RPT + q)] = acc[q];   // four 4-byte stores (see part_at)
#endif
#if RVCX_GRU_B128 == 5          // experiment: ~64 idle cycles between the counter reaching zero and the barrier
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_sleep 1" ::: "memory");
#endif
#if RVCX_GRU_B128 == 6          // experiment: the barrier twice
    lds_barrier();
#endif
    lds_barrier();
    // ---- gates for the owned units, publish h_t[ju] as a {value, step+1} granule
    if (tid < GRU_U) {
      float sr = 0.f, sz = 0.f, sn = 0.f;
#pragma unroll
      for (int p = 0; p < GRU_NCS; ++p) {
#if RVCX_GRU_B128 == 4          // experiment: b128 stores, but every sum read by its own ds_read_b32 (no read2st64 pairs)
        float pr, pz, pn;
        asm volatile("ds_read_b32 %0, %3\n\tds_read_b32 %1, %4\n\tds_read_b32 %2, %5\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(pr), "=&v"(pz), "=&v"(pn)
                     : "v"((unsigned)(size_t)&part[p][tid]), "v"((unsigned)(size_t)&part[p][GRU_U + tid]),
                       "v"((unsigned)(size_t)&part[p][2 * GRU_U + tid])
                     : "memory");
        sr += pr;
        sz += pz;
        sn += pn;
#elif RVCX_GRU_B128 == 7        // experiment: b128 stores, compiler-scheduled reads that cannot be paired (volatile)
        sr += *(volatile float*)&part[p][tid];
        sz += *(volatile float*)&part[p][GRU_U + tid];
        sn += *(volatile float*)&part[p][2 * GRU_U + tid];
#else
        sr += part[p][part_at<GRU_NRG>(tid)];
        sz += part[p][part_at<GRU_NRG>(GRU_U + tid)];
        sn += part[p][part_at<GRU_NRG>(2 * GRU_U + tid)];
#endif
      }
      const float ghr = sr + bh_r, ghz = sz + bh_z, ghn = sn + bh_n;
      // hardware exp2 / rcp (1-2 ulp): the gate chain is on the serial critical path of every step
      const float r = fast_sigmoid(g_r + ghr);
      const float z = fast_sigmoid(g_z + ghz);
      const float n = fast_tanh(g_n + r * ghn);
      const float hn = (1.f - z) * n + z * hs[ju];
      Granule gr;
      gr.s.v = hn;
      gr.s.tag = step + 1;
      __hip_atomic_store(xb + (step & 1) * H + ju, gr.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      yb[(long)ju * T + t] = hn;
      hs[ju] = hn;   // own slice: no other thread reads hs[ju] before the barrier below
    } else if (tid >= GATHER0 && tid < GATHER0 + (H - GRU_U)) {
      // ---- gather the other workgroups' units of h_t (whole waves only: a wave that held both publishing and polling
      // lanes could run its polling branch first and never publish -- every workgroup of the cluster would wait)
      int k = tid - GATHER0;                   // 0 .. H-U-1 over the foreign units
      if (k >= c * GRU_U) k += GRU_U;          // skip the own slice
      Granule gr;
      unsigned spins = 0;
      do {
        gr.u = __hip_atomic_load(xb + (step & 1) * H + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gr.s.tag == step + 1) break;
        if (++spins > GRU_SPIN_LIMIT) {
          sfail = 1;
          break;
        }
      } while (true);
      hs[k] = gr.s.v;
    }
    lds_barrier();
    if (sfail) break;     // block-uniform: every thread sees the flag after the barrier
  }
  if (tid == 0 && sfail) atomicExch(err, 1);
}

size_t bigru_scratch_bytes(int B) { return (size_t)B * 2 * 2 * GRU_H * sizeof(unsigned long long); }

void launch_bigru(const float* gi, const float* whh_t, const float* bhh, float* y, int B, int T, int H,
                  void* scratch, int* err, hipStream_t stream, const int* lens) {
  RVCX_CHECK(H == GRU_H, "bigru: hidden size must be 256 (RMVPE)");
  // the cluster kernel needs all 2*B*NC workgroups co-resident (they spin on each other)
  static const int nc = getenv("RVCX_GRU_NC") ? atoi(getenv("RVCX_GRU_NC")) : GRU_NC;
  static const bool plain = getenv("RVCX_GRU_PLAIN") && atoi(getenv("RVCX_GRU_PLAIN")) != 0;   // debugging: one WG per direction
  if (scratch && err && 2 * B * nc <= 128 && !g_gru_no_cluster && !plain) {
    unsigned long long* xbuf = static_cast<unsigned long long*>(scratch);
    RVCX_HIP(hipMemsetAsync(scratch, 0, bigru_scratch_bytes(B), stream));
    static int colocate = -1;
    if (colocate < 0) colocate = getenv("RVCX_GRU_COLOCATE") ? atoi(getenv("RVCX_GRU_COLOCATE")) : 1;
    const int nq = 2 * B;
    const int grid = colocate ? 8 * nc * cdiv(nq, 8) : nc * nq;
    if (nc == 8)
      hipLaunchKernelGGL((bigru_cluster_kernel<8, 16>), dim3(grid), dim3(GruGeom<8, 16>::THREADS), 0, stream, gi, whh_t, bhh,
                         y, xbuf, err, T, nq, colocate, lens);
    else
      hipLaunchKernelGGL((bigru_cluster_kernel<4, 32>), dim3(grid), dim3(GruGeom<4, 32>::THREADS), 0, stream, gi, whh_t, bhh,
                         y, xbuf, err, T, nq, colocate, lens);
  } else {
    hipLaunchKernelGGL(bigru_kernel<GRU_H>, dim3(2, B), dim3(3 * GRU_H), 0, stream, gi, whh_t, bhh, y, T, lens);
  }
  RVCX_HIP(hipGetLastError());
}

}  // namespace rvcx
