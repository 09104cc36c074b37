// Stride-1 implicit-GEMM conv with compile-time tiling -- the kernel that carries the NSF-HiFi-GAN
// decoder, the flow, the TextEncoder, the RMVPE U-Net and every Linear of the hot path.
//
// Differences from the generic kernel in conv.hip: the K-chunk shape (KKT taps x CIC channels), the LDS
// row pitch (BN + HALO) and every loop bound are template constants, so the k-loop is fully unrolled
// with immediate-offset ds_reads feeding back-to-back v_mfma_f32_32x32x2_f32, and the staging code is a
// handful of coalesced loads (weights: exactly A_FLOATS/1024 float4 per thread; input: CIC/4 rows per
// wave, 64 consecutive floats per wave-instruction, zero-filled at the sequence edges).
//
//   for ci0 in Cin step CIC:            stage X[ci0:ci0+CIC][n0+off_min : +BN+halo]      -> Bs[CIC][WROW]
//     for kk0 in ksize step KKT:        stage Wp[kk0:kk0+KKT][ci0:ci0+CIC][co0:co0+BM]   -> As[KKT][CIC][BM]
//       for kkl, cp (unrolled):         A frag As[kkl][2cp+h][wm+i], B frag Bs[2cp+h][wn+i+tap(kk)] -> MFMA
#include <cmath>
#include <cstdlib>

#include "conv.h"
#include "conv_device.h"

// timing ablations (RVCX_CONV_DBG=<mask>) are compiled in only with -DRVCX_ABLATION: even uniform
// branches in the unrolled k-loop cost the small-channel layers ~15 %
#ifdef RVCX_ABLATION
#define RVCX_DBG(a, bit) ((a).dbg & (bit))
#else
#define RVCX_DBG(a, bit) 0
#endif

namespace rvcx {

// Buffer loads through a 128-bit resource descriptor: 32-bit byte offsets and hardware range checking -- an
// offset >= num_records returns 0, so the zero fill of the sequence edges / padded channels / padded taps is
// a select on the OFFSET and every load is unconditional (the compiler turned `valid ? *p : 0` into
// branches that serialised the staging loads one memory round trip after the other).
struct BufRsrc {
#if defined(__HIP_DEVICE_COMPILE__)
  __amdgpu_buffer_rsrc_t r;
#endif
};
constexpr int kBufOob = 0x7ffffff0;
__device__ __forceinline__ BufRsrc make_rsrc(const float* base, int bytes) {
  BufRsrc b;
#if defined(__HIP_DEVICE_COMPILE__)
  b.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000);
#endif
  return b;
}
__device__ __forceinline__ float buf_load1(const BufRsrc& b, int byte_off) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b.r, byte_off, 0, 0));
#else
  return 0.f;
#endif
}
__device__ __forceinline__ float4 buf_load4(const BufRsrc& b, int byte_off) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(b.r, byte_off, 0, 0));
#else
  return make_float4(0.f, 0.f, 0.f, 0.f);
#endif
}
__device__ __forceinline__ void wait_all_memory() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_s_waitcnt(0);
#endif
}

// ---- single-buffered variant (register staging, 2 barriers per stage, small LDS footprint -> more
// resident workgroups): better for the long-sequence NSF layers where 3-5 co-resident blocks already
// hide the staging latency.
// STRIDE = 2 (HuBERT extractor): the input tile is staged at full input resolution and the MFMA B
// fragments read it with a stride-2 LDS pattern (2-way bank conflict, the LDS is far from its limit).
template <int BM, int BN, int WR, int WC, int KKT, int CIC, int HALO, int STRIDE = 1>
__global__ __launch_bounds__(256, 3) void conv_fast_sb_kernel(const ConvArgs a) {
  constexpr int WM = BM / (32 * WR), WN = BN / (32 * WC);
  constexpr int WROW = BN * STRIDE + HALO;
  constexpr int A_FLOATS = KKT * CIC * BM;
  constexpr int NA4 = A_FLOATS / 1024;
  constexpr int NBJ = (WROW + 63) / 64;
  constexpr int NBR = CIC / 4;
  static_assert(WR * WC == 4 && A_FLOATS % 1024 == 0 && CIC % 4 == 0, "bad tile");
  __shared__ float As[A_FLOATS];
  __shared__ float Bs[CIC * WROW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WC, wc = wave % WC;
  const int i = lane & 31, h = lane >> 5;
  const int b = blockIdx.z / a.splitk, ks = blockIdx.z - b * a.splitk;
#ifdef RVCX_ABLATION
  long long tr0 = 0, tr1 = 0, tr2 = 0, cy0 = 0;
  if (a.trace && tid == 0) {
    tr0 = wall_clock64();
    cy0 = clock64();   // s_memtime: shader clock
  }
#endif
  // Optional first-round stagger (RVCX_STAGGER, off by default): delays the k-th co-resident workgroup of a CU
  // by k/occupancy of a block time.  The per-workgroup trace shows the phases of co-resident workgroups are
  // already spread (a CU has no workgroup in its main loop < 1 % of the time), so this buys nothing.
  if (a.stagger > 0) {
    const long lin = blockIdx.x + (long)gridDim.x * (blockIdx.y + (long)gridDim.y * blockIdx.z);
    if (lin < a.stagger_blocks) {
      if (tid == 0) {
        const int slot = __builtin_amdgcn_s_getreg(63492) & 15;   // HW_ID.wave_id: slot of this wave on its SIMD
        const long long t_end = wall_clock64() + (long long)slot * a.stagger;
        while (wall_clock64() < t_end) __builtin_amdgcn_s_sleep(16);
      }
      __syncthreads();
    }
  }
  const int co0 = blockIdx.y * BM;
  const int n0 = blockIdx.x * BN;
  const int len_in = a.lens_in ? a.lens_in[b] : a.Tin;
  const int in_base = n0 * STRIDE + a.off_min;
  const int wuse = BN * STRIDE + a.wrow;          // a.wrow = off_max - off_min for this family
  const int pre_act = a.pre_act;
  const float pre_slope = a.pre_slope;
  const BufRsrc xr = make_rsrc(a.x + (long)b * a.x_bs, a.Cin_g * a.x_cs * 4);
  const BufRsrc wr_ = make_rsrc(a.w, a.ksize * a.Cin_gp * a.Cout_gp * 4);

  f32x16 acc[WM][WN];
#pragma unroll
  for (int m = 0; m < WM; ++m)
#pragma unroll
    for (int n = 0; n < WN; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  // chunk-invariant addressing.  Weights: float4 #j of this thread is row (kkl, cil), columns c4*4..+3.
  // Input: wave w stages rows w, w+4, ...; 64 consecutive positions per wave-instruction.
  int a_kkl[NA4], a_off[NA4];
#pragma unroll
  for (int j = 0; j < NA4; ++j) {
    const int idx = tid + j * 256;
    const int row = idx / (BM / 4), c4 = idx % (BM / 4);
    a_kkl[j] = row / CIC;
    a_off[j] = co0 + c4 * 4 < a.Cout_gp ? ((row % CIC) * a.Cout_gp + co0 + c4 * 4) * 4 : kBufOob;
  }
  int b_off[NBJ];
#pragma unroll
  for (int j = 0; j < NBJ; ++j) {
    const int p = lane + 64 * j, pos = in_base + p;
    b_off[j] = (p < wuse && pos >= 0 && pos < len_in) ? pos * 4 : kBufOob;
  }
  const float* Ap = As + h * BM + wr * (WM * 32) + i;
  const float* Bp = Bs + h * WROW + (wc * (WN * 32) + i) * STRIDE;
  const int nkk = (a.ksize + KKT - 1) / KKT;
  const int nci = a.Cin_gp / CIC;
  const int cb0 = ks * nci / a.splitk, cb1 = (ks + 1) * nci / a.splitk;   // this split's ci chunks
  const int nst = (cb1 - cb0) * nkk;
  const int kstride = a.Cin_gp * a.Cout_gp * 4;    // bytes between taps of the packed weights
  const int xrow = a.x_cs * 4;

  // register-staged software pipeline: the loads of stage st+1 are in flight while stage st computes
  float4 ra[NA4];
  float rb[NBR][NBJ];
  auto fetch_b = [&](int ci0) {
#pragma unroll
    for (int rr = 0; rr < NBR; ++rr) {
      const int ci = ci0 + wave + 4 * rr;
      const int rowb = ci * xrow;   // rows >= Cin_g land beyond num_records and read as 0
#pragma unroll
      for (int j = 0; j < NBJ; ++j) rb[rr][j] = buf_load1(xr, b_off[j] == kBufOob ? kBufOob : rowb + b_off[j]);
    }
  };
  auto fetch_a = [&](int ci0, int kk0) {
    const int wbase = ci0 * a.Cout_gp * 4;
#pragma unroll
    for (int j = 0; j < NA4; ++j) {
      const int kk = kk0 + a_kkl[j];
      ra[j] = buf_load4(wr_, (kk < a.ksize && a_off[j] != kBufOob) ? wbase + kk * kstride + a_off[j] : kBufOob);
    }
  };
  auto commit = [&](int kk0) {
    if (kk0 == 0) {
#pragma unroll
      for (int rr = 0; rr < NBR; ++rr) {
        const int r = wave + 4 * rr;
#pragma unroll
        for (int j = 0; j < NBJ; ++j) {
          const int p = lane + 64 * j;
          float v = rb[rr][j];
          if (pre_act == ACT_LRELU) v = v > 0.f ? v : v * pre_slope;
          if (NBJ * 64 == WROW || p < WROW) Bs[r * WROW + p] = v;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NA4; ++j) *reinterpret_cast<float4*>(As + (tid + j * 256) * 4) = ra[j];
  };

  // The weight tile (L2-resident) is fetched one stage ahead; the input tile (streamed from HBM, the long
  // latency under load) as soon as its registers are free, i.e. at the first tap-stage of the previous channel
  // chunk -- issued after the weights so that the next commit's vmcnt wait does not include it.
  int ci0 = cb0 * CIC, kk0 = 0;
  if (nst > 0) {
    fetch_b(ci0);
    fetch_a(ci0, 0);
  }
  for (int st = 0; st < nst; ++st) {
    if (!RVCX_DBG(a, 32) || st == 0) {
      __syncthreads();            // every wave is done reading the previous stage
      commit(kk0);
      __syncthreads();
    }
    int kk1 = kk0 + KKT, ci1 = ci0;
    if (kk1 >= a.ksize) {
      kk1 = 0;
      ci1 += CIC;
    }
    if (!RVCX_DBG(a, 16)) {
      if (st + 1 < nst) fetch_a(ci1, kk1);
      if (kk0 == 0 && ci0 + CIC < cb1 * CIC) fetch_b(ci0 + CIC);
    }
#ifdef RVCX_ABLATION
    if (a.trace && tid == 0 && st == 0) tr1 = wall_clock64();
#endif
#pragma unroll
    for (int kkl = 0; kkl < KKT; ++kkl) {
      const int kk = kk0 + kkl;
      if ((KKT == 1 || kk < a.ksize) && !RVCX_DBG(a, 4)) {
        const int tp = (kk / a.kw) * a.rowpitch + (kk % a.kw) * a.dil - a.pad - a.off_min;
        const float* Bt = Bp + tp;
#pragma unroll
        for (int cp = 0; cp < CIC / 2; ++cp) {
          float av[WM], bv[WN];
#pragma unroll
          for (int m = 0; m < WM; ++m) av[m] = Ap[(kkl * CIC + 2 * cp) * BM + m * 32];
#pragma unroll
          for (int n = 0; n < WN; ++n) bv[n] = Bt[2 * cp * WROW + n * 32 * STRIDE];
#pragma unroll
          for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int n = 0; n < WN; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[n], acc[m][n], 0, 0, 0);
        }
      }
    }
    kk0 = kk1;
    ci0 = ci1;
  }

  const int len_out = a.lens_out ? a.lens_out[b] : 0x7fffffff;
  const int co_w = co0 + wr * (WM * 32) + 4 * h, nn_w = n0 + wc * (WN * 32) + i;
#ifdef RVCX_ABLATION
  if (a.trace && tid == 0) tr2 = wall_clock64();
#endif
  if (RVCX_DBG(a, 8)) return;
  if (a.splitk > 1) {
    // raw partial sums -> part[ks][b][co][nn]; conv_splitk_finish_kernel reduces and applies the epilogue
    float* pb = a.part + ((long)ks * a.B + b) * a.Cout_g * a.Nout;
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
      for (int n = 0; n < WN; ++n) {
        const int nn = nn_w + n * 32;
        if (nn < a.Nout) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = co_w + m * 32 + (r & 3) + 8 * (r >> 2);
            if (co < a.Cout_g) pb[(long)co * a.Nout + nn] = acc[m][n][r];
          }
        }
      }
    return;
  }
  if (fast_epilogue_ok(a)) {
    store_tile_fast(a, b, co_w, nn_w, acc[0][0], len_out);
    if constexpr (WN > 1) store_tile_fast(a, b, co_w, nn_w + 32, acc[0][1], len_out);
    if constexpr (WM > 1) {
      store_tile_fast(a, b, co_w + 32, nn_w, acc[1][0], len_out);
      if constexpr (WN > 1) store_tile_fast(a, b, co_w + 32, nn_w + 32, acc[1][1], len_out);
    }
#ifdef RVCX_ABLATION
    if (a.trace && tid == 0) {
      __builtin_amdgcn_s_waitcnt(0);
      const long lin = blockIdx.x + (long)gridDim.x * (blockIdx.y + (long)gridDim.y * blockIdx.z);
      long long* tp = a.trace + lin * 6;
      tp[0] = __builtin_amdgcn_s_getreg(63492);               // HW_REG_HW_ID
      tp[1] = (long long)(__builtin_amdgcn_s_getreg((20) | (31 << 11)) & 15) | ((clock64() - cy0) << 8);   // XCC_ID | shader cycles of the block
      tp[2] = tr0;
      tp[3] = tr1;
      tp[4] = tr2;
      tp[5] = wall_clock64();
    }
#endif
  } else if (a.out_mode == OUT_SHUF1D) {
    store_tile_shuf1d(a, b, co_w, nn_w, acc[0][0], len_out);
    if constexpr (WN > 1) store_tile_shuf1d(a, b, co_w, nn_w + 32, acc[0][1], len_out);
    if constexpr (WM > 1) {
      store_tile_shuf1d(a, b, co_w + 32, nn_w, acc[1][0], len_out);
      if constexpr (WN > 1) store_tile_shuf1d(a, b, co_w + 32, nn_w + 32, acc[1][1], len_out);
    }
  } else {
    store_tile(a, b, 0, co_w, nn_w, acc[0][0], len_out);
    if constexpr (WN > 1) store_tile(a, b, 0, co_w, nn_w + 32, acc[0][1], len_out);
    if constexpr (WM > 1) {
      store_tile(a, b, 0, co_w + 32, nn_w, acc[1][0], len_out);
      if constexpr (WN > 1) store_tile(a, b, 0, co_w + 32, nn_w + 32, acc[1][1], len_out);
    }
  }
}

// sum the split-K partial slabs in a fixed order (deterministic) and run the fused epilogue.
// Round 6: four consecutive positions per thread (16-byte loads) with the slabs' loads of eight segments in flight before
// their adds (one element per thread and a load -> add chain per segment read 20 MB at 2.2 TB/s: 9.4 us of every split
// deep-level conv of a single clip); the order of the adds -- 0 + p0 + p1 + ... -- is unchanged: same bits.
__global__ __launch_bounds__(256) void conv_splitk_finish_kernel(const ConvArgs a) {
  const long per = (long)a.Cout_g * a.Nout, total = per * a.B;
  const int S = a.splitk;
  if ((per & 3) == 0) {                                         // four consecutive elements are one aligned 16-byte word of a slab
    const long slab = per * a.B;                                // floats between two segments' values of one element
    for (long idx = (blockIdx.x * 256L + threadIdx.x) * 4; idx < total; idx += (long)gridDim.x * 1024) {
      const int b = idx / per;
      const long r = idx - (long)b * per;
      int co = r / a.Nout, nn = r - (long)co * a.Nout;
      const float* p0 = a.part + (long)b * per + r;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      for (int s0 = 0; s0 < S; s0 += 8) {
        float4 q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          q[j] = s0 + j < S ? *reinterpret_cast<const float4*>(p0 + (long)(s0 + j) * slab) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (s0 + j < S) {
            v[0] += q[j].x;
            v[1] += q[j].y;
            v[2] += q[j].z;
            v[3] += q[j].w;
          }
      }
      const int len_out = a.lens_out ? a.lens_out[b] : 0x7fffffff;
#pragma unroll
      for (int e = 0; e < 4; ++e) {                            // (a row of a 606-position map ends inside a word)
        store_elem(a, b, co, nn, v[e], len_out);
        if (++nn == a.Nout) {
          nn = 0;
          ++co;
        }
      }
    }
    return;
  }
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int b = idx / per;
    const long r = idx - (long)b * per;
    const int co = r / a.Nout, nn = r - (long)co * a.Nout;
    float v = 0.f;
    for (int s = 0; s < S; ++s) v += a.part[((long)s * a.B + b) * per + r];
    const int len_out = a.lens_out ? a.lens_out[b] : 0x7fffffff;
    store_elem(a, b, co, nn, v, len_out);
  }
}

// Cin = 1 convolutions (NSF noise convs k = 2*stride, HuBERT conv0 k10 s5; round 5: the RMVPE U-Net's first 3 x 3 conv on
// the row-padded mel map, taps (kk / kw) rowpitch + (kk % kw) dil): an FIR per output channel, bound by the output write --
// plain vector FMAs, 16 output channels per thread, weights broadcast from LDS.
constexpr int kCin1MaxK = 192;
__global__ __launch_bounds__(256) void conv_cin1_kernel(const ConvArgs a) {
  __shared__ float ws[kCin1MaxK * 16];
  const int co0 = blockIdx.y * 16, b = blockIdx.z;
  const int K = a.ksize;
  for (int idx = threadIdx.x; idx < K * 16; idx += 256) {
    const int kk = idx >> 4, j = idx & 15;
    ws[idx] = co0 + j < a.Cout_gp ? a.w[(long)kk * a.Cin_gp * a.Cout_gp + co0 + j] : 0.f;
  }
  __syncthreads();
  const int nn = blockIdx.x * 256 + threadIdx.x;
  if (nn >= a.Nout) return;
  const int len_in = a.lens_in ? a.lens_in[b] : a.Tin;
  const int len_out = a.lens_out ? a.lens_out[b] : 0x7fffffff;
  const float* xb = a.x + (long)b * a.x_bs;
  const int p0 = nn * a.stride - a.pad;
  const int kw = a.kw, rowpitch = a.rowpitch;
  float acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  const int last = max(len_in - 1, 0);
#pragma unroll 4
  for (int kk = 0; kk < K; ++kk) {
    const int pos = kw == K ? p0 + kk * a.dil : p0 + (kk / kw) * rowpitch + (kk % kw) * a.dil;
    float xv = xb[min(max(pos, 0), last)];                 // always in range: no branch around the load
    xv = (pos >= 0 && pos < len_in) ? xv : 0.f;
    if (a.pre_act == ACT_LRELU) xv = xv > 0.f ? xv : xv * a.pre_slope;
    const float4* w4 = reinterpret_cast<const float4*>(ws + kk * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 w = w4[q];
      acc[4 * q + 0] = fmaf(w.x, xv, acc[4 * q + 0]);
      acc[4 * q + 1] = fmaf(w.y, xv, acc[4 * q + 1]);
      acc[4 * q + 2] = fmaf(w.z, xv, acc[4 * q + 2]);
      acc[4 * q + 3] = fmaf(w.w, xv, acc[4 * q + 3]);
    }
  }
#pragma unroll
  for (int j = 0; j < 16; ++j)
    if (co0 + j < a.Cout_g) store_elem(a, b, co0 + j, nn, acc[j], len_out);
}

// Cout = 1 convolutions (NSF conv_post: 32 -> 1, k = 7 over 1.5 M positions per clip; rvc/lib/algorithm/nsf.py:142-144).
// On a 32 x N MFMA tile 31 of 32 output rows are padding and the launch ran at 1 TB/s (198 us); the layer is a
// Cin x k tap FIR bound by reading its input once: 196 MB at HBM speed.  A thread owns 4 consecutive outputs, per input
// channel it loads the three aligned 16-byte words that cover its taps (neighbouring threads share two of them through
// the vector L1), applies the pre-activation once per value and runs k x 4 FMAs; the weights are wave-uniform.
// Exact fp32 (FMA chain, fixed order: channels ascending, taps ascending) -- nothing to split, no range guard needed.
template <int K>
__global__ __launch_bounds__(256) void conv_cout1_kernel(const ConvArgs a) {
  constexpr int PADL = 4;                       // aligned words: positions t0 - 4 .. t0 + 7 cover taps t0 - pad .. t0 + 3 + K-1-pad
  __shared__ float ws[64 * K];
  const int b = blockIdx.z, Cin = a.Cin_g;
  for (int idx = threadIdx.x; idx < Cin * K; idx += 256) {
    const int ci = idx / K, kk = idx - ci * K;
    ws[idx] = a.w[((long)kk * a.Cin_gp + ci) * a.Cout_gp];
  }
  __syncthreads();
  const long t0 = (blockIdx.x * 256L + threadIdx.x) * 4;
  if (t0 >= a.Nout) return;
  const int len_in = a.lens_in ? a.lens_in[b] : a.Tin;
  const int len_out = a.lens_out ? a.lens_out[b] : 0x7fffffff;
  const float* xb = a.x + (long)b * a.x_bs;
  const float slope = a.pre_act == ACT_LRELU ? a.pre_slope : 1.f;      // pre_act in {none, lrelu}
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const bool interior = t0 >= PADL && t0 + 8 <= len_in;
  for (int ci = 0; ci < Cin; ++ci) {
    const float* xc = xb + (long)ci * a.x_cs;
    float v[12];
    if (interior) {
      const float4 q0 = *reinterpret_cast<const float4*>(xc + t0 - 4);
      const float4 q1 = *reinterpret_cast<const float4*>(xc + t0);
      const float4 q2 = *reinterpret_cast<const float4*>(xc + t0 + 4);
      v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w;
      v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
      v[8] = q2.x; v[9] = q2.y; v[10] = q2.z; v[11] = q2.w;
    } else {
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        const long pos = t0 - 4 + j;
        v[j] = (pos >= 0 && pos < len_in) ? xc[pos] : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < 12; ++j) v[j] = fmaxf(v[j], v[j] * slope);          // leaky_relu for 0 <= slope <= 1 (1: identity)
#pragma unroll
    for (int kk = 0; kk < K; ++kk) {
      const float w = ws[ci * K + kk];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = fmaf(w, v[PADL + q + kk - (K - 1) / 2], acc[q]);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (t0 + q < a.Nout) store_elem(a, b, 0, (int)(t0 + q), acc[q], len_out);
}

static bool conv_cout1_ok(const ConvArgs& a) {
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  return a.Cout_g == 1 && a.groups == 1 && a.stride == 1 && a.dil == 1 && a.kw == a.ksize && a.ksize == 7 && a.pad == 3 &&
         a.Cin_g >= 8 && a.Cin_g <= 64 && a.out_mode == OUT_NORMAL && !a.x_split && !a.y_split && a.acc2_mode == ACC2_NONE &&
         (a.pre_act == ACT_NONE || (a.pre_act == ACT_LRELU && a.pre_slope >= 0.f && a.pre_slope <= 1.f)) && a.x_cs % 4 == 0 &&
         a.x_bs % 4 == 0 && al(a.x) && a.Nout == a.Tin;
}

namespace {

struct FastCfg {
  int bm, bn, halo, cic;   // cic = 32 marks the k=1 (Linear) variants; halo < 0: stride-2 variant (halo = -halo)
  float eff;
  void (*kern)(const ConvArgs);      // long-N launches
};

const FastCfg kFast[] = {
    // 1-D convs (halo <= 64 covers k=11 d=5)
    {128, 128, 64, 16, 1.00f, conv_fast_sb_kernel<128, 128, 2, 2, 2, 16, 64>},
    {128, 64, 64, 16, 0.85f, conv_fast_sb_kernel<128, 64, 4, 1, 2, 16, 64>},
    {64, 256, 64, 16, 1.00f, conv_fast_sb_kernel<64, 256, 1, 4, 4, 16, 64>},
    {64, 128, 64, 16, 0.85f, conv_fast_sb_kernel<64, 128, 2, 2, 4, 16, 64>},
    {64, 64, 64, 16, 0.70f, conv_fast_sb_kernel<64, 64, 2, 2, 4, 16, 64>},
    {32, 256, 64, 16, 0.90f, conv_fast_sb_kernel<32, 256, 1, 4, 4, 16, 64>},
    // 3x3 convs on row-padded maps (halo = 2*Wp + 2 <= 262)
    {128, 128, 320, 16, 1.00f, conv_fast_sb_kernel<128, 128, 2, 2, 2, 16, 320>},
    {64, 128, 320, 16, 0.85f, conv_fast_sb_kernel<64, 128, 2, 2, 4, 16, 320>},
    {32, 128, 320, 16, 0.70f, conv_fast_sb_kernel<32, 128, 1, 4, 4, 16, 320>},
    // Linear layers (k = 1)
    {128, 128, 0, 32, 1.00f, conv_fast_sb_kernel<128, 128, 2, 2, 1, 32, 0>},
    {128, 64, 0, 32, 0.85f, conv_fast_sb_kernel<128, 64, 4, 1, 1, 32, 0>},
    {64, 64, 0, 32, 0.70f, conv_fast_sb_kernel<64, 64, 2, 2, 1, 32, 0>},
    // stride-2 1-D convs (HuBERT feature extractor k=3 / k=2)
    {128, 128, -64, 16, 1.00f, conv_fast_sb_kernel<128, 128, 2, 2, 2, 16, 64, 2>},
    {128, 64, -64, 16, 0.85f, conv_fast_sb_kernel<128, 64, 4, 1, 2, 16, 64, 2>},
    {64, 64, -64, 16, 0.70f, conv_fast_sb_kernel<64, 64, 2, 2, 4, 16, 64, 2>},
};
constexpr int kNumFast = sizeof(kFast) / sizeof(kFast[0]);

}  // namespace

ConvOverride g_conv_override;

namespace {
int g_occ[32] = {0};          // resident workgroups per CU of each sb kernel
float g_stagger_scale = 0.f;  // RVCX_STAGGER=<scale> enables the first-round stagger (measured: no gain, off)
}  // namespace

void conv_fast_init() {
  for (int t = 0; t < kNumFast; ++t) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(kFast[t].kern), 256, 0) != hipSuccess)
      n = 3;
    g_occ[t] = std::max(1, n);
  }
  if (const char* e = getenv("RVCX_STAGGER")) g_stagger_scale = (float)atof(e);
  if (const char* e = getenv("RVCX_CONV_TILE")) g_conv_override.tile = atoi(e);   // tuning runs only
}

void conv_fast_describe(ConvProfile* p) {
  p->bm[7] = 16;
  p->bn[7] = 256;
  p->halo[7] = 300000;   // slot 7: conv_cin1_kernel
  p->bm[60] = 1;
  p->bn[60] = 1024;
  p->halo[60] = 300001;   // slot 60: conv_cout1_kernel
  p->bm[kConvtThinSlot] = 128;
  p->bn[kConvtThinSlot] = 32;
  p->halo[kConvtThinSlot] = 300002;   // convt_thin_kernel
  p->bm[kConv3ThinSlot] = 32;
  p->bn[kConv3ThinSlot] = 30;
  p->halo[kConv3ThinSlot] = 300004;   // conv3_thin_kernel
  p->bm[kConvDeepSlot] = 64;
  p->bn[kConvDeepSlot] = 320;
  p->halo[kConvDeepSlot] = 300003;    // conv_ws_kernel (conv_deep.hip)
  for (int t = 0; t < kNumFast; ++t) {
    const int s = 8 + t;
    p->bm[s] = kFast[t].bm;
    p->bn[s] = kFast[t].bn;
    p->halo[s] = kFast[t].cic == 32 ? 100000 : (kFast[t].halo < 0 ? 200000 : kFast[t].halo * 10);
  }
}

void launch_splitk_finish(const ConvArgs& a, hipStream_t stream) {
  const long total = (long)a.B * a.Cout_g * a.Nout;
  const long work = (((long)a.Cout_g * a.Nout) & 3) == 0 ? cdiv64(total, 4) : total;   // four elements per thread when aligned
  hipLaunchKernelGGL(conv_splitk_finish_kernel, dim3((unsigned)std::min<long>(cdiv64(work, 256), 4096)), dim3(256), 0,
                     stream, a);
}

int launch_conv_fast(ConvArgs& a, hipStream_t stream) {
  if (a.Cin_g == 1 && a.groups == 1 && a.ksize <= kCin1MaxK && a.kw >= 1 && a.ksize % a.kw == 0) {
    hipLaunchKernelGGL(conv_cin1_kernel, dim3(cdiv(a.Nout, 256), cdiv(a.Cout_g, 16), a.B), dim3(256), 0, stream, a);
    RVCX_HIP(hipGetLastError());
    return 7;
  }
  if (convt_thin_ok(a)) {
    launch_convt_thin(a, stream);
    return kConvtThinSlot;
  }
  RVCX_CHECK(!a.nz_har || a.out_mode == OUT_SHUF1D, "conv: a fused noise conv needs a polyphase ConvTranspose1d launch");
  if (conv_cout1_ok(a)) {
    static const bool on = !getenv("RVCX_COUT1") || atoi(getenv("RVCX_COUT1")) != 0;
    if (on) {
      hipLaunchKernelGGL(conv_cout1_kernel<7>, dim3((unsigned)cdiv64(cdiv64(a.Nout, 4), 256), 1, a.B), dim3(256), 0, stream, a);
      RVCX_HIP(hipGetLastError());
      return 60;
    }
  }
  if ((a.stride != 1 && !(a.stride == 2 && a.kw == a.ksize)) || a.Cin_gp % 16 != 0) return -1;
  if (a.groups != 1 && !a.w_h3) return -1;      // grouped layers: only the split-fp16 tiles take them (blockIdx.z = group)
  // the staging loads address x (per batch item) and the packed weights through 32-bit buffer offsets
  if ((long)a.Cin_gp * a.x_cs * 4 >= kBufOob || (long)a.ksize * a.Cin_gp * a.Cout_gp * 4 >= kBufOob) return -1;
  int off_min = 1 << 30, off_max = -(1 << 30);
  for (int kk = 0; kk < a.ksize; ++kk) {
    const int o = conv_tap_off(a, kk);
    off_min = std::min(off_min, o);
    off_max = std::max(off_max, o);
  }
  const int halo = off_max - off_min;
  if (conv3_thin_ok(a)) {
    launch_conv3_thin(a, stream);
    return kConv3ThinSlot;
  }
  if (conv_deep_ok(a)) {
    launch_conv_deep(a, stream);
    return kConvDeepSlot;
  }
  {
    const int slot = launch_conv_h3(a, halo, off_min, stream);
    if (slot >= 0) return slot;
  }
  if (a.groups != 1) return -1;
  // joint (tile, split-K) choice from a small cost model fitted on MI355X (round 2: tools/sweep_conv.py, since removed; refits and checks:
  // tools/sweep_tiles_1d.py, tools/sweep_unet.py, tools/bench_conv.py -> profiles/sweep_tiles_1d_r04.txt):
  //   block time = 2 bm bn (K/S + ovh_t) / (577 GFLOP/s x eff_t x f(c)),  c = blocks S / 256 blocks per CU,
  // f(c) = MFMA utilisation of a CU with c co-resident blocks (0.45 alone .. 1.0 from four up), ovh_t = the
  // prologue + epilogue of a block in K-steps, the busiest CU runs ceil(c) blocks; split-K adds the finish
  // pass (S + 1 slabs through HBM at ~3 TB/s + one launch).
  struct TileFit {
    float eff, ovh;
  };
  static const TileFit kFit[15] = {{0.964f, 201.f}, {0.922f, 129.f}, {0.968f, 231.f}, {0.954f, 150.f}, {0.964f, 169.f},
                                   {1.00f, 184.f},  {0.93f, 260.f},  {0.95f, 180.f},  {1.00f, 150.f},  {1.00f, 213.f},
                                   {0.95f, 150.f},  {0.95f, 120.f},  {0.964f, 201.f}, {0.922f, 129.f}, {0.964f, 169.f}};
  const bool lin = a.ksize == 1 && a.Cin_gp % 32 == 0;
  const double kdepth = (double)a.ksize * a.Cin_gp;
  // batch invariance (see launch_conv_h3): split-K is decided for one batch item, the tile for the real batch
  const long cap_item = a.part_cap_item > 0 ? a.part_cap_item : a.part_cap;
  auto select = [&](int Bsel, int forcedS, int* tile_out, int* s_out) {
    int best = -1, S = 1;
    double best_t = 1e300;
    for (int t = 0; t < kNumFast; ++t) {
      const FastCfg& F = kFast[t];
      if ((F.halo < 0) != (a.stride == 2)) continue;
      if (a.stride == 2) {
        if (halo > -F.halo) continue;
      } else {
        if ((F.cic == 32) != lin) continue;   // Linear layers go to the k=1 variants
        if (halo > F.halo) continue;
        if (F.halo == 320 && halo <= 64) continue;
      }
      if (g_conv_override.tile >= 0 && g_conv_override.tile != t) continue;
      const long blocks = (long)cdiv(a.Cout_gp, F.bm) * cdiv(a.Nout, F.bn) * Bsel;
      const int nci = a.Cin_gp / F.cic;
      for (int s = 1; s <= 8; s *= 2) {
        if (s > 1 && (!a.part || nci / s < 1 || kdepth / s < 128.0 || (long)s * a.Cout_g * a.Nout > cap_item ||
                      (long)s * a.B * a.Cout_g * a.Nout > a.part_cap))
          break;
        if (g_conv_override.splitk > 0 && g_conv_override.splitk != s) continue;
        if (forcedS > 0 && s != forcedS) continue;
        const double c = (double)blocks * s / 256.0;
        const double f = std::min(1.0, 0.45 + 0.55 * (std::max(c, 1.0) - 1.0) / 3.0);
        double us = std::ceil(c) * 2.0 * F.bm * F.bn * (kdepth / s + kFit[t].ovh) / (577e3 * kFit[t].eff * f);
        if (s > 1) us += 3.0 + (double)(s + 1) * Bsel * a.Cout_g * a.Nout * 4.0 / 3e6;
        if (us < best_t) {
          best_t = us;
          best = t;
          S = s;
        }
      }
    }
    *tile_out = best;
    *s_out = S;
  };
  int best = -1, S = 1;
  select(1, 0, &best, &S);
  if (best >= 0 && a.B > 1) {
    int b2 = -1, s2 = 1;
    select(a.B, S, &b2, &s2);
    if (b2 >= 0 && s2 == S) best = b2;
  }
  if (best < 0) return -1;
  const FastCfg& F = kFast[best];
  a.off_min = off_min;
  a.wrow = halo;
  const long blocks = (long)cdiv(a.Nout, F.bn) * cdiv(a.Cout_gp, F.bm) * a.B;
  a.splitk = S;
  {
    static int dbg = -1;
    if (dbg < 0) dbg = getenv("RVCX_CONV_DBG") ? atoi(getenv("RVCX_CONV_DBG")) : 0;
    a.dbg = dbg;
  }
  dim3 grid(cdiv(a.Nout, F.bn), cdiv(a.Cout_gp, F.bm), a.B * S);
  {
    // spacing = one block's exclusive time on its CU (block wall time / occupancy), only worth it when the
    // launch runs for several rounds
    const int occ = g_occ[best];
    const double excl_us = 2.0 * F.bm * F.bn * (kdepth / S + kFit[best].ovh) / (577e3 * kFit[best].eff);
    const bool multi_round = blocks * S >= 2L * 256 * occ;
    a.stagger = (multi_round && g_stagger_scale > 0.f) ? (int)(excl_us * 100.0 * g_stagger_scale) : 0;
    a.stagger_blocks = 256 * occ;
  }
  hipLaunchKernelGGL(F.kern, grid, dim3(256), 0, stream, a);
  if (S > 1) launch_splitk_finish(a, stream);
  RVCX_HIP(hipGetLastError());
  return 8 + best;
}

}  // namespace rvcx
