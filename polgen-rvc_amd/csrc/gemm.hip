// GEMM-shaped kernels for the Linear layers of the transformer sections (HuBERT q/k/v, out-proj, fc1, fc2 -- call site
// rvc/infer/pipeline.py:228-236; the north star's "HuBERT transformer QKV/FFN MFMA GEMMs").
//
//     Y[r][co] = act( sum_k W[co][k] X[r][k] + bias[co] ) + R[r][co]         r = b T + t : batch folded into the rows
//
// Activations are TIME-MAJOR here: a row is one frame, its channels are contiguous -- LayerNorm becomes one wavefront
// per row (ops.hip: layernorm_tm), and the B operand of the matrix cores (8 consecutive k per lane) is ONE 16-byte
// load.  Between layers the activations travel already in the fp16 hi/lo split form the MFMAs consume
// ("XS": row r = [chunk of 16 channels][op {hi, S lo}][h][8 halves], 64 bytes per chunk, the same bytes as fp32): the
// producer's epilogue splits (and range-checks, conv.h: kH3ActLimit), the consumer stages with plain 16-byte copies.
//
// gemm_h3_kernel: fp32-grade products from three fp16 MFMAs (conv_h3.hip has the arithmetic), M = Cout from the layer's
// existing fp16 hi/lo weight image, N = rows.  Tile BM x BN, 4 waves as 2 x 2, a stage = 2 chunks (32 channels) of both
// operands, LDS double-buffered, ONE barrier per stage: while the MFMAs of stage s run, the registers fetched for stage
// s + 1 are committed to the other buffer and the loads of stage s + 2 are issued.  No split-K: the k-order is
// fixed, so every tile gives the same bits and a batched call equals its single runs.
// gemm_f32_kernel: the exact-fp32 form (v_mfma_f32_32x32x2_f32) for layers pinned after an fp16-range overflow.
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <utility>

#include "conv.h"
#include "conv_device.h"
#include "gemm.h"
#include "h3_device.h"

namespace rvcx {

namespace {

constexpr int kKC = 2;                 // chunks of 16 channels per stage
constexpr int kBnMax = 128;

// One 32 x 32 accumulator tile -> the requested outputs.  Lane (j = lane & 31, hl = lane >> 5) owns row r0 + j and the
// channels c0 + 8 g + 4 hl + q (g, q = 0..3): four runs of 4 consecutive channels = one 16-byte store per run (fp32
// time-major) or one 8-byte hi + one 8-byte lo store (split form).
// four consecutive channels c .. c + 3 of one row: bias, activation, residual, length mask, the requested outputs
__device__ __forceinline__ void gemm_store4(const GemmArgs& a, int c, long row, bool live, float (&v)[4], bool& ovf) {
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] += (a.bias ? a.bias[c + q] : 0.f);
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = apply_act(v[q], a.act, 0.f);
  if (a.res) {
    const float4 r = *reinterpret_cast<const float4*>(a.res + row * a.ld_res + c);
    v[0] += r.x;
    v[1] += r.y;
    v[2] += r.z;
    v[3] += r.w;
  }
  if (!live) v[0] = v[1] = v[2] = v[3] = 0.f;
  if (a.y) *reinterpret_cast<float4*>(a.y + row * a.ld_y + c) = make_float4(v[0], v[1], v[2], v[3]);
  if (a.y_cf) {                                    // channel-first (B, cout, T): what the attention kernels read
    float* yc = a.y_cf + (row / a.T) * a.cf_bs + (long)c * a.T + row % a.T;
#pragma unroll
    for (int q = 0; q < 4; ++q) yc[(long)q * a.T] = v[q];
  }
  if (a.ys) {
    typedef _Float16 half4 __attribute__((ext_vector_type(4)));
    half4 hi, lo;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      ovf |= !(fabsf(v[q]) < kH3ActLimit);
      const _Float16 vh = (_Float16)v[q];
      hi[q] = vh;
      lo[q] = (_Float16)((v[q] - (float)vh) * kH3Scale);
    }
    char* e = static_cast<char*>(a.ys) + row * a.ld_ys + (c >> 4) * 64 + ((c >> 3) & 1) * 16 + (c & 7) * 2;
    *reinterpret_cast<half4*>(e) = hi;
    *reinterpret_cast<half4*>(e + 32) = lo;
  }
}

__device__ __forceinline__ void gemm_store_tile(const GemmArgs& a, int c0, long row, int hl, const f32x16& t, bool& ovf) {
  if (row >= a.rows) return;
  const bool live = !a.lens || (int)(row % a.T) < a.lens[row / a.T];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int c = c0 + 8 * g + 4 * hl;
    if (c >= a.cout) continue;                       // cout is a multiple of 4 (checked by the launcher)
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = t[4 * g + q];
    gemm_store4(a, c, row, live, v, ovf);
  }
}

// K-split launches (NSEG workgroups per tile, blockIdx.z = segment): the raw segment sums, part[seg][row][cout]
__device__ __forceinline__ void gemm_store_part(const GemmArgs& a, int seg, int c0, long row, int hl, const f32x16& t) {
  if (row >= a.rows) return;
  float* pb = a.part + ((long)seg * a.rows + row) * a.cout;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int c = c0 + 8 * g + 4 * hl;
    if (c < a.cout) *reinterpret_cast<float4*>(pb + c) = make_float4(t[4 * g], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3]);
  }
}

// NSEG: the K range is summed in NSEG equal segments, each from zero, and the segment sums are added left to right --
// the CANONICAL order of a long-K layer (launch_gemm: K >= 2048), whatever the tile and whether the segments are computed
// by one workgroup (SPLIT = false: a second accumulator set) or by NSEG workgroups (SPLIT = true: blockIdx.z = segment,
// raw sums to GemmArgs::part, gemm_finish_kernel adds them in the same order): a batched call, which never splits, still
// equals its single runs bit for bit.
template <int BM, int BN, int NSEG = 1, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void gemm_h3_kernel(const GemmArgs a) {
  constexpr int WM = BM / 64, WN = BN / 64;            // 4 waves as 2 x 2
  constexpr int BNP = BN + 1;                           // odd pitch (in 16-byte elements): conflict-free commits
  constexpr int A_ST = kKC * 4 * BM, B_ST = kKC * 4 * BNP;
  constexpr int NA = kKC * 4 * BM / 256, NB = kKC * 4 * BN / 256;
  static_assert(WM >= 1 && WN >= 1 && NA * 256 == kKC * 4 * BM && NB * 256 == kKC * 4 * BN, "bad tile");
  extern __shared__ uint4 gemm_lds[];
  uint4(*As)[A_ST] = reinterpret_cast<uint4(*)[A_ST]>(gemm_lds);                 // [buf][cl][op*2 + h][co]
  uint4(*Bs)[B_ST] = reinterpret_cast<uint4(*)[B_ST]>(gemm_lds + 2 * A_ST);      // [buf][cl][op*2 + h][p]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int i = lane & 31, h = lane >> 5;
  const int co0 = blockIdx.y * BM;
  const long n0 = (long)blockIdx.x * BN;
  const int nchunk = a.cin_p / 16;
  const int nst_all = (nchunk + kKC - 1) / kKC;
  const int seg_len = nst_all / NSEG;                    // stages per segment (the launcher guarantees an even count)
  const int st0 = SPLIT ? (int)blockIdx.z * seg_len : 0; // this workgroup's first stage
  const int nst = SPLIT ? seg_len : nst_all;
  const H3Rsrc wres = h3_rsrc(a.w_h3, nchunk * 4 * a.cout_p * 16);
  // rows of this tile as one buffer: an offset past the last valid byte reads zeros (rows beyond `rows`, chunks
  // beyond cin_p)
  const long rows_here = a.rows - n0 < BN ? a.rows - n0 : BN;
  const H3Rsrc xres = h3_rsrc(static_cast<const char*>(a.xs) + n0 * a.ld_xs, (int)(rows_here * a.ld_xs));
  const int slab = 4 * a.cout_p * 16;

  int a_off[NA], a_cl[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int e = tid + 256 * j;                        // (cl, oph, co)
    const int co = e % BM, rest = e / BM;
    a_cl[j] = rest >> 2;
    a_off[j] = co0 + co < a.cout_p ? ((rest & 3) * a.cout_p + co0 + co) * 16 : kH3Oob;
  }
  int b_off[NB], b_dst[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int e = tid + 256 * j;                        // (p, cl, oph): 8 consecutive lanes = 128 contiguous bytes of a row
    const int l8 = e & (4 * kKC - 1), p = e / (4 * kKC);
    b_off[j] = p < rows_here ? p * a.ld_xs + l8 * 16 : kH3Oob;
    b_dst[j] = l8 * BNP + p;
  }
  const int row_bytes = a.cin_p * 4;                    // bytes of a row that hold data

  // Two register sets: stage s travels in set s & 1.  Loads stay in flight across the barriers (a raw s_barrier that
  // waits for this wave's LDS traffic only -- __syncthreads() would also drain vmcnt and serialise every stage behind
  // its own load latency): a stage is fetched two iterations before it is committed to LDS.
  uint4 ra[2][NA], rb[2][NB];
  auto fetch = [&](int st, uint4 (&qa)[NA], uint4 (&qb)[NB]) {
    const int c0 = (st0 + st) * kKC;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int ch = c0 + a_cl[j];
      qa[j] = h3_load4(wres, (a_off[j] != kH3Oob && ch < nchunk) ? ch * slab + a_off[j] : kH3Oob);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int col = c0 * 64 + ((tid + 256 * j) & (4 * kKC - 1)) * 16;
      qb[j] = h3_load4(xres, (b_off[j] != kH3Oob && col < row_bytes) ? b_off[j] + c0 * 64 : kH3Oob);
    }
  };
  auto commit = [&](int buf, const uint4 (&qa)[NA], const uint4 (&qb)[NB]) {
#pragma unroll
    for (int j = 0; j < NA; ++j) As[buf][tid + 256 * j] = qa[j];
#pragma unroll
    for (int j = 0; j < NB; ++j) Bs[buf][b_dst[j]] = qb[j];
  };
  auto lds_barrier = [&]() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int m = 0; m < WM; ++m)
#pragma unroll
    for (int n = 0; n < WN; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  auto compute = [&](int buf) {
#pragma unroll
    for (int cl = 0; cl < kKC; ++cl) {
      half8 af[3][WM], bf[2][WN];
#pragma unroll
      for (int m = 0; m < WM; ++m) {
        af[0][m] = __builtin_bit_cast(half8, As[buf][(cl * 4 + 0 + h) * BM + wr * (WM * 32) + m * 32 + i]);
        af[2][m] = __builtin_bit_cast(half8, As[buf][(cl * 4 + 2 + h) * BM + wr * (WM * 32) + m * 32 + i]);
        af[1][m] = af[0][m] * (_Float16)(1.f / kH3Scale);
      }
#pragma unroll
      for (int op = 0; op < 2; ++op)
#pragma unroll
        for (int n = 0; n < WN; ++n)
          bf[op][n] = __builtin_bit_cast(half8, Bs[buf][(cl * 4 + op * 2 + h) * BNP + wc * (WN * 32) + n * 32 + i]);
#pragma unroll
      for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n) {
          acc[m][n] = h3_mfma(af[0][m], bf[0][n], acc[m][n]);   // (S wh) xh
          acc[m][n] = h3_mfma(af[1][m], bf[1][n], acc[m][n]);   // wh (S xl)
          acc[m][n] = h3_mfma(af[2][m], bf[0][n], acc[m][n]);   // (S wl) xh
        }
    }
  };
  // iteration st: stage st + 1 (set q) goes to LDS buffer (st + 1) & 1 -- nobody reads it any more: every wave passed
  // the barrier behind compute(st - 1) -- its registers are refilled with stage st + 3, then the MFMAs of stage st run
  auto iter = [&](int st, uint4 (&qa)[NA], uint4 (&qb)[NB]) {
    if (st + 1 < nst) commit((st + 1) & 1, qa, qb);
    fetch(st + 3, qa, qb);             // unconditional: stages past the end read zeros (buffer bounds), and the number of
    if (st < nst) compute(st & 1);     // loads in flight stays known to the compiler -> counted vmcnt waits, never vmcnt(0)
    lds_barrier();
  };

  // ragged batches: a tile whose rows all lie behind one item's last frame has nothing to multiply -- its outputs are
  // the zeros gemm_store_tile writes for dead rows (workgroup-uniform: no barrier is skipped by a part of the group)
  bool dead = false;
  if (a.lens) {
    const long last = (n0 + BN <= a.rows ? n0 + BN : a.rows) - 1;
    const long b0 = n0 / a.T;
    dead = last / a.T == b0 && (int)(n0 - b0 * a.T) >= a.lens[b0];
  }
  if (!dead) {
    fetch(0, ra[0], rb[0]);
    fetch(1, ra[1], rb[1]);
    commit(0, ra[0], rb[0]);
    fetch(2, ra[0], rb[0]);
    lds_barrier();
    if constexpr (NSEG > 1 && !SPLIT) {
      f32x16 tot[WM][WN];
      for (int st = 0; st < nst; st += 2) {
        iter(st, ra[1], rb[1]);
        iter(st + 1, ra[0], rb[0]);
        if ((st + 2) % seg_len == 0) {     // a segment ends here (seg_len is even): fold its sum, start the next from zero
          const bool first = st + 2 == seg_len;
#pragma unroll
          for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int n = 0; n < WN; ++n)
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                tot[m][n][r] = first ? acc[m][n][r] : tot[m][n][r] + acc[m][n][r];
                acc[m][n][r] = 0.f;
              }
        }
      }
#pragma unroll
      for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n) acc[m][n] = tot[m][n];
    } else {
      for (int st = 0; st < nst; st += 2) {
        iter(st, ra[1], rb[1]);
        iter(st + 1, ra[0], rb[0]);        // st + 1 == nst on an odd stage count: nothing to multiply, see iter
      }
    }
  }

  if constexpr (SPLIT) {
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
      for (int n = 0; n < WN; ++n)
        gemm_store_part(a, (int)blockIdx.z, co0 + wr * (WM * 32) + m * 32, n0 + wc * (WN * 32) + n * 32 + i, h, acc[m][n]);
    return;
  }
  constexpr float inv = 1.f / kH3Scale;
  bool ovf = false;
#pragma unroll
  for (int m = 0; m < WM; ++m)
#pragma unroll
    for (int n = 0; n < WN; ++n) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] *= inv;
      gemm_store_tile(a, co0 + wr * (WM * 32) + m * 32, n0 + wc * (WN * 32) + n * 32 + i, h, acc[m][n], ovf);
    }
  if (ovf) report_h3_overflow(a.ovf, a.ovf_next, a.seq);
}


// ---- gemm_bd_kernel (round 6): the same arithmetic with the ACTIVATIONS fed straight from global memory ----------------
// gemm_h3_kernel stages both operands through LDS: per 16-channel chunk a 64 x 64 wave tile reads eight 16-byte fragments
// for twelve MFMAs, and a 128 x 128 workgroup commits another 16 KB -- 125 B / clk / CU of LDS traffic with two
// workgroups per CU against the ~115 the LDS delivers for 16-byte reads (LABNOTES "Round 4 (a)"): the batched Linear
// layers of HuBERT sat at 150 - 185 TFLOP/s because of it.  The split input image is ALREADY the MFMA B fragment (a row's
// 16-byte element (op, h) of a chunk is exactly what lane (row, h) feeds), so here a wave loads its own 64 rows' fragments
// with plain 16-byte buffer loads -- no LDS write, no LDS read, no sharing needed: the four waves of a workgroup own
// DIFFERENT rows -- and only the weights (shared by all four waves) go through LDS.  Tile: 128 output channels x 256 rows
// (a wave = 128 x 64: eight accumulators); per chunk and wave 8 weight-fragment reads for 24 MFMAs (0.33 per MFMA against
// 0.67), LDS traffic per CU a third.  Same k-order (chunk-major, hh / hl / lh), same epilogue: bit-identical to every
// gemm_h3 tile.  WN = 1 (128 x 128) is the form for the long-K layers, whose four K segments need a second accumulator set.
template <typename F, int... N>
__device__ __forceinline__ void gemm_for_tiles(std::integer_sequence<int, N...>, F&& f) {
  (f(std::integral_constant<int, N>{}), ...);
}

template <int WN, int NSEG>
__global__ __launch_bounds__(256, 2) void gemm_bd_kernel(const GemmArgs a) {
  constexpr int BM = 128, WM = 4, BN = 4 * WN * 32;
  constexpr int A_ST = kKC * 4 * BM;                     // 16-byte elements of one weight stage
  constexpr int NA = A_ST / 256;
  __shared__ uint4 As[2][A_ST];                          // [buf][cl][op*2 + h][co]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int co0 = blockIdx.y * BM;
  const long n0 = (long)blockIdx.x * BN;
  const int nchunk = a.cin_p / 16;
  const int nst = (nchunk + kKC - 1) / kKC;
  const int seg_len = nst / NSEG;
  const H3Rsrc wres = h3_rsrc(a.w_h3, nchunk * 4 * a.cout_p * 16);
  const long rows_here = a.rows - n0 < BN ? a.rows - n0 : BN;
  const H3Rsrc xres = h3_rsrc(static_cast<const char*>(a.xs) + n0 * a.ld_xs, (int)(rows_here * a.ld_xs));
  const int slab = 4 * a.cout_p * 16;
  const int row_bytes = a.cin_p * 4;

  int a_off[NA], a_cl[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int e = tid + 256 * j;                         // (cl, oph, co)
    const int co = e % BM, rest = e / BM;
    a_cl[j] = rest >> 2;
    a_off[j] = co0 + co < a.cout_p ? ((rest & 3) * a.cout_p + co0 + co) * 16 : kH3Oob;
  }
  // this lane's rows: block n of the wave's 64-row strip; byte offset of (row, h) inside the tile's buffer
  int b_off[WN];
#pragma unroll
  for (int n = 0; n < WN; ++n) {
    const int p = wave * (WN * 32) + n * 32 + i;
    b_off[n] = p < rows_here ? (int)(p * a.ld_xs) + h * 16 : kH3Oob;
  }

  // Registers: ONE weight set (stage st + 1 travels in it while stage st is multiplied) and one fragment set per chunk of a
  // stage (refilled with the next stage's chunk right behind its last use) -- two full sets of each spilled at 256 registers
  uint4 ra[NA];
  uint4 rb[kKC][2][WN];                                  // [cl][op][n]
  auto fetch_a = [&](int st) {
    const int c0 = st * kKC;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int ch = c0 + a_cl[j];
      ra[j] = h3_load4(wres, (a_off[j] != kH3Oob && ch < nchunk) ? ch * slab + a_off[j] : kH3Oob);
    }
  };
  auto fetch_b = [&](int st, auto clt) {
    constexpr int cl = decltype(clt)::value;
    const int cb = (st * kKC + cl) * 64;                  // byte offset of the chunk inside a row
#pragma unroll
    for (int op = 0; op < 2; ++op)
#pragma unroll
      for (int n = 0; n < WN; ++n)
        rb[cl][op][n] = h3_load4(xres, (b_off[n] != kH3Oob && cb < row_bytes) ? b_off[n] + cb + op * 32 : kH3Oob);
  };
  auto commit_a = [&](int buf) {
#pragma unroll
    for (int j = 0; j < NA; ++j) As[buf][tid + 256 * j] = ra[j];
  };
  auto lds_barrier = [&]() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int m = 0; m < WM; ++m)
#pragma unroll
    for (int n = 0; n < WN; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  auto compute = [&](int buf, auto clt) {
    constexpr int cl = decltype(clt)::value;
#pragma unroll
    for (int m = 0; m < WM; ++m) {
      const half8 af0 = __builtin_bit_cast(half8, As[buf][(cl * 4 + 0 + h) * BM + m * 32 + i]);
      const half8 af2 = __builtin_bit_cast(half8, As[buf][(cl * 4 + 2 + h) * BM + m * 32 + i]);
      const half8 af1 = af0 * (_Float16)(1.f / kH3Scale);
#pragma unroll
      for (int n = 0; n < WN; ++n) {
        const half8 bf0 = __builtin_bit_cast(half8, rb[cl][0][n]);
        const half8 bf1 = __builtin_bit_cast(half8, rb[cl][1][n]);
        acc[m][n] = h3_mfma(af0, bf0, acc[m][n]);          // (S wh) xh
        acc[m][n] = h3_mfma(af1, bf1, acc[m][n]);          // wh (S xl)
        acc[m][n] = h3_mfma(af2, bf0, acc[m][n]);          // (S wl) xh
      }
    }
  };
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  static_assert(kKC == 2, "gemm_bd: two chunks per stage");
  // iteration st: the weights of stage st + 1 (fetched during iteration st - 1) go to LDS buffer (st + 1) & 1 -- nobody reads
  // it any more -- and stage st + 2 is requested; each chunk's MFMAs are followed by the request for the same chunk of the
  // next stage into the fragment registers they just freed
  auto iter = [&](int st) {
    if (st + 1 < nst) commit_a((st + 1) & 1);
    fetch_a(st + 2);
    compute(st & 1, C0{});
    fetch_b(st + 1, C0{});
    compute(st & 1, C1{});
    fetch_b(st + 1, C1{});
    lds_barrier();
  };

  bool dead = false;
  if (a.lens) {
    const long last = (n0 + BN <= a.rows ? n0 + BN : a.rows) - 1;
    const long b0 = n0 / a.T;
    dead = last / a.T == b0 && (int)(n0 - b0 * a.T) >= a.lens[b0];
  }
  if (!dead) {
    fetch_a(0);
    fetch_b(0, C0{});
    fetch_b(0, C1{});
    commit_a(0);
    fetch_a(1);
    lds_barrier();
    if constexpr (NSEG > 1) {
      f32x16 tot[WM][WN];
      for (int st = 0; st < nst; ++st) {
        iter(st);
        if ((st + 1) % seg_len == 0) {       // a segment ends here: fold its sum, start the next from zero
          const bool first = st + 1 == seg_len;
#pragma unroll
          for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int n = 0; n < WN; ++n)
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                tot[m][n][r] = first ? acc[m][n][r] : tot[m][n][r] + acc[m][n][r];
                acc[m][n][r] = 0.f;
              }
        }
      }
#pragma unroll
      for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n) acc[m][n] = tot[m][n];
    } else {
      for (int st = 0; st < nst; ++st) iter(st);
    }
  }
  constexpr float inv = 1.f / kH3Scale;
  bool ovf = false;
  // compile-time loop over the eight accumulator tiles: a `#pragma unroll` loop around the inlined epilogue is silently left
  // rolled, the accumulators are then indexed at run time and live in scratch (conv_deep.hip has the same note)
  gemm_for_tiles(std::make_integer_sequence<int, WM * WN>{}, [&](auto tt) {
    constexpr int m = decltype(tt)::value / WN, n = decltype(tt)::value % WN;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][n][r] *= inv;
    gemm_store_tile(a, co0 + m * 32, n0 + wave * (WN * 32) + n * 32 + i, h, acc[m][n], ovf);
  });
  if (ovf) report_h3_overflow(a.ovf, a.ovf_next, a.seq);
}

// the second half of a K-split launch: segment sums added left to right (the canonical order, see gemm_h3_kernel), then the
// epilogue of the tile kernel.  One thread per (row, four channels).
template <int NSEG>
__global__ __launch_bounds__(256) void gemm_finish_kernel(const GemmArgs a) {
  const int c4n = a.cout / 4;
  const long total = a.rows * c4n;
  bool ovf = false;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long row = e / c4n;
    const int c = (int)(e - row * c4n) * 4;
    float4 p[NSEG];
#pragma unroll
    for (int sg = 0; sg < NSEG; ++sg) p[sg] = *reinterpret_cast<const float4*>(a.part + ((long)sg * a.rows + row) * a.cout + c);
    float v[4] = {p[0].x, p[0].y, p[0].z, p[0].w};
#pragma unroll
    for (int sg = 1; sg < NSEG; ++sg) {
      v[0] += p[sg].x;
      v[1] += p[sg].y;
      v[2] += p[sg].z;
      v[3] += p[sg].w;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] *= 1.f / kH3Scale;
    const bool live = !a.lens || (int)(row % a.T) < a.lens[row / a.T];
    gemm_store4(a, c, row, live, v, ovf);
  }
  if (ovf) report_h3_overflow(a.ovf, a.ovf_next, a.seq);
}

// Exact fp32 (v_mfma_f32_32x32x2_f32): 64 x 64 tile, 4 waves of 32 x 32, 64 channels per stage, the next stage's
// operands are fetched into registers while the current one is multiplied.  x is fp32 time-major.  Only layers pinned
// to fp32 after an fp16-range overflow run here.
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmArgs a) {
  constexpr int KT = 64, P = 65, NE = KT * 64 / 256;
  __shared__ float As[KT][64];                          // [k][co]
  __shared__ float Bs[KT][P];                           // [k][p]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int i = lane & 31, h = lane >> 5;
  const int co0 = blockIdx.y * 64;
  const long n0 = (long)blockIdx.x * 64;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float qa[NE], qb[NE];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      const int e = tid + 256 * j;
      const int k = e >> 6, c = e & 63;                 // A: 64 consecutive lanes = 256 contiguous bytes of a weight row
      qa[j] = (k0 + k < a.cin && co0 + c < a.cout_p) ? a.w[(long)(k0 + k) * a.cout_p + co0 + c] : 0.f;
      const int p = e >> 6, kb = e & 63;                // B: 64 consecutive lanes = 256 contiguous bytes of an x row
      qb[j] = (n0 + p < a.rows && k0 + kb < a.cin) ? a.x[(n0 + p) * a.ld_x + k0 + kb] : 0.f;
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < a.cin; k0 += KT) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      const int e = tid + 256 * j;
      As[e >> 6][e & 63] = qa[j];
      Bs[e & 63][e >> 6] = qb[j];
    }
    __syncthreads();
    if (k0 + KT < a.cin) fetch(k0 + KT);
#pragma unroll
    for (int s = 0; s < KT / 2; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[2 * s + h][wr * 32 + i], Bs[2 * s + h][wc * 32 + i], acc, 0, 0, 0);
  }
  bool ovf = false;
  gemm_store_tile(a, co0 + wr * 32, n0 + wc * 32 + i, h, acc, ovf);
  if (ovf) report_h3_overflow(a.ovf, a.ovf_next, a.seq);
}

constexpr int kNumGemmTiles = 4;
constexpr int kGemmSeg = 4;              // segments of a long-K layer's canonical summation order
struct GemmCfg {
  int bm, bn;
  size_t lds;
  void (*kern)(const GemmArgs);
  void (*kern_seg)(const GemmArgs);     // the same tile, K summed in kGemmSeg segments by one workgroup
  void (*kern_split)(const GemmArgs);   // one segment per workgroup (blockIdx.z)
};
template <int BM, int BN>
constexpr GemmCfg gemm_cfg() {
    return {BM, BN, (size_t)2 * kKC * 4 * (BM + BN + 1) * 16, gemm_h3_kernel<BM, BN>, gemm_h3_kernel<BM, BN, kGemmSeg, false>,
            gemm_h3_kernel<BM, BN, kGemmSeg, true>};
}
const GemmCfg kGemm[] = {gemm_cfg<128, 128>(), gemm_cfg<64, 128>(), gemm_cfg<128, 64>(), gemm_cfg<64, 64>()};
constexpr int kNumGemm = sizeof(kGemm) / sizeof(kGemm[0]);

}  // namespace

bool gemm_h3_enabled() {
  static const int mode = getenv("RVCX_GEMM") ? atoi(getenv("RVCX_GEMM")) : 1;
  return mode != 0 && conv_h3_enabled();
}

constexpr int kGemmSlot0 = 53, kGemmF32Slot = 57, kGemmBdSlot = 64;

void gemm_describe(ConvProfile* p) {
  for (int t = 0; t < kNumGemm; ++t) {
    p->bm[kGemmSlot0 + t] = kGemm[t].bm;
    p->bn[kGemmSlot0 + t] = kGemm[t].bn;
    p->halo[kGemmSlot0 + t] = 600000;
  }
  p->bm[kGemmF32Slot] = 64;
  p->bn[kGemmF32Slot] = 64;
  p->halo[kGemmF32Slot] = 600001;
  p->bm[kGemmBdSlot] = 128;
  p->bn[kGemmBdSlot] = 256;
  p->halo[kGemmBdSlot] = 600002;
}

void gemm_init() {      // per DEVICE: a second-GPU context of the same process needs the >64 KB dynamic-LDS attribute too
  static std::mutex mu;
  static uint64_t done = 0;
  int dev = 0;
  RVCX_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> g(mu);
  if ((done >> (dev & 63)) & 1) return;
  for (const auto& c : kGemm)
    for (auto k : {c.kern, c.kern_seg, c.kern_split})
      if (k)
        RVCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.lds));
  done |= 1ull << (dev & 63);
}

// returns the profile slot of the kernel that ran
int launch_gemm(GemmArgs a, hipStream_t stream) {
  RVCX_CHECK(a.rows > 0 && a.cout % 4 == 0 && a.T > 0, "gemm: bad shape");
  RVCX_CHECK(a.y || a.ys || a.y_cf, "gemm: no output");
  if (a.xs && a.w_h3 && gemm_h3_enabled()) {
    RVCX_CHECK(a.cin_p % 16 == 0 && (long)kBnMax * a.ld_xs < kH3Oob && (long)a.cin_p * a.cout_p * 4 < kH3Oob, "gemm: operand too large");
    // Tile choice (the k-order does not depend on it: every tile gives the same bits).  Two workgroups fit a CU; a
    // workgroup's time ~ its MFMA work + a fixed prologue / epilogue.
    const int forced = g_conv_override.tile >= 200 ? g_conv_override.tile - 200 : -1;   // tuning / tests (rvcx_conv_override)
    // Measured on the HuBERT-base shapes (tools/bench_gemm.py): the launches are small against the chip -- 1200 to 3600
    // output tiles of 32 x 32 for 1024 SIMDs at B = 1 -- so what counts is how many workgroups share a CU (they hide
    // each other's load latency), not operand reuse; the large tile only pays once every CU holds two of them.
    auto blocks = [&](int t) { return (long)cdiv(a.cout_p, kGemm[t].bm) * cdiv64(a.rows, kGemm[t].bn); };
    // Long-K layers (HuBERT fc2: K = 3072) sum K in kGemmSeg segments -- a function of the LAYER's shape only, so every launch
    // of the layer has the same bits.  At B = 1 such a launch is 300 workgroups walking 96 dependent stages (62 us at 15 %
    // matrix-pipe duty); when the chip is underfilled the segments go to separate workgroups (K-split) + a finish pass.
    const int nchunk = a.cin_p / 16, nst = (nchunk + kKC - 1) / kKC;
    static const int seg_mode = getenv("RVCX_GEMM_KSEG") ? atoi(getenv("RVCX_GEMM_KSEG")) : 1;   // 0 off, 1 auto, 2 never split, 3 always split
    const bool seg = seg_mode != 0 && nchunk >= 128 && nst % (2 * kGemmSeg) == 0;
    int best = 3;                                          // 64 x 64
    if (blocks(0) >= 512) best = 0;                        // 128 x 128 (its segmented form: 248 VGPRs, still two waves per SIMD)
    else if (blocks(1) >= 384) best = 1;                   // 64 x 128
    if (forced >= 0 && forced < kNumGemm && !(seg && !kGemm[forced].kern_seg)) best = forced;
    gemm_init();
    // Round 6: the B-direct tile (gemm_bd_kernel: 128 x 256, long-K layers 128 x 128).  Bit-identical to the other tiles and
    // OFF by default: measured (tools/bench_gemm.py, B = 16) 199 / 166 / 205 / 210 TFLOP/s on qkv / o / fc1 / fc2 against
    // 221 / 212 / 218 / 223 for the LDS-staged 128 x 128 tile, HuBERT of 16 clips 51.1 against 47.5 ms, C3 1438 against
    // 1453x -- a third of the LDS traffic buys nothing: the LDS was not what bounds these launches (LABNOTES 12).
    // RVCX_GEMM_BD=1 (auto: grids of >= 512 workgroups) / 2 (always), override tile 204 (tests).
    {
      static const int bd_mode = getenv("RVCX_GEMM_BD") ? atoi(getenv("RVCX_GEMM_BD")) : 0;      // 0 off, 1 auto, 2 always
      const bool want = forced == 4 || (forced < 0 && bd_mode != 0);
      const int force_split = g_conv_override.splitk;
      if (want && a.cout_p % 128 == 0 && !(seg && force_split >= 2)) {
        const int bn = seg ? 128 : 256;
        const long blocks_bd = (long)(a.cout_p / 128) * cdiv64(a.rows, bn);
        if (forced == 4 || bd_mode == 2 || blocks_bd >= 512) {
          dim3 grid((unsigned)cdiv64(a.rows, bn), a.cout_p / 128, 1);
          if (seg) hipLaunchKernelGGL((gemm_bd_kernel<1, kGemmSeg>), grid, dim3(256), 0, stream, a);
          else hipLaunchKernelGGL((gemm_bd_kernel<2, 1>), grid, dim3(256), 0, stream, a);
          RVCX_HIP(hipGetLastError());
          return kGemmBdSlot;
        }
      }
    }
    if (seg) {
      const bool fits = a.part && (long)kGemmSeg * a.rows * a.cout <= a.part_cap;
      const int force = g_conv_override.splitk;           // tests (rvcx_conv_override): 1 never, >= 2 always
      const bool split = fits && force != 1 && (force >= 2 || seg_mode == 3 || (seg_mode == 1 && blocks(best) < 512));
      if (split) {      // the tile of the K-split form: the largest whose four-fold grid still covers the chip
        static const int split_tile = getenv("RVCX_GEMM_SPLIT_TILE") ? atoi(getenv("RVCX_GEMM_SPLIT_TILE")) : -1;
        // fc2 of a 30 s clip (tools/bench_gemm.py): 128 x 64 46.5 us, 128 x 128 51, 64 x 64 55, 64 x 128 56 (one workgroup per tile: 62 - 69)
        best = kGemmSeg * blocks(2) >= 256 ? 2 : 3;
        if (split_tile >= 0 && split_tile < kNumGemmTiles) best = split_tile;
        if (forced >= 0 && forced < kNumGemm) best = forced;
      }
      const GemmCfg& F = kGemm[best];
      if (split) {
        dim3 grid((unsigned)cdiv64(a.rows, F.bn), cdiv(a.cout_p, F.bm), kGemmSeg);
        hipLaunchKernelGGL(F.kern_split, grid, dim3(256), F.lds, stream, a);
        const long total = a.rows * (a.cout / 4);
        hipLaunchKernelGGL(gemm_finish_kernel<kGemmSeg>, dim3((unsigned)std::min<long>(cdiv64(total, 256), 4096)), dim3(256), 0,
                           stream, a);
      } else {
        dim3 grid((unsigned)cdiv64(a.rows, F.bn), cdiv(a.cout_p, F.bm), 1);
        hipLaunchKernelGGL(F.kern_seg, grid, dim3(256), F.lds, stream, a);
      }
      RVCX_HIP(hipGetLastError());
      return kGemmSlot0 + best;
    }
    const GemmCfg& F = kGemm[best];
    dim3 grid((unsigned)cdiv64(a.rows, F.bn), cdiv(a.cout_p, F.bm), 1);
    hipLaunchKernelGGL(F.kern, grid, dim3(256), F.lds, stream, a);
    RVCX_HIP(hipGetLastError());
    return kGemmSlot0 + best;
  }
  RVCX_CHECK(a.x && a.w, "gemm: the exact-fp32 kernel needs fp32 activations and weights");
  dim3 grid((unsigned)cdiv64(a.rows, 64), cdiv(a.cout_p, 64), 1);
  hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, stream, a);
  RVCX_HIP(hipGetLastError());
  return kGemmF32Slot;
}

}  // namespace rvcx
