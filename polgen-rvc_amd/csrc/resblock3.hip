// A whole ResBlock1 with kernel size 3 (rvc/lib/algorithm/residuals.py:15-62: three steps
//     x = x + c2_s(leaky_relu(c1_s(leaky_relu(x)))),   c1_s dilated by d_s = 1, 3, 5,   c2_s dilation 1)
// as ONE kernel, for the narrow stages of the NSF decoder (C = 32 and 64).
//
// Round 5's phase stamps of the fused single step (resblock.hip) show the k = 3 steps at C = 32 / 64 spend 80 - 90 % of a
// tile's life outside their two k-loops -- and what is left after the persistent form took the latencies away is work that a
// step-by-step execution repeats three times: the trip of the whole activation tensor through HBM (393 MB per step and 30 s
// clip) and the fp32 -> split-fp16 conversion of the step's input.  The total receptive field of the three steps is only 12
// positions per side ((1 + 1) + (3 + 1) + (5 + 1)), so a workgroup that owns all C channels of N1 positions can run the three
// steps back to back on its own tile: the step's output stays in registers as the next residual (accumulator layout, fp32)
// and is written -- already activated and split -- into LDS as the next c1 input.  HBM: read x once, write y once.
//
// Every tensor of the tile lives on ONE grid of N1 columns (column j = position n0 - 12 + j); a conv reads its input at
// j + (tap - 1) d, so each step spoils d + 1 more columns at both edges of the tile; after three steps columns
// [12, N1 - 12) are exact and are the ones stored.  The pad columns beside the grid stay zero (they are read, never
// written), spoiled columns hold finite garbage that is never stored.
//
// Arithmetic: the single step's, operation for operation (split, MFMA order, epilogue order (acc / S + b2) + x, the
// leaky_relu form of the input conversion, zeros outside [0, len)), so the result is bit-identical to three launches of the
// fused step and to the six conv launches (tests/test_gpu_round5.py::test_resblock3_equals_three_fused_steps).
// Persistent like resblock_pair_persist_kernel: a workgroup walks a contiguous range of tiles, the next tile's input is
// requested during the third step and committed in that step's last c2 stage.
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <type_traits>

#include "conv.h"
#include "conv_device.h"
#include "h3_device.h"

namespace rvcx {

namespace {

constexpr int kHalo3 = 12;     // (1 + 1) + (3 + 1) + (5 + 1)
constexpr int kPad3 = 8;       // zero columns on both sides of the grid (dilation 5 reaches 5 columns out)

template <int C, int NT, int WR, int WC, int NCS>
__global__ __launch_bounds__(64 * WR * WC) void resblock3_kernel(const Block3Args a) {
  constexpr int K = 3;
  constexpr int THREADS = 64 * WR * WC;
  constexpr int N1 = 32 * NT, XP = N1 + 2 * kPad3;
  constexpr int WM = (C / 32) / WR, WN = NT / WC;
  constexpr int NCHUNK = C / 16, NST = NCHUNK / NCS;       // weight stages per conv
  constexpr int A_ELEMS = NCS * K * 4 * C, NA = (A_ELEMS + THREADS - 1) / THREADS;
  constexpr int B_TASKS = NCHUNK * 2 * N1, NBT = (B_TASKS + THREADS - 1) / THREADS;
  constexpr int BN_OUT = N1 - 2 * kHalo3;
  static_assert(WM >= 1 && WN >= 1 && WM * WR * 32 == C && WN * WC == NT, "bad tile");
  static_assert(NCHUNK % NCS == 0 && BN_OUT % 4 == 0, "bad tile");
  extern __shared__ uint4 lds[];
  constexpr int XS_ELEMS = NCHUNK * 4 * XP;
  uint4* Xs = lds;                               // split(lrelu(x_s))      [chunk][op][h][XP], column j at kPad3 + j
  uint4* Y1 = Xs + XS_ELEMS;                     // split(lrelu(c1 + b1))  same shape
  uint4* As = Y1 + XS_ELEMS;                     // [cl * K + kk][op][h][C]
  float* bias_s = reinterpret_cast<float*>(As + A_ELEMS);   // b1 / b2 of the three steps: [6][C]

  int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int wr = wave / WC, wc = wave % WC, i = lane & 31, h = lane >> 5, lr = lane >> 3, lc = (lane & 7) * 4;
  const int b = blockIdx.z;
  const int len = a.lens ? a.lens[b] : a.T;
  const float slope = a.slope;
  const H3Rsrc xr = h3_rsrc(a.x + (long)b * a.bs, C * a.cs * 4);
  const int xrow = a.cs * 4;
  constexpr int slab = 4 * C * 16;               // bytes of one (tap, chunk) weight slab
  constexpr float inv = 1.f / kH3Scale;

  // this workgroup's tiles (as in resblock_pair_persist_kernel)
  int first, count;
  const int ntiles_all = (a.T + BN_OUT - 1) / BN_OUT;
  {
    const int nt = ntiles_all, nwg = gridDim.x, w = blockIdx.x;
    if (a.xcd_order && nwg >= 8) {
      const int x = w & 7, j = w >> 3;
      const int q = nt >> 3, r = nt & 7;
      const int x0 = x * q + (x < r ? x : r), xn = q + (x < r ? 1 : 0);
      const int m = (nwg >> 3) + (x < (nwg & 7) ? 1 : 0);
      const int q2 = xn / m, r2 = xn % m;
      first = x0 + j * q2 + (j < r2 ? j : r2);
      count = q2 + (j < r2 ? 1 : 0);
    } else {
      const int q = nt / nwg, r = nt % nwg;
      first = w * q + (w < r ? w : r);
      count = q + (w < r ? 1 : 0);
    }
  }
  if (count <= 0) return;

  // once per launch: zero pad columns of both tiles, biases
  for (int e = tid; e < NCHUNK * 4 * 2 * kPad3; e += THREADS) {
    const int row = e / (2 * kPad3), c = e % (2 * kPad3);
    const int col = c < kPad3 ? c : N1 + c;          // [0, kPad3) and [kPad3 + N1, XP)
    Xs[row * XP + col] = make_uint4(0, 0, 0, 0);
    Y1[row * XP + col] = make_uint4(0, 0, 0, 0);
  }
  for (int e = tid; e < 6 * C; e += THREADS) {
    const int s = e / C, c = e - s * C;
    const float* src = (s & 1) ? a.b2[s >> 1] : a.b1[s >> 1];
    bias_s[e] = src ? src[c] : 0.f;
  }

  f32x16 acc[WM][WN], res[WM][WN];
  auto zero_acc = [&]() {
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
      for (int n = 0; n < WN; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  };
  int a_cl[NA], a_kk[NA], a_off[NA];
  int b_off[NBT], b_row[NBT], b_col[NBT];
  auto derive_thread = [&]() {
    int t0 = threadIdx.x;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(t0));                   // opaque: see resblock_pair_persist_kernel
#endif
    tid = t0;
    lane = tid & 63;
    wave = tid >> 6;
    wr = wave / WC;
    wc = wave % WC;
    i = lane & 31;
    h = lane >> 5;
    lr = lane >> 3;
    lc = (lane & 7) * 4;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int e = tid + THREADS * j;
      const int co = e % C, rest = e / C;          // rest = (slot * 2 + op) * 2 + h
      const int slot = rest / 4;
      a_cl[j] = slot / K;
      a_kk[j] = slot % K;
      a_off[j] = e < A_ELEMS ? ((rest % 4) * C + co) * 16 : kH3Oob;
    }
#pragma unroll
    for (int j = 0; j < NBT; ++j) {
      const int t = tid + THREADS * j;
      const int ch2 = t / N1;                      // chunk * 2 + h
      b_col[j] = t - ch2 * N1;
      b_row[j] = ch2;
    }
  };
  derive_thread();
  auto set_input_tile = [&](int tile) {
    const int base = tile * BN_OUT - kHalo3;
#pragma unroll
    for (int j = 0; j < NBT; ++j) {
      const int pos = base + b_col[j];
      const bool ok = tid + THREADS * j < B_TASKS && pos >= 0 && pos < len;
      b_off[j] = ok ? ((b_row[j] >> 1) * 16 + (b_row[j] & 1) * 8) * xrow + pos * 4 : kH3Oob;
    }
  };
  uint4 ra[NA];
  float rb[NBT][8];
  bool ovf = false;
  auto fetch_b = [&]() {
#pragma unroll
    for (int j = 0; j < NBT; ++j)
#pragma unroll
      for (int q = 0; q < 8; ++q) rb[j][q] = h3_load1(xr, b_off[j] == kH3Oob ? kH3Oob : b_off[j] + q * xrow);
  };
  // stage st of the tile's 6 convs: conv = st / NST (even: c1 of step conv / 2, odd: its c2), chunk set st % NST
  auto fetch_a = [&](int st) {
    const int cv = st / NST, cs = st - cv * NST;
    const void* wp = (cv & 1) ? a.w2[cv >> 1] : a.w1[cv >> 1];
    const H3Rsrc wr_ = h3_rsrc(wp, K * NCHUNK * 4 * C * 16);
#pragma unroll
    for (int j = 0; j < NA; ++j)
      ra[j] = h3_load4(wr_, a_off[j] != kH3Oob ? (a_kk[j] * NCHUNK + cs * NCS + a_cl[j]) * slab + a_off[j] : kH3Oob);
  };
  auto commit_a = [&]() {
#pragma unroll
    for (int j = 0; j < NA; ++j)
      if (NA * THREADS == A_ELEMS || tid + THREADS * j < A_ELEMS) As[tid + THREADS * j] = ra[j];
  };
  auto split8 = [&](const float* v, half8& hi, half8& lo) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float t = v[q];
      t = t > 0.f ? t : t * slope;                   // leaky_relu ahead of c1 (the input conversion's form)
      ovf |= !(fabsf(t) < kH3ActLimit);
      const _Float16 th = (_Float16)t;
      hi[q] = th;
      lo[q] = (_Float16)((t - (float)th) * kH3Scale);
    }
  };
  auto commit_b = [&]() {                             // rb -> Xs
#pragma unroll
    for (int j = 0; j < NBT; ++j) {
      if (NBT * THREADS == B_TASKS || tid + THREADS * j < B_TASKS) {
        half8 hi, lo;
        split8(rb[j], hi, lo);
        const int chunk = b_row[j] >> 1, hh = b_row[j] & 1;
        Xs[((chunk * 2 + 0) * 2 + hh) * XP + kPad3 + b_col[j]] = __builtin_bit_cast(uint4, hi);
        Xs[((chunk * 2 + 1) * 2 + hh) * XP + kPad3 + b_col[j]] = __builtin_bit_cast(uint4, lo);
      }
    }
  };
  // the residual of step 0: x in accumulator layout (column i of tile (m, n), rows (r & 3) + 8 (r >> 2) + 4 h)
  auto fetch_res = [&](int tile) {
    const int base = tile * BN_OUT - kHalo3;
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
      for (int n = 0; n < WN; ++n) {
        const int pos = base + wc * (WN * 32) + n * 32 + i;
        const bool ok = pos >= 0 && pos < len;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = wr * (WM * 32) + m * 32 + 4 * h + (r & 3) + 8 * (r >> 2);
          res[m][n][r] = h3_load1(xr, ok ? co * xrow + pos * 4 : kH3Oob);
        }
      }
  };
  // k-steps of one stage: chunk set `cs` (NCS chunks x 3 taps) of the tile Bt (column 0 = grid column -tap_step)
  auto compute = [&](const uint4* Bt, int cs, int tap_step) {
    constexpr int NS = NCS * K;
    half8 af[2][2][WM], bf[2][2][WN];
    auto load = [&](int buf, int s) {
      const int cl = s / K, kk = s % K;
      const uint4* Bc = Bt + (cs * NCS + cl) * 4 * XP + kk * tap_step;
#pragma unroll
      for (int m = 0; m < WM; ++m) {
        af[buf][0][m] = __builtin_bit_cast(half8, As[(((cl * K + kk) * 2 + 0) * 2 + h) * C + wr * (WM * 32) + m * 32 + i]);
        af[buf][1][m] = __builtin_bit_cast(half8, As[(((cl * K + kk) * 2 + 1) * 2 + h) * C + wr * (WM * 32) + m * 32 + i]);
      }
#pragma unroll
      for (int op = 0; op < 2; ++op)
#pragma unroll
        for (int n = 0; n < WN; ++n)
          bf[buf][op][n] = __builtin_bit_cast(half8, Bc[(op * 2 + h) * XP + wc * (WN * 32) + n * 32 + i]);
    };
    load(0, 0);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int cur = s & 1;
      if (s + 1 < NS) load(cur ^ 1, s + 1);
#pragma unroll
      for (int m = 0; m < WM; ++m) {
        const half8 wh = af[cur][0][m] * (_Float16)(1.f / kH3Scale);
#pragma unroll
        for (int n = 0; n < WN; ++n) {
          acc[m][n] = h3_mfma(af[cur][0][m], bf[cur][0][n], acc[m][n]);   // (S wh) xh
          acc[m][n] = h3_mfma(wh, bf[cur][1][n], acc[m][n]);              // wh (S xl)
          acc[m][n] = h3_mfma(af[cur][1][m], bf[cur][0][n], acc[m][n]);   // (S wl) xh
        }
      }
    }
    {
      constexpr int R = 2 * WM + 2 * WN, M = 3 * WM * WN;
      __builtin_amdgcn_sched_group_barrier(0x100, R, 0);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int j = 0; j < M; ++j) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (s + 1 < NS && j < R) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if (s + 1 < NS && R > M) __builtin_amdgcn_sched_group_barrier(0x100, R - M, 0);
      }
    }
  };
  // accumulator tile (m, n) -> split halves in `dst` (the c1 epilogue of the single step, shared by both epilogues here)
  typedef _Float16 half4 __attribute__((ext_vector_type(4)));
  auto store_split = [&](uint4* dst, int m, int n, const float (&v)[16]) {
    const int c_t = wr * (WM * 32) + m * 32;
    const int j = wc * (WN * 32) + n * 32 + i;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int cg = c_t + 8 * g;
      half4 hi, lo;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float t = v[4 * g + q];
        ovf |= !(fabsf(t) < kH3ActLimit);
        const _Float16 th = (_Float16)t;
        hi[q] = th;
        lo[q] = (_Float16)((t - (float)th) * kH3Scale);
      }
      char* e_hi = reinterpret_cast<char*>(dst + (((cg >> 4) * 2 + 0) * 2 + (g & 1)) * XP + kPad3 + j) + 8 * h;
      char* e_lo = reinterpret_cast<char*>(dst + (((cg >> 4) * 2 + 1) * 2 + (g & 1)) * XP + kPad3 + j) + 8 * h;
      *reinterpret_cast<half4*>(e_hi) = hi;
      *reinterpret_cast<half4*>(e_lo) = lo;
    }
  };

  // ---- first tile: input -> Xs, residual -> registers, first weights -> registers
  set_input_tile(first);
  fetch_b();
  fetch_res(first);
  fetch_a(0);
  __syncthreads();                                   // pad columns / biases are in place
  commit_b();

  for (int ti = 0; ti < count; ++ti) {
    const int tile = first + ti;
    const int n0 = tile * BN_OUT;
    const bool has_next = ti + 1 < count;
    int st = 0;                                      // weight stage of the tile (0 .. 6 NST - 1); ra holds stage st
#pragma unroll 1
    for (int step = 0; step < 3; ++step) {
      const int d = step == 0 ? a.dil[0] : (step == 1 ? a.dil[1] : a.dil[2]);
      // per step, not per tile: with the three steps unrolled the address terms of a whole tile stay live otherwise
      // (first build: 256 registers + 225 spilled)
      if (ti > 0 || step > 0) derive_thread();
      // ================================================================ c1: Y1 = lrelu(c1(Xs) + b1)
      zero_acc();
      for (int cs = 0; cs < NST; ++cs) {
        __syncthreads();
        commit_a();
        __syncthreads();
        ++st;
        fetch_a(st);                                 // next stage: more of c1, or the first of c2
        compute(Xs + kPad3 - d, cs, d);
      }
      if (step == 2 && has_next) {                   // Xs is read for the last time above: tile t + 1's input leaves HBM now
        set_input_tile(tile + 1);
        fetch_b();
      }
#pragma unroll
      for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n) {
          const int c_t = wr * (WM * 32) + m * 32;
          const int j = wc * (WN * 32) + n * 32 + i;
          const int pos = n0 - kHalo3 + j;
          const bool live = pos >= 0 && pos < len;
          float v[16];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 bq = *reinterpret_cast<const float4*>(bias_s + (2 * step) * C + c_t + 8 * g + 4 * h);
            const float b4[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              float t = acc[m][n][4 * g + q] * inv + b4[q];
              t = fmaxf(t, t * slope);
              v[4 * g + q] = live ? t : 0.f;
            }
          }
          store_split(Y1, m, n, v);
        }
      // ================================================================ c2: x' = (c2(Y1) + b2) + x
      zero_acc();
      for (int cs = 0; cs < NST; ++cs) {
        const bool last_stage = step == 2 && cs == NST - 1;
        __syncthreads();
        commit_a();
        if (last_stage && has_next) commit_b();      // Xs is idle from here to the next tile: its input is committed now
        __syncthreads();
        ++st;
        if (!last_stage) fetch_a(st);
        else if (has_next) fetch_a(0);               // tile t + 1's first weights
        compute(Y1 + kPad3 - 1, cs, 1);
      }
      // c2 epilogue: the new x in registers (residual of the next step) and, activated and split, in Xs (input of the next c1)
#pragma unroll
      for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n) {
          const int c_t = wr * (WM * 32) + m * 32;
          const int j = wc * (WN * 32) + n * 32 + i;
          const int pos = n0 - kHalo3 + j;
          const bool live = pos >= 0 && pos < len;
          float v[16];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 bq = *reinterpret_cast<const float4*>(bias_s + (2 * step + 1) * C + c_t + 8 * g + 4 * h);
            const float b4[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              float t = acc[m][n][4 * g + q] * inv + b4[q];
              t = t + res[m][n][4 * g + q];
              t = live ? t : 0.f;
              res[m][n][4 * g + q] = t;
              v[4 * g + q] = t > 0.f ? t : t * slope;              // leaky_relu ahead of the next c1
            }
          }
          if (step < 2) store_split(Xs, m, n, v);
        }
    }
    // ================================================================ store columns [12, N1 - 12): the wide epilogue
    {
      constexpr int P = 36;
      static_assert(THREADS / 64 * 32 * P * 4 <= XS_ELEMS * 16, "staging tiles must fit the Y1 region");
      __syncthreads();                                 // every wave is done reading Y1: the staging tiles overlay it
      float* stg = reinterpret_cast<float*>(Y1) + wave * (32 * P);
#pragma unroll
      for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n) {
#pragma unroll
          for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * h) * P + i] = res[m][n][r];
          __builtin_amdgcn_wave_barrier();
          const int col0 = wc * (WN * 32) + n * 32 + lc;                 // grid column of this lane's four values
          const int pos0 = n0 - kHalo3 + col0;
          const bool ok = col0 >= kHalo3 && col0 < N1 - kHalo3 && pos0 < a.T;     // kHalo3 and BN_OUT are multiples of 4
          float4 v[4], pv[4];
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const int co = wr * (WM * 32) + m * 32 + lr + 8 * p;
            const long off = (long)b * a.bs + (long)co * a.cs + pos0;
            v[p] = *reinterpret_cast<const float4*>(stg + (lr + 8 * p) * P + lc);
            pv[p] = (ok && a.acc2_mode != ACC2_NONE && a.acc2_mode != ACC2_SET) ? *reinterpret_cast<const float4*>(a.y2 + off)
                                                                                : make_float4(0.f, 0.f, 0.f, 0.f);
          }
          __builtin_amdgcn_wave_barrier();
          if (ok) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
              const int co = wr * (WM * 32) + m * 32 + lr + 8 * p;
              const long off = (long)b * a.bs + (long)co * a.cs + pos0;
              float o[4] = {v[p].x, v[p].y, v[p].z, v[p].w};           // already zero beyond len
              if (a.y) *reinterpret_cast<float4*>(a.y + off) = make_float4(o[0], o[1], o[2], o[3]);
              if (a.acc2_mode != ACC2_NONE) {
                const float pp[4] = {pv[p].x, pv[p].y, pv[p].z, pv[p].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  if (a.acc2_mode == ACC2_ADD) o[q] = pp[q] + o[q];
                  else if (a.acc2_mode == ACC2_ADD_DIV) o[q] = (pp[q] + o[q]) / a.acc2_div;
                }
                *reinterpret_cast<float4*>(a.y2 + off) = make_float4(o[0], o[1], o[2], o[3]);
              }
            }
          }
        }
    }
    if (has_next) {
      fetch_res(tile + 1);                             // lands during the next tile's first c1 loop
      __syncthreads();                                 // the staging tiles (Y1 region) have been read
      // the staging tiles ran over Y1's pad columns: zero them again (they feed the spoiled edge columns only, but a NaN
      // bit pattern there would trip the range guard's check of those columns)
      for (int e = tid; e < NCHUNK * 4 * 2 * kPad3; e += THREADS) {
        const int row = e / (2 * kPad3), c = e % (2 * kPad3);
        Y1[row * XP + (c < kPad3 ? c : N1 + c)] = make_uint4(0, 0, 0, 0);
      }
    }
  }
  if (ovf) report_h3_overflow(a.ovf, a.ovf_layer, a.seq);
}

struct B3Cfg {
  int C, n1, threads;
  size_t lds;
  void (*kern)(const Block3Args);
};
template <int C, int NT, int WR, int WC, int NCS>
constexpr B3Cfg make_b3() {
  constexpr size_t xs = (size_t)(C / 16) * 4 * (32 * NT + 2 * kPad3), as = (size_t)NCS * 3 * 4 * C;
  return {C, 32 * NT, 64 * WR * WC, (2 * xs + as) * 16 + 6 * C * 4, resblock3_kernel<C, NT, WR, WC, NCS>};
}
// C = 32: N1 = 512 (488 stored: 5 % halo), 8 waves of 32 x 64.  C = 64: N1 = 192 (168 stored: 14 %), 6 waves of 32 x 64, two
// chunks per weight stage (N1 = 256 would leave room for one chunk per stage only: twice the barriers)
const B3Cfg kB3[] = {make_b3<32, 16, 1, 8, 2>(), make_b3<64, 6, 2, 3, 2>()};

const B3Cfg* find_b3(int C) {
  for (const auto& c : kB3)
    if (c.C == C) return &c;
  return nullptr;
}

}  // namespace

bool resblock3_ok(const Block3Args& a) {
  // Measured (round 5, one box, serial mode): C = 32: 440 us against 3 x 158 for the three persistent single steps (-7 %);
  // C = 64: 785 us against 3 x 190 (+38 %: N1 = 192 on six waves is all the LDS allows, 14 % of every tile is halo, and
  // the step's cost was never its HBM trip -- it is the chain of short barrier-separated phases, which the whole-block
  // form keeps).  Default: C = 32 only.  RVCX_BLOCK3 = 0: off, 2: every shape the kernel supports (tests, A/B runs).
  static const int mode = getenv("RVCX_BLOCK3") ? atoi(getenv("RVCX_BLOCK3")) : 1;
  if (mode == 0 || (mode == 1 && a.C != 32 && !a.any_shape) || !resblock_pair_enabled() || !find_b3(a.C)) return false;
  for (int s = 0; s < 3; ++s)
    if (!a.w1[s] || !a.w2[s] || a.dil[s] < 1 || a.dil[s] > 5) return false;
  if (a.dil[0] + a.dil[1] + a.dil[2] + 3 > kHalo3) return false;
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (a.cs % 4 != 0 || a.T % 4 != 0 || a.bs % 4 != 0 || !al(a.x) || !al(a.y) || !al(a.y2)) return false;
  if ((long)a.C * a.cs * 4 >= kH3Oob) return false;
  return true;
}

void resblock3_init() {
  static std::mutex mu;
  static uint64_t done = 0;
  int dev = 0;
  RVCX_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> g(mu);
  if ((done >> (dev & 63)) & 1) return;
  for (const auto& c : kB3)
    RVCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(c.kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.lds));
  done |= 1ull << (dev & 63);
}

int resblock3_n1(int C) {
  const B3Cfg* c = find_b3(C);
  return c ? c->n1 : 0;
}

void launch_resblock3(const Block3Args& a, hipStream_t stream) {
  RVCX_CHECK(resblock3_ok(a), "resblock3: unsupported shape");
  RVCX_CHECK(a.y != a.x && a.y2 != a.x, "resblock3: in-place operation is a race (halo reads vs neighbours' stores)");
  resblock3_init();
  const B3Cfg& c = *find_b3(a.C);
  static const int ncu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return std::max(8, n);
  }();
  const int ntiles = cdiv(a.T, c.n1 - 2 * kHalo3);
  dim3 grid(std::min(ntiles, ncu), 1, a.B);
  hipLaunchKernelGGL(c.kern, grid, dim3(c.threads), c.lds, stream, a);
  RVCX_HIP(hipGetLastError());
}

}  // namespace rvcx
