// One step of ResBlock1 (rvc/lib/algorithm/residuals.py:45-53) as ONE kernel:
//     xt = c1(leaky_relu(x));  xt = c2(leaky_relu(xt));  x = xt + x
// c1: k taps, dilation d; c2: k taps, dilation 1; both C -> C channels, "same" zero padding.
//
// Unfused, the pair moves five activation tensors through HBM (read x, write xt, read xt, read x again as the
// residual, write y) and re-stages the same input tile once per 32-channel output block.  Here a workgroup owns
// ALL C channels of a tile of positions: the c1 output tile never leaves the CU -- it is written, already in the
// fp16 hi/lo split form and element order the MFMA B operand wants, into LDS and consumed from there by c2; the
// residual is re-read from the (L2-hot) x tile the workgroup just staged.  HBM traffic per pair: read x, write y.
//
// Arithmetic is exactly conv_h3_kernel's (same split, same MFMA order: chunks of 16 input channels ascending, taps
// ascending, three MFMAs {S wh . xh, wh . S xl, S wl . xh} per block), so the fused step is bit-identical to the two
// launches it replaces (tests/test_gpu_conv.py::test_fused_resblock_pair_equals_two_launches).
//
// Tile: N1 = 32 NT positions of the c1 output (= the c2 input tile incl. its (k-1)/2 halo on both sides); the
// workgroup stores BN = N1 - (k-1) output positions.  LDS: Y1[C/16][op][h][N1 + 16] + As[KKT][op][h][C] +
// Bs[op][h][N1 + 64], 16-byte elements (8 halves = the k-slice one lane feeds to one MFMA).
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <type_traits>

#include "conv.h"
#include "conv_device.h"
#include "h3_device.h"

namespace rvcx {

namespace {

constexpr int kPairHalo = 64;    // (k - 1) d <= 50 for k <= 11, d <= 5
constexpr int kPairPadY = 16;    // c2 reads up to k - 1 <= 10 columns past N1 (for output columns that are discarded)
#ifndef RVCX_PAIR_SCHED_FENCE
#define RVCX_PAIR_SCHED_FENCE 0
#endif
constexpr bool kSchedFence = RVCX_PAIR_SCHED_FENCE != 0;
#ifndef RVCX_PAIR_PIPE
#define RVCX_PAIR_PIPE 1
#endif
constexpr bool kPipe = RVCX_PAIR_PIPE != 0;      // pin the fragment-read / MFMA interleave (sched_group_barrier); 0: A/B builds
// Timing ablations (tools/ablate_pair.sh builds one library per value; results are garbage, only the clock counts):
//   1 c2: no weight commit, no barriers      2 c2 epilogue: no residual loads / stores      4 no fragment reads, no MFMAs
//   8 c1: no input conversion at commit     16 c1: no weight commit, one barrier per chunk
#ifndef RVCX_PAIR_ABL
#define RVCX_PAIR_ABL 0
#endif
constexpr int kAbl = RVCX_PAIR_ABL;

// K: taps (compile time: the k-loop is straight-line code, fragment reads of slot s+1 are issued ahead of the MFMAs
// of slot s).  A stage = NCS chunks of 16 input channels x a group of <= KKT taps ("slots"); NCS > 1 only with
// KKT == K (small kernels: more MFMA work between two barriers).
// TRIM: halo and pad sized for THIS kernel size (dilation <= 5) instead of the common 64 / 16 columns: the small-tile
// variants that put two workgroups on a CU need every kilobyte.
constexpr int pair_halo(int K, bool trim) { return trim ? (((K - 1) * 5 + 1) & ~1) : kPairHalo; }
constexpr int pair_pady(int K, bool trim) { return trim ? K - 1 : kPairPadY; }

// RESPF: when the residual x of the output tile is fetched.  0: in the epilogue, eight values at a time (four dependent
// load -> store round trips per wave: ~5 us of every workgroup, 3 ms of a 30 s clip -- round-3 ablation); 1: all of a
// wave's values in one go right behind the c2 loop; 2: ahead of the c2 loop, so that the loads fly under its MFMAs
// (where 16 WM WN more registers fit).  3: the WIDE epilogue -- every 32 x 32 accumulator tile goes through a 4.5 KB LDS
// tile of its wave and comes back transposed, a lane then owns 4 consecutive positions of one channel: residual loads and
// output stores are 16 bytes per lane, a quarter of the memory instructions (the dword stores were store-ISSUE bound).
// Needs 16-byte aligned rows (cs, T multiples of 4); the tile keeps a multiple of 4 output positions.
// OVL: the c1 input tile Bs OVERLAYS the Y1 region (Y1 is written only by the c1 epilogue, when nobody reads Bs any more
// -- one extra barrier per tile): the LDS then holds max(Y1, Bs) + As, which is what lets a 256-position tile of all 128
// channels fit, i.e. 8 waves of 64 x 64 (WM = WN = 2: eight fragment reads per twelve MFMAs instead of six per six).
// Round 4 (tools/mfma_peak.hip): ds_read_b128 delivers ~115 B/clk/CU, so at one fragment read per MFMA the LDS, not the
// matrix pipe, bounds the k-loop (measured: 1.9 us per stage against 1.2 us of MFMA time).
template <int C, int NT, int WR, int WC, int K, int KKT, int NCS, bool TRIM = false, int RESPF = 0, bool OVL = false>
__global__ __launch_bounds__(64 * WR * WC) void resblock_pair_kernel(const PairArgs a) {
  constexpr int THREADS = 64 * WR * WC;
  constexpr int N1 = 32 * NT, N1P = N1 + pair_pady(K, TRIM), WROW = N1 + pair_halo(K, TRIM);
  constexpr int WM = (C / 32) / WR, WN = NT / WC;
  constexpr int NCHUNK = C / 16;
  constexpr int NG = (K + KKT - 1) / KKT;                 // tap groups per chunk set
  constexpr int SL = NCS * KKT;                           // weight slots resident per stage
  constexpr int A_ELEMS = SL * 4 * C, NA = (A_ELEMS + THREADS - 1) / THREADS;
  constexpr int B_TASKS = NCS * 2 * WROW, NBT = (B_TASKS + THREADS - 1) / THREADS;
  constexpr int BN_OUT = RESPF == 3 ? ((N1 - (K - 1)) & ~3) : N1 - (K - 1), H2 = (K - 1) / 2;
  static_assert(WM >= 1 && WN >= 1 && WM * WR * 32 == C && WN * WC == NT, "bad tile");
  static_assert(NCS == 1 || KKT == K, "several chunks per stage only with all taps resident");
  static_assert(NCHUNK % NCS == 0, "chunk sets must tile the channels");
  extern __shared__ uint4 lds[];
  constexpr int Y1_ELEMS = NCHUNK * 4 * N1P, BS_ELEMS = NCS * 4 * WROW;
  uint4* Y1 = lds;                              // [chunk][op][h][N1P]
  uint4* As = Y1 + (OVL && BS_ELEMS > Y1_ELEMS ? BS_ELEMS : Y1_ELEMS);   // [slot][op][h][C]
  uint4* Bs = OVL ? Y1 : As + A_ELEMS;          // [cl][op][h][WROW]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WC, wc = wave % WC;
  const int i = lane & 31, h = lane >> 5;
  const int b = blockIdx.z;
  const int pad1 = (K - 1) * a.dil / 2;
  // XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs (block b -> XCD b % 8), each with its own L2:
  // with tile = blockIdx.x neighbouring tiles always sit on DIFFERENT L2s, so the halo columns two tiles share are
  // fetched from HBM twice and the cache line that straddles a tile boundary (BN_OUT is not a multiple of 32
  // positions) is written back half-filled by two L2s.  Here XCD x works through a contiguous range of tiles.
  int tile = blockIdx.x;
  if (a.xcd_order) {
    const int nt = gridDim.x, xcd = tile & 7, idx = tile >> 3, q = nt >> 3, r = nt & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int n0 = tile * BN_OUT;
  const int len = a.lens ? a.lens[b] : a.T;
  const int wuse = N1 + (K - 1) * a.dil;        // input columns the tile really needs
  const int in_base = n0 - H2 - pad1;           // position of input-tile column 0
  const float slope = a.slope;
  const H3Rsrc xr = h3_rsrc(a.x + (long)b * a.bs, C * a.cs * 4);
  const H3Rsrc w1r = h3_rsrc(a.w1, K * NCHUNK * 4 * C * 16);
  const H3Rsrc w2r = h3_rsrc(a.w2, K * NCHUNK * 4 * C * 16);
  const int xrow = a.cs * 4;
  constexpr int slab = 4 * C * 16;              // bytes of one (tap, chunk) weight slab

  // phase stamps of this workgroup (100 MHz real-time counter): 0 start, 1 first stage committed (inputs arrived),
  // 2 c1 loop done, 3 Y1 written, 4 c2 loop done, 5 end; 6 = tile, 7 = XCC id
  auto stamp = [&](int k) {
    if (a.trace && tid == 0) a.trace[((long)b * gridDim.x + blockIdx.x) * 8 + k] = (long long)wall_clock64();
  };
  stamp(0);
  if (a.trace && tid == 0) {
    a.trace[((long)b * gridDim.x + blockIdx.x) * 8 + 6] = tile;
    unsigned xcc = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
#endif
    a.trace[((long)b * gridDim.x + blockIdx.x) * 8 + 7] = xcc & 0xf;
  }
  f32x16 acc[WM][WN];
  auto zero_acc = [&]() {
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
      for (int n = 0; n < WN; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  };

  // stage-invariant addressing: weight element e = tid + THREADS j -> (slot = cl * KKT + kkl, op, h, co)
  int a_cl[NA], a_kkl[NA], a_off[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int e = tid + THREADS * j;
    const int co = e % C, rest = e / C;          // rest = (slot*2 + op)*2 + h
    const int slot = rest / 4;
    a_cl[j] = slot / KKT;
    a_kkl[j] = slot % KKT;
    a_off[j] = e < A_ELEMS ? ((rest % 4) * C + co) * 16 : kH3Oob;
  }
  int b_off[NBT], b_row[NBT];                    // input task t -> (cl, h, p): 8 channels at one position
#pragma unroll
  for (int j = 0; j < NBT; ++j) {
    const int t = tid + THREADS * j;
    const int ch2 = t / WROW, p = t - ch2 * WROW;   // ch2 = cl*2 + h
    const int pos = in_base + p;
    b_row[j] = (ch2 >> 1) * 16 + (ch2 & 1) * 8;
    b_off[j] = (t < B_TASKS && p < wuse && pos >= 0 && pos < len) ? pos * 4 : kH3Oob;
  }

  uint4 ra[NA];
  float rb[NBT][8];
  bool ovf = false;
  auto fetch_b = [&](int chunk0) {
#pragma unroll
    for (int j = 0; j < NBT; ++j) {
      const int row0 = (chunk0 * 16 + b_row[j]) * xrow;
#pragma unroll
      for (int q = 0; q < 8; ++q) rb[j][q] = h3_load1(xr, b_off[j] == kH3Oob ? kH3Oob : row0 + q * xrow + b_off[j]);
    }
  };
  auto fetch_a = [&](const H3Rsrc& wr_, int chunk0, int kk0) {
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int kk = kk0 + a_kkl[j];
      const bool ok = a_off[j] != kH3Oob && kk < K;
      ra[j] = h3_load4(wr_, ok ? (kk * NCHUNK + chunk0 + a_cl[j]) * slab + a_off[j] : kH3Oob);
    }
  };
  auto commit_a = [&]() {
#pragma unroll
    for (int j = 0; j < NA; ++j)
      if (NA * THREADS == A_ELEMS || tid + THREADS * j < A_ELEMS) As[tid + THREADS * j] = ra[j];
  };
  auto commit_b = [&]() {
#pragma unroll
    for (int j = 0; j < NBT; ++j) {
      const int t = tid + THREADS * j;
      if (NBT * THREADS == B_TASKS || t < B_TASKS) {
        half8 hi, lo;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          float v = rb[j][q];
          if (kAbl & 8) {
            hi[q] = (_Float16)0.f;
            lo[q] = __builtin_bit_cast(_Float16, (unsigned short)(__builtin_bit_cast(unsigned, v) & 1));
            continue;
          }
          v = v > 0.f ? v : v * slope;                 // leaky_relu ahead of c1
          ovf |= !(fabsf(v) < kH3ActLimit);
          const _Float16 vh = (_Float16)v;
          hi[q] = vh;
          lo[q] = (_Float16)((v - (float)vh) * kH3Scale);
        }
        const int ch2 = t / WROW, p = t - ch2 * WROW;
        const int cl = ch2 >> 1, hh = ch2 & 1;
        Bs[((cl * 2 + 0) * 2 + hh) * WROW + p] = __builtin_bit_cast(uint4, hi);
        Bs[((cl * 2 + 1) * 2 + hh) * WROW + p] = __builtin_bit_cast(uint4, lo);
      }
    }
  };
  // The k-steps of one stage: slots (cl, kkl), TAPS taps per chunk.  Bt: tile of chunk cl at Bt + cl * cl_pitch,
  // [op][h][pitch]; tap kk of the group starting at kk0 reads column + (kk0 + kkl) * tap_step.  Fragments of the
  // next slot are read before the MFMAs of the current one are issued.
  auto compute = [&](auto taps_tag, const uint4* Bt, int pitch, int cl_pitch, int kk0, int tap_step) {
    constexpr int TAPS = decltype(taps_tag)::value;
    constexpr int NS = NCS * TAPS;
    if (kAbl & 4) return;
    half8 af[2][2][WM], bf[2][2][WN];
    auto load = [&](int buf, int s) {
      const int cl = s / TAPS, kkl = s % TAPS;
      const int tp = (kk0 + kkl) * tap_step;
      const uint4* Bc = Bt + cl * cl_pitch;
#pragma unroll
      for (int m = 0; m < WM; ++m) {
        af[buf][0][m] = __builtin_bit_cast(half8, As[(((cl * KKT + kkl) * 2 + 0) * 2 + h) * C + wr * (WM * 32) + m * 32 + i]);
        af[buf][1][m] = __builtin_bit_cast(half8, As[(((cl * KKT + kkl) * 2 + 1) * 2 + h) * C + wr * (WM * 32) + m * 32 + i]);
      }
#pragma unroll
      for (int op = 0; op < 2; ++op)
#pragma unroll
        for (int n = 0; n < WN; ++n)
          bf[buf][op][n] = __builtin_bit_cast(half8, Bc[(op * 2 + h) * pitch + wc * (WN * 32) + n * 32 + i + tp]);
    };
    load(0, 0);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int cur = s & 1;
      if (s + 1 < NS) load(cur ^ 1, s + 1);
      if (kSchedFence) __builtin_amdgcn_sched_barrier(0);
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
      if (kSchedFence) __builtin_amdgcn_sched_barrier(0);
    }
    // The software pipeline above, PINNED (round 4).  Left alone the scheduler sinks every ds_read next to its first use
    // -- the ISA was {ds_read, s_waitcnt lgkmcnt(0), v_mfma} per MFMA: each wave pays the LDS latency per matrix
    // instruction, the matrix pipe idled more than half of the k-loop.  The groups below order the block as: the first
    // slot's fragment reads, then per slot one MFMA / one read of the NEXT slot alternating; the compiler's counted
    // lgkmcnt waits then find the data already there.  Same instructions, same arithmetic order per accumulator.
    if (kPipe) {
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
  using Full = std::integral_constant<int, KKT>;
  using Last = std::integral_constant<int, K - (NG - 1) * KKT>;
  constexpr float inv = 1.f / kH3Scale;

  // ================================================================ phase 1: Y1 = lrelu(c1(lrelu(x)) + b1)
  zero_acc();
  fetch_b(0);
  fetch_a(w1r, 0, 0);
  // The tap groups of a chunk set are unrolled at compile time: one loop body holds the Full and the Last form of the
  // k-steps, so the accumulators keep their registers across the loop (as a runtime choice inside one loop the two forms
  // were joined by 32 v_mov_b64 per stage behind an s_nop that drained the matrix pipe).
  for (int chunk = 0; chunk < NCHUNK; chunk += NCS) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const bool last_stage = chunk + NCS >= NCHUNK && g == NG - 1;
      if (!(kAbl & 16) || g == 0) __syncthreads();
      if (g == 0) commit_b();
      if (!(kAbl & 16)) commit_a();
      if (!(kAbl & 16) || g == 0) __syncthreads();
      if (chunk == 0 && g == 0) stamp(1);
      if (!last_stage) fetch_a(w1r, g + 1 == NG ? chunk + NCS : chunk, g + 1 == NG ? 0 : (g + 1) * KKT);
      else fetch_a(w2r, 0, 0);                       // first weights of c2 fly during the c1 epilogue
      if (g == 0 && chunk + NCS < NCHUNK) fetch_b(chunk + NCS);
      if (NG > 1 && g == NG - 1) compute(Last{}, Bs, WROW, 4 * WROW, g * KKT, a.dil);
      else compute(Full{}, Bs, WROW, 4 * WROW, g * KKT, a.dil);
    }
  }
  stamp(2);
  if (OVL) __syncthreads();                      // Y1 overlays the input tile: every wave is done reading Bs
  // c1 epilogue: bias, leaky_relu (the one ahead of c2), zero outside the sequence, split, into LDS
#pragma unroll
  for (int m = 0; m < WM; ++m)
#pragma unroll
    for (int n = 0; n < WN; ++n) {
      const int c_t = wr * (WM * 32) + m * 32;
      const int j = wc * (WN * 32) + n * 32 + i;
      const int pos1 = n0 - H2 + j;
      const bool live = pos1 >= 0 && pos1 < len;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        typedef _Float16 half4 __attribute__((ext_vector_type(4)));
        const int cg = c_t + 8 * g;
        half4 hi, lo;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v = acc[m][n][4 * g + q] * inv + (a.b1 ? a.b1[cg + 4 * h + q] : 0.f);
          v = fmaxf(v, v * slope);
          v = live ? v : 0.f;
          ovf |= !(fabsf(v) < kH3ActLimit);
          const _Float16 vh = (_Float16)v;
          hi[q] = vh;
          lo[q] = (_Float16)((v - (float)vh) * kH3Scale);
        }
        char* e_hi = reinterpret_cast<char*>(Y1 + (((cg >> 4) * 2 + 0) * 2 + (g & 1)) * N1P + j) + 8 * h;
        char* e_lo = reinterpret_cast<char*>(Y1 + (((cg >> 4) * 2 + 1) * 2 + (g & 1)) * N1P + j) + 8 * h;
        *reinterpret_cast<half4*>(e_hi) = hi;
        *reinterpret_cast<half4*>(e_lo) = lo;
      }
    }

  if (ovf) report_h3_overflow(a.ovf, a.ovf_layer, a.seq);

  // ================================================================ phase 2: y = c2(Y1) + b2 + x
  float resv[RESPF ? WM : 1][RESPF ? WN : 1][16];
  auto load_res = [&]() {
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
      for (int n = 0; n < WN; ++n) {
        const int col = wc * (WN * 32) + n * 32 + i;
        const int pos = n0 + col;
        const bool ok = col < BN_OUT && pos < a.T;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = wr * (WM * 32) + m * 32 + 4 * h + (r & 3) + 8 * (r >> 2);
          if (RESPF == 2) {
            // Issued HERE, not where the compiler would like them (it sinks a plain load across the whole c2 loop to its
            // first use): inline asm stays put.  The compiler does not count these loads; they are older than every
            // load of the c2 loop, loads return in order, so its counted waits stay correct (they only wait longer),
            // and the epilogue waits vmcnt(0) by hand before the first use.
            const float* p = a.x + (long)b * a.bs + (ok ? (long)co * a.cs + pos : 0);
            float v;
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
#else
            v = *p;
#endif
            resv[RESPF ? m : 0][RESPF ? n : 0][r] = v;
          } else {
            resv[RESPF ? m : 0][RESPF ? n : 0][r] = h3_load1(xr, ok ? (co * a.cs + pos) * 4 : kH3Oob);
          }
        }
      }
  };
  if (RESPF == 2 && !(kAbl & 2)) load_res();
  zero_acc();
  stamp(3);
  if (kAbl & 1) __syncthreads();
  for (int chunk = 0; chunk < NCHUNK; chunk += NCS) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const bool last_stage = chunk + NCS >= NCHUNK && g == NG - 1;
      if (!(kAbl & 1)) {
        __syncthreads();
        commit_a();
        __syncthreads();
      }
      if (!last_stage && !(kAbl & 1)) fetch_a(w2r, g + 1 == NG ? chunk + NCS : chunk, g + 1 == NG ? 0 : (g + 1) * KKT);
      if (NG > 1 && g == NG - 1) compute(Last{}, Y1 + chunk * 4 * N1P, N1P, 4 * N1P, g * KKT, 1);
      else compute(Full{}, Y1 + chunk * 4 * N1P, N1P, 4 * N1P, g * KKT, 1);
    }
  }
  stamp(4);
  if constexpr (RESPF == 3) {
    constexpr int P = 36;                              // floats per staged row: 16-byte aligned, rows 4 banks apart
    static_assert(THREADS / 64 * 32 * P * 4 <= NCHUNK * 4 * N1P * 16, "staging tiles must fit the Y1 region");
    __syncthreads();                                   // every wave is done reading Y1: the staging tiles overlay it
    float* stg = reinterpret_cast<float*>(lds) + wave * (32 * P);
    const int lr = lane >> 3, lc = (lane & 7) * 4;
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
      for (int n = 0; n < WN; ++n) {
#pragma unroll
        for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * h) * P + i] = acc[m][n][r] * inv;
        __builtin_amdgcn_wave_barrier();               // same wave, in-order LDS queue: ordering for the compiler only
        const int col0 = wc * (WN * 32) + n * 32 + lc;
        const int pos0 = n0 + col0;
        const bool ok = col0 < BN_OUT && pos0 < a.T && !((kAbl & 2) && acc[m][n][0] != 12345.678f);
        float4 v[4], rv[4], pv[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int co = wr * (WM * 32) + m * 32 + lr + 8 * p;
          const long off = (long)b * a.bs + (long)co * a.cs + pos0;
          v[p] = *reinterpret_cast<const float4*>(stg + (lr + 8 * p) * P + lc);
          rv[p] = ok ? *reinterpret_cast<const float4*>(a.x + off) : make_float4(0.f, 0.f, 0.f, 0.f);
          pv[p] = (ok && a.acc2_mode != ACC2_NONE && a.acc2_mode != ACC2_SET) ? *reinterpret_cast<const float4*>(a.y2 + off)
                                                                              : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __builtin_amdgcn_wave_barrier();
        if (ok) {
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const int co = wr * (WM * 32) + m * 32 + lr + 8 * p;
            const long off = (long)b * a.bs + (long)co * a.cs + pos0;
            const float bb = a.b2 ? a.b2[co] : 0.f;
            float o[4] = {v[p].x + bb + rv[p].x, v[p].y + bb + rv[p].y, v[p].z + bb + rv[p].z, v[p].w + bb + rv[p].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = pos0 + q < len ? o[q] : 0.f;
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
    stamp(5);
    return;
  }
  // c2 epilogue: bias, residual (x itself: L2-hot, this workgroup staged it a moment ago), length mask, store
  if (RESPF == 1 && !(kAbl & 2)) load_res();
#if defined(__HIP_DEVICE_COMPILE__)
  if (RESPF == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#pragma unroll
  for (int m = 0; m < WM; ++m)
#pragma unroll
    for (int n = 0; n < WN; ++n) {
      const int col = wc * (WN * 32) + n * 32 + i;
      const int pos = n0 + col;
      if (col >= BN_OUT || pos >= a.T) continue;
      if ((kAbl & 2) && acc[m][n][0] != 12345.678f) continue;
      const bool live = pos < len;
      const int co_base = wr * (WM * 32) + m * 32 + 4 * h;
      const float* xb = a.x + (long)b * a.bs + pos;
      float* yb = a.y ? a.y + (long)b * a.bs + pos : nullptr;
      float* y2b = a.acc2_mode != ACC2_NONE ? a.y2 + (long)b * a.bs + pos : nullptr;
#pragma unroll
      for (int r0 = 0; r0 < 16; r0 += 8) {
        float bv[8], rv[8], pv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int co = co_base + ((r0 + q) & 3) + 8 * ((r0 + q) >> 2);
          bv[q] = a.b2 ? a.b2[co] : 0.f;
          rv[q] = RESPF ? resv[RESPF ? m : 0][RESPF ? n : 0][r0 + q] : xb[(long)co * a.cs];
          pv[q] = (y2b && a.acc2_mode != ACC2_SET) ? y2b[(long)co * a.cs] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int co = co_base + ((r0 + q) & 3) + 8 * ((r0 + q) >> 2);
          float v = acc[m][n][r0 + q] * inv + bv[q];
          v = v + rv[q];
          v = live ? v : 0.f;
          if (yb) yb[(long)co * a.cs] = v;
          if (y2b) {
            float* p2 = y2b + (long)co * a.cs;
            if (a.acc2_mode == ACC2_SET) *p2 = v;
            else if (a.acc2_mode == ACC2_ADD) *p2 = pv[q] + v;
            else *p2 = (pv[q] + v) / a.acc2_div;
          }
        }
      }
    }
}


// ====================================================================================================================
// Round 4: the A-DIRECT form of the fused step.
//
// Per-workgroup phase stamps of the kernel above (tools/pair_trace.py) showed the two k-loops are 91 % of a tile's
// life and run at ~60 % matrix-pipe duty: every stage pays two workgroup barriers around the weight commit (drain,
// commit, refill: ~0.6 us of a 1.9 us stage), and with one fragment read per MFMA the LDS (~115 B/clk/CU for
// ds_read_b128, tools/mfma_peak.hip) is the second bound right behind it.  Here the weight fragments never touch the
// LDS: the fp16 hi/lo weight image is already stored in MFMA A-fragment order (16 bytes per lane, 512 contiguous
// bytes per half wave), so every wave loads ITS fragments straight from L2 into registers, one tap group ahead
// (double-buffered register sets, ~1 us of prefetch distance).  What is left in LDS is the activation side:
//   * phase 2 (c2 over Y1) has NO barrier at all -- Y1 is read-only, waves run free;
//   * phase 1 (c1) keeps one barrier per 16-channel chunk: the converted input tile is double-buffered, the
//     conversion of chunk c + 1 happens in the shadow of chunk c's MFMAs;
//   * LDS reads per MFMA: 4 per 6 (32 x 64 wave tiles) or 4 per 12 (64 x 64) instead of 6 per 6.
// Same split, same k-order (chunks ascending, taps ascending, three MFMAs per block): bit-identical to the kernel
// above and to the two conv_h3 launches.
// G: taps per A register set.  OVL: the input tiles overlay Y1 (see above).
template <int G0, int NGRP, typename F>
__device__ __forceinline__ void for_each_group(F&& f) {      // f(integral_constant<int, g>) for g = G0 .. NGRP - 1, unrolled
  if constexpr (G0 < NGRP) {
    f(std::integral_constant<int, G0>{});
    for_each_group<G0 + 1, NGRP>(f);
  }
}

template <int C, int NT, int WR, int WC, int K, int G, bool OVL>
__global__ __launch_bounds__(64 * WR * WC) void resblock_pair_adir_kernel(const PairArgs a) {
  constexpr int THREADS = 64 * WR * WC;
  constexpr int N1 = 32 * NT, N1P = N1 + kPairPadY, WROW = N1 + kPairHalo;
  constexpr int WM = (C / 32) / WR, WN = NT / WC;
  constexpr int NCHUNK = C / 16, NG = (K + G - 1) / G;
  constexpr int B_TASKS = 2 * WROW, NBT = (B_TASKS + THREADS - 1) / THREADS;
  constexpr int BN_OUT = (N1 - (K - 1)) & ~3, H2 = (K - 1) / 2;
  constexpr int Y1_ELEMS = NCHUNK * 4 * N1P, BS_ELEMS = 4 * WROW;
  static_assert(WM >= 1 && WN >= 1 && WM * WR * 32 == C && WN * WC == NT, "bad tile");
  static_assert(NCHUNK % 2 == 0, "chunks are processed in pairs");
  static_assert(!OVL || 2 * BS_ELEMS <= Y1_ELEMS, "the two input tiles must fit the Y1 region");
  extern __shared__ uint4 lds[];
  uint4* Y1 = lds;                                  // [chunk][op][h][N1P]
  uint4* Bs = OVL ? Y1 : Y1 + Y1_ELEMS;             // [buf][op][h][WROW]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WC, wc = wave % WC;
  const int i = lane & 31, h = lane >> 5;
  const int b = blockIdx.z;
  const int pad1 = (K - 1) * a.dil / 2;
  int tile = blockIdx.x;
  if (a.xcd_order) {
    const int nt = gridDim.x, xcd = tile & 7, idx = tile >> 3, q = nt >> 3, r = nt & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int n0 = tile * BN_OUT;
  const int len = a.lens ? a.lens[b] : a.T;
  const int wuse = N1 + (K - 1) * a.dil;
  const int in_base = n0 - H2 - pad1;
  const float slope = a.slope;
  const H3Rsrc xr = h3_rsrc(a.x + (long)b * a.bs, C * a.cs * 4);
  const H3Rsrc w1r = h3_rsrc(a.w1, K * NCHUNK * 4 * C * 16);
  const H3Rsrc w2r = h3_rsrc(a.w2, K * NCHUNK * 4 * C * 16);
  const int xrow = a.cs * 4;
  constexpr int slab = 4 * C * 16;
  constexpr float inv = 1.f / kH3Scale;
  auto stamp = [&](int k) {
    if (a.trace && tid == 0) a.trace[((long)b * gridDim.x + blockIdx.x) * 8 + k] = (long long)wall_clock64();
  };
  stamp(0);
  if (a.trace && tid == 0) a.trace[((long)b * gridDim.x + blockIdx.x) * 8 + 6] = tile;

  f32x16 acc[WM][WN];
  auto zero_acc = [&]() {
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
      for (int n = 0; n < WN; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  };

  // ---- input tile staging: task t -> (h, p): 8 channels of one position (fp32 -> split fp16 at commit)
  int b_off[NBT], b_row[NBT];
#pragma unroll
  for (int j = 0; j < NBT; ++j) {
    const int t = tid + THREADS * j;
    const int hh = t / WROW, p = t - hh * WROW;
    const int pos = in_base + p;
    b_row[j] = hh * 8;
    b_off[j] = (t < B_TASKS && p < wuse && pos >= 0 && pos < len) ? pos * 4 : kH3Oob;
  }
  float rb[NBT][8];
  bool ovf = false;
  auto fetch_b = [&](int chunk) {
#pragma unroll
    for (int j = 0; j < NBT; ++j) {
      const int row0 = (chunk * 16 + b_row[j]) * xrow;
#pragma unroll
      for (int q = 0; q < 8; ++q) rb[j][q] = h3_load1(xr, b_off[j] == kH3Oob ? kH3Oob : row0 + q * xrow + b_off[j]);
    }
  };
  auto commit_b = [&](int buf) {
    uint4* Bd = Bs + buf * BS_ELEMS;
#pragma unroll
    for (int j = 0; j < NBT; ++j) {
      const int t = tid + THREADS * j;
      if (NBT * THREADS == B_TASKS || t < B_TASKS) {
        half8 hi, lo;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          float v = rb[j][q];
          v = v > 0.f ? v : v * slope;                 // leaky_relu ahead of c1
          ovf |= !(fabsf(v) < kH3ActLimit);
          const _Float16 vh = (_Float16)v;
          hi[q] = vh;
          lo[q] = (_Float16)((v - (float)vh) * kH3Scale);
        }
        const int hh = t / WROW, p = t - hh * WROW;
        Bd[(0 * 2 + hh) * WROW + p] = __builtin_bit_cast(uint4, hi);
        Bd[(1 * 2 + hh) * WROW + p] = __builtin_bit_cast(uint4, lo);
      }
    }
  };

  // ---- weight fragments: two register sets of G taps x {S wh, S wl} x WM row blocks, loaded straight from the image
  half8 aq[2][G][2][WM];
  const int a_lane = (h * C + wr * (WM * 32) + i) * 16;
  auto taps_of = [](int g) { return (NG > 1 && g == NG - 1) ? K - (NG - 1) * G : G; };
  auto load_aset = [&](auto set_tag, const H3Rsrc& wres, int chunk, int g) {
    constexpr int SET = decltype(set_tag)::value;
    const int kk0 = g * G;
#pragma unroll
    for (int kkl = 0; kkl < G; ++kkl)
      if (kkl < taps_of(g)) {
#pragma unroll
        for (int op = 0; op < 2; ++op)
#pragma unroll
          for (int m = 0; m < WM; ++m)
            aq[SET][kkl][op][m] = __builtin_bit_cast(
                half8, h3_load4s(wres, a_lane, ((kk0 + kkl) * NCHUNK + chunk) * slab + op * 2 * C * 16 + m * 512));
      }
  };
  // the k-steps of one stage: TAPS taps of one chunk, A from register set SET, B from LDS one slot ahead
  auto compute = [&](auto set_tag, auto taps_tag, const uint4* Bt, int pitch, int kk0, int tap_step) {
    constexpr int SET = decltype(set_tag)::value, TAPS = decltype(taps_tag)::value;
    half8 bf[2][2][WN];
    auto loadb = [&](int buf, int s) {
      const int tp = (kk0 + s) * tap_step;
#pragma unroll
      for (int op = 0; op < 2; ++op)
#pragma unroll
        for (int n = 0; n < WN; ++n)
          bf[buf][op][n] = __builtin_bit_cast(half8, Bt[(op * 2 + h) * pitch + wc * (WN * 32) + n * 32 + i + tp]);
    };
    loadb(0, 0);
#pragma unroll
    for (int s = 0; s < TAPS; ++s) {
      const int cur = s & 1;
      if (s + 1 < TAPS) loadb(cur ^ 1, s + 1);
#pragma unroll
      for (int m = 0; m < WM; ++m) {
        const half8 wh = aq[SET][s][0][m] * (_Float16)(1.f / kH3Scale);
#pragma unroll
        for (int n = 0; n < WN; ++n) {
          acc[m][n] = h3_mfma(aq[SET][s][0][m], bf[cur][0][n], acc[m][n]);   // (S wh) xh
          acc[m][n] = h3_mfma(wh, bf[cur][1][n], acc[m][n]);                  // wh (S xl)
          acc[m][n] = h3_mfma(aq[SET][s][1][m], bf[cur][0][n], acc[m][n]);   // (S wl) xh
        }
      }
    }
    if (kPipe) {      // pin the software pipeline: first slot's reads, then MFMAs / next slot's reads alternating
      constexpr int R = 2 * WN, M = 3 * WM * WN;
      __builtin_amdgcn_sched_group_barrier(0x100, R, 0);
#pragma unroll
      for (int s = 0; s < TAPS; ++s) {
#pragma unroll
        for (int j = 0; j < M; ++j) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (s + 1 < TAPS && j < R) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  // one stage of a phase: prefetch the next stage's weights into the other register set, then multiply
  auto stage = [&](auto cc_tag, auto g_tag, int chunk, const H3Rsrc& wres, bool have_next_chunk, const H3Rsrc* wres_after,
                   const uint4* Bt, int pitch, int tap_step) {
    constexpr int CC = decltype(cc_tag)::value, GG = decltype(g_tag)::value;
    constexpr int P = (CC * NG + GG) & 1;
    using Cur = std::integral_constant<int, P>;
    using Nxt = std::integral_constant<int, P ^ 1>;
    if constexpr (GG + 1 < NG) {
      load_aset(Nxt{}, wres, chunk, GG + 1);
    } else {
      if (have_next_chunk) load_aset(Nxt{}, wres, chunk + 1, 0);
      else if (wres_after) load_aset(Nxt{}, *wres_after, 0, 0);
    }
    constexpr int TAPS = (NG > 1 && GG == NG - 1) ? K - (NG - 1) * G : G;
    compute(Cur{}, std::integral_constant<int, TAPS>{}, Bt, pitch, GG * G, tap_step);
  };
  // compile-time loop over the tap groups of a chunk
  auto for_groups = [&](auto&& f) { for_each_group<0, NG>(f); };

  // ================================================================ phase 1: Y1 = lrelu(c1(lrelu(x)) + b1)
  zero_acc();
  fetch_b(0);
  load_aset(S0{}, w1r, 0, 0);
  commit_b(0);
  __syncthreads();
  stamp(1);
  if (NCHUNK > 1) fetch_b(1);
  for (int cp = 0; cp < NCHUNK; cp += 2) {
    // chunk cp (input tile buffer 0) then chunk cp + 1 (buffer 1)
    for_groups([&](auto g_tag) {
      constexpr int GG = decltype(g_tag)::value;
      stage(S0{}, g_tag, cp, w1r, true, nullptr, Bs, WROW, a.dil);
      if constexpr (GG == 0) {             // chunk cp + 1 into the other buffer, in the shadow of this chunk's MFMAs
        commit_b(1);
        if (cp + 2 < NCHUNK) fetch_b(cp + 2);
      }
    });
    __syncthreads();      // buffer 1 is complete; everyone is done with buffer 0
    for_groups([&](auto g_tag) {
      constexpr int GG = decltype(g_tag)::value;
      stage(S1{}, g_tag, cp + 1, w1r, cp + 2 < NCHUNK, &w2r, Bs + BS_ELEMS, WROW, a.dil);
      if constexpr (GG == 0) {
        if (cp + 2 < NCHUNK) {
          commit_b(0);
          if (cp + 3 < NCHUNK) fetch_b(cp + 3);
        }
      }
    });
    if (cp + 2 < NCHUNK) __syncthreads();
  }
  stamp(2);
  if (OVL) __syncthreads();                // Y1 overlays the input tiles: every wave is done reading them
  // c1 epilogue: bias, leaky_relu (the one ahead of c2), zero outside the sequence, split, into LDS
#pragma unroll
  for (int m = 0; m < WM; ++m)
#pragma unroll
    for (int n = 0; n < WN; ++n) {
      const int c_t = wr * (WM * 32) + m * 32;
      const int j = wc * (WN * 32) + n * 32 + i;
      const int pos1 = n0 - H2 + j;
      const bool live = pos1 >= 0 && pos1 < len;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        typedef _Float16 half4 __attribute__((ext_vector_type(4)));
        const int cg = c_t + 8 * g;
        half4 hi, lo;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v = acc[m][n][4 * g + q] * inv + (a.b1 ? a.b1[cg + 4 * h + q] : 0.f);
          v = fmaxf(v, v * slope);
          v = live ? v : 0.f;
          ovf |= !(fabsf(v) < kH3ActLimit);
          const _Float16 vh = (_Float16)v;
          hi[q] = vh;
          lo[q] = (_Float16)((v - (float)vh) * kH3Scale);
        }
        char* e_hi = reinterpret_cast<char*>(Y1 + (((cg >> 4) * 2 + 0) * 2 + (g & 1)) * N1P + j) + 8 * h;
        char* e_lo = reinterpret_cast<char*>(Y1 + (((cg >> 4) * 2 + 1) * 2 + (g & 1)) * N1P + j) + 8 * h;
        *reinterpret_cast<half4*>(e_hi) = hi;
        *reinterpret_cast<half4*>(e_lo) = lo;
      }
    }
  if (ovf) report_h3_overflow(a.ovf, a.ovf_layer, a.seq);
  zero_acc();
  __syncthreads();                          // Y1 is complete
  stamp(3);

  // ================================================================ phase 2: y = c2(Y1) + b2 + x  -- no barrier in the loop
  for (int cp = 0; cp < NCHUNK; cp += 2) {
    for_groups([&](auto g_tag) { stage(S0{}, g_tag, cp, w2r, true, nullptr, Y1 + cp * 4 * N1P, N1P, 1); });
    for_groups([&](auto g_tag) { stage(S1{}, g_tag, cp + 1, w2r, cp + 2 < NCHUNK, nullptr, Y1 + (cp + 1) * 4 * N1P, N1P, 1); });
  }
  stamp(4);
  // ---- wide epilogue (as above): 32 x 32 tiles transposed through a per-wave LDS tile, 16 bytes per lane
  {
    constexpr int P = 36;
    static_assert(THREADS / 64 * 32 * P * 4 <= Y1_ELEMS * 16, "staging tiles must fit the Y1 region");
    __syncthreads();                                   // every wave is done reading Y1: the staging tiles overlay it
    float* stg = reinterpret_cast<float*>(lds) + wave * (32 * P);
    const int lr = lane >> 3, lc = (lane & 7) * 4;
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
      for (int n = 0; n < WN; ++n) {
#pragma unroll
        for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * h) * P + i] = acc[m][n][r] * inv;
        __builtin_amdgcn_wave_barrier();
        const int col0 = wc * (WN * 32) + n * 32 + lc;
        const int pos0 = n0 + col0;
        const bool ok = col0 < BN_OUT && pos0 < a.T;
        float4 v[4], rv[4], pv[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int co = wr * (WM * 32) + m * 32 + lr + 8 * p;
          const long off = (long)b * a.bs + (long)co * a.cs + pos0;
          v[p] = *reinterpret_cast<const float4*>(stg + (lr + 8 * p) * P + lc);
          rv[p] = ok ? *reinterpret_cast<const float4*>(a.x + off) : make_float4(0.f, 0.f, 0.f, 0.f);
          pv[p] = (ok && a.acc2_mode != ACC2_NONE && a.acc2_mode != ACC2_SET) ? *reinterpret_cast<const float4*>(a.y2 + off)
                                                                              : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __builtin_amdgcn_wave_barrier();
        if (ok) {
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const int co = wr * (WM * 32) + m * 32 + lr + 8 * p;
            const long off = (long)b * a.bs + (long)co * a.cs + pos0;
            const float bb = a.b2 ? a.b2[co] : 0.f;
            float o[4] = {v[p].x + bb + rv[p].x, v[p].y + bb + rv[p].y, v[p].z + bb + rv[p].z, v[p].w + bb + rv[p].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = pos0 + q < len ? o[q] : 0.f;
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
  stamp(5);
}


// ====================================================================================================================
// Round 4, third form: SPLIT-PHASE staging.  The weights stay LDS-staged (shared by the waves of the CU: the A-direct
// form showed the L2 cannot feed every wave separately) but the two s_barriers per stage are gone from the k-loops.
// The weight tile is a ring of two stage buffers; a wave commits its share of stage s + 1 in the MIDDLE of its own
// stage s and signals through two monotonic LDS counters:
//     ready : + 1 per wave when its share of a stage is in LDS      (a stage is readable at  ready >= W * (index + 1))
//     done  : + 1 per wave when it has issued the last MFMA of a stage (a buffer is writable at done >= W * index)
// What a wave waits for happened about half a stage ago, so waves drift apart by up to half a stage instead of meeting
// at a barrier twice per stage: while one converts or commits, the others keep the matrix pipe fed.  The c1 input tile
// is double-buffered in the (not yet written) Y1 region and follows the same counters.  Same arithmetic order.
template <int C, int NT, int WR, int WC, int K, int G>
__global__ __launch_bounds__(64 * WR * WC) void resblock_pair_sp_kernel(const PairArgs a) {
  constexpr int THREADS = 64 * WR * WC, NW = WR * WC;
  constexpr int N1 = 32 * NT, N1P = N1 + kPairPadY, WROW = N1 + kPairHalo;
  constexpr int WM = (C / 32) / WR, WN = NT / WC;
  constexpr int NCHUNK = C / 16, NG = (K + G - 1) / G, NS = NCHUNK * NG;
  constexpr int A_ELEMS = G * 4 * C, NA = (A_ELEMS + THREADS - 1) / THREADS;
  constexpr int B_TASKS = 2 * WROW, NBT = (B_TASKS + THREADS - 1) / THREADS;
  constexpr int BN_OUT = (N1 - (K - 1)) & ~3, H2 = (K - 1) / 2;
  constexpr int Y1_ELEMS = NCHUNK * 4 * N1P, BS_ELEMS = 4 * WROW;
  static_assert(WM >= 1 && WN >= 1 && WM * WR * 32 == C && WN * WC == NT, "bad tile");
  constexpr int R0_ELEMS = Y1_ELEMS > 2 * BS_ELEMS ? Y1_ELEMS : 2 * BS_ELEMS;
  static_assert(G >= 2, "the commit sits behind the first tap of a stage");
  extern __shared__ uint4 lds[];
  uint4* Y1 = lds;                                  // [chunk][op][h][N1P]; phase 1: the two input tiles [buf][op][h][WROW]
  uint4* Bs = Y1;
  uint4* As = Y1 + R0_ELEMS;                        // [buf][tap][op][h][C]
  unsigned* cnt = reinterpret_cast<unsigned*>(As + 2 * A_ELEMS);   // {ready, done}

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WC, wc = wave % WC;
  const int i = lane & 31, h = lane >> 5;
  const int b = blockIdx.z;
  const int pad1 = (K - 1) * a.dil / 2;
  int tile = blockIdx.x;
  if (a.xcd_order) {
    const int nt = gridDim.x, xcd = tile & 7, idx = tile >> 3, q = nt >> 3, r = nt & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int n0 = tile * BN_OUT;
  const int len = a.lens ? a.lens[b] : a.T;
  const int wuse = N1 + (K - 1) * a.dil;
  const int in_base = n0 - H2 - pad1;
  const float slope = a.slope;
  const H3Rsrc xr = h3_rsrc(a.x + (long)b * a.bs, C * a.cs * 4);
  const H3Rsrc w1r = h3_rsrc(a.w1, K * NCHUNK * 4 * C * 16);
  const H3Rsrc w2r = h3_rsrc(a.w2, K * NCHUNK * 4 * C * 16);
  const int xrow = a.cs * 4;
  constexpr int slab = 4 * C * 16;
  constexpr float inv = 1.f / kH3Scale;
  auto stamp = [&](int k) {
    if (a.trace && tid == 0) a.trace[((long)b * gridDim.x + blockIdx.x) * 8 + k] = (long long)wall_clock64();
  };
  stamp(0);
  if (a.trace && tid == 0) a.trace[((long)b * gridDim.x + blockIdx.x) * 8 + 6] = tile;

  // ---- the two counters: one lane per wave adds, every wave polls (a broadcast LDS read)
  auto arrive = [&](int which) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave's LDS stores / fragment reads are performed
#endif
    if (lane == 0) __hip_atomic_fetch_add(&cnt[which], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  auto wait_for = [&](int which, unsigned target) {
    unsigned spins = 0;           // bounded: a protocol bug must end in wrong numbers (the tests see them), not in a hung GPU
    while (__hip_atomic_load(&cnt[which], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target && ++spins < (1u << 20))
      __builtin_amdgcn_s_sleep(1);
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::: "memory");
#endif
  };

  f32x16 acc[WM][WN];
  auto zero_acc = [&]() {
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
      for (int n = 0; n < WN; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  };

  // ---- weight staging: element e = tid + THREADS j -> (tap kkl, op, h, co) of one stage
  int a_kkl[NA], a_off[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int e = tid + THREADS * j;
    const int co = e % C, rest = e / C;              // rest = (kkl*2 + op)*2 + h
    a_kkl[j] = rest / 4;
    a_off[j] = e < A_ELEMS ? ((rest % 4) * C + co) * 16 : kH3Oob;
  }
  uint4 ra[NA];
  auto fetch_a = [&](const H3Rsrc& wres, int chunk, int g) {
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int kk = g * G + a_kkl[j];
      const bool ok = a_off[j] != kH3Oob && kk < K;
      ra[j] = h3_load4(wres, ok ? (kk * NCHUNK + chunk) * slab + a_off[j] : kH3Oob);
    }
  };
  auto commit_a = [&](int buf) {
#pragma unroll
    for (int j = 0; j < NA; ++j)
      if (NA * THREADS == A_ELEMS || tid + THREADS * j < A_ELEMS) As[buf * A_ELEMS + tid + THREADS * j] = ra[j];
  };
  // ---- input tile staging (phase 1)
  int b_off[NBT], b_row[NBT];
#pragma unroll
  for (int j = 0; j < NBT; ++j) {
    const int t = tid + THREADS * j;
    const int hh = t / WROW, p = t - hh * WROW;
    const int pos = in_base + p;
    b_row[j] = hh * 8;
    b_off[j] = (t < B_TASKS && p < wuse && pos >= 0 && pos < len) ? pos * 4 : kH3Oob;
  }
  float rb[NBT][8];
  bool ovf = false;
  auto fetch_b = [&](int chunk) {
#pragma unroll
    for (int j = 0; j < NBT; ++j) {
      const int row0 = (chunk * 16 + b_row[j]) * xrow;
#pragma unroll
      for (int q = 0; q < 8; ++q) rb[j][q] = h3_load1(xr, b_off[j] == kH3Oob ? kH3Oob : row0 + q * xrow + b_off[j]);
    }
  };
  auto commit_b = [&](int buf) {
    uint4* Bd = Bs + buf * BS_ELEMS;
#pragma unroll
    for (int j = 0; j < NBT; ++j) {
      const int t = tid + THREADS * j;
      if (NBT * THREADS == B_TASKS || t < B_TASKS) {
        half8 hi, lo;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          float v = rb[j][q];
          v = v > 0.f ? v : v * slope;
          ovf |= !(fabsf(v) < kH3ActLimit);
          const _Float16 vh = (_Float16)v;
          hi[q] = vh;
          lo[q] = (_Float16)((v - (float)vh) * kH3Scale);
        }
        const int hh = t / WROW, p = t - hh * WROW;
        Bd[(0 * 2 + hh) * WROW + p] = __builtin_bit_cast(uint4, hi);
        Bd[(1 * 2 + hh) * WROW + p] = __builtin_bit_cast(uint4, lo);
      }
    }
  };

  // the k-steps of one stage: TAPS taps; `mid` runs behind the MFMAs of the first tap (the commit of the next stage)
  auto compute = [&](auto taps_tag, const uint4* Ab, const uint4* Bt, int pitch, int kk0, int tap_step, auto&& mid) {
    constexpr int TAPS = decltype(taps_tag)::value;
    half8 af[2][2][WM], bf[2][2][WN];
    auto load = [&](int buf, int s) {
      const int tp = (kk0 + s) * tap_step;
#pragma unroll
      for (int m = 0; m < WM; ++m) {
        af[buf][0][m] = __builtin_bit_cast(half8, Ab[((s * 2 + 0) * 2 + h) * C + wr * (WM * 32) + m * 32 + i]);
        af[buf][1][m] = __builtin_bit_cast(half8, Ab[((s * 2 + 1) * 2 + h) * C + wr * (WM * 32) + m * 32 + i]);
      }
#pragma unroll
      for (int op = 0; op < 2; ++op)
#pragma unroll
        for (int n = 0; n < WN; ++n)
          bf[buf][op][n] = __builtin_bit_cast(half8, Bt[(op * 2 + h) * pitch + wc * (WN * 32) + n * 32 + i + tp]);
    };
    load(0, 0);
#pragma unroll
    for (int s = 0; s < TAPS; ++s) {
      const int cur = s & 1;
      if (s + 1 < TAPS) load(cur ^ 1, s + 1);
#pragma unroll
      for (int m = 0; m < WM; ++m) {
        const half8 wh = af[cur][0][m] * (_Float16)(1.f / kH3Scale);
#pragma unroll
        for (int n = 0; n < WN; ++n) {
          acc[m][n] = h3_mfma(af[cur][0][m], bf[cur][0][n], acc[m][n]);
          acc[m][n] = h3_mfma(wh, bf[cur][1][n], acc[m][n]);
          acc[m][n] = h3_mfma(af[cur][1][m], bf[cur][0][n], acc[m][n]);
        }
      }
      if (s == 0) mid();
    }
  };

  unsigned n_ready = 0, n_done = 0;        // this wave's wait targets so far (monotonic over the whole tile)
  // ================================================================ phase 1
  if (tid < 2) cnt[tid] = 0;
  zero_acc();
  fetch_b(0);
  fetch_a(w1r, 0, 0);
  __syncthreads();                           // counters are zero
  commit_b(0);
  commit_a(0);
  arrive(0);                                 // stage 0 (and input tile 0) committed by this wave
  if (NCHUNK > 1) fetch_b(1);
  {
    // registers hold: ra = weights of the stage to be committed next (stage sidx + 1), rb = input of chunk + 1
    // next stage of (chunk, g): (chunk, g + 1) or (chunk + 1, 0); behind the last stage of phase 1: w2's (0, 0)
    fetch_a(w1r, 1 / NG, 1 % NG);
  }
  stamp(1);
  int sidx = 0;
  for (int chunk = 0; chunk < NCHUNK; ++chunk) {
    for_each_group<0, NG>([&](auto g_tag) {
      constexpr int GG = decltype(g_tag)::value;
      constexpr int TAPS = (NG > 1 && GG == NG - 1) ? K - (NG - 1) * G : G;
      const int buf = sidx & 1;
      n_ready += NW;
      wait_for(0, n_ready);
      compute(std::integral_constant<int, TAPS>{}, As + buf * A_ELEMS, Bs + (chunk & 1) * BS_ELEMS, WROW, GG * G, a.dil, [&]() {
        // ---- middle of the stage: stage sidx + 1 goes to the other weight buffer (its last readers ran stage sidx - 1)
        wait_for(1, n_done);
        commit_a(buf ^ 1);
        if (GG == 0 && chunk + 1 < NCHUNK) commit_b((chunk + 1) & 1);     // last read during chunk - 1: long over
        arrive(0);
        // refill the registers: the stage after the next one; the input of the chunk after the next one
        const int t2 = chunk * NG + GG + 2;                                // the stage after the next one
        if (t2 < NS) fetch_a(w1r, t2 / NG, t2 % NG);
        else fetch_a(w2r, (t2 - NS) / NG, (t2 - NS) % NG);                // ... of phase 2 (its stage 0 or 1)
        if (GG == 0 && chunk + 2 < NCHUNK) fetch_b(chunk + 2);
      });
      arrive(1);
      n_done += NW;
      ++sidx;
    });
  }
  // here: every stage of phase 1 issued; As[sidx & 1] holds (or will hold) phase 2's stage 0, committed in the middle of
  // the last stage; ra holds phase 2's stage 1
  stamp(2);
  __syncthreads();                           // every wave is done reading the input tiles: Y1 may overwrite them
#pragma unroll
  for (int m = 0; m < WM; ++m)
#pragma unroll
    for (int n = 0; n < WN; ++n) {
      const int c_t = wr * (WM * 32) + m * 32;
      const int j = wc * (WN * 32) + n * 32 + i;
      const int pos1 = n0 - H2 + j;
      const bool live = pos1 >= 0 && pos1 < len;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        typedef _Float16 half4 __attribute__((ext_vector_type(4)));
        const int cg = c_t + 8 * g;
        half4 hi, lo;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v = acc[m][n][4 * g + q] * inv + (a.b1 ? a.b1[cg + 4 * h + q] : 0.f);
          v = fmaxf(v, v * slope);
          v = live ? v : 0.f;
          ovf |= !(fabsf(v) < kH3ActLimit);
          const _Float16 vh = (_Float16)v;
          hi[q] = vh;
          lo[q] = (_Float16)((v - (float)vh) * kH3Scale);
        }
        char* e_hi = reinterpret_cast<char*>(Y1 + (((cg >> 4) * 2 + 0) * 2 + (g & 1)) * N1P + j) + 8 * h;
        char* e_lo = reinterpret_cast<char*>(Y1 + (((cg >> 4) * 2 + 1) * 2 + (g & 1)) * N1P + j) + 8 * h;
        *reinterpret_cast<half4*>(e_hi) = hi;
        *reinterpret_cast<half4*>(e_lo) = lo;
      }
    }
  if (ovf) report_h3_overflow(a.ovf, a.ovf_layer, a.seq);
  zero_acc();
  __syncthreads();                           // Y1 is complete
  stamp(3);

  // ================================================================ phase 2 (the stage counter runs on: sidx = NS ...)
  for (int chunk = 0; chunk < NCHUNK; ++chunk) {
    for_each_group<0, NG>([&](auto g_tag) {
      constexpr int GG = decltype(g_tag)::value;
      constexpr int TAPS = (NG > 1 && GG == NG - 1) ? K - (NG - 1) * G : G;
      const int buf = sidx & 1;
      const bool has_next = !(chunk + 1 >= NCHUNK && GG + 1 >= NG);
      n_ready += NW;
      wait_for(0, n_ready);
      compute(std::integral_constant<int, TAPS>{}, As + buf * A_ELEMS, Y1 + chunk * 4 * N1P, N1P, GG * G, 1, [&]() {
        if (has_next) {
          wait_for(1, n_done);
          commit_a(buf ^ 1);
          arrive(0);
          const int t2 = chunk * NG + GG + 2;
          if (t2 < NS) fetch_a(w2r, t2 / NG, t2 % NG);
        }
      });
      arrive(1);
      n_done += NW;
      ++sidx;
    });
  }
  stamp(4);
  {
    constexpr int P = 36;
    static_assert(THREADS / 64 * 32 * P * 4 <= Y1_ELEMS * 16, "staging tiles must fit the Y1 region");
    __syncthreads();
    float* stg = reinterpret_cast<float*>(lds) + wave * (32 * P);
    const int lr = lane >> 3, lc = (lane & 7) * 4;
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
      for (int n = 0; n < WN; ++n) {
#pragma unroll
        for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * h) * P + i] = acc[m][n][r] * inv;
        __builtin_amdgcn_wave_barrier();
        const int col0 = wc * (WN * 32) + n * 32 + lc;
        const int pos0 = n0 + col0;
        const bool ok = col0 < BN_OUT && pos0 < a.T;
        float4 v[4], rv[4], pv[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int co = wr * (WM * 32) + m * 32 + lr + 8 * p;
          const long off = (long)b * a.bs + (long)co * a.cs + pos0;
          v[p] = *reinterpret_cast<const float4*>(stg + (lr + 8 * p) * P + lc);
          rv[p] = ok ? *reinterpret_cast<const float4*>(a.x + off) : make_float4(0.f, 0.f, 0.f, 0.f);
          pv[p] = (ok && a.acc2_mode != ACC2_NONE && a.acc2_mode != ACC2_SET) ? *reinterpret_cast<const float4*>(a.y2 + off)
                                                                              : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __builtin_amdgcn_wave_barrier();
        if (ok) {
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const int co = wr * (WM * 32) + m * 32 + lr + 8 * p;
            const long off = (long)b * a.bs + (long)co * a.cs + pos0;
            const float bb = a.b2 ? a.b2[co] : 0.f;
            float o[4] = {v[p].x + bb + rv[p].x, v[p].y + bb + rv[p].y, v[p].z + bb + rv[p].z, v[p].w + bb + rv[p].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = pos0 + q < len ? o[q] : 0.f;
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
  stamp(5);
}

// ====================================================================================================================
// Round 5: the PERSISTENT form of the fused step (RVCX_PAIR_VARIANT=10; the default where it is instantiated).
//
// Per-workgroup phase stamps of the kernel at the top of this file (tools/pair_trace.py, round 5) show how much of a
// tile's life is NOT in the two k-loops:   C = 128, k = 11: 9.5 of 94 us    C = 64, k = 3: 10.8 of 18.9 us
// C = 32, k = 3: 12.7 of 15.5 us -- launch + address set-up + the input tile's trip from HBM (2.2 - 5.7 us), the bias
// loads inside the c1 epilogue, and the residual's round trip inside the final epilogue (4.5 - 6.6 us).  None of it is
// work; all of it is latency that a workgroup with the CU to itself cannot hide.  Here a workgroup stays on its CU and
// walks a CONTIGUOUS range of tiles (XCD x owns a contiguous eighth of the tiles, its workgroups contiguous parts of
// that: neighbouring tiles share their halo in one L2, as before):
//   * the input tile of tile t + 1 is requested when tile t's c1 loop ends, converted and committed to LDS in the last
//     stage of tile t's c2 loop (the input staging area is idle in phase 2); its first weights are requested in that
//     stage too: the prologue of every tile but the first is a weight commit, not a round trip to HBM;
//   * the residual (the 16-byte-per-lane words of the wide epilogue) is requested in that same last stage and is there
//     when the epilogue asks for it;
//   * biases are read once per launch (into LDS; as registers they would cost 20 per lane and spill);
//   * no launch, no dispatch, no resource set-up per tile.
// The arithmetic is the first kernel's, instruction for instruction (same split, same k-order, same epilogues): the
// results are bit-identical to it and to the two conv launches (tests: RVCX_PAIR_VARIANT 0 / 10 vs unfused).
// EARLY: request tile t + 1's input and the first residual block as early as the staging registers allow (512-thread forms:
// 256 registers per lane); false: at the c1 epilogue / in the last c2 stage (768-thread forms: 168 registers)
template <int C, int NT, int WR, int WC, int K, int KKT, int NCS, bool EARLY>
__global__ __launch_bounds__(64 * WR * WC) void resblock_pair_persist_kernel(const PairArgs a) {
  constexpr int THREADS = 64 * WR * WC;
  constexpr int N1 = 32 * NT, N1P = N1 + kPairPadY, WROW = N1 + kPairHalo;
  constexpr int WM = (C / 32) / WR, WN = NT / WC;
  constexpr int NCHUNK = C / 16;
  constexpr int NG = (K + KKT - 1) / KKT;
  constexpr int SL = NCS * KKT;
  constexpr int A_ELEMS = SL * 4 * C, NA = (A_ELEMS + THREADS - 1) / THREADS;
  constexpr int B_TASKS = NCS * 2 * WROW, NBT = (B_TASKS + THREADS - 1) / THREADS;
  constexpr int BN_OUT = (N1 - (K - 1)) & ~3, H2 = (K - 1) / 2;
  static_assert(WM >= 1 && WN >= 1 && WM * WR * 32 == C && WN * WC == NT, "bad tile");
  static_assert(NCS == 1 || KKT == K, "several chunks per stage only with all taps resident");
  static_assert(NCHUNK % NCS == 0, "chunk sets must tile the channels");
  extern __shared__ uint4 lds[];
  constexpr int Y1_ELEMS = NCHUNK * 4 * N1P;
  uint4* Y1 = lds;                              // [chunk][op][h][N1P]
  uint4* As = Y1 + Y1_ELEMS;                    // [slot][op][h][C]
  uint4* Bs = As + A_ELEMS;                     // [cl][op][h][WROW]
  float* bias_s = reinterpret_cast<float*>(Bs + NCS * 4 * WROW);   // b1[C], b2[C]: read once per launch, 16-byte reads after

  // per-thread coordinates: NOT const -- they are re-derived at the top of every tile from an opaque copy of the thread id,
  // so that the hundred address terms built from them are recomputed per tile (a few dozen VALU instructions) instead of
  // being hoisted out of the tile loop and kept in registers for the whole launch (first build: 256 VGPRs + 28-154 spilled)
  int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int wr = wave / WC, wc = wave % WC;
  int i = lane & 31, h = lane >> 5;
  const int b = blockIdx.z;
  const int pad1 = (K - 1) * a.dil / 2;
  const int len = a.lens ? a.lens[b] : a.T;
  const int wuse = N1 + (K - 1) * a.dil;
  const float slope = a.slope;
  const H3Rsrc xr = h3_rsrc(a.x + (long)b * a.bs, C * a.cs * 4);
  const H3Rsrc w1r = h3_rsrc(a.w1, K * NCHUNK * 4 * C * 16);
  const H3Rsrc w2r = h3_rsrc(a.w2, K * NCHUNK * 4 * C * 16);
  const int xrow = a.cs * 4;
  constexpr int slab = 4 * C * 16;

  // ---- this workgroup's tiles: XCD (blockIdx.x & 7) owns a contiguous range, its workgroups contiguous parts of it
  int first, count;
  {
    const int nt = (a.T + BN_OUT - 1) / BN_OUT, nwg = gridDim.x, w = blockIdx.x;
    if (a.xcd_order && nwg >= 8) {
      const int x = w & 7, j = w >> 3;
      const int q = nt >> 3, r = nt & 7;
      const int x0 = x * q + (x < r ? x : r), xn = q + (x < r ? 1 : 0);          // tiles of this XCD
      const int m = (nwg >> 3) + (x < (nwg & 7) ? 1 : 0);                        // its workgroups
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
  const int ntiles_all = (a.T + BN_OUT - 1) / BN_OUT;
  int cur_tile = first;
  auto stamp = [&](int k) {
    if (a.trace && tid == 0) a.trace[((long)b * ntiles_all + cur_tile) * 8 + k] = (long long)wall_clock64();
  };

  f32x16 acc[WM][WN];
  auto zero_acc = [&]() {
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
      for (int n = 0; n < WN; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  };

  int a_cl[NA], a_kkl[NA], a_off[NA];
  int b_off[NBT], b_row[NBT];
  int lr = 0, lc = 0;
  auto derive_thread = [&]() {
    int t0 = threadIdx.x;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(t0));                   // opaque: nothing derived from it is loop-invariant to the compiler
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
      const int co = e % C, rest = e / C;
      const int slot = rest / 4;
      a_cl[j] = slot / KKT;
      a_kkl[j] = slot % KKT;
      a_off[j] = e < A_ELEMS ? ((rest % 4) * C + co) * 16 : kH3Oob;
    }
#pragma unroll
    for (int j = 0; j < NBT; ++j) {
      const int t = tid + THREADS * j;
      const int ch2 = t / WROW;
      b_row[j] = (ch2 >> 1) * 16 + (ch2 & 1) * 8;
    }
  };
  derive_thread();
  auto set_input_tile = [&](int tile) {           // where the input tasks of this thread read for `tile`
    const int in_base = tile * BN_OUT - H2 - pad1;
#pragma unroll
    for (int j = 0; j < NBT; ++j) {
      const int t = tid + THREADS * j;
      const int ch2 = t / WROW, p = t - ch2 * WROW;
      const int pos = in_base + p;
      b_off[j] = (t < B_TASKS && p < wuse && pos >= 0 && pos < len) ? pos * 4 : kH3Oob;
    }
  };

  uint4 ra[NA];
  float rb[NBT][8];
  bool ovf = false;
  auto fetch_b = [&](int chunk0) {
#pragma unroll
    for (int j = 0; j < NBT; ++j) {
      const int row0 = (chunk0 * 16 + b_row[j]) * xrow;
#pragma unroll
      for (int q = 0; q < 8; ++q) rb[j][q] = h3_load1(xr, b_off[j] == kH3Oob ? kH3Oob : row0 + q * xrow + b_off[j]);
    }
  };
  auto fetch_a = [&](const H3Rsrc& wr_, int chunk0, int kk0) {
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int kk = kk0 + a_kkl[j];
      const bool ok = a_off[j] != kH3Oob && kk < K;
      ra[j] = h3_load4(wr_, ok ? (kk * NCHUNK + chunk0 + a_cl[j]) * slab + a_off[j] : kH3Oob);
    }
  };
  auto commit_a = [&]() {
#pragma unroll
    for (int j = 0; j < NA; ++j)
      if (NA * THREADS == A_ELEMS || tid + THREADS * j < A_ELEMS) As[tid + THREADS * j] = ra[j];
  };
  auto commit_b = [&]() {
#pragma unroll
    for (int j = 0; j < NBT; ++j) {
      const int t = tid + THREADS * j;
      if (NBT * THREADS == B_TASKS || t < B_TASKS) {
        half8 hi, lo;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          float v = rb[j][q];
          v = v > 0.f ? v : v * slope;                 // leaky_relu ahead of c1
          ovf |= !(fabsf(v) < kH3ActLimit);
          const _Float16 vh = (_Float16)v;
          hi[q] = vh;
          lo[q] = (_Float16)((v - (float)vh) * kH3Scale);
        }
        const int ch2 = t / WROW, p = t - ch2 * WROW;
        const int cl = ch2 >> 1, hh = ch2 & 1;
        Bs[((cl * 2 + 0) * 2 + hh) * WROW + p] = __builtin_bit_cast(uint4, hi);
        Bs[((cl * 2 + 1) * 2 + hh) * WROW + p] = __builtin_bit_cast(uint4, lo);
      }
    }
  };
  auto compute = [&](auto taps_tag, const uint4* Bt, int pitch, int cl_pitch, int kk0, int tap_step) {
    constexpr int TAPS = decltype(taps_tag)::value;
    constexpr int NS = NCS * TAPS;
    half8 af[2][2][WM], bf[2][2][WN];
    auto load = [&](int buf, int s) {
      const int cl = s / TAPS, kkl = s % TAPS;
      const int tp = (kk0 + kkl) * tap_step;
      const uint4* Bc = Bt + cl * cl_pitch;
#pragma unroll
      for (int m = 0; m < WM; ++m) {
        af[buf][0][m] = __builtin_bit_cast(half8, As[(((cl * KKT + kkl) * 2 + 0) * 2 + h) * C + wr * (WM * 32) + m * 32 + i]);
        af[buf][1][m] = __builtin_bit_cast(half8, As[(((cl * KKT + kkl) * 2 + 1) * 2 + h) * C + wr * (WM * 32) + m * 32 + i]);
      }
#pragma unroll
      for (int op = 0; op < 2; ++op)
#pragma unroll
        for (int n = 0; n < WN; ++n)
          bf[buf][op][n] = __builtin_bit_cast(half8, Bc[(op * 2 + h) * pitch + wc * (WN * 32) + n * 32 + i + tp]);
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
    if (kPipe) {
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
  using Full = std::integral_constant<int, KKT>;
  using Last = std::integral_constant<int, K - (NG - 1) * KKT>;
  constexpr float inv = 1.f / kH3Scale;

  // biases, once per launch, into LDS (as registers they cost 20 per lane for the whole launch: spills)
  for (int c = tid; c < C; c += THREADS) {
    bias_s[c] = a.b1 ? a.b1[c] : 0.f;
    bias_s[C + c] = a.b2 ? a.b2[c] : 0.f;
  }

  // ---- first tile: its input chunk set 0 into Bs, its first weights into registers.  Loop invariant from here on:
  // at the top of a tile Bs holds the tile's chunk set 0 (converted), ra its first weight stage.
  set_input_tile(first);
  fetch_b(0);
  fetch_a(w1r, 0, 0);
  commit_b();

  for (int ti = 0; ti < count; ++ti) {
    const int tile = first + ti;
    cur_tile = tile;
    const int n0 = tile * BN_OUT;
    const bool has_next = ti + 1 < count;
    if (ti > 0) derive_thread();
    stamp(0);
    // ================================================================ phase 1: Y1 = lrelu(c1(lrelu(x)) + b1)
    zero_acc();
    for (int chunk = 0; chunk < NCHUNK; chunk += NCS) {
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const bool last_stage = chunk + NCS >= NCHUNK && g == NG - 1;
        __syncthreads();
        if (g == 0 && chunk > 0) commit_b();           // chunk set 0 was committed by the previous tile (or the preamble)
        commit_a();
        __syncthreads();
        if (chunk == 0 && g == 0) stamp(1);
        if (!last_stage) fetch_a(w1r, g + 1 == NG ? chunk + NCS : chunk, g + 1 == NG ? 0 : (g + 1) * KKT);
        else fetch_a(w2r, 0, 0);                       // first weights of c2 fly during the c1 epilogue
        if (g == 0 && chunk + NCS < NCHUNK) {
          fetch_b(chunk + NCS);
        } else if (EARLY && g == 0 && has_next) {
          // the tile's last input commit is behind us: the staging registers are free, tile t + 1's input leaves HBM now
          // (a loaded round trip is ~5 us; requested at the c1 epilogue it arrived 2-4 us late for the small shapes)
          set_input_tile(tile + 1);
          fetch_b(0);
        }
        if (NG > 1 && g == NG - 1) compute(Last{}, Bs, WROW, 4 * WROW, g * KKT, a.dil);
        else compute(Full{}, Bs, WROW, 4 * WROW, g * KKT, a.dil);
      }
    }
    stamp(2);
    if (!EARLY && has_next) {
      set_input_tile(tile + 1);
      fetch_b(0);
    }
    // c1 epilogue: bias, leaky_relu (the one ahead of c2), zero outside the sequence, split, into LDS
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
      for (int n = 0; n < WN; ++n) {
        const int c_t = wr * (WM * 32) + m * 32;
        const int j = wc * (WN * 32) + n * 32 + i;
        const int pos1 = n0 - H2 + j;
        const bool live = pos1 >= 0 && pos1 < len;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          typedef _Float16 half4 __attribute__((ext_vector_type(4)));
          const int cg = c_t + 8 * g;
          half4 hi, lo;
          const float4 bq = *reinterpret_cast<const float4*>(bias_s + cg + 4 * h);
          const float b4[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float v = acc[m][n][4 * g + q] * inv + b4[q];
            v = fmaxf(v, v * slope);
            v = live ? v : 0.f;
            ovf |= !(fabsf(v) < kH3ActLimit);
            const _Float16 vh = (_Float16)v;
            hi[q] = vh;
            lo[q] = (_Float16)((v - (float)vh) * kH3Scale);
          }
          char* e_hi = reinterpret_cast<char*>(Y1 + (((cg >> 4) * 2 + 0) * 2 + (g & 1)) * N1P + j) + 8 * h;
          char* e_lo = reinterpret_cast<char*>(Y1 + (((cg >> 4) * 2 + 1) * 2 + (g & 1)) * N1P + j) + 8 * h;
          *reinterpret_cast<half4*>(e_hi) = hi;
          *reinterpret_cast<half4*>(e_lo) = lo;
        }
      }

    // ================================================================ phase 2: y = c2(Y1) + b2 + x
    zero_acc();
    stamp(3);
    // The residual words of the wide epilogue.  Column block n = 0 is requested EARLY (all taps resident: at the start of
    // phase 2; otherwise with the first tap group of the last chunk set), blocks n >= 1 when the epilogue starts -- they
    // land while block 0 is transposed and stored.  Half the registers of requesting everything early.
    uint4 rv[WM][WN][4];
    auto request_res = [&](int n) {
#pragma unroll
      for (int m = 0; m < WM; ++m) {
        const int col0 = wc * (WN * 32) + n * 32 + lc;
        const int pos0 = n0 + col0;
        const bool ok = col0 < BN_OUT && pos0 < a.T;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int co = wr * (WM * 32) + m * 32 + lr + 8 * p;
          rv[m][n][p] = h3_load4(xr, ok ? (co * a.cs + pos0) * 4 : kH3Oob);
        }
      }
    };
    if (EARLY && NG == 1) request_res(0);
    for (int chunk = 0; chunk < NCHUNK; chunk += NCS) {
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const bool last_stage = chunk + NCS >= NCHUNK && g == NG - 1;
        __syncthreads();
        commit_a();
        if (last_stage && has_next) commit_b();        // Bs is idle in phase 2: tile t + 1's chunk set 0 is ready before its tile starts
        __syncthreads();
        if (!last_stage) fetch_a(w2r, g + 1 == NG ? chunk + NCS : chunk, g + 1 == NG ? 0 : (g + 1) * KKT);
        else if (has_next) fetch_a(w1r, 0, 0);       // tile t + 1's first weights
        if (EARLY ? (NG > 1 && g == 0 && chunk + NCS >= NCHUNK) : last_stage) request_res(0);
        if (NG > 1 && g == NG - 1) compute(Last{}, Y1 + chunk * 4 * N1P, N1P, 4 * N1P, g * KKT, 1);
        else compute(Full{}, Y1 + chunk * 4 * N1P, N1P, 4 * N1P, g * KKT, 1);
      }
    }
    stamp(4);
    {
      constexpr int P = 36;                              // floats per staged row: 16-byte aligned, rows 4 banks apart
      static_assert(THREADS / 64 * 32 * P * 4 <= NCHUNK * 4 * N1P * 16, "staging tiles must fit the Y1 region");
      __syncthreads();                                   // every wave is done reading Y1: the staging tiles overlay it
      float* stg = reinterpret_cast<float*>(lds) + wave * (32 * P);
#pragma unroll
      for (int n = 1; n < WN; ++n) request_res(n);
#pragma unroll
      for (int n = 0; n < WN; ++n)
#pragma unroll
        for (int m = 0; m < WM; ++m) {
#pragma unroll
          for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * h) * P + i] = acc[m][n][r] * inv;
          __builtin_amdgcn_wave_barrier();
          const int col0 = wc * (WN * 32) + n * 32 + lc;
          const int pos0 = n0 + col0;
          const bool ok = col0 < BN_OUT && pos0 < a.T;
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
              const float bb = bias_s[C + co];
              const float4 rr = __builtin_bit_cast(float4, rv[m][n][p]);
              float o[4] = {v[p].x + bb + rr.x, v[p].y + bb + rr.y, v[p].z + bb + rr.z, v[p].w + bb + rr.w};
#pragma unroll
              for (int q = 0; q < 4; ++q) o[q] = pos0 + q < len ? o[q] : 0.f;
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
    stamp(5);
  }
  if (ovf) report_h3_overflow(a.ovf, a.ovf_layer, a.seq);
}

struct PairCfg {
  int C, K, n1, threads, variant, bn_out;
  bool wide;                     // wide epilogue: rows must be 16-byte aligned
  size_t lds;
  void (*kern)(const PairArgs);
  bool persist = false;          // one workgroup per CU walks a range of tiles (grid = CUs, not tiles)
  int per_cu = 1;                // persistent workgroups per CU (small tiles: two, so that one's VALU phases meet the other's loops)
};
template <int C, int NT, int WR, int WC, int K, int KKT, int NCS, int V = 0, bool TRIM = false, int RESPF = 0, bool OVL = false>
constexpr PairCfg make_cfg() {
  constexpr size_t y1 = (size_t)(C / 16) * 4 * (32 * NT + pair_pady(K, TRIM)), as = (size_t)NCS * KKT * 4 * C,
                   bs = (size_t)NCS * 4 * (32 * NT + pair_halo(K, TRIM));
  return {C, K, 32 * NT, 64 * WR * WC, V, RESPF == 3 ? ((32 * NT - (K - 1)) & ~3) : 32 * NT - (K - 1), RESPF == 3,
          (OVL ? (y1 > bs ? y1 : bs) + as : y1 + as + bs) * 16,
          resblock_pair_kernel<C, NT, WR, WC, K, KKT, NCS, TRIM, RESPF, OVL>};
}
template <int C, int NT, int WR, int WC, int K, int KKT, int NCS, bool EARLY = true, int V = 10, int PER_CU = 1>
constexpr PairCfg make_persist() {
  constexpr size_t y1 = (size_t)(C / 16) * 4 * (32 * NT + kPairPadY), as = (size_t)NCS * KKT * 4 * C,
                   bs = (size_t)NCS * 4 * (32 * NT + kPairHalo);
  return {C, K, 32 * NT, 64 * WR * WC, V, (32 * NT - (K - 1)) & ~3, true, (y1 + as + bs) * 16 + 2 * C * 4,
          resblock_pair_persist_kernel<C, NT, WR, WC, K, KKT, NCS, EARLY>, true, PER_CU};
}
template <int C, int NT, int WR, int WC, int K, int G, int V, bool OVL = false>
constexpr PairCfg make_adir() {
  constexpr size_t y1 = (size_t)(C / 16) * 4 * (32 * NT + kPairPadY), bs = (size_t)2 * 4 * (32 * NT + kPairHalo);
  return {C, K, 32 * NT, 64 * WR * WC, V, (32 * NT - (K - 1)) & ~3, true, (OVL ? y1 : y1 + bs) * 16,
          resblock_pair_adir_kernel<C, NT, WR, WC, K, G, OVL>};
}
template <int C, int NT, int WR, int WC, int K, int G, int V>
constexpr PairCfg make_sp() {
  constexpr size_t y1 = (size_t)(C / 16) * 4 * (32 * NT + kPairPadY), bs2 = (size_t)2 * 4 * (32 * NT + kPairHalo),
                   as = (size_t)2 * G * 4 * C;
  return {C, K, 32 * NT, 64 * WR * WC, V, (32 * NT - (K - 1)) & ~3, true, ((y1 > bs2 ? y1 : bs2) + as) * 16 + 64,
          resblock_pair_sp_kernel<C, NT, WR, WC, K, G>};
}
// (C, K) instantiations of the RVC v2 decoders (resblock kernels 3 / 7 / 11): 512 threads, one workgroup per CU
const PairCfg kPair[] = {
    make_cfg<32, 16, 1, 8, 3, 3, 2, 0, false, 2>(),  make_cfg<32, 16, 1, 8, 7, 7, 1, 0, false, 2>(),  make_cfg<32, 16, 1, 8, 11, 11, 1, 0, false, 2>(),
    make_cfg<64, 8, 2, 4, 3, 3, 2, 0, false, 2>(),   make_cfg<64, 8, 2, 4, 7, 7, 1, 0, false, 2>(),   make_cfg<64, 8, 2, 4, 11, 11, 1, 0, false, 2>(),
    // C = 128: k = 3 and k = 11 run 768 threads on N1 = 192 (three waves per SIMD: +12 % / +6 % in tools/bench_pair.py),
    // k = 7 keeps 512 threads with all 7 taps resident (the 768-thread form would spill)
    make_cfg<128, 6, 4, 3, 3, 3, 1, 0, false, 2>(),  make_cfg<128, 4, 4, 2, 7, 7, 1, 0, false, 2>(),  make_cfg<128, 6, 4, 3, 11, 4, 1, 0, false, 1>(),
    // C = 256 (NSF stage 0): the c1 tile of all 256 channels is 1 KB per position, N1 = 96 fills the LDS (154 KB with two
    // weight taps resident); 12 waves of 64 x 32.  RVCX_PAIR_C256=0: the two conv_h3 launches instead
    make_cfg<256, 3, 4, 3, 3, 2, 1, 0, false, 1>(),  make_cfg<256, 3, 4, 3, 7, 2, 1, 0, false, 1>(),  make_cfg<256, 3, 4, 3, 11, 2, 1, 0, false, 1>(),
    // RVCX_PAIR_VARIANT=1: the alternatives, for A/B runs
    make_cfg<128, 4, 4, 2, 3, 3, 2, 1>(),  make_cfg<128, 6, 4, 3, 7, 4, 1, 1>(),  make_cfg<128, 4, 4, 2, 11, 6, 1, 1>(),
    // RVCX_PAIR_VARIANT=2: small tiles, TWO workgroups per CU (<= 80 KB of LDS, <= 168 registers): one computes while the
    // other is in its prologue / epilogue.  The timing ablations of round 3 (tools/ablate_pair.sh) showed the single
    // resident workgroup serialises its phases: MFMA time + everything else = the whole launch, and the residual loads /
    // output stores of the c2 epilogue alone are 80 us of every launch.
    make_cfg<32, 8, 1, 4, 3, 3, 2, 2, true>(),   make_cfg<32, 8, 1, 4, 7, 7, 1, 2, true>(),   make_cfg<32, 8, 1, 4, 11, 11, 1, 2, true>(),
    make_cfg<64, 4, 2, 2, 3, 3, 2, 2, true>(),   make_cfg<64, 4, 2, 2, 7, 7, 1, 2, true>(),   make_cfg<64, 4, 2, 2, 11, 4, 1, 2, true>(),
    make_cfg<128, 3, 2, 3, 3, 2, 1, 2, true>(),  make_cfg<128, 3, 2, 3, 7, 2, 1, 2, true>(),  make_cfg<128, 3, 2, 3, 11, 2, 1, 2, true>(),
    // variant 4: the wide (LDS-transposed, 16 bytes per lane) epilogue -- the default whenever the rows are 16-byte aligned
    make_cfg<32, 16, 1, 8, 3, 3, 2, 4, false, 3>(),  make_cfg<32, 16, 1, 8, 7, 7, 1, 4, false, 3>(),  make_cfg<32, 16, 1, 8, 11, 11, 1, 4, false, 3>(),
    make_cfg<64, 8, 2, 4, 3, 3, 2, 4, false, 3>(),   make_cfg<64, 8, 2, 4, 7, 7, 1, 4, false, 3>(),   make_cfg<64, 8, 2, 4, 11, 11, 1, 4, false, 3>(),
    make_cfg<128, 6, 4, 3, 3, 3, 1, 4, false, 3>(),  make_cfg<128, 4, 4, 2, 7, 7, 1, 4, false, 3>(),  make_cfg<128, 6, 4, 3, 11, 4, 1, 4, false, 3>(),
    make_cfg<256, 3, 4, 3, 3, 2, 1, 4, false, 3>(),  make_cfg<256, 3, 4, 3, 7, 2, 1, 4, false, 3>(),  make_cfg<256, 3, 4, 3, 11, 2, 1, 4, false, 3>(),
    // variant 5 (round 4): 8 waves of 64 x 64 (C >= 64) / 32 x 128 (C = 32) on twice as wide tiles, input tile overlaid on Y1
    make_cfg<32, 32, 1, 8, 3, 3, 2, 5, false, 3, true>(),  make_cfg<32, 32, 1, 8, 7, 7, 1, 5, false, 3, true>(),  make_cfg<32, 32, 1, 8, 11, 11, 1, 5, false, 3, true>(),
    make_cfg<64, 16, 1, 8, 3, 3, 2, 5, false, 3, true>(),  make_cfg<64, 16, 1, 8, 7, 3, 1, 5, false, 3, true>(),  make_cfg<64, 16, 1, 8, 11, 4, 1, 5, false, 3, true>(),
    make_cfg<128, 8, 2, 4, 3, 3, 1, 5, false, 3, true>(),  make_cfg<128, 8, 2, 4, 7, 3, 1, 5, false, 3, true>(),  make_cfg<128, 8, 2, 4, 11, 3, 1, 5, false, 3, true>(),
    // variant 6 (round 4): A-direct (weights from L2 into registers, barrier-free c2 loop), the same wave tiles as variant 4
    make_adir<32, 16, 1, 8, 3, 3, 6>(),  make_adir<32, 16, 1, 8, 7, 4, 6>(),  make_adir<32, 16, 1, 8, 11, 4, 6>(),
    make_adir<64, 8, 2, 4, 3, 3, 6>(),   make_adir<64, 8, 2, 4, 7, 4, 6>(),   make_adir<64, 8, 2, 4, 11, 4, 6>(),
    make_adir<128, 6, 4, 3, 3, 3, 6>(),  make_adir<128, 6, 4, 3, 7, 2, 6>(),  make_adir<128, 6, 4, 3, 11, 2, 6>(),
    // variant 7: A-direct on 8 waves of 64 x 64 (C >= 64) / 32 x 128 (C = 32), input tiles overlaid on Y1 where needed
    make_adir<32, 16, 1, 8, 3, 3, 7>(),  make_adir<32, 16, 1, 8, 7, 4, 7>(),  make_adir<32, 16, 1, 8, 11, 4, 7>(),
    make_adir<64, 16, 1, 8, 3, 2, 7, true>(),  make_adir<64, 16, 1, 8, 7, 2, 7, true>(),  make_adir<64, 16, 1, 8, 11, 2, 7, true>(),
    make_adir<128, 8, 2, 4, 3, 2, 7, true>(),  make_adir<128, 8, 2, 4, 7, 2, 7, true>(),  make_adir<128, 8, 2, 4, 11, 2, 7, true>(),
    // variant 8 (round 4): split-phase staging -- no s_barrier in the k-loops, LDS counters instead (see the kernel)
    make_sp<32, 16, 1, 8, 3, 3, 8>(),  make_sp<32, 16, 1, 8, 7, 4, 8>(),  make_sp<32, 16, 1, 8, 11, 4, 8>(),
    make_sp<64, 8, 2, 4, 3, 3, 8>(),   make_sp<64, 8, 2, 4, 7, 4, 8>(),   make_sp<64, 8, 2, 4, 11, 4, 8>(),
    make_sp<128, 6, 4, 3, 3, 3, 8>(),  make_sp<128, 6, 4, 3, 7, 3, 8>(),  make_sp<128, 6, 4, 3, 11, 3, 8>(),
    // (variant 9, round 5: variant 2's small tiles -- two workgroups per CU -- with the wide epilogue: -6 ... -8 % at C <= 64 / k = 3,
    // +50 % at C = 128; superseded by the persistent form and removed)
    // variant 10 (round 5): PERSISTENT workgroups on variant 4's tiles -- the default for these shapes (RVCX_PAIR_PERSIST=0: variant 4).
    // Measured per shape against variant 4 (tools/bench_pair.py, same box; profiles/pair_persist_r05.txt): k = 3 -11 ... -15 %,
    // C = 32 -8 ... -12 %, C = 64 / 128 at k = 7 -4 ... -6 %, C = 64 k = 11 -3 %.  C = 128, k = 11 is NOT here: on 768 threads (168
    // registers) the form spills inside its k-loops and loses 3 %; on 512 threads (variant 12) it loses 10 %.
    // EARLY = true (input of tile t + 1 requested during the last chunk set of c1) is slower on every multi-stage shape: vector
    // memory loads return IN ORDER, so the weight fetch of the next stage waits behind the ~5 us HBM round trip of the input tile.
    make_persist<32, 16, 1, 8, 3, 3, 2, false>(),  make_persist<32, 16, 1, 8, 7, 7, 1, false>(),  make_persist<32, 16, 1, 8, 11, 11, 1, false>(),
    make_persist<64, 8, 2, 4, 3, 3, 2, false>(),   make_persist<64, 8, 2, 4, 7, 7, 1, false>(),   make_persist<64, 8, 2, 4, 11, 11, 1, false>(),
    // C = 128: k = 3 on 768 threads (12 waves of 32 x 64); k = 7 and k = 11 on 512 threads with N1 = 192 -- eight waves of 32 x 96
    // (WN = 3: eight fragment reads per nine product blocks), 220-235 registers: -6 ... -10 % / -3.5 % against the forms above them
    make_persist<128, 6, 4, 3, 3, 3, 1, false>(),  make_persist<128, 6, 4, 2, 7, 4, 1, false>(),  make_persist<128, 6, 4, 2, 11, 4, 1, false>(),
    // A/B forms: 11 = EARLY requests; C = 128, k = 11 on 768 threads.  (12 = C = 128 on 512 threads with N1 = 128: +10 %, removed)
    make_persist<32, 16, 1, 8, 3, 3, 2, true, 11>(),  make_persist<32, 16, 1, 8, 7, 7, 1, true, 11>(),  make_persist<32, 16, 1, 8, 11, 11, 1, true, 11>(),
    make_persist<64, 8, 2, 4, 3, 3, 2, true, 11>(),   make_persist<64, 8, 2, 4, 7, 7, 1, true, 11>(),   make_persist<64, 8, 2, 4, 11, 11, 1, true, 11>(),
    make_persist<128, 6, 4, 3, 11, 4, 1, false, 11>(),  make_persist<128, 4, 4, 2, 7, 7, 1, false, 11>(),
    // (variant 15, round 5: persistent AND two workgroups per CU on half-width tiles: slower on all six C <= 64 shapes, removed)
    // variant 13 (round 5 A/B): C = 128, k = 3 on the 512-thread N1 = 192 tile that k = 7 / 11 use by default (slower at k = 3: 0.292 vs 0.278 ms)
    make_persist<128, 6, 4, 2, 3, 3, 1, false, 13>(),
    // RVCX_PAIR_VARIANT=3: the round-2 form (residual fetched eight values at a time inside the epilogue), for A/B runs
    make_cfg<32, 16, 1, 8, 3, 3, 2, 3>(),  make_cfg<32, 16, 1, 8, 7, 7, 1, 3>(),  make_cfg<32, 16, 1, 8, 11, 11, 1, 3>(),
    make_cfg<64, 8, 2, 4, 3, 3, 2, 3>(),   make_cfg<64, 8, 2, 4, 7, 7, 1, 3>(),   make_cfg<64, 8, 2, 4, 11, 11, 1, 3>(),
    make_cfg<128, 6, 4, 3, 3, 3, 1, 3>(),  make_cfg<128, 4, 4, 2, 7, 7, 1, 3>(),  make_cfg<128, 6, 4, 3, 11, 4, 1, 3>(),
};
constexpr int kPairBase = 12;
// rows 16-byte aligned: what the wide epilogue needs
bool pair_aligned(const PairArgs& a) {
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  return a.cs % 4 == 0 && a.T % 4 == 0 && a.bs % 4 == 0 && al(a.x) && al(a.y) && al(a.y2);
}

const PairCfg* find_cfg(const PairArgs& a) {
  const int C = a.C, K = a.k;
  static const int variant = getenv("RVCX_PAIR_VARIANT") ? atoi(getenv("RVCX_PAIR_VARIANT")) : 0;
  static const bool c256 = !getenv("RVCX_PAIR_C256") || atoi(getenv("RVCX_PAIR_C256")) != 0;
  if (C == 256 && (!c256 || K != 3)) return nullptr;     // k = 7 / 11 at C = 256: the two launches are faster (bench_pair)
  static const bool persist = !getenv("RVCX_PAIR_PERSIST") || atoi(getenv("RVCX_PAIR_PERSIST")) != 0;
  if (variant == 0 && persist && pair_aligned(a))
    for (int i = kPairBase; i < (int)(sizeof(kPair) / sizeof(kPair[0])); ++i)
      if (kPair[i].variant == 10 && kPair[i].C == C && kPair[i].K == K) return &kPair[i];
  const int want = variant != 0 ? variant : (pair_aligned(a) ? 4 : 0);
  if (want != 0)
    for (int i = kPairBase; i < (int)(sizeof(kPair) / sizeof(kPair[0])); ++i)
      if (kPair[i].variant == want && kPair[i].C == C && kPair[i].K == K && (!kPair[i].wide || pair_aligned(a)))
        return &kPair[i];
  for (int i = 0; i < kPairBase; ++i)
    if (kPair[i].C == C && kPair[i].K == K) return &kPair[i];
  return nullptr;
}

}  // namespace

bool resblock_pair_enabled() {
  static const int mode = getenv("RVCX_FUSE") ? atoi(getenv("RVCX_FUSE")) : 1;
  return mode != 0 && conv_h3_enabled();
}

bool resblock_pair_ok(const PairArgs& a) {
  if (!resblock_pair_enabled() || !a.w1 || !a.w2) return false;
  if (a.dil < 1 || a.dil > 5 || (a.k - 1) * a.dil > kPairHalo - 14) return false;
  if ((long)a.C * a.cs * 4 >= kH3Oob || (long)a.k * a.C * a.C * 4 >= kH3Oob) return false;
  return find_cfg(a) != nullptr;
}

int resblock_pair_slot(int C) { return C == 32 ? 50 : (C == 64 ? 51 : (C == 128 ? 52 : 58)); }

void resblock_pair_describe(ConvProfile* p) {
  for (int i = 0; i < kPairBase; ++i) {
    const auto& c = kPair[i];
    const int s = resblock_pair_slot(c.C);
    p->bm[s] = c.C;
    p->bn[s] = c.n1;
    p->halo[s] = 500000 + c.C;
  }
}

void resblock_pair_init() {      // once per device, also when several contexts start on different threads
  static std::mutex mu;
  static uint64_t done = 0;
  int dev = 0;
  RVCX_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> g(mu);
  if ((done >> (dev & 63)) & 1) return;
  for (const auto& c : kPair)
    RVCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(c.kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)c.lds));
  done |= 1ull << (dev & 63);
}

void launch_resblock_pair(const PairArgs& a, hipStream_t stream) {
  RVCX_CHECK(resblock_pair_ok(a), "resblock pair: unsupported shape");
  RVCX_CHECK(a.y != a.x && a.y2 != a.x, "resblock pair: in-place operation is a race (halo reads vs neighbours' stores)");
  resblock_pair_init();
  const PairCfg& c = *find_cfg(a);
  const int bn_out = c.bn_out;
  dim3 grid(cdiv(a.T, bn_out), 1, a.B);
  if (c.persist) {
    static const int ncu = [] {
      int dev = 0, n = 256;
      if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
      return std::max(8, n);
    }();
    grid.x = std::min<unsigned>(grid.x, (unsigned)(ncu * c.per_cu));
  }
  static const int xcd = getenv("RVCX_PAIR_XCD") ? atoi(getenv("RVCX_PAIR_XCD")) : 1;
  PairArgs b = a;
  b.xcd_order = xcd;
  hipLaunchKernelGGL(c.kern, grid, dim3(c.threads), c.lds, stream, b);
  RVCX_HIP(hipGetLastError());
}

}  // namespace rvcx
