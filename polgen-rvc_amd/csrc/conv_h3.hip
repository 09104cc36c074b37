// Dense stride-1 1-D conv on the fp16 matrix cores with fp32-grade products ("h3": hi/lo split, 3 MFMAs).
//
// The fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the fp16 rate.  Every fp32 operand is split into two
// fp16 numbers, w = wh + wl, x = xh + xl (11 + 11 significant bits, |error| <= 2^-22 |w|), and
//     w x  ~=  wh xh + wh xl + wl xh                       (the dropped wl xl term is 2^-22 relative)
// Each fp16 x fp16 product is exact in fp32 and the sums accumulate in fp32 inside v_mfma_f32_32x32x16_f16, so the
// result carries ~2^-21 relative error per product -- the size of fp32 rounding noise, not of fp16.  To keep the
// low parts out of the fp16 subnormal range everything is computed at scale S = 256:
//     S w x ~= (S wh) xh + wh (S xl) + (S wl) xh           operands A: {S wh, wh = (S wh)/S, S wl}   B: {xh, S xl}
// and the epilogue multiplies by 1/S (exact).  Weights are split once at load (ctx.hip: pack_h3); activations are
// split while they are committed to LDS.  3 MFMAs of 32 cycles replace 8 of 64: 5.3x fewer matrix-pipe cycles,
// which moves the NSF decoder's ResBlock convs from the MFMA roofline to the HBM one.
//
// LDS images (16-byte elements = 8 halves = the k-slice a lane feeds to one MFMA):
//   As[kkl][op {S wh, S wl}][h][co]   (wh itself is re-derived in registers: one LDS read and a third of the weight
//                        image saved)   co contiguous: lane (i, h) reads element (h, co0 + i)      -> linear, conflict-free
//   Bs[op][h][p]         p  contiguous: lane (j, h) reads element (h, n + j + tap)  -> taps are plain offsets
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "conv.h"
#include "conv_device.h"
#include "h3_device.h"

#ifdef RVCX_ABLATION
#define H3_DBG(a, bit) ((a).dbg & (bit))
#else
#define H3_DBG(a, bit) 0
#endif

namespace rvcx {

// HALO: input-tile halo (64: 1-D k <= 11 d <= 5; 320: 3x3 on row-padded maps).  STRIDE 2: HuBERT extractor.
// LIN: k = 1 layers -- a stage is KKT 16-channel chunks (each with its own input tile) instead of KKT taps.
// XS: the input arrives pre-split (ConvArgs::x_split) -- a separate instantiation so that the fp32-input staging
// code keeps its straight-line, batched loads.
template <int BM, int BN, int WR, int WC, int KKT, int HALO, int STRIDE, bool LIN, bool XS = false>
// (min waves per SIMD: 3 = at most 168 VGPRs.  The large Linear tiles and the 32x512 3x3 tile need more registers --
// forced under 168 they spilled 68..213 of them -- and run two waves per SIMD instead.)
__global__ __launch_bounds__(256, ((LIN && BM * BN >= 8192) || BN >= 512) ? 2 : 3) void conv_h3_kernel(const ConvArgs a) {
  constexpr int WM = BM / (32 * WR), WN = BN / (32 * WC);
  constexpr int WROW = LIN ? BN : BN * STRIDE + HALO;
  constexpr int NBTILE = LIN ? KKT : 1;               // input tiles resident per stage
  constexpr int A_ELEMS = KKT * 2 * 2 * BM;           // 16-byte elements per stage: ops {S wh, S wl}
  constexpr int NA = (A_ELEMS + 255) / 256;
  constexpr int B_TASKS = NBTILE * 2 * WROW;          // (tile, h, position): 8 channels each
  constexpr int NBT = (B_TASKS + 255) / 256;
  static_assert(WR * WC == 4, "bad tile");
  __shared__ uint4 As[A_ELEMS];
  __shared__ uint4 Bs[NBTILE * 2 * 2 * WROW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WC, wc = wave % WC;
  const int i = lane & 31, h = lane >> 5;
  // blockIdx.z = (item, group, K split).  A group of a grouped layer (HuBERT's pos_conv: 16 x {48 -> 48, 128 taps}) is a dense
  // layer of its own on Cin_g input rows, its own weight image and Cout_g output channels; grouped launches never split K.
  const int bz = blockIdx.z / a.splitk, ks = blockIdx.z - bz * a.splitk;
  const int b = bz / a.groups, grp = bz - b * a.groups;
  const int co0 = blockIdx.y * BM;
  const int n0 = blockIdx.x * BN;
  const int len_in = a.lens_in ? a.lens_in[b] : a.Tin;
  const int in_base = n0 * STRIDE + a.off_min;
  const int wuse = LIN ? BN : BN * STRIDE + a.wrow;
  const int pre_act = a.pre_act;
  const float pre_slope = a.pre_slope;
  const int nchunk = a.Cin_gp / 16;
  constexpr bool xsplit = XS;
  static_assert(!XS || !LIN, "pre-split input: tap tiles only");
  const H3Rsrc xr = xsplit ? h3_rsrc(static_cast<const char*>(a.x_split) + (long)b * a.x_bs * 4, a.Cin_g * a.x_cs * 4)
                           : h3_rsrc(a.x + (long)b * a.x_bs + (long)grp * a.Cin_g * a.x_cs, a.Cin_g * a.x_cs * 4);
  const H3Rsrc wr_ = h3_rsrc(static_cast<const char*>(a.w_h3) + (long)grp * a.ksize * nchunk * 4 * a.Cout_gp * 16,
                             a.ksize * nchunk * 4 * a.Cout_gp * 16);

  f32x16 acc[WM][WN];
#pragma unroll
  for (int m = 0; m < WM; ++m)
#pragma unroll
    for (int n = 0; n < WN; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  // chunk-invariant addressing
  int a_kkl[NA], a_off[NA];     // element e = tid + 256 j -> (kkl, op, h, co); byte offset inside a (kk, chunk) slab
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int e = tid + 256 * j;
    const int co = e % BM, rest = e / BM;              // rest = (kkl*2 + op)*2 + h
    a_kkl[j] = rest / 4;
    const bool ok = e < A_ELEMS && co0 + co < a.Cout_gp;
    a_off[j] = ok ? ((rest % 4) * a.Cout_gp + co0 + co) * 16 : kH3Oob;
  }
  int b_off[NBT], b_row[NBT];    // task t = tid + 256 j -> (tile, h, p): position byte offset, first channel row
#pragma unroll
  for (int j = 0; j < NBT; ++j) {
    const int t = tid + 256 * j;
    const int th = t / WROW, p = t - th * WROW;        // th = tile*2 + h
    const int pos = in_base + p;
    b_row[j] = (th >> 1) * 16 + (th & 1) * 8;
    b_off[j] = (t < B_TASKS && p < wuse && pos >= 0 && pos < len_in) ? pos * 4 : kH3Oob;
  }
  // pre-split input: task t = tid + 256 j -> (op*2 + h, p): one 16-byte element per task, no conversion at commit
  constexpr int NBS = LIN ? 1 : (4 * WROW + 255) / 256;
  static_assert(LIN || NBS * 4 <= NBT * 8, "split-input staging must fit the register set of the fp32 path");
  const int xrow = a.x_cs * 4;
  const int slab = 4 * a.Cout_gp * 16;                 // bytes of one (kk, chunk) weight slab
  // this split's chunk range, and the stage structure: LIN -> stages of KKT chunks; else (chunk, tap-group)
  const int cb0 = ks * nchunk / a.splitk, cb1 = (ks + 1) * nchunk / a.splitk;
  const int nkk = LIN ? 1 : (a.ksize + KKT - 1) / KKT;
  // ragged batches: a tile that lies entirely behind its item's last valid output has nothing to compute -- no stages,
  // and the epilogue stores the zeros it stores for every position beyond lens_out (workgroup-uniform)
  const int len_out = a.lens_out ? a.lens_out[b] : 0x7fffffff;
  const bool dead = (a.out_mode == OUT_SHUF1D ? n0 * a.sh_s - a.sh_pad : n0) >= len_out;
  const int nst = dead ? 0 : (LIN ? (cb1 - cb0 + KKT - 1) / KKT : (cb1 - cb0) * nkk);

  uint4 ra[NA];
  float rb[NBT][8];
  bool ovf = false;                                    // an activation that does not fit fp16 (see kH3ActLimit)
  auto fetch_b = [&](int chunk) {                      // LIN: chunks chunk .. chunk+KKT-1 (clipped to cb1)
    if constexpr (xsplit) {
      const int cbyte = chunk * 4 * a.x_cs * 16;
#pragma unroll
      for (int j = 0; j < NBS; ++j) {
        const int t = tid + 256 * j;                     // offsets recomputed per chunk: cheaper than 5 live registers
        const int oph = t / WROW, p = t - oph * WROW;
        const int pos = in_base + p;
        const bool ok = t < 4 * WROW && p < wuse && pos >= 0 && pos < len_in;
        const uint4 v = h3_load4(xr, ok ? cbyte + (oph * a.x_cs + pos) * 16 : kH3Oob);
        rb[(4 * j + 0) / 8][(4 * j + 0) % 8] = __builtin_bit_cast(float, v.x);   // compile-time indices: registers
        rb[(4 * j + 1) / 8][(4 * j + 1) % 8] = __builtin_bit_cast(float, v.y);
        rb[(4 * j + 2) / 8][(4 * j + 2) % 8] = __builtin_bit_cast(float, v.z);
        rb[(4 * j + 3) / 8][(4 * j + 3) % 8] = __builtin_bit_cast(float, v.w);
      }
    } else {
#pragma unroll
    for (int j = 0; j < NBT; ++j) {
      const int crow = chunk * 16 + b_row[j];
      const bool cok = !LIN || crow < cb1 * 16;
      const int row0 = crow * xrow;                    // rows >= Cin_g read as 0 (beyond num_records)
#pragma unroll
      for (int q = 0; q < 8; ++q)
        rb[j][q] = h3_load1(xr, (b_off[j] == kH3Oob || !cok) ? kH3Oob : row0 + q * xrow + b_off[j]);
    }
    }
  };
  auto fetch_a = [&](int chunk, int kk0) {
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int kk = LIN ? 0 : kk0 + a_kkl[j];
      const int ch = LIN ? chunk + a_kkl[j] : chunk;
      const bool ok = a_off[j] != kH3Oob && (LIN ? ch < cb1 : kk < a.ksize);
      ra[j] = h3_load4(wr_, ok ? (kk * nchunk + ch) * slab + a_off[j] : kH3Oob);
    }
  };
  auto commit = [&](int kk0) {
    if constexpr (xsplit) {
      if (kk0 == 0) {
#pragma unroll
        for (int j = 0; j < NBS; ++j) {
          const int t = tid + 256 * j;
          if (NBS * 256 == 4 * WROW || t < 4 * WROW) {
            uint4 v;
            v.x = __builtin_bit_cast(unsigned, rb[(4 * j + 0) / 8][(4 * j + 0) % 8]);
            v.y = __builtin_bit_cast(unsigned, rb[(4 * j + 1) / 8][(4 * j + 1) % 8]);
            v.z = __builtin_bit_cast(unsigned, rb[(4 * j + 2) / 8][(4 * j + 2) % 8]);
            v.w = __builtin_bit_cast(unsigned, rb[(4 * j + 3) / 8][(4 * j + 3) % 8]);
            Bs[t] = v;                                   // Bs index (op*2 + h) * WROW + p == t
          }
        }
      }
    } else if (LIN || kk0 == 0) {
#pragma unroll
      for (int j = 0; j < NBT; ++j) {
        const int t = tid + 256 * j;
        if (NBT * 256 == B_TASKS || t < B_TASKS) {
          half8 hi, lo;
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            float v = rb[j][q];
            if (pre_act == ACT_LRELU) v = v > 0.f ? v : v * pre_slope;
            ovf |= !(fabsf(v) < kH3ActLimit);
            const _Float16 vh = (_Float16)v;
            hi[q] = vh;
            lo[q] = (_Float16)((v - (float)vh) * kH3Scale);
          }
          const int th = t / WROW, p = t - th * WROW;
          const int tile = th >> 1, hh = th & 1;
          Bs[(tile * 4 + hh) * WROW + p] = __builtin_bit_cast(uint4, hi);
          Bs[(tile * 4 + 2 + hh) * WROW + p] = __builtin_bit_cast(uint4, lo);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NA; ++j)
      if (NA * 256 == A_ELEMS || tid + 256 * j < A_ELEMS) As[tid + 256 * j] = ra[j];
  };

  int chunk = cb0, kk0 = 0;
  if (nst > 0) {
    fetch_b(cb0);
    fetch_a(cb0, 0);
  }
  for (int st = 0; st < nst; ++st) {
    if (!H3_DBG(a, 32) || st == 0) {
      __syncthreads();
      commit(kk0);
      __syncthreads();
    }
    int kk1 = kk0, chunk1 = chunk;
    if (LIN) {
      chunk1 += KKT;
    } else {
      kk1 += KKT;
      if (kk1 >= a.ksize) {
        kk1 = 0;
        chunk1 += 1;
      }
    }
    if (!H3_DBG(a, 16)) {
      if (st + 1 < nst) fetch_a(chunk1, kk1);
      if (LIN) {
        if (st + 1 < nst) fetch_b(chunk1);
      } else if (kk0 == 0 && chunk + 1 < cb1) {
        fetch_b(chunk + 1);
      }
    }
#pragma unroll
    for (int kkl = 0; kkl < KKT; ++kkl) {
      const int kk = kk0 + kkl;
      if ((LIN ? (chunk + kkl < cb1) : (KKT == 1 || kk < a.ksize)) && !H3_DBG(a, 4)) {
        const int tp = LIN ? 0 : (kk / a.kw) * a.rowpitch + (kk % a.kw) * a.dil - a.pad - a.off_min;
        const uint4* Bt = Bs + (LIN ? kkl * 4 * WROW : 0);
        half8 af[3][WM], bf[2][WN];     // af[0] = S wh and af[2] = S wl from LDS; af[1] = wh = af[0] / S in registers
#pragma unroll
        for (int m = 0; m < WM; ++m) {
          af[0][m] = __builtin_bit_cast(half8, As[((kkl * 2 + 0) * 2 + h) * BM + wr * (WM * 32) + m * 32 + i]);
          af[2][m] = __builtin_bit_cast(half8, As[((kkl * 2 + 1) * 2 + h) * BM + wr * (WM * 32) + m * 32 + i]);
          af[1][m] = af[0][m] * (_Float16)(1.f / kH3Scale);   // exact: a power-of-two scaling (4 v_pk_mul_f16)
        }
#pragma unroll
        for (int op = 0; op < 2; ++op)
#pragma unroll
          for (int n = 0; n < WN; ++n)
            bf[op][n] = __builtin_bit_cast(half8, Bt[(op * 2 + h) * WROW + (wc * (WN * 32) + n * 32 + i) * STRIDE + tp]);
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
          for (int n = 0; n < WN; ++n) {
            acc[m][n] = h3_mfma(af[0][m], bf[0][n], acc[m][n]);   // (S wh) xh
            acc[m][n] = h3_mfma(af[1][m], bf[1][n], acc[m][n]);   // wh (S xl)
            acc[m][n] = h3_mfma(af[2][m], bf[0][n], acc[m][n]);   // (S wl) xh
          }
      }
    }
    kk0 = kk1;
    chunk = chunk1;
  }

  if (ovf) report_h3_overflow(a.ovf, a.ovf_layer, a.seq);
  constexpr float inv = 1.f / kH3Scale;
#pragma unroll
  for (int m = 0; m < WM; ++m)
#pragma unroll
    for (int n = 0; n < WN; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] *= inv;

  const int co_w = co0 + wr * (WM * 32) + 4 * h, nn_w = n0 + wc * (WN * 32) + i;
  if (H3_DBG(a, 8)) return;
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
  if (a.y_split) {
    const int c_t = co0 + wr * (WM * 32);
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
      for (int n = 0; n < WN; ++n) store_tile_split(a, b, c_t + m * 32, nn_w + n * 32, h, acc[m][n], len_out);
  } else if (fast_epilogue_ok(a) && a.groups == 1) {
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
      for (int n = 0; n < WN; ++n) store_tile_fast(a, b, co_w + m * 32, nn_w + n * 32, acc[m][n], len_out);
  } else if (a.out_mode == OUT_SHUF1D) {
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
      for (int n = 0; n < WN; ++n) store_tile_shuf1d(a, b, co_w + m * 32, nn_w + n * 32, acc[m][n], len_out);
  } else {
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
      for (int n = 0; n < WN; ++n) store_tile(a, b, grp, co_w + m * 32, nn_w + n * 32, acc[m][n], len_out);
  }
}

namespace {
struct H3Cfg {
  int bm, bn, halo, stride;   // halo 0 + lin: k = 1 layers
  bool lin;
  float ovh;                  // per-block prologue + epilogue in 16-channel k-steps of MFMA time (cost model)
  float eff;                  // relative main-loop efficiency (wave tile 32x64 reuses fragments, 32x32 does not)
  void (*kern)(const ConvArgs);
  void (*kern_xs)(const ConvArgs);   // same tile reading a pre-split input (null: none)
};
const H3Cfg kH3[] = {
    // 1-D, halo <= 64.  Round 4 (tools/sweep_tiles_1d.py, profiles/sweep_tiles_1d_r04.txt): the 64 x 64 tile beats both wide
    // tiles on EVERY 1-D shape of the path -- 1.1x on the C = 256 ResBlock convs, 1.4-1.6x on conv_pre / the first two
    // upsamplers / C = 128 k = 7, 1.9-2.4x on the thin upsamplers and C <= 64 (64 -> 64, 2 taps, 767 k positions: 175 against
    // 427 us) -- and the round-2 efficiencies (1.00 / 0.97 / 0.75, fitted before the staging and epilogue rewrites) sent most
    // of them to the wide ones.  The wide tiles stay selectable (RVCX_CONV_TILE, tested bit-identical) at efficiencies that
    // keep them out of the automatic choice; the 64 x 64 entry keeps its own number, so its split-K decisions are unchanged.
    {32, 256, 64, 1, false, 18.f, 0.40f, conv_h3_kernel<32, 256, 1, 4, 4, 64, 1, false>, conv_h3_kernel<32, 256, 1, 4, 4, 64, 1, false, true>},
    {64, 128, 64, 1, false, 16.f, 0.45f, conv_h3_kernel<64, 128, 2, 2, 2, 64, 1, false>, conv_h3_kernel<64, 128, 2, 2, 2, 64, 1, false, true>},
    {64, 64, 64, 1, false, 16.f, 0.75f, conv_h3_kernel<64, 64, 2, 2, 4, 64, 1, false>, conv_h3_kernel<64, 64, 2, 2, 4, 64, 1, false, true>},
    // 3x3 on row-padded maps
    {32, 128, 320, 1, false, 22.f, 1.00f, conv_h3_kernel<32, 128, 1, 4, 4, 320, 1, false>, conv_h3_kernel<32, 128, 1, 4, 4, 320, 1, false, true>},
    {64, 64, 320, 1, false, 20.f, 0.80f, conv_h3_kernel<64, 64, 2, 2, 4, 320, 1, false>, conv_h3_kernel<64, 64, 2, 2, 4, 320, 1, false, true>},
    // stride 2
    {64, 128, 64, 2, false, 18.f, 0.50f, conv_h3_kernel<64, 128, 2, 2, 2, 64, 2, false>, conv_h3_kernel<64, 128, 2, 2, 2, 64, 2, false, true>},   // (the same finding: 64 x 64 is 1.25x faster on the extractor's layers)
    {64, 64, 64, 2, false, 16.f, 0.80f, conv_h3_kernel<64, 64, 2, 2, 4, 64, 2, false>, conv_h3_kernel<64, 64, 2, 2, 4, 64, 2, false, true>},
    // k = 1
    {64, 64, 0, 1, true, 16.f, 0.95f, conv_h3_kernel<64, 64, 2, 2, 4, 0, 1, true>, nullptr},
    {128, 64, 0, 1, true, 18.f, 1.00f, conv_h3_kernel<128, 64, 4, 1, 2, 0, 1, true>, nullptr},
    // wide 3x3 tiles for the shallow U-Net levels (C = 16/32): the 2*Wp+2 halo is amortised over more outputs
    {32, 256, 320, 1, false, 22.f, 1.10f, conv_h3_kernel<32, 256, 1, 4, 4, 320, 1, false>, conv_h3_kernel<32, 256, 1, 4, 4, 320, 1, false, true>},
    {32, 512, 320, 1, false, 22.f, 1.15f, conv_h3_kernel<32, 512, 1, 4, 4, 320, 1, false>, conv_h3_kernel<32, 512, 1, 4, 4, 320, 1, false, true>},
    // Round 4 measured four more forms of the small-channel 3x3 tile (tools/sweep_unet.py, U-Net levels 0 / 1, B = 1 and 8)
    // and kept none: 32 x 512 on EIGHT waves (the 2 Wp + 2 halo staged once per 512 outputs instead of once per 128: 51 / 317
    // us against 41 / 290 for 32 x 128), 32 x 128 with min-waves 4 (the tile already runs four workgroups per CU: 104 VGPRs,
    // 37 KB of LDS), all nine taps of a chunk in one stage on 4 and on 8 waves (two barriers instead of six: slower than
    // both), and the 64 x 64 tile forced on the deep levels (faster in the isolated sweep at S = 1, nothing in the model,
    // where the per-item split-K factor stands).  These launches are bound by the latency chain of a workgroup (load ->
    // convert -> LDS -> 27 MFMAs -> store, ~10 us of life for 128 outputs), not by staging volume, occupancy or the matrix
    // pipe: what would help is a persistent kernel that prefetches tile t + 1 under tile t, or a fused ConvBlockRes.
    // Deep levels (C >= 128, <= 7272 positions) at B = 1: ~28 us per launch whatever the split-K factor (1 ... 32) or the
    // taps per stage (a nine-tap 64 x 64 tile: 24.6 / 26.5 / 29.3 us on levels 3 / 4 / 5 against 28.3 / 28.3 / 29.5): with
    // one wave per SIMD a k-step costs ~0.25 us of dependent LDS-read -> MFMA latency, five times its matrix time.
};
constexpr int kNumH3 = sizeof(kH3) / sizeof(kH3[0]);
// profile slots: tiles 0 .. 10 -> 39 .. 49 (50 .. 63 belong to the fused steps, the Linear tiles and the streaming kernels),
// later tiles -> 28 .. 38 (8 .. 27: the fp32 conv_fast family)
constexpr int h3_slot(int t) { return t < 11 ? 39 + t : 28 + (t - 11); }
static_assert(kNumH3 <= 22, "conv_h3: out of profile slots");
}  // namespace

thread_local bool g_force_fp32 = false;
thread_local bool g_gru_no_cluster = false;
thread_local int g_gru_drop_member = 0;

bool conv_h3_enabled() {
  static const int mode = getenv("RVCX_H3") ? atoi(getenv("RVCX_H3")) : 1;   // initialised once, thread-safe
  return mode != 0 && !g_force_fp32;
}

bool conv_h3_configured() {
  const bool saved = g_force_fp32;
  g_force_fp32 = false;
  const bool r = conv_h3_enabled();
  g_force_fp32 = saved;
  return r;
}

void conv_h3_describe(ConvProfile* p) {
  for (int t = 0; t < kNumH3; ++t) {
    const int s = h3_slot(t);
    p->bm[s] = kH3[t].bm;
    p->bn[s] = kH3[t].bn;
    p->halo[s] = 400000 + (kH3[t].lin ? 1 : (kH3[t].stride == 2 ? 2 : kH3[t].halo));
  }
}

bool conv_h3_split_ok(const ConvArgs& a) {
  if (!a.w_h3 || !conv_h3_enabled()) return false;
  if (a.groups != 1 || a.ksize == 1 || a.out_mode != OUT_NORMAL) return false;
  if (a.stride != 1 && !(a.stride == 2 && a.kw == a.ksize)) return false;      // stride 2: the HuBERT extractor's 1-D layers
  if (a.Cin_g % 16 != 0 || a.Cout_g % 16 != 0 || a.Cin_g != a.Cin_gp) return false;
  {
    int lo = 1 << 30, hi = -(1 << 30);
    for (int kk = 0; kk < a.ksize; ++kk) {
      lo = std::min(lo, conv_tap_off(a, kk));
      hi = std::max(hi, conv_tap_off(a, kk));
    }
    if (hi - lo > 320) return false;
  }
  if ((long)a.Cin_gp * a.x_cs * 4 >= kH3Oob || (long)a.Cout_g * a.y_cs * 4 >= kH3Oob) return false;
  return true;
}

// returns a profile slot (>= 0) when the launch was taken, -1 otherwise
int launch_conv_h3(ConvArgs& a, int halo, int off_min, hipStream_t stream) {
  const bool split = a.x_split || a.y_split;
  RVCX_CHECK(!split || conv_h3_split_ok(a), "conv_h3: pre-split activations on a launch that cannot take them");
  if (!a.w_h3 || !conv_h3_enabled()) return -1;
  if (a.Cin_gp % 16 != 0 || (a.stride != 1 && !(a.stride == 2 && a.kw == a.ksize))) return -1;
  if (a.groups != 1 && (split || a.ksize == 1 || a.stride != 1 || a.out_mode != OUT_NORMAL || a.Cin_g != a.Cin_gp)) return -1;
  if ((long)a.Cin_gp * a.x_cs * 4 >= kH3Oob || (long)a.ksize * a.Cin_gp * a.Cout_gp * 4 >= kH3Oob) return -1;
  const bool lin = a.ksize == 1 && a.stride == 1;
  const int nchunk = a.Cin_gp / 16;
  // cost model in the shape of conv_fast's: the three fp16 MFMAs of a (tap, 16-channel) k-step cost 96 matrix-pipe
  // cycles for 32x32x16 MACs; fitted on the NSF shapes: a CU retires ~39 k-steps of a 32x32 tile per microsecond with
  // >= 4 blocks resident, a block's prologue + epilogue cost ~18 k-steps per 32x32 tile
  // Batch invariance: the split-K factor changes the fp32 summation order, so it is decided for ONE batch item
  // (exactly the decision a single-utterance run takes); the tile -- which does not change the order of the k-loop
  // -- is then chosen for the real batch under that factor.  A batched conversion is therefore bit-identical to
  // converting its utterances one by one.
  const long cap_item = a.part_cap_item > 0 ? a.part_cap_item : a.part_cap;
  // The 32 x 256 / 32 x 512 3x3 tiles stay selectable by override (and tested) but out of the automatic choice: measured
  // on the U-Net level-0 shape 77 / 111 us against 40 us for 32 x 128 (tools/bench_unet_l0.py); with them C3 ran 5 % slower
  static const bool wide3x3 = getenv("RVCX_WIDE3X3") && atoi(getenv("RVCX_WIDE3X3")) != 0;
  auto select = [&](int Bsel, int forcedS, int* tile_out, int* s_out) {
    int best = -1, S = 1;
    double best_t = 1e300;
    for (int t = 0; t < kNumH3; ++t) {
      const H3Cfg& F = kH3[t];
      if (g_conv_override.tile >= 100 && g_conv_override.tile - 100 != t) continue;
      if (F.lin != lin || F.stride != a.stride) continue;
      if (a.x_split && !F.kern_xs) continue;
      if (!wide3x3 && g_conv_override.tile < 100 && F.halo == 320 && F.bn >= 256) continue;
      if (!lin && (halo > F.halo || (F.halo == 320 && halo <= 64))) continue;
      const long blocks = (long)cdiv(a.Cout_gp, F.bm) * cdiv(a.Nout, F.bn) * Bsel * a.groups;
      const double ksteps = (double)a.ksize * nchunk;
      for (int s = 1; s <= ((split || a.groups != 1) ? 1 : 8); s *= 2) {
        if (s > 1 && (!a.part || nchunk / s < 1 || ksteps / s < 8.0 || (long)s * a.Cout_g * a.Nout > cap_item ||
                      (long)s * a.B * a.Cout_g * a.Nout > a.part_cap))
          break;
        if (g_conv_override.splitk > 0 && g_conv_override.splitk != s) continue;
        if (forcedS > 0 && s != forcedS) continue;
        const double c = (double)blocks * s / 256.0;
        const double f = std::min(1.0, 0.45 + 0.55 * (std::max(c, 1.0) - 1.0) / 3.0);
        double us = std::ceil(c) * (F.bm / 32) * (F.bn / 32) * (ksteps / s + F.ovh) / (39.0 * F.eff * f);
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
  static const int tile2_mask = getenv("RVCX_TILE2_MASK") ? atoi(getenv("RVCX_TILE2_MASK")) : 255;
  const int cls = (a.out_mode != OUT_NORMAL) ? 8 : (lin ? 2 : (halo > 64 ? 1 : 4));
  int cls2 = cls;
  if (a.x_split || a.y_split) cls2 |= 16;
  const bool tile_for_batch = (tile2_mask & cls) && (!(cls2 & 16) || (tile2_mask & 16));
  if (best >= 0 && a.B > 1 && tile_for_batch) {
    int b2 = -1, s2 = 1;
    select(a.B, S, &b2, &s2);
    if (b2 >= 0 && s2 == S) best = b2;
  }
  if (best < 0) return -1;
  const H3Cfg& F = kH3[best];
  a.off_min = off_min;
  a.wrow = halo;
  a.splitk = S;
  {
    static int dbg = -1;
    if (dbg < 0) dbg = getenv("RVCX_CONV_DBG") ? atoi(getenv("RVCX_CONV_DBG")) : 0;
    a.dbg = dbg;
  }
  dim3 grid(cdiv(a.Nout, F.bn), cdiv(a.Cout_gp, F.bm), a.B * a.groups * S);
  {
    static const bool log = getenv("RVCX_CONV_LOG") != nullptr;
    if (log)
      fprintf(stderr, "h3 tile %d (%dx%d halo %d) S %d grid %u %u %u B %d Cin %d Cout %d k %d Nout %d xs %d ys %d\n", best,
              F.bm, F.bn, F.halo, S, grid.x, grid.y, grid.z, a.B, a.Cin_g, a.Cout_g, a.ksize, a.Nout, a.x_split != nullptr,
              a.y_split != nullptr);
  }
  hipLaunchKernelGGL(a.x_split ? F.kern_xs : F.kern, grid, dim3(256), 0, stream, a);
  if (S > 1) launch_splitk_finish(a, stream);
  RVCX_HIP(hipGetLastError());
  return h3_slot(best);   // profile slots (conv_h3_describe)
}

}  // namespace rvcx
