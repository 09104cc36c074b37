// conv_ws: a WEIGHT-STATIONARY large-tile form of the split-fp16 conv for the layers the 64 x 64 tile serves worst --
// the deep levels of the RMVPE U-Net (rvc/lib/predictors/RMVPE.py:140-320: 3 x 3 convs with 64 ... 512 channels on 28 288 ...
// 624 positions and their 2 x 2 polyphase ConvTranspose2d, ~90 launches per clip, each on a 24 - 41 us floor whatever its
// shape) -- round 6, replaces round 5's conv_deep_kernel (split-K across the waves of a 64 x 32 tile, both operands
// straight from global memory: measured slower, it re-read 380 MB of operands per conv through the L2 -> CU path).
//
// What bounded those launches (LABNOTES "Round 5", conv_h3.hip's tile table): a 64 x 64 output tile needs (64 + 64) x K
// operand values, so a K = 4608 conv over 624 positions moves 190 MB from L2 to the CUs for 2.9 GFLOP; a wave owns ONE 32 x 32
// accumulator, i.e. four LDS fragment reads and three DEPENDENT MFMAs per k-step (1.33 reads per MFMA: LDS-bound, and with one
// wave per SIMD every k-step is a serial read -> MFMA -> MFMA -> MFMA chain); a stage is 12 MFMAs between two barriers.
// Here a workgroup owns 64 output channels x 320 positions (four waves as 2 x 2, a wave = 32 channels x FIVE 32-position
// blocks): per k-step two weight-fragment reads + ten input-fragment reads feed fifteen MFMAs on five independent
// accumulators (0.8 reads per MFMA); a stage is one 16-channel chunk x ALL taps (135 MFMAs per wave at 3 x 3) in a
// double-buffered LDS ring, one barrier per stage, the next stage's operands requested before the stage's MFMAs start.
// Operand traffic per conv falls 6x (weights once per 320 positions instead of once per 64).
//
// K is cut into S segments of nchunk / S chunks -- S a function of the layer's SHAPE only (never of the batch size: the
// summation order is part of the result): each segment is accumulated from zero, the segment sums are added left to
// right.  A launch that fills the chip without it (a batch) walks the S segments inside one workgroup (FUSED: a second
// accumulator set, the common epilogue, no second launch); an underfilled one (a single clip) gives them to S workgroups
// (blockIdx.z) and conv_splitk_finish_kernel adds the slabs in the same order: bit-identical, so a batched conversion
// still equals its single runs.  The k-order inside a segment (chunk-major, taps ascending, hh / hl / lh per tap) is
// conv_h3's: with equal segment bounds the two kernels agree bit for bit (tests/test_gpu_round6.py).
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <utility>

#include "conv.h"
#include "conv_device.h"
#include "h3_device.h"

namespace rvcx {

namespace {

constexpr int kWsBM = 64, kWsNB = 5, kWsBN = 2 * kWsNB * 32;     // 64 x 320
constexpr int kWsMaxHalo = 96;

// compile-time loop over the accumulator blocks: a `#pragma unroll` loop around the inlined epilogues is silently left
// rolled (-Wno-pass-failed), the accumulators are then indexed at run time and live in scratch (LABNOTES, round 5)
template <typename F, int... N>
__device__ __forceinline__ void ws_for_blocks(std::integer_sequence<int, N...>, F&& f) {
  (f(std::integral_constant<int, N>{}), ...);
}

// KS: taps (compile time: the stage loop is straight-line); MODE 0: one K segment per workgroup (blockIdx.z), raw sums to the
// split-K slabs; 1: one workgroup per tile, a single segment (S = 1) + epilogue; 2: one workgroup walks the S segments
// (a second accumulator set) + epilogue
template <int KS, int MODE>
__global__ __launch_bounds__(256, 1) void conv_ws_kernel(const ConvArgs a) {
  constexpr bool FUSED = MODE != 0, MULTI = MODE == 2;
  extern __shared__ uint4 ws_lds[];
  constexpr int NB = kWsNB, BN = kWsBN;
  constexpr int A_ST = KS * 4 * kWsBM;                 // 16-byte elements of one weight stage: [tap][op {S wh, S wl}][h][co]
  const int wrow = BN + a.wrow;                        // input-tile columns (halo = a.wrow <= kWsMaxHalo)
  const int B_ST = 4 * wrow;                           // [op {hi, S lo}][h][p]
  // (LDS is always addressed as ws_lds[offset]: a pointer table would be a generic-address-space initialiser)
  auto a_base = [&](int buf) { return buf * A_ST; };
  auto b_base = [&](int buf) { return 2 * A_ST + buf * B_ST; };

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int i = lane & 31, h = lane >> 5;
  const int bz = blockIdx.z;
  const int b = FUSED ? bz : bz / a.splitk, ks = FUSED ? 0 : bz - b * a.splitk;
  const int co0 = blockIdx.y * kWsBM, n0 = blockIdx.x * BN;
  const int len_in = a.lens_in ? a.lens_in[b] : a.Tin;
  const int len_out = a.lens_out ? a.lens_out[b] : 0x7fffffff;
  const bool dead = (a.out_mode == OUT_SHUF1D ? n0 * a.sh_s - a.sh_pad : n0) >= len_out;
  const int nchunk = a.Cin_gp / 16;
  const int in_base = n0 + a.off_min;
  const int pre_act = a.pre_act;
  const float pre_slope = a.pre_slope;
  const H3Rsrc xr = h3_rsrc(a.x + (long)b * a.x_bs, a.Cin_g * a.x_cs * 4);
  const H3Rsrc wres = h3_rsrc(a.w_h3, KS * nchunk * 4 * a.Cout_gp * 16);
  const int slab = 4 * a.Cout_gp * 16;                 // bytes of one (tap, chunk) weight slab
  const int xrow = a.x_cs * 4;

  // weights: thread t stages element (op*2 + h = t / 64, co = t % 64) of every tap
  const int a_off = co0 + (tid & 63) < a.Cout_gp ? ((tid >> 6) * a.Cout_gp + co0 + (tid & 63)) * 16 : kH3Oob;
  // input: task t = tid + 256 j -> (h, p): 8 channels of one position
  constexpr int NBT = (2 * (BN + kWsMaxHalo) + 255) / 256;
  int b_off[NBT], b_row[NBT];
#pragma unroll
  for (int j = 0; j < NBT; ++j) {
    const int t = tid + 256 * j;
    const int hh = t / wrow, p = t - hh * wrow;
    const int pos = in_base + p;
    b_row[j] = hh * 8;
    b_off[j] = (t < 2 * wrow && pos >= 0 && pos < len_in) ? pos * 4 : kH3Oob;
  }
  // tap offsets inside the staged tile
  int tp[KS];
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) tp[kk] = (kk / a.kw) * a.rowpitch + (kk % a.kw) * a.dil - a.pad - a.off_min;

  uint4 ra[KS];
  float rb[NBT][8];
  bool ovf = false;
  auto fetch = [&](int chunk) {
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) ra[kk] = h3_load4(wres, a_off != kH3Oob ? (kk * nchunk + chunk) * slab + a_off : kH3Oob);
#pragma unroll
    for (int j = 0; j < NBT; ++j) {
      const int row0 = (chunk * 16 + b_row[j]) * xrow;     // rows >= Cin_g read as 0 (beyond num_records)
#pragma unroll
      for (int q = 0; q < 8; ++q) rb[j][q] = h3_load1(xr, b_off[j] == kH3Oob ? kH3Oob : row0 + q * xrow + b_off[j]);
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) ws_lds[a_base(buf) + kk * 256 + tid] = ra[kk];
#pragma unroll
    for (int j = 0; j < NBT; ++j) {
      const int t = tid + 256 * j;
      if (t < 2 * wrow) {
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
        const int hh = t / wrow, p = t - hh * wrow;
        ws_lds[b_base(buf) + hh * wrow + p] = __builtin_bit_cast(uint4, hi);
        ws_lds[b_base(buf) + (2 + hh) * wrow + p] = __builtin_bit_cast(uint4, lo);
      }
    }
  };
  auto lds_barrier = [&]() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
  };

  f32x16 acc[NB];
  auto zero_acc = [&]() {
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  };
  zero_acc();
  auto compute = [&](int buf) {
    const uint4* Ab = ws_lds + a_base(buf);
    const uint4* Bb = ws_lds + b_base(buf);
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const half8 af0 = __builtin_bit_cast(half8, Ab[(kk * 4 + 0 + h) * kWsBM + wr * 32 + i]);
      const half8 af2 = __builtin_bit_cast(half8, Ab[(kk * 4 + 2 + h) * kWsBM + wr * 32 + i]);
      const half8 af1 = af0 * (_Float16)(1.f / kH3Scale);
      half8 bf0[NB], bf1[NB];
      const int col = wc * (NB * 32) + i + tp[kk];
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        bf0[n] = __builtin_bit_cast(half8, Bb[h * wrow + col + n * 32]);
        bf1[n] = __builtin_bit_cast(half8, Bb[(2 + h) * wrow + col + n * 32]);
      }
      // per accumulator the order is conv_h3's (hh, hl, lh); across the five accumulators the MFMAs are independent
#pragma unroll
      for (int n = 0; n < NB; ++n) acc[n] = h3_mfma(af0, bf0[n], acc[n]);   // (S wh) xh
#pragma unroll
      for (int n = 0; n < NB; ++n) acc[n] = h3_mfma(af1, bf1[n], acc[n]);   // wh (S xl)
#pragma unroll
      for (int n = 0; n < NB; ++n) acc[n] = h3_mfma(af2, bf0[n], acc[n]);   // (S wl) xh
    }
  };

  // segment bounds exactly as conv_h3's split-K: segment s = chunks [s nchunk / S, (s + 1) nchunk / S)
  const int S = MODE == 1 ? 1 : a.splitk;
  const int c_begin = FUSED ? 0 : ks * nchunk / S, c_end = FUSED ? nchunk : (ks + 1) * nchunk / S;
  f32x16 tot[MULTI ? NB : 1];      // the running sum of the finished segments, from zero like the finish kernel's
#pragma unroll
  for (int n = 0; n < (MULTI ? NB : 1); ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) tot[n][r] = 0.f;
  if (!dead) {
    fetch(c_begin);
    commit(0);
    lds_barrier();
    int seg = 0, seg_end = FUSED ? nchunk / S : c_end;      // FUSED: first chunk behind the current segment
    for (int c = c_begin; c < c_end; ++c) {
      const int buf = (c - c_begin) & 1;
      if (c + 1 < c_end) fetch(c + 1);
      compute(buf);
      if (c + 1 < c_end) commit(buf ^ 1);
      lds_barrier();
      if constexpr (MULTI) {
        if (c + 1 == seg_end) {                   // a segment ends: fold its sum (left to right), start the next from zero
#pragma unroll
          for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) tot[n][r] += acc[n][r];
          zero_acc();
          ++seg;
          seg_end = (seg + 1) * nchunk / S;
        }
      }
    }
    if constexpr (MULTI) {
#pragma unroll
      for (int n = 0; n < NB; ++n) acc[n] = tot[n];
    }
  }
  if (ovf) report_h3_overflow(a.ovf, a.ovf_layer, a.seq);
  constexpr float inv = 1.f / kH3Scale;
#pragma unroll
  for (int n = 0; n < NB; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] *= inv;

  const int co_w = co0 + wr * 32 + 4 * h, nn_w = n0 + wc * (NB * 32) + i;
  if constexpr (!FUSED) {
    float* pb = a.part + ((long)ks * a.B + b) * a.Cout_g * a.Nout;      // the slab layout of conv_h3's split-K
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      const int nn = nn_w + n * 32;
      if (nn < a.Nout) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co_w + (r & 3) + 8 * (r >> 2);
          if (co < a.Cout_g) pb[(long)co * a.Nout + nn] = acc[n][r];
        }
      }
    }
  } else {
    using Blocks = std::make_integer_sequence<int, NB>;
    if (fast_epilogue_ok(a)) {
      ws_for_blocks(Blocks{}, [&](auto nt) { store_tile_fast(a, b, co_w, nn_w + nt.value * 32, acc[nt.value], len_out); });
    } else if (a.out_mode == OUT_SHUF1D) {
      ws_for_blocks(Blocks{}, [&](auto nt) { store_tile_shuf1d(a, b, co_w, nn_w + nt.value * 32, acc[nt.value], len_out); });
    } else {
      ws_for_blocks(Blocks{}, [&](auto nt) { store_tile(a, b, 0, co_w, nn_w + nt.value * 32, acc[nt.value], len_out); });
    }
  }
}

struct WsKern {
  int ks;
  void (*split)(const ConvArgs);
  void (*single)(const ConvArgs);
  void (*multi)(const ConvArgs);
};
template <int KS>
constexpr WsKern ws_kern() {
  return {KS, conv_ws_kernel<KS, 0>, conv_ws_kernel<KS, 1>, conv_ws_kernel<KS, 2>};
}
const WsKern kWs[] = {ws_kern<3>(), ws_kern<4>(), ws_kern<5>(), ws_kern<7>(), ws_kern<9>(), ws_kern<11>()};

size_t ws_lds_bytes(int ks, int halo) { return (size_t)2 * (ks * 4 * kWsBM + 4 * (kWsBN + halo)) * 16; }

int ws_mode() {      // 0 off; 1 (default): the 2-D maps of the F0 U-Net; 2: + the 1-D layers with >= 64 k-steps (tuning)
  static const int m = getenv("RVCX_CONV_WS") ? atoi(getenv("RVCX_CONV_WS")) : 1;
  return m;
}

}  // namespace

// the number of K segments of a layer: a function of its shape (channels, taps, positions PER ITEM) only
int conv_ws_segments(const ConvArgs& a) {
  const int nchunk = a.Cin_gp / 16;
  const long base = (long)(a.Cout_gp / kWsBM) * cdiv(a.Nout, kWsBN);
  // as many segments as fill the chip with ONE workgroup per CU (a second round of workgroups doubles the launch's time),
  // each of at least two chunks (a one-chunk segment has nothing to overlap its loads with)
  int S = 1;
  while (base * S * 2 <= 256 && S < 16 && nchunk % (2 * S) == 0 && nchunk / (2 * S) >= 2) S *= 2;
  if (g_conv_override.splitk > 0 && nchunk % g_conv_override.splitk == 0) S = g_conv_override.splitk;   // tests
  return S;
}

bool conv_deep_ok(const ConvArgs& a) {
  const int mode = g_conv_override.tile == 163 ? 2 : (g_conv_override.tile >= 0 ? 0 : ws_mode());    // tile 163: forced (tests / tools)
  if (mode == 0 || !a.w_h3 || !conv_h3_enabled()) return false;
  if (a.groups != 1 || a.stride != 1 || a.Cin_gp % 16 != 0 || a.Cout_gp % kWsBM != 0 || a.Cin_g != a.Cin_gp) return false;
  if (a.x_split || a.y_split || a.nz_har || !a.x) return false;
  if (a.pre_act != ACT_NONE && a.pre_act != ACT_LRELU) return false;
  bool have = false;
  for (const auto& k : kWs) have |= k.ks == a.ksize;
  if (!have) return false;
  int lo = 1 << 30, hi = -(1 << 30);
  for (int kk = 0; kk < a.ksize; ++kk) {
    lo = std::min(lo, conv_tap_off(a, kk));
    hi = std::max(hi, conv_tap_off(a, kk));
  }
  if (hi - lo > kWsMaxHalo) return false;
  if ((long)a.Cin_gp * a.x_cs * 4 >= kH3Oob || (long)a.ksize * a.Cin_gp * a.Cout_gp * 4 >= kH3Oob) return false;
  const int ksteps = a.ksize * (a.Cin_gp / 16);
  const bool map2d = a.kw < a.ksize;
  if (mode == 1) {
    // Default: the 3 x 3 convs of the F0 U-Net with >= 256 input channels (levels 4 / 5, the intermediate layers, the first
    // decoder blocks).  Measured in round 6 (tools/bench_ws.py, tools/prof_rmvpe.py; LABNOTES 12): in a batch of 16 the
    // 512 -> 512 layers take 199 us against 262 on the 64 x 64 tile and 256 -> 256 211 against 311; a single clip is level
    // (32 us either way: two launches and the slabs' trip through HBM are the floor); below 256 channels, on the 2 x 2
    // polyphase ConvTranspose2d and on every 1-D layer of the synthesizer the 64 x 64 tile at three workgroups per CU is
    // as fast or faster (the tile form is not what bounds the split-fp16 kernels) -- those stay where they were.
    if (!map2d || a.ksize != 9 || a.Cin_gp < 256 || a.Nout > 32768) return false;
  } else {
    if (ksteps < 32) return false;
  }
  // the segment slabs must fit the caller's scratch, per item and for the batch
  const int S = conv_ws_segments(a);
  const long cap_item = a.part_cap_item > 0 ? a.part_cap_item : a.part_cap;
  if (S > 1 && a.part && (long)S * a.Cout_g * a.Nout > cap_item) return false;      // (no scratch at all: the fused form)
  return true;
}

void launch_conv_deep(const ConvArgs& a_in, hipStream_t stream) {
  ConvArgs a = a_in;
  static std::mutex mu;
  static uint64_t done = 0;
  int dev = 0;
  RVCX_HIP(hipGetDevice(&dev));
  {
    std::lock_guard<std::mutex> g(mu);
    if (!((done >> (dev & 63)) & 1)) {
      for (const auto& k : kWs)
        for (auto fn : {k.split, k.single, k.multi})
          RVCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)ws_lds_bytes(k.ks, kWsMaxHalo)));
      done |= 1ull << (dev & 63);
    }
  }
  int lo = 1 << 30, hi = -(1 << 30);
  for (int kk = 0; kk < a.ksize; ++kk) {
    lo = std::min(lo, conv_tap_off(a, kk));
    hi = std::max(hi, conv_tap_off(a, kk));
  }
  a.off_min = lo;
  a.wrow = hi - lo;
  const WsKern* K = nullptr;
  for (const auto& k : kWs)
    if (k.ks == a.ksize) K = &k;
  RVCX_CHECK(K != nullptr, "conv_ws: unsupported tap count");
  const int S = conv_ws_segments(a);
  a.splitk = S;
  const long base = (long)(a.Cout_gp / kWsBM) * cdiv(a.Nout, kWsBN);
  // one workgroup per tile walks all segments when that fills the chip, or when the batch's slabs would not fit the scratch
  static const int fuse_at = getenv("RVCX_CONV_WS_FUSE_AT") ? atoi(getenv("RVCX_CONV_WS_FUSE_AT")) : 160;
  const bool fused = S == 1 || base * a.B >= fuse_at || !a.part || (long)S * a.B * a.Cout_g * a.Nout > a.part_cap;
  const size_t lds = ws_lds_bytes(a.ksize, a.wrow);
  dim3 grid(cdiv(a.Nout, kWsBN), a.Cout_gp / kWsBM, fused ? a.B : a.B * S);
  hipLaunchKernelGGL(!fused ? K->split : (S == 1 ? K->single : K->multi), grid, dim3(256), lds, stream, a);
  if (!fused) launch_splitk_finish(a, stream);
  RVCX_HIP(hipGetLastError());
}

}  // namespace rvcx
