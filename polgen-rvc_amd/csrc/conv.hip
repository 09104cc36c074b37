// Implicit-GEMM conv on v_mfma_f32_32x32x2_f32 -- see conv.h for the design.
#include "conv.h"
#include "conv_device.h"
#include "gemm.h"

#include <algorithm>
#include <cmath>

namespace rvcx {

// BM x BN output tile per 256-thread block; waves arranged WR x WC; each wave owns
// WM x WN MFMA tiles of 32x32.
template <int BM, int BN, int WR, int WC>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvArgs a) {
  constexpr int WM = BM / (32 * WR), WN = BN / (32 * WC);
  static_assert(WR * WC == 4 && WM >= 1 && WN >= 1, "bad tile");
  extern __shared__ float smem[];
  float* As = smem;                                   // [kk_chunk][ci_chunk][BM]
  float* Bs = smem + a.kk_chunk * a.ci_chunk * BM;    // [ci_chunk][wrow]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WC, wc = wave % WC;
  const int i = lane & 31, h = lane >> 5;
  const int b = blockIdx.z;
  const int mt_per_g = (a.Cout_gp + BM - 1) / BM;
  const int g = blockIdx.y / mt_per_g, mt = blockIdx.y % mt_per_g;
  const int co0 = mt * BM;
  const int n0 = blockIdx.x * BN;
  const int len_in = a.lens_in ? a.lens_in[b] : a.Tin;
  const float* xg = a.x + (long)b * a.x_bs + (long)(g * a.Cin_g) * a.x_cs;
  const float* wg = a.w + (long)g * a.ksize * a.Cin_gp * a.Cout_gp;
  const int ci_chunk = a.ci_chunk, kk_chunk = a.kk_chunk, wrow = a.wrow;

  f32x16 acc[WM][WN];
#pragma unroll
  for (int m = 0; m < WM; ++m)
#pragma unroll
    for (int n = 0; n < WN; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  const int in_base = n0 * a.stride + a.off_min;
  const int pre_act = a.pre_act;
  const float pre_slope = a.pre_slope;

  for (int ci0 = 0; ci0 < a.Cin_gp; ci0 += ci_chunk) {
    __syncthreads();
    // ---- stage the input tile: ci_chunk rows x wrow positions (coalesced along time)
    for (int r = wave; r < ci_chunk; r += 4) {
      const int ci = ci0 + r;
      const bool cvalid = ci < a.Cin_g;
      const float* xr = xg + (long)ci * a.x_cs;
      float* br = Bs + r * wrow;
      for (int p = lane; p < wrow; p += 64) {
        const int pos = in_base + p;
        float v = 0.f;
        if (cvalid && pos >= 0 && pos < len_in) {
          v = xr[pos];
          if (pre_act == ACT_LRELU) v = v > 0.f ? v : v * pre_slope;
        }
        br[p] = v;
      }
    }
    for (int kk0 = 0; kk0 < a.ksize; kk0 += kk_chunk) {
      if (kk0 > 0) __syncthreads();
      // ---- stage the weight tile: rows (kk, ci) of BM contiguous output channels
      const int nrows = kk_chunk * ci_chunk;
      for (int idx = tid; idx < nrows * (BM / 4); idx += 256) {
        const int row = idx / (BM / 4), c4 = idx % (BM / 4);
        const int kkl = row / ci_chunk, cil = row - kkl * ci_chunk;
        const int kk = kk0 + kkl, co = co0 + c4 * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kk < a.ksize && co < a.Cout_gp)
          v = *reinterpret_cast<const float4*>(wg + ((long)kk * a.Cin_gp + ci0 + cil) * a.Cout_gp + co);
        *reinterpret_cast<float4*>(As + row * BM + c4 * 4) = v;
      }
      __syncthreads();
      const int kkn = min(kk_chunk, a.ksize - kk0);
      for (int kkl = 0; kkl < kkn; ++kkl) {
        const int kk = kk0 + kkl;
        const int off = (kk / a.kw) * a.rowpitch + (kk % a.kw) * a.dil - a.pad - a.off_min;
        const float* Ap = As + (kkl * ci_chunk + h) * BM + wr * (WM * 32) + i;
        const float* Bp = Bs + h * wrow + (wc * (WN * 32) + i) * a.stride + off;
        for (int cp = 0; cp < ci_chunk / 2; ++cp) {
          float av[WM], bv[WN];
#pragma unroll
          for (int m = 0; m < WM; ++m) av[m] = Ap[cp * 2 * BM + m * 32];
#pragma unroll
          for (int n = 0; n < WN; ++n) bv[n] = Bp[cp * 2 * wrow + n * 32 * a.stride];
#pragma unroll
          for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int n = 0; n < WN; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[n], acc[m][n], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue (explicit per-tile calls: the accumulators must stay statically indexed)
  const int len_out = a.lens_out ? a.lens_out[b] : 0x7fffffff;
  const int co_w = co0 + wr * (WM * 32) + 4 * h, nn_w = n0 + wc * (WN * 32) + i;
  store_tile(a, b, g, co_w, nn_w, acc[0][0], len_out);
  if constexpr (WN > 1) store_tile(a, b, g, co_w, nn_w + 32, acc[0][1], len_out);
  if constexpr (WM > 1) {
    store_tile(a, b, g, co_w + 32, nn_w, acc[1][0], len_out);
    if constexpr (WN > 1) store_tile(a, b, g, co_w + 32, nn_w + 32, acc[1][1], len_out);
  }
}

// ---------------------------------------------------------------------------------------
struct TileCfg {
  int bm, bn;
  float eff;
  void (*kern)(const ConvArgs);
};

static const TileCfg kTiles[] = {
    {128, 128, 1.00f, conv_mfma_kernel<128, 128, 2, 2>},
    {64, 256, 1.00f, conv_mfma_kernel<64, 256, 1, 4>},
    {32, 256, 0.92f, conv_mfma_kernel<32, 256, 1, 4>},
    {128, 64, 0.85f, conv_mfma_kernel<128, 64, 4, 1>},
    {64, 128, 0.85f, conv_mfma_kernel<64, 128, 2, 2>},
    {64, 64, 0.75f, conv_mfma_kernel<64, 64, 2, 2>},
    {32, 128, 0.75f, conv_mfma_kernel<32, 128, 1, 4>},
};
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);
constexpr int kMaxLdsFloats = 16384;  // 64 KiB per block -> 2 blocks/CU of the 160 KiB

void conv_init() {
  conv_fast_init();
  for (int t = 0; t < kNumTiles; ++t)
    RVCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kTiles[t].kern),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
}

double conv_flops(const ConvArgs& a) {
  return 2.0 * a.B * a.groups * (double)a.Cout_g * a.Nout * (double)a.ksize * a.Cin_g;
}

static bool plan_lds(ConvArgs& a, int bm, int bn) {
  int off_min = 1 << 30, off_max = -(1 << 30);
  for (int kk = 0; kk < a.ksize; ++kk) {
    int o = conv_tap_off(a, kk);
    off_min = std::min(off_min, o);
    off_max = std::max(off_max, o);
  }
  a.off_min = off_min;
  a.wrow = (bn - 1) * a.stride + (off_max - off_min) + 1;
  int ci = 0;
  for (int c = std::min(16, a.Cin_gp) & ~1; c >= 2; c -= 2) {
    if (a.Cin_gp % c != 0) continue;
    long lim = (c == 2) ? kMaxLdsFloats * 3 / 4 : kMaxLdsFloats / 2;
    if ((long)c * a.wrow <= lim) {
      ci = c;
      break;
    }
  }
  if (ci == 0) return false;
  int a_budget = kMaxLdsFloats - ci * a.wrow;
  int kkc = std::min(a.ksize, a_budget / (ci * bm));
  if (kkc < 1) return false;
  kkc = std::min(kkc, std::max(1, 8192 / (ci * bm)));
  a.ci_chunk = ci;
  a.kk_chunk = kkc;
  return true;
}

namespace {
struct ProfRec {
  hipEvent_t a, b;
  int tile;
  double flops;
  int cin, cout, k, nout, stride, B;
};
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
}  // namespace

void conv_profile_begin() {
  g_prof.clear();
  g_prof_on = true;
}

static std::string g_prof_csv;
const char* conv_profile_csv() { return g_prof_csv.c_str(); }

void conv_profile_end(ConvProfile* out) {
  g_prof_on = false;
  g_prof_csv = "tile,B,cin,cout,k,stride,nout,gflop,ms,tflops\n";
  for (int t = 0; t < kNumTiles; ++t) {
    out->bm[t] = kTiles[t].bm;
    out->bn[t] = kTiles[t].bn;
    out->halo[t] = -1;
  }
  conv_fast_describe(out);
  conv_h3_describe(out);
  resblock_pair_describe(out);
  out->bm[kBlock3Slot] = 64;
  out->bn[kBlock3Slot] = resblock3_n1(64);
  out->halo[kBlock3Slot] = 500003;
  gemm_describe(out);
  for (auto& r : g_prof) {
    RVCX_HIP(hipEventSynchronize(r.b));
    float ms = 0.f;
    RVCX_HIP(hipEventElapsedTime(&ms, r.a, r.b));
    out->launches[r.tile] += 1;
    out->flops[r.tile] += r.flops;
    out->ms[r.tile] += ms;
    char line[256];
    snprintf(line, sizeof line, "%d,%d,%d,%d,%d,%d,%d,%.3f,%.4f,%.1f\n", r.tile, r.B, r.cin, r.cout, r.k, r.stride,
             r.nout, r.flops / 1e9, ms, r.flops / (ms * 1e-3) / 1e12);
    g_prof_csv += line;
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  g_prof.clear();
}

void conv_launch_pair(const PairArgs& a, double flops, hipStream_t stream) {
  if (!g_prof_on) {
    launch_resblock_pair(a, stream);
    return;
  }
  ProfRec rec;
  RVCX_HIP(hipEventCreate(&rec.a));
  RVCX_HIP(hipEventCreate(&rec.b));
  rec.tile = resblock_pair_slot(a.C);
  rec.flops = flops;
  rec.cin = a.C; rec.cout = a.C; rec.k = a.k; rec.nout = a.T; rec.stride = a.dil; rec.B = a.B;
  RVCX_HIP(hipEventRecord(rec.a, stream));
  launch_resblock_pair(a, stream);
  RVCX_HIP(hipEventRecord(rec.b, stream));
  g_prof.push_back(rec);
}

void conv_launch_block3(const Block3Args& a, double flops, hipStream_t stream) {
  if (!g_prof_on) {
    launch_resblock3(a, stream);
    return;
  }
  ProfRec rec;
  RVCX_HIP(hipEventCreate(&rec.a));
  RVCX_HIP(hipEventCreate(&rec.b));
  rec.tile = kBlock3Slot;
  rec.flops = flops;
  rec.cin = a.C; rec.cout = a.C; rec.k = 3; rec.nout = a.T; rec.stride = 135; rec.B = a.B;
  RVCX_HIP(hipEventRecord(rec.a, stream));
  launch_resblock3(a, stream);
  RVCX_HIP(hipEventRecord(rec.b, stream));
  g_prof.push_back(rec);
}

void conv_launch_gemm(const GemmArgs& a, double flops, hipStream_t stream) {
  if (!g_prof_on) {
    launch_gemm(a, stream);
    return;
  }
  ProfRec rec;
  RVCX_HIP(hipEventCreate(&rec.a));
  RVCX_HIP(hipEventCreate(&rec.b));
  rec.flops = flops;
  rec.cin = a.cin; rec.cout = a.cout; rec.k = 1; rec.nout = (int)a.rows; rec.stride = 1; rec.B = (int)(a.rows / a.T);
  RVCX_HIP(hipEventRecord(rec.a, stream));
  rec.tile = launch_gemm(a, stream);
  RVCX_HIP(hipEventRecord(rec.b, stream));
  g_prof.push_back(rec);
}

void launch_conv(ConvArgs a, hipStream_t stream) {
  RVCX_CHECK(a.Cin_gp % 2 == 0 && a.Cout_gp % 32 == 0, "conv: unpadded weights");
  RVCX_CHECK(a.Nout > 0 && a.B > 0, "conv: empty problem");
  if (g_prof_on) {   // the stride-1 fast family records under tile slot 7
    ProfRec rec;
    RVCX_HIP(hipEventCreate(&rec.a));
    RVCX_HIP(hipEventCreate(&rec.b));
    rec.tile = 0;
    rec.flops = conv_flops(a);
    rec.cin = a.Cin_g * a.groups; rec.cout = a.Cout_g * a.groups; rec.k = a.ksize; rec.nout = a.Nout;
    rec.stride = a.stride; rec.B = a.B;
    RVCX_HIP(hipEventRecord(rec.a, stream));
    const int slot = launch_conv_fast(a, stream);
    if (slot >= 0) {
      rec.tile = slot;
      RVCX_HIP(hipEventRecord(rec.b, stream));
      g_prof.push_back(rec);
      return;
    }
    (void)hipEventDestroy(rec.a);
    (void)hipEventDestroy(rec.b);
  } else if (launch_conv_fast(a, stream) >= 0) {
    return;
  }
  int best = -1;
  double best_t = 1e300;
  ConvArgs best_a = a;
  for (int t = 0; t < kNumTiles; ++t) {
    ConvArgs c = a;
    if (!plan_lds(c, kTiles[t].bm, kTiles[t].bn)) continue;
    long mt = cdiv(a.Cout_gp, kTiles[t].bm), nt = cdiv(a.Nout, kTiles[t].bn);
    long blocks = mt * nt * a.groups;   // per batch item: the choice must not depend on the batch size
    long rounds = (blocks + 511) / 512;  // 256 CUs x 2 resident blocks
    double tm = (double)rounds * kTiles[t].bm * kTiles[t].bn / kTiles[t].eff;
    if (tm < best_t) {
      best_t = tm;
      best = t;
      best_a = c;
    }
  }
  RVCX_CHECK(best >= 0, "conv: no tile configuration fits LDS");
  const TileCfg& T = kTiles[best];
  dim3 grid(cdiv(a.Nout, T.bn), cdiv(a.Cout_gp, T.bm) * a.groups, a.B);
  size_t lds = (size_t)(best_a.kk_chunk * best_a.ci_chunk * T.bm + best_a.ci_chunk * best_a.wrow) * sizeof(float);
  ProfRec rec;
  if (g_prof_on) {
    RVCX_HIP(hipEventCreate(&rec.a));
    RVCX_HIP(hipEventCreate(&rec.b));
    rec.tile = best;
    rec.flops = conv_flops(a);
    rec.cin = a.Cin_g * a.groups; rec.cout = a.Cout_g * a.groups; rec.k = a.ksize; rec.nout = a.Nout;
    rec.stride = a.stride; rec.B = a.B;
    RVCX_HIP(hipEventRecord(rec.a, stream));
  }
  hipLaunchKernelGGL(T.kern, grid, dim3(256), lds, stream, best_a);
  if (g_prof_on) {
    RVCX_HIP(hipEventRecord(rec.b, stream));
    g_prof.push_back(rec);
  }
  RVCX_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------
int conv_cin_pad(int cin_g) { return cin_g <= 16 ? round_up(cin_g, 2) : round_up(cin_g, 16); }
int conv_cout_pad(int cout_g) { return round_up(cout_g, 32); }

std::vector<float> pack_conv_weight(const float* w, int cout, int cin_g, int k, int groups) {
  const int cout_g = cout / groups;
  const int cip = conv_cin_pad(cin_g), cop = conv_cout_pad(cout_g);
  std::vector<float> out((size_t)groups * k * cip * cop, 0.f);
  for (int g = 0; g < groups; ++g)
    for (int co = 0; co < cout_g; ++co)
      for (int ci = 0; ci < cin_g; ++ci)
        for (int kk = 0; kk < k; ++kk)
          out[(((size_t)g * k + kk) * cip + ci) * cop + co] =
              w[((size_t)(g * cout_g + co) * cin_g + ci) * k + kk];
  return out;
}

PolyPhase1d pack_convtranspose1d(const float* w, const float* bias, int cin, int cout, int k, int s,
                                 int p) {
  // y[co][t] = sum_ci sum_kk W[ci][co][kk] x[ci][(t+p-kk)/s].  With u=t+p, q=u/s, ph=u%s,
  // kk = ph + m*s, the input index is q-m: a conv over q with taps m' = M-1-m at offset
  // m' - (M-1), producing channel co*s+ph (phase fastest: the 4 consecutive accumulator rows a lane owns are 4
  // consecutive output samples, so the shuffle store writes 16 contiguous bytes per lane), stored at t = q*s + ph - p.
  PolyPhase1d r;
  const int M = cdiv(k, s);
  r.taps = M;
  r.pad = M - 1;
  r.cout_total = s * cout;
  std::vector<float> wc((size_t)s * cout * cin * M, 0.f);  // (Cout', Cin, M)
  for (int ph = 0; ph < s; ++ph)
    for (int co = 0; co < cout; ++co)
      for (int ci = 0; ci < cin; ++ci)
        for (int mp = 0; mp < M; ++mp) {
          int kk = ph + (M - 1 - mp) * s;
          if (kk < k)
            wc[(((size_t)(co * s + ph)) * cin + ci) * M + mp] = w[((size_t)ci * cout + co) * k + kk];
        }
  r.w = pack_conv_weight(wc.data(), s * cout, cin, M, 1);
  r.bias.resize((size_t)s * cout, 0.f);
  if (bias)
    for (int ph = 0; ph < s; ++ph)
      for (int co = 0; co < cout; ++co) r.bias[(size_t)co * s + ph] = bias[co];
  return r;
}

}  // namespace rvcx
