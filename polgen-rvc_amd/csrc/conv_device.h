// Device-side pieces shared by the conv kernels: activation and the fused epilogue.
#pragma once
#include "conv.h"

namespace rvcx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// an activation did not fit the fp16 hi/lo split (conv.h: kH3ActLimit): tell the context and the layer
__device__ __forceinline__ void report_h3_overflow(int* ovf, int* layer, int seq) {
  if (ovf) atomicOr(ovf, kErrH3Overflow);
  if (layer) atomicMax(layer, 0x7fffffff - seq);     // larger = earlier launch of this call
}

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  switch (act) {
    case ACT_LRELU: return v > 0.f ? v : v * slope;
    case ACT_RELU: return v > 0.f ? v : 0.f;
    case ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    case ACT_TANH: return tanhf(v);
    case ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
    default: return v;
  }
}


// The stage's NSF noise conv evaluated in place of a residual read (ConvArgs::nz_*): Conv1d(1, C, k, stride, padding) of the
// harmonic source at output (c, t), the conv_cin1 kernel's arithmetic (FMA chain over ascending taps, then the bias).
__device__ __forceinline__ float noise_conv_at(const ConvArgs& a, int b, int c, int t) {
  const int nz_len = a.nz_lens ? a.nz_lens[b] : a.nz_len;
  const float* hb = a.nz_har + (long)b * a.nz_bs;
  const long p0 = (long)t * a.nz_stride - a.nz_pad;
  float n = 0.f;
  for (int j = 0; j < a.nz_k; ++j) {
    const long p = p0 + j;
    const float hvv = (p >= 0 && p < nz_len) ? hb[p] : 0.f;
    n = fmaf(a.nz_w[(long)j * a.nz_wstride + c], hvv, n);
  }
  return n + (a.nz_b ? a.nz_b[c] : 0.f);
}

// one output element: bias -> activation -> residual -> length mask -> store (by output mode)
__device__ __forceinline__ void store_elem(const ConvArgs& a, int b, int cg, int nn, float v, int len_out) {
  if (a.bias) v += a.bias[cg];
  v = apply_act(v, a.act, a.act_slope);
  if (a.out_mode == OUT_NORMAL) {
    if (a.res) v += a.res[(long)b * a.res_bs + (long)cg * a.res_cs + nn];
    if (nn >= len_out) v = 0.f;
    if (a.zero_wp > 0) {
      const int col = nn % a.zero_wp;
      if (col == 0 || col == a.zero_wp - 1) v = 0.f;
    }
    if (a.y) a.y[(long)b * a.y_bs + (long)cg * a.y_cs + nn] = v;
    if (a.acc2_mode != ACC2_NONE) {
      float* p2 = a.y2 + (long)b * a.y2_bs + (long)cg * a.y2_cs + nn;
      if (a.acc2_mode == ACC2_SET) *p2 = v;
      else if (a.acc2_mode == ACC2_ADD) *p2 = *p2 + v;
      else *p2 = (*p2 + v) / a.acc2_div;
    }
  } else if (a.out_mode == OUT_SHUF1D) {
    const int c = cg / a.sh_s, ph = cg - c * a.sh_s;   // packed channel = c * stride + phase
    const int t = nn * a.sh_s + ph - a.sh_pad;
    if (t >= 0 && t < a.sh_tout) {
      if (a.res) v += a.res[(long)b * a.res_bs + (long)c * a.res_cs + t];
      if (a.nz_har) v += noise_conv_at(a, b, c, t);
      if (t >= len_out) v = 0.f;
      a.y[(long)b * a.y_bs + (long)c * a.y_cs + t] = v;
    }
  } else if (a.out_mode == OUT_SHUF2D) {
    const int ph = cg / a.sh_cout, c = cg - ph * a.sh_cout;
    const int pa = ph >> 1, pb = ph & 1;
    const int irow = nn / a.wp_in, jj = nn - irow * a.wp_in;
    if (nn >= len_out) v = 0.f;          // rows below a shorter item's last one stay zero
    if (jj != 0 && jj != a.wp_in - 1) {
      const long rowbase = (long)b * a.y_bs + (long)c * a.y_cs + (long)(2 * irow + pa) * a.wp_out;
      a.y[rowbase + 2 * (jj - 1) + pb + 1] = v;
      if (jj == 1 && pb == 0) a.y[rowbase] = 0.f;
      if (jj == a.wp_in - 2 && pb == 1) a.y[rowbase + a.wp_out - 1] = 0.f;
    }
  } else {  // OUT_TRANSPOSED: y[b][n][c]
    if (a.res) v += a.res[(long)b * a.res_bs + (long)nn * a.res_cs + cg];
    if (nn >= len_out) v = 0.f;
    a.y[(long)b * a.y_bs + (long)nn * a.y_cs + cg] = v;
  }
}

// one 32x32 accumulator tile.  C layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5);
// co_base already includes the 4*(lane>>5) term.
__device__ __forceinline__ void store_tile(const ConvArgs& a, int b, int g, int co_base, int nn, const f32x16& t,
                                           int len_out) {
  if (nn >= a.Nout) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = co_base + (r & 3) + 8 * (r >> 2);
    if (co < a.Cout_g) store_elem(a, b, g * a.Cout_g + co, nn, t[r], len_out);
  }
}

// Lean epilogue for the common case: OUT_NORMAL (pad-column zeroing of row-padded maps included), activation in {none, lrelu,
// relu} (all three are max(v, slope*v)), optional bias / residual / secondary accumulator.  Everything
// that is uniform is decided once per tile, so an element costs ~8 instructions instead of ~50.
__device__ __forceinline__ bool fast_epilogue_ok(const ConvArgs& a) {
  return a.out_mode == OUT_NORMAL && a.act <= ACT_RELU;
}

__device__ __forceinline__ void store_tile_fast(const ConvArgs& a, int b, int co_base, int nn, const f32x16& t,
                                                int len_out) {
  if (nn >= a.Nout) return;
  const float slope = a.act == ACT_NONE ? 1.f : (a.act == ACT_RELU ? 0.f : a.act_slope);
  bool live = nn < len_out;
  if (a.zero_wp > 0) {   // a lane owns one output column: one modulo per tile, not per element
    const int col = nn % a.zero_wp;
    live = live && col != 0 && col != a.zero_wp - 1;
  }
  float* yb = a.y ? a.y + (long)b * a.y_bs + nn : nullptr;
  const float* rb = a.res ? a.res + (long)b * a.res_bs + nn : nullptr;
  float* y2b = a.acc2_mode != ACC2_NONE ? a.y2 + (long)b * a.y2_bs + nn : nullptr;
  // The loads of 8 elements are issued before their first store: y / y2 may alias res (in-place residual
  // updates), so the compiler would otherwise keep every load behind the previous store and the epilogue
  // becomes 16 serial memory round trips per tile (measured: 68 us per workgroup instead of ~15).
  // Batches of 8 keep the register footprint below the main loop's (occupancy stays at 3 waves/SIMD).
#pragma unroll
  for (int r0 = 0; r0 < 16; r0 += 8) {
    float bv[8], rv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int r = r0 + q;
      const int co = co_base + (r & 3) + 8 * (r >> 2);
      const int cc = co < a.Cout_g ? co : 0;
      bv[q] = a.bias ? a.bias[cc] : 0.f;
      rv[q] = rb ? rb[cc * a.res_cs] : 0.f;
    }
    if (y2b) {   // last conv of a ResBlock: running mean over the blocks (read-modify-write of y2)
      float pv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int r = r0 + q;
        const int co = co_base + (r & 3) + 8 * (r >> 2);
        const int cc = co < a.Cout_g ? co : 0;
        pv[q] = a.acc2_mode != ACC2_SET ? y2b[cc * a.y2_cs] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int r = r0 + q;
        const int co = co_base + (r & 3) + 8 * (r >> 2);
        if (co < a.Cout_g) {
          float v = t[r] + bv[q];
          v = fmaxf(v, v * slope) + rv[q];
          v = live ? v : 0.f;
          if (yb) yb[co * a.y_cs] = v;
          float* p2 = y2b + co * a.y2_cs;
          if (a.acc2_mode == ACC2_SET) *p2 = v;
          else if (a.acc2_mode == ACC2_ADD) *p2 = pv[q] + v;
          else *p2 = (pv[q] + v) / a.acc2_div;
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int r = r0 + q;
        const int co = co_base + (r & 3) + 8 * (r >> 2);
        if (co < a.Cout_g) {
          float v = t[r] + bv[q];
          v = fmaxf(v, v * slope) + rv[q];
          yb[co * a.y_cs] = live ? v : 0.f;
        }
      }
    }
  }
}

// Polyphase ConvTranspose1d shuffle store (OUT_SHUF1D) without a division per element: packed channel cg = c*s + ph,
// so the 4 consecutive accumulator rows of a register group are 4 consecutive (c, ph) pairs -- one division per
// group, then increments.  Residual loads of a group are issued before its stores.
__device__ __forceinline__ void store_tile_shuf1d(const ConvArgs& a, int b, int co_base, int nn, const f32x16& t,
                                                  int len_out) {
  if (nn >= a.Nout) return;
  const int s = a.sh_s;
  float* yb = a.y + (long)b * a.y_bs;
  const float* rb = a.res ? a.res + (long)b * a.res_bs : nullptr;
  const int tbase = nn * s - a.sh_pad;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int cg0 = co_base + 8 * g;
    int c = cg0 / s, ph = cg0 - c * s;
    int cc[4], tt[4];
    bool ok[4];
    float bv[4], rv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      cc[q] = c;
      tt[q] = tbase + ph;
      ok[q] = cg0 + q < a.Cout_g && tt[q] >= 0 && tt[q] < a.sh_tout;
      bv[q] = (a.bias && cg0 + q < a.Cout_g) ? a.bias[cg0 + q] : 0.f;
      rv[q] = (rb && ok[q]) ? rb[(long)c * a.res_cs + tt[q]] : 0.f;
      if (a.nz_har && ok[q]) rv[q] += noise_conv_at(a, b, c, tt[q]);
      if (++ph == s) {
        ph = 0;
        ++c;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (ok[q]) {
        float v = apply_act(t[4 * g + q] + bv[q], a.act, a.act_slope) + rv[q];
        if (tt[q] >= len_out) v = 0.f;
        yb[(long)cc[q] * a.y_cs + tt[q]] = v;
      }
    }
  }
}

}  // namespace rvcx
