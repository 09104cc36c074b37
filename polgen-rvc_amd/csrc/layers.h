// Thin helpers that turn packed layers into ConvArgs for the common tensor layouts:
//   1-D maps  (B, C, T)            time contiguous
//   2-D maps  (B, C, H, Wp=W+2)    row-padded: columns 0 and Wp-1 are kept at zero so a 3x3
//                                  conv becomes a flat 1-D conv with taps {-Wp-1 .. Wp+1}
#pragma once
#include "ctx.h"

namespace rvcx {

// dense (B,C,T) -> (B,Cout,Tout) conv1d
inline ConvArgs conv1d_args(const ConvW& w, const float* x, float* y, int B, int Tin, int Tout,
                            int stride = 1, int dil = 1, int pad = 0) {
  ConvArgs a;
  conv_set_weights(a, w);
  a.x = x;
  a.y = y;
  a.B = B;
  a.Tin = Tin;
  a.Nout = Tout;
  a.stride = stride;
  a.dil = dil;
  a.pad = pad;
  a.x_bs = (long)w.cin * Tin;
  a.x_cs = Tin;
  a.y_bs = (long)w.cout * Tout;
  a.y_cs = Tout;
  return a;
}

inline void conv_set_res(ConvArgs& a, const float* res, int C, int T) {
  a.res = res;
  a.res_bs = (long)C * T;
  a.res_cs = T;
}

// polyphase ConvTranspose1d layer
struct ConvT1dW {
  ConvW w;        // cout = stride*cout_real, k = taps
  int stride = 1, pad_t = 0, cout_real = 0, k_orig = 0, taps_pad = 0;
};

inline ConvArgs convT1d_args(const ConvT1dW& L, const float* x, float* y, int B, int Tin, int Tout) {
  ConvArgs a;
  conv_set_weights(a, L.w);
  a.x = x;
  a.y = y;
  a.B = B;
  a.Tin = Tin;
  a.Nout = cdiv(Tout + L.pad_t, L.stride);
  a.stride = 1;
  a.dil = 1;
  a.pad = L.taps_pad;
  a.x_bs = (long)L.w.cin * Tin;
  a.x_cs = Tin;
  a.y_bs = (long)L.cout_real * Tout;
  a.y_cs = Tout;
  a.out_mode = OUT_SHUF1D;
  a.sh_s = L.stride;
  a.sh_pad = L.pad_t;
  a.sh_cout = L.cout_real;
  a.sh_tout = Tout;
  return a;
}

// 3x3 conv on a row-padded 2-D map: x (B,Cin,H,Wp) -> y (B,Cout,H,Wp)
inline ConvArgs conv2d_args(const ConvW& w, const float* x, float* y, int B, int H, int Wp) {
  ConvArgs a;
  conv_set_weights(a, w);
  a.x = x;
  a.y = y;
  a.B = B;
  a.Tin = H * Wp;
  a.Nout = H * Wp;
  a.stride = 1;
  a.dil = 1;
  if (w.k == 9) {
    a.kw = 3;
    a.rowpitch = Wp;
    a.pad = Wp + 1;
  } else {  // 1x1
    a.kw = 1;
    a.rowpitch = 0;
    a.pad = 0;
  }
  a.x_bs = (long)w.cin * H * Wp;
  a.x_cs = H * Wp;
  a.y_bs = (long)w.cout * H * Wp;
  a.y_cs = H * Wp;
  a.zero_wp = Wp;
  return a;
}

// polyphase ConvTranspose2d 3x3 stride 2 pad 1 outpad 1: x (B,Cin,H,Wp) -> y (B,Cout,2H,2W+2)
struct ConvT2dW {
  ConvW w;  // cout = 4*cout_real, k = 4 (taps (dy,dx) in {0,1}^2)
  int cout_real = 0;
};

inline ConvArgs convT2d_args(const ConvT2dW& L, const float* x, float* y, int B, int H, int Wp) {
  ConvArgs a;
  conv_set_weights(a, L.w);
  const int W = Wp - 2, Wpo = 2 * W + 2;
  a.x = x;
  a.y = y;
  a.B = B;
  a.Tin = H * Wp;
  a.Nout = H * Wp;
  a.kw = 2;
  a.rowpitch = Wp;
  a.dil = 1;
  a.pad = 0;
  a.stride = 1;
  a.x_bs = (long)L.w.cin * H * Wp;
  a.x_cs = H * Wp;
  a.y_bs = (long)L.cout_real * 2 * H * Wpo;
  a.y_cs = 2 * H * Wpo;
  a.out_mode = OUT_SHUF2D;
  a.sh_cout = L.cout_real;
  a.wp_in = Wp;
  a.wp_out = Wpo;
  return a;
}

ConvT1dW make_convT1d(Ctx& c, const float* w, const float* bias, int cin, int cout, int k, int s, int p);
// w (Cin,Cout,3,3); scale/shift (per Cout) fold an eval-mode BatchNorm that follows (may be null)
ConvT2dW make_convT2d(Ctx& c, const float* w, const float* scale, const float* shift, int cin, int cout);

}  // namespace rvcx
