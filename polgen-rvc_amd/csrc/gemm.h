// Time-major GEMM for the Linear layers of the transformer sections (gemm.hip).
#pragma once
#include "conv.h"

namespace rvcx {

struct GemmArgs {
  // input, one of: xs = rows in fp16 hi/lo split form (row r at xs + r * ld_xs bytes, 64 bytes per 16-channel chunk:
  // {hi h0, hi h1, S lo h0, S lo h1} x 8 halves), x = fp32 rows (ld_x floats apart; the exact-fp32 kernel only)
  const void* xs = nullptr;
  long ld_xs = 0;
  const float* x = nullptr;
  long ld_x = 0;
  const void* w_h3 = nullptr;    // ConvW::w_h3 of a k = 1 layer: H3[chunk][op][h][cout_p][8 halves]
  const float* w = nullptr;      // ConvW::w: Wp[cin_p][cout_p] fp32 (exact-fp32 kernel)
  const float* bias = nullptr;
  const float* res = nullptr;    // fp32 time-major residual, added after the activation
  long ld_res = 0;
  float* y = nullptr;            // fp32 time-major output (rows x cout, ld_y floats apart) or null
  long ld_y = 0;
  void* ys = nullptr;            // split time-major output (ld_ys bytes apart) or null
  long ld_ys = 0;
  float* y_cf = nullptr;         // fp32 channel-first output (B, cout, T) or null (q / k / v for the attention kernels)
  long cf_bs = 0;                // its batch stride in floats
  const int* lens = nullptr;     // per item valid rows (null: T); rows beyond are stored as zeros
  long rows = 0;                 // B * T
  int T = 1;                     // rows per batch item
  int cin = 0, cin_p = 0, cout = 0, cout_p = 0;
  int act = ACT_NONE;            // ACT_NONE / ACT_RELU / ACT_GELU
  float* part = nullptr;         // K-split scratch offered by the caller (launch_gemm decides): kGemmSeg x rows x cout floats
  long part_cap = 0;             // floats
  int* ovf = nullptr;            // device error word (conv.h)
  int* ovf_next = nullptr;       // overflow word of the layer that CONSUMES ys: this kernel writes its split input
  int seq = 0;
};

bool gemm_h3_enabled();          // RVCX_GEMM (default on) and the split-fp16 kernels enabled
int launch_gemm(GemmArgs a, hipStream_t stream);                                // raw launch; profile slot or -1
void conv_launch_gemm(const GemmArgs& a, double flops, hipStream_t stream);     // + profile record (conv.hip)
void gemm_init();                // kernel attributes, once per process
void gemm_describe(ConvProfile* p);

}  // namespace rvcx
