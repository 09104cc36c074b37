// Implicit-GEMM convolution on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact f32).
//
// One kernel family covers every conv / linear layer of the RVC hot path:
//   conv1d (any k, stride, dilation, groups), linear (k=1), ConvTranspose1d (polyphase:
//   a conv with stride*Cout output channels + shuffle store), Conv2d 3x3 on a row-padded
//   map (flattened to 1-D with taps at {-Wp-1..Wp+1}), ConvTranspose2d 3x3 s2 (polyphase,
//   4*Cout channels + 2-D shuffle store).
//
// GEMM view:  M = Cout (per group), N = output positions, K = ksize * Cin (per group).
// Weights are pre-packed on the host as Wp[g][kk][ci_pad][co_pad] (co contiguous) so the A
// fragment of lane (i = lane&31, h = lane>>5) for k-step (kk, ci pair cp) is
// Wp[kk][2cp+h][co0+i]; the B fragment is X[2cp+h][(n0+j)*stride + off(kk)] read from an
// LDS tile of the input (time contiguous -> conflict-free ds_read_b32).
#pragma once
#include "common.h"

namespace rvcx {

enum Act { ACT_NONE = 0, ACT_LRELU = 1, ACT_RELU = 2, ACT_GELU = 3, ACT_TANH = 4, ACT_SIGMOID = 5 };
enum OutMode { OUT_NORMAL = 0, OUT_SHUF1D = 1, OUT_SHUF2D = 2, OUT_TRANSPOSED = 3 };
enum Acc2 { ACC2_NONE = 0, ACC2_SET = 1, ACC2_ADD = 2, ACC2_ADD_DIV = 3 };

struct ConvArgs {
  const float* x = nullptr;
  const float* w = nullptr;      // packed Wp[g][kk][Cin_gp][Cout_gp]
  const void* w_h3 = nullptr;    // optional fp16 hi/lo split image of w (conv_h3.hip); null -> fp32 MFMA only
  // conv_h3 only: activations exchanged already split, XS[c/16][op {hi, S*lo}][(c%16)/8][t][8 halves] (same bytes as
  // fp32, 16-byte elements = exactly the kernel's LDS input-tile elements).  x_split replaces x, y_split replaces y.
  const void* x_split = nullptr;
  void* y_split = nullptr;
  const float* bias = nullptr;   // [groups*Cout_g] or null
  const float* res = nullptr;    // residual added after the activation (same indexing as y)
  float* y = nullptr;            // may be null when only y2 is wanted
  float* y2 = nullptr;           // secondary accumulator output (OUT_NORMAL only)
  const int* lens_in = nullptr;  // per-batch valid input positions (null -> Tin)
  const int* lens_out = nullptr; // per-batch valid output positions; beyond -> 0 is stored
  int B = 1;
  int Cin_g = 0, Cin_gp = 0, Cout_g = 0, Cout_gp = 0, groups = 1;
  int Tin = 0;                   // input positions per (batch, channel)
  int Nout = 0;                  // output positions computed per batch (GEMM N)
  int ksize = 1, kw = 1, rowpitch = 0, dil = 1, pad = 0, stride = 1;
  long x_bs = 0, y_bs = 0, res_bs = 0, y2_bs = 0;
  int x_cs = 0, y_cs = 0, res_cs = 0, y2_cs = 0;
  int pre_act = ACT_NONE;
  float pre_slope = 0.f;
  int act = ACT_NONE;
  float act_slope = 0.f;
  int out_mode = OUT_NORMAL;
  int sh_s = 1, sh_pad = 0, sh_cout = 0, sh_tout = 0;  // OUT_SHUF1D
  int wp_in = 0, wp_out = 0;                            // OUT_SHUF2D
  int zero_wp = 0;               // >0: force the pad columns of a row-padded 2-D map to 0
  int acc2_mode = ACC2_NONE;
  float acc2_div = 1.f;
  // OUT_SHUF1D only: the NSF noise conv of the stage (Conv1d(1, C, k = nz_k, stride nz_stride, padding nz_pad) over the
  // harmonic source, nsf.py:128-129) evaluated inside the epilogue instead of being read back as `res`: adds
  // nz_b[c] + sum_j nz_w[j * nz_wstride + c] * har[t * nz_stride + j - nz_pad] to output (c, t).  The thin ConvTranspose1d
  // kernel (convt_thin.hip, nz_k <= 4) and the shuffle-store epilogues of the tiled kernels (conv_device.h) implement it.
  const float* nz_har = nullptr;   // (B, nz_len_row) harmonic source
  const float* nz_w = nullptr;     // packed Cin = 1 weights: tap j of channel c at nz_w[j * nz_wstride + c]
  const float* nz_b = nullptr;
  const int* nz_lens = nullptr;    // per-item valid source samples (null: nz_len)
  int nz_k = 0, nz_stride = 1, nz_pad = 0, nz_wstride = 0;
  long nz_bs = 0;
  int nz_len = 0;
  // split-K scratch offered by the caller (per stream); the launcher decides whether to use it
  float* part = nullptr;
  long part_cap = 0;             // floats
  long part_cap_item = 0;        // floats per batch item the split-K DECISION may count on (see launch_conv_h3)
  // filled by the launcher
  int ci_chunk = 0, kk_chunk = 0, wrow = 0, off_min = 0;
  int splitk = 1;
  long long* trace = nullptr;    // RVCX_ABLATION: per-workgroup {hw_id, xcc, t_start, t_stage0, t_mainloop_end, t_end}
  int stagger = 0;               // start delay per resident-workgroup slot in 100 MHz ticks (first round only)
  int stagger_blocks = 0;        // workgroups of the first round (linear id below this get the delay)
  int* ovf = nullptr;            // device error word: bit 1 is set when an activation does not fit the fp16 hi/lo split
  int* ovf_layer = nullptr;      // the LAYER's own device word (ConvW::ovf_word): receives 0x7fffffff - seq, so the host finds the
  int seq = 0;                   // first offender of a call in launch order and pins that layer to the exact-fp32 kernels
  int dbg = 0;                   // timing ablations only (RVCX_CONV_DBG): 1 skip weight staging, 2 skip input staging, 4 skip MFMAs
};

// tap offset of kernel element kk relative to n*stride (before subtracting off_min)
inline int conv_tap_off(const ConvArgs& a, int kk) {
  return (kk / a.kw) * a.rowpitch + (kk % a.kw) * a.dil - a.pad;
}

// One ResBlock1 step fused into one kernel (resblock.hip): y = x + c2(lrelu(c1(lrelu(x)) + b1)) + b2
struct PairArgs {
  const float* x = nullptr;      // input and residual, (B, C, T) with batch stride bs / channel stride cs
  float* y = nullptr;            // output (same strides) or null when only y2 is wanted
  float* y2 = nullptr;           // running mean over the ResBlocks of a stage (acc2_mode), same strides
  const void* w1 = nullptr;      // fp16 hi/lo images of the two convs (ConvW::w_h3)
  const void* w2 = nullptr;
  const float* b1 = nullptr;
  const float* b2 = nullptr;
  const int* lens = nullptr;     // per-item valid positions (null: T)
  int B = 1, C = 0, T = 0;
  long bs = 0;
  int cs = 0;
  int k = 1, dil = 1;            // c1: k taps, dilation dil; c2: k taps, dilation 1
  float slope = 0.1f;
  int acc2_mode = ACC2_NONE;
  float acc2_div = 1.f;
  int xcd_order = 1;             // tiles in XCD-contiguous order (resblock.hip); 0: tile = blockIdx.x (RVCX_PAIR_XCD=0)
  int* ovf = nullptr;            // as ConvArgs::ovf
  int* ovf_layer = nullptr;      // as ConvArgs::ovf_layer (the pair reports as its first conv)
  int seq = 0;
  long long* trace = nullptr;    // profiling (rvcx_bench_resblock_pair, RVCX_PAIR_TRACE): 8 s_memrealtime stamps per workgroup
};
// A whole ResBlock1 with kernel size 3 (three steps, dilations dil[0..2]) in one kernel (resblock3.hip)
struct Block3Args {
  const float* x = nullptr;      // (B, C, T), batch stride bs, channel stride cs
  float* y = nullptr;            // block output or null when only y2 is wanted
  float* y2 = nullptr;           // running mean over the ResBlocks of a stage (acc2_mode)
  const void* w1[3] = {nullptr, nullptr, nullptr};   // fp16 hi/lo images of convs1[s] / convs2[s]
  const void* w2[3] = {nullptr, nullptr, nullptr};
  const float* b1[3] = {nullptr, nullptr, nullptr};
  const float* b2[3] = {nullptr, nullptr, nullptr};
  int dil[3] = {1, 3, 5};
  const int* lens = nullptr;
  int B = 1, C = 0, T = 0;
  long bs = 0;
  int cs = 0;
  float slope = 0.1f;
  int acc2_mode = ACC2_NONE;
  float acc2_div = 1.f;
  int xcd_order = 1;
  bool any_shape = false;        // kernel-level tests: every shape the kernel supports, not only the ones the pipeline sends it
  int* ovf = nullptr;
  int* ovf_layer = nullptr;
  int seq = 0;
};
bool resblock3_ok(const Block3Args& a);
void launch_resblock3(const Block3Args& a, hipStream_t stream);
void conv_launch_block3(const Block3Args& a, double flops, hipStream_t stream);   // + profile record (conv.hip)
int resblock3_n1(int C);
constexpr int kBlock3Slot = 62;

// fp16 hi/lo split kernels hold activations as fp16 halves: |x| >= 65504 (attention K / V: >= 255) would become
// inf.  Every split kernel checks what it converts, raises bit 1 of the context's device error word and stamps its
// layer's own word with the launch sequence number.  The API entry point then pins the FIRST offending layer of the
// call to the exact-fp32 kernels for the life of its model (a region flag: sticky) and repeats the call -- once per
// model and layer, not once per request.  g_force_fp32 (thread-local: everything on fp32) is the last resort.
constexpr float kH3ActLimit = 6.0e4f;
constexpr int kErrGruTimeout = 1, kErrH3Overflow = 2;
extern thread_local bool g_force_fp32;
extern thread_local int g_gru_drop_member;   // test hook: the next BiGRU cluster launch loses one workgroup (a real device-side time-out)
extern thread_local bool g_gru_no_cluster;   // this attempt of the API call runs the single-workgroup BiGRU kernel (gru.hip)
bool resblock_pair_enabled();                                // RVCX_FUSE (default on) and the h3 kernels enabled
bool resblock_pair_ok(const PairArgs& a);
void launch_resblock_pair(const PairArgs& a, hipStream_t stream);              // raw launch (resblock.hip)
void resblock_pair_init();                                                     // kernel attributes, once per process
void conv_launch_pair(const PairArgs& a, double flops, hipStream_t stream);    // + profile record (conv.hip)
int resblock_pair_slot(int C);
struct ConvProfile;
void resblock_pair_describe(ConvProfile* p);

struct ConvProfile {
  static constexpr int kMaxTiles = 72;   // 0-7 generic tiles, 8.. stride-1 family (sb), 24.. (DMA double-buffered)
  long launches[kMaxTiles] = {0};
  double flops[kMaxTiles] = {0};
  double ms[kMaxTiles] = {0};
  int bm[kMaxTiles] = {0}, bn[kMaxTiles] = {0}, halo[kMaxTiles] = {0};
};
void conv_profile_begin();                          // start recording one event pair per conv launch
void conv_profile_end(ConvProfile* out);            // sync, accumulate, stop
const char* conv_profile_csv();                     // per-launch table of the last profile (CSV text)
void conv_init();                                   // raise dynamic-LDS limits once
void launch_conv(ConvArgs a, hipStream_t stream);   // picks kernel family + tile, launches
int launch_conv_fast(ConvArgs& a, hipStream_t stream);    // stride-1 compile-time-tiled family; profile slot or -1
void conv_fast_describe(ConvProfile* p);
int launch_conv_h3(ConvArgs& a, int halo, int off_min, hipStream_t stream);   // fp16x3 split kernels; slot or -1
bool convt_thin_ok(const ConvArgs& a);                       // ConvTranspose1d(C -> C/2, k = 4, s = 2) on the thin kernel?
void launch_convt_thin(const ConvArgs& a, hipStream_t stream);
void convt_thin_init();
constexpr int kConvtThinSlot = 61;
bool conv3_thin_ok(const ConvArgs& a);                       // 3x3, C = 16 / 32, many positions: the streaming kernel (conv3_thin.hip)?
void launch_conv3_thin(const ConvArgs& a, hipStream_t stream);
constexpr int kConv3ThinSlot = 59;
bool conv_deep_ok(const ConvArgs& a);                        // long K, few positions per item: split-K inside the workgroup (conv_deep.hip)
void launch_conv_deep(const ConvArgs& a, hipStream_t stream);
constexpr int kConvDeepSlot = 63;
bool conv_h3_enabled();      // RVCX_H3 on and the calling thread is not in an exact-fp32 rerun
bool conv_h3_configured();   // RVCX_H3 on (what checkpoint loading looks at)
bool conv_h3_split_ok(const ConvArgs& a);   // may this launch read / write pre-split activations?
void launch_splitk_finish(const ConvArgs& a, hipStream_t stream);   // deterministic reduction of the split-K slabs + epilogue
void conv_h3_describe(ConvProfile* p);
struct ConvOverride { int tile = -1, variant = -1, splitk = -1; };
extern ConvOverride g_conv_override;   // tuning sweeps only (rvcx_conv_override)
void conv_fast_init();
double conv_flops(const ConvArgs& a);               // 2*M*N*K of the *real* (unpadded) problem

// ---------------- host-side weight packing ------------------------------------------------
int conv_cin_pad(int cin_g);
int conv_cout_pad(int cout_g);
// w: (Cout, Cin_g, K) row-major, Cout = groups*Cout_g.  Returns Wp[g][kk][Cin_gp][Cout_gp].
std::vector<float> pack_conv_weight(const float* w, int cout, int cin_g, int k, int groups);
// ConvTranspose1d weight (Cin, Cout, K), stride s, padding p -> polyphase conv with s*Cout
// output channels and M = ceil((K + extra)/s) taps on the low-rate input.  See conv.hip.
struct PolyPhase1d {
  std::vector<float> w;     // packed
  std::vector<float> bias;  // replicated per phase (s*Cout)
  int taps = 0;             // ksize of the equivalent conv
  int pad = 0;              // its left padding (taps look back: x[q - m])
  int cout_total = 0;
};
PolyPhase1d pack_convtranspose1d(const float* w, const float* bias, int cin, int cout, int k, int s,
                                 int p);

}  // namespace rvcx
