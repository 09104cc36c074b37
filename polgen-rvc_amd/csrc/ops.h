// Non-GEMM kernels (normalisation, scans, resampling, element-wise glue) + attention/GRU launchers.
#pragma once
#include "common.h"

namespace rvcx {

// ---- attention.hip
void launch_attention(const float* q, const float* k, const float* v, float* out, int B, int H, int D, int T,
                      int ld, long in_bs, long out_bs, float scale, const float* emb_rel_k, const float* emb_rel_v, int window,
                      const int* lens, float* scratch, float* split_scratch, hipStream_t stream,
                      int* ovf = nullptr /* device error word (fp16-split overflow bit), see conv.h */,
                      int* ovf_layer = nullptr, int seq = 0 /* this call's own word + launch number (conv.h) */,
                      bool allow_h3 = true /* false: the exact-fp32 kernel (a call pinned after an overflow) */);
size_t attention_scratch_floats(int B, int H, int T, int window);
size_t attention_split_floats(int B, int H, int T);
double attention_flops(int B, int H, int D, int T);

// ---- gru.hip
// gi: (B, T, 2*3H) input projections (+b_ih) for [fwd | rev]; whh_t: (2, H, 3H); bhh: (2, 3H);
// y: (B, 2H, T) channel-first (feeds the classifier conv directly)
// scratch: bigru_scratch_bytes(B) of device memory for the inter-CU exchange; err: device flag set to 1 if
// the cluster kernel timed out waiting for a partner workgroup (null scratch/err -> single-CU kernel)
size_t bigru_scratch_bytes(int B);
// one-time check (per device, synchronous) that a plain store inside one XCD is seen by a partner's sc1 poll -- what the
// cluster kernel's publish relies on; 0 switches the device to the write-through publish.  bigru_probe_state: 1 holds,
// 0 does not, -1 undecided / not probed.
int bigru_probe_publish();
int bigru_probe_state();
// lens (device, B ints or null): item b runs lens[b] <= T steps -- the reverse direction starts at frame lens[b] - 1 --
// while rows stay T frames apart; y beyond an item's length is not written
void launch_bigru(const float* gi, const float* whh, const float* bhh, float* y, int B, int T, int H,
                  void* scratch, int* err, hipStream_t stream, const int* lens = nullptr);

// ---- ops.hip
// LayerNorm over channels of (B,C,T): y = (x-mean)/sqrt(var+eps)*gamma+beta  [normalization.py:13-16]
void launch_layernorm_c(const float* x, const float* gamma, const float* beta, float* y, int B, int C, int T,
                        float eps, const int* lens, hipStream_t s);
// ---- time-major transformer section (gemm.hip): rows r = b T + t, channels contiguous
// LayerNorm of fp32 rows (ld_x floats apart) -> fp32 rows y (or null) and / or the fp16 hi/lo split rows ys the next GEMM
// stages (or null).  ovf / ovf_next / seq: range guard of the split form (conv.h), stamped for the consuming layer.
void launch_layernorm_tm(const float* x, long ld_x, const float* gamma, const float* beta, float* y, long ld_y, void* ys,
                         long ld_ys, long rows, int C, float eps, int* ovf, int* ovf_next, int seq, hipStream_t s);
// channel-first (B, C, T) fp32 (batch stride x_bs) -> time-major fp32 rows and / or split rows
void launch_cf_to_tm(const float* x, long x_bs, float* y, long ld_y, void* ys, long ld_ys, int B, int C, int T, int* ovf,
                     int* ovf_next, int seq, hipStream_t s);
// per-(b,c) normalisation over time + GELU (GroupNorm(C,C) of the HuBERT extractor)
// lens (device, B ints or null): statistics over the item's first lens[b] frames only (rows stay T apart); frames
// beyond are stored as zeros
void launch_groupnorm_gelu(const float* x, const float* gamma, const float* beta, float* y, int B, int C, int T,
                           float eps, hipStream_t s, const int* lens = nullptr);
// the same with the output already split for a conv_h3 tile that reads ConvArgs::x_split (stats: B*C*2 floats of scratch)
// HuBERT's first extractor layer (Cin = 1 FIR) + GroupNorm + GELU + split store without the fp32 map in between (ops.hip);
// Cp: floats between two taps of the packed fp32 weights (cin_gp * cout_gp); part: hubert_conv0_part_doubles(B, C, T)
// doubles of scratch, stats: B * C * 2 floats
size_t hubert_conv0_part_doubles(int B, int C, int T);
void launch_hubert_conv0_gn_gelu_split(const float* wav, long wav_bs, const float* w, int K, int stride, int Cp, const float* gamma,
                                       const float* beta, double* part, float* stats, void* y_split, int B, int C, int T,
                                       float eps, hipStream_t s, const int* lens, int* ovf, int* ovf_layer, int seq);
void launch_groupnorm_gelu_split(const float* x, const float* gamma, const float* beta, float* stats, void* y_split, int B,
                                 int C, int T, float eps, hipStream_t s, const int* lens, int* ovf, int* ovf_layer, int seq);
// (B,R,Cc) -> (B,Cc,R)
void launch_transpose(const float* x, float* y, int B, int R, int Cc, hipStream_t s);
// x[c][t] = lrelu((x[c][t] + emb[pitch[t]][c]) * scale, slope) * (t<len)     [encoders.py:116-123]
void launch_embed_pitch(float* x, const float* emb, const int* pitch, int B, int C, int T, float scale,
                        float slope, const int* lens, hipStream_t s);
// z = (m + exp(logs) * noise * 0.66666) * mask ; stats = [m ; logs] (B,2C,T)   [synthesizers.py:174]
void launch_sample_z(const float* stats, const float* noise, float* z, int B, int C, int T, const int* lens,
                     hipStream_t s);
// y[c] = x[C-1-c]                                                            [residuals.py:81-88]
void launch_flip_channels(const float* x, float* y, int B, int C, int T, hipStream_t s);
// acts = tanh(a[:H]+g[:H]) * sigmoid(a[H:]+g[H:]); a (B,2H,T), g (B,gstride) at offset goff  [commons.py:79-86]
void launch_wn_gate(const float* a, const float* g, int goff, int gstride, float* acts, int B, int H, int T,
                    hipStream_t s);
// x = (x + rs[:H]) * mask ; out (+)= rs[H:]   (last layer: out += rs)          [modules.py:76-83]
void launch_wn_res_skip(float* x, float* out, const float* rs, int B, int H, int T, int last, int first,
                        const int* lens, hipStream_t s);
// x1 = (x1 - m) * mask on channels [half, 2*half) of x (B,2*half,T)             [residuals.py:226-227]
void launch_coupling_sub(float* x, const float* m, int B, int half, int T, const int* lens, hipStream_t s);
// x (B,C,T) *= mask
void launch_mask(float* x, int B, int C, int T, const int* lens, hipStream_t s);
// x[b][c][t] += g[b][c]
void launch_add_channel_bias(float* x, const float* g, int B, int C, int T, hipStream_t s);
// NSF harmonic source: f0 (B,T) -> har (B,T*upp) = tanh(w*(sine*uv + namp*noise) + b)  [generators.py:117-156, nsf.py:36-40]
void launch_sine_source(const float* f0, const float* noise, float* har, int B, int T, int upp, float sr,
                        const float* lin_wb, const int* lens, double* scratch, hipStream_t s);
// standard normal noise, Philox4x32-10 + Box-Muller
void launch_randn(float* out, size_t n, uint64_t seed, uint64_t offset, hipStream_t s);
// tanh(conv_post) is done in the conv epilogue; final leaky_relu(0.01) is its prologue.

// ---- RMVPE helpers
// reflect-pad 1-D signals: x (B,n) -> y (B,n+2p) written with batch stride y_bs
// ns (device, B ints or null): item b holds ns[b] <= n samples (rows of x stay x_bs = n apart unless given); its padded
// signal is ns[b] + 2p long and the rest of the row (up to n + 2p) is written as zeros
void launch_reflect_pad(const float* x, float* y, int B, int n, int p, long y_bs, hipStream_t s, const int* ns = nullptr,
                        long x_bs = 0);
// small int arrays for the kernels' per-item length arguments: written by a kernel from by-value arguments (no host
// buffer has to outlive the call, no DMA in the launch sequence)
int* dev_ints(Arena& A, const int* v, int n, hipStream_t s);
void set_dev_ints(int* dst, const int* v, int n, hipStream_t s);      // the same into memory the caller owns
inline int* dev_ints(Arena& A, const std::vector<int>& v, hipStream_t s) { return dev_ints(A, v.data(), (int)v.size(), s); }
std::vector<int*> dev_ints_many(Arena& A, const std::vector<std::vector<int>>& vs, hipStream_t s);   // one launch for all of them
// |STFT|: ft (B, 2*nb, F) -> mag (B, nb, F) = sqrt(re^2 + im^2 + eps)   (eps: FCPE.py:147)
void launch_magnitude(const float* ft, float* mag, int B, int nb, int F, hipStream_t s, float eps = 0.f);
// log(clamp(mel,1e-5)) -> BN affine -> row-padded (B,1,Tp,Wp=130) with reflect padding of frames to Tp
// fs / tps (device, B ints each, or null): item b has fs[b] <= F frames reflect-padded to tps[b] <= Tp rows; rows beyond
// tps[b] are zero
void launch_mel_post(const float* mel, float* out, int B, int nmel, int F, int Tp, const float* bn,
                     hipStream_t s, const int* fs = nullptr, const int* tps = nullptr);
// y = log(max(x, floor))
void launch_log_clamp(const float* x, float* y, long n, float floor, hipStream_t s);
// 2x2 average pool on row-padded maps: (B*C, H, Wp) -> (B*C, H/2, W/2+2)
void launch_avgpool2(const float* x, float* y, int planes, int H, int Wp, long x_ps, long y_ps, hipStream_t s);
// copy channels of row-padded maps between buffers with different batch strides
void launch_copy_strided(const float* x, float* y, int B, long n, long x_bs, long y_bs, hipStream_t s);
// cnn output (B,3,T,Wp) -> GRU input (B, 3*128, T) channel-first (k = c*128+f)
void launch_gru_input(const float* x, float* y, int B, int Cc, int T, int Wp, hipStream_t s);
// salience (B,T,360) [row stride ld] -> f0 Hz (B,T)   [RMVPE.py:472-476,494-516]
void launch_decode_f0(const float* sal, float* f0, int B, int T, int ld, float thred, float f0_min, float f0_max,
                      hipStream_t s);
// f0 (n) -> f0 * 2^(pitch/12), coarse 1..255      [pipeline.py:183,193-201] (float64 arithmetic)
void launch_f0_coarse(const float* f0_in, float* f0_out, int* coarse, int n, double pitch, double f0_min,
                      double f0_max, hipStream_t s);

// f0 file override (pipeline.py:185-191): frames [start, start+count) of f0 / coarse (n frames) take rep (device, float64)
void launch_f0_override(const double* rep, int count, int start, float* f0, int* coarse, int n, double f0_min,
                        double f0_max, hipStream_t s);
std::vector<double> f0_file_track(const float* tbl, int rows);   // host: np.interp restated (see ops.hip)

// ---- audio.hip: librosa.resample stand-in (kind 0: "kaiser_hq", a Kaiser design to soxr_hq's published targets -- the
// default; kind 1: resampy's published "kaiser_best"; see the file header)
struct ResampleFilter {
  const double* win = nullptr;     // device: interp_win, nwin doubles
  const double* delta = nullptr;   // device: forward differences
  double scale = 1.0, time_increment = 1.0;
  int index_step = 512;            // kaiser_best: int(scale * table)
  int table = 512, nwin = 0, exact = 0;
};
long resample_out_len(long n, int sr_in, int sr_out);                // int(n * sr_out / sr_in)
int resample_default_kind();                                         // RVCX_RESAMPLER=kaiser_best -> 1, else 0
ResampleFilter make_resample_filter(Arena& A, int sr_in, int sr_out, hipStream_t s, int kind = -1);
// x: (n frames, channels) interleaved float64, averaged over the channels on the fly; y: n_out mono samples
void launch_resample_f64(const ResampleFilter& f, const double* x, long n, int channels, double* y, long n_out,
                         hipStream_t s);
void launch_resample_f32(const ResampleFilter& f, const float* x, long n, float* y, long n_out, hipStream_t s);

// ---- pipeline glue
// feats (C,T) -> x2 nearest upsample, protect mix; writes phone (C, 2T') cropped to p_len  [pipeline.py:252-270]
// ld_in / ld_out (0 = Th / p_len): row strides of feats / out when the item sits in a wider batch row
void launch_upsample_protect(const float* feats, const float* feats0, const float* pitchf, float* out, int C,
                             int Th, int p_len, float protect, int use_protect, hipStream_t s, int ld_in = 0,
                             int ld_out = 0);

}  // namespace rvcx
