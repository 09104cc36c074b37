// Model containers (weights folded + packed into the slab) and their forward passes.
#pragma once
#include <functional>

#include "../../include/rvcx.h"
#include "ctx.h"
#include "layers.h"

namespace rvcx {

// ------------------------------------------------------------------------------ Synthesizer
struct SynthModel {
  std::shared_ptr<WeightRegion> region = std::make_shared<WeightRegion>();   // freed with the model
  rvcx_synth_cfg cfg{};
  int upp = 1;
  // frames of z an output sample of the NSF decoder can depend on, each side (conv_pre + every stage's ConvTranspose1d and
  // ResBlocks + conv_post, rounded up + 2): what SynthIO::dec_skip must leave around the samples the caller keeps
  int dec_rf_frames = 0;
  // TextEncoder (rvc/lib/algorithm/encoders.py:76-126)
  ConvW emb_phone;
  const float* emb_pitch = nullptr;  // (256, hidden)
  struct EncLayer {
    ConvW qkv, o, ffn1, ffn2;
    AttFlag att;                                         // this layer's attention call (fp16-split range guard)
    const float *rel_k = nullptr, *rel_v = nullptr;      // (21, hidden/heads)
    const float *g1 = nullptr, *b1 = nullptr, *g2 = nullptr, *b2 = nullptr;
  };
  std::vector<EncLayer> enc;
  ConvW proj;
  // ResidualCouplingBlock (rvc/lib/algorithm/residuals.py:109-232), layers 0..3 = flows.{0,2,4,6}
  struct Flow {
    ConvW pre, post, cond, in_l[3], rs_l[3];
  };
  Flow flows[4];
  // GeneratorNSF (rvc/lib/algorithm/nsf.py:43-144)
  ConvW conv_pre, cond, conv_post;
  const float* lin_wb = nullptr;   // device {weight, bias} of m_source.l_linear (1 -> 1)
  struct Stage {
    ConvT1dW up;
    ConvW noise;
    int noise_stride = 1, noise_pad = 0;
    // round 5: a long noise FIR (stage 0: k = 80, stride 40) re-indexed like the STFT -- a dense k = 3 conv over the
    // hop-major transposed source X2[c][q] = har[stride q + c] (c < stride, zero rows up to `noise_dense_cin`); cin 0 = unused
    ConvW noise_dense;
    int noise_dense_cin = 0;
    int ch = 0;
    ConvW c1[4][3], c2[4][3];
  };
  std::vector<Stage> stages;
  const float* emb_g = nullptr;  // (spk, gin)
};

std::unique_ptr<SynthModel> synth_load(Ctx& c, const rvcx_synth_cfg& cfg, const TensorTable& t);

struct SynthIO {
  int B = 1, T = 0;
  const int* lens_host = nullptr;     // (B) or null = all T
  const float* phone_ct = nullptr;    // device (B, input_dim, T) channel-first
  const int* pitch = nullptr;         // device (B,T)
  const float* pitchf = nullptr;      // device (B,T)
  const int* sid_host = nullptr;      // (B)
  const float* z_noise = nullptr;     // device (B,inter,T)
  const float* src_noise = nullptr;   // device (B,T*upp)
  float* out = nullptr;               // device (B, T*upp)
  // Round 6: the decoder evaluates frames [dec_skip, len - dec_skip) of every item only (the source branch, TextEncoder and
  // flow stay whole); output samples outside that window read as zero.  VC.pipeline throws t_pad_tgt samples of every
  // decoder call away at both ends (pipeline.py:432-447): with dec_skip <= t_pad frames - SynthModel::dec_rf_frames the
  // samples it keeps are the ones the full evaluation gives (the decoder is convolutional; its source is computed whole).
  int dec_skip = 0;
  // optional taps for parity tests (device, may be null)
  float* stats_out = nullptr;         // (B, 2*inter, T) = [m_p ; logs_p]
  float* z_out = nullptr;             // (B, inter, T)
  hipEvent_t ev_decoder = nullptr;    // optional: recorded on the main stream when the NSF decoder section begins
};
size_t synth_arena_bytes(const SynthModel& m, int B, int T);
// stage_ev (optional): 4 caller-created events recorded at {start, enc_p done, flow done, decoder done} -- no sync
void synth_forward(Ctx& c, const SynthModel& m, const SynthIO& io, hipEvent_t* stage_ev /*4 or null*/);

// ------------------------------------------------------------------------------ RMVPE
struct RmvpeModel {
  std::shared_ptr<WeightRegion> region = std::make_shared<WeightRegion>();   // freed with the model
  rvcx_rmvpe_cfg cfg{};
  ConvW stft;        // (1026, 1, 1024) Hann-windowed Fourier basis, stride 160
  ConvW melfb;       // (128, 513) as a 1x1 conv
  const float* bn0 = nullptr;      // device {scale, shift} of the folded encoder.bn
  struct Block {
    ConvW c1, c2, sc;   // 3x3 + folded BN (x2), optional 1x1 shortcut
    bool has_sc = false;
    int cin = 0, cout = 0;
  };
  std::vector<std::vector<Block>> enc, inter;
  struct Dec {
    ConvT2dW up;
    std::vector<Block> blocks;
    int cout = 0;
  };
  std::vector<Dec> dec;
  ConvW cnn;         // 3x3, 16 -> 3, bias
  ConvW gru_ih;      // (2*3H, 384) both directions stacked, bias = b_ih
  const float* whh_t = nullptr;  // (2, H, 3H)
  const float* bhh = nullptr;    // (2, 3H)
  ConvW fc;          // (360, 512) + bias, sigmoid
};
std::unique_ptr<RmvpeModel> rmvpe_load(Ctx& c, const rvcx_rmvpe_cfg& cfg, const TensorTable& t);
size_t rmvpe_arena_bytes(const RmvpeModel& m, int B, int64_t n);
// audio: device (B,n) f32.  f0: device (B, 1+n/160).  hidden: device (B, frames, 360) or null.
// mel_out (optional): device (B, 128, frames) log-mel spectrogram (MelSpectrogram.forward, RMVPE.py:412-439)
// `after_shallow` (optional) runs on the host once the mel front end and the first U-Net encoder levels (the long
// launches) are enqueued: the pipeline enqueues HuBERT there, so that neither branch waits for the host
// ns_host (optional, B ints): a ragged batch -- item b holds ns_host[b] <= n samples in its row of n; its f0 row has
// 1 + ns_host[b]/160 valid frames and is bit-identical to what the call returns for that item alone in a row of n
void rmvpe_block_op(Ctx& c, const ConvW& c1, const ConvW& c2, const ConvW* sc, const float* x, float* y, float* tmp1,
                    float* tmp2, int B, int H, int Wp, const int* rows, hipStream_t s);
void rmvpe_forward(Ctx& c, const RmvpeModel& m, int B, const float* audio, int64_t n, float thred, float f0_min,
                   float f0_max, float* f0, float* hidden, hipStream_t s, float* mel_out = nullptr,
                   const std::function<void()>* after_shallow = nullptr, const int* ns_host = nullptr);

// ------------------------------------------------------------------------------ FCPE
struct FcpeModel {
  std::shared_ptr<WeightRegion> region = std::make_shared<WeightRegion>();   // freed with the model
  rvcx_fcpe_cfg cfg{};
  ConvW stft;        // shared Hann-windowed Fourier basis (n_fft 1024, hop 160)
  ConvW melfb;       // (128, 513) slaney-scale filterbank as a 1x1 conv
  ConvW stack0, stack3;
  const float *gn_g = nullptr, *gn_b = nullptr;
  struct Layer {
    const float *ln_g = nullptr, *ln_b = nullptr, *cln_g = nullptr, *cln_b = nullptr;
    ConvW qkv, out, pw1, pw2;      // to_q|to_k|to_v stacked; to_out; conformer 1x1 convs
    const float* proj = nullptr;   // (nb_features, dim_head) random-feature projection (a checkpoint buffer)
    const float *dw_w = nullptr, *dw_b = nullptr;   // depth-wise conv (inner, K), bias
  };
  std::vector<Layer> layers;
  const float *norm_g = nullptr, *norm_b = nullptr;
  ConvW dense;                     // weight-norm folded Linear(n_chans, 360)
  const float* cent_table = nullptr;
};
ConvW make_stft_conv(Ctx& c);      // rmvpe.hip

// ------------------------------------------------------------------------------ CREPE (crepe.hip; torchcrepe's model.Crepe)
struct CrepeModel {
  std::shared_ptr<WeightRegion> region = std::make_shared<WeightRegion>();
  int filters[6] = {0, 0, 0, 0, 0, 0};
  int in_features = 0;               // 4 * filters[5]
  ConvW conv[6];                     // conv[0]: the stride-4 / 512-tap layer as a dense 128-tap conv over 4 phase channels
  const float* bn_scale[6] = {};     // eval BatchNorm as x * scale + shift (applied after the ReLU)
  const float* bn_shift[6] = {};
  ConvW classifier;                  // Linear(in_features, 360) + sigmoid
  const double* band = nullptr;      // log of the Viterbi transition band, (360, 23)
  double log_far = 0.0, log_init = 0.0;
};
std::unique_ptr<CrepeModel> crepe_load(Ctx& c, const TensorTable& t);
int crepe_frames(int64_t n, int hop);
double crepe_quantile999(std::vector<float>& abs_scratch);     // host: np.quantile(|x|, 0.999); clobbers its argument
size_t crepe_arena_bytes(const CrepeModel& m, int64_t n, int hop);
void crepe_decode(Ctx& c, const CrepeModel& m, const float* probs, long F, int batch, float fmin, float fmax,
                  const float* dither, float* pitch, int* bins_out, hipStream_t s);
void crepe_forward(Ctx& c, const CrepeModel& m, const float* x, int64_t n, float scale, int hop, float fmin, float fmax,
                   const float* dither, float* pitch, float* probs_out, int* bins_out, hipStream_t s);
void launch_crepe_dither(float* out, long n, uint64_t seed, uint64_t offset, hipStream_t s);
void launch_crepe_resize(const float* pitch, long L, long p_len, float* f0, hipStream_t s);
std::unique_ptr<FcpeModel> fcpe_load(Ctx& c, const rvcx_fcpe_cfg& cfg, const TensorTable& t);
size_t fcpe_arena_bytes(const FcpeModel& m, int B, int64_t n);
// FCPEInfer.__call__ (FCPE.py:739-745): audio device (B,n) f32 -> f0 device (B, n/160 + 1) Hz, 0 = below `threshold`.
// sal_out (optional): (B, frames, 360) sigmoid salience;  mel_out (optional): (B, 128, frames) log-mel
void fcpe_forward(Ctx& c, const FcpeModel& m, int B, const float* audio, int64_t n, float threshold, float* f0,
                  float* sal_out, float* mel_out, hipStream_t s, const std::function<void()>* after_stack = nullptr);
// FCPEF0Predictor.post_process (FCPE.py:841-867) + the tail of VC.get_f0 (pipeline.py:183-201): raw Hz (B, F_in) ->
// f0 (float32 of the float64 result) and coarse (1..255), p_len frames each, rows `out_stride` apart
void fcpe_post_coarse(Ctx& c, const float* f0raw, int B, int F_in, int p_len, float* f0_out, int* coarse, long out_stride,
                      double pitch, double f0_min, double f0_max, hipStream_t s);

// ------------------------------------------------------------------------------ HuBERT
struct HubertModel {
  std::shared_ptr<WeightRegion> region = std::make_shared<WeightRegion>();   // freed with the model
  rvcx_hubert_cfg cfg{};
  std::vector<ConvW> convs;
  const float *gn_g = nullptr, *gn_b = nullptr, *ln0_g = nullptr, *ln0_b = nullptr;
  ConvW proj, pos_conv;
  ConvW final_proj;                                      // Linear(embed, final_dim): v1 voice models only (pipeline.py:236)
  bool has_final_proj = false;
  const float *eln_g = nullptr, *eln_b = nullptr;
  struct Layer {
    ConvW qkv, o, fc1, fc2;
    AttFlag att;                                         // this layer's attention call (fp16-split range guard)
    const float *ln1_g = nullptr, *ln1_b = nullptr, *ln2_g = nullptr, *ln2_b = nullptr;
  };
  std::vector<Layer> layers;
};
std::unique_ptr<HubertModel> hubert_load(Ctx& c, const rvcx_hubert_cfg& cfg, const TensorTable& t);
int hubert_frames(const HubertModel& m, int64_t n);
size_t hubert_arena_bytes(const HubertModel& m, int B, int64_t n);
// wav: device (B,n).  feats_ct: device (B, embed, T') channel-first.
// `after_extractor` (optional) runs on the host right after the conv feature extractor has been enqueued:
// the pipeline uses it to enqueue RMVPE's ~330 small launches while those long convs keep the GPU busy.
// ns_host (optional, B ints): a ragged batch -- item b holds ns_host[b] <= n samples, zeros behind them; its first
// hubert_frames(ns_host[b]) feature frames are bit-identical to what the call returns for that item alone in a row of n
void hubert_forward(Ctx& c, const HubertModel& m, int B, const float* wav, int64_t n, int output_layer,
                    float* feats_ct, hipStream_t s, const std::function<void()>* after_extractor = nullptr,
                    long wav_bs = 0 /* element stride between the B signals, 0 = n */, const int* ns_host = nullptr);

// What VC.vc feeds the retrieval blend and the synthesizer (pipeline.py:228-236): out_dim == embed_dim -> the output of
// transformer layer 12 (RVC v2); out_dim == final_proj's width (256) -> final_proj(output of layer 9) (RVC v1).
// feats_ct: device (B, out_dim, T') channel-first.  Other arguments as hubert_forward.
void hubert_features_for(Ctx& c, const HubertModel& m, int out_dim, int B, const float* wav, int64_t n, float* feats_ct,
                         hipStream_t s, long wav_bs = 0, const int* ns_host = nullptr);

// ------------------------------------------------------------------------------ retrieval index
struct IndexData {
  std::shared_ptr<WeightRegion> region = std::make_shared<WeightRegion>();   // freed with the model
  ConvW mat;                    // big_npy (N, dim) packed as a 1x1 conv (N out channels)
  const float* rows = nullptr;  // (N, dim) row-major for the gather
  const float* norms = nullptr; // |b|^2 (N)
  float max_norm = 0.f;         // max |b|^2 (error bound of the pre-filter, index.hip)
  int* exhaustive = nullptr;    // device counter: queries that needed the exhaustive search (rvcx_index_exhaustive)
  int64_t n = 0;
  int dim = 0;
  // IVF (faiss "IVF{nlist},Flat" searched with nprobe = 1, what RVC's training writes): the coarse centroids and
  // the inverted list every stored vector belongs to.  nlist == 0: flat index.
  ConvW cent;                   // (nlist, dim) as a 1x1 conv
  const float* cent_norms = nullptr;
  const int* assign = nullptr;  // (N) list id per stored vector
  int nlist = 0;
};
std::unique_ptr<IndexData> index_load(Ctx& c, const float* big_npy, int64_t n, int dim, const float* centroids = nullptr,
                                      int nlist = 0, const int32_t* assign = nullptr);
size_t index_arena_bytes(const IndexData& ix, int T);
// feats_ct (dim, T) channel-first in/out (device); ids (T,8) int64 / dist (T,8) optional device outputs
void index_blend(Ctx& c, const IndexData& ix, float* feats_ct, int T, float index_rate, int64_t* ids,
                 float* dist, hipStream_t s);

// ------------------------------------------------------------------------------ host helpers
// weight-norm fold  w = g * v / ||v||  over all dims except `dim` (0 or 2)
std::vector<float> wn_weight(const TensorTable& t, const std::string& prefix, int dim = 0);

}  // namespace rvcx
