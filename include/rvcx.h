/* rvcx.h -- C ABI of librvcx.so: the MI355X-native RVC v2 inference hot path.
 *
 * The reference (Bebra777228/PolGen-RVC) is pure Python; it has no FFI of its own.  Each
 * entry point below names the reference interface (file:line under /root/reference) whose
 * work it replaces; the Python mirror in polgen-rvc_amd/infer/{infer,pipeline}.py binds
 * them with ctypes behind the reference's own call signatures.
 *
 * Conventions: return 0 on success, negative on error (message via rvcx_last_error);
 * nothing throws across the ABI.  One context per GPU.  Every entry point takes the
 * context's own (recursive) mutex: threads that share a context QUEUE -- the reference
 * builds fresh model objects per request (rvc/scripts/voice_conversion.py:71-100), so
 * its concurrent requests are safe, and they stay safe here -- while different contexts
 * are fully concurrent.  Throughput comes from one rvcx_convert_batch call over many
 * utterances, not from threads.  rvcx_destroy must not race with any other call.  "hd" pointers may
 * be host or device memory (copied with hipMemcpyDefault); everything else is host.
 * All tensors are dense row-major float32 unless stated.
 */
#ifndef RVCX_H
#define RVCX_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rvcx_ctx rvcx_ctx;

/* one checkpoint tensor, borrowed for the duration of the load call only */
typedef struct {
  const char* name;
  const void* data;
  int32_t dtype; /* 0 = float32, 1 = float16, 2 = int64 */
  int32_t ndim;
  int64_t shape[4];
} rvcx_tensor;

/* Synthesizer(*cpt["config"]) hyper-parameters -- rvc/infer/infer.py:86-97,
 * rvc/lib/algorithm/synthesizers.py:14-36 */
typedef struct {
  int32_t inter_channels, hidden_channels, filter_channels, n_heads, n_layers, kernel_size;
  int32_t n_resblocks;          /* len(resblock_kernel_sizes) (<= 4) */
  int32_t res_kernels[4];
  int32_t res_dilations[4][3];
  int32_t n_ups;                /* len(upsample_rates) (<= 6) */
  int32_t up_rates[6], up_kernels[6];
  int32_t up_initial_channel, spk_embed_dim, gin_channels, sr, input_dim;
} rvcx_synth_cfg;

/* E2E(n_blocks, n_gru, kernel_size=(2,2), en_de_layers, inter_layers, in_channels,
 * en_out_channels) -- rvc/lib/predictors/RMVPE.py:340-352,452 */
typedef struct {
  int32_t n_blocks, en_de_layers, inter_layers, en_out_channels;
} rvcx_rmvpe_cfg;

/* FCPE(input_channel, out_dims, n_layers, n_chans) -- rvc/lib/predictors/FCPE.py:551-627, built from fcpe.pt's
 * "config" block at FCPE.py:715-733; heads / dim_head / nb_features / dw_kernel are the module defaults
 * (FCPE.py:445-446, 435, 315), read off the tensor shapes; mel_fmin / mel_fmax from config["mel"] */
typedef struct {
  int32_t n_layers, n_chans, input_channel, out_dims;
  int32_t heads, dim_head, nb_features, dw_kernel;
  float mel_fmin, mel_fmax;
} rvcx_fcpe_cfg;

/* fairseq HubertModel (hubert_base) geometry -- loaded at rvc/infer/infer.py:67-74 */
typedef struct {
  int32_t conv_dim, n_conv;
  int32_t conv_kernels[8], conv_strides[8];
  int32_t embed_dim, ffn_dim, heads, layers, pos_kernel, pos_groups;
} rvcx_hubert_cfg;

/* per-call conversion parameters -- the keyword set of VC.pipeline / rvc_infer
 * (rvc/infer/pipeline.py:289-311, rvc/infer/infer.py:109-128) plus Config's chunk geometry
 * (rvc/infer/infer.py:36-43) */
typedef struct {
  float pitch;            /* semitones */
  float f0_min, f0_max;
  float index_rate;
  float protect;
  float volume_envelope;
  int32_t sid;
  int32_t x_pad, x_query, x_center, x_max; /* seconds */
  uint64_t seed;          /* Philox seed for the two Gaussian draws when noise == NULL */
  int32_t f0_method;      /* RVCX_F0_RMVPE ("rmvpe" / "rmvpe+", pipeline.py:142-167) or RVCX_F0_FCPE ("fcpe", :169-181) */
  int32_t resample_sr;    /* VC.pipeline's resample_sr (pipeline.py:453-454): >= 16000 and != tgt_sr resamples the output
                             before the peak normalisation; 0 (what rvc_infer passes, infer.py:144) = off */
  int32_t hop_length;     /* VC.get_f0's hop_length: frame step of "mangio-crepe" in 16 kHz samples (pipeline.py:151-152);
                             <= 0: 128, the reference's default */
  int32_t reserved;
} rvcx_params;
enum { RVCX_F0_RMVPE = 0, RVCX_F0_FCPE = 1, RVCX_F0_CREPE = 2 /* "mangio-crepe", pipeline.py:86-117, 151-152 */ };

/* per-utterance extras of rvcx_convert_batch_ex */
typedef struct {
  /* VC.pipeline's f0_file (rvc/infer/pipeline.py:349-360): the parsed rows "time [s], f0 [Hz]" as float32 pairs in HOST
   * memory, or NULL.  VC.get_f0 turns them into a 100 Hz track with np.interp and overwrites the estimate from frame
   * x_pad * 100 on (pipeline.py:185-191); the library does exactly that (float32 / float64 steps as numpy takes them). */
  const float* inp_f0;
  int32_t inp_f0_rows;
  int32_t reserved;
  /* "mangio-crepe" only: the +-20 cent triangular dither torchcrepe adds to every decoded frame
   * (convert.bins_to_cents -> dither: scipy.stats.triang.rvs on the GLOBAL numpy RNG), one float32 per frame in HOST
   * memory (rvcx_crepe_frames(n_padded, hop) of them), or NULL: drawn from the call's Philox stream */
  const float* crepe_dither;
  int64_t crepe_dither_n;
} rvcx_utt_extra;

/* ---- lifecycle ------------------------------------------------------------------------ */
int rvcx_create(int device, rvcx_ctx** out);
void rvcx_destroy(rvcx_ctx* ctx);
const char* rvcx_last_error(rvcx_ctx* ctx); /* ctx may be NULL: last error of this thread */
const char* rvcx_version(void);

/* ---- model loading (host tensors in checkpoint layout; folded + packed + uploaded) ---- */
/* replaces load_hubert -- rvc/infer/infer.py:67-74 */
int rvcx_load_hubert(rvcx_ctx*, const rvcx_hubert_cfg*, const rvcx_tensor* tbl, int n);
/* replaces RMVPE0Predictor.__init__ -- rvc/lib/predictors/RMVPE.py:442-459 */
int rvcx_load_rmvpe(rvcx_ctx*, const rvcx_rmvpe_cfg*, const rvcx_tensor* tbl, int n);
/* replaces FCPEF0Predictor.__init__ / FCPEInfer.__init__ -- rvc/lib/predictors/FCPE.py:708-736, 806-826
 * (tbl: the checkpoint's "model" state_dict) */
int rvcx_load_fcpe(rvcx_ctx*, const rvcx_fcpe_cfg*, const rvcx_tensor* tbl, int n);
/* replaces torchcrepe.load.model (called by torchcrepe.predict, rvc/infer/pipeline.py:96): the state dict of
 * torchcrepe's model.Crepe -- conv{1..6}.weight (Cout, Cin, K, 1) / .bias, conv{1..6}_BN.*, classifier.*; the capacity
 * ("full" / "tiny" / ...) is read off the shapes.  torchcrepe is not vendored with the reference: parity unpinned */
int rvcx_load_crepe(rvcx_ctx*, const rvcx_tensor* tbl, int n);
/* replaces get_vc's Synthesizer construction -- rvc/infer/infer.py:78-105 */
int rvcx_load_synth(rvcx_ctx*, const rvcx_synth_cfg*, const rvcx_tensor* tbl, int n, int* model_id);
int rvcx_unload_synth(rvcx_ctx*, int model_id);
/* replaces faiss.read_index + reconstruct_n -- rvc/infer/pipeline.py:322-323.
 * big_npy is the (n, dim) float32 matrix of stored vectors; NULL/0 drops the index. */
int rvcx_load_index(rvcx_ctx*, const float* big_npy, int64_t n, int dim);
/* the same for a faiss "IVF{nlist},Flat" index, which is what RVC training writes: index.search then scans only
 * the inverted list of the query's nearest centroid (nprobe = 1, stored in the file; pipeline.py:242 uses it as
 * read).  centroids (nlist, dim): the coarse quantiser; assign (n): list id of every stored vector (row = id).
 * Lists with fewer than 8 vectors pad with id -1 / infinite distance, exactly as faiss does. */
int rvcx_load_index_ivf(rvcx_ctx*, const float* big_npy, int64_t n, int dim, const float* centroids, int nlist,
                        const int32_t* assign, int nprobe);

/* Folded weights live in per-model regions (freed by rvcx_unload_synth / rvcx_load_index(NULL) / a reload).
 * rvcx_weights_regions lists the device chunks of everything loaded, in a fixed order (HuBERT, RMVPE, voice
 * models by id, index): up to `cap` (pointer, used bytes) pairs are written, the total chunk count is returned.
 * Chunk sizes, offsets and *layout_hash depend on tensor SHAPES only, so ranks that loaded placeholder values
 * (zeros) with the same configurations report the same layout; rank 0's chunks can then be RCCL-broadcast into
 * them (polgen-rvc_amd/dist.py) instead of parsing / folding the checkpoints once per GPU.  New in rvcx -- the
 * reference is single-device.  After the chunks were overwritten, rvcx_weights_adopt re-reads
 * the value-dependent layer flags that travel in each region's header. */
int rvcx_weights_regions(rvcx_ctx*, int cap, void** dev_ptrs, int64_t* nbytes, uint64_t* layout_hash);
int rvcx_weights_adopt(rvcx_ctx*);
/* Same-device counterpart of the broadcast: copy the folded weights of `src` into this context, which must have
 * loaded the same configurations (placeholder values allowed) -- a second or third context per GPU (several
 * conversions in flight) then costs a device copy instead of parsing and folding again. */
int rvcx_weights_clone(rvcx_ctx*, rvcx_ctx* src);

/* ---- stage-level entry points (parity tests bind these) ------------------------------- */
/* RMVPE0Predictor.infer_from_audio_with_pitch -- rvc/lib/predictors/RMVPE.py:487-496.
 * audio (B, n) -> f0 (B, 1 + n/160) Hz; hidden (B, frames, 360) optional. */
int rvcx_rmvpe_f0(rvcx_ctx*, int B, const float* audio_hd, int64_t n, float thred, float f0_min,
                  float f0_max, float* f0_hd, float* hidden_hd);
int rvcx_rmvpe_frames(int64_t n);
/* MelSpectrogram.forward -- rvc/lib/predictors/RMVPE.py:412-439: audio (B, n) -> log-mel (B, 128, 1 + n/160) */
int rvcx_rmvpe_mel(rvcx_ctx*, int B, const float* audio_hd, int64_t n, float* mel_hd);
/* FCPEInfer.__call__(audio, sr=16000, threshold) -- rvc/lib/predictors/FCPE.py:739-745 (return_hz_f0, local_argmax
 * decoder): audio (B, n) -> f0 (B, n/160 + 1) Hz, 0 where the salience maximum is <= threshold.
 * salience (B, frames, 360) = the sigmoid output of FCPE.forward (FCPE.py:646), mel (B, 128, frames) =
 * Wav2Mel.__call__ transposed (FCPE.py:768-788); both optional. */
int rvcx_fcpe_f0(rvcx_ctx*, int B, const float* audio_hd, int64_t n, float threshold, float* f0_hd,
                 float* salience_hd, float* mel_hd);
int rvcx_fcpe_frames(int64_t n);
/* FCPEF0Predictor.compute_f0(x, p_len) as VC.get_f0 calls it (threshold 0.03, pipeline.py:169-179; FCPE.py:869-877)
 * followed by get_f0's own tail (pitch shift, coarse; pipeline.py:183-201): x (n samples) -> p_len frames. */
int rvcx_get_f0_fcpe_x(rvcx_ctx*, const float* x_hd, int64_t n, int64_t p_len, const rvcx_params* p, int32_t* coarse,
                       float* f0);
/* HubertModel.extract_features(source, padding_mask=False, output_layer=L)[0] --
 * call site rvc/infer/pipeline.py:228-236.  wav (B, n) -> feats (B, T', embed_dim). */
int rvcx_hubert_features(rvcx_ctx*, int B, const float* wav_hd, int64_t n, int output_layer,
                         float* feats_hd);
int rvcx_hubert_frames(rvcx_ctx*, int64_t n);
/* Synthesizer.infer -- rvc/lib/algorithm/synthesizers.py:163-188.
 * phone (B,T,input_dim), pitch (B,T) int32 coarse, pitchf (B,T) Hz, lens (B) valid frames,
 * sid (B).  z_noise (B,inter,T) / src_noise (B,T*upp) replace the two randn_like draws when
 * non-NULL (parity mode), otherwise Philox(seed).  out (B, T*upp). */
int rvcx_synth_infer(rvcx_ctx*, int model_id, int B, int T, const int32_t* lens,
                     const float* phone_hd, const int32_t* pitch_hd, const float* pitchf_hd,
                     const int32_t* sid, const float* z_noise_hd, const float* src_noise_hd,
                     uint64_t seed, float* out_hd);
/* the same with the NSF decoder evaluated on frames [dec_skip, len - dec_skip) of every item only (TextEncoder, flow and the
 * harmonic source stay whole); out samples outside that window read 0.  What VC.pipeline does with the result --
 * audio1[t_pad_tgt:-t_pad_tgt], rvc/infer/pipeline.py:432-447 -- makes the discarded ends dead work: with
 * dec_skip <= t_pad frames - rvcx_synth_dec_rf(model) the kept samples are those of the full evaluation (the decoder is
 * convolutional, nsf.py:100-144; rvcx_synth_dec_rf = its receptive field in frames, each side, + 2).  rvcx_convert_batch
 * uses this internally (RVCX_DEC_WINDOW=0 turns it off); no reference counterpart. */
int rvcx_synth_infer_window(rvcx_ctx*, int model_id, int B, int T, const int32_t* lens,
                            const float* phone_hd, const int32_t* pitch_hd, const float* pitchf_hd,
                            const int32_t* sid, const float* z_noise_hd, const float* src_noise_hd,
                            uint64_t seed, int dec_skip, float* out_hd);
int rvcx_synth_dec_rf(rvcx_ctx*, int model_id);
/* the same with the intermediates the reference returns beside the waveform (synthesizers.py:186-188):
 * stats (B, 2*inter, T) = [m_p ; logs_p] of the TextEncoder, zflow (B, inter, T) = z after the reverse flow */
int rvcx_synth_infer_taps(rvcx_ctx*, int model_id, int B, int T, const int32_t* lens,
                          const float* phone_hd, const int32_t* pitch_hd, const float* pitchf_hd,
                          const int32_t* sid, const float* z_noise_hd, const float* src_noise_hd,
                          uint64_t seed, float* out_hd, float* stats_hd, float* zflow_hd);
int rvcx_synth_upp(rvcx_ctx*, int model_id);
/* index.search(k=8) + weighted blend -- rvc/infer/pipeline.py:239-250.
 * feats (T, dim) in/out; ids (T,8) int64 and dist (T,8) optional. */
int rvcx_index_blend(rvcx_ctx*, float* feats_hd, int T, float index_rate, int64_t* ids_hd,
                     float* dist_hd);

/* ---- whole path ------------------------------------------------------------------------ */
/* capacity (in samples) the caller must provide per output buffer for an n-sample 16 kHz input;
 * the exact count VC.pipeline produces depends on the silence-aligned cut points (and is exactly
 * (n/160)*upp - 2*upp... for single-chunk clips); rvcx_convert_batch reports it in out_n */
int64_t rvcx_out_len(rvcx_ctx*, int model_id, int64_t n, const rvcx_params* p);
/* VC.pipeline for a batch of utterances -- rvc/infer/pipeline.py:289-467 with
 * f0_method = p->f0_method ("rmvpe+" or "fcpe"; that model must be loaded), pitch_guidance=1, resample_sr=0,
 * f0_file=None.
 * wav16k[i] (n[i] samples, 16 kHz mono f32, host or device); out[i] caller-allocated int16
 * buffers of rvcx_out_len samples (host or device); out_f32[i] optional (same capacity) float
 * waveform before int16 quantisation; out_n[i] receives the number of samples produced; noise[i] optional
 * packed parity noise (see rvcx_noise_len).
 * Utterances of one length class (rvcx_bucket_length: padded lengths within RVCX_BUCKET_FRAMES 10 ms frames, default
 * 128; clips long enough to be cut into chunks: equal lengths) are converted together as ragged micro-batches (B > 1
 * through HuBERT, RMVPE, TextEncoder and flow with per-item lengths; rvcx_micro_batch tells how many at a time;
 * rvcx_last_micro_batches what the last call did); every utterance's result is bit-identical to converting it alone.  Without parity noise utterance i draws its Gaussians from Philox(seed + i).
 * Batch conversion is listed as not done in the reference (TODO.md:11). */
int rvcx_convert_batch(rvcx_ctx*, int model_id, int B, const float* const* wav16k_hd,
                       const int64_t* n, const rvcx_params* p, const float* const* noise_hd,
                       int16_t* const* out_hd, float* const* out_f32_hd, int64_t* out_n);
/* the same with float64 input, the dtype rvc_infer hands to VC.pipeline (load_audio -> float64,
 * rvc/lib/my_utils.py:5-16): the float64 zero-phase high-pass (pipeline.py:329) then sees the reference's input */
int rvcx_convert_batch_f64(rvcx_ctx*, int model_id, int B, const double* const* wav16k_hd,
                           const int64_t* n, const rvcx_params* p, const float* const* noise_hd,
                           int16_t* const* out_hd, float* const* out_f32_hd, int64_t* out_n);
/* the same with per-utterance extras (`extra`: B entries or NULL); wav16k_hd[i] is float64 when wav_is_f64 != 0 */
int rvcx_convert_batch_ex(rvcx_ctx*, int model_id, int B, const void* const* wav16k_hd, int wav_is_f64,
                          const int64_t* n, const rvcx_params* p, const float* const* noise_hd,
                          const rvcx_utt_extra* extra, int16_t* const* out_hd, float* const* out_f32_hd, int64_t* out_n);
/* utterances of n samples converted per launch sequence (memory-bounded; RVCX_MAX_BATCH, RVCX_ARENA_GB) */
int rvcx_micro_batch(rvcx_ctx*, int model_id, int64_t n, const rvcx_params* p);
/* The sample count whose launch geometry an n-sample utterance is converted with (>= n): utterances with equal values
 * share micro-batches.  An uncut rmvpe / mangio-crepe clip: the longest clip of its length class; otherwise n itself. */
int64_t rvcx_bucket_length(rvcx_ctx*, int model_id, int64_t n, const rvcx_params* p);
/* Member counts of the micro-batches the last rvcx_convert_batch* call of this context formed (in launch order);
 * returns their number (counts receives at most cap of them). */
int rvcx_last_micro_batches(rvcx_ctx*, int32_t* counts, int cap);
/* floats of parity noise rvcx_convert_batch consumes for one n-sample utterance: for each
 * chunk in order, z_noise (inter*T) then src_noise (T*upp) -- the draw order of the reference */
int64_t rvcx_noise_len(rvcx_ctx*, int model_id, int64_t n, const rvcx_params* p);
/* VC.get_f0 -- rvc/infer/pipeline.py:132-201 on the reflect-padded, high-passed signal (the F0 model is chosen by
 * p->f0_method): returns coarse (int32) and f0 (Hz) of p_len frames for one utterance */
int rvcx_get_f0(rvcx_ctx*, const float* wav16k_hd, int64_t n, const rvcx_params* p,
                int32_t* coarse, float* f0, int64_t* p_len);
/* VC.get_f0(input_audio_path, x, p_len, pitch, "rmvpe+", ...) -- rvc/infer/pipeline.py:132-201 with the
 * reference's meaning of x: the ALREADY reflect-padded, high-passed signal (n samples).  Writes 1 + n/160 frames
 * of coarse (1..255) and f0 (Hz, shifted by p->pitch semitones), un-truncated like the reference's return. */
int rvcx_get_f0_x(rvcx_ctx*, const float* x_hd, int64_t n, const rvcx_params* p, int32_t* coarse, float* f0);
/* VC.get_f0 with everything the reference's does (pipeline.py:132-201): the F0 model p->f0_method names on the padded
 * signal x, pitch shift, the optional f0-file table inp_f0 (rows of (time, f0) float32 pairs in host memory, see
 * rvcx_utt_extra), coarse quantisation.  rmvpe+ returns 1 + n/160 frames, fcpe p_len frames (*frames). */
int rvcx_get_f0_x_ex(rvcx_ctx*, const float* x_hd, int64_t n, int64_t p_len, const rvcx_params* p, const float* inp_f0,
                     int inp_f0_rows, int32_t* coarse, float* f0, int64_t* frames);
/* frames torchcrepe.predict(..., pad=True) returns for n samples at frame step hop: 1 + n / hop */
int64_t rvcx_crepe_frames(int64_t n, int hop);
/* VC.get_f0_crepe up to the resize (rvc/infer/pipeline.py:90-106): x / quantile(|x|, 0.999), then
 * torchcrepe.predict(x, 16000, hop, fmin, fmax, model, batch_size = 2 * hop, pad = True) with its default Viterbi decoder.
 * dither: rvcx_crepe_frames(n, hop) floats (see rvcx_utt_extra) or NULL (Philox, `seed`).  Writes the pitch track (Hz) and,
 * when non-NULL, the network's sigmoid outputs (360, frames) and the decoded bins. */
int rvcx_crepe_predict(rvcx_ctx*, const float* x_hd, int64_t n, int hop, float fmin, float fmax, const float* dither_hd,
                       uint64_t seed, float* pitch_hd, float* probs_hd, int32_t* bins_hd);
/* op level: core.postprocess + decode.viterbi + convert.bins_to_frequency on given sigmoid outputs (360, F), one Viterbi
 * pass per `batch` frames */
int rvcx_op_crepe_decode(rvcx_ctx*, const float* probs_hd, int64_t F, int batch, float fmin, float fmax,
                         const float* dither_hd, float* pitch_hd, int32_t* bins_hd);
/* VC.get_f0(..., f0_method="mangio-crepe", hop_length = p->hop_length) on the padded signal x: get_f0_crepe incl. the
 * resize to p_len frames, pitch shift, f0-file table, coarse quantisation (pipeline.py:86-117, 151-152, 183-201) */
int rvcx_get_f0_crepe_x(rvcx_ctx*, const float* x_hd, int64_t n, int64_t p_len, const rvcx_params* p, const float* inp_f0,
                        int inp_f0_rows, const float* dither_hd, int64_t dither_n, int32_t* coarse, float* f0);
/* host only (no GPU needed): the 100 Hz track VC.get_f0 builds from an f0 file's rows (pipeline.py:186-189: delta_t in
 * float32, np.interp in float64).  Writes min(count, cap) values, returns count. */
int rvcx_f0_file_track(const float* inp_f0, int rows, double* track, int cap);
/* librosa.resample(librosa.to_mono(audio.T), orig_sr=sr_in, target_sr=sr_out) of load_audio (rvc/lib/my_utils.py:9-13):
 * x = (frames, channels) interleaved float64 (host or device), y = rvcx_resample_len(frames, ...) mono float64 samples.
 * Band-limited sinc interpolation (csrc/audio.hip).  librosa's default res_type is "soxr_hq": libsoxr is not vendored and
 * its coefficients are unpublished, its design targets are (pass-band flat to 0.9136 x Nyquist, -120.4 dB from 1.0 x
 * Nyquist, linear phase).  kind 0 (the default; -1 = default / RVCX_RESAMPLER): "kaiser_hq", a Kaiser-windowed sinc
 * designed to those targets (measured +-0.001 dB / -127 dB); kind 1: resampy's published "kaiser_best" (librosa's default
 * before 0.10) in its published arithmetic. */
int64_t rvcx_resample_len(int64_t n, int sr_in, int sr_out);
int rvcx_resample_f64(rvcx_ctx*, const double* x_hd, int64_t frames, int channels, int sr_in, int sr_out, double* y_hd);
int rvcx_resample_f64_kind(rvcx_ctx*, const double* x_hd, int64_t frames, int channels, int sr_in, int sr_out, int kind,
                           double* y_hd);
/* VC.vc(model, net_g, sid, audio0, pitch, pitchf, index, big_npy, index_rate, version="v2", protect) --
 * rvc/infer/pipeline.py:203-287: HuBERT -> (retrieval blend with the resident index when index_rate != 0) ->
 * x2 upsample / protect mix -> Synthesizer.infer.  audio0 (n samples of audio_pad); pitch / pitchf (n_pitch
 * frames, n_pitch >= rvcx_vc_frames(n)); out receives rvcx_vc_frames(n) * upp float32 samples (*out_n), the
 * un-trimmed audio1 of the reference.  z_noise / src_noise as in rvcx_synth_infer. */
int rvcx_vc(rvcx_ctx*, int model_id, const float* audio0_hd, int64_t n, const int32_t* pitch_hd,
            const float* pitchf_hd, int n_pitch, int sid, float index_rate, float protect,
            const float* z_noise_hd, const float* src_noise_hd, uint64_t seed, float* out_hd, int64_t* out_n);
int rvcx_vc_frames(rvcx_ctx*, int64_t n);

/* ---- instrumentation ------------------------------------------------------------------- */
/* per-stage GPU milliseconds (HIP events on the library's stream) of the last
 * rvcx_convert_batch: {highpass, rmvpe, hubert, index, enc_p, flow, decoder, post, total} */
int rvcx_last_timing(rvcx_ctx*, float* ms9);
/* HIP-event profile of the MFMA conv kernel family: begin=1 starts recording an event pair around
 * every conv launch on the library stream; begin=0 stops and returns, per tile configuration
 * (kind: -1 generic strided kernel, halo*10 for the stride-1
 * family, 100000/100001 its Linear variants), the launch count, algorithmic FLOPs (2*M*N*K of the unpadded problem) and the
 * summed kernel milliseconds, plus the tile shape (bm x bn).
 * The profile is PROCESS-wide: the two hooks serialise against each other, but launches of any other context of the
 * process that run while a profile is open are recorded into it -- profile with one context active. */
int rvcx_conv_profile(rvcx_ctx*, int begin, int64_t* launches, double* flops, double* ms, int32_t* bm,
                      int32_t* bn, int32_t* kind, int cap);
/* per-launch table (CSV text: tile,B,cin,cout,k,stride,nout,gflop,ms,tflops) of the last profile */
const char* rvcx_conv_profile_csv(rvcx_ctx*);
/* name and memory size of GPU `device` (no context needed) -- what Config._configure_gpu reads through
 * torch.cuda.get_device_name / get_device_properties, rvc/infer/infer.py:49-63.  -1 without a visible device. */
int rvcx_device_info(int device, char* name, int name_cap, int64_t* total_bytes);
/* free / total bytes of the context's GPU (hipMemGetInfo): what is left for further voice models and indices */
int rvcx_mem_info(rvcx_ctx*, int64_t* free_bytes, int64_t* total_bytes);
/* calls this context repeated because a split-fp16 kernel met an activation beyond
 * fp16 range (|x| >= 6e4; attention K / V >= 234): the default kernels form fp32-grade products from fp16 hi/lo
 * halves, which have fp16's exponent range.  The repeat is automatic and transparent; this counter reports it. */
int64_t rvcx_fp32_reruns(rvcx_ctx*);
/* layers (and attention calls) pinned to the exact-fp32 kernels since their models were loaded: a split-fp16 kernel that
 * meets an activation beyond fp16 range stamps its layer; the entry point pins the first offender of the call (launch
 * order) for the life of the model and repeats the call once -- later requests pay nothing. */
int64_t rvcx_fp32_layers(rvcx_ctx*);
/* which ones: one text line per layer the range guard pinned at run time since its model was loaded ("voice model 0: layer
 * 17 of 142 (conv 128 <- 128 x 7)"), in the order they were pinned; returns the length of the full text (buf receives at
 * most cap - 1 bytes + NUL).  Diagnostics for checkpoints with outlier channels; no reference counterpart. */
int rvcx_fp32_pinned(rvcx_ctx*, char* buf, int cap);
/* calls repeated with the single-workgroup BiGRU kernel because the cluster kernel's workgroups were not co-resident */
int64_t rvcx_gru_fallbacks(rvcx_ctx*);
/* the cluster BiGRU publishes h_t inside one XCD with a plain store and relies on the partners' sc1 polls seeing it (a
 * hardware observation, INTEGRATION.md "hardware assumptions").  The library checks that once per device when RMVPE is loaded
 * (or at the first BiGRU call); this runs the check if it has not run and reports it: 1 holds (plain publish in use), 0 does
 * not (the device uses the write-through publish), -1 undecided (no co-located pair ran side by side), -2 error. */
int rvcx_gru_publish_probe(rvcx_ctx*);
/* retrieval: queries whose 8 neighbours could not be certified from the split-fp16 pre-filter and were searched
 * exhaustively instead (csrc/index.hip) since the last call of this function; waits for the device.  -1: no index */
int64_t rvcx_index_exhaustive(rvcx_ctx*);
/* DEBUG HOOKS -- rvcx_debug_inject, rvcx_bench_resblock_pair, rvcx_bench_conv1d, rvcx_bench_gemm, rvcx_conv_override are
 * process-wide tuning / fault-injection levers.  They return -2 ("refused") unless the process was started with
 * RVCX_DEBUG=1 in its environment (read once); the product path never calls them. */
/* test hook: what = 1 makes the next call behave as if the BiGRU cluster kernel had timed out; what = 3 makes the next BiGRU
 * cluster launch of the calling thread lose one workgroup, so that its partners really run into the device-side time-out
 * (~1.5 s) and the call is repeated on the single-workgroup kernel; what = 2 reads and clears the raw device error word */
int rvcx_debug_inject(rvcx_ctx*, int what);
/* algorithmic FLOPs issued by conv/GEMM/attention launches since the last reset */
double rvcx_flop_counter(rvcx_ctx*, int reset);
void* rvcx_stream(rvcx_ctx*); /* hipStream_t the library launches on */

/* ---- kernel-level entry points (unit parity tests of the HIP kernels) ------------------ */
/* FCPEF0Predictor.post_process()[0] (FCPE.py:841-867) + the tail of VC.get_f0 (pipeline.py:183-201) on a given raw
 * track: raw (F_in) Hz with 0 = unvoiced -> f0 (p_len, float32 of the float64 result) and coarse (p_len) */
int rvcx_op_fcpe_post(rvcx_ctx*, const float* raw_hd, int F_in, int p_len, double pitch, double f0_min, double f0_max,
                      int32_t* coarse, float* f0);
/* y = act(conv1d(pre(x), w) + bias) + res ; x (B,Cin,Tin) w (Cout,Cin/groups,K) host f32 */
int rvcx_op_conv1d(rvcx_ctx*, const float* x, const float* w, const float* bias, const float* res,
                   float* y, int B, int Cin, int Tin, int Cout, int K, int stride, int dil,
                   int pad_left, int Tout, int groups, int pre_lrelu, float pre_slope, int act,
                   float act_slope, const int32_t* lens_in, const int32_t* lens_out);
/* one ResBlock1 step, y = x + c2(lrelu(c1(lrelu(x)) + b1)) + b2 -- rvc/lib/algorithm/residuals.py:45-53.
 * w1 / w2 (C, C, K); c1 has dilation dil, c2 dilation 1.  fused = 1: the single-kernel form (resblock.hip);
 * fused = 0: the two conv launches it replaces.  lens (B) optional per-item valid lengths. */
int rvcx_op_resblock_pair(rvcx_ctx*, const float* x, const float* w1, const float* b1, const float* w2,
                          const float* b2, float* y, int B, int C, int T, int K, int dil, float slope, int fused,
                          const int32_t* lens);
/* a whole ResBlock1 with kernel size 3 -- three steps x = x + c2_s(lrelu(c1_s(lrelu(x)) + b1_s)) + b2_s, c1_s dilated by
 * dils[s] (1, 3, 5 in every RVC v2 decoder), rvc/lib/algorithm/residuals.py:15-62 -- in ONE kernel (csrc/resblock3.hip; C = 32 /
 * 64, T a multiple of 4).  w1 / w2 (3, C, C, 3), b1 / b2 (3, C) or NULL.  Bit-identical to three rvcx_op_resblock_pair calls. */
int rvcx_op_resblock3(rvcx_ctx*, const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                      float* y, int B, int C, int T, const int32_t* dils, float slope, const int32_t* lens);
/* micro-benchmark of one ResBlock1 step on device-resident random data (fused kernel or the two launches) */
int rvcx_bench_resblock_pair(rvcx_ctx*, int B, int C, int T, int K, int dil, int fused, int iters,
                             float* ms_per_launch);
/* micro-benchmark of the conv kernel on device-resident random data: `iters` back-to-back launches of
 * y = conv1d(lrelu(x)) + bias + res, average milliseconds per launch (HIP events on the library stream) */
int rvcx_bench_conv1d(rvcx_ctx*, int B, int Cin, int Tin, int Cout, int K, int stride, int dil, int groups,
                      int iters, float* ms_per_launch);
/* tuning hook for the tile-selection sweeps (tools/sweep_tiles_1d.py, sweep_unet.py, bench_conv.py): force the conv_fast tile index, the
 * staging variant (ignored: the LDS-DMA variant was removed in round 2) and the split-K factor; -1 = heuristic.
 * Process-wide; never set by the product path. */
int rvcx_conv_override(int tile, int variant, int splitk);
/* ConvTranspose1d: w (Cin,Cout,K), padding p; Tout = (Tin-1)*s - 2p + K */
int rvcx_op_convtranspose1d(rvcx_ctx*, const float* x, const float* w, const float* bias, float* y,
                            int B, int Cin, int Tin, int Cout, int K, int stride, int pad,
                            int pre_lrelu, float pre_slope);
/* Conv2d 3x3 pad 1 (+bias, act, res) on (B,Cin,H,W) */
int rvcx_op_conv2d3x3(rvcx_ctx*, const float* x, const float* w, const float* bias, const float* res,
                      float* y, int B, int Cin, int H, int W, int Cout, int act);
/* One ConvBlockRes of the F0 model's U-Net (RMVPE.py:140-175, BatchNorm already folded into w / b by the caller):
 * y = relu(conv3x3(relu(conv3x3(x, w1) + b1), w2) + b2) + (wsc ? conv1x1(x, wsc) + bsc : x), through the model's own block
 * path (split hand-off between the two convs on large maps).  rows (B ints or NULL): valid rows of each item -- the rows
 * below are zero in x and come back zero (what a shorter member of a ragged batch sees). */
int rvcx_op_convblock2d(rvcx_ctx*, const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                        const float* wsc, const float* bsc, float* y, int B, int Cin, int Cout, int H, int W,
                        const int32_t* rows);
/* ConvTranspose2d 3x3 stride 2 pad 1 output_padding 1: (B,Cin,H,W) -> (B,Cout,2H,2W), w (Cin,Cout,3,3) */
int rvcx_op_convtranspose2d(rvcx_ctx*, const float* x, const float* w, const float* bias, float* y,
                            int B, int Cin, int H, int W, int Cout, int act);
/* softmax(q k^T [+ rel-pos bias]) v on (B, H*D, T) channel-first tensors; emb_rel_k/v (2w+1, D) or NULL */
int rvcx_op_attention(rvcx_ctx*, const float* q, const float* k, const float* v, float* out, int B,
                      int H, int D, int T, float scale, const float* emb_rel_k,
                      const float* emb_rel_v, int window, const int32_t* lens);
/* The time-major Linear kernel of the transformer sections (csrc/gemm.hip): x_cf (B, Cin, T) is turned into rows
 * r = b T + t (split form, or fp32 when exact_fp32), y = act(x W^T + bias) + res.  w (Cout, Cin); res_tm (B T, Cout) or
 * NULL; act as in rvcx_op_conv1d (0 none, 2 relu, 3 gelu).  Outputs, each (B T, Cout) unless noted: y_tm; y_cf
 * (B, Cout, T) or NULL; y_split = the split-form output decoded to fp32 (hi + lo) or NULL. */
int rvcx_op_gemm_tm(rvcx_ctx*, const float* x_cf, const float* w, const float* bias, const float* res_tm, int B, int T,
                    int Cin, int Cout, int act, int exact_fp32, float* y_tm, float* y_cf, float* y_split);
/* micro-benchmark of the time-major Linear kernel on device-resident random rows (split form in, fp32 rows out) */
int rvcx_bench_gemm(rvcx_ctx*, int64_t rows, int Cin, int Cout, int iters, float* ms_per_launch);
/* LayerNorm of time-major rows (rows, C): fp32 result and the decoded split-form result (NULL to skip) */
int rvcx_op_layernorm_tm(rvcx_ctx*, const float* x, const float* gamma, const float* beta, float* y, float* y_split,
                         int64_t rows, int C, float eps);
/* LayerNorm over channels of (B,C,T) */
int rvcx_op_layernorm_c(rvcx_ctx*, const float* x, const float* gamma, const float* beta, float* y,
                        int B, int C, int T, float eps);
/* bidirectional GRU: x (B,T,I) -> y (B,T,2H); weights in torch layout */
int rvcx_op_bigru(rvcx_ctx*, const float* x, const float* w_ih, const float* w_hh, const float* b_ih,
                  const float* b_hh, const float* w_ih_r, const float* w_hh_r, const float* b_ih_r,
                  const float* b_hh_r, float* y, int B, int T, int I, int H);
/* scipy.signal.filtfilt(bh, ah, x) of pipeline.py:19-22,329 (float64) */
int rvcx_op_highpass(rvcx_ctx*, const double* x, double* y, int64_t n);

/* ---- FLAC on the host (csrc/flac.hip; no GPU, no context) -------------------------------------------------------
 * rvc/infer/infer.py:153 writes WAV bytes whatever the extension of output_path; the mirror writes a real FLAC stream when
 * the path ends in ".flac"  and reads FLAC input where soundfile (rvc/lib/my_utils.py:9) is absent.
 * Encoder: 16-bit PCM, interleaved, 1-8 channels, block size 4096, CONSTANT / VERBATIM / FIXED 0-4 + partitioned Rice,
 * STREAMINFO with MD5.  Decoder: the whole format (LPC, wasted bits, Rice2 / escape partitions, stereo decorrelation,
 * 4-32 bits), CRC-8 / CRC-16 / MD5 checked.  Errors: negative return, text via rvcx_flac_last_error (per thread). */
int64_t rvcx_flac_encode_bound(int64_t frames, int channels);
int64_t rvcx_flac_encode_s16(const int16_t* pcm, int64_t frames, int channels, int sample_rate, uint8_t* out, int64_t cap);
int rvcx_flac_info(const uint8_t* data, int64_t n, int64_t* frames, int32_t* channels, int32_t* sample_rate, int32_t* bits);
/* decoded samples, interleaved, right-justified at the stream's sample size; returns frames decoded */
int64_t rvcx_flac_decode_s32(const uint8_t* data, int64_t n, int32_t* out, int64_t cap_samples);
const char* rvcx_flac_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RVCX_H */
