// VC.pipeline orchestration (rvc/infer/pipeline.py:289-467).
#pragma once
#include <functional>
#include "models.h"

namespace rvcx {

struct Geometry {
  long t_pad = 0, t_pad_tgt = 0, t_query = 0, t_center = 0, t_max = 0;
};
struct Chunk {
  long s, e;      // sample range of audio_pad handed to vc()
  long f0_off;    // first pitch frame
};

Geometry make_geometry(const rvcx_params& p, int tgt_sr);
std::vector<Chunk> plan_chunks(long n, const std::vector<long>& opt_ts, const Geometry& g);
long out_capacity(const SynthModel& m, long n, const rvcx_params& p);
long noise_len_for(const Ctx& c, const SynthModel& m, long n, const rvcx_params& p);

size_t highpass_ext_doubles(long n);   // scratch per signal the caller provides as `ext`
// B signals (element stride xs, 0 = n); y64 / y32 are written densely (B, n).  ns (device, B ints or null): item b holds
// ns[b] <= n samples -- it is filtered as a signal of exactly that length, the rest of its output row is zero
void launch_highpass(const float* x32, const double* x64, double* ext, double* y64, float* y32, long n,
                     hipStream_t s, int B = 1, long xs = 0, const int* ns = nullptr);

// one utterance of a rvcx_convert_batch call.  wav / wav64 / noise / out / out_f32 may be host or device memory.
struct UttIO {
  const float* wav = nullptr;      // 16 kHz mono float32 ...
  const double* wav64 = nullptr;   // ... or float64 (what load_audio returns in the reference, my_utils.py:16)
  long n = 0;
  const float* noise = nullptr;    // packed parity noise (rvcx_noise_len floats) or null (Philox)
  short* out = nullptr;            // capacity out_capacity()
  float* out_f32 = nullptr;        // optional, same capacity
  long out_n = 0;                  // produced samples (result)
  int seed_offset = 0;             // Philox stream of this utterance = params.seed + seed_offset
  const float* inp_f0 = nullptr;   // f0 file table, rows of (time [s], f0 [Hz]) float32 in HOST memory (pipeline.py:349-360)
  int inp_f0_rows = 0;
  const float* crepe_dither = nullptr;   // "mangio-crepe": per-frame dither in HOST memory or null (rvcx_utt_extra)
  long crepe_dither_n = 0;
};
// per batch item of get_f0_device: the crepe dither (host) and the item's Philox stream offset
struct F0Extra {
  const float* dither = nullptr;
  long dither_n = 0;
  int seed_offset = 0;
};
// VC.pipeline for a list of utterances: equal-length utterances run as micro-batches (B > 1 through every network).
void convert_batch(Ctx& c, int model_id, std::vector<UttIO>& utts, const rvcx_params& p, float* stage_ms /*9 or null*/);
int convert_micro_batch(Ctx& c, int model_id, long n, const rvcx_params& p);   // utterances per micro-batch at this length

// F0 back-end selected by params.f0_method: throws unless its model is resident; workspace for B signals of n_pad samples
void check_f0_backend(const Ctx& c, const rvcx_params& p);
size_t f0_arena_bytes(const Ctx& c, const rvcx_params& p, int B, long n_pad);
int crepe_hop(const rvcx_params& p);
void crepe_f0_device(Ctx& c, const float* x, long n, const rvcx_params& p, long p_len, const F0Extra* ex, float* f0raw,
                     hipStream_t s);   // VC.get_f0_crepe for one padded signal on the device          // "mangio-crepe" frame step: params.hop_length, 128 when unset
// VC.get_f0 on device for B equal-length reflect-padded signals: coarse/f0 rows of out_stride elements
// ns_host (optional, B ints, rmvpe only): ragged batch -- item b holds ns_host[b] <= n_pad padded samples in its row
long get_f0_device(Ctx& c, const float* apad, long n_pad, const rvcx_params& p, int* coarse, float* f0,
                   hipStream_t s, int B = 1, long out_stride = 0, const std::function<void()>* mid = nullptr,
                   const F0Extra* extra = nullptr, const int* ns_host = nullptr);
int bucket_frames();                                                   // class width of the ragged micro-batches (frames)
long bucket_length(long n, const rvcx_params& p, const Geometry& g);   // length whose geometry an n-sample utterance runs with

}  // namespace rvcx
