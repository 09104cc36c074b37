// VC.pipeline orchestration (rvc/infer/pipeline.py:289-467).
#pragma once
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

size_t highpass_ext_doubles(long n);   // scratch the caller provides as `ext`
void launch_highpass(const float* x32, const double* x64, double* ext, double* y64, float* y32, long n,
                     hipStream_t s);

// One utterance, everything on the device.  wav: device f32 (n).  noise: device packed parity noise or
// null (Philox).  out_pcm: device int16 (capacity out_capacity), out_f32 optional.  Returns samples.
long convert_one(Ctx& c, int model_id, const float* wav, long n, const rvcx_params& p, const float* noise,
                 short* out_pcm, float* out_f32, float* stage_ms /*9 or null*/);
size_t convert_arena_bytes(Ctx& c, int model_id, long n, const rvcx_params& p);

// VC.get_f0 on device for one utterance: coarse/f0 device arrays of p_len frames
long get_f0_device(Ctx& c, const float* apad, long n_pad, const rvcx_params& p, int* coarse, float* f0,
                   hipStream_t s);

}  // namespace rvcx
