// VC.pipeline (rvc/infer/pipeline.py:289-467) on the device: zero-phase high-pass, silence-aligned
// chunking, F0 once per utterance, per-chunk HuBERT -> (index blend) -> x2 upsample / protect ->
// Synthesizer.infer, trim, RMS envelope, peak normalise, int16.
#include <cmath>

#include "models.h"
#include "ops.h"
#include "pipeline.h"

namespace rvcx {

// scipy.signal.butter(N=5, Wn=48, btype="high", fs=16000) (pipeline.py:19-22), scipy 1.15.3
static const double BH[6] = {0.9699606451838447,  -4.849803225919223, 9.699606451838447,
                             -9.699606451838447, 4.849803225919223,  -0.9699606451838447};
static const double AH[6] = {1.0, -4.939001819168364, 9.757863526739543, -9.639544849413458,
                             4.761506797356209, -0.9408236532054606};
constexpr int PADLEN = 18;  // filtfilt default 3*max(len(a),len(b))

// scipy.signal.lfilter_zi: solve (I - companion(a)^T) zi = b[1:] - a[1:]*b[0]
static void lfilter_zi(double zi[5]) {
  double M[5][6];
  for (int i = 0; i < 5; ++i)
    for (int j = 0; j < 5; ++j) {
      // companion(a): first row -a[1:]/a[0], sub-diagonal ones.  A^T[i][j] = A[j][i]
      double Aji = (j == 0) ? -AH[i + 1] : ((j == i + 1) ? 1.0 : 0.0);
      M[i][j] = (i == j ? 1.0 : 0.0) - Aji;
    }
  for (int i = 0; i < 5; ++i) M[i][5] = BH[i + 1] - AH[i + 1] * BH[0];
  for (int col = 0; col < 5; ++col) {
    int piv = col;
    for (int r = col + 1; r < 5; ++r)
      if (std::fabs(M[r][col]) > std::fabs(M[piv][col])) piv = r;
    for (int k = 0; k < 6; ++k) std::swap(M[col][k], M[piv][k]);
    for (int r = 0; r < 5; ++r) {
      if (r == col) continue;
      const double f = M[r][col] / M[col][col];
      for (int k = col; k < 6; ++k) M[r][k] -= f * M[col][k];
    }
  }
  for (int i = 0; i < 5; ++i) zi[i] = M[i][5] / M[i][i];
}

struct IirCoef {
  double b[6], a[6], zi[5];
};

// The 5th-order Butterworth at 48 Hz / 16 kHz forgets its state (measured slice error against the full recursion: 2.7e-8 after 3072 samples of warm-up).  Each lane
// filters one IIR_CHUNK-sample slice after warming up on the IIR_WARM samples before it (the first slice
// starts from scipy's exact initial state zi*x[0]); slices are independent, so the serial recursion
// becomes ~1000 concurrent lanes.  Measured against scipy.signal.filtfilt: max |err| 6e-8 on a 30 s
// clip, the same size as the float64 rounding-order noise this ill-conditioned filter shows between any
// two evaluation orders (tests/test_gpu_pipeline.py::test_highpass_matches_scipy), and below float32 eps.
constexpr int IIR_CHUNK = 256;
constexpr int IIR_WARM = 3072;   // slice error vs the full recursion: 2.7e-8 at 3072 (= the 4096 floor), 7e-8 at 2560, 1.3e-6 at 2048

// scipy filtfilt(method="pad", padtype="odd"): odd extension by PADLEN samples on both sides
__global__ void odd_ext_kernel(const float* __restrict__ x32, const double* __restrict__ x64, double* ext, long n) {
  const long m = n + 2 * PADLEN;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < m; i += (long)gridDim.x * 256) {
    auto X = [&](long j) -> double { return x64 ? x64[j] : (double)x32[j]; };
    double v;
    if (i < PADLEN) v = 2.0 * X(0) - X(PADLEN - i);
    else if (i >= PADLEN + n) v = 2.0 * X(n - 1) - X(n - 2 - (i - PADLEN - n));
    else v = X(i - PADLEN);
    ext[i] = v;
  }
}

// direct-form-II-transposed step (scipy.signal.lfilter's recursion), float64
__device__ __forceinline__ double iir_step(const IirCoef& cf, double (&z)[5], double x) {
  const double y = cf.b[0] * x + z[0];
  z[0] = cf.b[1] * x + z[1] - cf.a[1] * y;
  z[1] = cf.b[2] * x + z[2] - cf.a[2] * y;
  z[2] = cf.b[3] * x + z[3] - cf.a[3] * y;
  z[3] = cf.b[4] * x + z[4] - cf.a[4] * y;
  z[4] = cf.b[5] * x - cf.a[5] * y;
  return y;
}

// one lane per slice; `rev` walks the buffers backwards (filtfilt's second, time-reversed pass)
__global__ void iir_slice_kernel(const double* __restrict__ in, double* __restrict__ out, long m, int rev,
                                 IirCoef cf) {
  const long p = blockIdx.x * (long)blockDim.x + threadIdx.x;
  const long s = p * IIR_CHUNK;
  if (s >= m) return;
  const long e = min(m, s + IIR_CHUNK);
  const long w0 = max(0L, s - IIR_WARM);
  double z[5];
  {
    const double x0 = in[rev ? m - 1 - w0 : w0];
    for (int k = 0; k < 5; ++k) z[k] = cf.zi[k] * x0;      // lfilter(..., zi = zi * x[0])
  }
  long i = w0;
  for (; i + 8 <= s; i += 8) {                              // warm-up: loads hoisted 8 at a time
    double xv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) xv[k] = in[rev ? m - 1 - (i + k) : i + k];
#pragma unroll
    for (int k = 0; k < 8; ++k) (void)iir_step(cf, z, xv[k]);
  }
  for (; i < s; ++i) (void)iir_step(cf, z, in[rev ? m - 1 - i : i]);
  for (; i + 8 <= e; i += 8) {
    double xv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) xv[k] = in[rev ? m - 1 - (i + k) : i + k];
#pragma unroll
    for (int k = 0; k < 8; ++k) out[rev ? m - 1 - (i + k) : i + k] = iir_step(cf, z, xv[k]);
  }
  for (; i < e; ++i) out[rev ? m - 1 - i : i] = iir_step(cf, z, in[rev ? m - 1 - i : i]);
}

__global__ void crop_ext_kernel(const double* ext, double* y64, float* y32, long n) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const double v = ext[PADLEN + i];
    if (y64) y64[i] = v;
    if (y32) y32[i] = (float)v;
  }
}

// ext: highpass_ext_doubles(n) doubles of scratch
void launch_highpass(const float* x32, const double* x64, double* ext, double* y64, float* y32, long n,
                     hipStream_t s) {
  RVCX_CHECK(n > PADLEN, "highpass: input shorter than filtfilt's pad length");
  static IirCoef cf;
  static bool ready = false;
  if (!ready) {
    for (int i = 0; i < 6; ++i) {
      cf.b[i] = BH[i];
      cf.a[i] = AH[i];
    }
    lfilter_zi(cf.zi);
    ready = true;
  }
  const long m = n + 2 * PADLEN;
  double* tmp = ext + m;
  const unsigned g = (unsigned)std::min<long>(cdiv64(m, 256), 65535);
  const int nslices = (int)cdiv64(m, IIR_CHUNK);
  hipLaunchKernelGGL(odd_ext_kernel, dim3(g), dim3(256), 0, s, x32, x64, ext, n);
  hipLaunchKernelGGL(iir_slice_kernel, dim3(cdiv(nslices, 64)), dim3(64), 0, s, ext, tmp, m, 0, cf);
  hipLaunchKernelGGL(iir_slice_kernel, dim3(cdiv(nslices, 64)), dim3(64), 0, s, tmp, ext, m, 1, cf);
  hipLaunchKernelGGL(crop_ext_kernel, dim3(g), dim3(256), 0, s, ext, y64, y32, n);
}

size_t highpass_ext_doubles(long n) { return 2 * ((size_t)n + 2 * PADLEN) + 16; }

// ---------------------------------------------------------------- chunk search (pipeline.py:330-344)
// audio_sum[i] = sum_{j<160} reflect_pad(audio,80)[i+j]  (same left-to-right order as the numpy loop)
__global__ void window_abs_sum_kernel(const double* audio, double* asum, long n) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    double acc = 0.0;
    for (int j = 0; j < 160; ++j) {
      long p = i + j - 80;
      if (p < 0) p = -p;
      if (p >= n) p = 2 * (n - 1) - p;
      acc += audio[p];
    }
    asum[i] = fabs(acc);
  }
}

__global__ void argmin_first_kernel(const double* v, long lo, long hi, long* out) {
  __shared__ double bv[256];
  __shared__ long bi[256];
  double best = INFINITY;
  long idx = hi;
  for (long i = lo + threadIdx.x; i < hi; i += 256)
    if (v[i] < best) {
      best = v[i];
      idx = i;
    }
  bv[threadIdx.x] = best;
  bi[threadIdx.x] = idx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      const double ov = bv[threadIdx.x + o];
      const long oi = bi[threadIdx.x + o];
      if (ov < bv[threadIdx.x] || (ov == bv[threadIdx.x] && oi < bi[threadIdx.x])) {
        bv[threadIdx.x] = ov;
        bi[threadIdx.x] = oi;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = bi[0];
}

// ---------------------------------------------------------------- post-processing
// librosa.feature.rms(frame_length=sr, hop=sr/2, center=True zero pad) -> n_frames values (f32)
template <typename T>
__global__ void frame_rms_kernel(const T* x, float* out, long n, int frame, int hop, int nframes) {
  __shared__ double red[256];
  const int f = blockIdx.x;
  const long start = (long)f * hop - frame / 2;
  double acc = 0.0;
  for (int i = threadIdx.x; i < frame; i += 256) {
    const long p = start + i;
    if (p >= 0 && p < n) {
      const double v = (double)x[p];
      acc += v * v;
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[f] = (float)sqrt(red[0] / (double)frame);
}

__device__ __forceinline__ float interp_linear(const float* v, int nin, long nout, long i) {
  // F.interpolate(mode="linear", align_corners=False): fp32 source index as torch computes it
  const float scale = (float)nin / (float)nout;
  float src = scale * ((float)i + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  const int i0 = (int)src;
  const int i1 = i0 + (i0 < nin - 1 ? 1 : 0);
  const float l1 = src - (float)i0, l0 = 1.f - l1;
  return l0 * v[i0] + l1 * v[i1];
}

// audio * rms1^(1-rate) * max(rms2,1e-6)^(rate-1)        (pipeline.py:46-61)
__global__ void envelope_kernel(float* audio, const float* rms1, int n1, const float* rms2, int n2, long n,
                                float rate) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float r1 = interp_linear(rms1, n1, n, i);
    float r2 = interp_linear(rms2, n2, n, i);
    r2 = fmaxf(r2, 1e-6f);
    audio[i] = audio[i] * (powf(r1, 1.f - rate) * powf(r2, rate - 1.f));
  }
}

__global__ void absmax_kernel(const float* x, long n, unsigned int* out) {
  float m = 0.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));   // non-negative floats order as uints
}

// audio_max = max|x| / 0.99; scale = 32768 (/ audio_max if > 1); int16 truncation  (pipeline.py:457-461)
__global__ void to_int16_kernel(const float* x, short* out, long n, const unsigned int* amax) {
  const float audio_max = __uint_as_float(*amax) / 0.99f;
  float mult = 32768.f;
  if (audio_max > 1.f) mult = 32768.f / audio_max;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float v = x[i] * mult;
    out[i] = (short)(int)v;
  }
}

__global__ void f64_to_f32_kernel(const double* x, float* y, long n) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = (float)x[i];
}

// ---------------------------------------------------------------- geometry helpers
Geometry make_geometry(const rvcx_params& p, int tgt_sr) {
  Geometry g;
  g.t_pad = 16000L * p.x_pad;
  g.t_pad_tgt = (long)tgt_sr * p.x_pad;
  g.t_query = 16000L * p.x_query;
  g.t_center = 16000L * p.x_center;
  g.t_max = 16000L * p.x_max;
  return g;
}

std::vector<Chunk> plan_chunks(long n, const std::vector<long>& opt_ts, const Geometry& g) {
  // pipeline.py:381-447: (start, end) of audio_pad per vc() call and the pitch frame offsets
  std::vector<Chunk> plan;
  const long n_pad = n + 2 * g.t_pad;
  long s = 0, t = -1;
  for (long t0 : opt_ts) {
    t = t0 / 160 * 160;
    plan.push_back({s, t + 2 * g.t_pad + 160, s / 160});
    s = t;
  }
  if (t < 0) plan.push_back({0, n_pad, 0});
  else plan.push_back({t, n_pad, t / 160});
  return plan;
}

static long chunk_frames(const HubertModel& h, long samples) {
  return std::min<long>(samples / 160, 2L * hubert_frames(h, samples));
}

long out_capacity(const SynthModel& m, long n, const rvcx_params& p) {
  // samples VC.pipeline can return at most: every chunk yields <= (its frames)*upp - 2*t_pad_tgt and
  // consecutive chunks overlap by 2*t_pad + one window (pipeline.py:385-397)
  const Geometry g = make_geometry(p, m.cfg.sr);
  const long nc = (g.t_center > 0 ? n / g.t_center : 0) + 2;
  return (n / 160 + nc + 2) * m.upp;
}

namespace {

struct StageClock {
  Ctx& c;
  float* ms;
  std::vector<hipEvent_t> ev;
  StageClock(Ctx& cc, float* m) : c(cc), ms(m) {}
  int mark(hipStream_t s) {
    if (!ms) return -1;
    hipEvent_t e;
    RVCX_HIP(hipEventCreate(&e));
    RVCX_HIP(hipEventRecord(e, s));
    ev.push_back(e);
    return (int)ev.size() - 1;
  }
  float between(int a, int b) {
    float t = 0.f;
    RVCX_HIP(hipEventElapsedTime(&t, ev[a], ev[b]));
    return t;
  }
  ~StageClock() {
    for (auto e : ev) (void)hipEventDestroy(e);
  }
};

// chunk cut points (pipeline.py:330-344).  Needs a device->host copy of a few indices.
std::vector<long> find_cut_points(Ctx& c, const double* audio64, long n, const Geometry& g) {
  std::vector<long> opt;
  if (n + 160 <= g.t_max) return opt;   // audio_pad (window/2 each side) length vs t_max
  Arena& A = c.arena;
  double* asum = A.alloc<double>((size_t)n);
  hipLaunchKernelGGL(window_abs_sum_kernel, dim3((unsigned)std::min<long>(cdiv64(n, 256), 65535)), dim3(256), 0,
                     c.stream, audio64, asum, n);
  std::vector<long> ts;
  for (long t = g.t_center; t < n; t += g.t_center) ts.push_back(t);
  long* dres = A.alloc<long>(ts.size() + 1);
  for (size_t i = 0; i < ts.size(); ++i) {
    const long lo = std::max<long>(0, ts[i] - g.t_query), hi = std::min<long>(n, ts[i] + g.t_query);
    hipLaunchKernelGGL(argmin_first_kernel, dim3(1), dim3(256), 0, c.stream, asum, lo, hi, dres + i);
  }
  opt.resize(ts.size());
  RVCX_HIP(hipMemcpyAsync(opt.data(), dres, ts.size() * sizeof(long), hipMemcpyDeviceToHost, c.stream));
  RVCX_HIP(hipStreamSynchronize(c.stream));
  return opt;
}

}  // namespace

size_t convert_arena_bytes(Ctx& c, int model_id, long n, const rvcx_params& p) {
  const SynthModel& M = *c.synths[model_id];
  const Geometry g = make_geometry(p, M.cfg.sr);
  const long n_pad = n + 2 * g.t_pad;
  const long max_chunk = std::min<long>(n_pad, g.t_center + 2 * g.t_query + 2 * g.t_pad + 320);
  const int Tmax = (int)(max_chunk / 160 + 2);
  size_t b = (size_t)n_pad * 40 + ((size_t)64 << 20);
  b += rmvpe_arena_bytes(*c.rmvpe, 1, n_pad);
  b += hubert_arena_bytes(*c.hubert, 1, max_chunk);
  b += synth_arena_bytes(M, 1, Tmax) + (size_t)Tmax * (M.cfg.input_dim * 3 + M.upp * 3 + M.cfg.inter_channels) * 4;
  b += (size_t)out_capacity(M, n, p) * 8;
  if (c.index) b += index_arena_bytes(*c.index, Tmax);
  return b;
}

long noise_len_for(const Ctx& c, const SynthModel& m, long n, const rvcx_params& p) {
  // only defined for the un-chunked case (cut points are data dependent); callers with long audio
  // hand in a generous buffer (capacity below)
  const Geometry g = make_geometry(p, m.cfg.sr);
  const long cap_frames = n / 160 + (g.t_center > 0 ? n / g.t_center : 0) * 202 + 2 * (g.t_pad / 160) + 4;
  (void)c;
  return cap_frames * (long)(m.cfg.inter_channels + m.upp);
}

long get_f0_device(Ctx& c, const float* apad, long n_pad, const rvcx_params& p, int* coarse, float* f0,
                   hipStream_t s) {
  // VC.get_f0 (pipeline.py:132-201) on the already reflect-padded signal
  const long F = 1 + n_pad / 160, p_len = n_pad / 160;
  float* f0raw = c.arena.alloc<float>((size_t)F);
  rmvpe_forward(c, *c.rmvpe, 1, apad, n_pad, 0.03f, p.f0_min, p.f0_max, f0raw, nullptr, s);
  launch_f0_coarse(f0raw, f0, coarse, (int)p_len, p.pitch, p.f0_min, p.f0_max, s);
  return p_len;
}

long convert_one(Ctx& c, int model_id, const float* wav, long n, const rvcx_params& p, const float* noise,
                 short* out_pcm, float* out_f32, float* stage_ms) {
  RVCX_CHECK(c.hubert && c.rmvpe, "convert: hubert / rmvpe not loaded");
  RVCX_CHECK(model_id >= 0 && model_id < (int)c.synths.size() && c.synths[model_id], "convert: bad model id");
  const SynthModel& M = *c.synths[model_id];
  const Geometry g = make_geometry(p, M.cfg.sr);
  RVCX_CHECK(M.cfg.sr == 100 * M.upp, "synth sample rate must be 100 * prod(upsample_rates)");
  RVCX_CHECK(n > g.t_pad, "convert: clip shorter than the reflect padding");
  hipStream_t s = c.stream;
  Arena& A = c.arena;
  StageClock clk(c, stage_ms);
  const int e0 = clk.mark(s);
  // ---- 1. zero-phase high-pass in float64 (pipeline.py:329)
  const long n_pad = n + 2 * g.t_pad;
  double* ext = A.alloc<double>(highpass_ext_doubles(n));
  double* a64 = A.alloc<double>((size_t)n);
  float* a32 = A.alloc<float>((size_t)n);
  launch_highpass(wav, nullptr, ext, a64, a32, n, s);
  const int e1 = clk.mark(s);
  // ---- 2. chunk plan
  std::vector<long> opt_ts = find_cut_points(c, a64, n, g);
  std::vector<Chunk> plan = plan_chunks(n, opt_ts, g);
  // ---- 3. reflect pad, F0 once per utterance (pipeline.py:348-380)
  float* apad = A.alloc<float>((size_t)n_pad);
  launch_reflect_pad(a32, apad, 1, (int)n, (int)g.t_pad, n_pad, s);
  const long p_len = n_pad / 160;
  int* coarse = A.alloc<int>((size_t)p_len + 8);
  float* f0 = A.alloc<float>((size_t)p_len + 8);
  // F0 runs on the second stream beside HuBERT (they only share the padded signal) out of its own arena, so
  // the main stream's arena resets cannot recycle its workspace.  One host thread feeds both streams: RMVPE's
  // ~330 small launches are enqueued right after the first chunk's HuBERT conv extractor (7 long kernels),
  // otherwise HuBERT would sit idle for the milliseconds the host needs to enqueue RMVPE.
  hipStream_t sf = c.serial ? s : c.stream2;
  RVCX_HIP(hipEventRecord(c.ev_fork, s));
  c.arena_f0.reset();
  c.arena_f0.reserve(rmvpe_arena_bytes(*c.rmvpe, 1, n_pad) + ((size_t)64 << 20));
  int r0 = -1, r1 = -1;
  bool f0_enqueued = false;
  const std::function<void()> enqueue_f0 = [&]() {
    if (f0_enqueued) return;
    f0_enqueued = true;
    RVCX_HIP(hipStreamWaitEvent(sf, c.ev_fork, 0));
    r0 = clk.mark(sf);
    c.arena.swap(c.arena_f0);
    try {
      get_f0_device(c, apad, n_pad, p, coarse, f0, sf);
    } catch (...) {
      c.arena.swap(c.arena_f0);
      throw;
    }
    c.arena.swap(c.arena_f0);
    r1 = clk.mark(sf);
    RVCX_HIP(hipEventRecord(c.ev_join, sf));
  };
  bool joined = false;
  const int e2 = clk.mark(s);
  // ---- 4. per-chunk vc() (pipeline.py:203-287)
  const long cap = out_capacity(M, n, p);
  float* outf = out_f32 ? out_f32 : A.alloc<float>((size_t)cap);
  long out_n = 0;
  const float* np = noise;
  float t_hub = 0, t_idx = 0, t_syn[3] = {0, 0, 0};
  const int E = c.hubert->cfg.embed_dim;
  RVCX_CHECK(E == M.cfg.input_dim, "hubert embed dim != synthesizer input_dim");
  for (const Chunk& ch : plan) {
    const size_t mk = A.mark();
    const long ns = ch.e - ch.s;
    const int Th = hubert_frames(*c.hubert, ns);
    const int T = (int)std::min<long>(ns / 160, 2L * Th);
    RVCX_CHECK(Th > 0 && T > 0, "convert: chunk too short");
    RVCX_CHECK((long)T * M.upp > 2 * g.t_pad_tgt, "convert: chunk shorter than its padding");
    const int h0 = clk.mark(s);
    float* feats = A.alloc<float>((size_t)E * Th);
    {
      const size_t mk2 = A.mark();
      hipStream_t sh = (c.stream_h && !c.serial) ? c.stream_h : s;
      if (sh != s) {
        RVCX_HIP(hipEventRecord(c.ev_hub, s));
        RVCX_HIP(hipStreamWaitEvent(sh, c.ev_hub, 0));
      }
      hubert_forward(c, *c.hubert, 1, apad + ch.s, ns, 12, feats, sh, &enqueue_f0);
      if (sh != s) {
        RVCX_HIP(hipEventRecord(c.ev_hub, sh));
        RVCX_HIP(hipStreamWaitEvent(s, c.ev_hub, 0));
      }
      A.reset(mk2);
    }
    const int h1 = clk.mark(s);
    const float* feats0 = feats;
    float* blended = feats;
    const bool use_protect = p.protect < 0.5f;
    if (c.index && p.index_rate != 0.f) {
      if (use_protect) {
        float* keep = A.alloc<float>((size_t)E * Th);
        RVCX_HIP(hipMemcpyAsync(keep, feats, (size_t)E * Th * sizeof(float), hipMemcpyDeviceToDevice, s));
        feats0 = keep;
      }
      const size_t mk2 = A.mark();
      index_blend(c, *c.index, blended, Th, p.index_rate, nullptr, nullptr, s);
      A.reset(mk2);
    }
    const int h2 = clk.mark(s);
    if (!joined) {   // first consumer of f0 / coarse
      RVCX_HIP(hipStreamWaitEvent(s, c.ev_join, 0));
      joined = true;
    }
    float* phone = A.alloc<float>((size_t)E * T);
    launch_upsample_protect(blended, feats0, f0 + ch.f0_off, phone, E, Th, T, p.protect, use_protect ? 1 : 0, s);
    // noise (parity: packed [z (inter*T) | src (T*upp)] per chunk in draw order; else Philox)
    const size_t nz = (size_t)M.cfg.inter_channels * T, nsrc = (size_t)T * M.upp;
    float *zn, *sn;
    if (np) {
      zn = const_cast<float*>(np);
      sn = const_cast<float*>(np) + nz;
      np += nz + nsrc;
    } else {
      zn = A.alloc<float>(nz);
      sn = A.alloc<float>(nsrc);
      const uint64_t off = (uint64_t)(&ch - &plan[0]) << 36;
      launch_randn(zn, nz, p.seed, off, s);
      launch_randn(sn, nsrc, p.seed, off + ((uint64_t)1 << 35), s);
    }
    float* wavout = A.alloc<float>(nsrc);
    SynthIO io;
    io.B = 1;
    io.T = T;
    io.phone_ct = phone;
    io.pitch = coarse + ch.f0_off;
    io.pitchf = f0 + ch.f0_off;
    int sid = p.sid;
    io.sid_host = &sid;
    io.z_noise = zn;
    io.src_noise = sn;
    io.out = wavout;
    float ms3[3] = {0, 0, 0};
    synth_forward(c, M, io, stage_ms ? ms3 : nullptr);
    const long keep_n = (long)nsrc - 2 * g.t_pad_tgt;
    RVCX_CHECK(out_n + keep_n <= cap, "convert: output capacity exceeded");
    RVCX_HIP(hipMemcpyAsync(outf + out_n, wavout + g.t_pad_tgt, (size_t)keep_n * sizeof(float),
                            hipMemcpyDeviceToDevice, s));
    out_n += keep_n;
    if (stage_ms) {
      RVCX_HIP(hipStreamSynchronize(s));
      t_hub += clk.between(h0, h1);
      t_idx += clk.between(h1, h2);
      for (int i = 0; i < 3; ++i) t_syn[i] += ms3[i];
    }
    A.reset(mk);   // stream-ordered reuse: later launches on the same stream see the finished chunk
  }
  const int e3 = clk.mark(s);
  // ---- 5. RMS envelope, peak normalise, int16 (pipeline.py:449-461)
  if (p.volume_envelope != 1.f) {
    const int tgt = M.cfg.sr;
    const int f1 = 16000 / 2 * 2, h1 = 16000 / 2, f2 = tgt / 2 * 2, h2 = tgt / 2;
    const int n1 = (int)(1 + (n + 2 * (f1 / 2) - f1) / h1), n2 = (int)(1 + (out_n + 2 * (f2 / 2) - f2) / h2);
    float* r1 = A.alloc<float>((size_t)n1);
    float* r2 = A.alloc<float>((size_t)n2);
    hipLaunchKernelGGL(frame_rms_kernel<double>, dim3(n1), dim3(256), 0, s, a64, r1, n, f1, h1, n1);
    hipLaunchKernelGGL(frame_rms_kernel<float>, dim3(n2), dim3(256), 0, s, outf, r2, out_n, f2, h2, n2);
    hipLaunchKernelGGL(envelope_kernel, dim3((unsigned)std::min<long>(cdiv64(out_n, 256), 65535)), dim3(256), 0, s,
                       outf, r1, n1, r2, n2, out_n, p.volume_envelope);
  }
  unsigned int* amax = A.alloc<unsigned int>(1);
  RVCX_HIP(hipMemsetAsync(amax, 0, sizeof(unsigned int), s));
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)std::min<long>(cdiv64(out_n, 256), 1024)), dim3(256), 0, s, outf,
                     out_n, amax);
  hipLaunchKernelGGL(to_int16_kernel, dim3((unsigned)std::min<long>(cdiv64(out_n, 256), 65535)), dim3(256), 0, s,
                     outf, out_pcm, out_n, amax);
  const int e4 = clk.mark(s);
  RVCX_HIP(hipGetLastError());
  if (stage_ms) {
    RVCX_HIP(hipStreamSynchronize(s));
    stage_ms[0] = clk.between(e0, e1);
    stage_ms[1] = r0 >= 0 ? clk.between(r0, r1) : 0.f;   // on stream2, overlapped with the HuBERT stage
    stage_ms[2] = t_hub;
    stage_ms[3] = t_idx;
    stage_ms[4] = t_syn[0];
    stage_ms[5] = t_syn[1];
    stage_ms[6] = t_syn[2];
    stage_ms[7] = clk.between(e3, e4);
    stage_ms[8] = clk.between(e0, e4);
  }
  return out_n;
}

}  // namespace rvcx
