// VC.pipeline (rvc/infer/pipeline.py:289-467) on the device: zero-phase high-pass, silence-aligned
// chunking, F0 once per utterance, per-chunk HuBERT -> (index blend) -> x2 upsample / protect ->
// Synthesizer.infer, trim, RMS envelope, peak normalise, int16.
#include <algorithm>
#include <chrono>
#include <array>
#include <cmath>
#include <cstdlib>
#include <functional>

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

// scipy filtfilt(method="pad", padtype="odd"): odd extension by PADLEN samples on both sides.
// All highpass kernels take a batch of equal-length signals: blockIdx.y = item, element strides xs / es / ys.
__global__ void odd_ext_kernel(const float* __restrict__ x32, const double* __restrict__ x64, double* ext, long n,
                               long xs, long es, const int* __restrict__ ns) {
  const long b = blockIdx.y;
  if (ns) n = ns[b];                       // ragged micro-batch: the item's own length (rows stay xs / es apart)
  const long m = n + 2 * PADLEN;
  ext += b * es;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < m; i += (long)gridDim.x * 256) {
    auto X = [&](long j) -> double { return x64 ? x64[b * xs + j] : (double)x32[b * xs + j]; };
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

// one lane per slice; `rev` walks the buffers backwards (filtfilt's second, time-reversed pass).
// Round 6: the lane's samples arrive 16 at a time, the NEXT block requested before the current one is filtered.  A pass is
// ~30 single-wave workgroups of 3328 serial steps; with the loads of a block issued and awaited in place (round 2-5: 8 at a
// time) every block paid a memory round trip -- 210 us per pass for ~60 us of dependent float64 FMAs (0.5 ms of a single
// clip's 25.5).  Same recursion, same order: the bits do not change.
constexpr int IIR_BLK = 16;
static_assert(IIR_CHUNK % (2 * IIR_BLK) == 0 && IIR_WARM % (2 * IIR_BLK) == 0, "warm-up and slice are whole block pairs");
template <bool rev>
__global__ void iir_slice_kernel(const double* __restrict__ in, double* __restrict__ out, long m,
                                 long es, IirCoef cf, const int* __restrict__ ns) {
  const long p = blockIdx.x * (long)blockDim.x + threadIdx.x;
  const long s = p * IIR_CHUNK;
  if (ns) m = ns[blockIdx.y] + 2 * PADLEN;
  if (s >= m) return;
  in += blockIdx.y * es;
  out += blockIdx.y * es;
  const long e = min(m, s + IIR_CHUNK);
  const long w0 = max(0L, s - IIR_WARM);
  double z[5];
  {
    const double x0 = in[rev ? m - 1 - w0 : w0];
    for (int k = 0; k < 5; ++k) z[k] = cf.zi[k] * x0;      // lfilter(..., zi = zi * x[0])
  }
  // straight-line bodies (no branch around a fetch: the compiler then waits for every load in flight before it issues the
  // next ones).  A fetch that would run past the signal is moved back inside it: its values are never used (the block loops
  // below only filter whole blocks inside [w0, e)).  One address per block, the elements at immediate offsets.
  auto fetch = [&](double (&v)[IIR_BLK], long i0) {
    const long c = min(i0, m - IIR_BLK);
    const double* p = rev ? in + (m - 1 - c) : in + c;
#pragma unroll
    for (int k = 0; k < IIR_BLK; ++k) v[k] = rev ? p[-k] : p[k];
    __builtin_amdgcn_sched_barrier(0);    // or the scheduler sinks these loads below the block they are meant to overlap
  };
  auto warm = [&](const double (&v)[IIR_BLK]) {
#pragma unroll
    for (int k = 0; k < IIR_BLK; ++k) (void)iir_step(cf, z, v[k]);
  };
  auto emit = [&](const double (&v)[IIR_BLK], long i0) {
    double* q = rev ? out + (m - 1 - i0) : out + i0;
#pragma unroll
    for (int k = 0; k < IIR_BLK; ++k) q[rev ? -k : k] = iir_step(cf, z, v[k]);
  };
  long i = w0;
  double xa[IIR_BLK], xb[IIR_BLK];
  fetch(xa, i);
  // s - w0 is IIR_WARM or s, both multiples of 2 * IIR_BLK: the warm-up ends on a pair boundary
  for (; i + 2 * IIR_BLK <= s; i += 2 * IIR_BLK) {
    fetch(xb, i + IIR_BLK);
    warm(xa);
    fetch(xa, i + 2 * IIR_BLK);
    warm(xb);
  }
  for (; i + 2 * IIR_BLK <= e; i += 2 * IIR_BLK) {
    fetch(xb, i + IIR_BLK);
    emit(xa, i);
    fetch(xa, i + 2 * IIR_BLK);
    emit(xb, i + IIR_BLK);
  }
  for (; i < e; ++i) {                                    // the last slice's ragged end
    const double y = iir_step(cf, z, in[rev ? m - 1 - i : i]);
    if (i >= s) out[rev ? m - 1 - i : i] = y;
  }
}

__global__ void crop_ext_kernel(const double* ext, double* y64, float* y32, long n, long es, long ys,
                                const int* __restrict__ ns) {
  const long b = blockIdx.y;
  ext += b * es;
  const long nb = ns ? ns[b] : n;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const double v = i < nb ? ext[PADLEN + i] : 0.0;    // beyond the item's length the row is zero
    if (y64) y64[b * ys + i] = v;
    if (y32) y32[b * ys + i] = (float)v;
  }
}

static const IirCoef& iir_coef() {
  // function-local static: initialised exactly once, also when several contexts start on different threads
  static const IirCoef cf = [] {
    IirCoef c;
    for (int i = 0; i < 6; ++i) {
      c.b[i] = BH[i];
      c.a[i] = AH[i];
    }
    lfilter_zi(c.zi);
    return c;
  }();
  return cf;
}

// B equal-length signals (element stride xs in, n out); ext: B * highpass_ext_doubles(n) doubles of scratch
void launch_highpass(const float* x32, const double* x64, double* ext, double* y64, float* y32, long n,
                     hipStream_t s, int B, long xs, const int* ns) {
  RVCX_CHECK(n > PADLEN, "highpass: input shorter than filtfilt's pad length");
  const IirCoef& cf = iir_coef();
  if (xs == 0) xs = n;
  const long m = n + 2 * PADLEN;
  const long es = (long)highpass_ext_doubles(n);
  double* tmp = ext + m;
  const unsigned g = (unsigned)std::min<long>(cdiv64(m, 256), 65535);
  const int nslices = (int)cdiv64(m, IIR_CHUNK);
  hipLaunchKernelGGL(odd_ext_kernel, dim3(g, B), dim3(256), 0, s, x32, x64, ext, n, xs, es, ns);
  hipLaunchKernelGGL(iir_slice_kernel<false>, dim3(cdiv(nslices, 64), B), dim3(64), 0, s, ext, tmp, m, es, cf, ns);
  hipLaunchKernelGGL(iir_slice_kernel<true>, dim3(cdiv(nslices, 64), B), dim3(64), 0, s, tmp, ext, m, es, cf, ns);
  hipLaunchKernelGGL(crop_ext_kernel, dim3(g, B), dim3(256), 0, s, ext, y64, y32, n, es, n, ns);
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

// (round 6: one atomic per workgroup instead of one per wave -- 4096 atomics on one word were 45 of the kernel's 49 us)
__global__ __launch_bounds__(256) void absmax_kernel(const float* x, long n, unsigned int* out) {
  __shared__ float part[4];
  float m = 0.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0)
    atomicMax(out, __float_as_uint(fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]))));   // non-negative floats order as uints
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
  RVCX_CHECK(p.x_pad > 0 && p.x_query > 0 && p.x_center > 0 && p.x_max > 0,
             "chunk geometry: x_pad, x_query, x_center and x_max must be positive (Config, infer.py:36-43)");
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

// number of cut points VC.pipeline places in an n-sample clip: len(range(t_center, n, t_center)) when the clip
// (plus the half-window pad) exceeds t_max, else none  (pipeline.py:330-344)
static long cut_count(long n, const Geometry& g) {
  if (n + 160 <= g.t_max) return 0;
  return (n - 1) / g.t_center;
}

long out_capacity(const SynthModel& m, long n, const rvcx_params& p) {
  // samples VC.pipeline can return at most: every chunk yields <= (its frames)*upp - 2*t_pad_tgt and
  // consecutive chunks overlap by 2*t_pad + one window (pipeline.py:385-397)
  const Geometry g = make_geometry(p, m.cfg.sr);
  const long cap = (n / 160 + cut_count(n, g) + 4) * m.upp;
  // resample_sr (pipeline.py:453-454): the output leaves at another rate
  if (p.resample_sr >= 16000 && p.resample_sr != m.cfg.sr) return std::max(cap, resample_out_len(cap, m.cfg.sr, p.resample_sr) + 8);
  return cap;
}

long noise_len_for(const Ctx& c, const SynthModel& m, long n, const rvcx_params& p) {
  // Upper bound for any placement of the cut points: the chunks tile the padded clip and every cut adds one
  // overlap of 2*t_pad + window samples (pipeline.py:385-397); each frame draws inter + upp Gaussians.
  (void)c;
  const Geometry g = make_geometry(p, m.cfg.sr);
  const long n_pad = n + 2 * g.t_pad;
  const long cap_frames = n_pad / 160 + cut_count(n, g) * (2 * g.t_pad / 160 + 1) + 2;
  return cap_frames * (long)(m.cfg.inter_channels + m.upp);
}

namespace {

struct StageClock {
  bool on;
  std::vector<hipEvent_t> ev;
  explicit StageClock(bool enabled) : on(enabled) {}
  int mark(hipStream_t s) {
    if (!on) return -1;
    hipEvent_t e;
    RVCX_HIP(hipEventCreate(&e));
    RVCX_HIP(hipEventRecord(e, s));
    ev.push_back(e);
    return (int)ev.size() - 1;
  }
  float between(int a, int b) {
    if (a < 0 || b < 0) return 0.f;
    float t = 0.f;
    RVCX_HIP(hipEventElapsedTime(&t, ev[a], ev[b]));
    return t;
  }
  ~StageClock() {
    for (auto e : ev) (void)hipEventDestroy(e);
  }
};

// one utterance of the call, with everything the stages exchange (device pointers)
struct Utt {
  int io = 0;                 // index into the caller's arrays
  long n = 0, n_pad = 0, p_len = 0, cap = 0;
  const float* noise = nullptr;   // packed parity noise on the device (or null)
  long noise_cap = 0;
  double* rep = nullptr;          // f0-file track on the device (pipeline.py:185-191) or null
  int rep_n = 0;
  short* pcm = nullptr;
  float* outf = nullptr;
  long out_n = 0;
  // front-end products (live in the micro-batch's front set)
  const double* a64 = nullptr;
  const float* apad = nullptr;
  const int* coarse = nullptr;
  const float* f0 = nullptr;
  std::vector<Chunk> plan;
};

struct Job {                  // one vc() call of the reference: chunk `ci` of utterance `u` (index in the micro-batch)
  int u, ci;
  long s, e, f0_off, out_off, noise_off;
  int Th, T;
};

}  // namespace

size_t convert_item_bytes(Ctx& c, int model_id, long n, const rvcx_params& p) {
  // arena bytes one utterance of a micro-batch needs beyond the call-long buffers (front set x2 + work area)
  const SynthModel& M = *c.synths[model_id];
  const Geometry g = make_geometry(p, M.cfg.sr);
  const long n_pad = n + 2 * g.t_pad;
  const long max_chunk = std::min<long>(n_pad, g.t_center + 2 * g.t_query + 2 * g.t_pad + 320);
  const int Tmax = (int)(max_chunk / 160 + 2);
  size_t b = 2 * ((size_t)n_pad * 40 + (1 << 16));                    // two front sets
  b += 2 * (size_t)c.hubert->cfg.embed_dim *
       (size_t)((n_pad + (cut_count(n, g) + 1) * (2 * g.t_pad + 640)) / 320 + 8) * 4;   // HuBERT features of both sets
  b += hubert_arena_bytes(*c.hubert, 1, max_chunk) - ((size_t)64 << 20) + (size_t)max_chunk * 4;
  b += synth_arena_bytes(M, 1, Tmax) - ((size_t)64 << 20) +
       (size_t)Tmax * ((size_t)M.cfg.input_dim * 3 + (size_t)M.upp * 3 + M.cfg.inter_channels + 8) * 4;
  if (c.index) b += index_arena_bytes(*c.index, Tmax);
  return b;
}

size_t convert_call_bytes(Ctx& c, int model_id, long n, const rvcx_params& p, bool f64, bool noise) {
  // call-long buffers of one utterance: staged input, parity noise, float + int16 output
  const SynthModel& M = *c.synths[model_id];
  size_t b = (size_t)n * (f64 ? 8 : 4) + (size_t)out_capacity(M, n, p) * 6 + 1024;
  if (noise) b += (size_t)noise_len_for(c, M, n, p) * 4 + 256;
  return b;
}

// ---- length classes (ragged micro-batches)
// Uncut utterances whose padded length falls into the same class of `bucket_frames()` 10 ms frames share a micro-batch:
// every network up to the decoder is launched with the class's geometry (the longest length of the class: tiles,
// split-K factors, key splits -- everything that shapes an fp32 summation order) and per-item length arrays carry
// each utterance's own arithmetic (filter ends, reflect padding, frame counts, statistics, masks, recurrence
// length).  A single utterance takes exactly the same path with B = 1, so a batch item is bit-identical to its single
// run whatever shares the batch with it; the NSF decoder runs every utterance at its own length.
// RVCX_BUCKET_FRAMES: class width in frames (a multiple of 32, the F0 U-Net's row granularity; default 128 = 1.28 s);
// 0 = equal lengths only.  Tiles that lie entirely behind an item's length are skipped (conv_h3, gemm_h3), so the
// padding of a class costs little.
int bucket_frames() {
  static const int v = [] {
    const char* e = getenv("RVCX_BUCKET_FRAMES");
    int f = e ? atoi(e) : 128;      // measured on C5 (tools/sweep_bucket.sh): 64 / 128 / 256 within noise of each other, 32 slower
    if (f <= 0) return 0;
    return std::max(32, (f + 31) / 32 * 32);
  }();
  return v;
}

// the sample count whose geometry an n-sample utterance is run with (>= n)
long bucket_length(long n, const rvcx_params& p, const Geometry& g) {
  const int cf = bucket_frames();
  // cut clips keep their own geometry; FCPE's network takes equal lengths only (crepe runs one item at a time anyway)
  if (cf == 0 || p.f0_method == RVCX_F0_FCPE || n + 160 > g.t_max) return n;
  const long k = ((n + 2 * g.t_pad) / 160) / cf;
  return (k + 1) * cf * 160 - 1 - 2 * g.t_pad;          // the longest clip whose padded frame count is (k + 1) cf - 1
}

int convert_micro_batch(Ctx& c, int model_id, long n, const rvcx_params& p) {
  // 16: the BiGRU cluster kernel's co-residency bound (2 directions x 16 items x 4 workgroups = 128); C5 +3 %, C3 +1 % over 8
  static const int env_max = getenv("RVCX_MAX_BATCH") ? std::max(1, atoi(getenv("RVCX_MAX_BATCH"))) : 16;
  // Activation budget: RVCX_ARENA_GB if set; else 100 GB (64 x 30 s: micro-batches of 16 instead of 11, +1 %) but never
  // more than 70 % of what the device has FREE right now -- a second context on the same GPU, or a smaller GPU, must not
  // turn the default into an out-of-memory error (the arena only ever grows: what this context already holds counts as free).
  // The free-memory probe runs ONCE per context state (first conversion after a load / unload: api_call clears the cached
  // value on every non-repeating entry point), on the context's OWN device -- rvcx_micro_batch reaches this function
  // without api_call's hipSetDevice -- so that the micro-batch partition is a function of (inputs, resident models,
  // environment), not of what another context happened to hold at the moment of the call.  The 70 % rule is a heuristic
  // against the obvious failure (two contexts, a smaller GPU), not an out-of-memory guarantee: two contexts created
  // together can each see the same free bytes.
  size_t budget = (size_t)(getenv("RVCX_ARENA_GB") ? atoi(getenv("RVCX_ARENA_GB")) : 100) << 30;
  if (!getenv("RVCX_ARENA_GB")) {
    if (c.arena_budget == 0) {
      int cur = -1;
      (void)hipGetDevice(&cur);
      if (cur != c.device) (void)hipSetDevice(c.device);
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0) {
        const size_t mine = c.arena.capacity() + c.arena_f0.capacity() + c.arena_hub.capacity();
        c.arena_budget = std::max<size_t>(1, std::min(budget, (size_t)((double)(free_b + mine) * 0.7)));
      } else {
        c.arena_budget = budget;
      }
      if (cur >= 0 && cur != c.device) (void)hipSetDevice(cur);
    }
    budget = c.arena_budget;
  }
  const size_t per = convert_item_bytes(c, model_id, n, p) + f0_arena_bytes(c, p, 1, n + 32000L * p.x_pad);
  return (int)std::max<size_t>(1, std::min<size_t>((size_t)env_max, budget / std::max<size_t>(per, 1)));
}

// the F0 back-end VC.get_f0 dispatches on (pipeline.py:142-181): which model must be resident, and its workspace
void check_f0_backend(const Ctx& c, const rvcx_params& p) {
  if (p.f0_method == RVCX_F0_RMVPE) {
    RVCX_CHECK(c.rmvpe != nullptr, "get_f0: rmvpe not loaded");
  } else if (p.f0_method == RVCX_F0_FCPE) {
    RVCX_CHECK(c.fcpe != nullptr, "get_f0: fcpe not loaded");
  } else if (p.f0_method == RVCX_F0_CREPE) {
    RVCX_CHECK(c.crepe != nullptr, "get_f0: crepe not loaded");
  } else {
    fail("get_f0: unknown f0_method " + std::to_string(p.f0_method));
  }
}

size_t f0_arena_bytes(const Ctx& c, const rvcx_params& p, int B, long n_pad) {
  check_f0_backend(c, p);
  if (p.f0_method == RVCX_F0_FCPE) return fcpe_arena_bytes(*c.fcpe, B, n_pad) + (size_t)B * (n_pad / 160 + 8) * 16;
  if (p.f0_method == RVCX_F0_CREPE)   // one item at a time
    return crepe_arena_bytes(*c.crepe, n_pad, crepe_hop(p)) + (size_t)B * (n_pad / 160 + 8) * 16;
  return rmvpe_arena_bytes(*c.rmvpe, B, n_pad);
}

int crepe_hop(const rvcx_params& p) { return p.hop_length > 0 ? p.hop_length : 128; }

// VC.get_f0_crepe (pipeline.py:86-117) for one padded signal on the device: f0raw receives p_len frames
void crepe_f0_device(Ctx& c, const float* x, long n, const rvcx_params& p, long p_len, const F0Extra* ex, float* f0raw,
                     hipStream_t s) {
  const CrepeModel& m = *c.crepe;
  const int hop = crepe_hop(p);
  const long F = crepe_frames(n, hop);
  // x /= np.quantile(np.abs(x), 0.999): an order statistic of the whole signal -- read back and selected on the host
  // (this optional back-end is not on the headline path; the frames are renormalised to unit variance right after)
  std::vector<float> h((size_t)n);
  RVCX_HIP(hipMemcpyAsync(h.data(), x, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, s));
  RVCX_HIP(hipStreamSynchronize(s));
  const float scale = (float)crepe_quantile999(h);
  RVCX_CHECK(std::isfinite(scale), "crepe: non-finite samples in the signal");
  if (!(scale > 0.f)) {
    // a silent clip: the reference divides by 0, every frame turns NaN and get_f0_crepe's nan_to_num leaves an all-zero
    // track (pipeline.py:91,111-118) -- this item's f0 is zero, the rest of the batch goes on
    RVCX_HIP(hipMemsetAsync(f0raw, 0, (size_t)p_len * sizeof(float), s));
    return;
  }
  float* dith = c.arena.alloc<float>((size_t)F);
  if (ex && ex->dither) {
    RVCX_CHECK(ex->dither_n >= F, "crepe: the dither array is shorter than the frame count");
    RVCX_HIP(hipMemcpyAsync(dith, ex->dither, (size_t)F * sizeof(float), hipMemcpyHostToDevice, s));
  } else {
    launch_crepe_dither(dith, F, p.seed + 0x63726570ull, (uint64_t)(ex ? ex->seed_offset : 0) << 32, s);
  }
  float* pitch = c.arena.alloc<float>((size_t)F);
  crepe_forward(c, m, x, n, scale, hop, p.f0_min, p.f0_max, dith, pitch, nullptr, nullptr, s);
  launch_crepe_resize(pitch, F, p_len, f0raw, s);
}

long get_f0_device(Ctx& c, const float* apad, long n_pad, const rvcx_params& p, int* coarse, float* f0,
                   hipStream_t s, int B, long out_stride, const std::function<void()>* mid, const F0Extra* extra,
                   const int* ns_host) {
  // VC.get_f0 (pipeline.py:132-201) on already reflect-padded signals (B, n_pad); coarse / f0 rows of out_stride
  const long F = 1 + n_pad / 160, p_len = n_pad / 160;
  check_f0_backend(c, p);
  RVCX_CHECK(!ns_host || p.f0_method != RVCX_F0_FCPE, "get_f0: ragged batches are an rmvpe / mangio-crepe feature");
  float* f0raw = c.arena.alloc<float>((size_t)B * F);
  if (p.f0_method == RVCX_F0_CREPE) {   // pipeline.py:151-152: get_f0_crepe(x, f0_min, f0_max, p_len, hop_length)
    if (mid) (*mid)();
    for (int b = 0; b < B; ++b) {      // one item at a time, each at ITS OWN length (rows stay n_pad apart)
      const size_t mark = c.arena.mark();
      const long nb = ns_host ? ns_host[b] : n_pad, pl = nb / 160;
      crepe_f0_device(c, apad + (size_t)b * n_pad, nb, p, pl, extra ? extra + b : nullptr, f0raw + (size_t)b * F, s);
      c.arena.reset(mark);
      launch_f0_coarse(f0raw + (size_t)b * F, f0 + (size_t)b * out_stride, coarse + (size_t)b * out_stride, (int)pl,
                       p.pitch, p.f0_min, p.f0_max, s);
    }
    return p_len;
  }
  if (p.f0_method == RVCX_F0_FCPE) {   // pipeline.py:169-181: threshold 0.03, compute_f0(x, p_len)
    fcpe_forward(c, *c.fcpe, B, apad, n_pad, 0.03f, f0raw, nullptr, nullptr, s, mid);
    fcpe_post_coarse(c, f0raw, B, (int)F, (int)p_len, f0, coarse, out_stride, p.pitch, p.f0_min, p.f0_max, s);
    return p_len;
  }
  rmvpe_forward(c, *c.rmvpe, B, apad, n_pad, 0.03f, p.f0_min, p.f0_max, f0raw, nullptr, s, nullptr, mid, ns_host);
  for (int b = 0; b < B; ++b)
    launch_f0_coarse(f0raw + (size_t)b * F, f0 + (size_t)b * out_stride, coarse + (size_t)b * out_stride,
                     (int)(ns_host ? ns_host[b] / 160 : p_len), p.pitch, p.f0_min, p.f0_max, s);
  return p_len;
}

// VC.pipeline for a list of utterances (pipeline.py:289-467).
//
// Utterances of equal length form micro-batches that go through every network as one launch sequence with B > 1
// (chunks of cut utterances are batch items too when their lengths agree): the serial BiGRU recurrence, the ~600
// small launches of the front end and the TextEncoder/flow section are paid once per micro-batch instead of once
// per clip.  Three streams: `sf` runs the front end (upload, float64 zero-phase high-pass, cut search, reflect
// pad, RMVPE F0) of micro-batch k+1 while the main stream is still in the NSF decoder of micro-batch k; HuBERT keeps
// its CU-masked stream; finished PCM leaves behind its micro-batch on the main stream.  No host synchronisation inside the loop except
// the cut-point read-back of clips longer than x_max seconds.
void convert_batch(Ctx& c, int model_id, std::vector<UttIO>& ios, const rvcx_params& p, float* stage_ms) {
  RVCX_CHECK(c.hubert != nullptr, "convert: hubert not loaded");
  check_f0_backend(c, p);
  RVCX_CHECK(model_id >= 0 && model_id < (int)c.synths.size() && c.synths[model_id], "convert: bad model id");
  const SynthModel& M = *c.synths[model_id];
  const Geometry g = make_geometry(p, M.cfg.sr);
  RVCX_CHECK(M.cfg.sr == 100 * M.upp, "synth sample rate must be 100 * prod(upsample_rates)");
  // the feature width VC.vc works with (pipeline.py:228-236): the HuBERT's embed_dim for RVC v2 voice models, the width of
  // its final_proj (256) for RVC v1 ones (output layer 9 + final_proj: hubert_features_for)
  const int E = M.cfg.input_dim, inter = M.cfg.inter_channels;
  RVCX_CHECK(E == c.hubert->cfg.embed_dim || (c.hubert->has_final_proj && E == c.hubert->final_proj.cout),
             "the voice model's input_dim is neither the HuBERT's embed_dim (v2) nor its final_proj width (v1)");
  const int NB = (int)ios.size();
  if (NB == 0) return;
  hipStream_t s = c.stream;
  hipStream_t sf = c.serial ? s : c.stream2;
  hipStream_t sio = (c.serial || !c.stream_io) ? s : c.stream_io;   // copy-out stream (the main stream: see ctx.h)
  Arena& A = c.arena;
  StageClock clk(stage_ms != nullptr);

  // ---- micro-batches: utterances sorted by length (stable); utterances of one length class (bucket_length: the same
  // launch geometry `nd`) are grouped, at most `mb_max` per group.  `ragged`: the members differ from nd.
  std::vector<int> order(NB);
  for (int i = 0; i < NB; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return ios[a].n < ios[b].n; });
  struct MB {
    int first, count;
    long nd;        // the length whose geometry every launch of the micro-batch takes (>= every member's length)
    bool ragged;
  };
  std::vector<MB> mbs;
  long n_max = 0;
  bool any_f64 = false, any_noise = false;
  for (int i = 0; i < NB;) {
    const long n = ios[order[i]].n;
    RVCX_CHECK(n > 18, "convert: the clip must be longer than the zero-phase filter's edge padding (18 samples; scipy's filtfilt raises for it too)");
    const long nd = bucket_length(n, p, g);
    const int cap = convert_micro_batch(c, model_id, nd, p);
    int j = i;
    while (j < NB && bucket_length(ios[order[j]].n, p, g) == nd) ++j;
    // the class's members in ceil(m / cap) micro-batches of (almost) equal size: 17 members are 9 + 8, not 16 + 1
    const int m = j - i, parts = (m + cap - 1) / cap;
    for (int q = 0; q < parts; ++q) {
      const int a = i + (int)((long)m * q / parts), b = i + (int)((long)m * (q + 1) / parts);
      bool ragged = false;
      for (int t = a; t < b; ++t) ragged |= ios[order[t]].n != nd;
      mbs.push_back({a, b - a, nd, ragged});
    }
    i = j;
  }
  c.last_mbs.clear();
  for (const auto& mb : mbs) c.last_mbs.push_back(mb.count);
  size_t call_bytes = 0, mb_bytes = 0, f0_bytes = 0, hub_bytes = 0;
  for (const auto& u : ios) {
    any_f64 |= u.wav64 != nullptr;
    any_noise |= u.noise != nullptr;
    n_max = std::max(n_max, u.n);
    call_bytes += convert_call_bytes(c, model_id, u.n, p, u.wav64 != nullptr, u.noise != nullptr);
    if (p.resample_sr >= 16000) call_bytes += (size_t)out_capacity(M, u.n, p) * 4 + ((size_t)1 << 20);
    if (u.inp_f0) call_bytes += (size_t)65536 * sizeof(double) + 256;      // delta_t is an int16
  }
  for (const auto& mb : mbs) {
    const long n = mb.nd;
    mb_bytes = std::max(mb_bytes, (size_t)mb.count * convert_item_bytes(c, model_id, n, p));
    f0_bytes = std::max(f0_bytes, f0_arena_bytes(c, p, mb.count, n + 2 * g.t_pad));
    const long n_pad = n + 2 * g.t_pad;
    const long max_chunk = std::min<long>(n_pad, g.t_center + 2 * g.t_query + 2 * g.t_pad + 320);
    hub_bytes = std::max(hub_bytes, hubert_arena_bytes(*c.hubert, mb.count, max_chunk) + (size_t)mb.count * max_chunk * 8);
  }
  (void)any_f64;
  (void)any_noise;
  c.ensure_splitk((int)std::max<size_t>(1, [&] {
    int m = 1;
    for (const auto& mb : mbs) m = std::max(m, mb.count);
    return (size_t)m;
  }()));
  A.reserve(call_bytes + mb_bytes + ((size_t)96 << 20));
  A.reset();
  c.arena_f0.reset();
  c.arena_f0.reserve(f0_bytes + ((size_t)64 << 20));
  c.arena_hub.reset();
  c.arena_hub.reserve(hub_bytes + ((size_t)64 << 20));

  // ---- call-long buffers
  std::vector<std::vector<double>> f0_tracks;      // host copies stay alive until the call's last synchronisation
  f0_tracks.reserve(NB);
  std::vector<Utt> utts(NB);
  for (int i = 0; i < NB; ++i) {
    Utt& u = utts[i];
    u.io = i;
    u.n = ios[i].n;
    u.n_pad = u.n + 2 * g.t_pad;
    u.p_len = u.n_pad / 160;
    u.cap = out_capacity(M, u.n, p);
    u.pcm = A.alloc<short>((size_t)u.cap);
    u.outf = A.alloc<float>((size_t)u.cap);
    if (ios[i].noise) {
      u.noise_cap = noise_len_for(c, M, u.n, p);
      u.noise = A.alloc<float>((size_t)u.noise_cap);
    }
    if (ios[i].inp_f0 && ios[i].inp_f0_rows > 0) {       // f0 file: the replacement track, built like the reference does
      f0_tracks.push_back(f0_file_track(ios[i].inp_f0, ios[i].inp_f0_rows));
      const std::vector<double>& tr = f0_tracks.back();
      if (!tr.empty()) {
        u.rep_n = (int)tr.size();
        u.rep = A.alloc<double>(tr.size());
        RVCX_HIP(hipMemcpyAsync(u.rep, tr.data(), tr.size() * sizeof(double), hipMemcpyHostToDevice, sf));
      }
    }
  }
  // two front sets (micro-batch k+1's front end runs while k is in the synthesizer)
  struct Front {
    void* wav = nullptr;
    double *ext = nullptr, *a64 = nullptr, *asum = nullptr;
    float *a32 = nullptr, *apad = nullptr, *f0 = nullptr;
    int* coarse = nullptr;
    long* cuts = nullptr;
    float* feats = nullptr;     // HuBERT features of every chunk of the micro-batch, group after group
    size_t feats_cap = 0;
    int* ns = nullptr;          // device: the members' sample counts (ragged micro-batches)
  } fr[2];
  size_t front_items = 0;
  for (const auto& mb : mbs) front_items = std::max(front_items, (size_t)mb.count);
  {
    size_t wmax = 0, emax = 0, nmax = 0, pmax = 0, fmax = 0, cmax = 0, hmax = 0;
    for (const auto& mb : mbs) {
      const long n = mb.nd;
      const size_t k = (size_t)mb.count;
      wmax = std::max(wmax, k * (size_t)n * 8);
      emax = std::max(emax, k * highpass_ext_doubles(n));
      nmax = std::max(nmax, k * (size_t)n);
      pmax = std::max(pmax, k * (size_t)(n + 2 * g.t_pad));
      fmax = std::max(fmax, k * (size_t)((n + 2 * g.t_pad) / 160 + 8));
      cmax = std::max(cmax, k * (size_t)(cut_count(n, g) + 1));
      // every chunk carries 2 * t_pad samples of context beyond its share of the clip
      hmax = std::max(hmax, k * (size_t)E * (size_t)((n + 2 * g.t_pad + (cut_count(n, g) + 1) * (2 * g.t_pad + 640)) / 320 + 8));
    }
    for (auto& f : fr) {
      f.wav = A.alloc<char>(wmax);
      f.ext = A.alloc<double>(emax);    // also the |window sum| scratch of the cut search (same size class)
      f.a64 = A.alloc<double>(nmax);
      f.a32 = A.alloc<float>(nmax);
      f.apad = A.alloc<float>(pmax);
      f.f0 = A.alloc<float>(fmax);
      f.coarse = A.alloc<int>(fmax);
      f.cuts = A.alloc<long>(cmax);
      f.feats = A.alloc<float>(hmax);
      f.feats_cap = hmax;
      f.ns = A.alloc<int>(front_items);
    }
  }
  // resample_sr (pipeline.py:453-454): librosa.resample(audio_opt, orig_sr=tgt_sr, target_sr=resample_sr) ahead of the peak
  // normalisation -- hard-wired off by rvc_infer (infer.py:144), reachable through VC.pipeline
  const bool resamp = p.resample_sr >= 16000 && p.resample_sr != M.cfg.sr;
  ResampleFilter rs_filter;
  if (resamp) rs_filter = make_resample_filter(A, M.cfg.sr, p.resample_sr, s);
  const size_t work_mark = A.mark();

  float t_hp = 0, t_f0 = 0, t_hub = 0, t_idx = 0, t_syn[3] = {0, 0, 0}, t_post = 0, t_wait_hub = 0, t_wait_f0 = 0;
  struct Span {
    int a, b;
    float* acc;
  };
  std::vector<Span> spans;
  // RVCX_FRONT_DELAY (experiments): 0 (default) = the next micro-batch's front end starts as soon as its buffers are
  // free, 1 = when the main stream reaches this micro-batch's synthesizer, 2 = when it reaches the NSF decoder.
  // Measured on C3 (64 x 30 s): 1036 / 1030 / 1031 x -- at B = 8 HuBERT and the F0 model are throughput-bound like the
  // decoder, so WHERE they overlap the main stream does not matter: the sum of the kernel times is what it is.
  static const int front_delay = getenv("RVCX_FRONT_DELAY") ? atoi(getenv("RVCX_FRONT_DELAY")) : 0;
  // ---- front end of one micro-batch on `sf`: upload, high-pass, cut search, reflect pad.  Returns after the
  // (rare) cut-point read-back, so the chunk plan is known to the host.
  auto front = [&](int k) {
    const MB& mb = mbs[k];
    Front& f = fr[k & 1];
    const long n = mb.nd, n_pad = n + 2 * g.t_pad;      // the micro-batch's geometry; a member holds ios[].n <= n samples
    const int Bm = mb.count;
    if (sf != s && k >= 2) RVCX_HIP(hipStreamWaitEvent(sf, c.ev_done[k & 1], 0));   // set k&1 was micro-batch k-2's
    if (sf != s && k >= 1 && front_delay) RVCX_HIP(hipStreamWaitEvent(sf, c.ev_syn[(k - 1) & 1], 0));
    const int h0 = clk.mark(sf);        // behind the waits: the span is the front end's own work, not its queueing
    const bool f64 = ios[order[mb.first]].wav64 != nullptr;
    for (int b = 0; b < Bm; ++b) {
      const UttIO& io = ios[order[mb.first + b]];
      RVCX_CHECK((io.wav64 != nullptr) == f64, "convert: float32 and float64 inputs mixed in one call");
      if (f64) RVCX_HIP(hipMemcpyAsync(static_cast<double*>(f.wav) + (size_t)b * n, io.wav64, (size_t)io.n * 8, hipMemcpyDefault, sf));
      else RVCX_HIP(hipMemcpyAsync(static_cast<float*>(f.wav) + (size_t)b * n, io.wav, (size_t)io.n * 4, hipMemcpyDefault, sf));
      Utt& u = utts[order[mb.first + b]];
      if (io.noise) RVCX_HIP(hipMemcpyAsync(const_cast<float*>(u.noise), io.noise, (size_t)u.noise_cap * 4, hipMemcpyDefault, sf));
    }
    const int* d_ns = nullptr;
    if (mb.ragged) {
      std::vector<int> nsv(Bm);
      for (int b = 0; b < Bm; ++b) nsv[b] = (int)ios[order[mb.first + b]].n;
      set_dev_ints(f.ns, nsv.data(), Bm, sf);
      d_ns = f.ns;
    }
    launch_highpass(f64 ? nullptr : static_cast<const float*>(f.wav), f64 ? static_cast<const double*>(f.wav) : nullptr,
                    f.ext, f.a64, f.a32, n, sf, Bm, n, d_ns);
    // chunk cut points (pipeline.py:330-344): only clips longer than t_max are cut
    const long ncut = cut_count(ios[order[mb.first]].n, g);     // cut clips are never ragged: n is their own length
    RVCX_CHECK(ncut == 0 || !mb.ragged, "internal: a cut clip in a ragged micro-batch");
    std::vector<long> cuts((size_t)Bm * ncut);
    if (ncut > 0) {
      for (int b = 0; b < Bm; ++b) {
        double* asum = f.ext + (size_t)b * highpass_ext_doubles(n);   // the filter scratch is free again
        hipLaunchKernelGGL(window_abs_sum_kernel, dim3((unsigned)std::min<long>(cdiv64(n, 256), 65535)), dim3(256), 0,
                           sf, f.a64 + (size_t)b * n, asum, n);
        long i = 0;
        for (long t = g.t_center; t < n; t += g.t_center, ++i) {
          const long lo = std::max<long>(0, t - g.t_query), hi = std::min<long>(n, t + g.t_query);
          hipLaunchKernelGGL(argmin_first_kernel, dim3(1), dim3(256), 0, sf, asum, lo, hi, f.cuts + (size_t)b * ncut + i);
        }
        RVCX_CHECK(i == ncut, "internal: cut count");
      }
      RVCX_HIP(hipMemcpyAsync(cuts.data(), f.cuts, cuts.size() * sizeof(long), hipMemcpyDeviceToHost, sf));
    }
    launch_reflect_pad(f.a32, f.apad, Bm, (int)n, (int)g.t_pad, n_pad, sf, d_ns);   // zeros behind a shorter member
    if (sf != s) RVCX_HIP(hipEventRecord(c.ev_front[k & 1], sf));
    spans.push_back({h0, clk.mark(sf), &t_hp});
    if (ncut > 0) RVCX_HIP(hipStreamSynchronize(sf));
    const long stride = n_pad / 160 + 8;
    for (int b = 0; b < Bm; ++b) {
      Utt& u = utts[order[mb.first + b]];
      u.a64 = f.a64 + (size_t)b * n;
      u.apad = f.apad + (size_t)b * n_pad;
      u.coarse = f.coarse + (size_t)b * stride;
      u.f0 = f.f0 + (size_t)b * stride;
      std::vector<long> opt(cuts.begin() + (size_t)b * ncut, cuts.begin() + (size_t)(b + 1) * ncut);
      u.plan = plan_chunks(u.n, opt, g);
    }
  };

  // ---- F0 of one micro-batch on `sf` out of its own arena (pipeline.py:362-380: once per utterance)
  std::vector<int> f0_ev0(mbs.size(), -1), f0_ev1(mbs.size(), -1);
  auto enqueue_f0 = [&](int k, const std::function<void()>* mid) {
    const MB& mb = mbs[k];
    Front& f = fr[k & 1];
    const long n_pad = mb.nd + 2 * g.t_pad;
    f0_ev0[k] = clk.mark(sf);
    c.arena_f0.reset();
    c.arena.swap(c.arena_f0);
    try {
      std::vector<F0Extra> fx(mb.count);
      for (int b = 0; b < mb.count; ++b) {
        const UttIO& io = ios[order[mb.first + b]];
        fx[b].dither = io.crepe_dither;
        fx[b].dither_n = io.crepe_dither_n;
        fx[b].seed_offset = io.seed_offset;
      }
      std::vector<int> nsp(mb.count);
      for (int b = 0; b < mb.count; ++b) nsp[b] = (int)utts[order[mb.first + b]].n_pad;
      get_f0_device(c, f.apad, n_pad, p, f.coarse, f.f0, sf, mb.count, n_pad / 160 + 8, mid, fx.data(),
                    mb.ragged ? nsp.data() : nullptr);
    } catch (...) {
      c.arena.swap(c.arena_f0);
      throw;
    }
    c.arena.swap(c.arena_f0);
    for (int b = 0; b < mb.count; ++b) {                 // f0 file override (pipeline.py:185-191): x_pad * 100 frames in
      const Utt& u = utts[order[mb.first + b]];
      const long stride = n_pad / 160 + 8;
      if (u.rep)
        launch_f0_override(u.rep, u.rep_n, (int)(g.t_pad / 160), f.f0 + (size_t)b * stride, f.coarse + (size_t)b * stride,
                           (int)(u.n_pad / 160), p.f0_min, p.f0_max, sf);
    }
    f0_ev1[k] = clk.mark(sf);
    if (sf != s) RVCX_HIP(hipEventRecord(c.ev_join, sf));
  };

  const int e_begin = clk.mark(s);
  std::vector<std::array<hipEvent_t, 4>> syn_ev;   // synth_forward's own stage events (resolved after the final sync)

  front(0);
  const bool use_protect = p.protect < 0.5f;
  const bool use_index = c.index && p.index_rate != 0.f;
  // HuBERT's stream: the decoder's first auxiliary stream, idle while the front end runs (no fifth stream: api.hip).
  // RVCX_HUBERT_ON=main: the main stream
  static const bool hub_on_main = getenv("RVCX_HUBERT_ON") && std::string(getenv("RVCX_HUBERT_ON")) == "main";
  static const bool hub_on_aux1 = getenv("RVCX_HUBERT_ON") && std::string(getenv("RVCX_HUBERT_ON")) == "aux1";
  hipStream_t sh = (c.serial || hub_on_main) ? s : (hub_on_aux1 ? c.aux[1] : c.aux[0]);

  // ---- chunk jobs of a micro-batch, grouped by chunk length (order of first appearance); group gi's HuBERT
  // features live at fr[k & 1].feats + feats_off[gi]
  struct Grp {
    std::vector<int> jobs;
    long ns = 0;        // launch geometry of the group: chunk samples, HuBERT frames, synthesizer frames ...
    int Th = 0, T = 0;
    bool ragged = false;   // ... which a member's own (Job::e - s, Th, T) may fall short of
  };
  struct Plan {
    std::vector<Job> jobs;
    std::vector<Grp> groups;
    std::vector<size_t> feats_off;
  } plans[2];
  auto plan_jobs = [&](int k) {
    const MB& mb = mbs[k];
    Plan& P = plans[k & 1];
    P.jobs.clear();
    P.groups.clear();
    P.feats_off.clear();
    for (int b = 0; b < mb.count; ++b) {
      Utt& u = utts[order[mb.first + b]];
      long out_off = 0, noise_off = 0;
      for (size_t ci = 0; ci < u.plan.size(); ++ci) {
        const Chunk& ch = u.plan[ci];
        const long ns = ch.e - ch.s;
        const int Th = hubert_frames(*c.hubert, ns);
        const int T = (int)std::min<long>(ns / 160, 2L * Th);
        RVCX_CHECK(Th > 0 && T > 0, "convert: chunk too short");
        // a cut may land within x_pad seconds of the clip's end: the reference then trims that chunk's output to
        // nothing (audio1[t_pad_tgt:-t_pad_tgt], pipeline.py:441-447) but still runs it -- so do we
        RVCX_CHECK((long)T * M.upp >= 2 * g.t_pad_tgt,
                   "convert: the clip is shorter than what the trim of the x_pad padding removes (the reference's "
                   "audio1[t_pad_tgt:-t_pad_tgt] is empty there and np.abs(audio_opt).max() raises)");
        P.jobs.push_back({b, (int)ci, ch.s, ch.e, ch.f0_off, out_off, noise_off, Th, T});
        out_off += (long)T * M.upp - 2 * g.t_pad_tgt;
        noise_off += (long)T * (inter + M.upp);
      }
      RVCX_CHECK(out_off > 0,
                 "convert: nothing is left of the clip after the trim of the x_pad padding (the reference's "
                 "np.abs(audio_opt).max() raises on the empty array; at x_pad = 1 a clip needs 400 samples)");
      RVCX_CHECK(out_off <= u.cap, "convert: output capacity exceeded");
      RVCX_CHECK(!u.noise || noise_off <= u.noise_cap, "convert: parity noise buffer shorter than the chunk plan needs");
      u.out_n = out_off;
    }
    if (mb.ragged) {
      // one job per member (uncut clips), one group with the class's geometry
      Grp G;
      G.ns = mb.nd + 2 * g.t_pad;
      G.Th = hubert_frames(*c.hubert, G.ns);
      G.T = (int)std::min<long>(G.ns / 160, 2L * G.Th);
      G.ragged = true;
      RVCX_CHECK((int)P.jobs.size() == mb.count, "internal: a ragged micro-batch holds uncut clips only");
      for (int j = 0; j < (int)P.jobs.size(); ++j) G.jobs.push_back(j);
      P.groups.push_back(std::move(G));
    } else {
      for (int j = 0; j < (int)P.jobs.size(); ++j) {
        size_t gi = 0;
        for (; gi < P.groups.size(); ++gi)
          if (P.groups[gi].ns == P.jobs[j].e - P.jobs[j].s) break;
        if (gi == P.groups.size()) {
          P.groups.emplace_back();
          P.groups[gi].ns = P.jobs[j].e - P.jobs[j].s;
          P.groups[gi].Th = P.jobs[j].Th;
          P.groups[gi].T = P.jobs[j].T;
        }
        P.groups[gi].jobs.push_back(j);
      }
    }
    size_t off = 0;
    for (const auto& grp : P.groups) {
      P.feats_off.push_back(off);
      off += (size_t)grp.jobs.size() * E * grp.Th;
    }
    RVCX_CHECK(off <= fr[k & 1].feats_cap, "internal: HuBERT feature buffer smaller than the chunk plan");
  };

  // ---- HuBERT of one micro-batch out of its own arena, on aux[0] -- the stream the decoder's second ResBlock branch uses.
  // It depends on the micro-batch's front end only, but it is ENQUEUED behind synth_forward(k - 1): in a batched call
  // HuBERT k queues behind decoder k - 1's last aux[0] branch, so the HuBERT / decoder overlap of rounds 2-3 (a stream of
  // its own) is gone -- traded, with a measured net gain, for not paying the fifth stream's shared hardware queue (api.hip;
  // DESIGN "Four streams, not five").  What still overlaps: HuBERT k beside the F0 model of k (front stream).
  std::vector<int> hub_ev0(mbs.size(), -1), hub_ev1(mbs.size(), -1);
  auto enqueue_hubert = [&](int k) {
    const MB& mb = mbs[k];
    Plan& P = plans[k & 1];
    Front& f = fr[k & 1];
    if (sh != sf) RVCX_HIP(hipStreamWaitEvent(sh, c.ev_front[k & 1], 0));   // apad of this micro-batch is ready
    hub_ev0[k] = clk.mark(sh);
    c.arena_hub.reset();
    for (size_t gi = 0; gi < P.groups.size(); ++gi) {
      const auto& grp = P.groups[gi].jobs;
      const int G = (int)grp.size();
      const Job& j0 = P.jobs[grp[0]];
      const long ns = P.groups[gi].ns;
      std::vector<int> nsv(G);
      for (int q = 0; q < G; ++q) nsv[q] = (int)(P.jobs[grp[q]].e - P.jobs[grp[q]].s);
      c.arena.swap(c.arena_hub);       // allocations below come from HuBERT's arena (stream-ordered on `sh`)
      try {
        const size_t mk = c.arena.mark();
        // chunk audio (G, ns): strided view of apad when the items are consecutive utterances cut at the same place
        const float* wav = utts[order[mb.first + j0.u]].apad + j0.s;
        long wav_bs = ns;
        bool uniform = true;
        for (int q = 1; q < G; ++q) uniform &= P.jobs[grp[q]].s == j0.s && P.jobs[grp[q]].u == j0.u + q;
        if (P.groups[gi].ragged) {
          wav_bs = mb.nd + 2 * g.t_pad;      // the members' rows of the front set's apad (zeros behind each member)
        } else if (G > 1 && uniform) {
          wav_bs = utts[order[mb.first + j0.u]].n_pad;
        } else if (G > 1) {
          float* wg = c.arena.alloc<float>((size_t)G * ns);
          for (int q = 0; q < G; ++q)
            RVCX_HIP(hipMemcpyAsync(wg + (size_t)q * ns, utts[order[mb.first + P.jobs[grp[q]].u]].apad + P.jobs[grp[q]].s,
                                    (size_t)ns * 4, hipMemcpyDeviceToDevice, sh));
          wav = wg;
        }
        hubert_features_for(c, *c.hubert, E, G, wav, ns, f.feats + P.feats_off[gi], sh, wav_bs,
                            P.groups[gi].ragged ? nsv.data() : nullptr);
        c.arena.reset(mk);
      } catch (...) {
        c.arena.swap(c.arena_hub);
        throw;
      }
      c.arena.swap(c.arena_hub);
    }
    hub_ev1[k] = clk.mark(sh);
    if (sh != s) RVCX_HIP(hipEventRecord(c.ev_hubdone[k & 1], sh));
  };
  // One host thread feeds both branches of the front end, and the GPU runs them in this order: the F0 model's mel + U-Net
  // first (throughput work, 3.7 ms of a single clip), then its recurrence (3.3 ms on 16 CUs) with ALL of HuBERT beside it.
  // HuBERT is enqueued from the F0 model's hook behind the U-Net (rmvpe.hip: RVCX_HUBERT_AFTER) and its stream waits there on the
  // GPU too (RVCX_HUBERT_GATE=0: no wait).  Round 4, once the two streams really ran side by side (api.hip: the fifth
  // stream): starting HuBERT beside the shallow U-Net levels (rounds 2-3) slows the U-Net by more than HuBERT gains --
  // C2 on one box, HuBERT enqueued behind level 2 / 3 / 4 / the intermediate layers / the whole U-Net: 1041-1055 /
  // 1039-1050 / 1044-1064 / 1058-1072 / 1078-1081x; batched calls do not care (C3 1290 / 1295x with / without the wait).
  // Other F0 methods have no such hook: HuBERT is enqueued behind them and does not wait.
  int mid_mark = -1;
  // Round 6: the wait is for front ends that run EXPOSED and latency-bound -- a single utterance, or the first micro-batch
  // of a call when it is small (B = 2 / 3 / 4 in one call: 1360 / 1433 / 1474x with the wait, 1311 / 1408 / 1453 without;
  // B = 8: level).  A front end that runs beside the previous micro-batch's decoder is a throughput-type neighbour and HuBERT
  // beside its U-Net is free: C3 1468 - 1471 -> 1486 - 1491x without the wait (tools/sweep_c3_knobs.sh,
  // tools/sweep_gate_small_batches.sh).  RVCX_HUBERT_GATE: 0 never, 1 always, unset = this rule.
  static const int hub_gate_env = getenv("RVCX_HUBERT_GATE") ? atoi(getenv("RVCX_HUBERT_GATE")) : -1;
  auto enqueue_models = [&](int k) {
    bool hub_done = false, from_hook = true;
    const std::function<void()> mid = [&]() {
      if (hub_done) return;
      hub_done = true;
      if (k == 0) mid_mark = clk.mark(sf);
      const bool hub_gate = hub_gate_env < 0 ? (mbs[k].count <= 1 || (k == 0 && mbs[k].count < 8)) : hub_gate_env != 0;
      if (hub_gate && from_hook && sh != sf) {
        RVCX_HIP(hipEventRecord(c.ev_hub, sf));
        RVCX_HIP(hipStreamWaitEvent(sh, c.ev_hub, 0));
      }
      enqueue_hubert(k);
    };
    enqueue_f0(k, &mid);
    from_hook = false;
    if (!hub_done) mid();
  };

  static const bool host_trace = getenv("RVCX_HOST_TRACE") != nullptr;   // where does the enqueueing thread spend its time?
  const auto ht0 = std::chrono::steady_clock::now();
  auto ht = [&](const char* what, int k) {
    if (host_trace)
      fprintf(stderr, "[host] mb %d %-14s %9.3f ms\n", k, what,
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ht0).count());
  };
  plan_jobs(0);
  enqueue_models(0);
  ht("hubert+f0", 0);
  for (int k = 0; k < (int)mbs.size(); ++k) {
    const MB& mb = mbs[k];
    const int Bm = mb.count;
    Plan& P = plans[k & 1];
    const std::vector<Job>& jobs = P.jobs;
    A.reset(work_mark);
    const int w0 = clk.mark(s);
    if (sf != s) RVCX_HIP(hipStreamWaitEvent(s, c.ev_front[k & 1], 0));   // a64 (envelope) of this micro-batch is ready
    if (sh != s) RVCX_HIP(hipStreamWaitEvent(s, c.ev_hubdone[k & 1], 0)); // and its HuBERT features
    const int w1 = clk.mark(s);
    spans.push_back({w0, w1, &t_wait_hub});
    bool joined = false;
    for (size_t gi = 0; gi < P.groups.size(); ++gi) {
      const auto& grp = P.groups[gi].jobs;
      const bool ragged = P.groups[gi].ragged;
      const size_t mk = A.mark();
      const int G = (int)grp.size();
      const int Th = P.groups[gi].Th, T = P.groups[gi].T;      // the group's geometry; a member's own: Job::Th, Job::T
      const size_t nz = (size_t)inter * T, nsrc = (size_t)T * M.upp;
      const int h1 = clk.mark(s);
      float* feats = fr[k & 1].feats + P.feats_off[gi];
      const float* feats0 = feats;
      if (use_index) {
        if (use_protect) {
          float* keep = A.alloc<float>((size_t)G * E * Th);
          RVCX_HIP(hipMemcpyAsync(keep, feats, (size_t)G * E * Th * sizeof(float), hipMemcpyDeviceToDevice, s));
          feats0 = keep;
        }
        for (int q = 0; q < G; ++q) {
          const size_t mk2 = A.mark();
          const int Thq = jobs[grp[q]].Th;
          float* fq = feats + (size_t)q * E * Th;
          if (Thq == Th) {
            index_blend(c, *c.index, fq, Th, p.index_rate, nullptr, nullptr, s);
          } else {       // a shorter member of a ragged group: searched as the (E, Thq) matrix of its single run
            float* cq = A.alloc<float>((size_t)E * Thq);
            RVCX_HIP(hipMemcpy2DAsync(cq, (size_t)Thq * 4, fq, (size_t)Th * 4, (size_t)Thq * 4, E, hipMemcpyDeviceToDevice, s));
            index_blend(c, *c.index, cq, Thq, p.index_rate, nullptr, nullptr, s);
            RVCX_HIP(hipMemcpy2DAsync(fq, (size_t)Th * 4, cq, (size_t)Thq * 4, (size_t)Thq * 4, E, hipMemcpyDeviceToDevice, s));
          }
          A.reset(mk2);
        }
      }
      const int h2 = clk.mark(s);
      spans.push_back({h1, h2, &t_idx});
      ht("index", k);
      if (!joined) {   // first consumer of f0 / coarse
        const int j0m = clk.mark(s);
        if (sf != s) RVCX_HIP(hipStreamWaitEvent(s, c.ev_join, 0));
        spans.push_back({j0m, clk.mark(s), &t_wait_f0});
        joined = true;
      }
      float* phone = A.alloc<float>((size_t)G * E * T);
      int* pitch = A.alloc<int>((size_t)G * T);
      float* pitchf = A.alloc<float>((size_t)G * T);
      float* zn = A.alloc<float>((size_t)G * nz);
      float* sn = A.alloc<float>((size_t)G * nsrc);
      std::vector<int> sids(G, p.sid), lens_q(G);
      if (ragged) {      // frames behind a shorter member: zero phone / pitch rows (masked in the networks anyway)
        RVCX_HIP(hipMemsetAsync(phone, 0, (size_t)G * E * T * sizeof(float), s));
        RVCX_HIP(hipMemsetAsync(pitch, 0, (size_t)G * T * sizeof(int), s));
        RVCX_HIP(hipMemsetAsync(pitchf, 0, (size_t)G * T * sizeof(float), s));
      }
      for (int q = 0; q < G; ++q) {
        const Job& j = jobs[grp[q]];
        const Utt& u = utts[order[mb.first + j.u]];
        const int Tq = j.T;
        lens_q[q] = Tq;
        launch_upsample_protect(feats + (size_t)q * E * Th, feats0 + (size_t)q * E * Th, u.f0 + j.f0_off,
                                phone + (size_t)q * E * T, E, j.Th, Tq, p.protect, use_protect ? 1 : 0, s, Th, T);
        RVCX_HIP(hipMemcpyAsync(pitch + (size_t)q * T, u.coarse + j.f0_off, (size_t)Tq * 4, hipMemcpyDeviceToDevice, s));
        RVCX_HIP(hipMemcpyAsync(pitchf + (size_t)q * T, u.f0 + j.f0_off, (size_t)Tq * 4, hipMemcpyDeviceToDevice, s));
        // noise: parity = packed [z (inter*T) | src (T*upp)] per chunk in draw order (the member's own T); else Philox
        // with the utterance's own stream (seed + position in the call) so that a batch item equals its single run
        if (u.noise) {
          const size_t nzq = (size_t)inter * Tq, nsq = (size_t)Tq * M.upp;
          RVCX_HIP(hipMemcpy2DAsync(zn + (size_t)q * nz, (size_t)T * 4, u.noise + j.noise_off, (size_t)Tq * 4, (size_t)Tq * 4,
                                    inter, hipMemcpyDeviceToDevice, s));
          RVCX_HIP(hipMemcpyAsync(sn + (size_t)q * nsrc, u.noise + j.noise_off + nzq, nsq * 4, hipMemcpyDeviceToDevice, s));
        } else {
          const uint64_t seed = p.seed + (uint64_t)ios[u.io].seed_offset;
          const uint64_t off = (uint64_t)j.ci << 36;
          launch_randn(zn + (size_t)q * nz, nz, seed, off, s);
          launch_randn(sn + (size_t)q * nsrc, nsrc, seed, off + ((uint64_t)1 << 35), s);
        }
      }
      float* wavout = A.alloc<float>((size_t)G * nsrc);
      if (gi == 0 && sf != s && front_delay == 1) RVCX_HIP(hipEventRecord(c.ev_syn[k & 1], s));
      SynthIO io;
      if (gi == 0 && sf != s && front_delay == 2) io.ev_decoder = c.ev_syn[k & 1];
      io.B = G;
      io.T = T;
      io.lens_host = ragged ? lens_q.data() : nullptr;
      {
        // Round 6: only the part of each decoder call that survives the trim below (audio1[t_pad_tgt:-t_pad_tgt],
        // pipeline.py:432-447) is evaluated, plus the decoder's receptive field around it (synth.hip, SynthIO::dec_skip; a
        // multiple of four frames so that every shifted pointer stays 16-byte aligned).  RVCX_DEC_WINDOW=0: everything.
        static const bool window_on = !getenv("RVCX_DEC_WINDOW") || atoi(getenv("RVCX_DEC_WINDOW")) != 0;
        const int pad_frames = (int)(g.t_pad_tgt / M.upp);
        io.dec_skip = window_on ? std::max(0, pad_frames - M.dec_rf_frames) & ~3 : 0;
      }
      io.phone_ct = phone;
      io.pitch = pitch;
      io.pitchf = pitchf;
      io.sid_host = sids.data();
      io.z_noise = zn;
      io.src_noise = sn;
      io.out = wavout;
      if (stage_ms) {
        syn_ev.emplace_back();
        for (auto& e : syn_ev.back()) RVCX_HIP(hipEventCreate(&e));
      }
      ht("pre-synth", k);
      synth_forward(c, M, io, stage_ms ? syn_ev.back().data() : nullptr);
      ht("synth", k);
      for (int q = 0; q < G; ++q) {
        const Job& j = jobs[grp[q]];
        const Utt& u = utts[order[mb.first + j.u]];
        const long keep_n = (long)j.T * M.upp - 2 * g.t_pad_tgt;
        if (keep_n > 0)
          RVCX_HIP(hipMemcpyAsync(u.outf + j.out_off, wavout + (size_t)q * nsrc + g.t_pad_tgt,
                                  (size_t)keep_n * sizeof(float), hipMemcpyDeviceToDevice, s));
      }
      A.reset(mk);   // stream-ordered reuse: later launches on the same stream see the finished group
    }
    // ---- RMS envelope, peak normalise, int16 (pipeline.py:449-461), then the PCM leaves on the copy stream
    const int e3 = clk.mark(s);
    for (int b = 0; b < Bm; ++b) {
      Utt& u = utts[order[mb.first + b]];
      const size_t mk = A.mark();
      if (p.volume_envelope != 1.f) {
        const int tgt = M.cfg.sr;
        const int f1 = 16000 / 2 * 2, h1 = 16000 / 2, f2 = tgt / 2 * 2, h2 = tgt / 2;
        const int n1 = (int)(1 + (u.n + 2 * (f1 / 2) - f1) / h1), n2 = (int)(1 + (u.out_n + 2 * (f2 / 2) - f2) / h2);
        float* r1 = A.alloc<float>((size_t)n1);
        float* r2 = A.alloc<float>((size_t)n2);
        hipLaunchKernelGGL(frame_rms_kernel<double>, dim3(n1), dim3(256), 0, s, u.a64, r1, u.n, f1, h1, n1);
        hipLaunchKernelGGL(frame_rms_kernel<float>, dim3(n2), dim3(256), 0, s, u.outf, r2, u.out_n, f2, h2, n2);
        hipLaunchKernelGGL(envelope_kernel, dim3((unsigned)std::min<long>(cdiv64(u.out_n, 256), 65535)), dim3(256), 0, s,
                           u.outf, r1, n1, r2, n2, u.out_n, p.volume_envelope);
      }
      if (resamp) {
        const long nr = resample_out_len(u.out_n, M.cfg.sr, p.resample_sr);
        RVCX_CHECK(nr <= u.cap, "convert: resampled output exceeds the capacity");
        float* r = A.alloc<float>((size_t)std::max<long>(nr, 1));
        launch_resample_f32(rs_filter, u.outf, u.out_n, r, nr, s);
        RVCX_HIP(hipMemcpyAsync(u.outf, r, (size_t)nr * sizeof(float), hipMemcpyDeviceToDevice, s));
        u.out_n = nr;
      }
      unsigned int* amax = A.alloc<unsigned int>(1);
      RVCX_HIP(hipMemsetAsync(amax, 0, sizeof(unsigned int), s));
      hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)std::min<long>(cdiv64(u.out_n, 256), 512)), dim3(256), 0, s, u.outf,
                         u.out_n, amax);
      hipLaunchKernelGGL(to_int16_kernel, dim3((unsigned)std::min<long>(cdiv64(u.out_n, 256), 65535)), dim3(256), 0, s,
                         u.outf, u.pcm, u.out_n, amax);
      (void)mk;   // amax / rms buffers stay allocated until the micro-batch's work area is reset
    }
    RVCX_HIP(hipGetLastError());
    const int e4 = clk.mark(s);
    spans.push_back({e3, e4, &t_post});
    if (sio != s) {
      RVCX_HIP(hipEventRecord(c.ev_io, s));
      RVCX_HIP(hipStreamWaitEvent(sio, c.ev_io, 0));
    }
    for (int b = 0; b < Bm; ++b) {
      Utt& u = utts[order[mb.first + b]];
      UttIO& io = ios[u.io];
      RVCX_HIP(hipMemcpyAsync(io.out, u.pcm, (size_t)u.out_n * sizeof(short), hipMemcpyDefault, sio));
      if (io.out_f32) RVCX_HIP(hipMemcpyAsync(io.out_f32, u.outf, (size_t)u.out_n * sizeof(float), hipMemcpyDefault, sio));
      io.out_n = u.out_n;
    }
    if (sf != s) RVCX_HIP(hipEventRecord(c.ev_done[k & 1], s));
    ht("post+d2h", k);
    // ---- front end of the next micro-batch (the main stream still has this one's synthesizer queued)
    if (k + 1 < (int)mbs.size()) {
      front(k + 1);
      ht("front", k + 1);
      plan_jobs(k + 1);
      enqueue_models(k + 1);
      ht("hubert+f0", k + 1);
    }
  }
  const int e_end = clk.mark(s);
  // the device error word travels with the call's own synchronisation (Ctx::snapshot_dev_err): every kernel that can set it
  // has been joined into the main stream by now (F0 and HuBERT streams through ev_join / ev_hubdone, the branch streams
  // at each stage's end)
  c.snapshot_dev_err(s);
  RVCX_HIP(hipStreamSynchronize(s));
  if (sf != s) RVCX_HIP(hipStreamSynchronize(sf));
  if (sio != s) RVCX_HIP(hipStreamSynchronize(sio));
  if (stage_ms) {
    for (const auto& sp : spans) *sp.acc += clk.between(sp.a, sp.b);
    for (size_t k = 0; k < mbs.size(); ++k) t_f0 += clk.between(f0_ev0[k], f0_ev1[k]);
    for (size_t k = 0; k < mbs.size(); ++k) t_hub += clk.between(hub_ev0[k], hub_ev1[k]);
    for (auto& ev : syn_ev) {
      for (int i = 0; i < 3; ++i) {
        float t = 0.f;
        RVCX_HIP(hipEventElapsedTime(&t, ev[i], ev[i + 1]));
        t_syn[i] += t;
      }
      for (auto& e : ev) (void)hipEventDestroy(e);
    }
    if (getenv("RVCX_HOST_TRACE") && !mbs.empty())     // device-side times of the first micro-batch's front end, from its first event
      fprintf(stderr, "[front] F0 model %.2f .. %.2f ms (shallow U-Net levels enqueued and done by %.2f), HuBERT %.2f .. %.2f ms\n",
              clk.between(0, f0_ev0[0]), clk.between(0, f0_ev1[0]), clk.between(0, mid_mark), clk.between(0, hub_ev0[0]),
              clk.between(0, hub_ev1[0]));
    if (getenv("RVCX_HOST_TRACE"))
      fprintf(stderr, "[wait] main stream waited %.1f ms for HuBERT / front sets and %.1f ms for F0\n", t_wait_hub, t_wait_f0);
    stage_ms[0] = t_hp;
    stage_ms[1] = t_f0;     // on the front stream, overlapped with HuBERT / the previous micro-batch's decoder
    stage_ms[2] = t_hub;    // on HuBERT's stream, beside the F0 model and (k >= 1) the previous micro-batch's decoder
    stage_ms[3] = t_idx;
    stage_ms[4] = t_syn[0];
    stage_ms[5] = t_syn[1];
    stage_ms[6] = t_syn[2];
    stage_ms[7] = t_post;
    stage_ms[8] = clk.between(e_begin, e_end);
  }
}

}  // namespace rvcx
