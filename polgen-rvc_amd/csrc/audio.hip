// Sample-rate conversion on the device: stands for the two librosa.resample calls of the path --
// load_audio (rvc/lib/my_utils.py:12-13: file rate -> 16 kHz, float64) and VC.pipeline's resample_sr branch
// (rvc/infer/pipeline.py:453-454: tgt_sr -> resample_sr on the float32 output).
//
// librosa.resample's default res_type is "soxr_hq" (librosa >= 0.10; the reference pins no version).  libsoxr is not
// vendored by the reference and its coefficients are not published as a formula, but its HQ recipe is (soxr.c,
// soxr_quality_spec: 20-bit precision): linear phase, pass-band flat to 0.9136 x Nyquist(out), stop-band from 1.0 x
// Nyquist(out) at -120.4 dB.  Round 6 -- the DEFAULT filter here ("kaiser_hq") is a Kaiser-windowed sinc DESIGNED TO THOSE
// TARGETS: cut-off at the middle of the transition band (roll-off 0.9568), beta 12.82 (125 dB), 96 zero crossings per wing
// (the Kaiser length for a 0.0864 x Nyquist transition), 2048 table samples per crossing with linear interpolation
// (table error -141 dB), every tap evaluated at its EXACT table position (p_i = (frac + i scale) 2048: no integer
// table step, so no constant gain error).  Measured (tests/test_resample_spec.py): +-0.001 dB to 7.31 kHz, below
// -127 dB from 8.0 kHz at 44.1 k / 48 k -> 16 k -- inside soxr_hq's published targets; the sample-level difference to
// any other filter meeting them is bounded by the ripple / stop-band figures outside the 7.31 ... 8 kHz transition band.
// resampy's published "kaiser_best" (librosa's default before 0.10: 64 zero crossings, 512 table samples, roll-off
// 0.9475937167399596, beta 14.769656459379492, integer table step int(scale * 512)) stays selectable (RVCX_RESAMPLER=
// kaiser_best, rvcx_resample_f64_kind) in its published arithmetic.  oracle/audio.py restates both in numpy.
//
// One lane per output sample, left wing then right wing; ~2 x 290 taps per sample at 48 -> 16 kHz.
// A 30 s stereo 48 kHz upload becomes 16 kHz mono in ~0.1 ms instead of ~1 s of host time -- at 1000x real time the
// conversion itself takes 30 ms, so a host-side resampler would be the whole request.
#include <cmath>
#include <cstdlib>
#include <vector>

#include "common.h"
#include "ops.h"

namespace rvcx {

namespace {

struct FilterSpec {
  int zeros, precision;
  double rolloff, beta;
  bool exact;          // taps at their exact table positions (kaiser_hq); false: resampy's integer table step
};
constexpr FilterSpec kSpecs[2] = {
    {96, 11, 0.9568, 12.82, true},                                   // 0: kaiser_hq -- designed to soxr_hq's published targets
    {64, 9, 0.9475937167399596, 14.769656459379492, false},          // 1: resampy's published kaiser_best
};

long double bessel_i0(long double x) {          // power series: converges in < 60 terms for x <= 15
  long double s = 1.0L, t = 1.0L;
  const long double q = x * x / 4.0L;
  for (int k = 1; k < 200; ++k) {
    t *= q / ((long double)k * k);
    s += t;
    if (t < s * 1e-22L) break;
  }
  return s;
}

// interp_win (right half of the filter) and interp_delta (its forward difference), optionally scaled by `gain`
void build_window(const FilterSpec& F, double gain, std::vector<double>& win, std::vector<double>& delta) {
  const int table = 1 << F.precision, n = F.zeros * table, nwin = n + 1;
  win.resize(nwin);
  delta.assign(nwin, 0.0);
  const long double i0b = bessel_i0((long double)F.beta);
  const double pi = 3.14159265358979323846;
  for (int i = 0; i <= n; ++i) {
    const double xz = (double)i / table;                          // np.linspace(0, num_zeros, n + 1)
    const double arg = F.rolloff * xz;
    const double sinc = arg == 0.0 ? 1.0 : std::sin(pi * arg) / (pi * arg);
    // np.kaiser(2 n + 1, beta)[n + i]: alpha = n, argument beta * sqrt(1 - (i / n)^2)
    const long double r = (long double)i / n;
    const long double taper = bessel_i0((long double)F.beta * sqrtl(std::max(0.0L, 1.0L - r * r))) / i0b;
    win[i] = (double)taper * (F.rolloff * sinc) * gain;
  }
  // exact tap positions: the last tap of a wing can lie within rounding of the window's edge, where a Kaiser window is not
  // zero (1e-8 of the peak) -- the final table sample is zero, so the weight runs out continuously (oracle/audio.py)
  if (F.exact) win[n] = 0.0;
  for (int i = 0; i < n; ++i) delta[i] = win[i + 1] - win[i];
}

template <typename TIn, typename TOut, typename TAcc>
__global__ void resample_kernel(const TIn* __restrict__ x, long n_orig, long x_stride, int channels, TOut* __restrict__ y,
                                long n_out, const ResampleFilter f) {
  const long t = blockIdx.x * 256L + threadIdx.x;
  if (t >= n_out) return;
  const double* __restrict__ win = f.win;
  const double* __restrict__ delta = f.delta;
  // `channels` > 1: the input is interleaved (frames, channels) and is averaged on the fly (librosa.to_mono)
  auto X = [&](long i) -> double {
    if (channels == 1) return (double)x[i * x_stride];
    double s = 0.0;
    for (int c = 0; c < channels; ++c) s += (double)x[i * x_stride + c];
    return s / channels;
  };
  const double time_register = (double)t * f.time_increment;
  const long n = (long)time_register;
  double frac = f.scale * (time_register - (double)n);
  TAcc acc = (TAcc)0;
  if (f.exact) {
    // tap i of the left wing sits at table position (frac + i scale) table: floor + linear interpolation per tap
    const double step = f.scale * f.table;
    double p0 = frac * f.table;
    for (long i = 0; i <= n; ++i) {
      const double p = p0 + (double)i * step;
      const long idx = (long)p;
      if (idx >= f.nwin - 1) break;
      const double w = win[idx] + (p - (double)idx) * delta[idx];
      acc = (TAcc)((double)acc + w * X(n - i));
    }
    p0 = (f.scale - frac) * f.table;
    for (long k = 0; n + k + 1 < n_orig; ++k) {
      const double p = p0 + (double)k * step;
      const long idx = (long)p;
      if (idx >= f.nwin - 1) break;
      const double w = win[idx] + (p - (double)idx) * delta[idx];
      acc = (TAcc)((double)acc + w * X(n + k + 1));
    }
    y[t] = (TOut)acc;
    return;
  }
  // resampy's published loop: a common fractional offset, an INTEGER table step
  const int index_step = f.index_step;
  double index_frac = frac * f.table;
  int offset = (int)index_frac;
  double eta = index_frac - offset;
  long i_max = min(n + 1, (long)((f.nwin - offset) / index_step));
  for (long i = 0; i < i_max; ++i) {
    const double w = win[offset + i * index_step] + eta * delta[offset + i * index_step];
    acc = (TAcc)((double)acc + w * X(n - i));
  }
  frac = f.scale - frac;
  index_frac = frac * f.table;
  offset = (int)index_frac;
  eta = index_frac - offset;
  const long k_max = min(n_orig - n - 1, (long)((f.nwin - offset) / index_step));
  for (long k = 0; k < k_max; ++k) {
    const double w = win[offset + k * index_step] + eta * delta[offset + k * index_step];
    acc = (TAcc)((double)acc + w * X(n + k + 1));
  }
  y[t] = (TOut)acc;
}

}  // namespace

long resample_out_len(long n, int sr_in, int sr_out) { return (long)((double)n * ((double)sr_out / (double)sr_in)); }

int resample_default_kind() {
  static const int k = [] {
    const char* e = getenv("RVCX_RESAMPLER");
    return (e && std::string(e) == "kaiser_best") ? 1 : 0;
  }();
  return k;
}

ResampleFilter make_resample_filter(Arena& A, int sr_in, int sr_out, hipStream_t s, int kind) {
  if (kind < 0) kind = resample_default_kind();
  RVCX_CHECK(kind == 0 || kind == 1, "resample: unknown filter kind");
  const FilterSpec& F = kSpecs[kind];
  ResampleFilter f;
  const double ratio = (double)sr_out / (double)sr_in;
  f.scale = std::min(1.0, ratio);
  f.time_increment = 1.0 / ratio;
  f.table = 1 << F.precision;
  f.nwin = F.zeros * f.table + 1;
  f.exact = F.exact ? 1 : 0;
  f.index_step = (int)(f.scale * f.table);
  RVCX_CHECK(f.index_step >= 1, "resample: ratio too small");
  std::vector<double> win, delta;
  build_window(F, ratio < 1.0 ? ratio : 1.0, win, delta);
  double* d = A.alloc<double>(2 * (size_t)f.nwin);
  RVCX_HIP(hipMemcpyAsync(d, win.data(), (size_t)f.nwin * sizeof(double), hipMemcpyHostToDevice, s));
  RVCX_HIP(hipMemcpyAsync(d + f.nwin, delta.data(), (size_t)f.nwin * sizeof(double), hipMemcpyHostToDevice, s));
  RVCX_HIP(hipStreamSynchronize(s));          // the host vectors die with this frame
  f.win = d;
  f.delta = d + f.nwin;
  return f;
}

void launch_resample_f64(const ResampleFilter& f, const double* x, long n, int channels, double* y, long n_out,
                         hipStream_t s) {
  if (n_out <= 0) return;
  hipLaunchKernelGGL((resample_kernel<double, double, double>), dim3((unsigned)cdiv64(n_out, 256)), dim3(256), 0, s, x, n,
                     (long)channels, channels, y, n_out, f);
}

void launch_resample_f32(const ResampleFilter& f, const float* x, long n, float* y, long n_out, hipStream_t s) {
  if (n_out <= 0) return;
  // float32 in -> float32 out with a float32 running sum: what the published loop does on a float32 array
  hipLaunchKernelGGL((resample_kernel<float, float, float>), dim3((unsigned)cdiv64(n_out, 256)), dim3(256), 0, s, x, n, 1L, 1,
                     y, n_out, f);
}

}  // namespace rvcx
