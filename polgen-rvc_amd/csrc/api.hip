// C ABI of librvcx.so (include/rvcx.h).  Nothing throws across this boundary.
#include "../../include/rvcx.h"

#include "ctx.h"
#include "layers.h"
#include "models.h"
#include "ops.h"

using namespace rvcx;

struct rvcx_ctx {
  Ctx c;
};

static thread_local std::string g_last_error;

#define API_BEGIN(ctxp)                       \
  Ctx* C = (ctxp) ? &(ctxp)->c : nullptr;     \
  try {                                       \
    if (!C) fail("null context");             \
    RVCX_HIP(hipSetDevice(C->device));

#define API_END                               \
    return 0;                                 \
  } catch (const std::exception& e) {         \
    g_last_error = e.what();                  \
    if (C) {                                  \
      C->last_error = e.what();               \
      C->arena.reset();                       \
    }                                         \
    (void)hipGetLastError();                  \
    return -1;                                \
  }

extern "C" {

const char* rvcx_version(void) { return "rvcx 0.1.0 (gfx950)"; }

int rvcx_create(int device, rvcx_ctx** out) {
  try {
    int n = 0;
    RVCX_HIP(hipGetDeviceCount(&n));
    if (n <= 0) fail("no HIP device visible: librvcx has no CPU fallback");
    if (device < 0 || device >= n) fail("device index out of range");
    RVCX_HIP(hipSetDevice(device));
    auto* h = new rvcx_ctx();
    h->c.device = device;
    RVCX_HIP(hipStreamCreateWithFlags(&h->c.stream, hipStreamNonBlocking));
    RVCX_HIP(hipStreamCreateWithFlags(&h->c.stream2, hipStreamNonBlocking));
    RVCX_HIP(hipEventCreateWithFlags(&h->c.ev_fork, hipEventDisableTiming));
    RVCX_HIP(hipEventCreateWithFlags(&h->c.ev_join, hipEventDisableTiming));
    conv_init();
    h->c.arena.reserve((size_t)256 << 20);
    *out = h;
    return 0;
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return -1;
  }
}

void rvcx_destroy(rvcx_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->c.device);
  (void)hipDeviceSynchronize();
  delete ctx;
}

const char* rvcx_last_error(rvcx_ctx* ctx) {
  return ctx ? ctx->c.last_error.c_str() : g_last_error.c_str();
}

void* rvcx_stream(rvcx_ctx* ctx) { return ctx ? (void*)ctx->c.stream : nullptr; }

double rvcx_flop_counter(rvcx_ctx* ctx, int reset) {
  if (!ctx) return 0.0;
  double f = ctx->c.flops;
  if (reset) ctx->c.flops = 0.0;
  return f;
}

// ------------------------------------------------------------------------------------------
// kernel-level entry points: host in, host out.  Weights are packed into the slab on each
// call (test-only path), activations live in the arena.
// ------------------------------------------------------------------------------------------
static float* to_dev(Ctx& c, const float* h, size_t n) {
  float* d = c.arena.alloc<float>(n);
  RVCX_HIP(hipMemcpyAsync(d, h, n * sizeof(float), hipMemcpyHostToDevice, c.stream));
  return d;
}
static int* to_dev_i(Ctx& c, const int32_t* h, size_t n) {
  if (!h) return nullptr;
  int* d = c.arena.alloc<int>(n);
  RVCX_HIP(hipMemcpyAsync(d, h, n * sizeof(int), hipMemcpyHostToDevice, c.stream));
  return d;
}
static void to_host(Ctx& c, float* h, const float* d, size_t n) {
  RVCX_HIP(hipMemcpyAsync(h, d, n * sizeof(float), hipMemcpyDeviceToHost, c.stream));
  RVCX_HIP(hipStreamSynchronize(c.stream));
}
static void ensure_slab(Ctx& c) { c.slab.init((size_t)3 << 30); }

int rvcx_op_conv1d(rvcx_ctx* ctx, const float* x, const float* w, const float* bias, const float* res,
                   float* y, int B, int Cin, int Tin, int Cout, int K, int stride, int dil,
                   int pad_left, int Tout, int groups, int pre_lrelu, float pre_slope, int act,
                   float act_slope, const int32_t* lens_in, const int32_t* lens_out) {
  API_BEGIN(ctx)
  ensure_slab(*C);
  size_t nx = (size_t)B * Cin * Tin, ny = (size_t)B * Cout * Tout;
  C->arena.reserve((nx + 2 * ny) * 4 + (64 << 20));
  C->arena.reset();
  ConvW L = make_conv(*C, w, bias, Cout, Cin / groups, K, groups);
  float* dx = to_dev(*C, x, nx);
  float* dy = C->arena.alloc<float>(ny);
  ConvArgs a = conv1d_args(L, dx, dy, B, Tin, Tout, stride, dil, pad_left);
  if (res) conv_set_res(a, to_dev(*C, res, ny), Cout, Tout);
  if (pre_lrelu) {
    a.pre_act = ACT_LRELU;
    a.pre_slope = pre_slope;
  }
  a.act = act;
  a.act_slope = act_slope;
  a.lens_in = to_dev_i(*C, lens_in, B);
  a.lens_out = to_dev_i(*C, lens_out, B);
  C->conv(a);
  to_host(*C, y, dy, ny);
  C->arena.reset();
  API_END
}

int rvcx_op_convtranspose1d(rvcx_ctx* ctx, const float* x, const float* w, const float* bias, float* y,
                            int B, int Cin, int Tin, int Cout, int K, int stride, int pad,
                            int pre_lrelu, float pre_slope) {
  API_BEGIN(ctx)
  ensure_slab(*C);
  const int Tout = (Tin - 1) * stride - 2 * pad + K;
  size_t nx = (size_t)B * Cin * Tin, ny = (size_t)B * Cout * Tout;
  C->arena.reserve((nx + ny) * 4 + (64 << 20));
  C->arena.reset();
  ConvT1dW L = make_convT1d(*C, w, bias, Cin, Cout, K, stride, pad);
  float* dx = to_dev(*C, x, nx);
  float* dy = C->arena.alloc<float>(ny);
  ConvArgs a = convT1d_args(L, dx, dy, B, Tin, Tout);
  if (pre_lrelu) {
    a.pre_act = ACT_LRELU;
    a.pre_slope = pre_slope;
  }
  C->conv(a);
  to_host(*C, y, dy, ny);
  C->arena.reset();
  API_END
}

// host helpers: dense (B,C,H,W) <-> row-padded (B,C,H,W+2)
static std::vector<float> pad_rows(const float* x, size_t planes, int H, int W) {
  const int Wp = W + 2;
  std::vector<float> o(planes * H * Wp, 0.f);
  for (size_t p = 0; p < planes; ++p)
    for (int h = 0; h < H; ++h)
      std::memcpy(&o[(p * H + h) * Wp + 1], &x[(p * H + h) * W], (size_t)W * 4);
  return o;
}
static void unpad_rows(const std::vector<float>& xp, float* y, size_t planes, int H, int W) {
  const int Wp = W + 2;
  for (size_t p = 0; p < planes; ++p)
    for (int h = 0; h < H; ++h)
      std::memcpy(&y[(p * H + h) * W], &xp[(p * H + h) * Wp + 1], (size_t)W * 4);
}

int rvcx_op_conv2d3x3(rvcx_ctx* ctx, const float* x, const float* w, const float* bias, const float* res,
                      float* y, int B, int Cin, int H, int W, int Cout, int act) {
  API_BEGIN(ctx)
  ensure_slab(*C);
  const int Wp = W + 2;
  size_t nx = (size_t)B * Cin * H * Wp, ny = (size_t)B * Cout * H * Wp;
  C->arena.reserve((nx + 2 * ny) * 4 + (64 << 20));
  C->arena.reset();
  ConvW L = make_conv(*C, w, bias, Cout, Cin, 9, 1);
  std::vector<float> xp = pad_rows(x, (size_t)B * Cin, H, W);
  float* dx = to_dev(*C, xp.data(), nx);
  float* dy = C->arena.alloc<float>(ny);
  ConvArgs a = conv2d_args(L, dx, dy, B, H, Wp);
  std::vector<float> rp;
  if (res) {
    rp = pad_rows(res, (size_t)B * Cout, H, W);
    conv_set_res(a, to_dev(*C, rp.data(), ny), Cout, H * Wp);
  }
  a.act = act;
  C->conv(a);
  std::vector<float> yp(ny);
  to_host(*C, yp.data(), dy, ny);
  // the kernel must keep the pad columns at exactly zero
  for (size_t r = 0; r < (size_t)B * Cout * H; ++r)
    if (yp[r * Wp] != 0.f || yp[r * Wp + Wp - 1] != 0.f) fail("conv2d: pad column not zero");
  unpad_rows(yp, y, (size_t)B * Cout, H, W);
  C->arena.reset();
  API_END
}

int rvcx_op_convtranspose2d(rvcx_ctx* ctx, const float* x, const float* w, const float* bias, float* y,
                            int B, int Cin, int H, int W, int Cout, int act) {
  API_BEGIN(ctx)
  ensure_slab(*C);
  const int Wp = W + 2, Wpo = 2 * W + 2;
  size_t nx = (size_t)B * Cin * H * Wp, ny = (size_t)B * Cout * 2 * H * Wpo;
  C->arena.reserve((nx + ny) * 4 + (64 << 20));
  C->arena.reset();
  ConvT2dW L = make_convT2d(*C, w, nullptr, bias, Cin, Cout);
  std::vector<float> xp = pad_rows(x, (size_t)B * Cin, H, W);
  float* dx = to_dev(*C, xp.data(), nx);
  float* dy = C->arena.alloc<float>(ny);
  RVCX_HIP(hipMemsetAsync(dy, 0xff, ny * 4, C->stream));  // NaN fill: every element must be written
  ConvArgs a = convT2d_args(L, dx, dy, B, H, Wp);
  a.act = act;
  C->conv(a);
  std::vector<float> yp(ny);
  to_host(*C, yp.data(), dy, ny);
  for (size_t r = 0; r < (size_t)B * Cout * 2 * H; ++r)
    if (yp[r * Wpo] != 0.f || yp[r * Wpo + Wpo - 1] != 0.f) fail("convT2d: pad column not zero");
  unpad_rows(yp, y, (size_t)B * Cout, 2 * H, 2 * W);
  C->arena.reset();
  API_END
}

}  // extern "C"
