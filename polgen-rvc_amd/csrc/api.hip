// C ABI of librvcx.so (include/rvcx.h).  Nothing throws across this boundary.
#include "../../include/rvcx.h"

#include <cstdlib>
#include <mutex>

#include "ctx.h"
#include "layers.h"
#include "models.h"
#include "ops.h"
#include "pipeline.h"

using namespace rvcx;

// One context = one set of streams, arenas and resident models.  The reference builds fresh model objects for every
// request (rvc/scripts/voice_conversion.py:71-100), so two Gradio worker threads never share state there; here every
// thread of a process shares the one resident context (infer/_state.py) and ctypes releases the GIL -- so every entry
// point takes the context's mutex (recursive: an entry point may call another).  Calls on one context QUEUE; throughput
// comes from rvcx_convert_batch (one call, many utterances), not from threads.  Different contexts stay concurrent.
struct rvcx_ctx {
  Ctx c;
  std::recursive_mutex mu;
};
using CtxLock = std::unique_lock<std::recursive_mutex>;
static CtxLock lock_ctx(rvcx_ctx* h) { return h ? CtxLock(h->mu) : CtxLock(); }

static thread_local std::string g_last_error;

// Every entry point's body runs inside api_call():
//  * fp16-split range guard.  If a split-fp16 kernel reported an activation it could not represent
//    (Ctx::take_overflow), the FIRST offending layer of the call (launch order; every layer stamps its own device
//    word) is pinned to the exact-fp32 kernels for the life of its model and the call is repeated -- a one-off per
//    model and layer (rvcx_fp32_reruns counts the repeats, rvcx_fp32_layers the pinned layers).  If no layer can be
//    named (kernel-level test entry points pack their weights per call) or after kMaxAttempts - 1 repeats, the last
//    attempt runs everything on the exact-fp32 kernels (thread-local g_force_fp32).
//  * BiGRU cluster time-out.  The cluster kernel needs its workgroups co-resident; if a partner never showed up
//    (Ctx::check_dev_err -> GruTimeout) the call is repeated once with the single-workgroup GRU kernel.
// Bodies are written to be repeatable (they reset the arena first); load / unload entry points run once (repeat = false).
struct Fp32Scope {
  bool saved;
  explicit Fp32Scope(bool on) : saved(g_force_fp32) { g_force_fp32 = saved || on; }
  ~Fp32Scope() { g_force_fp32 = saved; }
};
struct GruScope {
  bool saved;
  explicit GruScope(bool on) : saved(g_gru_no_cluster) { g_gru_no_cluster = saved || on; }
  ~GruScope() { g_gru_no_cluster = saved; }
};
constexpr int kMaxAttempts = 6;

static std::vector<WeightRegion*> all_regions(Ctx& c, uint64_t* hash);

// pins the first layer (in launch order) whose activations left fp16 range; false: none of the resident models named one
static bool localize_overflow(Ctx& c) {
  WeightRegion* best_r = nullptr;
  int best = 0, best_i = -1;
  for (WeightRegion* r : all_regions(c, nullptr)) {
    int i = -1;
    const int v = r->first_overflow(&i);       // also clears the region's words
    if (v > best) {
      best = v;
      best_i = i;
      best_r = r;
    }
  }
  if (!best_r) return false;
  best_r->drop_flag(best_i);
  return true;
}

static void reset_after_failure(Ctx& c) {
  (void)hipDeviceSynchronize();    // side streams may still use arena memory
  c.arena.reset();
  c.arena_f0.reset();
  c.arena_hub.reset();
}

template <typename F>
static int api_call(rvcx_ctx* ctxp, bool repeat, F&& body) {
  Ctx* C = ctxp ? &ctxp->c : nullptr;
  CtxLock guard = lock_ctx(ctxp);
  try {
    if (!C) fail("null context");
    RVCX_HIP(hipSetDevice(C->device));
    if (!repeat) C->arena_budget = 0;     // loads / unloads change what is free: convert_micro_batch probes again
    const int last = repeat ? kMaxAttempts - 1 : 0;
    bool gru_plain = false;
    for (int attempt = 0; attempt <= last; ++attempt) {
      Fp32Scope fp32_scope(attempt > 0 && attempt == last);
      GruScope gru_scope(gru_plain);
      C->launch_seq = 0;
      C->err_snapshot = false;
      try {
        body(C);
      } catch (const GruTimeout&) {
        if (!repeat || gru_plain) throw;
        gru_plain = true;
        C->gru_fallbacks++;
        reset_after_failure(*C);
        --attempt;
        continue;
      }
      if (attempt < last && C->take_overflow()) {
        C->fp32_reruns++;
        if (!localize_overflow(*C)) attempt = last - 1;     // nobody to pin: everything on fp32 next
        continue;
      }
      if (attempt > 0 && attempt == last) (void)C->take_overflow();   // producers of split tensors may have re-raised the bit
      break;
    }
    return 0;
  } catch (const std::exception& e) {
    g_last_error = e.what();
    if (C) {
      C->last_error = e.what();
      reset_after_failure(*C);
    }
    (void)hipGetLastError();
    return -1;
  }
}

// Tuning / fault-injection hooks (rvcx_conv_override, rvcx_debug_inject, rvcx_bench_*) are process-wide levers a serving
// process must never meet by accident: they are refused (-2) unless the process was started with RVCX_DEBUG=1
// (read once; tests/conftest.py and tools/ set it).
static bool debug_hooks_enabled() {
  static const bool on = [] {
    const char* e = getenv("RVCX_DEBUG");
    return e && atoi(e) != 0;
  }();
  return on;
}
#define REQUIRE_DEBUG(ctxp, name)                                                             \
  if (!debug_hooks_enabled()) {                                                               \
    g_last_error = name ": debug / tuning hook refused (start the process with RVCX_DEBUG=1)"; \
    if ((ctxp) != nullptr) {                                                                  \
      CtxLock dbg_guard_ = lock_ctx((rvcx_ctx*)(ctxp));                                       \
      ((rvcx_ctx*)(ctxp))->c.last_error = g_last_error;                                       \
    }                                                                                         \
    return -2;                                                                                \
  }

#define API_BEGIN(ctxp) return api_call((ctxp), true, [&](Ctx* C) {
#define API_BEGIN_ONCE(ctxp) return api_call((ctxp), false, [&](Ctx* C) {
#define API_END });

// copy n elements from host-or-device memory into the arena
template <typename T>
static T* any_to_dev(Ctx& c, const T* p, size_t n) {
  T* d = c.arena.alloc<T>(n);
  RVCX_HIP(hipMemcpyAsync(d, p, n * sizeof(T), hipMemcpyDefault, c.stream));
  return d;
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
    // Exactly four streams (main, F0 / front, two auxiliaries) and no more: HIP multiplexes a process's streams onto four
    // hardware queues, and a stream that shares one pays for it on every launch.  Rounds 2-3 gave HuBERT a fifth, CU-masked
    // stream ("masked streams get their own queue" -- not in this process): its 143 launches ran at twice their stand-alone
    // time whenever the F0 model was busy too, 4 ms per call in the batched workloads (round 4: HuBERT rides aux[0], which is
    // idle while the front end runs; C3 1220 -> 1290x, C5 1055 -> 1203x; tools/stream_queue_probe.hip shows the sharing).
    {
      // RMVPE ends in a latency-bound serial GRU: give its stream dispatch priority over HuBERT's wide kernels
      int lo = 0, hi = 0;
      RVCX_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
      const bool prio = !getenv("RVCX_F0_PRIORITY") || atoi(getenv("RVCX_F0_PRIORITY")) != 0;
      RVCX_HIP(hipStreamCreateWithPriority(&h->c.stream2, hipStreamNonBlocking, prio ? hi : lo));
    }
    RVCX_HIP(hipStreamCreateWithFlags(&h->c.aux[0], hipStreamNonBlocking));
    {
      // RVCX_HUBERT_CUS = N (round 6 probe, with RVCX_HUBERT_ON=aux1 RVCX_HUBERT_GATE=0): aux[1] -- unused by the default two
      // branch streams, created fourth like before so that it shares its hardware queue with the main stream, which idles
      // during a single clip's front end -- is confined to the first N CUs (bit i = XCD i mod 8, CU i / 8:
      // tools/cu_mask_probe.hip), for a HuBERT that runs beside the whole F0 model on its own part of the chip
      const int cus = getenv("RVCX_HUBERT_CUS") ? atoi(getenv("RVCX_HUBERT_CUS")) : 0;
      if (cus > 0 && cus < 256) {
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < cus; ++i) mask[i >> 5] |= 1u << (i & 31);
        RVCX_HIP(hipExtStreamCreateWithCUMask(&h->c.aux[1], 8, mask));
      } else {
        RVCX_HIP(hipStreamCreateWithFlags(&h->c.aux[1], hipStreamNonBlocking));
      }
    }
    for (auto& e : h->c.ev_aux) RVCX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    RVCX_HIP(hipEventCreateWithFlags(&h->c.ev_fork, hipEventDisableTiming));
    for (auto& e : h->c.ev_src) RVCX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    RVCX_HIP(hipEventCreateWithFlags(&h->c.ev_join, hipEventDisableTiming));
    RVCX_HIP(hipEventCreateWithFlags(&h->c.ev_hub, hipEventDisableTiming));
    for (auto& e : h->c.ev_hubdone) RVCX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : h->c.ev_syn) RVCX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    RVCX_HIP(hipEventCreateWithFlags(&h->c.ev_io, hipEventDisableTiming));
    for (auto& e : h->c.ev_front) RVCX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : h->c.ev_done) RVCX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    conv_init();
    resblock_pair_init();
    gemm_init();
    // Round 6: TWO branch streams (main + aux[0]) by default.  With the third (aux[1]) in use the process has four busy
    // streams besides the null stream, and the main stream's ~110 small TextEncoder / flow launches of a single clip run
    // 0.4 ms slower (the hardware-queue sharing of DESIGN "Batching and streams"): C2 1145 - 1151 -> 1179 - 1182x, C5 1276 -
    // 1280 -> 1291 - 1292x, C3 level (tools/sweep_c2_knobs.sh; all on the main stream: C2 1169 - 1177, C5 1219).
    h->c.resblock_streams = getenv("RVCX_RESBLOCK_STREAMS") ? atoi(getenv("RVCX_RESBLOCK_STREAMS")) : 2;
    // RVCX_SERIAL=1: every launch on the one main stream (rocprofv3 kernel durations are then each launch's own)
    h->c.serial_env = getenv("RVCX_SERIAL") && atoi(getenv("RVCX_SERIAL")) != 0;
    h->c.serial = h->c.serial_env;
    RVCX_HIP(hipMalloc(&h->c.dev_err, sizeof(int)));
    RVCX_HIP(hipMemset(h->c.dev_err, 0, sizeof(int)));
    RVCX_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->c.err_host), sizeof(int), hipHostMallocDefault));
    *h->c.err_host = 0;
    h->c.arena.reserve((size_t)256 << 20);
    {
      // One-time work a serving process should not pay inside its first request (round 2: 88 ms in the first call's
      // high-pass stage): the code object is loaded on the first launch, and every stream / hardware queue is
      // created on its first use.  One trivial launch per stream, then wait.
      float* w = h->c.arena.alloc<float>(256);
      hipStream_t ss[] = {h->c.stream, h->c.stream2, h->c.aux[0], h->c.aux[1]};
      for (hipStream_t s : ss)
        if (s) launch_randn(w, 64, 1, 0, s);
      RVCX_HIP(hipDeviceSynchronize());
      h->c.arena.reset();
    }
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
  // a copy per calling thread: another thread's failing call may replace the context's string at any time
  static thread_local std::string copy;
  if (!ctx) return g_last_error.c_str();
  CtxLock guard = lock_ctx(ctx);
  copy = ctx->c.last_error;
  return copy.c_str();
}

void* rvcx_stream(rvcx_ctx* ctx) { CtxLock ctx_guard_ = lock_ctx(ctx); return ctx ? (void*)ctx->c.stream : nullptr; }

int rvcx_device_info(int device, char* name, int name_cap, int64_t* total_bytes) {
  try {
    int n = 0;
    RVCX_HIP(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) fail("device index out of range");
    hipDeviceProp_t prop;
    RVCX_HIP(hipGetDeviceProperties(&prop, device));
    if (name && name_cap > 0) {
      snprintf(name, (size_t)name_cap, "%s", prop.name);
    }
    if (total_bytes) *total_bytes = (int64_t)prop.totalGlobalMem;
    return 0;
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return -1;
  }
}

int rvcx_mem_info(rvcx_ctx* ctx, int64_t* free_bytes, int64_t* total_bytes) {
  API_BEGIN_ONCE(ctx)
  size_t f = 0, t = 0;
  RVCX_HIP(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = (int64_t)f;
  if (total_bytes) *total_bytes = (int64_t)t;
  API_END
}

int64_t rvcx_fp32_reruns(rvcx_ctx* ctx) { CtxLock ctx_guard_ = lock_ctx(ctx); return ctx ? (int64_t)ctx->c.fp32_reruns : -1; }

int64_t rvcx_fp32_layers(rvcx_ctx* ctx) {
  CtxLock ctx_guard_ = lock_ctx(ctx);
  if (!ctx) return -1;
  int64_t n = 0;
  for (WeightRegion* r : all_regions(ctx->c, nullptr)) n += r->dropped();
  return n;
}

int rvcx_fp32_pinned(rvcx_ctx* ctx, char* buf, int cap) {
  CtxLock ctx_guard_ = lock_ctx(ctx);
  if (!ctx || !buf || cap <= 0) return -1;
  Ctx& c = ctx->c;
  std::string t;
  if (c.hubert) t += c.hubert->region->pinned_text("hubert");
  if (c.rmvpe) t += c.rmvpe->region->pinned_text("rmvpe");
  if (c.fcpe) t += c.fcpe->region->pinned_text("fcpe");
  if (c.crepe) t += c.crepe->region->pinned_text("crepe");
  for (size_t i = 0; i < c.synths.size(); ++i)
    if (c.synths[i]) t += c.synths[i]->region->pinned_text("voice model " + std::to_string(i));
  const int n = (int)std::min<size_t>(t.size(), (size_t)cap - 1);
  memcpy(buf, t.data(), (size_t)n);
  buf[n] = 0;
  return (int)t.size();
}

int64_t rvcx_gru_fallbacks(rvcx_ctx* ctx) { CtxLock ctx_guard_ = lock_ctx(ctx); return ctx ? (int64_t)ctx->c.gru_fallbacks : -1; }

int rvcx_gru_publish_probe(rvcx_ctx* ctx) {
  int state = -1;
  const int rc = api_call(ctx, false, [&](Ctx*) {
    (void)bigru_probe_publish();
    state = bigru_probe_state();
  });
  return rc ? -2 : state;
}

int64_t rvcx_index_exhaustive(rvcx_ctx* ctx) {
  CtxLock ctx_guard_ = lock_ctx(ctx);
  if (!ctx || !ctx->c.index || !ctx->c.index->exhaustive) return -1;
  int v = 0;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  (void)hipMemcpy(&v, ctx->c.index->exhaustive, sizeof(int), hipMemcpyDeviceToHost);
  (void)hipMemset(ctx->c.index->exhaustive, 0, sizeof(int));
  return v;
}

int rvcx_debug_inject(rvcx_ctx* ctx, int what) {
  REQUIRE_DEBUG(ctx, "rvcx_debug_inject")
  CtxLock ctx_guard_ = lock_ctx(ctx);
  if (ctx && what == 2) {          // read and clear the raw device error word (debugging builds set extra bits)
    int v = 0;
    (void)hipMemcpy(&v, ctx->c.dev_err, sizeof(int), hipMemcpyDeviceToHost);
    (void)hipMemset(ctx->c.dev_err, 0, sizeof(int));
    return v;
  }
  if (ctx && what == 3) {          // the next BiGRU cluster launch of this thread loses a member: its partners really time out
    g_gru_drop_member = 1;
    return 0;
  }
  if (!ctx || what != 1) return -1;
  ctx->c.inject_gru_timeout = true;
  return 0;
}

double rvcx_flop_counter(rvcx_ctx* ctx, int reset) {
  CtxLock ctx_guard_ = lock_ctx(ctx);
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
// kernel-level entry points pack their weights into a region that lives for the call only
#define TEMP_REGION(C) WeightRegion tmp_region_; RegionScope tmp_scope_(*(C), tmp_region_)

int rvcx_op_conv1d(rvcx_ctx* ctx, const float* x, const float* w, const float* bias, const float* res,
                   float* y, int B, int Cin, int Tin, int Cout, int K, int stride, int dil,
                   int pad_left, int Tout, int groups, int pre_lrelu, float pre_slope, int act,
                   float act_slope, const int32_t* lens_in, const int32_t* lens_out) {
  API_BEGIN(ctx)
  TEMP_REGION(C);
  size_t nx = (size_t)B * Cin * Tin, ny = (size_t)B * Cout * Tout;
  C->arena.reserve((nx + 2 * ny) * 4 + (64 << 20));
  C->arena.reset();
  ConvW L = make_conv(*C, w, bias, Cout, Cin / groups, K, groups, true);
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

int rvcx_op_resblock_pair(rvcx_ctx* ctx, const float* x, const float* w1, const float* b1, const float* w2,
                          const float* b2, float* y, int B, int Cc, int T, int K, int dil, float slope, int fused,
                          const int32_t* lens) {
  API_BEGIN(ctx)
  TEMP_REGION(C);
  const size_t n = (size_t)B * Cc * T;
  C->arena.reserve(n * 4 * 4 + (64 << 20));
  C->arena.reset();
  ConvW L1 = make_conv(*C, w1, b1, Cc, Cc, K, 1, true);
  ConvW L2 = make_conv(*C, w2, b2, Cc, Cc, K, 1, true);
  float* dx = to_dev(*C, x, n);
  float* dt = C->arena.alloc<float>(n);
  float* dy = C->arena.alloc<float>(n);
  RVCX_HIP(hipMemsetAsync(dy, 0xff, n * 4, C->stream));      // NaN fill: every element must be written
  const int* dl = to_dev_i(*C, lens, B);
  if (fused) {
    PairArgs pa;
    pa.x = dx;
    pa.y = dy;
    pa.w1 = (L1.w_h3 && *L1.h3_ok) ? L1.w_h3 : nullptr;
    pa.w2 = (L2.w_h3 && *L2.h3_ok) ? L2.w_h3 : nullptr;
    pa.b1 = L1.bias;
    pa.b2 = L2.bias;
    pa.lens = dl;
    pa.B = B;
    pa.C = Cc;
    pa.T = T;
    pa.bs = (long)Cc * T;
    pa.cs = T;
    pa.k = K;
    pa.dil = dil;
    pa.slope = slope;
    if (!resblock_pair_ok(pa) && !g_force_fp32) fail("resblock pair: shape not supported by the fused kernel");
  }
  if (fused && !g_force_fp32) {
    PairArgs pa;
    pa.x = dx;
    pa.y = dy;
    pa.w1 = L1.w_h3;
    pa.w2 = L2.w_h3;
    pa.b1 = L1.bias;
    pa.b2 = L2.bias;
    pa.lens = dl;
    pa.B = B;
    pa.C = Cc;
    pa.T = T;
    pa.bs = (long)Cc * T;
    pa.cs = T;
    pa.k = K;
    pa.dil = dil;
    pa.slope = slope;
    C->pair_on(pa, C->stream);
  } else {   // the two launches the fused kernel replaces (synth.hip's fallback path; also the exact-fp32 rerun)
    ConvArgs a = conv1d_args(L1, dx, dt, B, T, T, 1, dil, (K * dil - dil) / 2);
    a.pre_act = ACT_LRELU;
    a.pre_slope = slope;
    a.act = ACT_LRELU;
    a.act_slope = slope;
    a.lens_in = dl;
    a.lens_out = dl;
    ConvArgs a2 = conv1d_args(L2, dt, dy, B, T, T, 1, 1, (K - 1) / 2);
    const bool split = conv_h3_split_ok(a) && conv_h3_split_ok(a2);
    if (split) {
      a.y_split = dt;
      a.y = nullptr;
    }
    C->conv(a);
    if (split) a2.x_split = dt;
    conv_set_res(a2, dx, Cc, T);
    a2.lens_in = dl;
    a2.lens_out = dl;
    C->conv(a2);
  }
  to_host(*C, y, dy, n);
  C->arena.reset();
  API_END
}

int rvcx_op_resblock3(rvcx_ctx* ctx, const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                      float* y, int B, int Cc, int T, const int32_t* dils, float slope, const int32_t* lens) {
  API_BEGIN(ctx)
  TEMP_REGION(C);
  const size_t n = (size_t)B * Cc * T, wn = (size_t)Cc * Cc * 3;
  C->arena.reserve(n * 4 * 3 + (64 << 20));
  C->arena.reset();
  ConvW L1[3], L2[3];
  for (int s = 0; s < 3; ++s) {
    L1[s] = make_conv(*C, w1 + s * wn, b1 ? b1 + (size_t)s * Cc : nullptr, Cc, Cc, 3, 1, true);
    L2[s] = make_conv(*C, w2 + s * wn, b2 ? b2 + (size_t)s * Cc : nullptr, Cc, Cc, 3, 1, true);
  }
  float* dx = to_dev(*C, x, n);
  float* dy = C->arena.alloc<float>(n);
  RVCX_HIP(hipMemsetAsync(dy, 0xff, n * 4, C->stream));      // NaN fill: every element must be written
  const int* dl = to_dev_i(*C, lens, B);
  Block3Args a;
  a.x = dx;
  a.y = dy;
  for (int s = 0; s < 3; ++s) {
    a.w1[s] = (L1[s].w_h3 && *L1[s].h3_ok) ? L1[s].w_h3 : nullptr;
    a.w2[s] = (L2[s].w_h3 && *L2[s].h3_ok) ? L2[s].w_h3 : nullptr;
    a.b1[s] = L1[s].bias;
    a.b2[s] = L2[s].bias;
    a.dil[s] = dils ? dils[s] : 2 * s + 1;       // NULL: ResBlock1's own dilations (1, 3, 5), residuals.py:15-62
  }
  a.ovf_layer = L1[0].ovf_word;
  a.lens = dl;
  a.B = B;
  a.C = Cc;
  a.T = T;
  a.bs = (long)Cc * T;
  a.cs = T;
  a.slope = slope;
  a.any_shape = true;
  if (!resblock3_ok(a)) fail("resblock3: shape not supported by the whole-block kernel");
  C->block3_on(a, C->stream);
  to_host(*C, y, dy, n);
  C->arena.reset();
  API_END
}

int rvcx_bench_resblock_pair(rvcx_ctx* ctx, int B, int Cc, int T, int K, int dil, int fused, int iters,
                             float* ms_per_launch) {
  REQUIRE_DEBUG(ctx, "rvcx_bench_resblock_pair")
  API_BEGIN(ctx)
  TEMP_REGION(C);
  const size_t n = (size_t)B * Cc * T;
  C->arena.reserve(n * 4 * 4 + (64 << 20));
  C->arena.reset();
  std::vector<float> w((size_t)Cc * Cc * K), bias((size_t)Cc, 0.1f);
  for (size_t i = 0; i < w.size(); ++i) w[i] = ((float)((i * 2654435761u) % 2001) / 1000.f - 1.f) / std::sqrt((float)Cc * K);
  ConvW L1 = make_conv(*C, w.data(), bias.data(), Cc, Cc, K, 1, true);
  ConvW L2 = make_conv(*C, w.data(), bias.data(), Cc, Cc, K, 1, true);
  float* dx = C->arena.alloc<float>(n);
  float* dt = C->arena.alloc<float>(n);
  float* dy = C->arena.alloc<float>(n);
  launch_randn(dx, n, 1, 0, C->stream);
  if (getenv("RVCX_BENCH_ZERO")) RVCX_HIP(hipMemsetAsync(dx, 0, n * 4, C->stream));   // power / clock experiments only
  PairArgs pa;
  pa.x = dx;
  pa.y = dy;
  pa.w1 = L1.w_h3;
  pa.w2 = L2.w_h3;
  pa.b1 = L1.bias;
  pa.b2 = L2.bias;
  pa.B = B;
  pa.C = Cc;
  pa.T = T;
  pa.bs = (long)Cc * T;
  pa.cs = T;
  pa.k = K;
  pa.dil = dil;
  ConvArgs a = conv1d_args(L1, dx, dt, B, T, T, 1, dil, (K * dil - dil) / 2);
  a.pre_act = ACT_LRELU;
  a.pre_slope = 0.1f;
  a.act = ACT_LRELU;
  a.act_slope = 0.1f;
  ConvArgs a2 = conv1d_args(L2, dt, dy, B, T, T, 1, 1, (K - 1) / 2);
  const bool split = conv_h3_split_ok(a) && conv_h3_split_ok(a2);
  if (split) {
    a.y_split = dt;
    a.y = nullptr;
    a2.x_split = dt;
  }
  conv_set_res(a2, dx, Cc, T);
  auto once = [&]() {
    if (fused) {
      C->pair_on(pa, C->stream);
    } else {
      C->conv(a);
      C->conv(a2);
    }
  };
  if (fused && !resblock_pair_ok(pa)) fail("resblock pair: shape not supported by the fused kernel");
  once();
  if (fused && getenv("RVCX_PAIR_TRACE")) {     // one traced launch: per-workgroup phase stamps -> CSV (tools/pair_trace.py)
    const size_t cap = (size_t)B * (T / 32 + 64) * 8;
    long long* tr = C->arena.alloc<long long>(cap);
    RVCX_HIP(hipMemsetAsync(tr, 0, cap * 8, C->stream));
    pa.trace = tr;
    once();
    pa.trace = nullptr;
    std::vector<long long> h(cap);
    RVCX_HIP(hipMemcpyAsync(h.data(), tr, cap * 8, hipMemcpyDeviceToHost, C->stream));
    RVCX_HIP(hipStreamSynchronize(C->stream));
    if (FILE* f = fopen(getenv("RVCX_PAIR_TRACE"), "a")) {
      fprintf(f, "# C %d T %d K %d dil %d\n", Cc, T, K, dil);
      for (size_t w = 0; w * 8 < cap; ++w)
        if (h[w * 8]) {
          for (int k = 0; k < 8; ++k) fprintf(f, "%lld%c", h[w * 8 + k], k == 7 ? '\n' : ',');
        }
      fclose(f);
    }
  }
  hipEvent_t e0, e1;
  RVCX_HIP(hipEventCreate(&e0));
  RVCX_HIP(hipEventCreate(&e1));
  RVCX_HIP(hipEventRecord(e0, C->stream));
  for (int i = 0; i < iters; ++i) once();
  RVCX_HIP(hipEventRecord(e1, C->stream));
  RVCX_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  RVCX_HIP(hipEventElapsedTime(&ms, e0, e1));
  *ms_per_launch = ms / iters;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  C->arena.reset();
  API_END
}

int rvcx_conv_override(int tile, int variant, int splitk) {
  REQUIRE_DEBUG(nullptr, "rvcx_conv_override")
  rvcx::g_conv_override.tile = tile;
  rvcx::g_conv_override.variant = variant;
  rvcx::g_conv_override.splitk = splitk;
  return 0;
}

int rvcx_bench_conv1d(rvcx_ctx* ctx, int B, int Cin, int Tin, int Cout, int K, int stride, int dil, int groups,
                      int iters, float* ms_per_launch) {
  REQUIRE_DEBUG(ctx, "rvcx_bench_conv1d")
  API_BEGIN(ctx)
  TEMP_REGION(C);
  const int pad = (K * dil - dil) / 2;
  const int Tout = (Tin + 2 * pad - dil * (K - 1) - 1) / stride + 1;
  size_t nx = (size_t)B * Cin * Tin, ny = (size_t)B * Cout * Tout;
  C->arena.reserve((nx + 2 * ny) * 4 + (64 << 20));
  C->arena.reset();
  std::vector<float> w((size_t)Cout * (Cin / groups) * K), bias((size_t)Cout, 0.1f);
  for (size_t i = 0; i < w.size(); ++i) w[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
  ConvW L = make_conv(*C, w.data(), bias.data(), Cout, Cin / groups, K, groups, true);
  float* dx = C->arena.alloc<float>(nx);
  float* dy = C->arena.alloc<float>(ny);
  float* dr = C->arena.alloc<float>(ny);
  launch_randn(dx, nx, 1, 0, C->stream);
  launch_randn(dr, ny, 2, 0, C->stream);
  ConvArgs a = conv1d_args(L, dx, dy, B, Tin, Tout, stride, dil, pad);
  conv_set_res(a, dr, Cout, Tout);
  a.pre_act = ACT_LRELU;
  a.pre_slope = 0.1f;
  C->conv(a);
  long long* dtrace = nullptr;
  const long ntrace = 1L << 16;
  if (getenv("RVCX_TRACE")) {
    RVCX_HIP(hipMalloc(&dtrace, ntrace * 6 * sizeof(long long)));
    RVCX_HIP(hipMemset(dtrace, 0, ntrace * 6 * sizeof(long long)));
    a.trace = dtrace;
  }
  hipEvent_t e0, e1;
  RVCX_HIP(hipEventCreate(&e0));
  RVCX_HIP(hipEventCreate(&e1));
  RVCX_HIP(hipEventRecord(e0, C->stream));
  for (int i = 0; i < iters; ++i) C->conv(a);
  RVCX_HIP(hipEventRecord(e1, C->stream));
  RVCX_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  RVCX_HIP(hipEventElapsedTime(&ms, e0, e1));
  *ms_per_launch = ms / iters;
  if (dtrace) {
    std::vector<long long> ht((size_t)ntrace * 6);
    RVCX_HIP(hipMemcpy(ht.data(), dtrace, ht.size() * sizeof(long long), hipMemcpyDeviceToHost));
    FILE* f = fopen(getenv("RVCX_TRACE"), "w");
    for (long i = 0; i < ntrace; ++i)
      if (ht[i * 6 + 5]) fprintf(f, "%ld,%lld,%lld,%lld,%lld,%lld,%lld\n", i, ht[i*6], ht[i*6+1], ht[i*6+2], ht[i*6+3], ht[i*6+4], ht[i*6+5]);
    fclose(f);
    (void)hipFree(dtrace);
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  C->arena.reset();
  API_END
}

int rvcx_op_convtranspose1d(rvcx_ctx* ctx, const float* x, const float* w, const float* bias, float* y,
                            int B, int Cin, int Tin, int Cout, int K, int stride, int pad,
                            int pre_lrelu, float pre_slope) {
  API_BEGIN(ctx)
  TEMP_REGION(C);
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
  TEMP_REGION(C);
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

int rvcx_op_convblock2d(rvcx_ctx* ctx, const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                        const float* wsc, const float* bsc, float* y, int B, int Cin, int Cout, int H, int W,
                        const int32_t* rows) {
  API_BEGIN(ctx)
  TEMP_REGION(C);
  RVCX_CHECK(wsc || Cin == Cout, "op_convblock2d: Cin != Cout needs the 1x1 shortcut");
  const int Wp = W + 2;
  const size_t nx = (size_t)B * Cin * H * Wp, ny = (size_t)B * Cout * H * Wp;
  C->arena.reserve((nx + 3 * ny) * 4 + (64 << 20));
  C->arena.reset();
  ConvW c1 = make_conv(*C, w1, b1, Cout, Cin, 9, 1), c2 = make_conv(*C, w2, b2, Cout, Cout, 9, 1), sc;
  if (wsc) sc = make_conv(*C, wsc, bsc, Cout, Cin, 1, 1);
  std::vector<float> xp = pad_rows(x, (size_t)B * Cin, H, W);
  if (rows)                                       // what the model guarantees: nothing but zeros below an item's last row
    for (int b = 0; b < B; ++b) {
      RVCX_CHECK(rows[b] >= 0 && rows[b] <= H, "op_convblock2d: rows outside [0, H]");
      for (int c = 0; c < Cin; ++c)
        std::fill(xp.begin() + (((size_t)b * Cin + c) * H + rows[b]) * Wp, xp.begin() + (((size_t)b * Cin + c) + 1) * H * Wp, 0.f);
    }
  float* dx = to_dev(*C, xp.data(), nx);
  float* dy = C->arena.alloc<float>(ny);
  float* t1 = C->arena.alloc<float>(ny);
  float* t2 = C->arena.alloc<float>(ny);
  RVCX_HIP(hipMemsetAsync(dy, 0xff, ny * 4, C->stream));  // NaN fill: every element must be written
  RVCX_HIP(hipMemsetAsync(t1, 0xff, ny * 4, C->stream));
  rvcx::rmvpe_block_op(*C, c1, c2, wsc ? &sc : nullptr, dx, dy, t1, t2, B, H, Wp, rows, C->stream);
  std::vector<float> yp(ny);
  to_host(*C, yp.data(), dy, ny);
  for (size_t r = 0; r < (size_t)B * Cout * H; ++r)
    if (yp[r * Wp] != 0.f || yp[r * Wp + Wp - 1] != 0.f) fail("convblock2d: pad column not zero");
  unpad_rows(yp, y, (size_t)B * Cout, H, W);
  C->arena.reset();
  API_END
}

int rvcx_op_convtranspose2d(rvcx_ctx* ctx, const float* x, const float* w, const float* bias, float* y,
                            int B, int Cin, int H, int W, int Cout, int act) {
  API_BEGIN(ctx)
  TEMP_REGION(C);
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

// ------------------------------------------------------------------------------------------
// model loading
// ------------------------------------------------------------------------------------------
static TensorTable make_table(const rvcx_tensor* tbl, int n) {
  TensorTable t;
  for (int i = 0; i < n; ++i) {
    HostTensor h;
    h.data = tbl[i].data;
    h.dtype = tbl[i].dtype;
    for (int d = 0; d < tbl[i].ndim; ++d) h.shape.push_back(tbl[i].shape[d]);
    t.add(tbl[i].name, std::move(h));
  }
  return t;
}

int rvcx_load_synth(rvcx_ctx* ctx, const rvcx_synth_cfg* cfg, const rvcx_tensor* tbl, int n, int* model_id) {
  API_BEGIN_ONCE(ctx)
  TensorTable t = make_table(tbl, n);
  auto m = synth_load(*C, *cfg, t);
  int id = -1;
  for (size_t i = 0; i < C->synths.size(); ++i)
    if (!C->synths[i]) id = (int)i;
  if (id < 0) {
    C->synths.emplace_back();
    id = (int)C->synths.size() - 1;
  }
  C->synths[id] = std::move(m);
  *model_id = id;
  API_END
}

int rvcx_unload_synth(rvcx_ctx* ctx, int model_id) {
  API_BEGIN_ONCE(ctx)
  if (model_id < 0 || model_id >= (int)C->synths.size()) fail("bad model id");
  RVCX_HIP(hipDeviceSynchronize());
  C->synths[model_id].reset();   // frees the model's weight region
  API_END
}

static SynthModel& get_synth(Ctx& c, int id) {
  if (id < 0 || id >= (int)c.synths.size() || !c.synths[id]) fail("synth model not loaded");
  return *c.synths[id];
}

int rvcx_synth_upp(rvcx_ctx* ctx, int model_id) {
  CtxLock ctx_guard_ = lock_ctx(ctx);
  if (!ctx || model_id < 0 || model_id >= (int)ctx->c.synths.size() || !ctx->c.synths[model_id]) return -1;
  return ctx->c.synths[model_id]->upp;
}

// every weight region of the context in a fixed order: HuBERT, RMVPE, FCPE, voice models by id, index
static std::vector<WeightRegion*> all_regions(Ctx& c, uint64_t* hash) {
  std::vector<WeightRegion*> r;
  uint64_t h = 1469598103934665603ull;
  auto add = [&](WeightRegion* w, uint64_t tag) {
    h = (h ^ tag) * 1099511628211ull;
    h = (h ^ (w ? w->layout_hash() : 0)) * 1099511628211ull;
    if (w) r.push_back(w);
  };
  add(c.hubert ? c.hubert->region.get() : nullptr, 1);
  add(c.rmvpe ? c.rmvpe->region.get() : nullptr, 2);
  add(c.fcpe ? c.fcpe->region.get() : nullptr, 4);
  add(c.crepe ? c.crepe->region.get() : nullptr, 5);
  for (size_t i = 0; i < c.synths.size(); ++i) add(c.synths[i] ? c.synths[i]->region.get() : nullptr, 16 + i);
  add(c.index ? c.index->region.get() : nullptr, 3);
  if (hash) *hash = h;
  return r;
}

int rvcx_weights_regions(rvcx_ctx* ctx, int cap, void** dev_ptrs, int64_t* nbytes, uint64_t* layout_hash) {
  CtxLock ctx_guard_ = lock_ctx(ctx);
  Ctx* C = ctx ? &ctx->c : nullptr;
  try {
    if (!C) fail("null context");
    RVCX_HIP(hipSetDevice(C->device));
    int k = 0;
    for (WeightRegion* r : all_regions(*C, layout_hash))
      for (int i = 0; i < r->n_chunks(); ++i, ++k)
        if (k < cap) {
          dev_ptrs[k] = r->chunk_base(i);
          nbytes[k] = (int64_t)r->chunk_used(i);
        }
    return k;
  } catch (const std::exception& e) {
    g_last_error = e.what();
    if (C) C->last_error = e.what();
    return -1;
  }
}

int rvcx_weights_clone(rvcx_ctx* ctx, rvcx_ctx* src) {
  // both contexts' mutexes, acquired deadlock-free (two threads cloning A <- B and B <- A)
  CtxLock ga, gb;
  if (ctx && src && ctx != src) {
    ga = CtxLock(ctx->mu, std::defer_lock);
    gb = CtxLock(src->mu, std::defer_lock);
    std::lock(ga, gb);
  }
  API_BEGIN_ONCE(ctx)
  if (!src) fail("weights_clone: null source context");
  if (src->c.device != C->device) fail("weights_clone: contexts live on different devices (use the RCCL broadcast)");
  uint64_t ha = 0, hb = 0;
  std::vector<WeightRegion*> ra = all_regions(src->c, &ha), rb = all_regions(*C, &hb);
  if (ha != hb || ra.size() != rb.size()) fail("weights_clone: the two contexts hold different model layouts");
  RVCX_HIP(hipDeviceSynchronize());
  for (size_t r = 0; r < ra.size(); ++r) {
    if (ra[r]->n_chunks() != rb[r]->n_chunks()) fail("weights_clone: chunk lists differ");
    for (int i = 0; i < ra[r]->n_chunks(); ++i) {
      if (ra[r]->chunk_used(i) != rb[r]->chunk_used(i)) fail("weights_clone: chunk sizes differ");
      RVCX_HIP(hipMemcpy(rb[r]->chunk_base(i), ra[r]->chunk_base(i), ra[r]->chunk_used(i), hipMemcpyDeviceToDevice));
    }
    rb[r]->adopt();
  }
  API_END
}

int rvcx_weights_adopt(rvcx_ctx* ctx) {
  API_BEGIN_ONCE(ctx)
  RVCX_HIP(hipDeviceSynchronize());
  for (WeightRegion* r : all_regions(*C, nullptr)) r->adopt();
  API_END
}

static int synth_infer_impl(rvcx_ctx* ctx, int model_id, int B, int T, const int32_t* lens, const float* phone,
                            const int32_t* pitch, const float* pitchf, const int32_t* sid, const float* z_noise,
                            const float* src_noise, uint64_t seed, float* out, float* stats, float* zflow, int dec_skip);

int rvcx_synth_infer(rvcx_ctx* ctx, int model_id, int B, int T, const int32_t* lens, const float* phone,
                     const int32_t* pitch, const float* pitchf, const int32_t* sid, const float* z_noise,
                     const float* src_noise, uint64_t seed, float* out) {
  CtxLock ctx_guard_ = lock_ctx(ctx);
  return synth_infer_impl(ctx, model_id, B, T, lens, phone, pitch, pitchf, sid, z_noise, src_noise, seed, out, nullptr,
                          nullptr, 0);
}

int rvcx_synth_infer_window(rvcx_ctx* ctx, int model_id, int B, int T, const int32_t* lens, const float* phone,
                            const int32_t* pitch, const float* pitchf, const int32_t* sid, const float* z_noise,
                            const float* src_noise, uint64_t seed, int dec_skip, float* out) {
  CtxLock ctx_guard_ = lock_ctx(ctx);
  return synth_infer_impl(ctx, model_id, B, T, lens, phone, pitch, pitchf, sid, z_noise, src_noise, seed, out, nullptr,
                          nullptr, dec_skip < 0 ? 0 : dec_skip);
}

int rvcx_synth_dec_rf(rvcx_ctx* ctx, int model_id) {
  int rf = -1;
  const int rc = api_call(ctx, false, [&](Ctx* C) { rf = get_synth(*C, model_id).dec_rf_frames; });
  return rc ? -1 : rf;
}

int rvcx_synth_infer_taps(rvcx_ctx* ctx, int model_id, int B, int T, const int32_t* lens, const float* phone,
                          const int32_t* pitch, const float* pitchf, const int32_t* sid, const float* z_noise,
                          const float* src_noise, uint64_t seed, float* out, float* stats, float* zflow) {
  CtxLock ctx_guard_ = lock_ctx(ctx);
  return synth_infer_impl(ctx, model_id, B, T, lens, phone, pitch, pitchf, sid, z_noise, src_noise, seed, out, stats,
                          zflow, 0);
}

static int synth_infer_impl(rvcx_ctx* ctx, int model_id, int B, int T, const int32_t* lens, const float* phone,
                            const int32_t* pitch, const float* pitchf, const int32_t* sid, const float* z_noise,
                            const float* src_noise, uint64_t seed, float* out, float* stats, float* zflow, int dec_skip) {
  API_BEGIN(ctx)
  SynthModel& M = get_synth(*C, model_id);
  const int D = M.cfg.input_dim, inter = M.cfg.inter_channels;
  const size_t Tupp = (size_t)T * M.upp;
  C->ensure_splitk(B);
  C->arena.reserve(synth_arena_bytes(M, B, T) + (size_t)B * T * D * 8 + (size_t)B * Tupp * 8 +
                   (size_t)B * 3 * inter * T * 4 + 4096);
  C->arena.reset();
  float* ph = any_to_dev(*C, phone, (size_t)B * T * D);
  float* ph_ct = C->arena.alloc<float>((size_t)B * T * D);
  launch_transpose(ph, ph_ct, B, T, D, C->stream);
  SynthIO io;
  io.B = B;
  io.T = T;
  io.lens_host = lens;
  io.phone_ct = ph_ct;
  io.pitch = any_to_dev<int>(*C, pitch, (size_t)B * T);
  io.pitchf = any_to_dev(*C, pitchf, (size_t)B * T);
  io.sid_host = sid;
  float* zn = C->arena.alloc<float>((size_t)B * inter * T);
  float* sn = C->arena.alloc<float>((size_t)B * Tupp);
  if (z_noise) RVCX_HIP(hipMemcpyAsync(zn, z_noise, (size_t)B * inter * T * 4, hipMemcpyDefault, C->stream));
  else launch_randn(zn, (size_t)B * inter * T, seed, 0, C->stream);
  if (src_noise) RVCX_HIP(hipMemcpyAsync(sn, src_noise, (size_t)B * Tupp * 4, hipMemcpyDefault, C->stream));
  else launch_randn(sn, (size_t)B * Tupp, seed, (uint64_t)1 << 40, C->stream);
  io.z_noise = zn;
  io.src_noise = sn;
  float* dout = C->arena.alloc<float>((size_t)B * Tupp);
  io.out = dout;
  if (stats) io.stats_out = C->arena.alloc<float>((size_t)B * 2 * inter * T);
  if (zflow) io.z_out = C->arena.alloc<float>((size_t)B * inter * T);
  io.dec_skip = dec_skip;
  synth_forward(*C, M, io, nullptr);
  if (stats) RVCX_HIP(hipMemcpyAsync(stats, io.stats_out, (size_t)B * 2 * inter * T * 4, hipMemcpyDefault, C->stream));
  if (zflow) RVCX_HIP(hipMemcpyAsync(zflow, io.z_out, (size_t)B * inter * T * 4, hipMemcpyDefault, C->stream));
  RVCX_HIP(hipMemcpyAsync(out, dout, (size_t)B * Tupp * 4, hipMemcpyDefault, C->stream));
  RVCX_HIP(hipStreamSynchronize(C->stream));
  C->arena.reset();
  API_END
}

int rvcx_op_attention(rvcx_ctx* ctx, const float* q, const float* k, const float* v, float* out, int B, int H,
                      int D, int T, float scale, const float* emb_rel_k, const float* emb_rel_v, int window,
                      const int32_t* lens) {
  API_BEGIN(ctx)
  const size_t n = (size_t)B * H * D * T;
  C->arena.reserve(n * 16 + (attention_scratch_floats(B, H, T, window) + attention_split_floats(B, H, T)) * 4 + (64 << 20));
  C->arena.reset();
  float *dq = to_dev(*C, q, n), *dk = to_dev(*C, k, n), *dv = to_dev(*C, v, n);
  float* dout = C->arena.alloc<float>(n);
  float *ek = nullptr, *ev = nullptr, *scratch = nullptr;
  if (emb_rel_k) {
    ek = to_dev(*C, emb_rel_k, (size_t)(2 * window + 1) * D);
    ev = to_dev(*C, emb_rel_v, (size_t)(2 * window + 1) * D);
    scratch = C->arena.alloc<float>(attention_scratch_floats(B, H, T, window));
  }
  launch_attention(dq, dk, dv, dout, B, H, D, T, T, (long)H * D * T, (long)H * D * T, scale, ek, ev, window,
                   to_dev_i(*C, lens, B), scratch, C->arena.alloc<float>(attention_split_floats(B, H, T)), C->stream);
  to_host(*C, out, dout, n);
  C->arena.reset();
  API_END
}

// host view of split rows: hi + (S lo) / S
static void decode_xs(const std::vector<uint16_t>& raw, long rows, int Cc, float* out) {
  for (long r = 0; r < rows; ++r)
    for (int c = 0; c < Cc; ++c) {
      const size_t e = ((size_t)r * Cc * 2) + (size_t)(c >> 4) * 32 + ((c >> 3) & 1) * 8 + (c & 7);
      out[(size_t)r * Cc + c] = half_to_float(raw[e]) + half_to_float(raw[e + 16]) / 256.f;
    }
}

int rvcx_op_gemm_tm(rvcx_ctx* ctx, const float* x_cf, const float* w, const float* bias, const float* res_tm, int B,
                    int T, int Cin, int Cout, int act, int exact_fp32, float* y_tm, float* y_cf, float* y_split) {
  API_BEGIN(ctx)
  TEMP_REGION(C);
  const long R = (long)B * T;
  C->arena.reserve(((size_t)R * (3 * (size_t)Cin + 4 * (size_t)Cout)) * 4 + (64 << 20));
  C->arena.reset();
  hipStream_t s = C->stream;
  ConvW L = make_conv(*C, w, bias, Cout, Cin, 1, 1, true);
  float* dx = to_dev(*C, x_cf, (size_t)R * Cin);
  float* xf = C->arena.alloc<float>((size_t)R * Cin);
  float* xs = C->arena.alloc<float>((size_t)R * L.cin_gp);
  RVCX_HIP(hipMemsetAsync(xs, 0, (size_t)R * L.cin_gp * 4, s));
  const bool h3 = !exact_fp32 && conv_h3_ok(L) && gemm_h3_enabled() && Cin % 4 == 0;
  launch_cf_to_tm(dx, (long)Cin * T, xf, Cin, h3 ? xs : nullptr, (long)L.cin_gp * 4, B, Cin, T, C->dev_err, nullptr, 0, s);
  GemmArgs g = gemm_args(L, R, T);
  if (h3) g.xs = xs, g.ld_xs = (long)L.cin_gp * 4;
  else g.w_h3 = nullptr, g.x = xf, g.ld_x = Cin;
  g.act = act;
  if (res_tm) g.res = to_dev(*C, res_tm, (size_t)R * Cout), g.ld_res = Cout;
  float* dy = C->arena.alloc<float>((size_t)R * Cout);
  float* dc = C->arena.alloc<float>((size_t)R * Cout);
  float* ds = C->arena.alloc<float>((size_t)R * Cout);
  RVCX_HIP(hipMemsetAsync(dy, 0xff, (size_t)R * Cout * 4, s));
  RVCX_HIP(hipMemsetAsync(dc, 0xff, (size_t)R * Cout * 4, s));
  g.y = dy, g.ld_y = Cout;
  if (y_cf) g.y_cf = dc, g.cf_bs = (long)Cout * T;
  if (y_split && Cout % 16 == 0) g.ys = ds, g.ld_ys = (long)Cout * 4;
  C->gemm_on(g, s);
  to_host(*C, y_tm, dy, (size_t)R * Cout);
  if (y_cf) to_host(*C, y_cf, dc, (size_t)R * Cout);
  if (g.ys) {
    std::vector<uint16_t> raw((size_t)R * Cout * 2);
    RVCX_HIP(hipMemcpy(raw.data(), ds, raw.size() * 2, hipMemcpyDeviceToHost));
    decode_xs(raw, R, Cout, y_split);
  }
  C->arena.reset();
  API_END
}

int rvcx_bench_gemm(rvcx_ctx* ctx, int64_t rows, int Cin, int Cout, int iters, float* ms_per_launch) {
  REQUIRE_DEBUG(ctx, "rvcx_bench_gemm")
  API_BEGIN(ctx)
  TEMP_REGION(C);
  C->arena.reserve(((size_t)rows * ((size_t)Cin + 2 * (size_t)Cout)) * 4 + (64 << 20));
  C->arena.reset();
  hipStream_t s = C->stream;
  std::vector<float> w((size_t)Cout * Cin), bias((size_t)Cout, 0.1f);
  for (size_t i = 0; i < w.size(); ++i) w[i] = ((float)((i * 2654435761u) % 2001) / 1000.f - 1.f) / std::sqrt((float)Cin);
  ConvW L = make_conv(*C, w.data(), bias.data(), Cout, Cin, 1, 1, true);
  float* xf = C->arena.alloc<float>((size_t)rows * Cin);
  float* xs = C->arena.alloc<float>((size_t)rows * L.cin_gp);
  float* dy = C->arena.alloc<float>((size_t)rows * Cout);
  launch_randn(xf, (size_t)rows * Cin, 1, 0, s);
  // N(0,1) data read as a channel-first (1, Cin, rows) map -> split rows
  launch_cf_to_tm(xf, (long)Cin * rows, nullptr, 0, xs, (long)L.cin_gp * 4, 1, Cin, (int)rows, nullptr, nullptr, 0, s);
  GemmArgs g = gemm_args(L, rows, (int)rows);
  g.xs = xs, g.ld_xs = (long)L.cin_gp * 4;
  g.y = dy, g.ld_y = Cout;
  C->gemm_on(g, s);
  hipEvent_t e0, e1;
  RVCX_HIP(hipEventCreate(&e0));
  RVCX_HIP(hipEventCreate(&e1));
  RVCX_HIP(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) C->gemm_on(g, s);
  RVCX_HIP(hipEventRecord(e1, s));
  RVCX_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  RVCX_HIP(hipEventElapsedTime(&ms, e0, e1));
  *ms_per_launch = ms / iters;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  C->arena.reset();
  API_END
}

int rvcx_op_layernorm_tm(rvcx_ctx* ctx, const float* x, const float* gamma, const float* beta, float* y, float* y_split,
                         int64_t rows, int Cc, float eps) {
  API_BEGIN(ctx)
  const size_t n = (size_t)rows * Cc;
  C->arena.reserve(n * 16 + (64 << 20));
  C->arena.reset();
  float* dx = to_dev(*C, x, n);
  float* dy = C->arena.alloc<float>(n);
  float* ds = C->arena.alloc<float>(n);
  launch_layernorm_tm(dx, Cc, to_dev(*C, gamma, Cc), to_dev(*C, beta, Cc), dy, Cc, (y_split && Cc % 16 == 0) ? ds : nullptr,
                      (long)Cc * 4, rows, Cc, eps, C->dev_err, nullptr, 0, C->stream);
  to_host(*C, y, dy, n);
  if (y_split && Cc % 16 == 0) {
    std::vector<uint16_t> raw(n * 2);
    RVCX_HIP(hipMemcpy(raw.data(), ds, raw.size() * 2, hipMemcpyDeviceToHost));
    decode_xs(raw, rows, Cc, y_split);
  }
  C->arena.reset();
  API_END
}

int rvcx_op_layernorm_c(rvcx_ctx* ctx, const float* x, const float* gamma, const float* beta, float* y, int B,
                        int Cc, int T, float eps) {
  API_BEGIN(ctx)
  const size_t n = (size_t)B * Cc * T;
  C->arena.reserve(n * 8 + (64 << 20));
  C->arena.reset();
  float* dx = to_dev(*C, x, n);
  float* dy = C->arena.alloc<float>(n);
  launch_layernorm_c(dx, to_dev(*C, gamma, Cc), to_dev(*C, beta, Cc), dy, B, Cc, T, eps, nullptr, C->stream);
  to_host(*C, y, dy, n);
  C->arena.reset();
  API_END
}

int rvcx_load_rmvpe(rvcx_ctx* ctx, const rvcx_rmvpe_cfg* cfg, const rvcx_tensor* tbl, int n) {
  API_BEGIN_ONCE(ctx)
  TensorTable t = make_table(tbl, n);
  C->rmvpe = rmvpe_load(*C, *cfg, t);
  (void)bigru_probe_publish();   // the BiGRU's publish assumption, checked once per device while it is idle (gru.hip)
  API_END
}

int rvcx_load_crepe(rvcx_ctx* ctx, const rvcx_tensor* tbl, int n) {
  API_BEGIN_ONCE(ctx)
  TensorTable t = make_table(tbl, n);
  C->crepe = crepe_load(*C, t);
  API_END
}

int64_t rvcx_crepe_frames(int64_t n, int hop) { return hop > 0 ? 1 + n / hop : -1; }

int rvcx_crepe_predict(rvcx_ctx* ctx, const float* x, int64_t n, int hop, float fmin, float fmax, const float* dither,
                       uint64_t seed, float* pitch, float* probs, int32_t* bins) {
  API_BEGIN(ctx)
  if (!C->crepe) fail("crepe not loaded");
  if (!x || !pitch || n <= 0 || hop <= 0) fail("crepe_predict: bad argument");
  const long F = crepe_frames(n, hop);
  C->arena.reserve(crepe_arena_bytes(*C->crepe, n, hop) + (size_t)n * 8 + (size_t)F * (360 + 16) * 4 + (64 << 20));
  C->arena.reset();
  hipStream_t s = C->stream;
  float* dx = any_to_dev(*C, x, (size_t)n);
  std::vector<float> h((size_t)n);
  RVCX_HIP(hipMemcpyAsync(h.data(), dx, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, s));
  RVCX_HIP(hipStreamSynchronize(s));
  const float scale = (float)crepe_quantile999(h);
  if (!(scale > 0.f)) fail("crepe: the signal is silent (its 99.9 % quantile is 0)");
  float* dd = C->arena.alloc<float>((size_t)F);
  if (dither) RVCX_HIP(hipMemcpyAsync(dd, dither, (size_t)F * 4, hipMemcpyDefault, s));
  else launch_crepe_dither(dd, F, seed + 0x63726570ull, 0, s);
  float* dp = C->arena.alloc<float>((size_t)F);
  float* dpr = probs ? C->arena.alloc<float>((size_t)F * 360) : nullptr;
  int* db = bins ? C->arena.alloc<int>((size_t)F) : nullptr;
  crepe_forward(*C, *C->crepe, dx, n, scale, hop, fmin, fmax, dd, dp, dpr, db, s);
  RVCX_HIP(hipMemcpyAsync(pitch, dp, (size_t)F * 4, hipMemcpyDefault, s));
  if (probs) RVCX_HIP(hipMemcpyAsync(probs, dpr, (size_t)F * 360 * 4, hipMemcpyDefault, s));
  if (bins) RVCX_HIP(hipMemcpyAsync(bins, db, (size_t)F * 4, hipMemcpyDefault, s));
  RVCX_HIP(hipStreamSynchronize(s));
  C->check_dev_err();
  C->arena.reset();
  API_END
}

int rvcx_op_crepe_decode(rvcx_ctx* ctx, const float* probs, int64_t F, int batch, float fmin, float fmax, const float* dither,
                         float* pitch, int32_t* bins) {
  API_BEGIN(ctx)
  if (!C->crepe) fail("crepe not loaded");
  if (!probs || !dither || !pitch || F <= 0 || batch <= 0) fail("crepe_decode: bad argument");
  C->arena.reserve((size_t)F * (2 * 360 * 4 + 360 * 2 + 64) + (64 << 20));
  C->arena.reset();
  hipStream_t s = C->stream;
  float* dpr = any_to_dev(*C, probs, (size_t)F * 360);
  float* dd = any_to_dev(*C, dither, (size_t)F);
  float* dp = C->arena.alloc<float>((size_t)F);
  int* db = C->arena.alloc<int>((size_t)F);
  crepe_decode(*C, *C->crepe, dpr, (long)F, batch, fmin, fmax, dd, dp, db, s);
  RVCX_HIP(hipMemcpyAsync(pitch, dp, (size_t)F * 4, hipMemcpyDefault, s));
  if (bins) RVCX_HIP(hipMemcpyAsync(bins, db, (size_t)F * 4, hipMemcpyDefault, s));
  RVCX_HIP(hipStreamSynchronize(s));
  C->arena.reset();
  API_END
}

int rvcx_get_f0_crepe_x(rvcx_ctx* ctx, const float* x, int64_t n, int64_t p_len, const rvcx_params* p, const float* inp_f0,
                        int inp_f0_rows, const float* dither, int64_t dither_n, int32_t* coarse, float* f0) {
  API_BEGIN(ctx)
  if (!p || !x || !coarse || !f0 || n <= 0 || p_len <= 0) fail("get_f0_crepe: bad argument");
  if (!C->crepe) fail("crepe not loaded");
  const std::vector<double> track = f0_file_track(inp_f0, inp_f0 ? inp_f0_rows : 0);
  C->arena.reserve(crepe_arena_bytes(*C->crepe, n, crepe_hop(*p)) + (size_t)n * 8 + (size_t)p_len * 48 + track.size() * 8 +
                   (64 << 20));
  C->arena.reset();
  hipStream_t s = C->stream;
  float* dx = any_to_dev(*C, x, (size_t)n);
  float* fraw = C->arena.alloc<float>((size_t)p_len);
  int* dc = C->arena.alloc<int>((size_t)p_len);
  float* df = C->arena.alloc<float>((size_t)p_len);
  F0Extra ex;
  ex.dither = dither;
  ex.dither_n = dither ? dither_n : 0;
  crepe_f0_device(*C, dx, n, *p, p_len, &ex, fraw, s);
  launch_f0_coarse(fraw, df, dc, (int)p_len, p->pitch, p->f0_min, p->f0_max, s);
  if (!track.empty()) {
    double* rep = C->arena.alloc<double>(track.size());
    RVCX_HIP(hipMemcpyAsync(rep, track.data(), track.size() * 8, hipMemcpyHostToDevice, s));
    launch_f0_override(rep, (int)track.size(), 100 * p->x_pad, df, dc, (int)p_len, p->f0_min, p->f0_max, s);
  }
  RVCX_HIP(hipMemcpyAsync(coarse, dc, (size_t)p_len * 4, hipMemcpyDefault, s));
  RVCX_HIP(hipMemcpyAsync(f0, df, (size_t)p_len * 4, hipMemcpyDefault, s));
  RVCX_HIP(hipStreamSynchronize(s));
  C->check_dev_err();
  C->arena.reset();
  API_END
}

int rvcx_load_fcpe(rvcx_ctx* ctx, const rvcx_fcpe_cfg* cfg, const rvcx_tensor* tbl, int n) {
  API_BEGIN_ONCE(ctx)
  TensorTable t = make_table(tbl, n);
  C->fcpe = fcpe_load(*C, *cfg, t);
  API_END
}

int rvcx_fcpe_f0(rvcx_ctx* ctx, int B, const float* audio, int64_t n, float threshold, float* f0, float* salience,
                 float* mel) {
  API_BEGIN(ctx)
  if (!C->fcpe) fail("fcpe not loaded");
  C->ensure_splitk(B);
  const int F = (int)(1 + n / 160), nb = C->fcpe->cfg.out_dims;
  C->arena.reserve(fcpe_arena_bytes(*C->fcpe, B, n) + (size_t)B * (n + (size_t)F * (nb + 130)) * 4);
  C->arena.reset();
  float* da = any_to_dev(*C, audio, (size_t)B * n);
  float* df0 = C->arena.alloc<float>((size_t)B * F);
  float* ds = salience ? C->arena.alloc<float>((size_t)B * F * nb) : nullptr;
  float* dm = mel ? C->arena.alloc<float>((size_t)B * 128 * F) : nullptr;
  fcpe_forward(*C, *C->fcpe, B, da, n, threshold, df0, ds, dm, C->stream);
  RVCX_HIP(hipMemcpyAsync(f0, df0, (size_t)B * F * 4, hipMemcpyDefault, C->stream));
  if (salience) RVCX_HIP(hipMemcpyAsync(salience, ds, (size_t)B * F * nb * 4, hipMemcpyDefault, C->stream));
  if (mel) RVCX_HIP(hipMemcpyAsync(mel, dm, (size_t)B * 128 * F * 4, hipMemcpyDefault, C->stream));
  RVCX_HIP(hipStreamSynchronize(C->stream));
  C->check_dev_err();
  C->arena.reset();
  API_END
}

int rvcx_op_fcpe_post(rvcx_ctx* ctx, const float* raw, int F_in, int p_len, double pitch, double f0_min, double f0_max,
                      int32_t* coarse, float* f0) {
  API_BEGIN(ctx)
  if (F_in <= 0 || p_len <= 0) fail("fcpe_post: empty track");
  C->arena.reserve((size_t)(F_in + 5L * p_len) * 4 + (64 << 20));
  C->arena.reset();
  float* dr = any_to_dev(*C, raw, (size_t)F_in);
  int* dc = C->arena.alloc<int>((size_t)p_len);
  float* df = C->arena.alloc<float>((size_t)p_len);
  fcpe_post_coarse(*C, dr, 1, F_in, p_len, df, dc, p_len, pitch, f0_min, f0_max, C->stream);
  RVCX_HIP(hipMemcpyAsync(coarse, dc, (size_t)p_len * 4, hipMemcpyDefault, C->stream));
  RVCX_HIP(hipMemcpyAsync(f0, df, (size_t)p_len * 4, hipMemcpyDefault, C->stream));
  RVCX_HIP(hipStreamSynchronize(C->stream));
  C->arena.reset();
  API_END
}

int rvcx_get_f0_fcpe_x(rvcx_ctx* ctx, const float* x, int64_t n, int64_t p_len, const rvcx_params* p, int32_t* coarse,
                       float* f0) {
  API_BEGIN(ctx)
  if (!C->fcpe) fail("fcpe not loaded");
  if (p_len <= 0) fail("get_f0: p_len must be positive");
  const long F = 1 + n / 160;
  C->arena.reserve(fcpe_arena_bytes(*C->fcpe, 1, n) + (size_t)n * 8 + (size_t)(F + p_len) * 32 + (64 << 20));
  C->arena.reset();
  float* dx = any_to_dev(*C, x, (size_t)n);
  float* fraw = C->arena.alloc<float>((size_t)F);
  int* dc = C->arena.alloc<int>((size_t)p_len);
  float* df = C->arena.alloc<float>((size_t)p_len);
  fcpe_forward(*C, *C->fcpe, 1, dx, n, 0.03f, fraw, nullptr, nullptr, C->stream);
  fcpe_post_coarse(*C, fraw, 1, (int)F, (int)p_len, df, dc, p_len, p->pitch, p->f0_min, p->f0_max, C->stream);
  RVCX_HIP(hipMemcpyAsync(coarse, dc, (size_t)p_len * 4, hipMemcpyDefault, C->stream));
  RVCX_HIP(hipMemcpyAsync(f0, df, (size_t)p_len * 4, hipMemcpyDefault, C->stream));
  RVCX_HIP(hipStreamSynchronize(C->stream));
  C->check_dev_err();
  C->arena.reset();
  API_END
}

int rvcx_load_hubert(rvcx_ctx* ctx, const rvcx_hubert_cfg* cfg, const rvcx_tensor* tbl, int n) {
  API_BEGIN_ONCE(ctx)
  TensorTable t = make_table(tbl, n);
  C->hubert = hubert_load(*C, *cfg, t);
  API_END
}

int rvcx_rmvpe_f0(rvcx_ctx* ctx, int B, const float* audio, int64_t n, float thred, float f0_min, float f0_max,
                  float* f0, float* hidden) {
  API_BEGIN(ctx)
  if (!C->rmvpe) fail("rmvpe not loaded");
  C->ensure_splitk(B);
  const int F = (int)(1 + n / 160);
  C->arena.reserve(rmvpe_arena_bytes(*C->rmvpe, B, n) + (size_t)B * (n + (size_t)F * 362) * 4);
  C->arena.reset();
  float* da = any_to_dev(*C, audio, (size_t)B * n);
  float* df0 = C->arena.alloc<float>((size_t)B * F);
  float* dh = hidden ? C->arena.alloc<float>((size_t)B * F * 360) : nullptr;
  rmvpe_forward(*C, *C->rmvpe, B, da, n, thred, f0_min, f0_max, df0, dh, C->stream);
  RVCX_HIP(hipMemcpyAsync(f0, df0, (size_t)B * F * 4, hipMemcpyDefault, C->stream));
  if (hidden) RVCX_HIP(hipMemcpyAsync(hidden, dh, (size_t)B * F * 360 * 4, hipMemcpyDefault, C->stream));
  RVCX_HIP(hipStreamSynchronize(C->stream));
  C->check_dev_err();
  C->arena.reset();
  API_END
}

int rvcx_rmvpe_mel(rvcx_ctx* ctx, int B, const float* audio, int64_t n, float* mel) {
  API_BEGIN(ctx)
  if (!C->rmvpe) fail("rmvpe not loaded");
  C->ensure_splitk(B);
  const int F = (int)(1 + n / 160);
  C->arena.reserve(rmvpe_arena_bytes(*C->rmvpe, B, n) + (size_t)B * (n + (size_t)F * 130) * 4);
  C->arena.reset();
  float* da = any_to_dev(*C, audio, (size_t)B * n);
  float* df0 = C->arena.alloc<float>((size_t)B * F);
  float* dm = C->arena.alloc<float>((size_t)B * 128 * F);
  rmvpe_forward(*C, *C->rmvpe, B, da, n, 0.03f, 50.f, 1100.f, df0, nullptr, C->stream, dm);
  RVCX_HIP(hipMemcpyAsync(mel, dm, (size_t)B * 128 * F * 4, hipMemcpyDefault, C->stream));
  RVCX_HIP(hipStreamSynchronize(C->stream));
  C->check_dev_err();
  C->arena.reset();
  API_END
}

int rvcx_hubert_frames(rvcx_ctx* ctx, int64_t n) {
  CtxLock ctx_guard_ = lock_ctx(ctx);
  if (!ctx || !ctx->c.hubert) return -1;
  return hubert_frames(*ctx->c.hubert, n);
}

int rvcx_hubert_features(rvcx_ctx* ctx, int B, const float* wav, int64_t n, int output_layer, float* feats) {
  API_BEGIN(ctx)
  if (!C->hubert) fail("hubert not loaded");
  const int T = hubert_frames(*C->hubert, n), E = C->hubert->cfg.embed_dim;
  if (T <= 0) fail("hubert: input too short");
  C->ensure_splitk(B);
  C->arena.reserve(hubert_arena_bytes(*C->hubert, B, n) + (size_t)B * (n + (size_t)2 * T * E) * 4);
  C->arena.reset();
  float* dw = any_to_dev(*C, wav, (size_t)B * n);
  float* fct = C->arena.alloc<float>((size_t)B * E * T);
  float* ftc = C->arena.alloc<float>((size_t)B * E * T);
  hipStream_t st = C->stream;
  hubert_forward(*C, *C->hubert, B, dw, n, output_layer, fct, st);
  launch_transpose(fct, ftc, B, E, T, st);   // (B,E,T) -> (B,T,E)
  RVCX_HIP(hipMemcpyAsync(feats, ftc, (size_t)B * E * T * 4, hipMemcpyDefault, st));
  RVCX_HIP(hipStreamSynchronize(st));
  C->arena.reset();
  API_END
}

int rvcx_op_bigru(rvcx_ctx* ctx, const float* x, const float* w_ih, const float* w_hh, const float* b_ih,
                  const float* b_hh, const float* w_ih_r, const float* w_hh_r, const float* b_ih_r,
                  const float* b_hh_r, float* y, int B, int T, int I, int H) {
  API_BEGIN(ctx)
  TEMP_REGION(C);
  const int H3 = 3 * H;
  C->arena.reserve(((size_t)B * T * (2 * I + 2 * H3 + 4 * H)) * 4 + (64 << 20));
  C->arena.reset();
  std::vector<float> wih((size_t)2 * H3 * I), bih((size_t)2 * H3), whh_t((size_t)2 * H * H3), bhh((size_t)2 * H3);
  const float* wi[2] = {w_ih, w_ih_r};
  const float* wh[2] = {w_hh, w_hh_r};
  const float* bi[2] = {b_ih, b_ih_r};
  const float* bh[2] = {b_hh, b_hh_r};
  for (int d = 0; d < 2; ++d) {
    std::memcpy(&wih[(size_t)d * H3 * I], wi[d], (size_t)H3 * I * 4);
    std::memcpy(&bih[(size_t)d * H3], bi[d], (size_t)H3 * 4);
    std::memcpy(&bhh[(size_t)d * H3], bh[d], (size_t)H3 * 4);
    for (int j = 0; j < H3; ++j)
      for (int k = 0; k < H; ++k) whh_t[((size_t)d * H + k) * H3 + j] = wh[d][(size_t)j * H + k];
  }
  ConvW Wih = make_conv(*C, wih.data(), bih.data(), 2 * H3, I, 1, 1);
  const float* dwhh = C->slab.upload(whh_t);
  const float* dbhh = C->slab.upload(bhh);
  float* dx = to_dev(*C, x, (size_t)B * T * I);
  float* dxt = C->arena.alloc<float>((size_t)B * T * I);
  launch_transpose(dx, dxt, B, T, I, C->stream);   // (B,T,I) -> (B,I,T)
  float* gi = C->arena.alloc<float>((size_t)B * T * 2 * H3);
  ConvArgs a = conv1d_args(Wih, dxt, gi, B, T, T);
  a.out_mode = OUT_TRANSPOSED;
  a.y_bs = (long)T * 2 * H3;
  a.y_cs = 2 * H3;
  C->conv(a);
  float* gy = C->arena.alloc<float>((size_t)B * 2 * H * T);
  void* gscr = C->arena.alloc<unsigned long long>(bigru_scratch_bytes(B) / 8);
  launch_bigru(gi, dwhh, dbhh, gy, B, T, H, gscr, C->dev_err, C->stream);
  float* gyt = C->arena.alloc<float>((size_t)B * 2 * H * T);
  launch_transpose(gy, gyt, B, 2 * H, T, C->stream);  // (B,2H,T) -> (B,T,2H)
  to_host(*C, y, gyt, (size_t)B * T * 2 * H);
  C->check_dev_err();
  C->arena.reset();
  API_END
}

int rvcx_load_index(rvcx_ctx* ctx, const float* big_npy, int64_t n, int dim) {
  API_BEGIN_ONCE(ctx)
  if (!big_npy || n == 0) {
    C->index.reset();
  } else {
      C->index = index_load(*C, big_npy, n, dim);
  }
  API_END
}

int rvcx_load_index_ivf(rvcx_ctx* ctx, const float* big_npy, int64_t n, int dim, const float* centroids, int nlist,
                        const int32_t* assign, int nprobe) {
  API_BEGIN_ONCE(ctx)
  if (!big_npy || n <= 0 || !centroids || nlist <= 0 || !assign) fail("load_index_ivf: null argument");
  if (nprobe != 1) fail("load_index_ivf: only nprobe = 1 (what RVC index files carry) is implemented");
  C->index = index_load(*C, big_npy, n, dim, centroids, nlist, assign);
  API_END
}

int rvcx_index_blend(rvcx_ctx* ctx, float* feats, int T, float index_rate, int64_t* ids, float* dist) {
  API_BEGIN(ctx)
  if (!C->index) fail("index not loaded");
  const int D = C->index->dim;
  C->arena.reserve(index_arena_bytes(*C->index, T) + (size_t)T * (2 * D + 24) * 4 + (64 << 20));
  C->arena.reset();
  float* f = any_to_dev(*C, feats, (size_t)T * D);
  float* fct = C->arena.alloc<float>((size_t)T * D);
  launch_transpose(f, fct, 1, T, D, C->stream);
  int64_t* dids = C->arena.alloc<int64_t>((size_t)T * 8);
  float* ddist = C->arena.alloc<float>((size_t)T * 8);
  index_blend(*C, *C->index, fct, T, index_rate, dids, ddist, C->stream);
  launch_transpose(fct, f, 1, D, T, C->stream);
  RVCX_HIP(hipMemcpyAsync(feats, f, (size_t)T * D * 4, hipMemcpyDefault, C->stream));
  if (ids) RVCX_HIP(hipMemcpyAsync(ids, dids, (size_t)T * 8 * 8, hipMemcpyDefault, C->stream));
  if (dist) RVCX_HIP(hipMemcpyAsync(dist, ddist, (size_t)T * 8 * 4, hipMemcpyDefault, C->stream));
  RVCX_HIP(hipStreamSynchronize(C->stream));
  C->arena.reset();
  API_END
}

int64_t rvcx_out_len(rvcx_ctx* ctx, int model_id, int64_t n, const rvcx_params* p) {
  CtxLock ctx_guard_ = lock_ctx(ctx);
  if (!ctx || model_id < 0 || model_id >= (int)ctx->c.synths.size() || !ctx->c.synths[model_id]) return -1;
  return out_capacity(*ctx->c.synths[model_id], n, *p);
}

int64_t rvcx_noise_len(rvcx_ctx* ctx, int model_id, int64_t n, const rvcx_params* p) {
  CtxLock ctx_guard_ = lock_ctx(ctx);
  if (!ctx || model_id < 0 || model_id >= (int)ctx->c.synths.size() || !ctx->c.synths[model_id]) return -1;
  return noise_len_for(ctx->c, *ctx->c.synths[model_id], n, *p);
}

static int convert_impl(rvcx_ctx* ctx, int model_id, int B, const float* const* wav32, const double* const* wav64,
                        const int64_t* n, const rvcx_params* p, const float* const* noise, int16_t* const* out,
                        float* const* out_f32, int64_t* out_n, const rvcx_utt_extra* extra = nullptr) {
  API_BEGIN(ctx)
  (void)get_synth(*C, model_id);
  if (B < 0 || (B > 0 && (!n || !p || !out || (!wav32 && !wav64)))) fail("convert_batch: null argument");
  std::vector<UttIO> ios((size_t)B);
  for (int i = 0; i < B; ++i) {
    UttIO& u = ios[i];
    u.wav = wav32 ? wav32[i] : nullptr;
    u.wav64 = wav64 ? wav64[i] : nullptr;
    u.n = n[i];
    u.noise = noise ? noise[i] : nullptr;
    u.out = out[i];
    u.out_f32 = out_f32 ? out_f32[i] : nullptr;
    u.seed_offset = i;
    if (extra) {
      u.inp_f0 = extra[i].inp_f0;
      u.inp_f0_rows = extra[i].inp_f0 ? extra[i].inp_f0_rows : 0;
      u.crepe_dither = extra[i].crepe_dither;
      u.crepe_dither_n = extra[i].crepe_dither ? extra[i].crepe_dither_n : 0;
    }
    if (!(u.wav || u.wav64) || !u.out) fail("convert_batch: null buffer for utterance " + std::to_string(i));
  }
  static const bool timing = !getenv("RVCX_STAGE_TIMING") || atoi(getenv("RVCX_STAGE_TIMING")) != 0;
  float ms[9] = {0};
  convert_batch(*C, model_id, ios, *p, timing ? ms : nullptr);
  C->check_dev_err();
  for (int k = 0; k < 9; ++k) C->timing[k] = ms[k];
  if (out_n)
    for (int i = 0; i < B; ++i) out_n[i] = ios[i].out_n;
  C->arena.reset();
  API_END
}

int rvcx_convert_batch(rvcx_ctx* ctx, int model_id, int B, const float* const* wav16k, const int64_t* n,
                       const rvcx_params* p, const float* const* noise, int16_t* const* out, float* const* out_f32,
                       int64_t* out_n) {
  CtxLock ctx_guard_ = lock_ctx(ctx);
  return convert_impl(ctx, model_id, B, wav16k, nullptr, n, p, noise, out, out_f32, out_n);
}

int rvcx_convert_batch_f64(rvcx_ctx* ctx, int model_id, int B, const double* const* wav16k, const int64_t* n,
                           const rvcx_params* p, const float* const* noise, int16_t* const* out,
                           float* const* out_f32, int64_t* out_n) {
  CtxLock ctx_guard_ = lock_ctx(ctx);
  return convert_impl(ctx, model_id, B, nullptr, wav16k, n, p, noise, out, out_f32, out_n);
}

int rvcx_convert_batch_ex(rvcx_ctx* ctx, int model_id, int B, const void* const* wav16k, int wav_is_f64,
                          const int64_t* n, const rvcx_params* p, const float* const* noise,
                          const rvcx_utt_extra* extra, int16_t* const* out, float* const* out_f32, int64_t* out_n) {
  CtxLock ctx_guard_ = lock_ctx(ctx);
  return convert_impl(ctx, model_id, B, wav_is_f64 ? nullptr : reinterpret_cast<const float* const*>(wav16k),
                      wav_is_f64 ? reinterpret_cast<const double* const*>(wav16k) : nullptr, n, p, noise, out, out_f32,
                      out_n, extra);
}

int rvcx_micro_batch(rvcx_ctx* ctx, int model_id, int64_t n, const rvcx_params* p) {
  CtxLock ctx_guard_ = lock_ctx(ctx);
  if (!ctx || !p || model_id < 0 || model_id >= (int)ctx->c.synths.size() || !ctx->c.synths[model_id] ||
      !ctx->c.hubert)
    return -1;
  try {
    return convert_micro_batch(ctx->c, model_id, n, *p);
  } catch (const std::exception& e) {
    ctx->c.last_error = e.what();
    return -1;
  }
}

int64_t rvcx_bucket_length(rvcx_ctx* ctx, int model_id, int64_t n, const rvcx_params* p) {
  CtxLock ctx_guard_ = lock_ctx(ctx);
  if (!ctx || !p || model_id < 0 || model_id >= (int)ctx->c.synths.size() || !ctx->c.synths[model_id]) return -1;
  try {
    return bucket_length(n, *p, make_geometry(*p, ctx->c.synths[model_id]->cfg.sr));
  } catch (const std::exception& e) {
    ctx->c.last_error = e.what();
    return -1;
  }
}

int rvcx_last_micro_batches(rvcx_ctx* ctx, int32_t* counts, int cap) {
  CtxLock ctx_guard_ = lock_ctx(ctx);
  if (!ctx) return -1;
  const auto& v = ctx->c.last_mbs;
  for (int i = 0; i < (int)v.size() && i < cap && counts; ++i) counts[i] = v[i];
  return (int)v.size();
}

int rvcx_get_f0(rvcx_ctx* ctx, const float* wav16k, int64_t n, const rvcx_params* p, int32_t* coarse, float* f0,
                int64_t* p_len) {
  API_BEGIN(ctx)
  const long t_pad = 16000L * p->x_pad, n_pad = n + 2 * t_pad;
  C->arena.reserve(f0_arena_bytes(*C, *p, 1, n_pad) + (size_t)n_pad * 48 + (64 << 20));
  C->arena.reset();
  float* dw = any_to_dev(*C, wav16k, (size_t)n);
  double* ext = C->arena.alloc<double>(highpass_ext_doubles(n));
  float* a32 = C->arena.alloc<float>((size_t)n);
  launch_highpass(dw, nullptr, ext, nullptr, a32, n, C->stream);
  float* apad = C->arena.alloc<float>((size_t)n_pad);
  launch_reflect_pad(a32, apad, 1, (int)n, (int)t_pad, n_pad, C->stream);
  const long pl = n_pad / 160;
  int* dc = C->arena.alloc<int>((size_t)pl + 8);
  float* df = C->arena.alloc<float>((size_t)pl + 8);
  get_f0_device(*C, apad, n_pad, *p, dc, df, C->stream);
  RVCX_HIP(hipMemcpyAsync(coarse, dc, (size_t)pl * 4, hipMemcpyDefault, C->stream));
  RVCX_HIP(hipMemcpyAsync(f0, df, (size_t)pl * 4, hipMemcpyDefault, C->stream));
  RVCX_HIP(hipStreamSynchronize(C->stream));
  C->check_dev_err();
  *p_len = pl;
  C->arena.reset();
  API_END
}

int rvcx_get_f0_x(rvcx_ctx* ctx, const float* x, int64_t n, const rvcx_params* p, int32_t* coarse, float* f0) {
  API_BEGIN(ctx)
  if (!C->rmvpe) fail("rmvpe not loaded");
  const long F = 1 + n / 160;
  C->arena.reserve(rmvpe_arena_bytes(*C->rmvpe, 1, n) + (size_t)n * 8 + (size_t)F * 32 + (64 << 20));
  C->arena.reset();
  float* dx = any_to_dev(*C, x, (size_t)n);
  float* fraw = C->arena.alloc<float>((size_t)F);
  int* dc = C->arena.alloc<int>((size_t)F);
  float* df = C->arena.alloc<float>((size_t)F);
  rmvpe_forward(*C, *C->rmvpe, 1, dx, n, 0.03f, p->f0_min, p->f0_max, fraw, nullptr, C->stream);
  launch_f0_coarse(fraw, df, dc, (int)F, p->pitch, p->f0_min, p->f0_max, C->stream);
  RVCX_HIP(hipMemcpyAsync(coarse, dc, (size_t)F * 4, hipMemcpyDefault, C->stream));
  RVCX_HIP(hipMemcpyAsync(f0, df, (size_t)F * 4, hipMemcpyDefault, C->stream));
  RVCX_HIP(hipStreamSynchronize(C->stream));
  C->check_dev_err();
  C->arena.reset();
  API_END
}

int rvcx_get_f0_x_ex(rvcx_ctx* ctx, const float* x, int64_t n, int64_t p_len, const rvcx_params* p, const float* inp_f0,
                     int inp_f0_rows, int32_t* coarse, float* f0, int64_t* frames) {
  API_BEGIN(ctx)
  if (!p || !x) fail("get_f0: null argument");
  check_f0_backend(*C, *p);
  if (p->f0_method == RVCX_F0_CREPE) fail("get_f0: mangio-crepe takes its dither through rvcx_get_f0_crepe_x");
  const bool fcpe = p->f0_method == RVCX_F0_FCPE;
  const long F = fcpe ? (long)p_len : 1 + n / 160;        // rmvpe+: un-truncated; fcpe: compute_f0 resizes to p_len
  if (F <= 0) fail("get_f0: p_len must be positive");
  const std::vector<double> track = f0_file_track(inp_f0, inp_f0 ? inp_f0_rows : 0);
  C->arena.reserve(f0_arena_bytes(*C, *p, 1, n) + (size_t)n * 8 + (size_t)(F + n / 160 + 8) * 48 + track.size() * 8 +
                   (64 << 20));
  C->arena.reset();
  hipStream_t s = C->stream;
  float* dx = any_to_dev(*C, x, (size_t)n);
  float* fraw = C->arena.alloc<float>((size_t)(1 + n / 160));
  int* dc = C->arena.alloc<int>((size_t)F);
  float* df = C->arena.alloc<float>((size_t)F);
  if (fcpe) {
    fcpe_forward(*C, *C->fcpe, 1, dx, n, 0.03f, fraw, nullptr, nullptr, s);
    fcpe_post_coarse(*C, fraw, 1, (int)(1 + n / 160), (int)F, df, dc, F, p->pitch, p->f0_min, p->f0_max, s);
  } else {
    rmvpe_forward(*C, *C->rmvpe, 1, dx, n, 0.03f, p->f0_min, p->f0_max, fraw, nullptr, s);
    launch_f0_coarse(fraw, df, dc, (int)F, p->pitch, p->f0_min, p->f0_max, s);
  }
  if (!track.empty()) {
    double* rep = C->arena.alloc<double>(track.size());
    RVCX_HIP(hipMemcpyAsync(rep, track.data(), track.size() * 8, hipMemcpyHostToDevice, s));
    launch_f0_override(rep, (int)track.size(), 100 * p->x_pad, df, dc, (int)F, p->f0_min, p->f0_max, s);
  }
  RVCX_HIP(hipMemcpyAsync(coarse, dc, (size_t)F * 4, hipMemcpyDefault, s));
  RVCX_HIP(hipMemcpyAsync(f0, df, (size_t)F * 4, hipMemcpyDefault, s));
  RVCX_HIP(hipStreamSynchronize(s));
  C->check_dev_err();
  if (frames) *frames = F;
  C->arena.reset();
  API_END
}

int rvcx_f0_file_track(const float* inp_f0, int rows, double* track, int cap) {
  try {
    const std::vector<double> t = f0_file_track(inp_f0, rows);
    for (size_t i = 0; i < t.size() && (int)i < cap; ++i) track[i] = t[i];
    return (int)t.size();
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return -1;
  }
}

int64_t rvcx_resample_len(int64_t n, int sr_in, int sr_out) {
  return (sr_in > 0 && sr_out > 0 && n >= 0) ? (int64_t)resample_out_len((long)n, sr_in, sr_out) : -1;
}

int rvcx_resample_f64_kind(rvcx_ctx* ctx, const double* x, int64_t frames, int channels, int sr_in, int sr_out, int kind,
                           double* y) {
  API_BEGIN(ctx)
  if (!x || !y || frames <= 0 || channels < 1 || sr_in <= 0 || sr_out <= 0) fail("resample: bad argument");
  const long n_out = resample_out_len((long)frames, sr_in, sr_out);
  C->arena.reserve(((size_t)frames * channels + (size_t)n_out) * 8 + ((size_t)8 << 20));
  C->arena.reset();
  hipStream_t s = C->stream;
  double* dx = any_to_dev(*C, x, (size_t)frames * channels);
  double* dy = C->arena.alloc<double>((size_t)std::max<long>(n_out, 1));
  const ResampleFilter f = make_resample_filter(C->arena, sr_in, sr_out, s, kind);
  launch_resample_f64(f, dx, (long)frames, channels, dy, n_out, s);
  RVCX_HIP(hipMemcpyAsync(y, dy, (size_t)n_out * 8, hipMemcpyDefault, s));
  RVCX_HIP(hipStreamSynchronize(s));
  C->arena.reset();
  API_END
}

int rvcx_resample_f64(rvcx_ctx* ctx, const double* x, int64_t frames, int channels, int sr_in, int sr_out, double* y) {
  return rvcx_resample_f64_kind(ctx, x, frames, channels, sr_in, sr_out, -1, y);
}

int rvcx_vc_frames(rvcx_ctx* ctx, int64_t n) {
  CtxLock ctx_guard_ = lock_ctx(ctx);
  if (!ctx || !ctx->c.hubert) return -1;
  const int Th = hubert_frames(*ctx->c.hubert, n);
  if (Th <= 0) return -1;
  return (int)std::min<long>(n / 160, 2L * Th);
}

int rvcx_vc(rvcx_ctx* ctx, int model_id, const float* audio0, int64_t n, const int32_t* pitch, const float* pitchf,
            int n_pitch, int sid, float index_rate, float protect, const float* z_noise, const float* src_noise,
            uint64_t seed, float* out, int64_t* out_n) {
  API_BEGIN(ctx)
  SynthModel& M = get_synth(*C, model_id);
  if (!C->hubert) fail("hubert not loaded");
  if (!pitch || !pitchf) fail("vc: non-f0 models cannot run in the reference either (generators.py:57-77)");
  const int E = M.cfg.input_dim, inter = M.cfg.inter_channels;      // v2: the HuBERT's embed_dim; v1: its final_proj width
  RVCX_CHECK(E == C->hubert->cfg.embed_dim || (C->hubert->has_final_proj && E == C->hubert->final_proj.cout),
             "the voice model's input_dim is neither the HuBERT's embed_dim (v2) nor its final_proj width (v1)");
  const int Th = hubert_frames(*C->hubert, n);
  RVCX_CHECK(Th > 0, "vc: chunk too short");
  const int T = (int)std::min<long>(n / 160, 2L * Th);       // p_len clamp, pipeline.py:257-262
  RVCX_CHECK(n_pitch >= T, "vc: pitch / pitchf shorter than the chunk's frame count");
  const size_t nz = (size_t)inter * T, nsrc = (size_t)T * M.upp;
  const bool use_index = C->index && index_rate != 0.f, use_protect = protect < 0.5f;
  size_t need = hubert_arena_bytes(*C->hubert, 1, n) + synth_arena_bytes(M, 1, T) + (size_t)n * 4 +
                ((size_t)T * ((size_t)3 * E + inter + 3 * M.upp + 16)) * 4;
  if (use_index) need += index_arena_bytes(*C->index, Th);
  C->arena.reserve(need);
  C->arena.reset();
  hipStream_t s = C->stream;
  float* dw = any_to_dev(*C, audio0, (size_t)n);
  int* dp = any_to_dev<int>(*C, pitch, (size_t)T);
  float* dpf = any_to_dev(*C, pitchf, (size_t)T);
  float* feats = C->arena.alloc<float>((size_t)E * Th);
  {
    const size_t mk = C->arena.mark();
    hubert_features_for(*C, *C->hubert, E, 1, dw, n, feats, s);     // v2: layer 12; v1: final_proj(layer 9)
    C->arena.reset(mk);
  }
  const float* feats0 = feats;
  if (use_index) {
    if (use_protect) {
      float* keep = C->arena.alloc<float>((size_t)E * Th);
      RVCX_HIP(hipMemcpyAsync(keep, feats, (size_t)E * Th * 4, hipMemcpyDeviceToDevice, s));
      feats0 = keep;
    }
    const size_t mk = C->arena.mark();
    index_blend(*C, *C->index, feats, Th, index_rate, nullptr, nullptr, s);
    C->arena.reset(mk);
  }
  float* phone = C->arena.alloc<float>((size_t)E * T);
  launch_upsample_protect(feats, feats0, dpf, phone, E, Th, T, protect, use_protect ? 1 : 0, s);
  float* zn = C->arena.alloc<float>(nz);
  float* sn = C->arena.alloc<float>(nsrc);
  if (z_noise) RVCX_HIP(hipMemcpyAsync(zn, z_noise, nz * 4, hipMemcpyDefault, s));
  else launch_randn(zn, nz, seed, 0, s);
  if (src_noise) RVCX_HIP(hipMemcpyAsync(sn, src_noise, nsrc * 4, hipMemcpyDefault, s));
  else launch_randn(sn, nsrc, seed, (uint64_t)1 << 35, s);
  float* wavout = C->arena.alloc<float>(nsrc);
  SynthIO io;
  io.B = 1;
  io.T = T;
  io.phone_ct = phone;
  io.pitch = dp;
  io.pitchf = dpf;
  io.sid_host = &sid;
  io.z_noise = zn;
  io.src_noise = sn;
  io.out = wavout;
  synth_forward(*C, M, io, nullptr);
  RVCX_HIP(hipMemcpyAsync(out, wavout, nsrc * 4, hipMemcpyDefault, s));
  RVCX_HIP(hipStreamSynchronize(s));
  if (out_n) *out_n = (int64_t)nsrc;
  C->arena.reset();
  API_END
}

// The per-launch profile is PROCESS-wide state (conv.hip): the two hooks below serialise against each other on one mutex, but
// launches of ANOTHER context that run while a profile is open are recorded into it too (rvcx.h says so).
static std::mutex g_profile_mu;

int rvcx_conv_profile(rvcx_ctx* ctx, int begin, int64_t* launches, double* flops, double* ms, int32_t* bm,
                      int32_t* bn, int32_t* kind, int cap) {
  std::lock_guard<std::mutex> prof_guard(g_profile_mu);
  API_BEGIN_ONCE(ctx)
  C->serial = begin != 0 || C->serial_env;
  if (begin) {
    conv_profile_begin();
  } else {
    ConvProfile p;
    conv_profile_end(&p);
    for (int t = 0; t < cap && t < ConvProfile::kMaxTiles; ++t) {
      launches[t] = p.launches[t];
      flops[t] = p.flops[t];
      ms[t] = p.ms[t];
      bm[t] = p.bm[t];
      bn[t] = p.bn[t];
      kind[t] = p.halo[t];
    }
  }
  API_END
}

const char* rvcx_conv_profile_csv(rvcx_ctx*) {
  std::lock_guard<std::mutex> g(g_profile_mu);
  return conv_profile_csv();
}

int rvcx_last_timing(rvcx_ctx* ctx, float* ms9) {
  CtxLock ctx_guard_ = lock_ctx(ctx);
  if (!ctx) return -1;
  for (int k = 0; k < 9; ++k) ms9[k] = ctx->c.timing[k];
  return 0;
}

int rvcx_op_highpass(rvcx_ctx* ctx, const double* x, double* y, int64_t n) {
  API_BEGIN(ctx)
  C->arena.reserve((size_t)n * 32 + (64 << 20));
  C->arena.reset();
  double* dx = C->arena.alloc<double>((size_t)n);
  RVCX_HIP(hipMemcpyAsync(dx, x, (size_t)n * 8, hipMemcpyHostToDevice, C->stream));
  double* ext = C->arena.alloc<double>(highpass_ext_doubles(n));
  double* dy = C->arena.alloc<double>((size_t)n);
  launch_highpass(nullptr, dx, ext, dy, nullptr, n, C->stream);
  RVCX_HIP(hipMemcpyAsync(y, dy, (size_t)n * 8, hipMemcpyDeviceToHost, C->stream));
  RVCX_HIP(hipStreamSynchronize(C->stream));
  C->arena.reset();
  API_END
}

// ------------------------------------------------------------------------------------------
// entry points not implemented yet return an error (never a silent fallback)
// ------------------------------------------------------------------------------------------
#define NOT_IMPL(ctxp, name)                    \
  do {                                          \
    g_last_error = name ": not implemented";    \
    if (ctxp) (ctxp)->c.last_error = g_last_error; \
    return -1;                                  \
  } while (0)

int rvcx_rmvpe_frames(int64_t n) { return (int)(1 + n / 160); }
int rvcx_fcpe_frames(int64_t n) { return (int)(1 + n / 160); }

}  // extern "C"
