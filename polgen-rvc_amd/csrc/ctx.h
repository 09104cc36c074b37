// rvcx context: one per GPU.  Owns the stream, the activation arena, the weight slab and
// the loaded models.
#pragma once
#include <memory>

#include "common.h"
#include "conv.h"

namespace rvcx {

struct SynthModel;
struct RmvpeModel;
struct HubertModel;
struct IndexData;

// contiguous slab for folded/packed weights (so the whole set can be RCCL-broadcast)
class WeightSlab {
 public:
  ~WeightSlab() {
    if (base_) (void)hipFree(base_);
  }
  void init(size_t bytes) {
    if (base_) return;
    RVCX_HIP(hipMalloc(&base_, bytes));
    cap_ = bytes;
  }
  float* upload(const std::vector<float>& h) { return upload(h.data(), h.size()); }
  float* upload(const float* h, size_t n) {
    size_t bytes = (std::max<size_t>(n, 1) * sizeof(float) + 255) & ~size_t(255);
    if (off_ + bytes > cap_) fail("weight slab exhausted");
    float* d = reinterpret_cast<float*>(static_cast<char*>(base_) + off_);
    if (n) RVCX_HIP(hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice));
    off_ += bytes;
    return d;
  }
  void* base() const { return base_; }
  size_t used() const { return off_; }

 private:
  void* base_ = nullptr;
  size_t cap_ = 0, off_ = 0;
};

struct StageTimer {
  hipEvent_t ev[16];
  bool made = false;
  void make() {
    if (made) return;
    for (auto& e : ev) RVCX_HIP(hipEventCreate(&e));
    made = true;
  }
};

struct Ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;  // second stream: RMVPE runs beside HuBERT
  hipStream_t stream_h = nullptr; // optional CU-masked stream for HuBERT (RVCX_HUBERT_CUS): leaves CUs free for RMVPE
  hipEvent_t ev_hub = nullptr;
  hipStream_t aux[2] = {nullptr, nullptr};   // the three ResBlocks of an NSF stage run side by side
  hipEvent_t ev_aux[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipEvent_t ev_src[2] = {nullptr, nullptr};   // NSF source branch (sine + noise convs) beside TextEncoder/flow
  Arena arena;
  Arena arena_f0;                 // RMVPE workspace: lives on stream2 across the main stream's arena resets
  WeightSlab slab;
  std::string last_error;
  double flops = 0.0;
  bool resblock_streams = false;  // run the ResBlocks of an NSF stage on 3 streams
  bool serial = false;            // profiling: keep every launch on the main stream (true per-kernel times)
  int* dev_err = nullptr;         // device flag: a kernel gave up waiting (checked after each API call)
  void check_dev_err();
  float timing[9] = {0};
  StageTimer timer;
  std::unique_ptr<HubertModel> hubert;
  std::unique_ptr<RmvpeModel> rmvpe;
  std::vector<std::unique_ptr<SynthModel>> synths;
  std::unique_ptr<IndexData> index;
  Ctx();
  ~Ctx();
  // split-K partial-sum scratch, one per stream (RMVPE and HuBERT run concurrently)
  float* splitk_buf[4] = {nullptr, nullptr, nullptr, nullptr};
  static constexpr long kSplitKFloats = 32L << 20;   // 128 MiB each
  void conv(const ConvArgs& a) { conv_on(a, stream); }
  void conv_on(ConvArgs a, hipStream_t s) {
    flops += conv_flops(a);
    const int si = (s == stream2) ? 1 : (s == aux[0] ? 2 : (s == aux[1] ? 3 : 0));
    if (!splitk_buf[si]) RVCX_HIP(hipMalloc(&splitk_buf[si], kSplitKFloats * sizeof(float)));
    a.part = splitk_buf[si];
    a.part_cap = kSplitKFloats;
    launch_conv(a, s);
  }
};

// packed conv layer living in the weight slab
struct ConvW {
  const float* w = nullptr;
  const void* w_h3 = nullptr;     // fp16 hi/lo split image (layers that may run on conv_h3_kernel)
  const float* bias = nullptr;
  int cin = 0, cout = 0, k = 1, groups = 1;
  int cin_gp = 0, cout_gp = 0;
};

ConvW make_conv(Ctx& c, const float* w, const float* bias, int cout, int cin_g, int k, int groups = 1,
               bool h3 = true);

// fill the channel/pad fields of ConvArgs from a packed layer
inline void conv_set_weights(ConvArgs& a, const ConvW& w) {
  a.w = w.w;
  a.w_h3 = w.w_h3;
  a.bias = w.bias;
  a.groups = w.groups;
  a.Cin_g = w.cin / w.groups;
  a.Cout_g = w.cout / w.groups;
  a.Cin_gp = w.cin_gp;
  a.Cout_gp = w.cout_gp;
  a.ksize = w.k;
  a.kw = w.k;
}

}  // namespace rvcx
