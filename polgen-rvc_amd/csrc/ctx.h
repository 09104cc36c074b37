// rvcx context: one per GPU.  Owns the stream, the activation arena, the weight slab and
// the loaded models.
#pragma once
#include <memory>

#include "common.h"
#include "conv.h"
#include "gemm.h"

namespace rvcx {

struct SynthModel;
struct RmvpeModel;
struct FcpeModel;
struct CrepeModel;
struct HubertModel;
struct IndexData;

// Folded/packed weights live in per-model REGIONS (one per HuBERT, RMVPE, voice model, index): a region is a
// short list of device chunks filled by bump allocation and freed when its model is unloaded.  The chunk
// sequence, every offset and the region's layout hash depend on tensor SHAPES only -- never on values -- so
// every rank of a multi-GPU job builds the same layout and rank 0 can RCCL-broadcast the chunks
// (rvcx_weights_regions).  Value-dependent facts (may a layer use its fp16 hi/lo image?) are flag bytes in a
// header at the start of chunk 0: they travel with the broadcast and are re-read by rvcx_weights_adopt.
class WeightRegion {
 public:
  static constexpr size_t kChunkBytes = (size_t)32 << 20;
  static constexpr int kMaxFlags = 4096;
  static constexpr size_t kHeaderBytes = kMaxFlags;
  WeightRegion() {
    flags_.reset(new uint8_t[kMaxFlags]);
    std::memset(flags_.get(), 0, kMaxFlags);
  }
  ~WeightRegion() {
    for (auto& c : chunks_) (void)hipFree(c.base);
    if (ovf_words_) (void)hipFree(ovf_words_);
  }
  WeightRegion(const WeightRegion&) = delete;
  WeightRegion& operator=(const WeightRegion&) = delete;
  float* upload(const std::vector<float>& h) { return upload(h.data(), h.size()); }
  float* upload(const float* h, size_t n) {
    float* d = static_cast<float*>(reserve(std::max<size_t>(n, 1) * sizeof(float)));
    if (n && h) RVCX_HIP(hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice));
    return d;
  }
  // bytes of device memory, 256-byte aligned, contents undefined
  void* reserve(size_t bytes) {
    bytes = (bytes + 255) & ~size_t(255);
    if (chunks_.empty()) {
      new_chunk(std::max(kChunkBytes, bytes + kHeaderBytes));
      chunks_[0].off = kHeaderBytes;    // flag header
    }
    if (chunks_.back().off + bytes > chunks_.back().cap) new_chunk(std::max(kChunkBytes, bytes));
    Chunk& c = chunks_.back();
    void* d = static_cast<char*>(c.base) + c.off;
    mix(chunks_.size() - 1);
    mix(c.off);
    mix(bytes);
    c.off += bytes;
    return d;
  }
  // a value-dependent yes/no that must agree on every rank after a broadcast; the returned pointer stays valid
  // for the life of the region and is what kernels' launchers consult
  uint8_t* new_flag(bool v) {
    if (nflags_ >= kMaxFlags) fail("weight region: flag table full");
    flags_[nflags_] = v ? 1 : 0;
    return &flags_[nflags_++];
  }
  // Device word of a flag: split-fp16 kernels stamp it (conv.h: ovf_layer) when an activation of that layer leaves
  // fp16 range.  Not part of the broadcast layout (a plain per-region scratch array, zero when idle).
  int* ovf_word(const uint8_t* flag) {
    if (!ovf_words_) {
      RVCX_HIP(hipMalloc(&ovf_words_, kMaxFlags * sizeof(int)));
      RVCX_HIP(hipMemset(ovf_words_, 0, kMaxFlags * sizeof(int)));
    }
    return ovf_words_ + (flag - flags_.get());
  }
  // After a call that reported an overflow: the largest stamp of this region (0: none) and its flag index.
  int first_overflow(int* index) {
    if (!ovf_words_) return 0;
    std::vector<int> w(kMaxFlags);
    RVCX_HIP(hipMemcpy(w.data(), ovf_words_, kMaxFlags * sizeof(int), hipMemcpyDeviceToHost));
    int best = 0;
    for (int i = 0; i < nflags_; ++i)
      if (w[i] > best) {
        best = w[i];
        *index = i;
      }
    if (best) RVCX_HIP(hipMemset(ovf_words_, 0, kMaxFlags * sizeof(int)));
    return best;
  }
  void clear_overflow() {
    if (ovf_words_) RVCX_HIP(hipMemset(ovf_words_, 0, kMaxFlags * sizeof(int)));
  }
  // pin a layer to the exact-fp32 kernels for the life of the region (sticky); the device header follows
  void drop_flag(int index) {
    flags_[index] = 0;
    ++dropped_;
    pinned_.push_back(index);
    if (!chunks_.empty()) RVCX_HIP(hipMemcpy(chunks_[0].base, flags_.get(), kMaxFlags, hipMemcpyHostToDevice));
  }
  int dropped() const { return dropped_; }
  // what a flag stands for ("conv 128 <- 128 x 7", "attention"): host-side, for rvcx_fp32_pinned; not part of the layout
  void describe_flag(const uint8_t* flag, const std::string& what) {
    const size_t i = (size_t)(flag - flags_.get());
    if (desc_.size() <= i) desc_.resize(i + 1);
    desc_[i] = what;
  }
  std::string pinned_text(const std::string& region) const {
    std::string t;
    for (int i : pinned_)
      t += region + ": layer " + std::to_string(i) + " of " + std::to_string(nflags_) + " (" +
           ((size_t)i < desc_.size() && !desc_[i].empty() ? desc_[i] : std::string("?")) + ")\n";
    return t;
  }
  void seal() {     // publish the host flags into the device header (end of a load)
    if (chunks_.empty()) reserve(256);
    RVCX_HIP(hipMemcpy(chunks_[0].base, flags_.get(), kMaxFlags, hipMemcpyHostToDevice));
  }
  void adopt() {    // re-read the flags from the device header (after a broadcast wrote it)
    if (chunks_.empty()) return;
    RVCX_HIP(hipMemcpy(flags_.get(), chunks_[0].base, kMaxFlags, hipMemcpyDeviceToHost));
  }
  int n_chunks() const { return (int)chunks_.size(); }
  void* chunk_base(int i) const { return chunks_[i].base; }
  size_t chunk_used(int i) const { return chunks_[i].off; }
  size_t used() const {
    size_t t = 0;
    for (auto& c : chunks_) t += c.off;
    return t;
  }
  uint64_t layout_hash() const { return hash_; }

 private:
  struct Chunk {
    void* base;
    size_t cap, off;
  };
  void new_chunk(size_t bytes) {
    Chunk c{nullptr, bytes, 0};
    RVCX_HIP(hipMalloc(&c.base, bytes));
    chunks_.push_back(c);
  }
  void mix(uint64_t v) {
    for (int i = 0; i < 8; ++i) {
      hash_ ^= (v >> (8 * i)) & 0xff;
      hash_ *= 1099511628211ull;
    }
  }
  std::vector<Chunk> chunks_;
  std::unique_ptr<uint8_t[]> flags_;
  int* ovf_words_ = nullptr;
  int nflags_ = 0, dropped_ = 0;
  std::vector<int> pinned_;            // flag indices dropped at run time, in order
  std::vector<std::string> desc_;
  uint64_t hash_ = 1469598103934665603ull;
};

// what the loaders call: uploads go to the region currently being filled (RegionScope)
class WeightSlab {
 public:
  float* upload(const std::vector<float>& h) { return cur().upload(h); }
  float* upload(const float* h, size_t n) { return cur().upload(h, n); }
  uint8_t* new_flag(bool v) { return cur().new_flag(v); }
  WeightRegion& cur() {
    if (!cur_) fail("internal: weight upload outside a RegionScope");
    return *cur_;
  }
  WeightRegion* cur_ = nullptr;
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
  hipEvent_t ev_hub = nullptr;
  hipEvent_t ev_syn[2] = {nullptr, nullptr};       // the main stream reached micro-batch k's synthesizer
  hipEvent_t ev_hubdone[2] = {nullptr, nullptr};   // HuBERT features of micro-batch k (front set k & 1) are ready
  hipStream_t aux[2] = {nullptr, nullptr};   // the three ResBlocks of an NSF stage run side by side
  hipEvent_t ev_aux[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // NOTE: no further streams.  HIP multiplexes streams onto 4 hardware queues per process; a fifth stream made the
  // RMVPE stream share a queue (RMVPE 10.8 -> 16.4 ms, the whole clip 31.6 -> 42.3 ms).  D2H copies ride the main stream.
  hipStream_t stream_io = nullptr;             // always null: see above
  hipEvent_t ev_io = nullptr;
  hipEvent_t ev_front[2] = {nullptr, nullptr}; // front end (high-pass .. reflect pad) of micro-batch k is ready
  hipEvent_t ev_done[2] = {nullptr, nullptr};  // main stream is done with micro-batch k's front set
  hipEvent_t ev_src[2] = {nullptr, nullptr};   // NSF source branch (sine + noise convs) beside TextEncoder/flow
  Arena arena;
  size_t arena_budget = 0;        // activation budget of convert_micro_batch, probed once per context state (0: not yet)
  Arena arena_f0;                 // RMVPE workspace: lives on stream2 across the main stream's arena resets
  Arena arena_hub;                // HuBERT workspace: its own arena because HuBERT k+1 is enqueued (on aux[0], behind decoder k's branch there) before the main arena of k is released
  WeightSlab slab;
  std::string last_error;
  double flops = 0.0;
  int resblock_streams = 0;       // the ResBlocks of an NSF stage on 3 streams (1), on the main stream + aux[0] (2), on the main stream (0)
  bool serial = false;            // profiling: keep every launch on the main stream (true per-kernel times)
  bool serial_env = false;        // RVCX_SERIAL=1: serial for the whole life of the context
  int* dev_err = nullptr;         // device error word (conv.h: kErrGruTimeout, kErrH3Overflow), read after each API call
  // Round 6: a call that ends in its own stream synchronisation copies the word into pinned host memory just in front of
  // it (snapshot_dev_err); check_dev_err / take_overflow then read that copy instead of issuing two synchronous 4-byte
  // hipMemcpy's of their own (~40 us each of a single clip's wall time).  Valid for the one call that took it.
  int* err_host = nullptr;
  bool err_snapshot = false;
  void snapshot_dev_err(hipStream_t s);   // enqueue the copy on s; the caller synchronises s before check_dev_err
  void check_dev_err();           // throws on a GRU timeout; an fp16-split overflow is left for take_overflow()
  bool take_overflow();           // true (and the bit cleared) when a split kernel met an activation beyond fp16
  long fp32_reruns = 0;           // calls repeated because of that (each repeat pins one layer to fp32, or, last resort, all)
  long gru_fallbacks = 0;         // calls repeated with the single-workgroup BiGRU kernel after a cluster time-out
  bool inject_gru_timeout = false;
  int launch_seq = 0;             // launches of split-fp16 kernels since the call began (orders the layers' overflow stamps)
  float timing[9] = {0};
  std::vector<int> last_mbs;      // member counts of the micro-batches of the last convert_batch call
  StageTimer timer;
  std::unique_ptr<HubertModel> hubert;
  std::unique_ptr<RmvpeModel> rmvpe;
  std::unique_ptr<FcpeModel> fcpe;
  std::unique_ptr<CrepeModel> crepe;
  std::vector<std::unique_ptr<SynthModel>> synths;
  std::unique_ptr<IndexData> index;
  Ctx();
  ~Ctx();
  // split-K partial-sum scratch, one per stream (RMVPE and HuBERT run concurrently)
  float* splitk_buf[4] = {nullptr, nullptr, nullptr, nullptr};   // main, stream2, aux0 (HuBERT and a decoder branch: one stream, in order), aux1
  static constexpr long kSplitKFloats = 32L << 20;   // 128 MiB per batch item each
  int splitk_items = 1;                              // batch items the buffers are sized for (ensure_splitk)
  // Called before a micro-batch is enqueued: the split-K decision is taken per item (batch invariance), so the
  // scratch must hold `items` times what one item may use.  Re-allocates (after a device sync) when it grows.
  void ensure_splitk(int items) {
    if (items <= splitk_items) return;
    RVCX_HIP(hipDeviceSynchronize());
    for (auto& p : splitk_buf)
      if (p) {
        (void)hipFree(p);
        p = nullptr;
      }
    splitk_items = items;
  }
  void pair_on(const PairArgs& a, hipStream_t s) {
    const double f = 2.0 * 2.0 * a.B * (double)a.C * a.C * a.k * a.T;   // both convs, 2 M N K each
    flops += f;
    PairArgs b = a;
    b.ovf = dev_err;
    b.seq = ++launch_seq;
    conv_launch_pair(b, f, s);
  }
  void block3_on(const Block3Args& a, hipStream_t s) {
    const double f = 3.0 * 2.0 * 2.0 * a.B * (double)a.C * a.C * 3 * a.T;   // three steps of two convs, 2 M N K each
    flops += f;
    Block3Args b = a;
    b.ovf = dev_err;
    b.seq = ++launch_seq;
    conv_launch_block3(b, f, s);
  }
  void gemm_on(GemmArgs a, hipStream_t s) {
    const double f = 2.0 * (double)a.rows * a.cout * a.cin;
    flops += f;
    a.ovf = dev_err;
    a.seq = ++launch_seq;
    {   // the stream's split-K scratch (conv_on): K-split launches of long-K layers
      const int si = (s == stream2) ? 1 : (s == aux[0] ? 2 : (s == aux[1] ? 3 : 0));
      if (!splitk_buf[si]) RVCX_HIP(hipMalloc(&splitk_buf[si], kSplitKFloats * splitk_items * sizeof(float)));
      a.part = splitk_buf[si];
      a.part_cap = kSplitKFloats * splitk_items;
    }
    conv_launch_gemm(a, f, s);
  }
  void conv(const ConvArgs& a) { conv_on(a, stream); }
  void conv_on(ConvArgs a, hipStream_t s) {
    flops += conv_flops(a);
    const int si = (s == stream2) ? 1 : (s == aux[0] ? 2 : (s == aux[1] ? 3 : 0));
    if (!splitk_buf[si]) RVCX_HIP(hipMalloc(&splitk_buf[si], kSplitKFloats * splitk_items * sizeof(float)));
    a.part = splitk_buf[si];
    a.part_cap = kSplitKFloats * splitk_items;
    a.part_cap_item = kSplitKFloats;
    a.ovf = dev_err;
    a.seq = ++launch_seq;
    launch_conv(a, s);
  }
};

// fills `r` for the duration of a load call
struct RegionScope {
  Ctx& c;
  WeightRegion* prev;
  RegionScope(Ctx& cc, WeightRegion& r) : c(cc), prev(cc.slab.cur_) { cc.slab.cur_ = &r; }
  ~RegionScope() { c.slab.cur_ = prev; }
};

// packed conv layer living in a weight region
struct ConvW {
  const float* w = nullptr;
  const void* w_h3 = nullptr;     // fp16 hi/lo split image (space is reserved whenever the SHAPE allows it)
  const uint8_t* h3_ok = nullptr; // region flag: the image is usable (no weight overflowed fp16 at scale S, and no activation
                                  // of this layer has left fp16 range since the model was loaded)
  int* ovf_word = nullptr;        // the layer's device overflow word (WeightRegion::ovf_word)
  const float* bias = nullptr;
  int cin = 0, cout = 0, k = 1, groups = 1;
  int cin_gp = 0, cout_gp = 0;
};

// an attention call has no weights of its own: its "layer" is a flag + overflow word in the model's region
struct AttFlag {
  const uint8_t* ok = nullptr;
  int* word = nullptr;
  bool h3() const { return !ok || *ok; }
};
inline AttFlag make_att_flag(Ctx& c) {
  AttFlag f;
  f.ok = c.slab.new_flag(true);
  f.word = c.slab.cur().ovf_word(f.ok);
  c.slab.cur().describe_flag(f.ok, "attention call");
  return f;
}

ConvW make_conv(Ctx& c, const float* w, const float* bias, int cout, int cin_g, int k, int groups = 1,
               bool h3 = true);

// may this layer run on the split-fp16 kernels right now?
inline bool conv_h3_ok(const ConvW& w) { return w.w_h3 && w.h3_ok && *w.h3_ok && conv_h3_enabled(); }

// a Linear (k = 1) layer as the time-major GEMM sees it (gemm.h); the caller adds the operands
inline GemmArgs gemm_args(const ConvW& w, long rows, int T) {
  GemmArgs a;
  a.w_h3 = conv_h3_ok(w) ? w.w_h3 : nullptr;
  a.w = w.w;
  a.bias = w.bias;
  a.cin = w.cin;
  a.cin_p = w.cin_gp;
  a.cout = w.cout;
  a.cout_p = w.cout_gp;
  a.rows = rows;
  a.T = T;
  return a;
}

// fill the channel/pad fields of ConvArgs from a packed layer
inline void conv_set_weights(ConvArgs& a, const ConvW& w) {
  a.w = w.w;
  a.w_h3 = (w.w_h3 && w.h3_ok && *w.h3_ok) ? w.w_h3 : nullptr;
  a.ovf_layer = w.ovf_word;
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
