// rvcx -- shared host-side plumbing for the HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace rvcx {

struct Error : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// the BiGRU cluster kernel gave up waiting for a partner workgroup (gru.hip): api_call repeats with the plain kernel
struct GruTimeout : Error {
  using Error::Error;
};

[[noreturn]] inline void fail(const std::string& msg) { throw Error(msg); }

#define RVCX_HIP(expr)                                                                   \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess)                                                                \
      ::rvcx::fail(std::string(#expr) + " -> " + hipGetErrorString(_e) + " @" + __FILE__ + \
                   ":" + std::to_string(__LINE__));                                      \
  } while (0)

#define RVCX_CHECK(cond, msg)                                                             \
  do {                                                                                   \
    if (!(cond)) ::rvcx::fail(std::string("check failed: ") + #cond + " -- " + (msg));    \
  } while (0)

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int round_up(int a, int b) { return cdiv(a, b) * b; }

// ---------------------------------------------------------------------------------------
// Device arena: one big hipMalloc'd slab, bump allocation, mark/release.  No hipMalloc in
// the hot loop (graph-capture friendly, no allocator jitter).  Grows by re-allocation only
// between top-level calls (reserve()).
// ---------------------------------------------------------------------------------------
class Arena {
 public:
  ~Arena() { release_all(); }
  void reserve(size_t bytes) {
    if (bytes <= cap_) return;
    RVCX_CHECK(off_ == 0, "arena grow while in use");
    release_all();
    RVCX_HIP(hipMalloc(&base_, bytes));
    cap_ = bytes;
  }
  template <typename T>
  T* alloc(size_t n) {
    size_t bytes = (n * sizeof(T) + 255) & ~size_t(255);
    if (off_ + bytes > cap_)
      fail("arena exhausted: need " + std::to_string(off_ + bytes) + " have " + std::to_string(cap_));
    T* p = reinterpret_cast<T*>(static_cast<char*>(base_) + off_);
    off_ += bytes;
    if (off_ > peak_) peak_ = off_;
    return p;
  }
  size_t mark() const { return off_; }
  void reset(size_t m = 0) { off_ = m; }
  void swap(Arena& o) {
    std::swap(base_, o.base_);
    std::swap(cap_, o.cap_);
    std::swap(off_, o.off_);
    std::swap(peak_, o.peak_);
  }
  size_t capacity() const { return cap_; }
  size_t peak() const { return peak_; }

 private:
  void release_all() {
    if (base_) (void)hipFree(base_);
    base_ = nullptr;
    cap_ = off_ = 0;
  }
  void* base_ = nullptr;
  size_t cap_ = 0, off_ = 0, peak_ = 0;
};

// ---------------------------------------------------------------------------------------
// Host-side view of the checkpoint tensors handed across the C-ABI (include/rvcx.h).
// ---------------------------------------------------------------------------------------
struct HostTensor {
  const void* data = nullptr;
  int dtype = 0;  // 0 f32, 1 f16, 2 i64
  std::vector<int64_t> shape;
  int64_t numel() const {
    int64_t n = 1;
    for (auto s : shape) n *= s;
    return n;
  }
};

float half_to_float(uint16_t h);

class TensorTable {
 public:
  void add(const std::string& name, HostTensor t) { map_[name] = std::move(t); }
  bool has(const std::string& name) const { return map_.count(name) != 0; }
  const HostTensor& raw(const std::string& name) const {
    auto it = map_.find(name);
    if (it == map_.end()) fail("checkpoint tensor missing: " + name);
    return it->second;
  }
  // float32 copy of a tensor (f16 checkpoints are widened, as net_g.float() does, infer.py:102)
  std::vector<float> f32(const std::string& name) const;
  std::vector<int64_t> shape(const std::string& name) const { return raw(name).shape; }

 private:
  std::unordered_map<std::string, HostTensor> map_;
};

// persistent device buffer for weights
class DeviceWeights {
 public:
  ~DeviceWeights() {
    for (void* p : ptrs_) (void)hipFree(p);
  }
  float* upload(const std::vector<float>& h) {
    float* d = nullptr;
    size_t bytes = std::max<size_t>(h.size(), 1) * sizeof(float);
    RVCX_HIP(hipMalloc(&d, bytes));
    if (!h.empty()) RVCX_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    ptrs_.push_back(d);
    bytes_ += bytes;
    return d;
  }
  size_t bytes() const { return bytes_; }

 private:
  std::vector<void*> ptrs_;
  size_t bytes_ = 0;
};

}  // namespace rvcx
