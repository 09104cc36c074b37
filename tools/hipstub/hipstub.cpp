// A stand-in for the HIP runtime, for the HOST-SIDE sanitizer build only (make host-asan; SURVEY.md 5 "race detection /
// sanitizers").  The ~35 runtime entry points librvcx.so imports are implemented on plain host memory: "device"
// allocations are malloc (so AddressSanitizer sees every host-side copy into or out of them) or lazily committed mmap
// above 256 MB (activation arenas), copies are memcpy, kernel launches do nothing, events and streams are counted objects.
// What runs for real under ASan / UBSan is everything the library does on the host: checkpoint folding and packing, weight
// regions, the micro-batch planner, arena arithmetic, f0-file interpolation, error paths.  Never linked into the product.
#include <hip/hip_runtime_api.h>
#include <sys/mman.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

namespace {
std::mutex g_mu;
std::map<void*, size_t> g_big;                 // mmap'd "device" blocks
constexpr size_t kBig = (size_t)256 << 20;
long g_launches = 0, g_live_events = 0, g_live_streams = 0;
struct Obj {
  int kind;
};
}  // namespace

extern "C" {

hipError_t hipGetDeviceCount(int* n) {
  *n = 1;
  return hipSuccess;
}
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int* d) {
  *d = 0;
  return hipSuccess;
}
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) {
  *v = 256;
  return hipSuccess;
}
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
  memset(p, 0, sizeof(*p));
  snprintf(p->name, sizeof(p->name), "hipstub (host memory, no kernels)");
  p->totalGlobalMem = (size_t)288 << 30;
  p->multiProcessorCount = 256;
  return hipSuccess;
}
hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) {
  *lo = 0;
  *hi = -1;
  return hipSuccess;
}
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : "hipstub error"; }
hipError_t hipMemGetInfo(size_t* f, size_t* t) {
  *f = (size_t)280 << 30;
  *t = (size_t)288 << 30;
  return hipSuccess;
}

hipError_t hipMalloc(void** p, size_t n) {
  if (n == 0) n = 1;
  if (n >= kBig) {
    void* m = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (m == MAP_FAILED) return hipErrorOutOfMemory;
    std::lock_guard<std::mutex> g(g_mu);
    g_big[m] = n;
    *p = m;
    return hipSuccess;
  }
  *p = malloc(n);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipHostFree(void* p);
hipError_t hipFree(void* p) {
  if (!p) return hipSuccess;
  {
    std::lock_guard<std::mutex> g(g_mu);
    auto it = g_big.find(p);
    if (it != g_big.end()) {
      munmap(p, it->second);
      g_big.erase(it);
      return hipSuccess;
    }
  }
  free(p);
  return hipSuccess;
}
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) {
  if (n) memmove(d, s, n);
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) {
  if (n) memmove(d, s, n);
  return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t) {
  for (size_t r = 0; r < h; ++r) memmove((char*)d + r * dp, (const char*)s + r * sp, w);
  return hipSuccess;
}
// big fills are what a kernel-less run does not need: keep the lazily committed arenas uncommitted
hipError_t hipMemset(void* d, int v, size_t n) {
  if (n && n < kBig) memset(d, v, n);
  return hipSuccess;
}
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { return hipMemset(d, v, n); }

hipError_t hipHostFree(void* p) { return hipFree(p); }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
  *s = reinterpret_cast<hipStream_t>(new Obj{1});
  ++g_live_streams;
  return hipSuccess;
}
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned f, int) { return hipStreamCreateWithFlags(s, f); }
hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, uint32_t, const uint32_t*) { return hipStreamCreateWithFlags(s, 0); }
hipError_t hipStreamDestroy(hipStream_t s) {
  delete reinterpret_cast<Obj*>(s);
  --g_live_streams;
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) {
  *e = reinterpret_cast<hipEvent_t>(new Obj{2});
  ++g_live_events;
  return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) {
  delete reinterpret_cast<Obj*>(e);
  --g_live_events;
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) {
  *ms = 0.001f;
  return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void*, int, size_t) {
  *n = 2;
  return hipSuccess;
}
hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t) {
  ++g_launches;
  return hipSuccess;
}
// what the host pass of a .hip file emits around a <<<>>> launch and at load time
hipError_t __hipPushCallConfiguration(dim3, dim3, size_t, hipStream_t) { return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3* g, dim3* b, size_t* sh, hipStream_t* s) {
  *g = dim3(1);
  *b = dim3(1);
  *sh = 0;
  *s = nullptr;
  return hipSuccess;
}
void** __hipRegisterFatBinary(const void*) {
  static void* h = nullptr;
  return &h;
}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipUnregisterFatBinary(void**) {}

long hipstub_launches(void) { return g_launches; }
long hipstub_live_objects(void) { return g_live_events + g_live_streams; }

}  // extern "C"
