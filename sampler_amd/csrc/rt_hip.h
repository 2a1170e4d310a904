// rt_hip.h -- the thin HIP runtime layer under the C ABI (device memory, stream,
// events, launches).  tests/hipemu/rt_emu.h provides the same names on the host so the
// kernels can run under sanitizers in CI; the product library only ever uses this file.
#ifndef DWX_RT_HIP_H_
#define DWX_RT_HIP_H_

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "host_parallel.h"

namespace dwx {
namespace rt {

inline void check(hipError_t e, const char *what) {
  if (e != hipSuccess)
    throw std::runtime_error(std::string("HIP error in ") + what + ": " + hipGetErrorString(e));
}
#define DWX_HIP(x) ::dwx::rt::check((x), #x)

typedef hipStream_t stream_t;
typedef hipEvent_t event_t;

inline void init_device(int dev) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    throw std::runtime_error("no HIP device available (the dwx sampler has no CPU fallback)");
  if (dev < 0 || dev >= n)
    throw std::runtime_error("HIP device ordinal " + std::to_string(dev) + " out of range");
  DWX_HIP(hipSetDevice(dev));
}
inline void set_device(int dev) { DWX_HIP(hipSetDevice(dev)); }
inline int device_count() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
  return n;
}

// (what to try once when the device is out of memory: dwx_api.cc points it at the device builds' block cache,
// whose idle blocks are the library's own to give back)
inline void (*&oom_hook())(int) {
  static void (*hook)(int) = nullptr;
  return hook;
}
inline void *dmalloc(size_t n) {
  void *p = nullptr;
  hipError_t e = hipMalloc(&p, n ? n : 16);
  if (e != hipSuccess && oom_hook()) {
    (void)hipGetLastError();
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) oom_hook()(dev);
    e = hipMalloc(&p, n ? n : 16);
  }
  DWX_HIP(e);
  return p;
}
inline void dfree(void *p) { if (p) (void)hipFree(p); }
// Large uploads go through the library's own pinned staging buffers (two 64 MiB chunks in turn, filled by a
// few host threads) instead of hipMemcpyAsync on the caller's pageable array.  The runtime's own path is a
// little faster (49 against 43 GB/s) but leaves the SOURCE pages registered with the driver: giving them back
// later -- munmap, MADV_DONTNEED, the process exit -- then runs at 10 GB/s instead of 100 (measured on the
// box: 16 GB released in 1.55 s after a plain hipMemcpy, in 0.11-0.15 s after a staged one; hipHostRegister /
// Unregister around the copy: 1.50 s).  `dw gibbs` on config 5's files spent 1.8 s of 9.5 leaving.
struct StagePair {
  char *buf[2] = {nullptr, nullptr};
  hipEvent_t ev[2] = {nullptr, nullptr};
  bool used[2] = {false, false};
  int device = -1;
  bool busy = false;
};
constexpr size_t STAGE_CHUNK = (size_t)64 << 20, STAGE_MIN = (size_t)256 << 20;
inline std::mutex &stage_mutex() { static std::mutex m; return m; }
inline std::vector<StagePair *> &stage_pool() { static std::vector<StagePair *> v; return v; }
inline StagePair *stage_acquire() {
  int dev = 0;
  DWX_HIP(hipGetDevice(&dev));
  {
    std::lock_guard<std::mutex> lk(stage_mutex());
    for (StagePair *p : stage_pool())
      if (!p->busy && p->device == dev) { p->busy = true; return p; }
  }
  StagePair *p = new StagePair();
  p->device = dev; p->busy = true;
  for (int i = 0; i < 2; ++i) {
    if (hipHostMalloc((void **)&p->buf[i], STAGE_CHUNK, hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&p->ev[i], hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      for (int j = 0; j < 2; ++j) { if (p->buf[j]) (void)hipHostFree(p->buf[j]); if (p->ev[j]) (void)hipEventDestroy(p->ev[j]); }
      delete p;
      return nullptr;      // (no pinned memory to be had: the caller copies the plain way)
    }
  }
  std::lock_guard<std::mutex> lk(stage_mutex());
  stage_pool().push_back(p);
  return p;
}
inline void stage_release(StagePair *p) {
  std::lock_guard<std::mutex> lk(stage_mutex());
  p->busy = false;
}
// free the idle staging pairs (their last copies complete first)
inline void stage_trim() {
  std::vector<StagePair *> dead;
  {
    std::lock_guard<std::mutex> lk(stage_mutex());
    auto &pool = stage_pool();
    for (size_t i = 0; i < pool.size();) {
      if (!pool[i]->busy) { dead.push_back(pool[i]); pool[i] = pool.back(); pool.pop_back(); } else ++i;
    }
  }
  for (StagePair *p : dead) {
    for (int i = 0; i < 2; ++i) {
      if (p->used[i]) (void)hipEventSynchronize(p->ev[i]);
      (void)hipHostFree(p->buf[i]);
      (void)hipEventDestroy(p->ev[i]);
    }
    delete p;
  }
}
inline void h2d(void *d, const void *h, size_t n, stream_t s) {
  if (!n) return;
  StagePair *p = n >= STAGE_MIN ? stage_acquire() : nullptr;
  if (!p) {
    DWX_HIP(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s));
    return;
  }
  const uint32_t nth = std::min(host_threads(), 16u);
  try {
    size_t k = 0;
    for (size_t off = 0; off < n; off += STAGE_CHUNK, ++k) {
      const size_t len = std::min(STAGE_CHUNK, n - off);
      const int b = (int)(k & 1);
      if (p->used[b]) DWX_HIP(hipEventSynchronize(p->ev[b]));      // (the copy that last read this chunk)
      char *dst = p->buf[b];
      const char *src = (const char *)h + off;
      parallel_ranges(len, nth, [&](uint64_t x, uint64_t y) { std::memcpy(dst + x, src + x, y - x); }, (uint64_t)1 << 20);
      DWX_HIP(hipMemcpyAsync((char *)d + off, dst, len, hipMemcpyHostToDevice, s));
      DWX_HIP(hipEventRecord(p->ev[b], s));
      p->used[b] = true;
    }
  } catch (...) {
    stage_release(p);
    throw;
  }
  stage_release(p);
}
inline void d2h(void *h, const void *d, size_t n, stream_t s) {
  if (n) DWX_HIP(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s));
}
inline void d2d(void *dst, const void *src, size_t n, stream_t s) {
  if (n) DWX_HIP(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, s));
}
inline void dmemset(void *d, int v, size_t n, stream_t s) {
  if (n) DWX_HIP(hipMemsetAsync(d, v, n, s));
}
inline stream_t stream_create() {
  stream_t s;
  DWX_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  return s;
}
inline void stream_destroy(stream_t s) { (void)hipStreamDestroy(s); }
inline void stream_sync(stream_t s) { DWX_HIP(hipStreamSynchronize(s)); }
inline event_t event_create() {
  event_t e;
  DWX_HIP(hipEventCreate(&e));
  return e;
}
// (ordering only: no timestamps taken)
inline event_t event_create_ordering() {
  event_t e;
  DWX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return e;
}
inline void stream_wait_event(stream_t s, event_t e) { DWX_HIP(hipStreamWaitEvent(s, e, 0)); }
inline void event_destroy(event_t e) { (void)hipEventDestroy(e); }
inline void event_record(event_t e, stream_t s) { DWX_HIP(hipEventRecord(e, s)); }
inline double event_elapsed_ms(event_t a, event_t b) {
  float ms = 0;
  DWX_HIP(hipEventElapsedTime(&ms, a, b));
  return ms;
}

// Stream capture + graph replay (a split learning sweep is a chain of dozens of short dependent
// launches: captured once per plan level, re-captured and patched in place every sweep -- the
// sweep counter and the step change, the topology does not -- and handed over as ONE launch).
typedef hipGraph_t graph_t;
typedef hipGraphExec_t graph_exec_t;
inline void capture_begin(stream_t s) { DWX_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed)); }
inline graph_t capture_end(stream_t s) {
  graph_t g = nullptr;
  DWX_HIP(hipStreamEndCapture(s, &g));
  return g;
}
// (after an error inside a capture: leave the stream usable, keep the first error)
inline void capture_abandon(stream_t s) {
  graph_t g = nullptr;
  (void)hipStreamEndCapture(s, &g);
  if (g) (void)hipGraphDestroy(g);
  (void)hipGetLastError();
}
inline graph_exec_t graph_instantiate(graph_t g) {
  graph_exec_t e = nullptr;
  DWX_HIP(hipGraphInstantiate(&e, g, nullptr, nullptr, 0));
  return e;
}
// patch an instantiated graph with the parameters of a freshly captured one; false: the topology
// differs (the caller instantiates anew)
inline bool graph_exec_update(graph_exec_t e, graph_t g) {
  hipGraphNode_t bad = nullptr;
  hipGraphExecUpdateResult res = hipGraphExecUpdateSuccess;
  const hipError_t rc = hipGraphExecUpdate(e, g, &bad, &res);
  if (rc != hipSuccess) (void)hipGetLastError();
  return rc == hipSuccess && res == hipGraphExecUpdateSuccess;
}
inline void graph_launch(graph_exec_t e, stream_t s) { DWX_HIP(hipGraphLaunch(e, s)); }
inline void graph_destroy(graph_t g) { if (g) (void)hipGraphDestroy(g); }
inline void graph_exec_destroy(graph_exec_t e) { if (e) (void)hipGraphExecDestroy(e); }

template <class K>
inline void allow_dynamic_lds(K kernel, size_t bytes) {
  if (bytes > 48 * 1024)
    DWX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
}

// workgroups of this kernel that are resident on the whole chip at once
template <class K>
inline unsigned resident_blocks(K kernel, unsigned block, size_t lds) {
  int dev = 0, cus = 0, per_cu = 0;
  DWX_HIP(hipGetDevice(&dev));
  DWX_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  DWX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, (int)block, lds));
  if (per_cu < 1) per_cu = 1;
  return (unsigned)(cus * per_cu);
}

inline unsigned cu_count() {
  int dev = 0, cus = 0;
  DWX_HIP(hipGetDevice(&dev));
  DWX_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  return (unsigned)std::max(1, cus);
}

// workgroups a kernel with a hand-rolled grid barrier may be launched with: one per CU -- all of them
// resident together whatever else the sampler has queued (the barrier's price list and the sc1
// hand-off forms of the CDNA guide are measured at one workgroup per CU)
inline unsigned grid_barrier_blocks() { return cu_count(); }

template <class K, class... A>
inline void launch(K kernel, unsigned grid, unsigned block, size_t lds, stream_t s, A... args) {
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), lds, s, args...);
  DWX_HIP(hipGetLastError());
}

}  // namespace rt
}  // namespace dwx
#endif
