// rt_emu.h -- TEST-ONLY stand-in for sampler_amd/csrc/rt_hip.h: "device" memory is
// host memory, a launch runs the grid on fibers (hip_emul.h).  Used only by
// tests/hipemu/Makefile to build libdwx_emu.so for sanitizer runs of the kernel source.
#ifndef DWX_RT_EMU_H_
#define DWX_RT_EMU_H_

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

#include "hip_emul.h"

namespace dwx {
namespace rt {

typedef void *stream_t;
typedef std::chrono::steady_clock::time_point *event_t;

inline void init_device(int dev) {
  if (dev != 0) throw std::runtime_error("emulated device ordinal must be 0");
}
inline void set_device(int) {}
inline int device_count() { return 1; }
inline void (*&oom_hook())(int) {
  static void (*hook)(int) = nullptr;
  return hook;
}
inline void *dmalloc(size_t n) { return malloc(n ? n : 16); }
inline void dfree(void *p) { free(p); }
inline void h2d(void *d, const void *h, size_t n, stream_t) { if (n) memcpy(d, h, n); }
inline void stage_trim() {}
inline void d2h(void *h, const void *d, size_t n, stream_t) { if (n) memcpy(h, d, n); }
inline void d2d(void *dst, const void *src, size_t n, stream_t) { if (n) memcpy(dst, src, n); }
inline void dmemset(void *d, int v, size_t n, stream_t) { if (n) memset(d, v, n); }
inline stream_t stream_create() { return (stream_t)1; }
inline void stream_destroy(stream_t) {}
inline void stream_sync(stream_t) {}
inline event_t event_create() { return new std::chrono::steady_clock::time_point(); }
inline event_t event_create_ordering() { return event_create(); }
inline void stream_wait_event(stream_t, event_t) {}   // (launches run to completion in order)
inline void event_destroy(event_t e) { delete e; }
inline void event_record(event_t e, stream_t) { *e = std::chrono::steady_clock::now(); }
inline double event_elapsed_ms(event_t a, event_t b) {
  return std::chrono::duration<double, std::milli>(*b - *a).count();
}
// "capture": launches between begin and end run eagerly (as everywhere in this harness); the
// graph objects are tokens and a graph launch has nothing left to do
typedef void *graph_t;
typedef void *graph_exec_t;
inline void capture_begin(stream_t) {}
inline graph_t capture_end(stream_t) { return (graph_t)1; }
inline void capture_abandon(stream_t) {}
inline graph_exec_t graph_instantiate(graph_t) { return (graph_exec_t)1; }
inline bool graph_exec_update(graph_exec_t, graph_t) { return true; }
inline void graph_launch(graph_exec_t, stream_t) {}
inline void graph_destroy(graph_t) {}
inline void graph_exec_destroy(graph_exec_t) {}

template <class K>
inline void allow_dynamic_lds(K, size_t) {}

// a deliberately small "chip" so the persistent tile loop iterates in tests
template <class K>
inline unsigned resident_blocks(K, unsigned, size_t) { return 3; }

inline unsigned cu_count() { return 5; }
// workgroups a kernel with a grid barrier may be launched with: the harness runs blocks one after
// another, so ONE (its barrier then passes trivially; the chunk loop, the rows and the in-launch
// update run as on the device)
inline unsigned grid_barrier_blocks() { return 1; }

template <class K, class... A>
inline void launch(K kernel, unsigned grid, unsigned block, size_t lds, stream_t, A... args) {
  emu::run_grid(grid, block, lds, [&]() { kernel(args...); });
}

}  // namespace rt
}  // namespace dwx
#endif
