// host_parallel.h -- the two host-side parallel primitives of the one-off setup work
// (graph compile, gradient incidence list, loader, dumps): a range splitter and a stable
// two-level counting sort.  They replace the reference's per-variable std::sort under
// mutexes (src/factor_graph.cc:90-199) and its serial loaders; nothing here runs per sweep.
#ifndef DWX_HOST_PARALLEL_H_
#define DWX_HOST_PARALLEL_H_

#include <stdint.h>

#include <stdio.h>
#include <stdlib.h>
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <exception>
#include <new>
#include <thread>
#include <vector>

namespace dwx {

// CPUs' worth of CPU time this process may use (cgroup v2 cpu.max, v1 cfs quota), rounded up; 0 = no limit
// found.  A container on a 256-thread host with a quota of 16 reports 256 hardware threads: 64 setup threads
// there are throttled as a group -- `dw gibbs` on config 5's files took 13.8-14.6 s with 64 threads, 12.3-13.0
// with 24-32 (profiles/r04/e2e_threads.json).
inline uint32_t cgroup_cpu_quota() {
  static const uint32_t quota = []() -> uint32_t {
    long long q = 0, period = 0;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char first[64] = {0};
      const int n = fscanf(f, "%63s %lld", first, &period);
      fclose(f);
      if (n == 2 && first[0] != 'm') q = atoll(first);
    } else {
      if (FILE *fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(fq, "%lld", &q) != 1) q = 0; fclose(fq); }
      if (FILE *fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(fp, "%lld", &period) != 1) period = 0; fclose(fp); }
    }
    if (q <= 0 || period <= 0) return 0u;
    return (uint32_t)std::min<long long>((q + period - 1) / period, 4096);
  }();
  return quota;
}

// threads for the host-side setup work: the caller's request, else DWX_HOST_THREADS, else the hardware
// concurrency capped at 64 -- and at twice the cgroup's CPU quota where there is one (the phases stall on page
// faults and on each other: two threads per granted CPU measured best over both sizes -- config 3: 1.34-1.52 s
// with 32 threads, 1.45-1.53 with 64, 1.56-1.79 with 24 at a quota of 16)
inline uint32_t host_threads(uint32_t requested = 0) {
  if (requested) return std::min(requested, 256u);
  if (const char *e = getenv("DWX_HOST_THREADS")) {
    const long v = atol(e);
    if (v > 0) return (uint32_t)std::min(v, 256L);
  }
  uint32_t n = std::min(std::max(1u, std::thread::hardware_concurrency()), 64u);
  if (const uint32_t q = cgroup_cpu_quota()) n = std::min(n, std::max(1u, 2 * q));
  return n;
}

// Give the pages of a large private mapping back from several threads, then unmap it.  Freeing is not
// free: a kernel that clears pages when they are released (init_on_free -- the GPU boxes: 20 GB/s from one
// thread, huge pages or not) made the teardown of config 5's 67 GB of host columns 4 s of an 18 s run.
// MADV_DONTNEED takes the address-space lock shared, so the slices really run side by side, and in 64 MiB
// steps, so that a writer (another thread's mmap) never waits longer than one step.
inline void unmap_parallel(void *p, size_t bytes) {
  constexpr size_t kStep = (size_t)64 << 20, kPerThread = (size_t)256 << 20;
  const uint32_t T = (uint32_t)std::min<size_t>(std::min(host_threads(), 32u), bytes / kPerThread);
  if (T >= 2) {
    const size_t steps = (bytes + kStep - 1) / kStep;
    std::atomic<size_t> next{0};
    auto work = [&]() {
      for (size_t i; (i = next.fetch_add(1)) < steps;)
        madvise((char *)p + i * kStep, std::min(kStep, bytes - i * kStep), MADV_DONTNEED);
    };
    std::vector<std::thread> th;
    for (uint32_t t = 1; t < T; ++t) {
      // (called from destructors: a thread that cannot be started is done without)
      try { th.emplace_back(work); } catch (...) { break; }
    }
    work();
    for (auto &t : th) t.join();
  }
  munmap(p, bytes);
}

// The live large mappings of every RawArray in the process (graph_compile.cc holds the list: one copy, in the
// library, for its own arrays and its callers').  A run that is DONE and leaves through _Exit (dw_cli.cc:
// quick_exit_if_done) hands all of their pages back from all threads first -- the exit itself clears them on
// one core (20 GB/s: 1.8 s for config 5's graph).  Worth it only since the uploads are staged (rt_hip.h): pages
// the runtime has copied from directly come back ten times slower, whoever frees them.
void mapping_registry_add(void *p, size_t bytes);
void mapping_registry_remove(void *p);
void mapping_registry_drop_all();

// Uninitialised array of trivially copyable T: unlike std::vector it does not zero-fill
// serially, so the first touch (and its page faults) happens inside the parallel loops.
// Large arrays are mapped directly and advised to use transparent huge pages: first touch
// of multi-GB columns in 4 KiB pages is what the setup phases otherwise spend their time on.
template <class T>
class RawArray {
 public:
  RawArray() = default;
  explicit RawArray(size_t n) { reset(n); }
  RawArray(const RawArray &) = delete;
  RawArray &operator=(const RawArray &) = delete;
  RawArray(RawArray &&o) noexcept : p_(o.p_), n_(o.n_), mapped_(o.mapped_) { o.p_ = nullptr; o.n_ = 0; o.mapped_ = 0; }
  RawArray &operator=(RawArray &&o) noexcept {
    if (this != &o) {
      release();
      p_ = o.p_; n_ = o.n_; mapped_ = o.mapped_;
      o.p_ = nullptr; o.n_ = 0; o.mapped_ = 0;
    }
    return *this;
  }
  ~RawArray() { release(); }
  void reset(size_t n) {
    release();
    n_ = n;
    if (!n) return;
    const size_t bytes = n * sizeof(T);
    if (bytes >= kHuge) {
      mapped_ = (bytes + kHuge - 1) / kHuge * kHuge;
      void *m = mmap(nullptr, mapped_, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
      if (m == MAP_FAILED) { mapped_ = 0; n_ = 0; throw std::bad_alloc(); }
      madvise(m, mapped_, MADV_HUGEPAGE);
      p_ = (T *)m;
      mapping_registry_add(m, mapped_);
    } else {
      p_ = (T *)malloc(bytes);
      if (!p_) { n_ = 0; throw std::bad_alloc(); }
    }
  }
  void clear() { reset(0); }
  size_t size() const { return n_; }
  T *data() { return p_; }
  const T *data() const { return p_; }
  T *begin() { return p_; }
  T *end() { return p_ + n_; }
  T &operator[](size_t i) { return p_[i]; }
  const T &operator[](size_t i) const { return p_[i]; }

 private:
  static constexpr size_t kHuge = (size_t)2 << 20;
  void release() {
    if (mapped_) { mapping_registry_remove(p_); unmap_parallel(p_, mapped_); } else free(p_);
    p_ = nullptr; n_ = 0; mapped_ = 0;
  }
  T *p_ = nullptr;
  size_t n_ = 0, mapped_ = 0;
};

// fn(part, begin, end) over [0, n) cut into at most n_threads contiguous parts; the first
// exception thrown by any part is rethrown on the caller's thread.
template <class Fn>
void parallel_parts(uint64_t n, uint32_t n_threads, Fn &&fn, uint64_t serial_below = 65536) {
  if (n_threads <= 1 || n < serial_below) {
    fn(0u, (uint64_t)0, n);
    return;
  }
  std::vector<std::thread> th;
  std::exception_ptr eptr = nullptr;
  std::atomic_flag lock = ATOMIC_FLAG_INIT;
  const uint64_t per = (n + n_threads - 1) / n_threads;
  for (uint32_t t = 0; t < n_threads; ++t) {
    const uint64_t b = std::min<uint64_t>(n, per * t), e = std::min<uint64_t>(n, per * (t + 1));
    if (b >= e) break;
    th.emplace_back([&, t, b, e]() {
      try {
        fn(t, b, e);
      } catch (...) {
        if (!lock.test_and_set()) eptr = std::current_exception();
      }
    });
  }
  for (auto &t : th) t.join();
  if (eptr) std::rethrow_exception(eptr);
}

// serial_below: item counts under it run on the caller's thread (the default suits
// per-variable / per-edge items; pass a small value when one item is a lot of work)
template <class Fn>
void parallel_ranges(uint64_t n, uint32_t n_threads, Fn &&fn, uint64_t serial_below = 65536) {
  parallel_parts(n, n_threads, [&](uint32_t, uint64_t b, uint64_t e) { fn(b, e); }, serial_below);
}

// In-place inclusive prefix sum of a[0 .. n) (two passes over contiguous parts: part totals,
// then the parts again with their offsets).
template <class T>
void parallel_inclusive_prefix(T *a, uint64_t n, uint32_t n_threads) {
  const uint32_t P = (n_threads <= 1 || n < (1u << 18)) ? 1u : n_threads;
  if (P == 1) {
    for (uint64_t i = 1; i < n; ++i) a[i] += a[i - 1];
    return;
  }
  std::vector<T> total(P + 1, 0);
  const uint64_t per = (n + P - 1) / P;
  parallel_parts(P, P, [&](uint32_t, uint64_t tb, uint64_t te) {
    for (uint64_t t = tb; t < te; ++t) {
      T acc = 0;
      for (uint64_t i = std::min(n, per * t); i < std::min(n, per * (t + 1)); ++i) acc += a[i];
      total[t + 1] = acc;
    }
  }, 0);
  for (uint32_t t = 0; t < P; ++t) total[t + 1] += total[t];
  parallel_parts(P, P, [&](uint32_t, uint64_t tb, uint64_t te) {
    for (uint64_t t = tb; t < te; ++t) {
      T acc = total[t];
      for (uint64_t i = std::min(n, per * t); i < std::min(n, per * (t + 1)); ++i) { acc += a[i]; a[i] = acc; }
    }
  }, 0);
}

// Stable parallel counting sort of generated records by an integer key in [0, n_keys).
//   produce(begin, end, emit): enumerates, in order, the records of the source range
//     [begin, end) of [0, n_src) calling emit(const Rec &); it runs TWICE per range (count,
//     then scatter) and must emit the same records both times.
//   key(rec) -> uint64_t < n_keys.
// Output: out = all records ordered by key, records of equal key in source order;
// start[k] .. start[k+1] = the records of key k (start has n_keys + 1 entries).
// Level 1 partitions by the high key bits into <= 4096 buckets (per-thread histograms,
// one streaming scatter), level 2 counting-sorts every bucket on its own (cache-sized
// counters), buckets handed out dynamically.
template <class Rec, class KeyFn, class Produce>
void parallel_group_by_key(uint64_t n_src, uint32_t n_threads, uint64_t n_keys, KeyFn &&key,
                           Produce &&produce, RawArray<Rec> &out, std::vector<uint64_t> &start,
                           uint64_t serial_below = 65536) {
  start.assign(n_keys + 1, 0);
  out.clear();
  if (n_keys == 0 || n_src == 0) return;
  uint32_t shift = 0;
  while ((n_keys >> shift) > 4096) ++shift;
  const uint64_t nb = ((n_keys - 1) >> shift) + 1;
  const uint32_t T = (n_threads <= 1 || n_src < serial_below) ? 1u : (uint32_t)std::min<uint64_t>(n_threads, n_src);
  std::vector<std::vector<uint64_t>> hist(T, std::vector<uint64_t>(nb, 0));
  parallel_parts(n_src, T, [&](uint32_t t, uint64_t b, uint64_t e) {
    std::vector<uint64_t> &h = hist[t];
    produce(b, e, [&](const Rec &r) { ++h[key(r) >> shift]; });
  }, 0);
  // bucket-major, thread-minor offsets keep the scatter stable
  std::vector<uint64_t> bucket_start(nb + 1, 0);
  uint64_t total = 0;
  for (uint64_t b = 0; b < nb; ++b) {
    bucket_start[b] = total;
    for (uint32_t t = 0; t < T; ++t) { const uint64_t c = hist[t][b]; hist[t][b] = total; total += c; }
  }
  bucket_start[nb] = total;
  RawArray<Rec> tmp(total);
  parallel_parts(n_src, T, [&](uint32_t t, uint64_t b, uint64_t e) {
    std::vector<uint64_t> &h = hist[t];
    produce(b, e, [&](const Rec &r) { tmp[h[key(r) >> shift]++] = r; });
  }, 0);
  out.reset(total);
  std::atomic<uint64_t> next{0};
  const uint64_t width = (uint64_t)1 << shift;
  parallel_parts(T, T, [&](uint32_t, uint64_t, uint64_t) {
    std::vector<uint64_t> cnt(width + 1);
    for (;;) {
      const uint64_t b = next.fetch_add(1);
      if (b >= nb) break;
      const uint64_t lo = bucket_start[b], hi = bucket_start[b + 1], k0 = b << shift;
      const uint64_t kn = std::min<uint64_t>(width, n_keys - k0);
      std::fill(cnt.begin(), cnt.begin() + kn + 1, 0);
      for (uint64_t i = lo; i < hi; ++i) ++cnt[key(tmp[i]) - k0 + 1];
      for (uint64_t k = 0; k < kn; ++k) cnt[k + 1] += cnt[k];
      for (uint64_t k = 0; k < kn; ++k) start[k0 + k] = lo + cnt[k];
      for (uint64_t i = lo; i < hi; ++i) out[lo + cnt[key(tmp[i]) - k0]++] = tmp[i];
    }
  }, 0);
  start[n_keys] = total;
}

}  // namespace dwx
#endif
