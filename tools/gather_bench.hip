// tools/gather_bench.hip -- the ceiling of what bounds the config-3 sweeps: random 4-byte
// gathers out of a table of T bytes (4 KB ... 16 MB: the CU's vector L1, the XCD's L2, the
// Infinity Cache), at 1-8 waves per SIMD, in the shapes the sweep kernels could use.
//
// Not product code: a calibration tool.  bench.py runs it (outside the timed region) to
// report `roofline.secondary`; profiles/r02/gather_bench.* keep its output and rocprof summary.
//
//   gather_bench [--quick] [--json]      one line per (mode, table, occupancy)
//
// Modes
//   global     global_load_dword, index from a per-lane hash (no index traffic at all)
//   nt         the same with the non-temporal bit
//   buffer     buffer_load_dword offen through a descriptor of exactly the table
//   x2         8-byte gathers (global_load_dwordx2) out of a table of pairs
//   sorted     the 64 lanes of a wave-instruction gather ascending, nearby addresses
//              (what a wid-sorted staging order over 16 tiles would see: ~0.6 lines per lane)
//   stream     the sweep's own shape: lane reads K 8-byte records (coalesced, non-temporal,
//              buffer descriptor), gathers table[rec.key], sums -- record stream + gathers,
//              nothing else.  THE calibrated ceiling of sweep8_kernel's staging phase.
//   scalar     gathers through the scalar cache: v_readlane + s_load_dword per lane
//   lds        the table (<= 64 KB) in LDS: ds_read_b32 with random addresses
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr int THREADS = 256;
enum Mode { M_GLOBAL, M_NT, M_BUFFER, M_X2, M_SORTED, M_STREAM, M_SCALAR, M_LDS, M_COUNT };
static const char *kModeName[M_COUNT] = {"global", "nt", "buffer", "x2", "sorted", "stream", "scalar", "lds"};

__device__ __forceinline__ uint32_t mix(uint32_t x) {   // one round of a cheap integer hash
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// U independent gathers in flight per lane and step; `iters` steps per lane.
template <int MODE, int U>
__global__ void __launch_bounds__(THREADS) gather_kernel(const uint32_t *__restrict__ table, uint32_t mask,
                                                         uint32_t iters, const u32x2 *__restrict__ recs,
                                                         uint32_t nrec, uint32_t *out) {
  extern __shared__ uint32_t lds[];
  const uint32_t tid = threadIdx.x, gid = blockIdx.x * THREADS + tid;
  uint32_t acc = 0, state = mix(gid * 2654435761u + 12345u);
  if (MODE == M_LDS) {
    for (uint32_t i = tid; i <= mask; i += THREADS) lds[i] = table[i];
    __syncthreads();
  }
  const __amdgpu_buffer_rsrc_t trs =
      __builtin_amdgcn_make_buffer_rsrc((void *)table, 0, (int)((mask + 1) * 4u), 0x00020000);
  // stream mode: workgroup b owns records [b * per, (b+1) * per), read in steps of U * 256
  const uint32_t per = MODE == M_STREAM ? (nrec / gridDim.x) / (U * THREADS) * (U * THREADS) : 0;
  const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(
      (void *)(recs + (size_t)blockIdx.x * per), 0, (int)(per * 8u), 0x00020000);
  const uint32_t steps = MODE == M_STREAM ? per / (U * THREADS) : iters;
  // (software pipelined like the sweep: the records of step i+1 are in flight during step i)
  u32x2 rec[U];
  if (MODE == M_STREAM) {
#pragma unroll
    for (int u = 0; u < U; ++u)
      rec[u] = __builtin_amdgcn_raw_buffer_load_b64(rrs, (int)(tid * 8u), (int)(u * THREADS * 8), 2);
  }
  for (uint32_t it = 0; it < steps; ++it) {
    uint32_t idx[U], v[U];
    if (MODE == M_STREAM) {
#pragma unroll
      for (int u = 0; u < U; ++u) idx[u] = rec[u].x & mask;
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = table[idx[u]];
      const uint32_t nxt = (it + 1 < steps ? it + 1 : it) * (U * THREADS * 8u);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        acc += rec[u].y;
        rec[u] = __builtin_amdgcn_raw_buffer_load_b64(rrs, (int)(tid * 8u), (int)(nxt + u * THREADS * 8), 2);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc += v[u];
      continue;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      state = state * 1664525u + 1013904223u;
      uint32_t r = mix(state);
      if (MODE == M_SORTED) {
        // wave-uniform random base, lanes ascending with a random stride of 0..38 words:
        // ~0.6 distinct 128-byte lines per lane
        const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
        r = base + (tid & 63u) * 19u + (r & 15u);
      }
      idx[u] = r & mask;
    }
    if (MODE == M_GLOBAL || MODE == M_SORTED) {
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = table[idx[u]];
    } else if (MODE == M_NT) {
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(&table[idx[u]]);
    } else if (MODE == M_BUFFER) {
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b32(trs, (int)(idx[u] * 4u), 0, 0);
    } else if (MODE == M_X2) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const u32x2 p = ((const u32x2 *)table)[idx[u] >> 1];
        v[u] = p.x ^ p.y;
      }
    } else if (MODE == M_LDS) {
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = lds[idx[u]];
    } else if (MODE == M_SCALAR) {
      // every lane's address through the scalar cache: 64 (readlane, s_load_dword) pairs per
      // gather instruction replaced.  The loaded values are summed on the scalar unit (a real
      // kernel would v_writelane them back: one more VALU op per lane).
      typedef const __attribute__((address_space(4))) uint32_t *cptr;
      const cptr ct = (cptr)(uintptr_t)table;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        uint32_t s = 0;
#pragma unroll
        for (int l = 0; l < 64; ++l) s += ct[(uint32_t)__builtin_amdgcn_readlane((int)idx[u], l)];
        v[u] = s;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  if (acc == 0x12345678u) out[gid] = acc;   // keeps the loads alive, (almost) never stores
}

struct Result { double gathers_per_s, clk_per_gather_cu, ms; uint64_t gathers; };

template <int MODE, int U>
Result run(const uint32_t *table, uint32_t words, int occ, int cus, double clock_ghz, const u32x2 *recs,
           uint32_t nrec, uint32_t *out, uint64_t target_gathers) {
  // occupancy by LDS padding: occ workgroups of 256 threads per CU = occ waves per SIMD
  size_t lds = (160 * 1024) / occ - 1024;
  if (MODE == M_LDS && lds < (size_t)words * 4) lds = (size_t)words * 4;
  auto k = gather_kernel<MODE, U>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int per_cu = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, THREADS, lds));
  const unsigned grid = (unsigned)(cus * std::min(per_cu, occ));
  const uint64_t lanes = (uint64_t)grid * THREADS;
  uint32_t iters = (uint32_t)std::max<uint64_t>(1, target_gathers / (lanes * U));
  if (MODE == M_SCALAR) iters = std::max<uint32_t>(1, iters / 8);
  uint64_t gathers = lanes * U * (uint64_t)iters;
  if (MODE == M_STREAM) {
    const uint32_t per = (nrec / grid) / (U * THREADS) * (U * THREADS);
    gathers = (uint64_t)per * grid;
  }
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {   // first = warm-up (table into the caches)
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(k, dim3(grid), dim3(THREADS), lds, 0, table, words - 1, iters, recs, nrec, out);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    if (rep) best = std::min(best, ms);
  }
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  Result r;
  r.ms = best; r.gathers = gathers;
  r.gathers_per_s = gathers / (best * 1e-3);
  r.clk_per_gather_cu = clock_ghz * 1e9 * cus / r.gathers_per_s;
  (void)per_cu;
  return r;
}

template <int MODE>
Result run_mode(const uint32_t *table, uint32_t words, int occ, int cus, double ghz, const u32x2 *recs, uint32_t nrec,
                uint32_t *out, uint64_t target) {
  return run<MODE, 12>(table, words, occ, cus, ghz, recs, nrec, out, target);
}

int main(int argc, char **argv) {
  bool quick = false, ceiling_only = false;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--quick")) quick = true;
    if (!strcmp(argv[i], "--ceiling")) ceiling_only = true;   // one line: stream mode, 4 MB, 3 waves/SIMD (bench.py)
  }
  int dev = 0, cus = 0, khz = 0;
  CK(hipGetDevice(&dev));
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, dev));
  const double ghz = khz / 1e6;
  const uint32_t max_words = 4u << 20;   // 16 MB
  std::vector<uint32_t> h(max_words);
  for (uint32_t i = 0; i < max_words; ++i) h[i] = i * 2654435761u;
  uint32_t *table = nullptr, *out = nullptr;
  CK(hipMalloc(&table, (size_t)max_words * 4));
  CK(hipMemcpy(table, h.data(), (size_t)max_words * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&out, (size_t)cus * 8 * THREADS * 4));
  // record stream for the stream mode: 100 M 8-byte records with random keys (config 3's sweep)
  const uint32_t nrec = quick ? 25u << 20 : 100u * 1000u * 1000u;
  u32x2 *recs = nullptr;
  CK(hipMalloc(&recs, (size_t)nrec * 8));
  {
    std::vector<u32x2> hr(1u << 22);
    uint32_t s = 777;
    for (auto &r : hr) { s = s * 1664525u + 1013904223u; r.x = (s >> 4) ^ (s << 9); r.y = 1; }
    for (size_t off = 0; off < nrec; off += hr.size())
      CK(hipMemcpy(recs + off, hr.data(), std::min<size_t>(hr.size(), nrec - off) * 8, hipMemcpyHostToDevice));
  }
  const uint64_t target = quick ? 50ull * 1000 * 1000 : 200ull * 1000 * 1000;
  printf("{\"device_cus\": %d, \"clock_ghz\": %.3f, \"records\": %u}\n", cus, ghz, nrec);
  const uint32_t sizes_kb[] = {4, 32, 256, 1024, 2048, 4096, 8192, 16384};
  const int occs[] = {1, 2, 3, 4, 8};
  for (int m = 0; m < M_COUNT; ++m) {
    for (uint32_t kb : sizes_kb) {
      if (m == M_LDS && kb > 32) continue;
      if (ceiling_only && !(m == M_STREAM && kb == 4096)) continue;
      if (quick && !(kb == 4 || kb == 1024 || kb == 4096 || kb == 16384)) continue;
      for (int occ : occs) {
        if (ceiling_only && occ != 3) continue;
        if (quick && (occ == 2 || occ == 8)) continue;
        const uint32_t words = kb * 256u;
        Result r{};
        switch (m) {
          case M_GLOBAL: r = run_mode<M_GLOBAL>(table, words, occ, cus, ghz, recs, nrec, out, target); break;
          case M_NT: r = run_mode<M_NT>(table, words, occ, cus, ghz, recs, nrec, out, target); break;
          case M_BUFFER: r = run_mode<M_BUFFER>(table, words, occ, cus, ghz, recs, nrec, out, target); break;
          case M_X2: r = run_mode<M_X2>(table, words, occ, cus, ghz, recs, nrec, out, target); break;
          case M_SORTED: r = run_mode<M_SORTED>(table, words, occ, cus, ghz, recs, nrec, out, target); break;
          case M_STREAM: r = run_mode<M_STREAM>(table, words, occ, cus, ghz, recs, nrec, out, target); break;
          case M_SCALAR: r = run_mode<M_SCALAR>(table, words, occ, cus, ghz, recs, nrec, out, target); break;
          case M_LDS: r = run_mode<M_LDS>(table, words, occ, cus, ghz, recs, nrec, out, target); break;
        }
        printf("{\"mode\": \"%s\", \"table_kb\": %u, \"waves_per_simd\": %d, \"gathers\": %llu, \"ms\": %.4f, "
               "\"gathers_per_s\": %.4g, \"clk_per_gather_per_cu\": %.3f}\n",
               kModeName[m], kb, occ, (unsigned long long)r.gathers, r.ms, r.gathers_per_s, r.clk_per_gather_cu);
        fflush(stdout);
      }
    }
  }
  CK(hipFree(table)); CK(hipFree(out)); CK(hipFree(recs));
  return 0;
}
