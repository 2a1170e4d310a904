// tools/sorted_bench.hip -- feasibility / ceiling of the weight-sorted super-tile sweep
// (DESIGN.md 3.1b): a super-tile = NV consecutive variables whose records (8 bytes:
// {weight id, owner slot | step index}) are stored SORTED BY WEIGHT ID; the workgroup streams
// them, gathers w32[wid] -- neighbouring lanes now ask for neighbouring weights: few L2 requests
// per wave-instruction instead of one per lane -- and accumulates the potential difference of
// the record's owner in LDS with a 64-bit fixed-point atomic add; a second phase turns every
// variable's sum into a draw (exp) and stores 4 bytes.
//
// Not product code: a calibration tool (run by hand; results in profiles/r03/sorted_bench.jsonl).
//
//   sorted_bench [--records N] [--weights W] [--ceiling]
// one JSON line per (NV, threads, workgroups per CU, sorted?, atomics?); --ceiling: only the
// product kernel's shape, on one full round of 256 super-tiles (bench.py: roofline.secondary)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
constexpr int K = 20;         // records in flight per lane (the product's SORT_K)
constexpr int PER_VAR = 10;   // records per variable

// fixed point 2^-32 through the round-to-nearest of an f64 add (|p| < 2^19)
__device__ __forceinline__ long long fix32(double p) {
  const double magic = 6755399441055744.0 / 4294967296.0;   // 1.5 * 2^52 * 2^-32
  const double s = p + magic;
  long long bits, mb;
  __builtin_memcpy(&bits, &s, 8);
  __builtin_memcpy(&mb, &magic, 8);
  return bits - mb;
}

template <int THREADS, bool ATOMICS>
__global__ void __launch_bounds__(THREADS) sorted_kernel(const u32x2 *__restrict__ recs, const float *__restrict__ w32,
                                                         uint32_t nv, uint32_t n_super, uint32_t *__restrict__ out) {
  extern __shared__ unsigned long long acc[];
  const uint32_t t = threadIdx.x;
  const uint32_t per = nv * PER_VAR;                 // records per super-tile
  long long sink = 0;
  for (uint32_t st = blockIdx.x; st < n_super; st += gridDim.x) {
    for (uint32_t i = t; i < nv; i += THREADS) acc[i] = 0;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(recs + (size_t)st * per), 0, (int)(per * 8u), 0x00020000);
    u32x2 rec[K];
#pragma unroll
    for (int k = 0; k < K; ++k) rec[k] = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(t * 8u), (int)(k * THREADS * 8), 2);
    const uint32_t steps = (per + K * THREADS - 1) / (K * THREADS);
    for (uint32_t it = 0; it < steps; ++it) {
      float w[K];
#pragma unroll
      for (int k = 0; k < K; ++k) w[k] = w32[rec[k].x];
      u32x2 cur[K];
#pragma unroll
      for (int k = 0; k < K; ++k) cur[k] = rec[k];
      const uint32_t nxt = (it + 1) * (K * THREADS * 8u);
#pragma unroll
      for (int k = 0; k < K; ++k) rec[k] = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(t * 8u), (int)(nxt + k * THREADS * 8), 2);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const long long q = fix32((double)w[k] * 2.0);
        const uint32_t owner = cur[k].y & 0xFFFFu;
        if (ATOMICS) atomicAdd(&acc[owner < nv ? owner : 0], (unsigned long long)q);
        else sink += q + owner;
      }
    }
    __syncthreads();
    for (uint32_t i = t; i < nv; i += THREADS) {
      const double x = (double)(long long)acc[i] * (1.0 / 4294967296.0);
      const float q = 0.37f * (1.0f + __expf((float)-x));
      out[(size_t)st * nv + i] = q < 1.0f ? 1u : 0u;
    }
    __syncthreads();
  }
  if (sink == 0x1234567) out[0] = 7;
}

// --rec6 (round 4): the same loop over 6-BYTE records {weight id: 24 bits, owner slot: 14,
// index of d: 10} -- a quarter less of the stream.  A lane takes two consecutive records per 12-byte
// load (buffer_load_dwordx3), K / 2 loads in flight; everything else as sorted_kernel.
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
template <int THREADS>
__global__ void __launch_bounds__(THREADS) sorted6_kernel(const uint32_t *__restrict__ recs, const float *__restrict__ w32,
                                                          uint32_t nv, uint32_t n_super, uint32_t *__restrict__ out) {
  extern __shared__ unsigned long long acc[];
  const uint32_t t = threadIdx.x;
  const uint32_t per = nv * PER_VAR;                 // records per super-tile (even)
  constexpr int L = K / 2;                           // 12-byte loads per lane and step
  for (uint32_t st = blockIdx.x; st < n_super; st += gridDim.x) {
    for (uint32_t i = t; i < nv; i += THREADS) acc[i] = 0;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void *)((const char *)recs + (size_t)st * per * 6u), 0, (int)(per * 6u), 0x00020000);
    u32x3 rec[L];
#pragma unroll
    for (int k = 0; k < L; ++k) rec[k] = __builtin_amdgcn_raw_buffer_load_b96(rs, (int)(t * 12u), (int)(k * THREADS * 12), 2);
    const uint32_t steps = (per + K * THREADS - 1) / (K * THREADS);
    for (uint32_t it = 0; it < steps; ++it) {
      float w[K];
      uint32_t own[K];
#pragma unroll
      for (int k = 0; k < L; ++k) {
        const u32x3 v = rec[k];
        const uint32_t b_lo = (v.y >> 16) | (v.z << 16);
        w[2 * k] = w32[v.x & 0xFFFFFFu];
        w[2 * k + 1] = w32[b_lo & 0xFFFFFFu];
        own[2 * k] = (v.x >> 24) | ((v.y & 0x3Fu) << 8);
        own[2 * k + 1] = (b_lo >> 24) | (((v.z >> 16) & 0x3Fu) << 8);
      }
      const uint32_t nxt = (it + 1) * (K * THREADS * 6u);
#pragma unroll
      for (int k = 0; k < L; ++k) rec[k] = __builtin_amdgcn_raw_buffer_load_b96(rs, (int)(t * 12u), (int)(nxt + k * THREADS * 12), 2);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const long long q = fix32((double)w[k] * 2.0);
        atomicAdd(&acc[own[k] < nv ? own[k] : 0], (unsigned long long)q);
      }
    }
    __syncthreads();
    for (uint32_t i = t; i < nv; i += THREADS) {
      const double x = (double)(long long)acc[i] * (1.0 / 4294967296.0);
      const float q = 0.37f * (1.0f + __expf((float)-x));
      out[(size_t)st * nv + i] = q < 1.0f ? 1u : 0u;
    }
    __syncthreads();
  }
}

struct Cfg { uint32_t nv; int threads, wg_per_cu; bool sorted, atomics; };

int main(int argc, char **argv) {
  uint64_t nrec = 100ull * 1000 * 1000;
  uint32_t W = 1u << 20;
  bool ceiling_only = false;    // one line: the product kernel's shape (16 384 variables, 1024 threads, 1 per CU)
  bool rec6 = false;            // the product's shape over 8-byte and over 6-byte records, side by side
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--rec6")) rec6 = true;
    if (!strcmp(argv[i], "--ceiling")) { ceiling_only = true; nrec = 42ull * 1000 * 1000; }
    if (i + 1 < argc && !strcmp(argv[i], "--records")) nrec = strtoull(argv[i + 1], nullptr, 10);
    if (i + 1 < argc && !strcmp(argv[i], "--weights")) W = (uint32_t)strtoul(argv[i + 1], nullptr, 10);
  }
  int dev = 0, cus = 0;
  CK(hipGetDevice(&dev));
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  float *w32 = nullptr;
  {
    std::vector<float> hw(W);
    for (uint32_t i = 0; i < W; ++i) hw[i] = (float)((i * 2654435761u >> 8) & 0xFFFF) / 65536.0f - 0.5f;
    CK(hipMalloc(&w32, (size_t)W * 4));
    CK(hipMemcpy(w32, hw.data(), (size_t)W * 4, hipMemcpyHostToDevice));
  }
  u32x2 *recs = nullptr;
  uint32_t *out = nullptr;
  CK(hipMalloc(&recs, nrec * 8 + (1u << 20)));
  CK(hipMalloc(&out, nrec / PER_VAR * 4 + (1u << 20)));
  printf("{\"device_cus\": %d, \"records\": %llu, \"weights\": %u}\n", cus, (unsigned long long)nrec, W);
  const Cfg cfgs[] = {
      {8192, 512, 2, false, true}, {8192, 512, 2, true, true}, {8192, 512, 2, true, false},
      {8192, 1024, 2, true, true}, {8192, 256, 2, true, true},
      {4096, 512, 4, true, true}, {4096, 256, 4, true, true},
      {16384, 1024, 1, true, true}, {16384, 512, 1, true, true},
      {2560, 256, 6, true, true},
  };
  std::vector<u32x2> h(nrec);
  for (const Cfg &c : cfgs) {
    if ((ceiling_only || rec6) && !(c.nv == 16384 && c.threads == 1024 && c.sorted && c.atomics)) continue;
    const uint32_t per = c.nv * PER_VAR;
    const uint32_t n_super = (uint32_t)(nrec / per);
    // super-tile contents: random weight ids (sorted or not), owners = a random permutation of the
    // variable slots, PER_VAR records each; 64 distinct super-tiles repeated
    const uint32_t distinct = std::min<uint32_t>(n_super, 64);
    {
      std::vector<std::thread> th;
      for (uint32_t s = 0; s < distinct; ++s)
        th.emplace_back([&, s]() {
          std::mt19937 rng(1234 + s);
          u32x2 *r = h.data() + (size_t)s * per;
          for (uint32_t i = 0; i < per; ++i) { r[i].x = rng() % W; r[i].y = (i / PER_VAR) | (1u << 16); }
          std::shuffle(r, r + per, rng);   // owners in random order ...
          if (c.sorted) std::sort(r, r + per, [](const u32x2 &a, const u32x2 &b) { return a.x < b.x; });   // ... by weight
        });
      for (auto &t : th) t.join();
    }
    for (uint32_t s = distinct; s < n_super; ++s)
      memcpy(h.data() + (size_t)s * per, h.data() + (size_t)(s % distinct) * per, (size_t)per * 8);
    CK(hipMemcpy(recs, h.data(), (size_t)n_super * per * 8, hipMemcpyHostToDevice));
    const size_t lds = (size_t)c.nv * 8;
    auto launch = [&](auto kern) {
      CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      int per_cu = 0;
      CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, c.threads, lds));
      const unsigned grid = (unsigned)(cus * std::min(per_cu, c.wg_per_cu));
      hipEvent_t a, b;
      CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(c.threads), lds, 0, recs, w32, c.nv, n_super, out);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        if (rep) best = std::min(best, ms);
      }
      CK(hipGetLastError());
      const double n = (double)n_super * per;
      printf("{\"nv\": %u, \"threads\": %d, \"wg_per_cu\": %d, \"occupancy_limit\": %d, \"sorted\": %s, \"atomics\": %s, "
             "\"records\": %.0f, \"ms\": %.4f, \"records_per_s\": %.4g, \"stream_GBps\": %.1f}\n",
             c.nv, c.threads, c.wg_per_cu, per_cu, c.sorted ? "true" : "false", c.atomics ? "true" : "false", n, best,
             n / (best * 1e-3), n * 8 / (best * 1e-3) / 1e9);
      fflush(stdout);
    };
    if (rec6) {
      // the same super-tiles as 6-byte records (d index 1), two per 12 bytes
      std::vector<uint8_t> h6((size_t)n_super * per * 6 + 16);
      for (size_t i = 0; i < (size_t)n_super * per; ++i) {
        const uint64_t v = (uint64_t)(h[i].x & 0xFFFFFFu) | ((uint64_t)(h[i].y & 0x3FFFu) << 24) | (1ull << 38);
        memcpy(h6.data() + i * 6, &v, 6);
      }
      uint32_t *recs6 = nullptr;
      CK(hipMalloc(&recs6, h6.size()));
      CK(hipMemcpy(recs6, h6.data(), h6.size(), hipMemcpyHostToDevice));
      launch(sorted_kernel<1024, true>);
      auto kern = sorted6_kernel<1024>;
      CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipEvent_t a, b;
      CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(kern, dim3(cus), dim3(1024), lds, 0, (const uint32_t *)recs6, w32, c.nv, n_super, out);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        if (rep) best = std::min(best, ms);
      }
      CK(hipGetLastError());
      const double n = (double)n_super * per;
      printf("{\"nv\": %u, \"threads\": 1024, \"wg_per_cu\": 1, \"record_bytes\": 6, \"records\": %.0f, \"ms\": %.4f, "
             "\"records_per_s\": %.4g, \"stream_GBps\": %.1f}\n", c.nv, n, best, n / (best * 1e-3), n * 6 / (best * 1e-3) / 1e9);
      CK(hipFree(recs6));
      continue;
    }
    if (c.threads == 256) { if (c.atomics) launch(sorted_kernel<256, true>); else launch(sorted_kernel<256, false>); }
    if (c.threads == 512) { if (c.atomics) launch(sorted_kernel<512, true>); else launch(sorted_kernel<512, false>); }
    if (c.threads == 1024) { if (c.atomics) launch(sorted_kernel<1024, true>); else launch(sorted_kernel<1024, false>); }
  }
  return 0;
}
