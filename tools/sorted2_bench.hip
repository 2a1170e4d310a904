// tools/sorted2_bench.hip -- feasibility of a weight-sorted sweep for tiles with PAIRWISE factors
// (config 3b / 5b: 6 unary + 8 pairwise memberships per variable): would the remedy of
// sorted_sweep_kernel pay there too?  One workgroup = one super-tile of NV
// variables; its 14 NV records (16 bytes: weight id, owner slot | flags, other endpoint, feature
// value) are sorted by weight id.
//   pass 1  stream the records, gather w32[wid] (sorted) and the other endpoint's assignment on
//           both chains (random inside a near window: offsets 1 / 7 / 101, or a far window:
//           V/8 + 3), add w * d(other) to the owner's two fixed-point sums in LDS
//   pass 2  a draw per variable and chain, assignments stored
//   pass 3  (learning) stream the records again, gather the other endpoint again, and add the
//           gradient of every pairwise record of an evidence owner to grad[wid] (int64 atomics in
//           sorted order; about a third of them are non-zero)
// Compare with sweep_kernel<LEARN=true> on config 3b: 1.036 ms per colour launch of 5 M variables
// (profiles/r03/summary_cfg3b.txt), inference first sweep 0.30 ms.
// Not product code.   sorted2_bench [--vars N]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int K = 6;
constexpr int PER_VAR = 14;

__device__ __forceinline__ long long fix32(double p) {
  const double magic = 6755399441055744.0 / 4294967296.0;
  const double s = p + magic;
  long long bits, mb;
  __builtin_memcpy(&bits, &s, 8);
  __builtin_memcpy(&mb, &magic, 8);
  return bits - mb;
}

template <int THREADS, bool LEARN>
__global__ void __launch_bounds__(THREADS) sorted2_kernel(const u32x4 *__restrict__ recs, const float *__restrict__ w32,
                                                          uint32_t *__restrict__ a_free, uint32_t *__restrict__ a_evid,
                                                          uint32_t nv, uint32_t n_super, uint32_t v_first,
                                                          unsigned long long *__restrict__ grad) {
  extern __shared__ unsigned long long acc[];     // [2 * nv]: free chain, evidence chain
  const uint32_t t = threadIdx.x;
  const uint32_t per = nv * PER_VAR;
  for (uint32_t st = blockIdx.x; st < n_super; st += gridDim.x) {
    for (uint32_t i = t; i < 2 * nv; i += THREADS) acc[i] = 0;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(recs + (size_t)st * per), 0, (int)(per * 16u), 0x00020000);
    const uint32_t steps = (per + K * THREADS - 1) / (K * THREADS);
    for (int pass = 0; pass < (LEARN ? 2 : 1); ++pass) {
      u32x4 rec[K];
#pragma unroll
      for (int k = 0; k < K; ++k) rec[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(t * 16u), (int)(k * THREADS * 16), 2);
      for (uint32_t it = 0; it < steps; ++it) {
        float w[K];
        uint32_t xf[K], xe[K];
        if (pass == 0) {
#pragma unroll
          for (int k = 0; k < K; ++k) w[k] = w32[rec[k].x & 0x7FFFFFFu];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) { xe[k] = a_evid[rec[k].z]; xf[k] = LEARN ? a_free[rec[k].z] : 0u; }
        u32x4 cur[K];
#pragma unroll
        for (int k = 0; k < K; ++k) cur[k] = rec[k];
        const uint32_t nxt = (it + 1) * (K * THREADS * 16u);
#pragma unroll
        for (int k = 0; k < K; ++k) rec[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(t * 16u), (int)(nxt + k * THREADS * 16), 2);
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const uint32_t owner = cur[k].y & 0x1FFFu;
          const bool pair = cur[k].y & 0x2000u, live = cur[k].y & 0x4000u;
          const uint32_t fbits = cur[k].w;
          float f;
          __builtin_memcpy(&f, &fbits, 4);
          if (pass == 0) {
            const double de = pair ? (xe[k] ? 2.0 : -2.0) : 2.0, df = pair ? (xf[k] ? 2.0 : -2.0) : 2.0;
            if (live) {
              atomicAdd(&acc[nv + owner], (unsigned long long)fix32((double)w[k] * (double)f * de));
              if (LEARN) atomicAdd(&acc[owner], (unsigned long long)fix32((double)w[k] * (double)f * df));
            }
          } else if (live && pair) {
            // gradient: owner's draws (LDS) against the other endpoint's values
            const uint32_t bits = (uint32_t)acc[owner];
            const int g = (int)((bits & 1u) == xf[k]) - (int)(((bits >> 1) & 1u) == xe[k]);
            if ((bits & 4u) && g) atomicAdd(&grad[cur[k].x & 0x7FFFFFFu], (unsigned long long)((long long)g << 31));
          }
        }
      }
      __syncthreads();
      if (pass == 0) {
        for (uint32_t i = t; i < nv; i += THREADS) {
          const uint32_t p = v_first + st * nv + i;
          const double xe_ = (double)(long long)acc[nv + i] * (1.0 / 4294967296.0);
          const double xf_ = (double)(long long)acc[i] * (1.0 / 4294967296.0);
          const uint32_t h = (p * 2654435761u) >> 8;
          const float r1 = (float)(h & 0xFFFF) / 65536.0f, r2 = (float)((h >> 8) & 0xFFFF) / 65536.0f;
          const uint32_t pe = r1 * (1.0f + __expf((float)-xe_)) < 1.0f ? 1u : 0u;
          const uint32_t pf = r2 * (1.0f + __expf((float)-xf_)) < 1.0f ? 1u : 0u;
          a_evid[p] = pe;
          if (LEARN) { a_free[p] = pf; acc[i] = pf | (pe << 1) | ((h & 1u) << 2); }   // (half the owners are evidence: they learn)
        }
        __syncthreads();
      }
    }
  }
}

int main(int argc, char **argv) {
  uint32_t V = 10u * 1000 * 1000, W = 1u << 20;
  for (int i = 1; i + 1 < argc; ++i)
    if (!strcmp(argv[i], "--vars")) V = (uint32_t)strtoul(argv[i + 1], nullptr, 10);
  int dev = 0, cus = 0;
  CK(hipGetDevice(&dev));
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  float *w32 = nullptr;
  uint32_t *af = nullptr, *ae = nullptr;
  unsigned long long *grad = nullptr;
  {
    std::vector<float> hw(W);
    for (uint32_t i = 0; i < W; ++i) hw[i] = (float)((i * 2654435761u >> 8) & 0xFFFF) / 65536.0f - 0.5f;
    CK(hipMalloc(&w32, (size_t)W * 4));
    CK(hipMemcpy(w32, hw.data(), (size_t)W * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&af, (size_t)V * 4)); CK(hipMalloc(&ae, (size_t)V * 4));
    CK(hipMemset(af, 0, (size_t)V * 4)); CK(hipMemset(ae, 0, (size_t)V * 4));
    CK(hipMalloc(&grad, (size_t)W * 8)); CK(hipMemset(grad, 0, (size_t)W * 8));
  }
  // one colour launch: the first half of the variables (the other colour is what they read)
  const uint32_t Vl = V / 2;
  struct Cfg { uint32_t nv; int threads, wg_per_cu; };
  const Cfg cfgs[] = {{4096, 512, 2}, {4096, 1024, 2}, {2048, 512, 4}, {8192, 1024, 1}};
  std::vector<u32x4> h;
  u32x4 *recs = nullptr;
  CK(hipMalloc(&recs, (size_t)Vl * PER_VAR * 16 + (1u << 20)));
  printf("{\"device_cus\": %d, \"vars_per_launch\": %u, \"records_per_launch\": %llu, \"weights\": %u}\n", cus, Vl,
         (unsigned long long)Vl * PER_VAR, W);
  for (const Cfg &c : cfgs) {
    const uint32_t per = c.nv * PER_VAR, n_super = Vl / c.nv;
    h.resize((size_t)n_super * per);
    const uint32_t distinct = std::min<uint32_t>(n_super, 64);
    {
      std::vector<std::thread> th;
      for (uint32_t s = 0; s < distinct; ++s)
        th.emplace_back([&, s]() {
          std::mt19937 rng(99 + s);
          u32x4 *r = h.data() + (size_t)s * per;
          const float one = 1.0f;
          uint32_t fb; memcpy(&fb, &one, 4);
          for (uint32_t i = 0; i < per; ++i) {
            const uint32_t owner = i / PER_VAR, j = i % PER_VAR;
            r[i].x = rng() % W;
            r[i].y = owner | (j >= 6 ? 0x2000u : 0u) | 0x4000u;
            r[i].z = 0;      // filled per super-tile below (depends on the super-tile's position)
            r[i].w = fb;
          }
          std::sort(r, r + per, [](const u32x4 &a, const u32x4 &b) { return a.x < b.x; });
        });
      for (auto &t : th) t.join();
    }
    {
      std::vector<std::thread> th;
      const uint32_t T = 16;
      for (uint32_t tt = 0; tt < T; ++tt)
        th.emplace_back([&, tt]() {
          const int offs[4] = {1, 7, 101, (int)(V / 8 + 3)};
          for (uint32_t s = tt; s < n_super; s += T) {
            u32x4 *r = h.data() + (size_t)s * per;
            if (s >= distinct) memcpy(r, h.data() + (size_t)(s % distinct) * per, (size_t)per * 16);
            std::mt19937 rng(7 + s);
            for (uint32_t i = 0; i < per; ++i) {
              const uint32_t owner = s * c.nv + (r[i].y & 0x1FFFu);
              const int o = offs[rng() & 3] * ((rng() & 4) ? 1 : -1);
              // the other endpoint lives in the OTHER colour: the second half of the variables
              r[i].z = (r[i].y & 0x2000u) ? Vl + (uint32_t)(((long long)owner + o + Vl) % Vl) : owner;
            }
          }
        });
      for (auto &t : th) t.join();
    }
    CK(hipMemcpy(recs, h.data(), (size_t)n_super * per * 16, hipMemcpyHostToDevice));
    const size_t lds = (size_t)c.nv * 16;
    auto launch = [&](auto kern, const char *what) {
      CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      int per_cu = 0;
      CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, c.threads, lds));
      const unsigned grid = (unsigned)std::min<uint32_t>(n_super, (uint32_t)(cus * std::min(per_cu, c.wg_per_cu)));
      hipEvent_t a, b;
      CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(c.threads), lds, 0, recs, w32, af, ae, c.nv, n_super, 0u, grad);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        if (rep) best = std::min(best, ms);
      }
      CK(hipGetLastError());
      printf("{\"what\": \"%s\", \"nv\": %u, \"threads\": %d, \"wg_per_cu\": %d, \"occupancy_limit\": %d, \"super_tiles\": %u, \"ms_per_colour_launch\": %.4f}\n",
             what, c.nv, c.threads, c.wg_per_cu, per_cu, n_super, best);
      fflush(stdout);
    };
    if (c.threads == 512) { launch(sorted2_kernel<512, true>, "learn"); launch(sorted2_kernel<512, false>, "infer"); }
    else { launch(sorted2_kernel<1024, true>, "learn"); launch(sorted2_kernel<1024, false>, "infer"); }
  }
  return 0;
}
