// tools/atomic_bench.hip -- how fast do scattered int64 atomic adds retire on MI355X?
//   mode agent : atomicAdd (agent scope) into ONE table                       -- what the sweeps did
//   mode u32 / u32wide : 4-byte agent-scope adds (dense words / at the 8-byte table's addresses)
//   mode xcd   : workgroup-scope atomic adds into the table copy of the issuing XCD (XCC_ID), i.e.
//                in that XCD's own L2; the eight copies are summed afterwards
// Prints Gatomics/s and checks the sums (lost updates would show).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ inline unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u; }

__device__ inline unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}

template <int MODE>
__global__ void __launch_bounds__(256) k(long long *tab, unsigned W, unsigned per_thread, unsigned *xcd_seen) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  long long *t = tab;
  if (MODE == 1) {
    const unsigned x = xcc_id();
    t = tab + (size_t)x * W;
    if (threadIdx.x == 0) atomicOr(&xcd_seen[0], 1u << x);
  }
  for (unsigned i = 0; i < per_thread; ++i) {
    const unsigned w = hash32(tid * per_thread + i) % W;
    if (MODE == 0) atomicAdd((unsigned long long *)&t[w], 1ull);
    else if (MODE == 2) atomicAdd((unsigned *)t + w, 1u);                       // 4-byte adds, dense table of W words
    else if (MODE == 3) atomicAdd((unsigned *)t + 2 * (size_t)w, 1u);           // 4-byte adds at the 8-byte adds' addresses
    else __hip_atomic_fetch_add(&t[w], 1ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

int main(int argc, char **argv) {
  unsigned W = 1000000, per = 64, blocks = 256 * 12;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--W")) W = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--per")) per = atoi(argv[++i]);
  }
  long long *tab; unsigned *seen;
  CHECK(hipMalloc(&tab, (size_t)8 * W * 8));
  CHECK(hipMalloc(&seen, 4));
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  const double total = (double)blocks * 256 * per;
  for (int mode = 0; mode < 4; ++mode) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      CHECK(hipMemset(tab, 0, (size_t)8 * W * 8)); CHECK(hipMemset(seen, 0, 4));
      CHECK(hipEventRecord(a));
      if (mode == 0) k<0><<<blocks, 256>>>(tab, W, per, seen);
      else if (mode == 1) k<1><<<blocks, 256>>>(tab, W, per, seen);
      else if (mode == 2) k<2><<<blocks, 256>>>(tab, W, per, seen);
      else k<3><<<blocks, 256>>>(tab, W, per, seen);
      CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
      float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    std::vector<long long> h((size_t)8 * W); unsigned s;
    CHECK(hipMemcpy(h.data(), tab, h.size() * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&s, seen, 4, hipMemcpyDeviceToHost));
    long long sum = 0;
    if (mode >= 2) { const unsigned *u = (const unsigned *)h.data(); for (size_t i = 0; i < h.size() * 2; ++i) sum += u[i]; }
    else for (long long v : h) sum += v;
    printf("{\"mode\": \"%s\", \"W\": %u, \"atomics\": %.0f, \"ms\": %.4f, \"G_per_s\": %.2f, \"sum_ok\": %s, \"xcd_mask\": %u}\n",
           mode == 0 ? "agent" : mode == 1 ? "xcd" : mode == 2 ? "u32" : "u32wide", W, total, best, total / best * 1e-6, sum == (long long)total ? "true" : "false", s);
  }
  return 0;
}
