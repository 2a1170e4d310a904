// device_build.hip -- the heavy phases of the host-side graph build as HIP kernels (SURVEY.md 8 f1,
// "optional GPU build"): what graph_compile.cc / dwx_api.cc do on the host's cores with counting
// sorts and per-record walks, done on the device the sampler is created on, from the columns that
// are uploaded anyway.  The host builders stay, as the checker (tests/test_gpu_parity.py compares
// the buffers byte for byte) and for the sanitizer harness, whose link has a stub for this file.
//
//   build_sorted_records   the weight-sorted second copy of the boolean all-unary tiles' records
//                          (graph_compile.cc: build_sorted_layout's radix sorts per super-tile)
//   build_static_tables    a plan level's static update counts T and curvature bounds h per chunk
//                          (dwx_api.cc: build_level's walk over every SGD-triggering variable's records)
//
// Sorting itself is rocPRIM's radix sort (a plain library sort, as the guide allows for plain
// library operations); everything around it is written here.  Replaces nothing of the reference
// (its construct_index sorts per variable with std::sort, src/factor_graph.cc:90-199).
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <stdexcept>
#include <string>

#include "device_build.h"

namespace dwx {
namespace devb {

namespace {
inline void check(hipError_t e, const char *what) {
  if (e != hipSuccess) throw std::runtime_error(std::string("HIP error in ") + what + ": " + hipGetErrorString(e));
}
#define DEVB_HIP(x) check((x), #x)

constexpr uint32_t EMIT_THREADS = 1024;

__device__ __forceinline__ float rec_d(const EdgeRec8 c) {
  const int sh = (int)((c.key >> REC8_HIT_SHIFT) & 3u) - 1, sm = (int)((c.key >> REC8_MISS_SHIFT) & 3u) - 1;
  return (float)(sh - sm) * c.f;     // exact: a factor in {-2 .. 2}
}

// One workgroup per super-tile: its records are ONE contiguous range of the variable-major stream
// (consecutive tiles), in (owner, row) order.  Every record with d != 0 is emitted, in that order
// (an ordered compaction: ballot + wave prefix + an LDS scan over the 16 waves), at the super-tile's
// offset of the key / value arrays: key = super-tile index << 32 | weight id, value = the SortRec8
// `od` word (index of d in the table of distinct values << owner bits | owner's slot).  A stable
// sort by key then leaves every super-tile's records sorted by (weight id, owner).
__global__ void __launch_bounds__(EMIT_THREADS)
emit_sorted_kernel(const TileDesc *tiles, const EdgeRec *edges, const EdgeRec8 *edges8, const SuperTile *supers,
                   uint32_t n_supers, const uint32_t *dbits, uint32_t n_dbits, unsigned long long *keys, uint32_t *vals) {
  __shared__ uint32_t s_e0[SORT_TV_SLOTS + 1], s_v0[SORT_TV_SLOTS + 1], s_wave[EMIT_THREADS / 64 + 1], s_db[SORT_MAX_DVALS];
  const uint32_t si = blockIdx.x, t = threadIdx.x;
  if (si >= n_supers) return;
  const SuperTile st = supers[si];
  for (uint32_t i = t; i < st.ntiles; i += EMIT_THREADS) { s_e0[i] = tiles[st.tile0 + i].e0; s_v0[i] = tiles[st.tile0 + i].v0; }
  if (t == 0) { const TileDesc last = tiles[st.tile0 + st.ntiles - 1]; s_e0[st.ntiles] = last.e0 + last.nedges; }
  for (uint32_t i = t; i < n_dbits; i += EMIT_THREADS) s_db[i] = dbits[i];
  __syncthreads();
  const uint64_t e_begin = s_e0[0], e_end = s_e0[st.ntiles];
  const uint64_t out0 = ((uint64_t)st.hi << 32) | st.lo;
  uint32_t done = 0;      // records emitted so far (workgroup-uniform)
  for (uint64_t base = e_begin; base < e_end; base += EMIT_THREADS) {
    const uint64_t e = base + t;
    const bool in = e < e_end;
    EdgeRec8 c{0u, 0.0f};
    if (in) c = edges8[e];
    const float dv = in ? rec_d(c) : 0.0f;
    const bool keep = dv != 0.0f;
    const unsigned long long bal = __ballot(keep);
    const uint32_t lane = t & 63u, wave = t >> 6;
    const uint32_t before = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t wbase = 0, total = 0;
    for (uint32_t w = 0; w < EMIT_THREADS / 64; ++w) { const uint32_t n = s_wave[w]; if (w < wave) wbase += n; total += n; }
    if (keep) {
      // the record's tile: the last one whose first record is <= e (<= 64 tiles: a short search)
      uint32_t lo = 0, hi = st.ntiles;
      while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (s_e0[mid] <= e) lo = mid; else hi = mid; }
      const uint32_t owner_lane = edges[e].packed >> EDGE_OWNER_SHIFT;
      const uint32_t slot = s_v0[lo] + owner_lane - st.v0;
      uint32_t bits;
      __builtin_memcpy(&bits, &dv, 4);
      uint32_t a = 0, b = n_dbits;      // lower_bound over the ascending bit patterns
      while (a < b) { const uint32_t mid = (a + b) >> 1; if (s_db[mid] < bits) a = mid + 1; else b = mid; }
      const uint64_t at = out0 + done + wbase + before;
      keys[at] = ((unsigned long long)si << 32) | (c.key & REC8_WID_MASK);
      vals[at] = ((a + 1u) << SORT_OWNER_BITS) | slot;
    }
    done += total;
    __syncthreads();      // s_wave is rewritten by the next chunk
  }
}

__global__ void __launch_bounds__(256)
compose_sorted_kernel(const unsigned long long *keys, const uint32_t *vals, uint64_t n, SortRec8 *out) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = SortRec8{(uint32_t)keys[i], vals[i]};
}
// ---- static update counts T and curvature bounds h of a plan level (dwx_api.cc: build_level (a)) ----
// The host's record_delta / for_each_record_bound, restated: how far one record can move its
// owner's potential between two of the owner's values (|sign(hit) - sign(miss)| |f| for a unary
// factor, 2 |f| (arity - 1) beyond; 0 for a fixed weight), and the Gershgorin share
// kappa * d * S of every record (S = the sum of the variable's deltas, boolean; the largest sum of
// one value row, categorical).  Same doubles in the same order as the host (-ffp-contract=off), then
// integers: the tables are the host's bit for bit whatever the order of the atomic adds.
__device__ __forceinline__ double unary_sign(uint32_t func, bool sat) {
  switch (func) {
    case FUNC_AND: case FUNC_ISTRUE: case FUNC_OR: case FUNC_IMPLY_NATURAL: return sat ? 1.0 : -1.0;
    case FUNC_EQUAL: return 1.0;
    default: return sat ? 1.0 : 0.0;
  }
}
__device__ __forceinline__ double record_delta(const EdgeRec r, const double *fval64, uint32_t e, bool owner_is_cat) {
  if (r.packed & EDGE_FIXED_FLAG) return 0.0;
  if (r.packed & EDGE_PRESIGNED) {
    float miss;
    __builtin_memcpy(&miss, &r.aux, 4);
    return fabs((double)r.fval - (double)miss);
  }
  const double f = (r.packed & EDGE_F64_FLAG) ? fval64[e] : (double)r.fval;
  const uint32_t ar = (r.packed & EDGE_INLINE2) ? 2u : ((r.packed >> EDGE_ARITY_SHIFT) & EDGE_ARITY_MASK);
  if (ar <= 1u) {
    const uint32_t fn = r.packed & EDGE_FUNC_MASK;
    const double hit = unary_sign(fn, owner_is_cat || r.aux == 1u);
    const double miss = unary_sign(fn, !owner_is_cat && r.aux == 0u);
    return fabs(hit - miss) * fabs(f);
  }
  return 2.0 * fabs(f) * (double)(ar - 1u);
}

// One workgroup per tile, a lane per variable; group_of[tile] = the row pair of the table the tile's
// variables add into (0xFFFFFFFF: none).  table[group][T[W] | h[W]], zeroed by the caller.
__global__ void __launch_bounds__(BLOCK_THREADS)
static_tables_kernel(const TileDesc *tiles, uint32_t n_tiles, const uint32_t *group_of, const uint32_t *v_meta,
                     const uint32_t *v_row, const uint32_t *row_ptr, const EdgeRec *edges, const double *fval64,
                     const uint8_t *w_fixed, uint32_t W, uint32_t learn_non_evidence, uint32_t noise_aware,
                     long long *table) {
  const uint32_t ti = blockIdx.x;
  if (ti >= n_tiles) return;
  const uint32_t k = group_of[ti];
  const TileDesc td = tiles[ti];
  if (k == 0xFFFFFFFFu || threadIdx.x >= td.nv) return;
  const uint32_t p = td.v0 + threadIdx.x, m = v_meta[p];
  const bool trig = learn_non_evidence || (!noise_aware && (m & VM_EVIDENCE)) || (noise_aware && (m & VM_TRUTHINESS));
  if (!trig) return;
  const bool cat = m & VM_CATEGORICAL;
  long long *row = table + (size_t)k * 2 * W, *hrow = row + W;
  const uint32_t r0 = v_row[p], r1 = v_row[p + 1];
  double S = 0.0;
  for (uint32_t r = r0; r < r1; ++r) {
    double sr = 0.0;
    for (uint32_t e = row_ptr[r]; e < row_ptr[r + 1]; ++e) sr += record_delta(edges[e], fval64, e, cat);
    S = cat ? (S > sr ? S : sr) : S + sr;
  }
  if (S != 0.0) {
    const double kappa = cat ? 0.5 : 0.25;
    for (uint32_t e = row_ptr[r0]; e < row_ptr[r1]; ++e) {
      const EdgeRec rec = edges[e];
      const double d = record_delta(rec, fval64, e, cat);
      if (d != 0.0) atomicAdd((unsigned long long *)&hrow[rec.wid], (unsigned long long)llrint(H_SCALE * (kappa * d * S)));
    }
  }
  if (cat) return;      // (their update counts depend on the samples: dynamic)
  const long long one = (long long)FIX_SCALE;
  for (uint32_t e = row_ptr[r0]; e < row_ptr[r0 + 1]; ++e) {
    const uint32_t wid = edges[e].wid;
    if (!w_fixed[wid]) atomicAdd((unsigned long long *)&row[wid], (unsigned long long)one);
  }
}

// max over the table's T entries and over its h entries -> out[0], out[1] (non-negative integers)
__global__ void __launch_bounds__(256)
table_max_kernel(const long long *table, uint64_t n_groups, uint32_t W, unsigned long long *out) {
  const uint64_t n = n_groups * 2 * W, stride = (uint64_t)gridDim.x * blockDim.x;
  unsigned long long mt = 0, mh = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const unsigned long long v = (unsigned long long)table[i];
    if ((i / W) & 1ull) mh = v > mh ? v : mh; else mt = v > mt ? v : mt;
  }
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned long long ot = __shfl_xor(mt, off, 64), oh = __shfl_xor(mh, off, 64);
    mt = ot > mt ? ot : mt; mh = oh > mh ? oh : mh;
  }
  if ((threadIdx.x & 63u) == 0) { atomicMax(&out[0], mt); atomicMax(&out[1], mh); }
}
}  // namespace

bool available() { return true; }

void build_static_tables(const TileDesc *d_tiles, uint32_t n_tiles, const uint32_t *h_group_of, const uint32_t *d_v_meta,
                         const uint32_t *d_v_row, const uint32_t *d_row_ptr, const EdgeRec *d_edges, const double *d_fval64,
                         const uint8_t *d_w_fixed, uint32_t W, uint32_t n_groups, bool learn_non_evidence, bool noise_aware,
                         long long *d_table, long long *t_max, long long *h_max, void *stream_v) {
  hipStream_t st = (hipStream_t)stream_v;
  *t_max = 0; *h_max = 0;
  if (!n_tiles || !n_groups || !W) return;
  uint32_t *d_group = nullptr;
  unsigned long long *d_max = nullptr;
  try {
    DEVB_HIP(hipMalloc(&d_group, (size_t)n_tiles * 4));
    DEVB_HIP(hipMalloc(&d_max, 16));
    DEVB_HIP(hipMemcpyAsync(d_group, h_group_of, (size_t)n_tiles * 4, hipMemcpyHostToDevice, st));
    DEVB_HIP(hipMemsetAsync(d_max, 0, 16, st));
    DEVB_HIP(hipMemsetAsync(d_table, 0, (size_t)n_groups * 2 * W * 8, st));
    hipLaunchKernelGGL(static_tables_kernel, dim3(n_tiles), dim3(BLOCK_THREADS), 0, st, d_tiles, n_tiles,
                       (const uint32_t *)d_group, d_v_meta, d_v_row, d_row_ptr, d_edges, d_fval64, d_w_fixed, W,
                       (uint32_t)learn_non_evidence, (uint32_t)noise_aware, d_table);
    DEVB_HIP(hipGetLastError());
    const uint64_t n = (uint64_t)n_groups * 2 * W;
    const unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, 256u * 16u);
    hipLaunchKernelGGL(table_max_kernel, dim3(grid), dim3(256), 0, st, (const long long *)d_table, (uint64_t)n_groups, W, d_max);
    DEVB_HIP(hipGetLastError());
    unsigned long long mx[2] = {0, 0};
    DEVB_HIP(hipMemcpyAsync(mx, d_max, 16, hipMemcpyDeviceToHost, st));
    DEVB_HIP(hipStreamSynchronize(st));
    *t_max = (long long)mx[0]; *h_max = (long long)mx[1];
  } catch (...) {
    (void)hipFree(d_group); (void)hipFree(d_max);
    throw;
  }
  (void)hipFree(d_group); (void)hipFree(d_max);
}

void build_sorted_records(const TileDesc *d_tiles, const EdgeRec *d_edges, const EdgeRec8 *d_edges8,
                          const SuperTile *d_supers, uint32_t n_supers, const uint32_t *h_dbits, uint32_t n_dbits,
                          uint64_t n_total, SortRec8 *d_out, void *stream_v) {
  hipStream_t st = (hipStream_t)stream_v;
  if (!n_supers || !n_total) return;
  if (n_dbits > SORT_MAX_DVALS) throw std::runtime_error("device build: too many distinct record deltas");
  unsigned long long *k0 = nullptr, *k1 = nullptr;
  uint32_t *v0 = nullptr, *v1 = nullptr, *d_dbits = nullptr;
  void *tmp = nullptr;
  auto cleanup = [&]() {
    (void)hipFree(k0); (void)hipFree(k1); (void)hipFree(v0); (void)hipFree(v1); (void)hipFree(d_dbits); (void)hipFree(tmp);
  };
  try {
    DEVB_HIP(hipMalloc(&k0, n_total * 8)); DEVB_HIP(hipMalloc(&k1, n_total * 8));
    DEVB_HIP(hipMalloc(&v0, n_total * 4)); DEVB_HIP(hipMalloc(&v1, n_total * 4));
    DEVB_HIP(hipMalloc(&d_dbits, std::max<size_t>(4, (size_t)n_dbits * 4)));
    if (n_dbits) DEVB_HIP(hipMemcpyAsync(d_dbits, h_dbits, (size_t)n_dbits * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(emit_sorted_kernel, dim3(n_supers), dim3(EMIT_THREADS), 0, st, d_tiles, d_edges, d_edges8, d_supers,
                       n_supers, (const uint32_t *)d_dbits, n_dbits, k0, v0);
    DEVB_HIP(hipGetLastError());
    // stable radix sort by (super-tile, weight id): only the bits that can be set
    uint32_t sbits = 0;
    while ((1ull << sbits) < (uint64_t)n_supers) ++sbits;
    const unsigned end_bit = 32u + (sbits ? sbits : 1u);
    size_t tmp_bytes = 0;
    DEVB_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, k0, k1, v0, v1, (size_t)n_total, 0u, end_bit, st));
    DEVB_HIP(hipMalloc(&tmp, std::max<size_t>(16, tmp_bytes)));
    DEVB_HIP(rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, v0, v1, (size_t)n_total, 0u, end_bit, st));
    const unsigned grid = (unsigned)std::min<uint64_t>((n_total + 255) / 256, 256u * 32u);
    hipLaunchKernelGGL(compose_sorted_kernel, dim3(grid), dim3(256), 0, st, (const unsigned long long *)k1,
                       (const uint32_t *)v1, n_total, d_out);
    DEVB_HIP(hipGetLastError());
    DEVB_HIP(hipStreamSynchronize(st));      // (the scratch arrays die here)
  } catch (...) {
    cleanup();
    throw;
  }
  cleanup();
}

}  // namespace devb
}  // namespace dwx
