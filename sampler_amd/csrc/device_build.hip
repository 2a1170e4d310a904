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
//   batch_curvature        the curvature estimate of one mini-batch: three power steps on its Hessian bound
//                          (dwx_api.cc: row_sum_bound)
//   build_incidence        a plan level's pull-gradient incidence list, sorted by (chunk, weight), and its
//                          block-pull tables (dwx_api.cc: build_level's two counting sorts + table fill)
//
// Sorting itself is rocPRIM's radix sort (a plain library sort, as the guide allows for plain
// library operations); everything around it is written here.  Replaces nothing of the reference
// (its construct_index sorts per variable with std::sort, src/factor_graph.cc:90-199).
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "device_build.h"

namespace dwx {
namespace devb {

namespace {
inline void check(hipError_t e, const char *what) {
  if (e != hipSuccess) throw std::runtime_error(std::string("HIP error in ") + what + ": " + hipGetErrorString(e));
}
#define DEVB_HIP(x) check((x), #x)

// The builds' large temporaries (sort keys / values / rocPRIM storage: 16 GB for config 5's incidence list) come
// from a cache of device blocks that are handed back, not freed: on this stack hipFree of a large block returns
// at once and the release is paid by a LATER hipMalloc -- about a second per 16 GB (tools: 4 x 4 GB allocated in
// 1 ms, or in 1.08 s after two rounds of free), which made `dw gibbs` on config 5's size vary by seconds from
// run to run.  A block serves any later request it is large enough for (smallest fit); release_scratch()
// really frees the idle ones.  Small requests (< 1 MiB) go to hipMalloc / hipFree as before.
}  // namespace
void release_scratch(int device, bool only_if_tight);
namespace {
constexpr size_t SCRATCH_MIN = (size_t)1 << 20;
struct ScratchBlock { void *p; size_t bytes; int device; bool busy; };
std::mutex g_scratch_mu;
std::vector<ScratchBlock> g_scratch;

// a persistent output of a build: plain hipMalloc, the idle cache blocks given back once if that fails
hipError_t output_malloc(void **p, size_t bytes) {
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) release_scratch(dev, false);
    e = hipMalloc(p, bytes);
  }
  return e;
}

void *scratch_take(size_t bytes) {
  bytes = std::max<size_t>(16, bytes);
  int dev = 0;
  DEVB_HIP(hipGetDevice(&dev));
  if (bytes >= SCRATCH_MIN) {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    ScratchBlock *best = nullptr;
    for (ScratchBlock &b : g_scratch)
      if (!b.busy && b.device == dev && b.bytes >= bytes && (!best || b.bytes < best->bytes)) best = &b;
    if (best) { best->busy = true; return best->p; }
  }
  void *p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) {
    // out of device memory with idle blocks in the cache that are all too small: give them back, once
    (void)hipGetLastError();
    release_scratch(dev, false);
    DEVB_HIP(hipMalloc(&p, bytes));
  }
  if (bytes >= SCRATCH_MIN) {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    g_scratch.push_back({p, bytes, dev, true});
  }
  return p;
}

void scratch_give(void *p) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    for (ScratchBlock &b : g_scratch)
      if (b.p == p) { b.busy = false; return; }
  }
  (void)hipFree(p);
}

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
// ---- the pull gradient's incidence list and block-pull tables of a plan level (build_level (b)) ----
// tile_info[tile]: group | mode << 30 (mode 1: every record of the tile's triggering variables, 2: only
// the pre-signed ones -- TILE_PULL_UNARY), 0xFFFFFFFF: the tile owns no entry.  An entry = (SGD-triggering
// boolean variable, non-fixed record of its row with dd = hit - miss != 0): key = group << 32 | weight
// id, value = slot << 32 | f32 bits of dd, slot = tile * 256 + lane.  Emitted in (tile, lane, record)
// order at the tile's offset (COUNT: only counted), so that the stable sort by key leaves every
// (group, weight)'s entries in tile order -- what the block tables need.
template <bool COUNT>
__global__ void __launch_bounds__(BLOCK_THREADS)
incidence_emit_kernel(const TileDesc *tiles, uint32_t n_tiles, const uint32_t *tile_info, const uint32_t *v_meta,
                      const uint32_t *v_row, const uint32_t *row_ptr, const EdgeRec *edges, uint32_t learn_non_evidence,
                      uint32_t noise_aware, unsigned long long *tile_count, const unsigned long long *tile_base,
                      unsigned long long *keys, unsigned long long *vals) {
  __shared__ uint32_t s_n[BLOCK_THREADS];
  const uint32_t ti = blockIdx.x, l = threadIdx.x;
  if (ti >= n_tiles) return;
  const uint32_t info = tile_info[ti];
  if (info == 0xFFFFFFFFu) { if (COUNT && l == 0) tile_count[ti] = 0; return; }
  const uint32_t group = info & 0x3FFFFFFFu;
  const bool unary_only = (info >> 30) == 2u;
  const TileDesc td = tiles[ti];
  uint32_t e0 = 0, e1 = 0;
  if (l < td.nv) {
    const uint32_t p = td.v0 + l, m = v_meta[p];
    if (learn_non_evidence || (!noise_aware && (m & VM_EVIDENCE))) { e0 = row_ptr[v_row[p]]; e1 = row_ptr[v_row[p] + 1]; }
  }
  auto wanted = [&](const EdgeRec r, float &dd) {
    if (r.packed & EDGE_FIXED_FLAG) return false;
    if (unary_only && !(r.packed & EDGE_PRESIGNED)) return false;
    float miss;
    __builtin_memcpy(&miss, &r.aux, 4);
    dd = r.fval - miss;      // exact: |hit| == |miss| or one of them is 0
    return dd != 0.0f;
  };
  uint32_t n = 0;
  for (uint32_t e = e0; e < e1; ++e) { float dd; n += wanted(edges[e], dd) ? 1u : 0u; }
  s_n[l] = n;
  __syncthreads();
  for (uint32_t off = 1; off < BLOCK_THREADS; off <<= 1) {      // inclusive scan
    const uint32_t v = l >= off ? s_n[l - off] : 0u;
    __syncthreads();
    s_n[l] += v;
    __syncthreads();
  }
  if (COUNT) { if (l == BLOCK_THREADS - 1) tile_count[ti] = s_n[l]; return; }
  unsigned long long at = tile_base[ti] + (s_n[l] - n);
  for (uint32_t e = e0; e < e1; ++e) {
    const EdgeRec r = edges[e];
    float dd;
    if (!wanted(r, dd)) continue;
    uint32_t bits;
    __builtin_memcpy(&bits, &dd, 4);
    keys[at] = ((unsigned long long)group << 32) | r.wid;
    vals[at] = ((unsigned long long)(ti * BLOCK_THREADS + l) << 32) | bits;
    ++at;
  }
}

// first index i with keys[i] >= q[j], for a few queries
__global__ void lower_bound_kernel(const unsigned long long *keys, uint64_t n, const unsigned long long *q, uint32_t nq,
                                   unsigned long long *out) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nq) return;
  uint64_t a = 0, b = n;
  while (a < b) { const uint64_t mid = (a + b) >> 1; if (keys[mid] < q[j]) a = mid + 1; else b = mid; }
  out[j] = a;
}

// which tiles own entries; the distinct dd bit patterns (an open-addressing set, 0xFFFFFFFF = empty)
constexpr uint32_t DSET_SLOTS = 1u << 16;
__global__ void __launch_bounds__(256)
incidence_scan_kernel(const unsigned long long *vals, uint64_t n, uint8_t *has, uint32_t *dset, uint32_t *dset_count) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  uint32_t last = 0xFFFFFFFFu;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const unsigned long long v = vals[i];
    uint8_t *h = &has[(uint32_t)(v >> 32) / BLOCK_THREADS];
    if (!*h) *h = 1;
    const uint32_t bits = (uint32_t)v;
    if (bits == last) continue;
    last = bits;
    uint32_t slot = (bits * 2654435761u) >> 16;
    for (uint32_t probe = 0; probe < DSET_SLOTS; ++probe, slot = (slot + 1) & (DSET_SLOTS - 1)) {
      const uint32_t cur = atomicCAS(&dset[slot], 0xFFFFFFFFu, bits);
      if (cur == 0xFFFFFFFFu) { atomicAdd(dset_count, 1u); break; }
      if (cur == bits) break;
    }
  }
}

// where each weight's entries start inside group `g` (entries [g0, g1) of the sorted arrays): w_at[w], w in [0, W]
__global__ void __launch_bounds__(256)
weight_start_kernel(const unsigned long long *keys, uint64_t g0, uint64_t g1, uint32_t group, uint32_t W, uint32_t *w_at) {
  const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w > W) return;
  const unsigned long long q = ((unsigned long long)group << 32) | w;
  uint64_t a = g0, b = g1;
  if (w == W) a = g1;
  else while (a < b) { const uint64_t mid = (a + b) >> 1; if (keys[mid] < q) a = mid + 1; else b = mid; }
  w_at[w] = (uint32_t)(a - g0);
}

// One lane per weight walks the weight's entries of the group (tile order: blocks ascending): the first
// `cap` entries of every (variable block, weight) go into the block table's row, the rest is counted
// (FILL) or copied, in walk order, to the group's kept list (!FILL; ovs = exclusive scan of the counts).
template <bool FILL>
__global__ void __launch_bounds__(256)
ell_walk_kernel(const unsigned long long *vals, uint64_t g0, const uint32_t *w_at, uint32_t W, const uint32_t *block_of,
                const uint32_t *tile0, uint32_t depth, uint64_t Wp, const uint32_t *dvals, uint32_t n_dvals, U32x4 *ell,
                uint32_t *ov, const uint32_t *ovs, const unsigned long long *keys, unsigned long long *kept_keys,
                unsigned long long *kept_vals, uint64_t kept_base) {
  const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= W) return;
  const uint32_t cap = BP_ROW * depth;
  uint32_t cur = 0xFFFFFFFFu, kk = 0, o = 0;
  const uint64_t out0 = FILL ? 0 : kept_base + ovs[w];
  for (uint64_t i = g0 + w_at[w]; i < g0 + w_at[w + 1]; ++i) {
    const unsigned long long v = vals[i];
    const uint32_t slot = (uint32_t)(v >> 32), ti = slot / BLOCK_THREADS, vb = block_of[ti];
    if (vb != cur) { cur = vb; kk = 0; }
    if (kk < cap) {
      if (FILL) {
        const uint32_t bits = (uint32_t)v;
        uint32_t a = 0, b = n_dvals;
        while (a < b) { const uint32_t mid = (a + b) >> 1; if (dvals[mid] < bits) a = mid + 1; else b = mid; }
        ell[((uint64_t)vb * depth + kk / BP_ROW) * Wp + w].v[kk % BP_ROW] =
            ((ti - tile0[vb]) * BLOCK_THREADS + slot % BLOCK_THREADS) | (a << BP_SLOT_BITS);
      }
    } else {
      if (!FILL) { kept_keys[out0 + o] = keys[i]; kept_vals[out0 + o] = v; }
      ++o;
    }
    ++kk;
  }
  if (FILL) ov[w] = o;
}

// the list columns of one group, padded to whole runs with neutral entries (the last entry's weight and slot, dd = 0)
__global__ void __launch_bounds__(256)
incidence_columns_kernel(const unsigned long long *keys, const unsigned long long *vals, uint64_t src0, uint64_t n,
                         uint64_t dst0, uint64_t padded, uint32_t *iw, uint32_t *is, float *id) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < padded; i += stride) {
    const uint64_t j = src0 + (i < n ? i : n - 1);
    const unsigned long long v = vals[j];
    iw[dst0 + i] = (uint32_t)keys[j];
    is[dst0 + i] = (uint32_t)(v >> 32);
    const uint32_t bits = i < n ? (uint32_t)v : 0u;
    float d;
    __builtin_memcpy(&d, &bits, 4);
    id[dst0 + i] = d;
  }
}
// ---- curvature estimate of one mini-batch (dwx_api.cc: row_sum_bound, the dense branch) ----
// Three power steps y = H x on the batch's curvature bound H = sum over its SGD-triggering variables of
// kappa d d^T, from x = 1: per variable dot = sum d x[wid], then y[wid] += kappa d dot -- as 64-bit
// FIXED-POINT atomic adds (scale = 2^62 / (R U): R the batch's records, U the largest possible term),
// so that the sums, and with them the plan decisions taken on lambda, do not depend on the order.
// The host's arithmetic, term for term; it spends 5 s per batch count on config 5's 10^9 records
// (two dependent random reads per record and step), the device ~50 ms.
__device__ __forceinline__ bool triggers_sgd(uint32_t m, uint32_t lne, uint32_t na) {
  return lne || (!na && (m & VM_EVIDENCE)) || (na && (m & VM_TRUTHINESS));
}
__global__ void __launch_bounds__(256)
curv_bounds_kernel(uint32_t p0, uint32_t p1, const uint32_t *v_meta, const uint32_t *v_row, const uint32_t *row_ptr,
                   const EdgeRec *edges, const double *fval64, uint32_t lne, uint32_t na, unsigned long long *out) {
  const uint32_t p = p0 + blockIdx.x * blockDim.x + threadIdx.x;
  double u = 0.0, dm = 0.0;
  unsigned long long n = 0;
  if (p < p1) {
    const uint32_t m = v_meta[p];
    if (triggers_sgd(m, lne, na)) {
      const bool cat = m & VM_CATEGORICAL;
      const uint32_t e0 = row_ptr[v_row[p]], e1 = row_ptr[v_row[p + 1]];
      double S = 0.0, dv = 0.0;
      for (uint32_t k = e0; k < e1; ++k) { const double d = record_delta(edges[k], fval64, k, cat); S += d; dv = d > dv ? d : dv; }
      u = (cat ? 0.5 : 0.25) * dv * S;
      dm = (cat ? 0.5 : 0.25) * dv * dv;
      n = e1 - e0;
    }
  }
  // (non-negative doubles order like their bit patterns)
  unsigned long long ub, db;
  __builtin_memcpy(&ub, &u, 8); __builtin_memcpy(&db, &dm, 8);
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned long long ou = __shfl_xor(ub, off, 64), od = __shfl_xor(db, off, 64), on = __shfl_xor(n, off, 64);
    ub = ou > ub ? ou : ub; db = od > db ? od : db; n += on;
  }
  if ((threadIdx.x & 63u) == 0) { atomicMax(&out[0], ub); atomicMax(&out[1], db); atomicAdd(&out[2], n); }
}
template <bool FIRST>
__global__ void __launch_bounds__(256)
curv_step_kernel(uint32_t p0, uint32_t p1, const uint32_t *v_meta, const uint32_t *v_row, const uint32_t *row_ptr,
                 const EdgeRec *edges, const double *fval64, uint32_t lne, uint32_t na, const double *x, double scale,
                 double dscale, long long *yfix, long long *dfix) {
  const uint32_t p = p0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= p1) return;
  const uint32_t m = v_meta[p];
  if (!triggers_sgd(m, lne, na)) return;
  const bool cat = m & VM_CATEGORICAL;
  const double kappa = cat ? 0.5 : 0.25;
  const uint32_t e0 = row_ptr[v_row[p]], e1 = row_ptr[v_row[p + 1]];
  double dot = 0.0;
  for (uint32_t k = e0; k < e1; ++k) { const EdgeRec r = edges[k]; dot += record_delta(r, fval64, k, cat) * x[r.wid]; }
  for (uint32_t k = e0; k < e1; ++k) {
    const EdgeRec r = edges[k];
    const double d = record_delta(r, fval64, k, cat);
    if (d == 0.0) continue;
    atomicAdd((unsigned long long *)&yfix[r.wid], (unsigned long long)llrint(scale * (kappa * d * dot)));
    if (FIRST) atomicAdd((unsigned long long *)&dfix[r.wid], (unsigned long long)llrint(dscale * (kappa * d * d)));
  }
}
constexpr uint32_t CURV_BLOCKS = 1024;
// y = yfix / scale; per-block partial sums of x.y, x.x, y.y (added up on the host in block order)
__global__ void __launch_bounds__(256)
curv_reduce_kernel(const long long *yfix, const double *x, uint32_t W, double scale, double *y, double *part) {
  __shared__ double s_r[3][256];
  double xy = 0.0, xx = 0.0, yy = 0.0;
  for (uint32_t w = blockIdx.x * 256u + threadIdx.x; w < W; w += CURV_BLOCKS * 256u) {
    const double yw = (double)yfix[w] / scale, xw = x[w];
    y[w] = yw;
    xy += xw * yw; xx += xw * xw; yy += yw * yw;
  }
  s_r[0][threadIdx.x] = xy; s_r[1][threadIdx.x] = xx; s_r[2][threadIdx.x] = yy;
  __syncthreads();
  for (uint32_t half = 128; half >= 1; half >>= 1) {
    if (threadIdx.x < half)
      for (int k = 0; k < 3; ++k) s_r[k][threadIdx.x] += s_r[k][threadIdx.x + half];
    __syncthreads();
  }
  if (threadIdx.x == 0) for (int k = 0; k < 3; ++k) part[3 * blockIdx.x + k] = s_r[k][0];
}
__global__ void __launch_bounds__(256)
curv_next_kernel(const double *y, double norm, uint32_t W, double *x, long long *yfix) {
  for (uint32_t w = blockIdx.x * 256u + threadIdx.x; w < W; w += CURV_BLOCKS * 256u) { x[w] = y[w] * norm; yfix[w] = 0; }
}
__global__ void __launch_bounds__(256)
curv_fill_kernel(uint32_t W, double *x, long long *yfix, long long *dfix) {
  for (uint32_t w = blockIdx.x * 256u + threadIdx.x; w < W; w += CURV_BLOCKS * 256u) { x[w] = 1.0; yfix[w] = 0; dfix[w] = 0; }
}
__global__ void __launch_bounds__(256)
curv_dmax_kernel(const long long *dfix, uint32_t W, unsigned long long *out) {
  unsigned long long mx = 0;
  for (uint32_t w = blockIdx.x * 256u + threadIdx.x; w < W; w += CURV_BLOCKS * 256u) { const unsigned long long v = (unsigned long long)dfix[w]; mx = v > mx ? v : mx; }
  for (int off = 32; off >= 1; off >>= 1) { const unsigned long long o = __shfl_xor(mx, off, 64); mx = o > mx ? o : mx; }
  if ((threadIdx.x & 63u) == 0) atomicMax(out, mx);
}
}  // namespace

bool available() { return true; }

CurvatureScratch::~CurvatureScratch() {
  (void)hipFree(x); (void)hipFree(y); (void)hipFree(yfix); (void)hipFree(dfix); (void)hipFree(part); (void)hipFree(small);
}

double batch_curvature(uint32_t p0, uint32_t p1, const uint32_t *d_v_meta, const uint32_t *d_v_row, const uint32_t *d_row_ptr,
                       const EdgeRec *d_edges, const double *d_fval64, bool learn_non_evidence, bool noise_aware, uint32_t W,
                       CurvatureScratch &sc, void *stream_v) {
  hipStream_t st = (hipStream_t)stream_v;
  if (p1 <= p0 || !W) return 0.0;
  if (!sc.x) {
    DEVB_HIP(hipMalloc(&sc.x, (size_t)W * 8)); DEVB_HIP(hipMalloc(&sc.y, (size_t)W * 8));
    DEVB_HIP(hipMalloc(&sc.yfix, (size_t)W * 8)); DEVB_HIP(hipMalloc(&sc.dfix, (size_t)W * 8));
    DEVB_HIP(hipMalloc(&sc.part, (size_t)CURV_BLOCKS * 3 * 8)); DEVB_HIP(hipMalloc(&sc.small, 64));
  }
  const uint32_t lne = learn_non_evidence, na = noise_aware;
  const unsigned vgrid = (p1 - p0 + 255) / 256;
  unsigned long long *small = (unsigned long long *)sc.small;
  DEVB_HIP(hipMemsetAsync(small, 0, 64, st));
  hipLaunchKernelGGL(curv_bounds_kernel, dim3(vgrid), dim3(256), 0, st, p0, p1, d_v_meta, d_v_row, d_row_ptr, d_edges, d_fval64, lne, na, small);
  unsigned long long h[4] = {0, 0, 0, 0};
  DEVB_HIP(hipMemcpyAsync(h, small, 32, hipMemcpyDeviceToHost, st));
  DEVB_HIP(hipStreamSynchronize(st));
  double U, D2;
  std::memcpy(&U, &h[0], 8); std::memcpy(&D2, &h[1], 8);
  const uint64_t R = h[2];
  if (!(U > 0.0) || !R) return 0.0;
  const double scale = std::ldexp(1.0, 62) / ((double)R * U), dscale = std::ldexp(1.0, 62) / ((double)R * D2);
  double *x = (double *)sc.x, *y = (double *)sc.y, *part = (double *)sc.part;
  long long *yfix = (long long *)sc.yfix, *dfix = (long long *)sc.dfix;
  hipLaunchKernelGGL(curv_fill_kernel, dim3(CURV_BLOCKS), dim3(256), 0, st, W, x, yfix, dfix);
  double lam = 0.0;
  std::vector<double> hp(CURV_BLOCKS * 3);
  for (int iter = 0; iter < 3; ++iter) {
    if (iter == 0)
      hipLaunchKernelGGL(curv_step_kernel<true>, dim3(vgrid), dim3(256), 0, st, p0, p1, d_v_meta, d_v_row, d_row_ptr, d_edges, d_fval64,
                         lne, na, (const double *)x, scale, dscale, yfix, dfix);
    else
      hipLaunchKernelGGL(curv_step_kernel<false>, dim3(vgrid), dim3(256), 0, st, p0, p1, d_v_meta, d_v_row, d_row_ptr, d_edges, d_fval64,
                         lne, na, (const double *)x, scale, dscale, yfix, dfix);
    hipLaunchKernelGGL(curv_reduce_kernel, dim3(CURV_BLOCKS), dim3(256), 0, st, (const long long *)yfix, (const double *)x, W, scale, y, part);
    DEVB_HIP(hipGetLastError());
    DEVB_HIP(hipMemcpyAsync(hp.data(), part, hp.size() * 8, hipMemcpyDeviceToHost, st));
    DEVB_HIP(hipStreamSynchronize(st));
    double xy = 0.0, xx = 0.0, yy = 0.0;
    for (uint32_t b = 0; b < CURV_BLOCKS; ++b) { xy += hp[3 * b]; xx += hp[3 * b + 1]; yy += hp[3 * b + 2]; }
    if (xx > 0) lam = std::max(lam, xy / xx);
    const double norm = yy > 0 ? 1.0 / std::sqrt(yy) : 0.0;
    hipLaunchKernelGGL(curv_next_kernel, dim3(CURV_BLOCKS), dim3(256), 0, st, (const double *)y, norm, W, x, yfix);
  }
  DEVB_HIP(hipMemsetAsync(small, 0, 8, st));
  hipLaunchKernelGGL(curv_dmax_kernel, dim3(CURV_BLOCKS), dim3(256), 0, st, (const long long *)dfix, W, small);
  DEVB_HIP(hipGetLastError());
  DEVB_HIP(hipMemcpyAsync(h, small, 8, hipMemcpyDeviceToHost, st));
  DEVB_HIP(hipStreamSynchronize(st));
  const double dmax = (double)(long long)h[0] / dscale;
  return std::max(lam, dmax);
}

void build_incidence(const TileDesc *d_tiles, const TileDesc *h_tiles, uint32_t n_tiles, const uint32_t *h_tile_info,
                     const uint32_t *d_v_meta, const uint32_t *d_v_row, const uint32_t *d_row_ptr, const EdgeRec *d_edges,
                     bool learn_non_evidence, bool noise_aware, uint32_t W, uint32_t n_groups, uint64_t block_pull_min_w,
                     uint32_t bp_tiles, Incidence &out, void *stream_v) {
  hipStream_t st = (hipStream_t)stream_v;
  out = Incidence();
  out.inc_begin.assign(n_groups, 0); out.inc_end.assign(n_groups, 0);
  std::vector<void *> scratch;
  auto dalloc = [&](size_t bytes) { void *p = scratch_take(bytes); scratch.push_back(p); return p; };
  auto release = [&]() { for (void *p : scratch) scratch_give(p); scratch.clear(); };
  U32x4 *ell_now = nullptr;
  // DWX_TIMING=1: wall time of the steps on stderr (each ends on a stream sync)
  const bool timing = getenv("DWX_TIMING") != nullptr;
  auto t_step = std::chrono::steady_clock::now();
  auto step = [&](const char *what) {
    if (!timing) return;
    (void)hipStreamSynchronize(st);
    const auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[devb incidence] %-26s %.3f s\n", what, std::chrono::duration<double>(t - t_step).count());
    t_step = t;
  };
  try {
    // ---- entries: count per tile, scan on the host (a few hundred thousand tiles), emit, sort ----
    uint32_t *d_info = (uint32_t *)dalloc((size_t)n_tiles * 4);
    unsigned long long *d_cnt = (unsigned long long *)dalloc((size_t)n_tiles * 8);
    DEVB_HIP(hipMemcpyAsync(d_info, h_tile_info, (size_t)n_tiles * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(incidence_emit_kernel<true>, dim3(n_tiles), dim3(BLOCK_THREADS), 0, st, d_tiles, n_tiles,
                       (const uint32_t *)d_info, d_v_meta, d_v_row, d_row_ptr, d_edges, (uint32_t)learn_non_evidence,
                       (uint32_t)noise_aware, d_cnt, (const unsigned long long *)nullptr, (unsigned long long *)nullptr,
                       (unsigned long long *)nullptr);
    DEVB_HIP(hipGetLastError());
    std::vector<unsigned long long> base(n_tiles + 1, 0);
    DEVB_HIP(hipMemcpyAsync(base.data() + 1, d_cnt, (size_t)n_tiles * 8, hipMemcpyDeviceToHost, st));
    DEVB_HIP(hipStreamSynchronize(st));
    for (uint32_t t = 0; t < n_tiles; ++t) base[t + 1] += base[t];
    const uint64_t n = base[n_tiles];
    out.n_entries = n;
    if (n + (uint64_t)n_groups * PULL_RUN >= 0xFFFFFFFFull) throw std::invalid_argument("incidence list exceeds 2^32-1 entries");
    if (!n) { release(); return; }
    DEVB_HIP(hipMemcpyAsync(d_cnt, base.data(), (size_t)n_tiles * 8, hipMemcpyHostToDevice, st));
    step("count");
    unsigned long long *k0 = (unsigned long long *)dalloc(n * 8), *k1 = (unsigned long long *)dalloc(n * 8);
    unsigned long long *v0 = (unsigned long long *)dalloc(n * 8), *v1 = (unsigned long long *)dalloc(n * 8);
    step("allocate 4 x n x 8 bytes");
    hipLaunchKernelGGL(incidence_emit_kernel<false>, dim3(n_tiles), dim3(BLOCK_THREADS), 0, st, d_tiles, n_tiles,
                       (const uint32_t *)d_info, d_v_meta, d_v_row, d_row_ptr, d_edges, (uint32_t)learn_non_evidence,
                       (uint32_t)noise_aware, (unsigned long long *)nullptr, (const unsigned long long *)d_cnt, k0, v0);
    DEVB_HIP(hipGetLastError());
    step("emit");
    uint32_t gbits = 0;
    while ((1ull << gbits) < (uint64_t)n_groups) ++gbits;
    {
      size_t tmp_bytes = 0;
      DEVB_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, k0, k1, v0, v1, (size_t)n, 0u, 32u + (gbits ? gbits : 1u), st));
      void *tmp = dalloc(tmp_bytes);
      DEVB_HIP(rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, v0, v1, (size_t)n, 0u, 32u + (gbits ? gbits : 1u), st));
    }
    step("radix sort");
    const unsigned long long *keys = k1, *vals = v1;      // (k0 / v0 are free again: the kept lists go there)
    // ---- where the groups start; which tiles own entries; the distinct deltas ----
    std::vector<unsigned long long> q(n_groups + 1), c_start(n_groups + 1);
    for (uint32_t k = 0; k <= n_groups; ++k) q[k] = (unsigned long long)k << 32;
    unsigned long long *d_q = (unsigned long long *)dalloc((n_groups + 1) * 8), *d_cs = (unsigned long long *)dalloc((n_groups + 1) * 8);
    DEVB_HIP(hipMemcpyAsync(d_q, q.data(), (n_groups + 1) * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(lower_bound_kernel, dim3((n_groups + 256) / 256), dim3(256), 0, st, keys, n, (const unsigned long long *)d_q,
                       n_groups + 1, d_cs);
    uint8_t *d_has = (uint8_t *)dalloc(n_tiles);
    uint32_t *d_dset = (uint32_t *)dalloc(DSET_SLOTS * 4 + 16), *d_dcount = d_dset + DSET_SLOTS;
    DEVB_HIP(hipMemsetAsync(d_has, 0, n_tiles, st));
    DEVB_HIP(hipMemsetAsync(d_dset, 0xFF, DSET_SLOTS * 4, st));
    DEVB_HIP(hipMemsetAsync(d_dcount, 0, 4, st));
    hipLaunchKernelGGL(incidence_scan_kernel, dim3(256 * 8), dim3(256), 0, st, vals, n, d_has, d_dset, d_dcount);
    DEVB_HIP(hipGetLastError());
    std::vector<uint8_t> has(n_tiles);
    std::vector<uint32_t> dset(DSET_SLOTS);
    uint32_t dcount = 0;
    DEVB_HIP(hipMemcpyAsync(c_start.data(), d_cs, (n_groups + 1) * 8, hipMemcpyDeviceToHost, st));
    DEVB_HIP(hipMemcpyAsync(has.data(), d_has, n_tiles, hipMemcpyDeviceToHost, st));
    DEVB_HIP(hipMemcpyAsync(dset.data(), d_dset, DSET_SLOTS * 4, hipMemcpyDeviceToHost, st));
    DEVB_HIP(hipMemcpyAsync(&dcount, d_dcount, 4, hipMemcpyDeviceToHost, st));
    DEVB_HIP(hipStreamSynchronize(st));
    step("group starts, owners, deltas");
    // ---- block pull: per group, runs of <= bp_tiles consecutive tiles started at tiles that own entries ----
    std::vector<uint32_t> dvals;
    bool enabled = W >= block_pull_min_w && dcount <= BP_MAX_DELTAS;
    if (enabled) {
      for (uint32_t b : dset) if (b != 0xFFFFFFFFu) dvals.push_back(b);
      std::sort(dvals.begin(), dvals.end());
    }
    const uint64_t Wp = ((uint64_t)W + BP_THREADS - 1) / BP_THREADS * BP_THREADS;
    std::vector<uint64_t> kept_n(n_groups);
    for (uint32_t k = 0; k < n_groups; ++k) kept_n[k] = c_start[k + 1] - c_start[k];
    uint64_t max_blocks = 0;
    unsigned long long *kk = k0, *kv = v0;      // kept lists (groups at their ORIGINAL starts: never longer)
    if (enabled) {
      std::vector<uint32_t> block_of(n_tiles, 0), tile0_all;
      std::vector<uint64_t> tile0_off(n_groups + 1, 0);
      std::vector<uint32_t> depth(n_groups, 0);
      for (uint32_t k = 0; k < n_groups; ++k) {
        tile0_off[k] = tile0_all.size();
        if (!kept_n[k]) continue;
        const size_t first = tile0_all.size();
        for (uint32_t ti = 0; ti < n_tiles; ++ti) {
          if (!has[ti] || (h_tile_info[ti] & 0x3FFFFFFFu) != k || h_tile_info[ti] == 0xFFFFFFFFu) continue;
          if (tile0_all.size() == first || ti >= tile0_all.back() + bp_tiles) tile0_all.push_back(ti);
          block_of[ti] = (uint32_t)(tile0_all.size() - first) - 1;
        }
        const uint64_t nvb = tile0_all.size() - first;
        const double lambda = (double)kept_n[k] / ((double)W * (double)nvb);
        if (lambda < 0.5) { tile0_all.resize(first); continue; }      // nearly empty rows: this group keeps its list
        depth[k] = lambda > 3.2 ? 2u : 1u;
      }
      tile0_off[n_groups] = tile0_all.size();
      uint32_t *d_block_of = (uint32_t *)dalloc((size_t)n_tiles * 4), *d_tile0 = (uint32_t *)dalloc(tile0_all.size() * 4 + 4);
      uint32_t *d_dvals = (uint32_t *)dalloc(dvals.size() * 4 + 4);
      uint32_t *d_wat = (uint32_t *)dalloc(((size_t)W + 1) * 4), *d_ov = (uint32_t *)dalloc((size_t)W * 4 + 4), *d_ovs = (uint32_t *)dalloc((size_t)W * 4 + 4);
      DEVB_HIP(hipMemcpyAsync(d_block_of, block_of.data(), (size_t)n_tiles * 4, hipMemcpyHostToDevice, st));
      if (!tile0_all.empty()) DEVB_HIP(hipMemcpyAsync(d_tile0, tile0_all.data(), tile0_all.size() * 4, hipMemcpyHostToDevice, st));
      if (!dvals.empty()) DEVB_HIP(hipMemcpyAsync(d_dvals, dvals.data(), dvals.size() * 4, hipMemcpyHostToDevice, st));
      size_t scan_bytes = 0;
      DEVB_HIP(rocprim::exclusive_scan(nullptr, scan_bytes, d_ov, d_ovs, 0u, (size_t)W, rocprim::plus<uint32_t>(), st));
      void *scan_tmp = dalloc(scan_bytes);
      out.bp.resize(n_groups);
      const unsigned wgrid = (W + 256) / 256;
      for (uint32_t k = 0; k < n_groups; ++k) {
        const uint64_t nvb = tile0_off[k + 1] - tile0_off[k];
        if (!depth[k] || !nvb) continue;
        const uint64_t g0 = c_start[k], g1 = c_start[k + 1];
        hipLaunchKernelGGL(weight_start_kernel, dim3(wgrid), dim3(256), 0, st, keys, g0, g1, k, W, d_wat);
        DEVB_HIP(output_malloc((void **)&ell_now, nvb * depth[k] * Wp * sizeof(U32x4)));
        DEVB_HIP(hipMemsetAsync(ell_now, 0xFF, nvb * depth[k] * Wp * sizeof(U32x4), st));
        hipLaunchKernelGGL(ell_walk_kernel<true>, dim3(wgrid), dim3(256), 0, st, vals, g0, (const uint32_t *)d_wat, W,
                           (const uint32_t *)d_block_of, (const uint32_t *)(d_tile0 + tile0_off[k]), depth[k], Wp,
                           (const uint32_t *)d_dvals, (uint32_t)dvals.size(), ell_now, d_ov, (const uint32_t *)nullptr, keys,
                           (unsigned long long *)nullptr, (unsigned long long *)nullptr, (uint64_t)0);
        DEVB_HIP(hipGetLastError());
        DEVB_HIP(rocprim::exclusive_scan(scan_tmp, scan_bytes, d_ov, d_ovs, 0u, (size_t)W, rocprim::plus<uint32_t>(), st));
        uint32_t last_ov = 0, last_ovs = 0;
        DEVB_HIP(hipMemcpyAsync(&last_ov, d_ov + (W - 1), 4, hipMemcpyDeviceToHost, st));
        DEVB_HIP(hipMemcpyAsync(&last_ovs, d_ovs + (W - 1), 4, hipMemcpyDeviceToHost, st));
        DEVB_HIP(hipStreamSynchronize(st));
        kept_n[k] = (uint64_t)last_ov + last_ovs;
        if (kept_n[k])
          hipLaunchKernelGGL(ell_walk_kernel<false>, dim3(wgrid), dim3(256), 0, st, vals, g0, (const uint32_t *)d_wat, W,
                             (const uint32_t *)d_block_of, (const uint32_t *)(d_tile0 + tile0_off[k]), depth[k], Wp,
                             (const uint32_t *)d_dvals, (uint32_t)dvals.size(), (U32x4 *)nullptr, (uint32_t *)nullptr,
                             (const uint32_t *)d_ovs, keys, kk, kv, g0);
        DEVB_HIP(hipGetLastError());
        Incidence::BlockTable &bt = out.bp[k];
        bt.d_ell = ell_now; ell_now = nullptr;
        bt.tile0.assign(tile0_all.begin() + tile0_off[k], tile0_all.begin() + tile0_off[k + 1]);
        bt.blocks = (uint32_t)nvb; bt.depth = depth[k];
        bt.total = g1 - g0; bt.on_list = kept_n[k];
        max_blocks = std::max(max_blocks, nvb);
      }
      if (!max_blocks) out.bp.clear();
    }
    step("block tables");
    // groups without a table keep all their entries: the kept arrays hold them at the same positions
    for (uint32_t k = 0; k < n_groups; ++k) {
      const uint64_t g0 = c_start[k], gn = c_start[k + 1] - c_start[k];
      const bool tabled = max_blocks && !out.bp.empty() && out.bp[k].blocks;
      if (tabled || !gn) continue;
      DEVB_HIP(hipMemcpyAsync(kk + g0, keys + g0, gn * 8, hipMemcpyDeviceToDevice, st));
      DEVB_HIP(hipMemcpyAsync(kv + g0, vals + g0, gn * 8, hipMemcpyDeviceToDevice, st));
    }
    // ---- the list columns, every group padded to whole runs ----
    std::vector<uint64_t> off(n_groups + 1, 0);
    for (uint32_t k = 0; k < n_groups; ++k) off[k + 1] = off[k] + (kept_n[k] + PULL_RUN - 1) / PULL_RUN * PULL_RUN;
    const uint64_t padded = off[n_groups];
    if (padded) {
      DEVB_HIP(output_malloc((void **)&out.d_inc_wid, padded * 4)); DEVB_HIP(output_malloc((void **)&out.d_inc_slot, padded * 4));
      DEVB_HIP(output_malloc((void **)&out.d_inc_d, padded * 4));
      for (uint32_t k = 0; k < n_groups; ++k) {
        out.inc_begin[k] = (uint32_t)off[k]; out.inc_end[k] = (uint32_t)off[k + 1];
        const uint64_t pn = off[k + 1] - off[k];
        if (!pn) continue;
        const unsigned grid = (unsigned)std::min<uint64_t>((pn + 255) / 256, 256u * 16u);
        hipLaunchKernelGGL(incidence_columns_kernel, dim3(grid), dim3(256), 0, st, (const unsigned long long *)kk,
                           (const unsigned long long *)kv, c_start[k], kept_n[k], off[k], pn, out.d_inc_wid, out.d_inc_slot, out.d_inc_d);
      }
      DEVB_HIP(hipGetLastError());
    }
    out.dvals = dvals;
    out.max_blocks = max_blocks;
    out.wp = Wp;
    DEVB_HIP(hipStreamSynchronize(st));
    step("list columns");
  } catch (...) {
    release();
    (void)hipFree(ell_now);
    for (auto &bt : out.bp) (void)hipFree(bt.d_ell);
    (void)hipFree(out.d_inc_wid); (void)hipFree(out.d_inc_slot); (void)hipFree(out.d_inc_d);
    out = Incidence();
    throw;
  }
  release();
  step("release scratch");
}

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
    scratch_give(k0); scratch_give(k1); scratch_give(v0); scratch_give(v1); scratch_give(d_dbits); scratch_give(tmp);
  };
  try {
    k0 = (unsigned long long *)scratch_take(n_total * 8); k1 = (unsigned long long *)scratch_take(n_total * 8);
    v0 = (uint32_t *)scratch_take(n_total * 4); v1 = (uint32_t *)scratch_take(n_total * 4);
    d_dbits = (uint32_t *)scratch_take(std::max<size_t>(4, (size_t)n_dbits * 4));
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
    tmp = scratch_take(tmp_bytes);
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

uint64_t scratch_bytes(int device);
void release_scratch(int device, bool only_if_tight) {
  if (only_if_tight) {
    // (kept for the plan levels built later unless it is a quarter or more of what is still free)
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && scratch_bytes(device) < free_b / 4) return;
  }
  std::vector<void *> dead;
  {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    for (size_t i = 0; i < g_scratch.size();) {
      if (!g_scratch[i].busy && (device < 0 || g_scratch[i].device == device)) {
        dead.push_back(g_scratch[i].p);
        g_scratch[i] = g_scratch.back();
        g_scratch.pop_back();
      } else {
        ++i;
      }
    }
  }
  for (void *p : dead) (void)hipFree(p);
}

uint64_t scratch_bytes(int device) {
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  uint64_t n = 0;
  for (const ScratchBlock &b : g_scratch)
    if (device < 0 || b.device == device) n += b.bytes;
  return n;
}

}  // namespace devb
}  // namespace dwx
