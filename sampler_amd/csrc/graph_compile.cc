// graph_compile.cc -- see graph_compile.h.  Host-only C++17, no HIP.
#include "graph_compile.h"

#include "host_parallel.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <functional>
#include <mutex>
#include <numeric>
#include <stdexcept>
#include <thread>

namespace dwx {
namespace {

struct LimitError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

constexpr uint32_t kUnset = 0xFFFFFFFFu;
constexpr uint64_t kInvalid64 = ~0ull;

inline bool is_linear_zero(double x) {  // src/common.h:114-116
  return x <= LINEAR_ZERO_THRESHOLD && x >= -LINEAR_ZERO_THRESHOLD;
}

inline bool known_func(uint32_t f) {
  switch (f) {
    case FUNC_IMPLY_NATURAL: case FUNC_OR: case FUNC_AND: case FUNC_EQUAL: case FUNC_ISTRUE:
    case FUNC_LINEAR: case FUNC_RATIO: case FUNC_LOGICAL: case FUNC_AND_CATEGORICAL:
    case FUNC_IMPLY_MLN:
      return true;
  }
  return false;
}

}  // namespace

uint64_t CompiledGraph::device_bytes() const {
  uint64_t b = 0;
  b += 4 * (V * 4 + 1);                       // v_meta, v_orig, v_init, v_row
  b += 4 * (R + 1) + (row_truth.empty() ? 0 : 8 * R);
  b += 16 * NIdx + 8 * edges8.size() + 8 * n_sorted + 32 * supers.size() + (edge_fval64.empty() ? 0 : 8 * NIdx) + 8 * NVif;
  b += 32 * tiles.size();
  b += 4 * V * 2 + 4 * R;                     // assignments x2, tallies
  b += W * (8 + 4 + 1 + 16 + 8);              // weights f64 + f32 copy, fixed, grad G/T, T static
  return b;
}

bool sorted_eligible(const TileDesc &td) {
  return (td.flags & TILE_SIMPLE) && !(td.flags & (TILE_CATEGORICAL | TILE_OUTSIDE));
}
float sorted_rec_d(const EdgeRec8 &c) {
  const int sh = (int)((c.key >> REC8_HIT_SHIFT) & 3u) - 1, sm = (int)((c.key >> REC8_MISS_SHIFT) & 3u) - 1;
  return (float)(sh - sm) * c.f;     // exact: a factor in {-2 .. 2}
}

// Super-tiles over the eligible tiles of `ranges` (tile ranges of one launch each) and their
// weight-sorted records.  One workgroup of sorted_sweep_kernel takes one super-tile, `slots`
// workgroups are resident and start in the order of their super-tiles.  full_rounds (the default
// layout of a graph): a run of eligible tiles is cut into full rounds of `slots` super-tiles of
// per_super tiles -- the bigger a super-tile, the denser its sorted gathers -- and ONE round of
// `slots` small ones for the rest, last (610 equal super-tiles on 512 slots would run as two rounds
// with the second a fifth full; cut into 1024 equal ones they are half as dense: config 3's learning
// sweep 0.36 -> 0.40 ms).  Otherwise (the layout of a split plan, one range per mini-batch chunk):
// `slots` equal super-tiles per run, so that every chunk's launch is as wide as the chip.
void build_sorted_layout(const CompiledGraph &g, const std::vector<std::pair<uint32_t, uint32_t>> &ranges,
                         uint32_t per_super, uint32_t slots, bool full_rounds, uint32_t nth, SortedLayout &out,
                         bool plan_only) {
  out.supers.clear(); out.recs.clear(); out.n = 0;
  if (g.edges8.size() == 0 || g.sort_dvals.empty()) return;
  per_super = std::max(1u, std::min(per_super, SORT_TV_SLOTS - 1));
  slots = std::max(1u, slots);
  auto emit = [&](uint32_t &i, uint32_t end, uint32_t per) {
    SuperTile st{};
    st.tile0 = i; st.v0 = g.tiles[i].v0;
    uint32_t nv = 0, j = i;
    while (j < end && j - i < per && nv + g.tiles[j].nv <= SUPER_NV_MAX) nv += g.tiles[j++].nv;
    st.ntiles = j - i; st.nv = nv;
    out.supers.push_back(st);
    i = j;
  };
  auto cut_run = [&](uint32_t a, uint32_t b) {           // eligible tiles [a, b)
    uint32_t i = a;
    if (full_rounds)
      while (b - i >= per_super * slots)
        for (uint32_t k = 0; k < slots; ++k) emit(i, b, per_super);
    if (i == b) return;
    const uint32_t per = std::min(per_super, std::max((b - i + slots - 1) / slots, std::min(8u, per_super)));
    while (i < b) emit(i, b, per);
  };
  for (const auto &rg : ranges) {
    uint32_t run0 = 0;
    bool open = false;
    for (uint32_t i = rg.first; i <= rg.second; ++i) {
      const bool stop = i == rg.second || !sorted_eligible(g.tiles[i]);
      if (stop && open) { cut_run(run0, i); open = false; }
      if (i < rg.second && sorted_eligible(g.tiles[i]) && !open) { open = true; run0 = i; }
    }
  }
  const size_t ns = out.supers.size();
  if (!ns) return;
  std::vector<uint64_t> count(ns + 1, 0);
  parallel_ranges(ns, nth, [&](uint64_t sb, uint64_t se) {
    for (uint64_t si = sb; si < se; ++si) {
      const SuperTile &st = out.supers[si];
      const TileDesc &t0 = g.tiles[st.tile0], &t1 = g.tiles[st.tile0 + st.ntiles - 1];
      uint64_t n = 0;
      for (uint64_t e = t0.e0; e < (uint64_t)t1.e0 + t1.nedges; ++e) n += sorted_rec_d(g.edges8[e]) != 0.0f;
      count[si + 1] = n;
    }
  }, 1);
  for (size_t i = 0; i < ns; ++i) count[i + 1] += count[i];
  out.n = count[ns];
  if (plan_only) {
    // (the records themselves are built on the device: device_build.hip)
    for (size_t si = 0; si < ns; ++si) {
      SuperTile &st = out.supers[si];
      st.lo = (uint32_t)count[si]; st.hi = (uint32_t)(count[si] >> 32); st.nrec = (uint32_t)(count[si + 1] - count[si]);
    }
    return;
  }
  // the records, super-tile by super-tile, sorted by (weight id, owner)
  out.recs.reset(out.n + 1);
  out.recs[out.n] = SortRec8{0u, 0u};
  const std::vector<uint32_t> &dbits = g.sort_dbits;
  parallel_ranges(ns, nth, [&](uint64_t sb, uint64_t se) {
    std::vector<uint64_t> keys, tmp;
    for (uint64_t si = sb; si < se; ++si) {
      SuperTile &st = out.supers[si];
      keys.clear();
      for (uint32_t ti = st.tile0; ti < st.tile0 + st.ntiles; ++ti) {
        const TileDesc &td = g.tiles[ti];
        for (uint32_t l = 0; l < td.nv; ++l) {
          const uint32_t p = td.v0 + l, slot = p - st.v0;
          for (uint32_t e = g.row_ptr[g.v_row[p]]; e < g.row_ptr[g.v_row[p + 1]]; ++e) {
            const EdgeRec8 &c = g.edges8[e];
            const float dv = sorted_rec_d(c);
            if (dv == 0.0f) continue;
            uint32_t bits; std::memcpy(&bits, &dv, 4);
            const uint32_t di = 1u + (uint32_t)(std::lower_bound(dbits.begin(), dbits.end(), bits) - dbits.begin());
            keys.push_back(((uint64_t)(c.key & REC8_WID_MASK) << 32) | (di << SORT_OWNER_BITS) | slot);
          }
        }
      }
      // by (weight id, owner): a stable LSD radix sort on the weight id (the keys come in
      // (owner, row) order, so ties end up ascending in the owner's slot)
      {
        const size_t n = keys.size();
        tmp.resize(n);
        uint32_t wmax = 0;
        for (uint64_t k : keys) wmax = std::max(wmax, (uint32_t)(k >> 32));
        uint64_t *src = keys.data(), *dst = tmp.data();
        for (uint32_t shift = 32; shift < 64 && (wmax >> (shift - 32)) != 0; shift += 11) {
          uint32_t cnt[2049] = {0};
          for (size_t i = 0; i < n; ++i) ++cnt[((src[i] >> shift) & 2047u) + 1];
          for (uint32_t b = 0; b < 2048; ++b) cnt[b + 1] += cnt[b];
          for (size_t i = 0; i < n; ++i) dst[cnt[(src[i] >> shift) & 2047u]++] = src[i];
          std::swap(src, dst);
        }
        if (src != keys.data()) std::memcpy(keys.data(), src, n * sizeof(uint64_t));
      }
      const uint64_t at = count[si];
      st.lo = (uint32_t)at; st.hi = (uint32_t)(at >> 32); st.nrec = (uint32_t)keys.size();
      for (size_t i = 0; i < keys.size(); ++i)
        out.recs[at + i] = SortRec8{(uint32_t)(keys[i] >> 32), (uint32_t)keys[i]};
    }
  }, 1);
}

void compile_graph(const dwx_graph_desc &d, const dwx_compile_opts &o, CompiledGraph &g,
                   bool *limit) {
  *limit = false;
  try {
    const uint64_t V = d.num_variables, F = d.num_factors, E = d.num_edges, W = d.num_weights;
    g.V = V; g.F = F; g.E = E; g.W = W;
    if (d.num_ghost_variables > V) throw std::runtime_error("num_ghost_variables > num_variables");
    const uint64_t Vo = V - d.num_ghost_variables;   // owned variables are ids [0, Vo)
    g.Vo = Vo;
    if (V >= kUnset || F >= kUnset || W >= kUnset || E >= kUnset)
      throw LimitError("graph exceeds the 32-bit compact layout (V, F, W, E must be < 2^32-1)");
    const uint32_t nth = host_threads(o.n_threads);
    g.tile_vars = o.tile_vars ? std::min(o.tile_vars, BLOCK_THREADS) : BLOCK_THREADS;
    const uint32_t arity_cap = o.conflict_arity_cap ? o.conflict_arity_cap : 256;
    if (F && d.fac_edge_offset[F] != E) throw std::runtime_error("fac_edge_offset[F] != num_edges");
    // DWX_TIMING=1: wall time of every compile phase on stderr
    const bool timing = getenv("DWX_TIMING") != nullptr;
    auto t_phase = std::chrono::steady_clock::now();
    auto phase = [&](const char *what) {
      if (!timing) return;
      auto t = std::chrono::steady_clock::now();
      fprintf(stderr, "[dwx compile] %-28s %.3f s\n", what, std::chrono::duration<double>(t - t_phase).count());
      t_phase = t;
    };

    // ---- variables (src/binary_format.cc:64-126) ----
    std::vector<uint8_t> is_cat(V);
    std::vector<uint32_t> card(V), assign_dense(V);
    g.var_is_evid.resize(V);
    {
      // (all host threads: at config 5's 10^8 variables the serial loops of this front end were 1.3 s)
      const uint32_t T = std::max(1u, nth);
      std::vector<uint32_t> part_max(T, 0);
      std::vector<uint8_t> part_cat(T, 0);
      parallel_parts(V, T, [&](uint32_t t, uint64_t vb, uint64_t ve) {
        uint32_t mx = 0;
        bool any = false;
        for (uint64_t v = vb; v < ve; ++v) {
          if (d.var_dtype[v] > 1)
            throw std::runtime_error("Only Boolean and Categorical variables are supported");
          is_cat[v] = d.var_dtype[v] == 1;
          if (d.var_cardinality[v] > MAX_CARD) throw LimitError("cardinality exceeds 2^24-1");
          card[v] = (uint32_t)d.var_cardinality[v];
          if (is_cat[v] && card[v] == 0) throw std::runtime_error("categorical variable with cardinality 0");
          g.var_is_evid[v] = d.var_role[v] >= 1;
          uint64_t init = g.var_is_evid[v] ? d.var_init_value[v] : 0;
          if (init >= kUnset && d.num_domains == 0) throw LimitError("initial value exceeds 32 bits");
          assign_dense[v] = (uint32_t)init;
          if (is_cat[v]) { any = true; mx = std::max(mx, card[v]); }
        }
        part_max[t] = mx; part_cat[t] = any;
      });
      for (uint32_t t = 0; t < T; ++t)
        if (part_cat[t]) { g.has_categorical = true; g.max_card = std::max(g.max_card, part_max[t]); }
    }

    phase("variables");
    // ---- domains (src/binary_format.cc:192-226): value -> index in file order ----
    // per block: (value, index) sorted by value, last index wins for duplicates
    std::vector<int64_t> dom_of(d.num_domains ? V : 0, -1);
    std::vector<uint64_t> ds_val;   // sorted values per block (same offsets as dom_offset)
    std::vector<uint32_t> ds_idx;
    std::vector<double> total_truth(d.num_domains ? V : 0, 0.0);   // (only domain blocks carry truthiness)
    if (d.num_domains) {
      ds_val.resize(d.dom_offset[d.num_domains]);
      ds_idx.resize(ds_val.size());
      for (uint64_t b = 0; b < d.num_domains; ++b) {
        uint64_t vid = d.dom_vid[b];
        if (vid >= V) throw std::runtime_error("domain block for unknown variable");
        uint64_t lo = d.dom_offset[b], hi = d.dom_offset[b + 1];
        if (!is_cat[vid]) throw std::runtime_error("domain block for a boolean variable");
        if (hi - lo != card[vid]) throw std::runtime_error("domain size != cardinality");
        std::vector<uint32_t> ord(hi - lo);
        std::iota(ord.begin(), ord.end(), 0u);
        std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t c) {
          return d.dom_value[lo + a] < d.dom_value[lo + c];
        });
        double tt = 0;
        for (uint64_t i = 0; i < ord.size(); ++i) {
          ds_val[lo + i] = d.dom_value[lo + ord[i]];
          ds_idx[lo + i] = ord[i];
          double tr = d.dom_truthiness[lo + ord[i]];
          if (!(tr >= 0 && tr <= 1)) throw std::runtime_error("truthiness outside [0,1]");
          bool last_of_run = (i + 1 == ord.size()) || d.dom_value[lo + ord[i + 1]] != ds_val[lo + i];
          if (last_of_run) tt += tr;   // unordered_map keeps one entry per value (the last)
        }
        total_truth[vid] = tt;
        dom_of[vid] = (int64_t)b;
      }
    }
    auto domain_index = [&](uint64_t vid, uint64_t val) -> uint64_t {  // src/variable.h:124-126
      if (dom_of.empty() || dom_of[vid] < 0) return val;
      uint64_t b = dom_of[vid], lo = d.dom_offset[b], hi = d.dom_offset[b + 1];
      // last element equal to val
      auto it = std::upper_bound(ds_val.begin() + lo, ds_val.begin() + hi, val);
      if (it == ds_val.begin() + lo || *(it - 1) != val)
        throw std::runtime_error("value " + std::to_string(val) + " not in the domain of variable " +
                                 std::to_string(vid));
      return ds_idx[(it - 1) - ds_val.begin()];
    };
    if (d.num_domains)
      for (uint64_t v = 0; v < V; ++v)
        if (dom_of[v] >= 0 && assign_dense[v]) {
          uint64_t init = g.var_is_evid[v] ? d.var_init_value[v] : 0;
          assign_dense[v] = (uint32_t)domain_index(v, init);
        }

    // ---- reference value numbering (src/factor_graph.cc:139-175) ----
    // (an exclusive prefix sum of the variables' row counts, by all threads: 10^8 variables on one were 0.15 s)
    g.ref_var_val_base.resize(V);
    auto rows_of = [&](uint64_t v) -> uint64_t { return v < Vo ? (is_cat[v] ? (uint64_t)card[v] : 1u) : 0u; };
    parallel_ranges(V, nth, [&](uint64_t b, uint64_t e) { for (uint64_t v = b; v < e; ++v) g.ref_var_val_base[v] = rows_of(v); });
    parallel_inclusive_prefix(g.ref_var_val_base.data(), V, nth);
    const uint64_t R = V ? g.ref_var_val_base[V - 1] : 0;
    parallel_ranges(V, nth, [&](uint64_t b, uint64_t e) { for (uint64_t v = b; v < e; ++v) g.ref_var_val_base[v] -= rows_of(v); });
    if (R >= kUnset) throw LimitError("number of value rows exceeds 2^32-1");
    g.R = R;
    g.value_sparse.assign(R, 0);
    std::vector<double> ref_truth;
    for (uint64_t v = 0; v < Vo && g.has_categorical; ++v) {
      if (!is_cat[v]) continue;
      uint64_t base = g.ref_var_val_base[v];
      if (!dom_of.empty() && dom_of[v] >= 0) {
        uint64_t b = dom_of[v], lo = d.dom_offset[b];
        if (ref_truth.empty()) ref_truth.assign(R, 0.0);
        for (uint32_t j = 0; j < card[v]; ++j) g.value_sparse[base + j] = kInvalid64;
        for (uint32_t i = 0; i < card[v]; ++i) {
          bool last_of_run = (i + 1 == card[v]) || ds_val[lo + i + 1] != ds_val[lo + i];
          if (!last_of_run) continue;
          uint32_t idx = ds_idx[lo + i];
          g.value_sparse[base + idx] = ds_val[lo + i];
          ref_truth[base + idx] = d.dom_truthiness[lo + idx];
        }
      } else {
        for (uint32_t j = 0; j < card[v]; ++j) g.value_sparse[base + j] = j;
      }
    }
    parallel_ranges(V, nth, [&](uint64_t vb, uint64_t ve) {
      for (uint64_t v = vb; v < ve; ++v) {
        if (!is_cat[v] && assign_dense[v] > 1)
          throw std::runtime_error("boolean variable " + std::to_string(v) + " has an initial value other than 0/1");
        if (is_cat[v] && g.var_is_evid[v] && assign_dense[v] >= card[v])
          throw std::runtime_error("evidence value of variable " + std::to_string(v) +
                                   " is outside its domain");
      }
    });
    for (double t : ref_truth) if (t != 0.0) { g.has_truthiness = true; break; }

    phase("domains, value numbering");
    // ---- factors: dense predicates + per-variable back-refs
    //      (src/binary_format.cc:128-190) ----
    RawArray<uint32_t> edge_dense(E);   // (every entry is written by the parallel pass below)
    constexpr double kMaxLearnFeature = 65536.0;
    std::atomic<bool> any_conflict{false};   // some factor ties two variables together (arity 2 .. cap)
    parallel_ranges(F, nth, [&](uint64_t fb, uint64_t fe) {
      bool conflict = false;
      for (uint64_t f = fb; f < fe; ++f) {
        uint64_t lo = d.fac_edge_offset[f], hi = d.fac_edge_offset[f + 1];
        if (hi < lo || hi > E) throw std::runtime_error("fac_edge_offset not monotone");
        if (hi - lo > MAX_ARITY) throw LimitError("factor arity exceeds 2^24-1");
        conflict = conflict || (hi - lo >= 2 && hi - lo <= arity_cap);
        if (!known_func(d.fac_func[f]))
          throw std::runtime_error("Unsupported FACTOR_FUNCTION_TYPE = " + std::to_string(d.fac_func[f]));
        if (d.fac_weight_id[f] >= W) throw std::runtime_error("factor references unknown weight");
        // the gradient of a learnable weight is summed in fixed point (2^-30, int64: exact,
        // order-independent, all-reducible): a feature value beyond 2^16 could overflow the sum
        // of a heavily tied weight, a non-finite one has no meaning at all
        {
          const double fv = d.fac_feature_value[f];
          if (!(fv == fv) || fv > 1.7e308 || fv < -1.7e308)
            throw std::runtime_error("feature value of factor " + std::to_string(f) + " is not finite");
          if (!d.w_is_fixed[d.fac_weight_id[f]] && (fv > kMaxLearnFeature || fv < -kMaxLearnFeature))
            throw LimitError("feature value of factor " + std::to_string(f) +
                             " exceeds 65536 on a learnable weight (fixed-point gradient range)");
        }
        for (uint64_t e = lo; e < hi; ++e) {
          const uint64_t vid = d.edge_vid[e];
          if (vid >= V) throw std::runtime_error("factor references unknown variable");
          const uint64_t dense = domain_index(vid, d.edge_equal_to[e]);
          if (dense >= kUnset) throw LimitError("predicate value exceeds 32 bits");
          edge_dense[e] = (uint32_t)dense;
        }
      }
      if (conflict) any_conflict.store(true, std::memory_order_relaxed);
    });
    // back-references grouped by variable, in factor order inside a variable (a stable
    // parallel counting sort; ghosts get none)
    struct BackRef { uint32_t vid, val, fid; };
    RawArray<BackRef> br;
    std::vector<uint64_t> start;
    parallel_group_by_key<BackRef>(
        F, nth, V, [](const BackRef &r) { return (uint64_t)r.vid; },
        [&](uint64_t fb, uint64_t fe, auto &&emit) {
          for (uint64_t f = fb; f < fe; ++f)
            for (uint64_t e = d.fac_edge_offset[f]; e < d.fac_edge_offset[f + 1]; ++e) {
              const uint64_t vid = d.edge_vid[e];
              if (vid >= Vo) continue;
              // booleans index under value 0
              emit(BackRef{(uint32_t)vid, is_cat[vid] ? edge_dense[e] : 0u, (uint32_t)f});
            }
        },
        br, start);
    phase("back-references");
    // ---- construct_index: sort by (value, fid), dedup (src/factor_graph.cc:177-197) ----
    std::vector<uint32_t> row_len(R, 0);
    std::vector<uint32_t> ucnt(V, 0);
    parallel_ranges(V, nth, [&](uint64_t vb, uint64_t ve) {
      auto less = [](const BackRef &a, const BackRef &c) { return a.val < c.val || (a.val == c.val && a.fid < c.fid); };
      for (uint64_t v = vb; v < ve; ++v) {
        uint64_t lo = start[v], hi = start[v + 1];
        if (lo == hi) continue;
        if (!std::is_sorted(br.begin() + lo, br.begin() + hi, less)) std::sort(br.begin() + lo, br.begin() + hi, less);
        const uint32_t lim = is_cat[v] ? card[v] : 1;
        uint64_t w = lo;
        for (uint64_t i = lo; i < hi; ++i) {
          if (i > lo && br[i].val == br[w - 1].val && br[i].fid == br[w - 1].fid) continue;
          if (br[i].val >= lim)
            throw std::runtime_error("predicate value outside the domain of variable " + std::to_string(v));
          br[w] = br[i]; ++w;
          ++row_len[g.ref_var_val_base[v] + br[i].val];
        }
        ucnt[v] = (uint32_t)(w - lo);
      }
    });
    g.ref_row_ptr.assign(R + 1, 0);
    for (uint64_t r = 0; r < R; ++r) g.ref_row_ptr[r + 1] = g.ref_row_ptr[r] + row_len[r];
    g.NIdx = g.ref_row_ptr[R];
    if (g.NIdx >= kUnset) throw LimitError("index entries exceed 2^32-1");
    g.ref_fidx.reset(g.NIdx);
    parallel_ranges(V, nth, [&](uint64_t vb, uint64_t ve) {
      for (uint64_t v = vb; v < ve; ++v) {
        uint64_t src = start[v], dst = g.ref_row_ptr[g.ref_var_val_base[v]];
        // the variable's unique (value, fid) pairs are sorted by value, so they are
        // exactly its rows back to back
        for (uint32_t i = 0; i < ucnt[v]; ++i) g.ref_fidx[dst + i] = br[src + i].fid;
      }
    });
    br.clear();
    phase("construct_index");
    // ---- chromatic partition: greedy colouring of the variable conflict graph
    //      (two variables conflict iff they share a factor of arity 2..cap) ----
    std::vector<uint32_t> color(V, kUnset);
    if (!any_conflict.load()) {
      // no factor of arity 2 .. cap: nothing conflicts, the greedy pass below would give every variable colour 0
      // after walking all of its index entries on ONE thread (0.7 s of config 5's 10^9 records)
      parallel_ranges(Vo, nth, [&](uint64_t b, uint64_t e) { std::fill(color.begin() + b, color.begin() + e, 0u); });
      g.n_colors = Vo ? 1 : 0;
    } else {
      std::vector<uint64_t> stamp;
      uint32_t ncol = 0;
      for (uint64_t v = 0; v < Vo; ++v) {
        uint64_t r0 = g.ref_var_val_base[v], r1 = r0 + (is_cat[v] ? card[v] : 1);
        for (uint64_t i = g.ref_row_ptr[r0]; i < g.ref_row_ptr[r1]; ++i) {
          uint32_t f = g.ref_fidx[i];
          uint64_t lo = d.fac_edge_offset[f], hi = d.fac_edge_offset[f + 1];
          if (hi - lo < 2 || hi - lo > arity_cap) continue;
          for (uint64_t e = lo; e < hi; ++e) {
            uint64_t u = d.edge_vid[e];
            if (u == v || u >= Vo || color[u] == kUnset) continue;   // ghosts: Hogwild across shards
            if (color[u] >= stamp.size()) stamp.resize(color[u] + 1, 0);
            stamp[color[u]] = v + 1;
          }
        }
        uint32_t c = 0;
        while (c < stamp.size() && stamp[c] == v + 1) ++c;
        color[v] = c;
        ncol = std::max(ncol, c + 1);
      }
      g.n_colors = Vo ? ncol : 0;
    }

    phase("colouring");
    // ---- device order: colour-major; inside a colour query variables before evidence
    //      variables (an inference sweep launches over the query part only, like the
    //      reference skips evidence, src/gibbs_sampler.h:157), booleans before
    //      categoricals, then id ----
    const uint32_t nkeys = std::max(1u, g.n_colors) * 4;
    auto key_of = [&](uint64_t v) { return color[v] * 4 + (g.var_is_evid[v] ? 2u : 0u) + is_cat[v]; };
    // (class sizes per thread part: the stable placement below runs over the same parts)
    const uint32_t order_T = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(nth, Vo / 65536 + 1));
    std::vector<std::vector<uint64_t>> part_hist(order_T, std::vector<uint64_t>(nkeys, 0));
    std::vector<uint64_t> part_query(order_T, 0);
    parallel_parts(Vo, order_T, [&](uint32_t t, uint64_t b, uint64_t e) {
      std::vector<uint64_t> &h = part_hist[t];
      uint64_t q = 0;
      for (uint64_t v = b; v < e; ++v) { ++h[key_of(v)]; q += !g.var_is_evid[v]; }
      part_query[t] = q;
    }, 0);
    std::vector<uint64_t> key_start(nkeys + 1, 0);
    for (uint32_t t = 0; t < order_T; ++t)
      for (uint32_t k = 0; k < nkeys; ++k) key_start[k + 1] += part_hist[t][k];
    for (uint32_t k = 0; k < nkeys; ++k) key_start[k + 1] += key_start[k];
    g.perm.reset(V); g.pos.reset(V);
    // All-unary graphs (no factor links two variables: the order inside a class is free, and the
    // only locality a sweep can have is in its weight gathers): inside a class, variables are
    // ordered by the weight id of their FIRST record.  The lanes that stage first records then
    // gather neighbouring weights -- one or two L2 lines per wave-instruction instead of one per
    // lane -- while every weight's records stay spread over the whole order but for that one
    // (the sweeps are bound by L2 requests, one per gathered weight).  Only for
    // lightly tied weights (<= 256 factors per weight on average): the order also puts a tenth of
    // every weight's factors into one stretch of the sweep, and a learning sweep that has to be cut
    // into mini-batches (heavily tied weights, section 3.5) wants every batch to see every weight's
    // factors evenly -- there the gathers hit few lines anyway.
    // (E == F alone would also hold for a mix of arity-0 and arity-2 factors: ask every factor)
    bool all_unary = E == F && F > 0;
    if (all_unary) {
      std::atomic<bool> ok{true};
      parallel_ranges(F, nth, [&](uint64_t fb, uint64_t fe) {
        for (uint64_t f = fb; f < fe && ok.load(std::memory_order_relaxed); ++f)
          if (d.fac_edge_offset[f + 1] - d.fac_edge_offset[f] != 1) ok.store(false, std::memory_order_relaxed);
      });
      all_unary = ok.load();
    }
    const bool by_weight = all_unary && W <= (1u << 24) && F / std::max<uint64_t>(W, 1) <= 256 && !o.no_weight_order;
    if (by_weight) {
      std::vector<uint32_t> first_wid(Vo, 0);
      parallel_ranges(Vo, nth, [&](uint64_t vb, uint64_t ve) {
        for (uint64_t v = vb; v < ve; ++v) {
          const uint64_t r0 = g.ref_row_ptr[g.ref_var_val_base[v]], r1 = g.ref_row_ptr[g.ref_var_val_base[v] + (is_cat[v] ? card[v] : 1)];
          first_wid[v] = r1 > r0 ? (uint32_t)d.fac_weight_id[g.ref_fidx[r0]] : 0u;
        }
      });
      // stable counting sort by first weight, class by class (an all-unary graph has one colour:
      // four classes) -- one counter array of W + 1 entries, not one per class (the
      // dense nkeys x (W + 1) table was 256 MB per colour group at 2^24 weights)
      std::vector<uint32_t> cnt(W + 2);
      for (uint32_t k = 0; k < nkeys; ++k) {
        if (key_start[k + 1] == key_start[k]) continue;
        std::fill(cnt.begin(), cnt.end(), 0u);
        for (uint64_t v = 0; v < Vo; ++v) if (key_of(v) == k) ++cnt[first_wid[v] + 1];
        for (size_t i = 1; i < cnt.size(); ++i) cnt[i] += cnt[i - 1];
        for (uint64_t v = 0; v < Vo; ++v) {
          if (key_of(v) != k) continue;
          const uint64_t p = key_start[k] + cnt[first_wid[v]]++;
          g.perm[p] = (uint32_t)v; g.pos[v] = (uint32_t)p;
          if (p != v) g.order_is_identity = false;
        }
      }
    } else {
      // stable counting sort by class, part by part: part t's variables of class k start where the earlier
      // parts' end (the serial loop over 10^8 variables was 0.15 s)
      for (uint32_t k = 0; k < nkeys; ++k) {
        uint64_t at = key_start[k];
        for (uint32_t t = 0; t < order_T; ++t) { const uint64_t n = part_hist[t][k]; part_hist[t][k] = at; at += n; }
      }
      std::atomic<bool> identity{true};
      parallel_parts(Vo, order_T, [&](uint32_t t, uint64_t b, uint64_t e) {
        std::vector<uint64_t> &cur = part_hist[t];
        bool same = true;
        for (uint64_t v = b; v < e; ++v) {
          const uint64_t p = cur[key_of(v)]++;
          g.perm[p] = (uint32_t)v; g.pos[v] = (uint32_t)p;
          same = same && p == v;
        }
        if (!same) identity.store(false, std::memory_order_relaxed);
      }, 0);
      if (!identity.load()) g.order_is_identity = false;
    }
    for (uint64_t v = Vo; v < V; ++v) { g.perm[v] = (uint32_t)v; g.pos[v] = (uint32_t)v; }  // ghosts last
    g.launch_off.clear();
    for (uint32_t c = 0; c < g.n_colors; ++c) g.launch_off.push_back(key_start[4 * c]);
    g.launch_off.push_back(Vo);
    g.n_query = 0;
    for (uint32_t t = 0; t < order_T; ++t) g.n_query += part_query[t];
    if (g.n_colors == 0) g.launch_off.assign(1, 0);

    phase("device order");
    // ---- device rows ----
    g.v_meta.reset(V); g.v_init.reset(V); g.v_row.reset(V + 1);
    g.v_row[0] = 0;
    parallel_ranges(V, nth, [&](uint64_t pb, uint64_t pe) {
      for (uint64_t p = pb; p < pe; ++p) {
        uint64_t v = g.perm[p];
        uint32_t m = (is_cat[v] ? VM_CATEGORICAL : 0) | (g.var_is_evid[v] ? VM_EVIDENCE : 0) |
                     (!total_truth.empty() && !is_linear_zero(total_truth[v]) ? VM_TRUTHINESS : 0) |
                     ((is_cat[v] ? card[v] : 2u) << VM_CARD_SHIFT);
        g.v_meta[p] = m;
        g.v_init[p] = assign_dense[v];
        g.v_row[p + 1] = v < Vo ? (is_cat[v] ? card[v] : 1) : 0;      // (row counts; prefix below)
      }
    });
    parallel_inclusive_prefix(g.v_row.data() + 1, V, nth);
    g.row_ptr.reset(R + 1);
    g.row_ptr[0] = 0;     // (every other entry is written below: each row belongs to an owned variable)
    if (g.has_truthiness) g.row_truth.assign(R, 0.0);
    parallel_ranges(Vo, nth, [&](uint64_t pb, uint64_t pe) {
      for (uint64_t p = pb; p < pe; ++p) {
        uint64_t v = g.perm[p], rb = g.ref_var_val_base[v];
        uint32_t nr = g.v_row[p + 1] - g.v_row[p];
        for (uint32_t j = 0; j < nr; ++j) {
          g.row_ptr[g.v_row[p] + j + 1] = row_len[rb + j];
          if (g.has_truthiness) g.row_truth[g.v_row[p] + j] = ref_truth[rb + j];
        }
      }
    });
    parallel_inclusive_prefix(g.row_ptr.data() + 1, R, nth);

    phase("device rows");
    // ---- vifs of factors with arity >= 2 ----
    // Factors of arity > VIF_PER_RECORD_ARITY keep one block of entries per factor in g.vifs,
    // shared by their records.  The entries of the small ones go to a scratch array first and
    // are laid out again per RECORD, in record order, once the tiles are known (below): the
    // edge-parallel staging then reads them as a stream beside the records instead of as one
    // random 16/24-byte gather per record.
    // (parallel prefix sums over the factors: per-part totals, then per-part fills)
    RawArray<uint32_t> vif_base(F);
    RawArray<VifRec> small_vifs;
    uint64_t nvif = 0, nsmall = 0;
    {
      const uint32_t T = std::max(1u, nth);
      std::vector<uint64_t> part(T + 1, 0), spart(T + 1, 0);
      parallel_parts(F, T, [&](uint32_t t, uint64_t fb, uint64_t fe) {
        uint64_t n = 0, m = 0;
        for (uint64_t f = fb; f < fe; ++f) {
          const uint64_t a = d.fac_edge_offset[f + 1] - d.fac_edge_offset[f];
          if (a > VIF_PER_RECORD_ARITY) n += a;
          else if (a >= 2) m += a;
        }
        part[t + 1] = n;
        spart[t + 1] = m;
      });
      for (uint32_t t = 0; t < T; ++t) { part[t + 1] += part[t]; spart[t + 1] += spart[t]; }
      nvif = part[T];
      nsmall = spart[T];
      if (nvif >= kUnset || nsmall >= kUnset) throw LimitError("vif entries exceed 2^32-1");
      parallel_parts(F, T, [&](uint32_t t, uint64_t fb, uint64_t fe) {
        uint64_t n = part[t], m = spart[t];
        for (uint64_t f = fb; f < fe; ++f) {
          const uint64_t a = d.fac_edge_offset[f + 1] - d.fac_edge_offset[f];
          if (a > VIF_PER_RECORD_ARITY) { vif_base[f] = (uint32_t)n; n += a; }
          else if (a >= 2) { vif_base[f] = (uint32_t)m; m += a; }
          else vif_base[f] = 0u;
        }
      });
    }
    g.NVif = nvif + nsmall;   // (the tile sizes below ask whether there are any; final count after the tiles)
    g.vifs.resize(nvif);
    small_vifs.reset(nsmall);
    parallel_ranges(F, nth, [&](uint64_t fb, uint64_t fe) {
      for (uint64_t f = fb; f < fe; ++f) {
        uint64_t lo = d.fac_edge_offset[f], hi = d.fac_edge_offset[f + 1];
        if (hi - lo < 2) continue;
        VifRec *dst = hi - lo > VIF_PER_RECORD_ARITY ? &g.vifs[vif_base[f]] : &small_vifs[vif_base[f]];
        for (uint64_t e = lo; e < hi; ++e) dst[e - lo] = VifRec{g.pos[d.edge_vid[e]], edge_dense[e]};
      }
    });

    phase("vifs");
    // ---- edge records, variable-major in device order ----
    g.edges.reset(g.NIdx);
    std::atomic<bool> need64{false};
    parallel_ranges(Vo, nth, [&](uint64_t pb, uint64_t pe) {
      for (uint64_t p = pb; p < pe; ++p) {
        uint64_t v = g.perm[p];
        uint64_t src = g.ref_row_ptr[g.ref_var_val_base[v]], dst = g.row_ptr[g.v_row[p]];
        uint64_t n = g.row_ptr[g.v_row[p + 1]] - dst;
        for (uint64_t i = 0; i < n; ++i) {
          uint32_t f = g.ref_fidx[src + i];
          uint64_t lo = d.fac_edge_offset[f], a = d.fac_edge_offset[f + 1] - lo;
          EdgeRec r;
          r.wid = (uint32_t)d.fac_weight_id[f];
          r.aux = a == 1 ? edge_dense[lo] : vif_base[f];
          r.packed = (uint32_t)d.fac_func[f] | ((uint32_t)a << EDGE_ARITY_SHIFT);
          double fv = d.fac_feature_value[f];
          r.fval = (float)fv;
          if (!((double)r.fval == fv)) { r.packed |= EDGE_F64_FLAG; need64 = true; }
          if (d.w_is_fixed[r.wid]) r.packed |= EDGE_FIXED_FLAG;
          if (a == 1 && !(r.packed & EDGE_F64_FLAG)) {
            // fold the unary sign function in (src/factor.h:112-299 at arity 1)
            auto usign = [&](bool sat) -> double {
              switch (d.fac_func[f]) {
                case FUNC_AND: case FUNC_ISTRUE: case FUNC_OR: case FUNC_IMPLY_NATURAL:
                  return sat ? 1.0 : -1.0;
                case FUNC_EQUAL: return 1.0;
                default: return sat ? 1.0 : 0.0;
              }
            };
            const uint32_t eq = edge_dense[lo];
            const double s_hit = is_cat[v] ? usign(true) : usign(eq == 1u);
            const double s_miss = is_cat[v] ? usign(false) : usign(eq == 0u);
            const float fa = (float)(s_hit * fv), fb = (float)(s_miss * fv);
            uint32_t bits;
            std::memcpy(&bits, &fb, 4);
            r.fval = fa; r.aux = bits; r.packed |= EDGE_PRESIGNED;
          }
          g.edges[dst + i] = r;
        }
      }
    });
    if (need64) {
      g.has_f64_fval = true;
      g.edge_fval64.resize(g.NIdx);
      parallel_ranges(Vo, nth, [&](uint64_t pb, uint64_t pe) {
        for (uint64_t p = pb; p < pe; ++p) {
          uint64_t v = g.perm[p];
          uint64_t src = g.ref_row_ptr[g.ref_var_val_base[v]], dst = g.row_ptr[g.v_row[p]];
          uint64_t n = g.row_ptr[g.v_row[p + 1]] - dst;
          for (uint64_t i = 0; i < n; ++i) g.edge_fval64[dst + i] = d.fac_feature_value[g.ref_fidx[src + i]];
        }
      });
    }

    phase("edge records");
    // ---- workgroup tiles: <= tile_vars variables of one type whose value rows and
    //      edge records fit the LDS budget; an oversized variable gets a tile alone ----
    // Default LDS budget per tile: 3072 records (48 KiB) for all-unary graphs; graphs with
    // wider factors stage 32-byte records when learning and keep many more registers live
    // per staged record, so they get 1536-record tiles (3 workgroups per CU either way).
    // Categorical graphs too: 1536 rows and records per tile (192 variables of 8 values) keep
    // three workgroups per CU resident with their row pointers and potential scratch --
    // config 4's inference sweep 0.39 -> 0.32 ms against 2048-row, 3072-record tiles.
    g.ecap = o.tile_edges ? std::min(o.tile_edges, MAX_ECAP)
                          : ((g.NVif > 0 || g.has_categorical) ? MAX_ECAP / 2 : MAX_ECAP);
    g.rcap = o.tile_rows ? o.tile_rows : (g.has_categorical ? 1536 : g.tile_vars);
    if (g.rcap < g.tile_vars && !g.has_categorical) g.rcap = g.tile_vars;
    g.tile_v.clear(); g.launch_tile.clear(); g.launch_query_tile_end.clear();
    // degree binning: a variable with more records than this leaves the lane-per-variable tiles
    // and is walked by a whole wave (TILE_WIDE); 0xFFFFFFFF switches the bin off
    g.wide_min = o.wide_min_records ? o.wide_min_records : WIDE_MIN_RECORDS_DEFAULT;
    std::vector<uint8_t> tile_wide;
    const uint64_t nl = g.launch_off.size() - 1;
    const uint32_t kTypeMask = VM_CATEGORICAL | VM_EVIDENCE;
    for (uint64_t l = 0; l < nl; ++l) {
      g.launch_tile.push_back((uint32_t)g.tile_v.size());
      uint64_t p = g.launch_off[l], pend = g.launch_off[l + 1];
      bool seen_evid = false;
      while (p < pend) {
        uint64_t t0 = p;
        uint32_t type = g.v_meta[p] & kTypeMask;
        if ((type & VM_EVIDENCE) && !seen_evid) {
          seen_evid = true;
          g.launch_query_tile_end.push_back((uint32_t)g.tile_v.size());
        }
        uint64_t rows = 0, edges = 0;
        bool wide = false;
        while (p < pend && p - t0 < g.tile_vars && (g.v_meta[p] & kTypeMask) == type) {
          uint64_t nr = g.v_row[p + 1] - g.v_row[p];
          uint64_t ne = g.row_ptr[g.v_row[p + 1]] - g.row_ptr[g.v_row[p]];
          const bool alone = ne > g.wide_min;      // mid- or high-degree: a tile of its own
          if (p > t0 && (alone || rows + nr > g.rcap || edges + ne > g.ecap)) break;
          rows += nr; edges += ne; ++p;
          if (rows > g.rcap || edges > g.ecap) break;   // oversized single variable
          if (alone) { wide = true; break; }
        }
        if (rows > g.rcap || edges > g.ecap) { ++g.n_giant_tiles; wide = false; }
        if (wide) ++g.n_wide_tiles;
        tile_wide.push_back(wide ? 1 : 0);
        g.tile_v.push_back((uint32_t)t0);
      }
      if (!seen_evid) g.launch_query_tile_end.push_back((uint32_t)g.tile_v.size());
    }
    g.launch_tile.push_back((uint32_t)g.tile_v.size());
    g.tile_v.push_back((uint32_t)Vo);
    g.tiles.resize(g.tile_v.size() - 1);
    std::atomic<uint32_t> n_terms2{0};
    parallel_ranges(g.tiles.size(), nth, [&](uint64_t ib, uint64_t ie) {
    for (uint64_t i = ib; i < ie; ++i) {
      const uint32_t v0 = g.tile_v[i], v1 = g.tile_v[i + 1];
      TileDesc t{};
      t.v0 = v0; t.nv = v1 - v0;
      t.r0 = g.v_row[v0]; t.nrows = g.v_row[v1] - g.v_row[v0];
      t.e0 = g.row_ptr[g.v_row[v0]]; t.nedges = g.row_ptr[g.v_row[v1]] - g.row_ptr[g.v_row[v0]];
      bool simple = true, terms2 = true, terms3 = true;
      for (uint32_t e = t.e0; e < t.e0 + t.nedges; ++e) {
        const uint32_t pk = g.edges[e].packed;
        const bool pre = (pk & EDGE_PRESIGNED) != 0;
        const uint32_t ar = (pk >> EDGE_ARITY_SHIFT) & EDGE_ARITY_MASK;
        simple = simple && pre;
        terms2 = terms2 && (pre || (ar == 2 && !(pk & EDGE_F64_FLAG)));
        // (the staged sign * feature value products must be f32-exact: every sign function is
        // integer-valued at arity <= 3 except RATIO, log2(3) with two unsatisfied body atoms)
        terms3 = terms3 && (pre || (!(pk & EDGE_F64_FLAG) &&
                                    (ar == 2 || (ar == 3 && (pk & EDGE_FUNC_MASK) != FUNC_RATIO))));
      }
      const bool cat = g.v_meta[v0] & VM_CATEGORICAL;
      // a categorical tile is staged edge-parallel with "the owner takes the value its own
      // predicate names" as every record's proposal: the owner must sit in the factor once
      bool cat_terms3 = cat && terms3 && !simple;
      for (uint32_t l = 0; l < t.nv && cat_terms3; ++l)
        for (uint32_t e = g.row_ptr[g.v_row[v0 + l]]; e < g.row_ptr[g.v_row[v0 + l + 1]] && cat_terms3; ++e) {
          const EdgeRec &r = g.edges[e];
          if (r.packed & EDGE_PRESIGNED) continue;
          const uint32_t ar = (r.packed >> EDGE_ARITY_SHIFT) & EDGE_ARITY_MASK;
          uint32_t mine = 0;
          for (uint32_t k = 0; k < ar; ++k) mine += small_vifs[r.aux + k].vid == v0 + l;
          cat_terms3 = mine == 1;
        }
      const bool giant = t.nrows > g.rcap || t.nedges > g.ecap;
      const uint32_t outside = giant ? TILE_GIANT : (tile_wide[i] ? TILE_WIDE : 0u);
      bool unit_rows = t.nrows == t.nedges && t.nrows > 0;
      for (uint32_t r = 0; r < t.nrows && unit_rows; ++r) unit_rows = g.row_ptr[t.r0 + r + 1] - g.row_ptr[t.r0 + r] == 1;
      t.flags = (simple ? TILE_SIMPLE : 0u) | (cat ? TILE_CATEGORICAL : 0u) | outside | (unit_rows ? TILE_UNIT_ROWS : 0u) |
                ((simple && !cat && W > LDS_AGG_MAX_W && !outside) ? TILE_PULL : 0u) |
                ((terms2 && !simple && !cat && t.nv <= 256 && !outside) ? TILE_TERMS2 : 0u) |
                ((terms3 && !simple && (cat ? cat_terms3 : !terms2) && t.nv <= 256 && !outside) ? TILE_TERMS3 : 0u);
      if ((t.flags & (TILE_TERMS2 | TILE_TERMS3)) && !cat && W > LDS_AGG_MAX_W && !o.no_pull_unary) t.flags |= TILE_PULL_UNARY;
      if (t.flags & (TILE_TERMS2 | TILE_TERMS3)) ++n_terms2;
      // arity-2 records carry their two vif entries themselves where the kernels that will
      // see the tile implement it (the K <= 6 builds; not the oversized-variable kernel) and
      // the predicates fit 7 bits
      if ((t.flags & TILE_TERMS2) && g.ecap <= 6 * BLOCK_THREADS) {
        bool ok = true;
        for (uint32_t e = t.e0; e < t.e0 + t.nedges && ok; ++e) {
          const EdgeRec &r = g.edges[e];
          if (r.packed & EDGE_PRESIGNED) continue;
          const VifRec *vp = &small_vifs[r.aux];
          ok = vp[0].equal_to <= INLINE2_PRED_MASK && vp[1].equal_to <= INLINE2_PRED_MASK;
        }
        if (ok) {
          t.flags |= TILE_INLINE2;
          for (uint32_t l = 0; l < t.nv; ++l)
            for (uint32_t e = g.row_ptr[g.v_row[v0 + l]]; e < g.row_ptr[g.v_row[v0 + l + 1]]; ++e) {
              EdgeRec &r = g.edges[e];
              if (r.packed & EDGE_PRESIGNED) continue;
              const VifRec a = small_vifs[r.aux], b = small_vifs[r.aux + 1];
              const uint32_t me = v0 + l;
              const bool a_me = a.vid == me, b_me = b.vid == me;
              r.aux = !a_me ? a.vid : (!b_me ? b.vid : me);
              r.packed = (r.packed & ~(EDGE_ARITY_MASK << EDGE_ARITY_SHIFT)) | EDGE_INLINE2 |
                         (a_me ? INLINE2_A_IS_OWNER : 0u) | (b_me ? INLINE2_B_IS_OWNER : 0u) |
                         (a.equal_to << INLINE2_PRED_A_SHIFT) | (b.equal_to << INLINE2_PRED_B_SHIFT);
            }
        }
      }
      // every record learns the lane of its owning variable inside the tile
      for (uint32_t l = 0; l < t.nv; ++l)
        for (uint32_t e = g.row_ptr[g.v_row[v0 + l]]; e < g.row_ptr[g.v_row[v0 + l + 1]]; ++e)
          g.edges[e].packed = (g.edges[e].packed & 0x00FFFFFFu) | ((l & 0xFFu) << EDGE_OWNER_SHIFT);
      g.tiles[i] = t;
    }
    }, 64);
    g.n_terms2_tiles = n_terms2;
    // ---- vif entries of the small factors, per record in record order ----
    // (records that became EDGE_INLINE2 carry theirs already and take no entries)
    {
      auto wants = [](const EdgeRec &r) -> uint32_t {
        if (r.packed & (EDGE_PRESIGNED | EDGE_INLINE2)) return 0u;
        const uint32_t a = (r.packed >> EDGE_ARITY_SHIFT) & EDGE_ARITY_MASK;
        return a >= 2 && a <= VIF_PER_RECORD_ARITY ? a : 0u;
      };
      const size_t nt = g.tiles.size();
      std::vector<uint64_t> tile_off(nt + 1, 0);
      parallel_ranges(nt, nth, [&](uint64_t tb, uint64_t te) {
        for (uint64_t i = tb; i < te; ++i) {
          const TileDesc &t = g.tiles[i];
          uint64_t n = 0;
          for (uint64_t e = t.e0; e < (uint64_t)t.e0 + t.nedges; ++e) n += wants(g.edges[e]);
          tile_off[i + 1] = n;
        }
      });
      for (size_t i = 0; i < nt; ++i) tile_off[i + 1] += tile_off[i];
      // (a graph whose per-record copies would not fit 32-bit bases keeps the small factors'
      // entries once per factor, behind the large ones: slower gathers, same results)
      const bool per_record = nvif + tile_off[nt] < kUnset && !o.no_record_vifs;
      if (!per_record && nvif + nsmall >= kUnset) throw LimitError("vif entries exceed 2^32-1");
      g.NVif = nvif + (per_record ? tile_off[nt] : nsmall);
      g.vifs.resize(g.NVif + 2);   // + 2 padding entries: branch-free pair loads of non-binary records
      if (!per_record) {
        parallel_ranges(nsmall, nth, [&](uint64_t b, uint64_t e) {
          for (uint64_t i = b; i < e; ++i) g.vifs[nvif + i] = small_vifs[i];
        });
        parallel_ranges(g.NIdx, nth, [&](uint64_t b, uint64_t e) {
          for (uint64_t i = b; i < e; ++i)
            if (wants(g.edges[i])) g.edges[i].aux += (uint32_t)nvif;
        });
      }
      if (per_record) parallel_ranges(nt, nth, [&](uint64_t tb, uint64_t te) {
        for (uint64_t i = tb; i < te; ++i) {
          const TileDesc &t = g.tiles[i];
          uint64_t at = nvif + tile_off[i];
          for (uint64_t e = t.e0; e < (uint64_t)t.e0 + t.nedges; ++e) {
            EdgeRec &r = g.edges[e];
            const uint32_t a = wants(r);
            if (!a) continue;
            for (uint32_t k = 0; k < a; ++k) g.vifs[at + k] = small_vifs[r.aux + k];
            r.aux = (uint32_t)at;
            at += a;
          }
        }
      });
    }
    // All-unary graph (every record pre-signed): the sweeps stream 8-byte records instead
    // (the 16-byte ones stay for the oversized-variable kernel and the terms table).
    {
      bool all_simple = g.NIdx > 0 && W <= REC8_WID_MASK && !o.no_compact_records;
      for (size_t i = 0; i < g.tiles.size() && all_simple; ++i) all_simple = (g.tiles[i].flags & TILE_SIMPLE) != 0;
      g.edges8.clear();
      if (all_simple) {
        g.edges8.reset(g.NIdx);
        parallel_ranges(g.NIdx, nth, [&](uint64_t eb, uint64_t ee) {
          for (uint64_t e = eb; e < ee; ++e) {
            const EdgeRec &r = g.edges[e];
            float miss;
            std::memcpy(&miss, &r.aux, 4);
            // hit = s_hit * f, miss = s_miss * f with signs in {-1, 0, +1}: recover (s, f)
            const float f = r.fval != 0.0f ? std::fabs(r.fval) : std::fabs(miss);
            auto code = [&](float x) -> uint32_t { return x == 0.0f ? 1u : (x < 0.0f ? 0u : 2u); };
            EdgeRec8 c;
            c.key = r.wid | ((r.packed & EDGE_FIXED_FLAG) ? REC8_FIXED : 0u) |
                    (code(r.fval) << REC8_HIT_SHIFT) | (code(miss) << REC8_MISS_SHIFT);
            c.f = f;
            g.edges8[e] = c;
          }
        });
      }
    }
    // What a multi-GPU driver may assume about the gradient sums (graph_compile.h: grad_shift)
    g.grad_shift = 0; g.grad_unit_max = 0; g.max_records_per_weight = 0;
    if (g.edges8.size() && !g.has_categorical && W > 0 && !o.no_narrow_info) {
      // (records per weight: a private histogram per thread where that fits -- 10^8 relaxed atomic
      // increments on one shared table were 0.2 s of config 3's compile -- else the shared table)
      uint32_t T = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(nth, g.NIdx / 65536 + 1));
      const bool private_hist = (uint64_t)W * 4 * std::min(T, 16u) <= (512ull << 20);
      if (private_hist) T = std::min(T, 16u);
      std::vector<uint64_t> all_or(T, 0), qmax(T, 0);
      std::vector<std::vector<uint32_t>> hist(private_hist ? T : 1);
      if (!private_hist) hist[0].assign(W, 0);
      parallel_parts(g.NIdx, T, [&](uint32_t t, uint64_t eb, uint64_t ee) {
        uint64_t o = 0, m = 0;
        if (private_hist) hist[t].assign(W, 0);
        uint32_t *per_w = hist[private_hist ? t : 0].data();
        for (uint64_t e = eb; e < ee; ++e) {
          const EdgeRec8 &c = g.edges8[e];
          const double dv = std::fabs((double)sorted_rec_d(c));
          if (dv == 0.0 || (c.key & REC8_FIXED)) continue;
          const uint64_t q = (uint64_t)std::llrint(FIX_SCALE * dv);
          o |= q; m = std::max(m, q);
          if (private_hist) ++per_w[c.key & REC8_WID_MASK];
          else __atomic_fetch_add(&per_w[c.key & REC8_WID_MASK], 1u, __ATOMIC_RELAXED);
        }
        all_or[t] = o; qmax[t] = m;
      }, 0);
      uint64_t o = 0, m = 0;
      for (uint32_t t = 0; t < T; ++t) { o |= all_or[t]; m = std::max(m, qmax[t]); }
      if (o) {
        g.grad_shift = (uint32_t)__builtin_ctzll(o);      // (the lowest set bit of any contribution)
        g.grad_unit_max = m >> g.grad_shift;
        std::vector<uint32_t> wmax(T, 0);
        parallel_parts(W, T, [&](uint32_t t, uint64_t wb, uint64_t we) {
          uint32_t mx = 0;
          for (uint64_t w = wb; w < we; ++w) {
            uint32_t n = 0;
            for (const auto &h : hist) if (!h.empty()) n += h[w];
            mx = std::max(mx, n);
          }
          wmax[t] = mx;
        }, 0);
        g.max_records_per_weight = *std::max_element(wmax.begin(), wmax.end());
      }
    }
    // Weight-sorted super-tiles over the boolean lane-bin tiles of a compact-record graph.
    g.sorted_recs.clear(); g.supers.clear(); g.sort_dvals.clear(); g.sort_dbits.clear(); g.n_sorted = 0;
    g.sorted_per_super = 0; g.sorted_slots = 0;
    {
      uint64_t min_w = 4096;   // (a smaller table sits in the CU's L1: the plain stream gathers as fast)
      if (const char *e = getenv("DWX_SORTED_MIN_W")) min_w = (uint64_t)std::max(0L, atol(e));   // test hook
      uint32_t want_super = o.super_tiles ? o.super_tiles : SUPER_TILES_DEFAULT;
      if (const char *e = getenv("DWX_SUPER_TILES")) want_super = (uint32_t)std::max(1L, atol(e));   // experiment hook
      const uint32_t per_super = std::min(want_super, SORT_TV_SLOTS - 1);   // (sorted_sweep_kernel's tile table)
      const uint32_t slots = o.sorted_slots ? o.sorted_slots : 256u * SORT_WG_PER_CU;
      if (g.edges8.size() && W >= min_w && !o.no_sorted_records) {
        // the distinct values of d = (sign(hit) - sign(miss)) * f over the eligible tiles' records
        // (few: feature values repeat); too many for the kernel's LDS table: no sorted copy
        const size_t nt = g.tiles.size();
        const uint32_t T = (uint32_t)std::max<size_t>(1, std::min<size_t>(nth, nt));
        std::vector<std::vector<uint32_t>> local(T);
        std::atomic<bool> any{false};
        parallel_parts(nt, T, [&](uint32_t t, uint64_t tb, uint64_t te) {
          std::vector<uint32_t> &v = local[t];
          uint32_t last = 0; bool have = false;
          for (uint64_t ti = tb; ti < te; ++ti) {
            const TileDesc &td = g.tiles[ti];
            if (!sorted_eligible(td)) continue;
            any.store(true, std::memory_order_relaxed);
            for (uint64_t e = td.e0; e < (uint64_t)td.e0 + td.nedges && v.size() <= 4 * SORT_MAX_DVALS; ++e) {
              const float dv = sorted_rec_d(g.edges8[e]);
              if (dv == 0.0f) continue;
              uint32_t bits; std::memcpy(&bits, &dv, 4);
              if (have && bits == last) continue;
              last = bits; have = true;
              v.push_back(bits);
              if (v.size() % 1024 == 0) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); }
            }
          }
          std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end());
        }, 0);
        std::vector<uint32_t> dbits;
        for (auto &v : local) dbits.insert(dbits.end(), v.begin(), v.end());
        std::sort(dbits.begin(), dbits.end()); dbits.erase(std::unique(dbits.begin(), dbits.end()), dbits.end());
        if (any.load() && dbits.size() + 1 <= SORT_MAX_DVALS) {
          g.sort_dbits = dbits;
          g.sort_dvals.push_back(0.0);
          for (uint32_t b : dbits) { float f; std::memcpy(&f, &b, 4); g.sort_dvals.push_back((double)f); }
          g.sorted_per_super = per_super; g.sorted_slots = slots;
          // the default layout: every launch's query tiles and evidence tiles on their own (an
          // inference sweep launches over the query part only), full rounds first
          std::vector<std::pair<uint32_t, uint32_t>> ranges;
          for (uint64_t l = 0; l < nl; ++l) {
            ranges.push_back({g.launch_tile[l], g.launch_query_tile_end[l]});
            ranges.push_back({g.launch_query_tile_end[l], g.launch_tile[l + 1]});
          }
          SortedLayout lay;
          build_sorted_layout(g, ranges, per_super, slots, true, nth, lay, o.defer_sorted_records != 0);
          g.supers.swap(lay.supers);
          g.sorted_recs = std::move(lay.recs);
          g.n_sorted = lay.n;
          g.sorted_deferred = o.defer_sorted_records != 0 && !g.supers.empty();
          if (g.supers.empty()) { g.sort_dvals.clear(); g.sort_dbits.clear(); }
        }
      }
    }
    phase("sorted records");
    g.giant_tiles.clear(); g.launch_giant.clear(); g.launch_giant_query_end.clear();
    g.wide_tiles.clear(); g.launch_wide.clear();
    for (uint64_t l = 0; l < nl; ++l) {
      g.launch_wide.push_back((uint32_t)g.wide_tiles.size());
      for (uint32_t i = g.launch_tile[l]; i < g.launch_tile[l + 1]; ++i)
        if (g.tiles[i].flags & TILE_WIDE) g.wide_tiles.push_back(i);
    }
    g.launch_wide.push_back((uint32_t)g.wide_tiles.size());
    for (uint64_t l = 0; l < nl; ++l) {
      g.launch_giant.push_back((uint32_t)g.giant_tiles.size());
      bool closed = false;
      for (uint32_t i = g.launch_tile[l]; i < g.launch_tile[l + 1]; ++i) {
        if (i == g.launch_query_tile_end[l]) { g.launch_giant_query_end.push_back((uint32_t)g.giant_tiles.size()); closed = true; }
        if (g.tiles[i].flags & TILE_GIANT) g.giant_tiles.push_back(i);
      }
      if (!closed) g.launch_giant_query_end.push_back((uint32_t)g.giant_tiles.size());
    }
    g.launch_giant.push_back((uint32_t)g.giant_tiles.size());

    phase("tiles");
    g.w_init.assign(d.w_initial_value, d.w_initial_value + W);
    g.w_fixed.assign(d.w_is_fixed, d.w_is_fixed + W);
  } catch (const LimitError &) {
    *limit = true;
    throw;
  }
}

// ---------------------------------------------------------------- registry of large host mappings
// (host_parallel.h: RawArray registers its direct mappings here)
namespace {
std::mutex g_map_mu;
std::vector<std::pair<void *, size_t>> g_maps;
}  // namespace

void mapping_registry_add(void *p, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_map_mu);
  g_maps.emplace_back(p, bytes);
}

void mapping_registry_remove(void *p) {
  std::lock_guard<std::mutex> lk(g_map_mu);
  for (size_t i = 0; i < g_maps.size(); ++i)
    if (g_maps[i].first == p) { g_maps[i] = g_maps.back(); g_maps.pop_back(); return; }
}

void mapping_registry_drop_all() {
  std::vector<std::pair<void *, size_t>> maps;
  {
    std::lock_guard<std::mutex> lk(g_map_mu);
    maps = g_maps;
  }
  constexpr size_t kStep = (size_t)64 << 20;
  struct Step { char *p; size_t n; };
  std::vector<Step> steps;
  for (const auto &m : maps)
    for (size_t off = 0; off < m.second; off += kStep) steps.push_back({(char *)m.first + off, std::min(kStep, m.second - off)});
  const uint32_t T = (uint32_t)std::min<size_t>(std::min(host_threads(), 32u), steps.size() / 4);
  std::atomic<size_t> next{0};
  auto work = [&]() {
    for (size_t i; (i = next.fetch_add(1)) < steps.size();) madvise(steps[i].p, steps[i].n, MADV_DONTNEED);
  };
  std::vector<std::thread> th;
  for (uint32_t t = 1; t < T; ++t) {
    try { th.emplace_back(work); } catch (...) { break; }
  }
  work();
  for (auto &t : th) t.join();
}

}  // namespace dwx
