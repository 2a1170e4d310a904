// graph_compile.h -- host-side graph compiler: from the columnar image of the
// reference's input files (dwx_graph_desc) to the compact device layout.
//
// Restates, MI355X-first, what the reference does in FactorGraph::load_factors /
// load_domains (dense predicate conversion, src/binary_format.cc:128-226) and
// FactorGraph::construct_index (sort + dedup of (value, factor) back-refs,
// src/factor_graph.cc:90-199), then adds what the device needs: a chromatic
// partition of the variable conflict graph, a colour-major variable order, 16-byte
// variable-major edge records and LDS-sized workgroup tiles.
#ifndef DWX_GRAPH_COMPILE_H_
#define DWX_GRAPH_COMPILE_H_

#include <stdint.h>
#include <string>
#include <utility>
#include <vector>

#include "../../include/dwx.h"
#include "device_types.h"
#include "host_parallel.h"

namespace dwx {

struct CompiledGraph {
  uint64_t V = 0, F = 0, E = 0, W = 0;
  uint64_t R = 0;      // value rows (== num_values)
  uint64_t NIdx = 0;   // edge records (== |factor_index| after dedup)
  uint64_t NVif = 0;   // vif entries of factors with arity >= 2
  uint64_t n_query = 0; // non-evidence owned variables
  uint64_t Vo = 0;      // owned (sampled) variables: ids [0, Vo); ids [Vo, V) are ghosts
  uint32_t n_colors = 0, n_giant_tiles = 0, max_card = 2, n_terms2_tiles = 0;
  bool has_categorical = false, has_truthiness = false, has_f64_fval = false;
  bool order_is_identity = true;
  uint32_t tile_vars = 256, ecap = 3072, rcap = 256;

  // ---- reference numbering (for dumps and parity) ----
  std::vector<uint64_t> ref_var_val_base;  // [V]
  std::vector<uint64_t> value_sparse;      // [R]   values[].value
  std::vector<uint64_t> ref_row_ptr;       // [R+1] into ref_fidx
  RawArray<uint32_t> ref_fidx;             // [NIdx] factor ids, reference order

  // ---- device order ----
  // (uninitialised storage, first touched by the parallel fills: six serial zero-fills of 400 MB each at
  // config 5's size otherwise)
  RawArray<uint32_t> perm;         // [V] position -> original variable id
  RawArray<uint32_t> pos;          // [V] original id -> position
  RawArray<uint32_t> v_meta;       // [V]
  RawArray<uint32_t> v_init;       // [V]
  RawArray<uint32_t> v_row;        // [V+1]
  RawArray<uint32_t> row_ptr;      // [R+1]
  std::vector<double> row_truth;   // [R] or empty
  RawArray<EdgeRec> edges;         // [NIdx] (uninitialised storage: first touch inside the parallel fill)
  RawArray<EdgeRec8> edges8;       // [NIdx] when every tile is TILE_SIMPLE (and W < 2^27), else empty
  // weight-sorted second copy of the boolean all-unary tiles' records (compact-record graphs
  // with enough weights for the gathers to matter), super-tile by super-tile
  RawArray<SortRec8> sorted_recs;   // [n_sorted] or empty
  std::vector<SuperTile> supers;    // ascending tile0; a super-tile never crosses a launch or its query end
  std::vector<double> sort_dvals;   // distinct (sign(hit) - sign(miss)) * f values, [0] = 0.0
  std::vector<uint32_t> sort_dbits; // ... their f32 bit patterns, ascending (entry i <-> sort_dvals[i + 1])
  uint64_t n_sorted = 0;
  uint32_t sorted_per_super = 0, sorted_slots = 0;   // the layout's parameters (0: no sorted copy)
  bool sorted_deferred = false;     // supers are planned (lo / hi / nrec set), sorted_recs is EMPTY: every sampler
                                    // builds the records on its device (device_build.hip)
  // All-boolean compact-record graphs: every gradient contribution is +-round(2^30 * d_r),
  // d_r = (sign(hit) - sign(miss)) * f -- a multiple of 2^grad_shift, at most grad_unit_max of
  // those units, and no weight has more than max_records_per_weight records (a multi-GPU driver
  // may then all-reduce the gradient sums as 32-bit counts).  grad_shift == 0: not known.
  uint32_t grad_shift = 0;
  uint64_t grad_unit_max = 0, max_records_per_weight = 0;
  std::vector<double> edge_fval64; // [NIdx] or empty
  std::vector<VifRec> vifs;        // [NVif]
  std::vector<uint32_t> tile_v;       // [n_tiles+1]
  std::vector<TileDesc> tiles;        // [n_tiles]
  std::vector<uint32_t> giant_tiles;  // indices of oversized tiles, launch-major
  std::vector<uint32_t> launch_giant; // [n_launches+1] into giant_tiles
  std::vector<uint32_t> launch_giant_query_end; // [n_launches]
  std::vector<uint32_t> wide_tiles;   // indices of TILE_WIDE tiles, launch-major
  std::vector<uint32_t> launch_wide;  // [n_launches+1] into wide_tiles
  uint32_t n_wide_tiles = 0, wide_min = 0;
  std::vector<uint32_t> launch_tile;  // [n_launches+1] into tile_v
  std::vector<uint32_t> launch_query_tile_end;  // [n_launches] end of the query-variable tiles
  std::vector<uint64_t> launch_off;   // [n_launches+1] variable positions
  std::vector<uint8_t> var_is_evid;   // [V] original order (for nsamples)

  std::vector<double> w_init;      // [W]
  std::vector<uint8_t> w_fixed;    // [W]

  uint64_t device_bytes() const;
};

// A weight-sorted layout: super-tiles + their sorted records (the graph's default one lives in
// CompiledGraph; a split mini-batch plan builds one per level, cut along its chunks).
struct SortedLayout {
  std::vector<SuperTile> supers;
  RawArray<SortRec8> recs;
  uint64_t n = 0;
};
// plan_only: the super-tiles with their record ranges (lo / hi / nrec) and out.n, no records
void build_sorted_layout(const CompiledGraph &g, const std::vector<std::pair<uint32_t, uint32_t>> &ranges,
                         uint32_t per_super, uint32_t slots, bool full_rounds, uint32_t n_threads, SortedLayout &out,
                         bool plan_only = false);

// Throws std::runtime_error (message for dwx_last_error) on malformed input;
// `limit` is set when the failure is a 32-bit layout limit.
void compile_graph(const dwx_graph_desc &d, const dwx_compile_opts &o, CompiledGraph &g,
                   bool *limit);

}  // namespace dwx
#endif
