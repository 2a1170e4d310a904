// device_build.h -- see device_build.hip: heavy phases of the graph build as HIP kernels.  Plain
// declarations over device pointers (no HIP types), so that dwx_api.cc calls them the same way in the
// product (device_build.hip) and in the tests' host harness (tests/hipemu/device_build_stub.cc:
// available() == false, the host builders run).
#ifndef DWX_DEVICE_BUILD_H_
#define DWX_DEVICE_BUILD_H_

#include <stdint.h>

#include <vector>

#include "device_types.h"

namespace dwx {
namespace devb {

bool available();

// The builds below take their large temporaries from a cache of device blocks that are handed back instead of
// freed (device_build.hip: a hipFree of gigabytes is paid by a later hipMalloc, a second per 16 GB).
// release_scratch really frees the idle blocks of `device` (-1: all) -- with only_if_tight, only when they are a
// quarter or more of the device memory that is still free; scratch_bytes is what the cache holds.
void release_scratch(int device, bool only_if_tight);
uint64_t scratch_bytes(int device);

// The weight-sorted records of `n_supers` super-tiles (SuperTile::lo/hi/nrec already planned: the
// records of super-tile i go to d_out[lo | hi << 32 .. + nrec)), n_total records in all, sorted by
// (weight id, owner) inside every super-tile -- byte for byte what build_sorted_layout (graph_compile.cc)
// produces on the host.  h_dbits: the ascending f32 bit patterns of the distinct record deltas
// (host memory).  Synchronises `stream` (a hipStream_t) before it returns.
void build_sorted_records(const TileDesc *d_tiles, const EdgeRec *d_edges, const EdgeRec8 *d_edges8,
                          const SuperTile *d_supers, uint32_t n_supers, const uint32_t *h_dbits, uint32_t n_dbits,
                          uint64_t n_total, SortRec8 *d_out, void *stream);

// A plan level's static tables (build_level (a) of dwx_api.cc): d_table[n_groups][T[W] | h[W]] --
// update counts of the boolean variables (fixed point 2^-30) and curvature bounds of all variables
// (2^-10) of the tiles of every group (h_group_of[tile], host memory; 0xFFFFFFFF: no group) -- bit for
// bit the host's; *t_max / *h_max: the largest entries.  Synchronises `stream`.
void build_static_tables(const TileDesc *d_tiles, uint32_t n_tiles, const uint32_t *h_group_of, const uint32_t *d_v_meta,
                         const uint32_t *d_v_row, const uint32_t *d_row_ptr, const EdgeRec *d_edges, const double *d_fval64,
                         const uint8_t *d_w_fixed, uint32_t W, uint32_t n_groups, bool learn_non_evidence, bool noise_aware,
                         long long *d_table, long long *t_max, long long *h_max, void *stream);

// A plan level's pull-gradient structures (build_level (b) of dwx_api.cc), built on the device: the
// incidence list of every (SGD-triggering boolean variable, non-fixed record with a non-zero delta)
// pair of the pull tiles, sorted by (group, weight), and -- graphs with >= block_pull_min_w weights and
// few distinct deltas -- per group the block-pull table (rows of BP_ROW x depth entries per (variable
// block, weight)), with what did not fit a row left on the list.  Bit for bit the host builder's
// structures.  h_tile_info[tile] = group | mode << 30 (1: all records, 2: pre-signed ones only),
// 0xFFFFFFFF: none.  The device buffers of the result belong to the caller (hipFree / rt::dfree).
struct Incidence {
  uint64_t n_entries = 0;
  std::vector<uint32_t> inc_begin, inc_end;      // per group: its part of the list columns (multiples of PULL_RUN)
  uint32_t *d_inc_wid = nullptr, *d_inc_slot = nullptr;
  float *d_inc_d = nullptr;
  struct BlockTable {
    U32x4 *d_ell = nullptr;
    std::vector<uint32_t> tile0;
    uint32_t blocks = 0, depth = 0;
    uint64_t total = 0, on_list = 0;             // the group's entries; those left on the list
  };
  std::vector<BlockTable> bp;                    // [groups], or empty: no block pull
  std::vector<uint32_t> dvals;                   // ascending f32 bit patterns of the distinct deltas
  uint64_t max_blocks = 0, wp = 0;
};
void build_incidence(const TileDesc *d_tiles, const TileDesc *h_tiles, uint32_t n_tiles, const uint32_t *h_tile_info,
                     const uint32_t *d_v_meta, const uint32_t *d_v_row, const uint32_t *d_row_ptr, const EdgeRec *d_edges,
                     bool learn_non_evidence, bool noise_aware, uint32_t W, uint32_t n_groups, uint64_t block_pull_min_w,
                     uint32_t bp_tiles, Incidence &out, void *stream);

// max(power-iteration estimate, largest diagonal entry) of the curvature bound of the mini-batch of
// variables [p0, p1) (row_sum_bound's dense branch in dwx_api.cc, WITHOUT its 1.1 safety factor): the
// host's arithmetic term for term, 64-bit fixed-point sums.  `sc`: scratch vectors kept between calls.
struct CurvatureScratch {
  void *x = nullptr, *y = nullptr, *yfix = nullptr, *dfix = nullptr, *part = nullptr, *small = nullptr;
  ~CurvatureScratch();
};
double batch_curvature(uint32_t p0, uint32_t p1, const uint32_t *d_v_meta, const uint32_t *d_v_row, const uint32_t *d_row_ptr,
                       const EdgeRec *d_edges, const double *d_fval64, bool learn_non_evidence, bool noise_aware, uint32_t W,
                       CurvatureScratch &sc, void *stream);

}  // namespace devb
}  // namespace dwx
#endif
