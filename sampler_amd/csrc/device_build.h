// device_build.h -- see device_build.hip: heavy phases of the graph build as HIP kernels.  Plain
// declarations over device pointers (no HIP types), so that dwx_api.cc calls them the same way in the
// product (device_build.hip) and in the tests' host harness (tests/hipemu/device_build_stub.cc:
// available() == false, the host builders run).
#ifndef DWX_DEVICE_BUILD_H_
#define DWX_DEVICE_BUILD_H_

#include <stdint.h>

#include "device_types.h"

namespace dwx {
namespace devb {

bool available();

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

}  // namespace devb
}  // namespace dwx
#endif
