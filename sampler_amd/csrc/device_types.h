// device_types.h -- records and launch parameters shared by the host graph compiler
// and the HIP kernels.  Plain C++ (no HIP types) so the host-only parts compile
// with g++.
#ifndef DWX_DEVICE_TYPES_H_
#define DWX_DEVICE_TYPES_H_

#include <stdint.h>

namespace dwx {

// FACTOR_FUNCTION_TYPE, /root/reference/src/common.h:35-48
enum : uint32_t {
  FUNC_IMPLY_NATURAL = 0, FUNC_OR = 1, FUNC_AND = 2, FUNC_EQUAL = 3, FUNC_ISTRUE = 4,
  FUNC_LINEAR = 7, FUNC_RATIO = 8, FUNC_LOGICAL = 9, FUNC_AND_CATEGORICAL = 12,
  FUNC_IMPLY_MLN = 13,
};

// One entry of a variable's (value) row: replaces factor_index[] + Factor (48 B) +
// the weight_id indirection of the reference (src/factor_graph.h:127-145) by ONE
// 16-byte record stored variable-major, so a workgroup tile's records are a single
// contiguous, 16-B-aligned HBM range.
struct alignas(16) EdgeRec {
  uint32_t wid;     // weight id
  uint32_t aux;     // arity >= 2: base index into vifs[]; arity == 1: dense equal_to of
                    // the (only) predicate, or -- EDGE_PRESIGNED -- the miss value
  uint32_t packed;  // bits 0-3 func id, bit 4: feature value needs the f64 side
                    // array, bit 5: weight is fixed, bit 6: pre-signed, bits 8-23
                    // arity, bits 24-31 lane of the owning variable within its tile
  float fval;       // feature value (exact when bit 4 is clear)
};
static_assert(sizeof(EdgeRec) == 16, "EdgeRec must be 16 bytes");

constexpr uint32_t EDGE_FUNC_MASK = 0xF;
constexpr uint32_t EDGE_F64_FLAG = 1u << 4;
constexpr uint32_t EDGE_FIXED_FLAG = 1u << 5;   // the record's weight is fixed (no SGD)
// Unary factor with an f32-exact feature value f: the host already folded the sign
// function in.  fval = sign(hit) * f and aux = float bits of sign(miss) * f, where for
// a boolean owner hit means "proposal == 1" and for a categorical owner "proposal ==
// the value of the row the record sits in" (exact: signs are -1, 0 or +1).
constexpr uint32_t EDGE_PRESIGNED = 1u << 6;
// Factor of arity 2 in a TILE_INLINE2 tile: the record carries the factor's two
// FactorToVariable entries itself instead of pointing into vifs[] (one dependent 16-byte
// load per record less).  aux = device position of the endpoint that is not the owner (the
// owner's own position if both are the owner); the arity field is re-used:
//   bit 8: position 0 is the owner, bit 9: position 1 is the owner,
//   bits 10-16 / 17-23: dense predicate value (equal_to, < 128) of position 0 / 1.
constexpr uint32_t EDGE_INLINE2 = 1u << 7;
constexpr uint32_t INLINE2_A_IS_OWNER = 1u << 8, INLINE2_B_IS_OWNER = 1u << 9;
constexpr uint32_t INLINE2_PRED_A_SHIFT = 10, INLINE2_PRED_B_SHIFT = 17, INLINE2_PRED_MASK = 0x7Fu;
constexpr uint32_t EDGE_ARITY_SHIFT = 8;      // bits 8-23: arity (unless EDGE_INLINE2)
constexpr uint32_t EDGE_ARITY_MASK = 0xFFFFu;
constexpr uint32_t EDGE_OWNER_SHIFT = 24;     // bits 24-31: lane of the owning variable in its tile
constexpr uint32_t MAX_ARITY = EDGE_ARITY_MASK;
// factors up to this arity keep their vif entries once per record, in record order (the
// edge-parallel staging streams them); larger factors share one block of entries
constexpr uint32_t VIF_PER_RECORD_ARITY = 3;

// Compact form of a pre-signed record, 8 bytes: what the sweep kernels stream when EVERY
// tile of the graph is TILE_SIMPLE (all-unary graphs: half the record stream of a sweep).
//   key: bits 0-26 weight id, bit 27 weight is fixed, bits 28-29 sign(hit) + 1,
//        bits 30-31 sign(miss) + 1;   f: the feature value (f32-exact by construction)
// so hit = sign(hit) * f and miss = sign(miss) * f are rebuilt with two bit operations.
struct alignas(8) EdgeRec8 {
  uint32_t key;
  float f;
};
static_assert(sizeof(EdgeRec8) == 8, "EdgeRec8 must be 8 bytes");
constexpr uint32_t REC8_WID_BITS = 27, REC8_WID_MASK = (1u << REC8_WID_BITS) - 1;
constexpr uint32_t REC8_FIXED = 1u << 27, REC8_HIT_SHIFT = 28, REC8_MISS_SHIFT = 30;

// Weight-sorted super-tiles (sorted_sweep_kernel): SUPER_TILES consecutive
// boolean all-unary tiles of one launch whose records exist a SECOND time, sorted by weight id
// -- neighbouring lanes of the record stream then gather neighbouring weights: a fraction of an
// L2 request per record instead of one.  A record adds w * d to its owner's potential
// difference pp - pn, d = (sign(hit) - sign(miss)) * f, summed in fixed point in LDS (pot_fix).
//   wid: the weight id;  od: bits 0-13 (SORT_OWNER_BITS) the owner's slot in the super-tile (variable
//   position - SuperTile::v0), bits 14-31 the index of d in the table of distinct values (entry 0 is
//   0.0: the zero-filled lanes past a super-tile's end add nothing).
struct alignas(8) SortRec8 {
  uint32_t wid, od;
};
static_assert(sizeof(SortRec8) == 8, "SortRec8 must be 8 bytes");
#ifndef DWX_SORT_OWNER_BITS
#define DWX_SORT_OWNER_BITS 14
#endif
constexpr uint32_t SORT_OWNER_BITS = DWX_SORT_OWNER_BITS, SORT_OWNER_MASK = (1u << SORT_OWNER_BITS) - 1;
constexpr uint32_t SUPER_NV_MAX = 1u << SORT_OWNER_BITS;      // 16 384 variables: 128 KiB of fixed-point sums
constexpr uint32_t SUPER_TILES_DEFAULT = SUPER_NV_MAX / 256;  // 64 full tiles
constexpr uint32_t SORT_WG_PER_CU = SORT_OWNER_BITS <= 13 ? 2 : 1;   // (14 bits: the CU's whole LDS, one 1024-thread workgroup per CU)
constexpr uint32_t SORT_TV_SLOTS = 2 * SUPER_TILES_DEFAULT;   // tile starts of a super-tile in LDS (+ end), padded
#ifndef DWX_SORT_THREADS
#define DWX_SORT_THREADS 1024
#endif
#ifndef DWX_SORT_K
#define DWX_SORT_K 20
#endif
constexpr uint32_t SORT_THREADS = DWX_SORT_THREADS;   // sorted_sweep_kernel's workgroup
constexpr int SORT_K = DWX_SORT_K;                    // ... and its records in flight per lane
// persist_learn8_kernel: every workgroup adds up all workgroups' gradient rows itself -- few weights only
constexpr uint32_t PERSIST_MAX_W = 128;
constexpr uint32_t SORT_MAX_DVALS = 1024;                     // distinct d values (8 KiB in LDS), else no sorted copy
struct alignas(16) SuperTile {
  uint32_t tile0, ntiles;   // tiles [tile0, tile0 + ntiles)
  uint32_t v0, nv;          // variables [v0, v0 + nv), nv <= SUPER_NV_MAX
  uint32_t lo, hi;          // sorted records [lo | hi << 32, ... + nrec)
  uint32_t nrec, pad;
};
static_assert(sizeof(SuperTile) == 32, "SuperTile must be 32 bytes");

// four entries of a block-pull row (aux_kernels.h)
struct alignas(16) U32x4 { uint32_t v[4]; };

// factor -> variable entry (src/variable.h:154-167), 8 bytes, device variable ids
struct alignas(8) VifRec {
  uint32_t vid;       // DEVICE position of the variable
  uint32_t equal_to;  // dense predicate value
};

// One workgroup tile: a run of consecutive variables (device order) of one colour and
// one type whose value rows and edge records fit the LDS budget -- or ONE oversized
// variable (rows > rcap or edges > ecap), processed straight from HBM.
struct alignas(16) TileDesc {
  uint32_t v0, nv;      // variables [v0, v0 + nv)
  uint32_t r0, nrows;   // value rows [r0, r0 + nrows)
  uint32_t e0, nedges;  // edge records [e0, e0 + nedges)
  uint32_t flags;       // TILE_*
  uint32_t pad1;
};
// every record of the tile is EDGE_PRESIGNED (unary factor, f32-exact feature value):
// the compute phase then needs no global load at all (everything it reads was staged)
constexpr uint32_t TILE_SIMPLE = 1u << 0;
constexpr uint32_t TILE_CATEGORICAL = 1u << 1;   // the tile's variables are categorical
// Learning, all-unary boolean tile of a graph with many weights: instead of scattering
// gradient atomics, the sweep records per variable whether (and with which sign) its two
// chains disagree -- two wave-ballot bit-planes -- and pull_grad_kernel gathers those bits
// through a weight-sorted incidence list.
constexpr uint32_t TILE_PULL = 1u << 2;
// Boolean tile whose every record is pre-signed or a factor of arity 2: an inference sweep
// evaluates all records edge-parallel in the staging pass (batched vif-pair loads, then
// batched neighbour gathers) and stages potential terms, as for TILE_SIMPLE.
constexpr uint32_t TILE_TERMS2 = 1u << 3;
// TILE_TERMS2 tile whose arity-2 records are EDGE_INLINE2 (only handed to kernels built with
// K <= 6, the ones that implement TILE_TERMS2)
constexpr uint32_t TILE_INLINE2 = 1u << 4;
// Tiles the LDS sweep kernels skip -- ONE variable each, processed straight from HBM by a
// cooperating group of lanes (degree-binned execution):
//   TILE_GIANT  rows > rcap or records > ecap: does not fit a tile at all; a whole workgroup
//               per variable (giant_kernel)
//   TILE_WIDE   fits, but has more than wide_min_records records (a lane-per-variable walk
//               would serialise them while its 255 neighbours idle); one WAVE per variable,
//               four variables per workgroup (wide_kernel)
// Boolean tile whose records are pre-signed or factors of arity 2 or 3 (f32-exact values) with at
// least one of arity 3: evaluated edge-parallel in the staging pass like TILE_TERMS2, through the
// general sign functions on batched loads (three factor->variable entries per record, then their
// assignments).  K <= 6 builds.
constexpr uint32_t TILE_TERMS3 = 1u << 7;
constexpr uint32_t TILE_GIANT = 1u << 5;
constexpr uint32_t TILE_WIDE = 1u << 6;
// boolean TILE_TERMS2 / TILE_TERMS3 tile of a graph with many weights: its PRE-SIGNED (unary) records
// take the pull gradient of TILE_PULL tiles -- the sweep publishes the variables' ballots, the
// incidence lists hold those records -- and only the non-unary records scatter atomics
constexpr uint32_t TILE_PULL_UNARY = 1u << 8;
// every value row of the tile holds exactly ONE record (config 4's shape: one unary factor per
// (variable, value)): row r of the tile starts at record e0 + r, the sweeps make the row pointers
// up instead of loading them (a quarter of config 4's traffic)
constexpr uint32_t TILE_UNIT_ROWS = 1u << 9;
constexpr uint32_t TILE_OUTSIDE = TILE_GIANT | TILE_WIDE;
constexpr uint32_t WIDE_MIN_RECORDS_DEFAULT = 192;
static_assert(sizeof(TileDesc) == 32, "TileDesc must be 32 bytes");

// v_meta bits
constexpr uint32_t VM_CATEGORICAL = 1u << 0;
constexpr uint32_t VM_EVIDENCE = 1u << 1;
constexpr uint32_t VM_TRUTHINESS = 1u << 2;  // total_truthiness not ~0
constexpr uint32_t VM_CARD_SHIFT = 8;
constexpr uint32_t MAX_CARD = (1u << 24) - 1;

// option flag bits (KernelParams::flags)
constexpr uint32_t OPT_SAMPLE_EVIDENCE = 1u << 0;
constexpr uint32_t OPT_LEARN_NON_EVIDENCE = 1u << 1;
constexpr uint32_t OPT_NOISE_AWARE = 1u << 2;
constexpr uint32_t OPT_HAS_F64_FVAL = 1u << 3;
constexpr uint32_t OPT_HAS_TRUTHINESS = 1u << 4;
// A learning sweep split into several mini-batches: per-batch update
// counts are not static any more (count them with atomics) and the whole-sweep pull-based
// gradient does not apply (scatter the gradient instead).
constexpr uint32_t OPT_DYNAMIC_T = 1u << 5;
constexpr uint32_t OPT_NO_PULL = 1u << 6;

constexpr double FIX_SCALE = 1073741824.0;  // 2^30: gradient fixed-point scale
// Per-weight curvature bounds h[w] (apply_kernel's saturating step) are summed in fixed point
// too -- order-independent, all-reducible -- at a coarser scale: a bound needs no more, and
// hubs with 10^5 records reach 10^10.
constexpr double H_SCALE = 1024.0;          // 2^10
constexpr double LINEAR_ZERO_THRESHOLD = 0.000001;  // src/common.h:15

constexpr uint32_t BLOCK_THREADS = 256;
constexpr uint32_t STAGE_UNROLL = 12;                       // edge records staged per lane
constexpr uint32_t MAX_ECAP = BLOCK_THREADS * STAGE_UNROLL;  // 3072 records = 48 KiB
// Few weights shared by many factors (the usual DeepDive tying): memory-side atomics on a
// handful of addresses serialise (14x slower per the MI355X guide), so each persistent
// workgroup accumulates G/T in LDS and flushes once at the end.
constexpr uint32_t LDS_AGG_MAX_W = 1024;
constexpr uint32_t MAX_PLAN_BATCHES = 64;                    // mini-batches per colour launch, at most
#ifndef DWX_PULL_RUN
#define DWX_PULL_RUN 16
#endif
constexpr uint32_t PULL_RUN = DWX_PULL_RUN;                  // incidence entries per lane
// Block pull (pull_ell_kernel, un-split sweeps of graphs with many weights): the incidence
// list as a table of BP_ROW * depth entries per (variable block, weight); a block's ballot
// pairs live in LDS.  Entry: bits 0-18 slot inside the block ((tile - first tile) * 256 +
// lane), bits 19-30 index into the table of distinct record deltas; BP_EMPTY = no entry.
constexpr uint32_t BP_THREADS = 1024;
constexpr uint32_t BP_TILES = 2048;               // tiles per variable block: 128 KiB of ballot pairs
constexpr uint32_t BP_ROW = 4;                    // entries per 16-byte row
#ifndef DWX_BP_UNROLL
#define DWX_BP_UNROLL 4
#endif
constexpr uint32_t BP_UNROLL = DWX_BP_UNROLL;     // weight groups in flight per lane
constexpr uint32_t BP_SLOT_BITS = 19, BP_SLOT_MASK = (1u << BP_SLOT_BITS) - 1;
constexpr uint32_t BP_EMPTY = 0xFFFFFFFFu;
constexpr uint32_t BP_DELTA_SLOTS = 4096, BP_MAX_DELTAS = BP_DELTA_SLOTS - 1;   // (the last slot stays 0: BP_EMPTY decodes to it)
// one block of ballot pairs + the table of deltas: exactly the 160 KiB of a CU
constexpr uint32_t BP_LDS_BYTES = BP_TILES * 64 + BP_DELTA_SLOTS * 8;
// sweep8_kernel<MULTI>: a run of at least this many sweeps over a tile with idle lanes is cut
// into this many slices per variable
constexpr uint32_t MULTI_SLICES = 4, MULTI_SLICE_MIN_SWEEPS = 256;
constexpr uint32_t ROWPTR_UNROLL = 2;                       // row pointers prefetched per lane
constexpr uint32_t ROWPTR_UNROLL_CAT = 7;                   // ... for categorical all-unary graphs (sweep8_kernel)                       // row pointers prefetched per lane

// Everything one sweep launch needs; passed by value.
struct KernelParams {
  // graph (read only)
  const uint32_t *v_meta;     // [V]
  const uint32_t *v_orig;     // [V] original variable id (Philox counter)
  const uint32_t *v_row;      // [V+1] first value row of each variable
  const uint32_t *v_init;     // [V] dense evidence value (assignment_dense)
  const uint32_t *row_ptr;    // [R+1] edge-record offsets
  const double *row_truth;    // [R] truthiness per value row, or null
  const EdgeRec *edges;       // [NIdx]
  const EdgeRec8 *edges8;     // [NIdx] compact records of an all-TILE_SIMPLE graph, or null
  const void *edge_terms;     // [NIdx] 16-byte {w*sign(hit)*f, w*sign(miss)*f} (f64) of the pre-signed
                              // records under the CURRENT weights, or null (inference sweeps only)
  const double *edge_fval64;  // [NIdx] or null
  const VifRec *vifs;         // [NVif]
  const TileDesc *tiles;      // [n_tiles]
  // state
  uint32_t *assign_free;      // [V]
  uint32_t *assign_evid;      // [V]
  uint32_t *tally;            // [R]
  const float *w32;           // [W] sampling copy of the weights, rounded to f32 (fits L2)
  long long *grad;            // [2W]: G then T (fixed point)
  unsigned long long *delta;  // [n_tiles*4*2] per wave: {chains disagree, free < evid} ballots
  // launch
  uint64_t seed, sweep;
  uint64_t vid_offset;        // global id of local variable 0 (Philox counter)
  uint32_t tile_begin;        // first tile of this launch
  uint32_t tile_end;          // one past the last tile of this launch
  uint32_t num_weights;
  uint32_t flags;
  uint32_t ecap, rcap;        // LDS capacities (edge records / value rows per tile)
  uint32_t lds_pot_off;       // byte offset of the potentials scratch in dynamic LDS
  uint32_t lds_edge_off;      // byte offset of the staged edge records
  uint32_t lds_w_off;         // byte offset of the staged f32 weights (learning kernel)
  uint32_t lds_agg_off;       // byte offset of the per-workgroup gradient accumulators
                              // (int64[2W], learning kernel, only when W <= LDS_AGG_MAX_W), else 0
  uint32_t n_sweeps;          // MULTI builds of the inference sweep: sweeps [sweep, sweep + n_sweeps) in one launch
};

// A split learning sweep of a few-weights graph as ONE persistent launch (persist_learn8_kernel,
// sweep_kernels.h): what the chunk loop and the in-launch updates need besides KernelParams.
struct PersistArgs {
  const uint32_t *chunk_tiles;   // [2 * n_chunks]: tiles [t0, t1) of chunk c
  uint32_t n_chunks, grid;       // grid == gridDim.x: every workgroup resident (one per CU at most)
  long long *rows;               // [2][grid][row_stride]: the workgroups' gradient sums of chunk c, parity c & 1
  uint32_t row_stride;           // 2W rounded up to whole 128-byte lines (16 entries)
  uint32_t *bar;                 // [0] arrivals (monotonic over the launch), [1] give-up flag; zeroed before every launch
  const long long *t_static;     // [n_chunks][T[W] | h[W]]: static update counts and curvature bounds per chunk
  double *weights;               // [W] master weights: read at the start, written back by workgroup 0 at the end
  float *w32;
  const uint8_t *w_fixed;
  double stepsize, reg_param;
  int32_t l2;
  uint32_t spin_limit;           // polls of one barrier before a workgroup gives up (and takes the launch down)
  uint32_t lds_w64_off;          // dynamic LDS: f64 weights [W], f32 copy [W] right behind
  uint32_t lds_red_off;          // ... reduction scratch of the in-launch update (BLOCK_THREADS x 8 bytes) + flag
};
constexpr uint32_t PERSIST_MIN_CHUNKS = 8;     // fewer mini-batches: plain launches (nothing to win)

// A mini-batch of a split learning sweep of a few-weights graph as ONE launch (sweep8_merged_kernel,
// persist_kernels.h): the previous mini-batch's update runs as the kernel's prologue, by every
// workgroup for itself, into an LDS copy of the f32 weights.
struct MergeArgs {
  const long long *prev_grad;    // [2W] gradient sums / counts of the previous mini-batch, or null (the sweep's first)
  long long *zero;               // [2W] the buffer the NEXT mini-batch accumulates into: zeroed by workgroup 0, or null
  const double *w_src;           // [W] weights before the previous mini-batch's update
  double *w_dst;                 // [W] ... and after it (written by workgroup 0; null with prev_grad == null)
  float *w32_dst;
  const uint8_t *w_fixed;
  const long long *t_static;     // [T[W] | h[W]] of the previous mini-batch, or null
  double stepsize, reg_param;
  int32_t l2;
  uint32_t lds_lw32_off;         // dynamic LDS: the f32 weights [W]
};

}  // namespace dwx
#endif
