/*
 * dwx.h -- C ABI of the MI355X-native DimmWitted Gibbs sweep ("dwx").
 *
 * This is the drop-in boundary for the reference's ONE hot path: the seam between
 * the epoch driver (DimmWitted::learn / inference, /root/reference/src/dimmwitted.cc:
 * 121-207) and the per-replica sampler (class GibbsSampler,
 * /root/reference/src/gibbs_sampler.h:18-57).  The reference has no FFI of its own;
 * every entry point below names the internal C++ interface it replaces.  Plain
 * pointers and sizes only -- no C++ or torch types cross this boundary.
 *
 * Conventions: every function returns 0 on success and a negative DWX_E_* code on
 * error (dwx_last_error() gives the text; the reference assert()s/abort()s in the
 * same situations).  Handles are opaque.  Host arrays passed in are copied during
 * the call; the library owns all device memory until *_destroy.  One host thread
 * drives one sampler handle; *_async calls enqueue on the handle's HIP stream and
 * dwx_wait() is the barrier, exactly like GibbsSampler::sample()/wait().
 *
 * There is NO CPU fallback: dwx_sampler_create fails with DWX_E_DEVICE when no HIP
 * device is usable.  dwx_graph_* functions are host-only (graph compilation) and
 * work without a GPU.
 */
#ifndef DWX_H_
#define DWX_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DWX_VERSION 1

enum {
  DWX_OK = 0,
  DWX_E_INVALID = -1,  /* malformed graph / argument (reference: assert/abort)   */
  DWX_E_LIMIT = -2,    /* graph exceeds the compact 32-bit layout               */
  DWX_E_DEVICE = -3,   /* HIP error or no device                                */
  DWX_E_NOMEM = -4
};

/* Columnar image of the reference's binary input files
 * (/root/reference/doc/binary_format.md; loaders src/binary_format.cc:48-226).
 * Ids must be dense (0..N-1), as FactorGraph::safety_check demands
 * (src/factor_graph.cc:201-219). */
typedef struct dwx_graph_desc {
  uint64_t num_variables, num_factors, num_edges, num_weights;
  const uint8_t *var_role;         /* isEvidence byte; evidence iff >= 1          */
  const uint64_t *var_init_value;  /* initialValue as in the file                 */
  const uint16_t *var_dtype;       /* 0 boolean, 1 categorical                    */
  const uint64_t *var_cardinality;
  uint64_t num_domains;            /* categorical domain blocks (may be 0)        */
  const uint64_t *dom_vid;         /* [num_domains]                               */
  const uint64_t *dom_offset;      /* [num_domains+1] into dom_value/truthiness   */
  const uint64_t *dom_value;
  const double *dom_truthiness;
  const uint16_t *fac_func;        /* FACTOR_FUNCTION_TYPE (src/common.h:35-48)   */
  const uint64_t *fac_edge_offset; /* [num_factors+1]                             */
  const uint64_t *fac_weight_id;
  const double *fac_feature_value; /* finite; |value| <= 65536 on a learnable weight (DWX_E_LIMIT
                                      beyond: gradients are summed in 2^-30 fixed point, int64);
                                      |value| below ~5e-10 contributes no gradient            */
  const uint64_t *edge_vid;
  const uint64_t *edge_equal_to;   /* equalPredicate as in the file               */
  const double *w_initial_value;   /* indexed by weight id                        */
  const uint8_t *w_is_fixed;
  /* Sharding: the LAST num_ghost_variables variables are ghosts -- remote variables
   * that local factors read.  They hold an assignment (refreshed by the caller's halo
   * exchange) but are never sampled, tallied or given rows.  0 for a whole graph. */
  uint64_t num_ghost_variables;
} dwx_graph_desc;

/* Graph-compilation knobs; zero-initialise for defaults. */
typedef struct dwx_compile_opts {
  uint32_t tile_vars;          /* variables per workgroup tile (default 256, max 256)  */
  uint32_t tile_edges;         /* edge records staged in LDS per tile (default 3072)   */
  uint32_t tile_rows;          /* value rows per tile (default 256 bool / 1536 categ.) */
  uint32_t conflict_arity_cap; /* factors wider than this do not constrain the
                                  colouring (Hogwild reads, as in the reference);
                                  default 256                                          */
  uint32_t n_threads;          /* host threads for the build (0 = DWX_HOST_THREADS, else the hardware
                                  concurrency capped at 64 and at twice the cgroup's CPU quota)  */
  uint32_t no_compact_records; /* 1: keep 16-byte records even for all-unary graphs
                                  (default 0: such graphs stream 8-byte records)       */
  uint32_t no_weight_order;    /* 1: keep variables in id order inside a class (default 0: an
                                  all-unary graph orders them by the weight id of their first
                                  factor -- locality for the weight gathers)              */
  uint32_t wide_min_records;   /* degree binning: a variable with more edge records than this
                                  is walked by a whole wave instead of one lane (default 192;
                                  0xFFFFFFFF: never)                                   */
  uint32_t no_record_vifs;     /* 1: factors of arity 2-3 keep their factor->variable entries once
                                  per factor (default 0: once per edge record, in record order,
                                  streamed by the staging pass -- 2-3x the entries; chosen
                                  automatically when the copies would not fit 32-bit bases) */
  uint32_t no_pull_unary;      /* 1: tiles with non-unary factors scatter the gradient of all their
                                  records (default 0: their unary records take the pull gradient
                                  of all-unary tiles when the graph has more than 1024 weights) */
  uint32_t no_sorted_records;  /* 1: no weight-sorted second copy of the records of boolean all-unary
                                  tiles (default 0: compact-record graphs with >= 4096 weights get
                                  one, and their sweeps gather weights in sorted order)        */
  uint32_t super_tiles;        /* tiles per weight-sorted super-tile, at most (default 64 = 16 384 variables:
                                  128 KiB of fixed-point sums, the CU's LDS)                            */
  uint32_t sorted_slots;       /* workgroups of sorted_sweep_kernel resident at once (default 256: one
                                  1024-thread workgroup per CU of an MI355X); runs of tiles are cut into
                                  multiples of it                                                       */
  uint32_t defer_sorted_records; /* set by dwx_graph_create itself where the library can build on the device
                                  (always in the product; DWX_HOST_BUILD=1 in the environment keeps the host
                                  builder): the weight-sorted copy is only PLANNED here and every sampler
                                  builds its records on its own device (device_build.hip)               */
  uint32_t no_narrow_info;     /* 1: do not scan the records for dwx_graph_info.grad_shift / grad_unit_max /
                                  max_records_per_weight (they read 0: "nothing is known, send int64") -- a
                                  pass over every record with a histogram per weight that only multi-GPU
                                  drivers need; `dw gibbs` on one GPU sets it (2 s at config 5's size)      */
} dwx_compile_opts;

typedef struct dwx_graph_info {
  uint64_t num_variables, num_factors, num_edges, num_weights;
  uint64_t num_owned_variables; /* num_variables - ghosts: the variables this graph samples */
  uint64_t num_values;         /* value rows: 1 per boolean, cardinality per categorical
                                  (== FactorGraph size.num_values)                     */
  uint64_t num_index_entries;  /* |factor_index| after dedup (src/factor_graph.cc:177) */
  uint64_t num_vif_entries;    /* factor->variable entries kept for arity >= 2         */
  uint64_t num_colors, num_launches, num_tiles, num_giant_tiles;   /* giant: workgroup-per-variable */
  uint64_t max_cardinality;
  uint64_t num_query_variables; /* non-evidence variables (sampled by an inference sweep
                                   unless --sample_evidence)                            */
  uint64_t device_bytes;       /* bytes the sampler will hold in HBM                   */
  uint32_t has_categorical, order_is_identity;
  uint64_t num_wide_tiles;     /* variables walked by a wave each (degree bin, see
                                  dwx_compile_opts.wide_min_records)                   */
  uint64_t num_staged_tiles;   /* tiles whose non-unary factors (arity <= 3) are evaluated
                                  edge-parallel while the tile is staged  */
  uint64_t num_super_tiles;    /* weight-sorted super-tiles (sorted_sweep_kernel) and the records of */
  uint64_t num_sorted_records; /* their sorted second copy; 0: no sorted copy             */
  /* Multi-GPU drivers: on an all-boolean all-unary graph every contribution to DWX_BUF_GRAD's
   * gradient sums is a multiple of 2^grad_shift (fixed point 2^-30), at most grad_unit_max such
   * units, and no weight has more than max_records_per_weight records -- so the sums may travel as
   * 32-bit counts (G >> grad_shift) while ranks x max_records_per_weight x grad_unit_max < 2^31.
   * grad_shift == 0: nothing is known (categorical variables, non-unary factors): send int64. */
  uint64_t grad_shift, grad_unit_max, max_records_per_weight;
} dwx_graph_info;

/* Runtime options of one sampler (the CmdParser fields the hot path reads:
 * src/gibbs_sampler.cc:46-48, src/inference_result.h:66-85). */
typedef struct dwx_options {
  int32_t device;              /* HIP device ordinal                                   */
  int32_t sample_evidence;     /* --sample_evidence                                    */
  int32_t learn_non_evidence;  /* --learn_non_evidence                                 */
  int32_t noise_aware;         /* --noise_aware                                        */
  int32_t regularization;      /* 0 = l1, 1 = l2 (reference enum order)                */
  int32_t plan_layouts;        /* weight-sorted layouts of the LEARNING sweeps' own (cut along a
                                  split plan's mini-batches, or -- un-split -- over whole launches
                                  instead of their query / evidence parts: a few % per sweep for
                                  a second copy of the records each):
                                  0 = default: built with the plan level (the records are sorted on the
                                  device in milliseconds; a DWX_HOST_BUILD=1 run builds them on the host
                                  -- 0.3 s per level -- once the level has run 2048 sweeps),
                                  1 = always with the level, 2 = never                      */
  double reg_param;            /* -b / --reg_param                                     */
  double step_cap;             /* bound on stepsize x R of one SGD mini-batch (see
                                  dwx_sgd_plan); <= 0: never split a sweep; default 1.5
                                  (dwx_default_options)                                 */
  uint64_t seed;               /* Philox key                                           */
  uint64_t var_id_offset;      /* added to local variable ids in the Philox counter: the
                                  global id of this shard's variable 0 (multi-GPU)     */
} dwx_options;

typedef struct dwx_graph dwx_graph;
typedef struct dwx_sampler dwx_sampler;

const char *dwx_last_error(void);
int dwx_version(void);
void dwx_default_options(dwx_options *o);

/* ---- graph compilation (host only) --------------------------------------------
 * Replaces FactorGraph::load_* conversions + construct_index
 * (src/binary_format.cc:128-226, src/factor_graph.cc:90-199) and adds what the
 * device needs: chromatic partition, colour-major variable order, LDS tiles. */
int dwx_graph_create(const dwx_graph_desc *desc, const dwx_compile_opts *opts, dwx_graph **out);
void dwx_graph_destroy(dwx_graph *g);
int dwx_graph_get_info(const dwx_graph *g, dwx_graph_info *out);
/* Execution schedule: order[num_owned_variables] = original variable ids in device order;
 * launch_off[num_launches+1] = offsets into order; every launch is an independent
 * set of the variable conflict graph. */
int dwx_graph_get_schedule(const dwx_graph *g, uint64_t *order, uint64_t *launch_off);
/* mask[num_variables] (reference numbering): 1 where the device sums the variable's potentials
 * in fixed point (2^-32; boolean variables of an all-unary graph on the lane path -- the sums
 * are then independent of the order of the records, which the weight-sorted sweep relies on),
 * 0 where it adds f64 terms in the reference's row order (src/factor_graph.h:127-145).  Test
 * hook: the CPU oracle's schedule mode follows it, so device-vs-oracle parity stays exact. */
int dwx_graph_get_fixed_point_mask(const dwx_graph *g, uint8_t *mask);
/* Value table for result dumps (src/inference_result.cc:211-243):
 * var_val_base[V] (reference numbering) and value_sparse[num_values]. */
int dwx_graph_get_values(const dwx_graph *g, uint64_t *var_val_base, uint64_t *value_sparse);
/* Device positions of variables (index into DWX_BUF_ASSIGN_*), for halo exchange:
 * out[i] = position of original variable id vids[i]. */
int dwx_graph_get_positions(const dwx_graph *g, const uint64_t *vids, uint64_t n, uint64_t *out);
/* Reference-order CSR, for parity checks against construct_index:
 * index_base/index_len[num_values], factor_index[num_index_entries]. */
int dwx_graph_get_index(const dwx_graph *g, uint64_t *index_base, uint64_t *index_len,
                        uint64_t *factor_index);

/* ---- sampler (device) -----------------------------------------------------------
 * Replaces GibbsSampler(pfg, weights, numa_nodes, nthread, nodeid, opts)
 * (src/gibbs_sampler.cc:5-18) incl. the InferenceResult it owns
 * (src/inference_result.cc:24-42). */
int dwx_sampler_create(const dwx_graph *g, const dwx_options *opts, dwx_sampler **out);
/* Optional: pay the HIP runtime's start-up (context creation on `device`) now -- e.g. on a
 * helper thread while dwx_graph_create, which is host-only, runs.  dwx_sampler_create does
 * the same on its own if this was never called.  (No reference counterpart: the reference
 * has no device.) */
int dwx_device_init(int32_t device);
/* Number of usable HIP devices (0 without a GPU: not an error). */
int dwx_device_count(int32_t *count);
/* Frees the sampler's device buffers and hands the device builds' idle scratch blocks (sort buffers
 * kept in a cache inside the library: INTEGRATION.md, "Device memory a caller may notice") back to the runtime. */
void dwx_sampler_destroy(dwx_sampler *s);

/* GibbsSampler::sample(i_epoch) (src/gibbs_sampler.cc:20-25): one inference sweep. */
int dwx_sample_async(dwx_sampler *s);
/* The inference loop of DimmWitted::inference (src/dimmwitted.cc:131-156: `for i_epoch:
 * sample(i_epoch); wait()`), n_sweeps of its iterations in one call: state, tallies and sweep
 * counter afterwards are bit for bit those of n_sweeps calls of dwx_sample_async.  On a graph
 * whose factors are all unary (no variable reads another, so nothing a sweep reads changes
 * between sweeps) the sweeps share ONE launch: every record is read once, every variable draws
 * n_sweeps times with the uniforms its separate sweeps would have used.  Any other graph runs
 * the n_sweeps sweeps one after the other. */
int dwx_sample_n_async(dwx_sampler *s, uint32_t n_sweeps);
/* GibbsSampler::sample_sgd(stepsize) (src/gibbs_sampler.cc:27-33): one learning
 * sweep = dwx_sgd_plan + (accumulate, apply)* + dwx_sgd_finish. */
int dwx_sample_sgd_async(dwx_sampler *s, double stepsize);
/* GibbsSampler::wait() (src/gibbs_sampler.cc:35-38). */
int dwx_wait(dwx_sampler *s);

/* A learning sweep in pieces, for drivers that put a collective between accumulation and
 * update (multi-GPU: all-reduce DWX_BUF_GRAD; replaces the dormant
 * InferenceResult::merge_gradients_from, src/inference_result.cc:57-62).
 *
 * The reference applies every factor's SGD update immediately; the device accumulates a
 * mini-batch and applies it in one step.  Two things keep that step faithful to the
 * reference's sequential updates:
 *  (1) the step of every weight SATURATES at the inverse of its own curvature bound h[w]
 *      (dwx_sgd_apply_async: w -= (1 - exp(-c stepsize)) / c * (G + reg T w), c = h[w] + reg T
 *      -- the end point of the gradient flow the T sequential updates follow).  A weight tied
 *      to millions of factors can therefore never overshoot, whatever the step size; a weight
 *      with few factors takes the reference's own step.  h is a static table (Gershgorin row
 *      sums of the batch's curvature), part of DWX_BUF_TSTATIC*.
 *  (2) dwx_sgd_plan sizes the mini-batches for the given step size: with R = the (estimated)
 *      largest eigenvalue of a batch's curvature in weight space (how strongly all updates of
 *      one batch interact), every colour launch is cut into `batches` runs of tiles carrying
 *      equal SGD work, batches = the first power of two >= stepsize * R(1) / step_cap with
 *      stepsize * R(batches) <= step_cap (1 for configs 2-3 of BASELINE.json at their quoted
 *      step; dozens for heavily tied weights with a large step; never more than 64).  Inside a
 *      batch all draws see the weights of its start, so the cut decides how closely the run
 *      follows the reference from epoch to epoch -- not whether it is stable.
 * effective_stepsize reports the SMALLEST step any weight takes under the plan (== stepsize
 * unless a weight's bound saturates it).  force_batches != 0 overrides the cut (all ranks of a
 * multi-GPU run must use the same value).  A plan is a list of chunks (consecutive device-order
 * runs of variables, never crossing a colour; n_chunks <= batches * colours -- tiles that learn
 * nothing ride along with a neighbour); with batches == 1 the update is applied once after
 * the last chunk, otherwise after every chunk.
 *   dwx_sgd_plan -> n_chunks;  for c in chunks: dwx_sgd_accumulate_async(c) [+ collective,
 *   + dwx_sgd_apply_async where due];  dwx_sgd_finish. */
int dwx_sgd_plan(dwx_sampler *s, double stepsize, uint32_t force_batches, uint32_t *batches,
                 uint32_t *n_chunks, double *effective_stepsize);
/* The curvature estimate R(batches) dwx_sgd_plan works with (cached per batch count).  A
 * multi-GPU driver takes the maximum over ranks once per batch count and can then derive
 * the common plan for any step size without a per-sweep agreement round. */
int dwx_sgd_curvature(dwx_sampler *s, uint32_t batches, double *lambda);
/* Multi-GPU only: make the current split plan's static-count table n_rows tall (rows
 * beyond this rank's own chunks are zero), so that all ranks can sum equally sized tables
 * and every rank can apply the update of a chunk it idles through. */
int dwx_sgd_plan_rows(dwx_sampler *s, uint32_t n_rows);
/* Multi-GPU only: run the CURRENT split plan with per-record gradient atomics and dynamic
 * update counts (the T half of DWX_BUF_GRAD) although this rank has per-chunk tables --
 * because some other rank has none (its chunk count exceeds the table limit) and all ranks
 * must put the same vector through the collective.  No effect on an un-split plan; reset by
 * the next dwx_sgd_plan. */
int dwx_sgd_plan_force_dynamic(dwx_sampler *s, int on);
/* chunk_range[2 * n_chunks]: chunk c covers positions [chunk_range[2c], chunk_range[2c+1]) of
 * the schedule order (dwx_graph_get_schedule).  The chunks of a split plan go round-robin over
 * the colour launches (the reference's id-order scan meets the colours interleaved), so their
 * ranges are not ascending. */
int dwx_sgd_get_chunks(dwx_sampler *s, uint64_t *chunk_range);
int dwx_sgd_accumulate_async(dwx_sampler *s, uint32_t chunk);
int dwx_sgd_apply_async(dwx_sampler *s);
int dwx_sgd_finish(dwx_sampler *s);
/* Multi-GPU (shards), between dwx_sgd_accumulate_async and dwx_sgd_apply_async, when every rank's
 * dwx_graph_info.grad_shift allows it: the gradient sums G[0, W) of DWX_BUF_GRAD as 32-bit counts
 * (int32)(G[i] >> shift) in a device buffer of the library (*dev32, *n elements) for the caller's
 * all-reduce -- half the bytes of the int64 vector -- and back, G[i] = (int64)count << shift.
 * Replaces nothing in the reference (its replicas average weights once per epoch,
 * src/dimmwitted.cc:209-216); dwx_wait fails if a sum was not a multiple of 2^shift or did not fit. */
int dwx_grad_pack32_async(dwx_sampler *s, uint32_t shift, void **dev32, uint64_t *n);
int dwx_grad_unpack32_async(dwx_sampler *s, uint32_t shift);
/* The general form: bits = 32 as above, or bits = 16 -- two counts per 32-bit word, word i =
 * c[2i] + 65536 c[2i+1] as arithmetic, *n_words = ceil(W / 2): the all-reduce's 32-bit sums of the
 * words are the packed sums of the counts while every SUMMED count stays inside (-2^15, 2^15), i.e.
 * while (sum over ranks of max_records_per_weight) x grad_unit_max < 2^15 -- a quarter of the int64
 * bytes (config 5a: 2 MB per mini-batch at W = 1 M).  RCCL has no 16-bit integer type, hence words. */
int dwx_grad_pack_async(dwx_sampler *s, uint32_t shift, uint32_t bits, void **dev32, uint64_t *n_words);
int dwx_grad_unpack_async(dwx_sampler *s, uint32_t shift, uint32_t bits);

/* infrs.weight_values access (src/dimmwitted.cc:209-216 merge/average, :245-258 dump) */
int dwx_get_weights(dwx_sampler *s, double *out);
int dwx_set_weights(dwx_sampler *s, const double *in);
/* Replica averaging, the reference's n_datacopy semantics (DimmWitted::update_weights +
 * copy_weights_to, src/dimmwitted.cc:209-216,199; InferenceResult::merge_weights_from /
 * average_weights, src/inference_result.cc:68-79): every replica (one sampler per GPU on
 * the WHOLE graph, own seed) runs dwx_sample_sgd_async; the driver then sums
 * DWX_BUF_WEIGHTS over the replicas in place (RCCL all-reduce on dwx_stream) and calls this
 * on every replica: weights /= n_replicas (fixed weights are restored verbatim), f32
 * sampling copy refreshed.  Replaces SURVEY.md 8(b)'s dwx_allreduce_weights: the
 * collective itself stays with the caller's communicator. */
int dwx_average_weights_async(dwx_sampler *s, uint32_t n_replicas);
/* InferenceResult::clear_variabletally / sample_tallies + agg_nsamples
 * (src/inference_result.cc:107-127); arrays are in the REFERENCE numbering. */
int dwx_clear_tallies(dwx_sampler *s);
int dwx_get_tallies(dwx_sampler *s, uint64_t *tallies, uint64_t *nsamples);
/* assignments_free (chain 0) / assignments_evid (chain 1), original variable order */
int dwx_get_assignments(dwx_sampler *s, int chain, uint64_t *out);
int dwx_set_assignments(dwx_sampler *s, int chain, const uint64_t *in);
/* sweep counter (the Philox counter word); starts at 0, +1 per sweep */
int dwx_get_sweep(dwx_sampler *s, uint64_t *out);
int dwx_set_sweep(dwx_sampler *s, uint64_t sweep);

/* Raw device buffers, so the caller's collective library (RCCL via
 * torch.distributed or rccl.h) can reduce / exchange in place. */
enum {
  DWX_BUF_WEIGHTS = 0,      /* double[W]                                            */
  DWX_BUF_GRAD = 1,         /* int64[2W]: fixed-point (2^-30) gradient sums G[W],
                               then update counts T[W]                             */
  DWX_BUF_ASSIGN_FREE = 2,  /* uint32[V] in device order                           */
  DWX_BUF_ASSIGN_EVID = 3,  /* uint32[V] in device order                           */
  DWX_BUF_TALLIES = 4,      /* uint32[num_values] in device order                  */
  DWX_BUF_TSTATIC = 5,      /* int64[2W]: per-sweep update counts T[W] of boolean variables
                               (fixed point 2^-30), then the curvature bounds h[W] of
                               all variables (fixed point 2^-10); static; a multi-GPU
                               driver sums it across shards once after create      */
  DWX_BUF_TSTATIC_PLAN = 6, /* int64[rows][2W]: the same pair per chunk of the current
                               SPLIT plan (valid between dwx_sgd_plan and dwx_sgd_finish;
                               null / 0 bytes when the plan counts dynamically).  A
                               multi-GPU driver calls dwx_sgd_plan_rows(n_chunks of the
                               slowest rank) and sums the table across shards ONCE per
                               batch count (the library caches it per batch count)   */
  DWX_BUF_SORTED_RECORDS = 7,      /* test hooks: the weight-sorted record copy of the graph's default  */
  DWX_BUF_SORTED_RECORDS_PLAN = 8  /* layout / of the current plan level's own (8 bytes per record;
                                      null / 0 bytes when there is none): the device build
                                      (device_build.hip) is checked against the host builder's bytes  */
};
int dwx_device_buffer(dwx_sampler *s, int which, void **dev_ptr, uint64_t *nbytes);
/* Copy between host memory and a device pointer obtained from dwx_device_buffer /
 * dwx_halo_buffer (to_device != 0: host -> device), ordered after everything already queued
 * on the sampler's stream; returns when the copy is done.  For drivers without a device
 * runtime of their own (the host-staged test communicator of `dw gibbs --comm host`). */
int dwx_buffer_copy(dwx_sampler *s, void *dst, const void *src, uint64_t nbytes, int to_device);
/* Halo exchange support (multi-GPU with factors that cross the shard boundary; SURVEY.md 8e:
 * "exchange halo assignments ... grouped in one ncclGroupStart/End").  A halo list is one side
 * of the exchange with one peer: `vids` = local variable ids (owned variables a peer ghosts:
 * a SEND list; ghost variables a peer owns: a RECEIVE list, dwx_graph_desc.num_ghost_variables)
 * and a device buffer for both chains.  dwx_halo_pack_async gathers the listed variables'
 * assignments of the selected chains (bit 0: free chain, bit 1: evidence chain; selected
 * chains back to back, see dwx_halo_message_bytes) into the buffer; the caller moves buffers between
 * ranks with its collective library on dwx_stream (ncclSend / ncclRecv); dwx_halo_unpack_async
 * scatters a received buffer into the ghosts.  All on the sampler's stream, no host
 * synchronisation.  (No reference counterpart: the reference's threads share one address
 * space and read each other's assignments directly, src/gibbs_sampler.h:198-215.) */
typedef struct dwx_halo dwx_halo;
int dwx_halo_create(dwx_sampler *s, const uint64_t *vids, uint64_t n, dwx_halo **out);
void dwx_halo_destroy(dwx_halo *h);
int dwx_halo_buffer(dwx_halo *h, void **dev_ptr, uint64_t *nbytes);
/* Bytes at the start of the buffer that dwx_halo_pack_async(h, chains) fills and
 * dwx_halo_unpack_async(h, chains) reads -- what to send / receive.  Values travel as 1 bit
 * (every listed variable boolean), 8 bits (every listed cardinality <= 256) or 32 bits each,
 * the selected chains' blocks back to back, each padded to 8 bytes; the peer's list of the same
 * variables gives the same number. */
int dwx_halo_message_bytes(dwx_halo *h, int chains, uint64_t *nbytes);
int dwx_halo_pack_async(dwx_halo *h, int chains);
int dwx_halo_unpack_async(dwx_halo *h, int chains);

/* The HIP stream (hipStream_t) the sampler enqueues on. */
int dwx_stream(dwx_sampler *s, void **stream);

/* Device time of the sweep kernels only, measured with HIP events on the sampler's
 * stream around every launch since the last reset: total milliseconds, number of
 * kernel launches and number of sweeps (a learning sweep of several colours or
 * mini-batches is ONE sweep of several launches).  kind: 0 = inference sweep kernels, 1 = learning
 * sweep kernels, 2 = the pull-based gradient kernel of learning sweeps, 3 = no time: launches =
 * sweeps = the split learning sweeps dwx_sample_sgd_async handed over as ONE graph launch since
 * the sampler was created (DWX_GRAPH=n in the environment: a sweep of >= n mini-batches is
 * captured and replayed as a hipGraph from the plan level's second sweep on, never while timing
 * is enabled; off by default -- measured, it buys nothing: DESIGN.md section 3.5); 4 = no time:
 * launches = sweeps = the split learning sweeps that ran as ONE persistent launch (DWX_PERSIST=1 in the
 * environment at dwx_sampler_create; all-unary graphs with at most 128 weights, >= 8 mini-batches:
 * chunk loop, grid barrier and update inside the kernel; off by default -- measured, it is slower
 * than the launches it replaces: sampler_amd/csrc/persist_kernels.h); 5 = no time: launches = sweeps =
 * the split learning sweeps that ran with ONE launch per mini-batch -- the update of a mini-batch as the
 * prologue of the next one's sweep kernel (all-unary graphs with at most 1024 weights and no
 * degree-binned variable; the default for them, DWX_NO_MERGED_APPLY=1 at dwx_sampler_create disables it). */
int dwx_kernel_time(dwx_sampler *s, int kind, double *ms, uint64_t *launches, uint64_t *sweeps);
int dwx_kernel_time_reset(dwx_sampler *s, int enable);

/* Test hook: evaluate one factor function on the device.  sat[i] = whether the
 * predicate of position i holds; mirrors test/factor_test.cc's truth tables. */
int dwx_test_factor_sign(int device, int func, uint64_t arity, const uint8_t *sat, double *out);
/* Test hook: the device's random number generator (the reference draws with erand48,
 * src/gibbs_sampler.h:177,204,230; the device with the counter-based Philox4x32-10 of Salmon
 * et al., SC'11).  out = philox(key, ctr), the raw block function (Random123's known-answer
 * vectors apply); uniforms = the two 53-bit uniforms the sweep kernels draw for
 * (seed = key, variable id = ctr[0..1], sweep = ctr[2..3]). */
int dwx_test_philox(int device, const uint32_t key[2], const uint32_t ctr[4], uint32_t out[4],
                    double uniforms[2]);

#ifdef __cplusplus
}
#endif
#endif /* DWX_H_ */
