/*
 * dw_oracle.h -- C API of the CPU ORACLE for the DimmWitted Gibbs-sweep hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load liboracle.so.  Nothing under sampler_amd/
 * includes, links or calls anything in this directory.
 *
 * Parity status: PINNED.  In reference mode (erand48 RNG, id-order scan) this
 * restatement reproduces byte-for-byte the output files of the real reference
 * (`dw gibbs -t 1 -c 1`, built by oracle/build_ref.sh) on all seven fixtures of
 * /root/reference/test (tests/golden/, tests/test_oracle_golden.py) and the
 * gtest known answers of test/sampler_test.cc, test/factor_graph_test.cc and
 * test/factor_test.cc.
 */
#ifndef DW_ORACLE_H_
#define DW_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Columnar image of the reference's binary input files (doc/binary_format.md).
 * Field-for-field identical to dwx_graph_desc in include/dwx.h so one ctypes
 * structure feeds both sides; the two headers do not include each other. */
typedef struct orc_graph_desc {
  uint64_t num_variables, num_factors, num_edges, num_weights;
  const uint8_t *var_role;        /* isEvidence byte; evidence iff >= 1      */
  const uint64_t *var_init_value; /* initialValue as in the file             */
  const uint16_t *var_dtype;      /* 0 boolean, 1 categorical                */
  const uint64_t *var_cardinality;
  uint64_t num_domains;           /* number of domain blocks (may be 0)      */
  const uint64_t *dom_vid;        /* [num_domains]                           */
  const uint64_t *dom_offset;     /* [num_domains+1] into dom_value/truth    */
  const uint64_t *dom_value;
  const double *dom_truthiness;
  const uint16_t *fac_func;       /* FACTOR_FUNCTION_TYPE                    */
  const uint64_t *fac_edge_offset;/* [num_factors+1]                         */
  const uint64_t *fac_weight_id;
  const double *fac_feature_value;
  const uint64_t *edge_vid;
  const uint64_t *edge_equal_to;  /* equalPredicate as in the file           */
  const double *w_initial_value;  /* indexed by weight id                    */
  const uint8_t *w_is_fixed;
  /* Sharding: the LAST num_ghost_variables variables are ghosts -- remote variables
   * that local factors read.  They hold an assignment (refreshed by the caller's halo
   * exchange) but are never sampled, tallied or given rows.  0 for a whole graph. */
  uint64_t num_ghost_variables;
} orc_graph_desc;

typedef struct orc_opts {
  int32_t sample_evidence;
  int32_t learn_non_evidence;
  int32_t noise_aware;
  int32_t regularization; /* 0 = L1, 1 = L2 (reference enum order REG_L1, REG_L2) */
  double reg_param;
} orc_opts;

typedef struct orc_sampler orc_sampler;

/* Load-time conversions + construct_index + InferenceResult init. NULL on error. */
orc_sampler *orc_create(const orc_graph_desc *desc, const orc_opts *opts);
void orc_destroy(orc_sampler *s);
const char *orc_last_error(void);

/* ---- sizes / state access (pointers stay valid until orc_destroy) ---- */
uint64_t orc_num_values(const orc_sampler *s);
uint64_t orc_num_index_entries(const orc_sampler *s);      /* |factor_index| after dedup */
double *orc_weights(orc_sampler *s);                        /* weight_values[W]  */
uint64_t *orc_assignments(orc_sampler *s, int chain);       /* 0 = free, 1 = evid */
uint64_t *orc_tallies(orc_sampler *s);                      /* sample_tallies[Val] */
uint64_t *orc_nsamples(orc_sampler *s);                     /* agg_nsamples[V]   */
const uint64_t *orc_var_val_base(const orc_sampler *s);     /* [V]               */
const uint64_t *orc_value_sparse(const orc_sampler *s);     /* values[].value    */
const uint64_t *orc_value_index_base(const orc_sampler *s); /* values[].factor_index_base */
const uint64_t *orc_value_index_len(const orc_sampler *s);  /* values[].factor_index_length */
const uint64_t *orc_factor_index(const orc_sampler *s);
const uint64_t *orc_var_assignment_dense(const orc_sampler *s);
void orc_clear_tallies(orc_sampler *s);

/* ---- reference mode: erand48 RNG, contiguous id-range shards ---- */
/* (Re)create n_workers GibbsSamplerThread-equivalents; seeds are taken from a
 * private glibc random_r stream seeded with 1 == un-seeded rand(), three draws
 * per worker in construction order (src/gibbs_sampler.cc:49). */
void orc_ref_set_workers(orc_sampler *s, uint32_t n_workers);
void orc_ref_set_seed(orc_sampler *s, uint32_t worker, uint16_t s0, uint16_t s1, uint16_t s2);
void orc_ref_sample_single_variable(orc_sampler *s, uint32_t worker, uint64_t vid);
void orc_ref_sample_sgd_single_variable(orc_sampler *s, uint32_t worker, uint64_t vid,
                                        double stepsize);
void orc_ref_sgd_on_variable(orc_sampler *s, uint64_t vid, double stepsize);
/* one epoch over all shards; threaded != 0 runs the workers as std::threads
 * (Hogwild, like the reference), else sequentially in worker order. */
void orc_ref_sample(orc_sampler *s, int threaded);
void orc_ref_sample_sgd(orc_sampler *s, double stepsize, int threaded);
/* DimmWitted::learn + inference with n_datacopy = 1 (src/dimmwitted.cc:121-207) */
void orc_ref_learn(orc_sampler *s, uint64_t n_epoch, double stepsize, double decay, int threaded);
void orc_ref_inference(orc_sampler *s, uint64_t n_epoch, int threaded);

/* potential(variable, proposal) on a chain with the current weights */
double orc_potential(orc_sampler *s, uint64_t vid, uint64_t proposal, int chain);
/* truth-table hook: sign of one factor function over explicit satisfied-bits
 * (sat[i] = whether position i's predicate holds). */
double orc_factor_sign(int func, uint64_t arity, const uint8_t *sat);
double orc_logadd(double a, double b);
double orc_erand48(uint16_t xsubi[3]);

/* ---- schedule mode: mirrors the device semantics ----
 * Counter-based Philox4x32-10 keyed by seed, counter (vid, sweep); variables are
 * visited launch by launch in the given order (each launch must be an independent
 * set); learning accumulates fixed-point gradient sums per weight and applies
 * them once per sweep. */
typedef struct orc_schedule {
  uint64_t n_order;            /* number of scheduled variables (== V)          */
  const uint64_t *order;       /* original variable ids in execution order      */
  uint64_t n_launches;
  const uint64_t *launch_off;  /* [n_launches+1] offsets into order             */
} orc_schedule;

void orc_sched_sample(orc_sampler *s, const orc_schedule *sch, uint64_t seed, uint64_t sweep);
void orc_sched_sample_sgd(orc_sampler *s, const orc_schedule *sch, uint64_t seed, uint64_t sweep,
                          double stepsize);
/* the two halves of orc_sched_sample_sgd: sample + accumulate G, T and the curvature bounds
 * H of the visited variables; then the saturating update (see orc_sched_apply) + clear */
void orc_sched_accumulate(orc_sampler *s, const orc_schedule *sch, uint64_t seed, uint64_t sweep);
void orc_sched_apply(orc_sampler *s, double stepsize);
/* the same with explicit curvature bounds (int64[W], fixed point 2^-10) instead of the
 * accumulated ones; orc_sched_curvature computes them for a list of variables */
void orc_sched_apply_h(orc_sampler *s, double stepsize, const int64_t *hess);
void orc_sched_curvature(orc_sampler *s, const orc_schedule *sch, int64_t *out);
/* int64[3W]: fixed-point gradient sums G[W], update counts T[W], curvature bounds H[W] */
int64_t *orc_grad(orc_sampler *s);
/* global id of local variable 0: added to ids in the Philox counter (shards) */
void orc_set_var_id_offset(orc_sampler *s, uint64_t off);
/* device semantics: FactorGraph::potential multiplies with the weight rounded to f32
 * (the device's L2-resident sampling copy); off in reference mode */
void orc_set_sampling_weight_f32(orc_sampler *s, int on);
/* schedule mode: mask[V] != 0 where the device sums a boolean variable's potentials in fixed
 * point (dwx_graph_get_fixed_point_mask); null clears it */
void orc_set_fixed_point_mask(orc_sampler *s, const uint8_t *mask);
/* returns 1 if every launch of the schedule is an independent set */
int orc_sched_check_independent(orc_sampler *s, const orc_schedule *sch);
/* two uniforms in [0,1) from Philox4x32-10 (test hook) */
/* Philox4x32-10 block function in place: ctr <- philox(key, ctr) (Random123 KATs) */
void orc_philox4x32_10(const uint32_t key[2], uint32_t ctr[4]);
void orc_philox_uniforms(uint64_t seed, uint64_t vid, uint64_t sweep, double out[2]);

#ifdef __cplusplus
}
#endif
#endif /* DW_ORACLE_H_ */
