// tile_walk.h -- what one variable does once its tile is staged: the row walks (potentials in
// the reference's order), the boolean / categorical draws, sample_sgd_single_variable with its
// gradient rows, in every weight-placement mode (W_*) -- lane, wave and workgroup walks share this
// code.  Part of sweep_kernels.h.
#ifndef DWX_TILE_WALK_H_
#define DWX_TILE_WALK_H_

#include "factor_functions.h"

namespace dwx {

// ---------------------------------------------------------------- tile view
// Where a lane reads its row pointers / edge records / weights / potential scratch
// from: the LDS-staged tile (normal) or HBM directly (a variable too big for a tile).
// WMODE says where the weight of an edge record lives:
//   W_GLOBAL   gather w32[rec.wid] from memory (oversized variables only)
//   W_ARRAY    staged f32 array parallel to the staged records (learning kernel:
//              the records keep their weight id for the gradient scatter)
//   W_INRECORD the staging pass overwrote rec.wid with the f32 weight bits
//              (inference kernel: no extra LDS)
//   W_TERMS    (inference, SIMPLE tiles) the staging pass replaced each record by its
//              two potential terms (w*(s1*f), w*(s0*f)); the row walk only adds
//   W_TERMS8   (inference on the 8-byte terms table) the table's entries as they are: w * f with
//              the two sign codes in its lowest mantissa bits; the row walk decodes and adds
//   W_COOP     (wide_kernel) a whole WAVE walks one variable straight from HBM: lane l takes
//              records l, l + 64, ... of a row, the 64 partial sums are combined by a butterfly
//              (all lanes get the total); every lane then follows the same decisions, side
//              effects happen once
//   W_COOPB    (giant_kernel) the same with a whole WORKGROUP of GIANT_THREADS lanes and an
//              LDS tree for the sums
//   W_PRESUM   (giant_decide_kernel) the potentials of a boolean variable are already summed
//              (TileView::presum); one lane decides, the gradient rows are walked elsewhere
//   W_LREC     (learning, categorical TILE_TERMS3 tiles) the staged records are LearnRecs: weight and
//              the four products per record come out of LDS, for the draws and the gradient alike
enum { W_GLOBAL = 0, W_ARRAY = 1, W_INRECORD = 2, W_TERMS = 3, W_TERMS8 = 4, W_COOP = 5, W_COOPB = 6, W_PRESUM = 7, W_LREC = 8, W_FIXSUM = 9 };
#ifndef DWX_GIANT_PIECE
#define DWX_GIANT_PIECE 4096   // (one batched step of 4 records per lane: half the latency of 8192, hub graph learning 16.8 -> 13.5 ms)
#endif
constexpr uint32_t GIANT_PIECE = DWX_GIANT_PIECE;     // records per workgroup of a boolean oversized variable
#ifndef DWX_GIANT_THREADS
#define DWX_GIANT_THREADS 1024
#endif
constexpr uint32_t GIANT_THREADS = DWX_GIANT_THREADS;   // lanes per oversized variable (giant kernels)
constexpr uint32_t COOP_U = 4;             // records per lane and step of a cooperative walk

DWX_DEV uint32_t wave_lane() { return threadIdx.x & 63u; }

// sum over all lanes of the workgroup through an LDS tree (every lane gets the total; must be
// reached by every lane of the workgroup)
DWX_DEV double block_sum_all(double v) {
  __shared__ double s_red[GIANT_THREADS];
  const uint32_t t = threadIdx.x;
  s_red[t] = v;
  __syncthreads();
  for (uint32_t s = blockDim.x / 2; s > 0; s >>= 1) {
    if (t < s) s_red[t] += s_red[t + s];
    __syncthreads();
  }
  const double r = s_red[0];
  __syncthreads();
  return r;
}

// the cooperating group of a W_COOP / W_COOPB walk
template <int WMODE>
struct Coop {
  static constexpr bool on = WMODE == W_COOP || WMODE == W_COOPB;
  static DWX_DEV uint32_t lane() { return WMODE == W_COOPB ? threadIdx.x : wave_lane(); }
  static DWX_DEV uint32_t stride() { return WMODE == W_COOPB ? blockDim.x : 64u; }
  static DWX_DEV double sum(double v) { return WMODE == W_COOPB ? block_sum_all(v) : DWX_WAVE_SUM_F64(v); }
};

struct alignas(16) EdgeTerms { double t1, t0; };
// Staged form of a record of a TILE_TERMS2 / TILE_TERMS3 tile in a learning sweep: the four
// sign * feature_value products (free / evidence chain x the owner's proposal "hits" / "misses"
// -- 1 / 0 for a boolean owner, the row's value / any other for a categorical one), evaluated
// edge-parallel in the staging pass; exact in f32 (signs of factors of arity <= 3 are small
// integers -- RATIO at arity 3 is kept out -- and such a tile only holds f32-exact feature values).
struct alignas(16) LearnRec {
  uint32_t wid, packed;
  float w, sf1, sf0, se1, se0;
  uint32_t pad;
};
static_assert(sizeof(LearnRec) == 32, "LearnRec must be 32 bytes");
// The same in 16 bytes, for builds whose staged tiles hold only pre-signed and arity-2 records (TV_PAIR,
// config 3b / 5b): a binary sign is -1, 0 or +1, so the four products of an arity-2 record are its feature
// value `a` times four 2-bit codes (sign + 1: free chain proposal 1 / 0 in bits 0-1 / 2-3, evidence chain in
// 4-5 / 6-7 of the word in `b`); a pre-signed record keeps its two terms in a (hit) and b (miss).  Half the
// LDS of LearnRec: four workgroups per CU instead of three (round 4).  wf: weight id | flags (W < 2^30).
struct alignas(16) LearnRec16 {
  uint32_t wf;
  float w, a, b;
};
static_assert(sizeof(LearnRec16) == 16, "LearnRec16 must be 16 bytes");
constexpr uint32_t LR16_PRESIGNED = 1u << 31, LR16_FIXED = 1u << 30, LR16_WID_MASK = (1u << 30) - 1u;
// one accessor set over both forms (chain: 0 free / 1 evidence; value: the owner's proposal)
DWX_DEV float lrec_term(const LearnRec &r, uint32_t chain, uint32_t value) {
  return chain ? (value ? r.se1 : r.se0) : (value ? r.sf1 : r.sf0);
}
DWX_DEV uint32_t lrec_wid(const LearnRec &r) { return r.wid; }
DWX_DEV bool lrec_fixed(const LearnRec &r) { return r.packed & EDGE_FIXED_FLAG; }
DWX_DEV bool lrec_presigned(const LearnRec &r) { return r.packed & EDGE_PRESIGNED; }
DWX_DEV float lrec_term(const LearnRec16 &r, uint32_t chain, uint32_t value) {
  uint32_t codes;
  __builtin_memcpy(&codes, &r.b, 4);
  const int sign = (int)((codes >> (chain * 4u + (value ? 0u : 2u))) & 3u) - 1;
  const float binary = (float)sign * r.a;      // (a pre-signed record's `b` is a float: this value is then unused)
  return (r.wf & LR16_PRESIGNED) ? (value ? r.a : r.b) : binary;
}
DWX_DEV uint32_t lrec_wid(const LearnRec16 &r) { return r.wf & LR16_WID_MASK; }
DWX_DEV bool lrec_fixed(const LearnRec16 &r) { return r.wf & LR16_FIXED; }
DWX_DEV bool lrec_presigned(const LearnRec16 &r) { return r.wf & LR16_PRESIGNED; }
// Table entry of a record of a TILE_INLINE2 tile (build_terms_kernel), overlaying an EdgeRec:
// wf = w * |f|-signed product (f64 of two f32: exact), `other` = device position of the other
// endpoint (the owner's for a unary record), bits: func id in 0-3; unary: TAB2_UNARY, TAB2_C1
// (t1 = wf, else 0) and c0 + 1 in two bits (t0 = c0 * wf); arity 2: the INLINE2 field.
struct alignas(16) TabRec2 { double wf; uint32_t other; uint32_t bits; };
constexpr uint32_t TAB2_UNARY = 1u << 4, TAB2_C1 = 1u << 5, TAB2_C0_SHIFT = 6;
DWX_DEV double u32x2_to_double(uint32_t lo, uint32_t hi) {
  const unsigned long long u = (unsigned long long)lo | ((unsigned long long)hi << 32);
  double d; __builtin_memcpy(&d, &u, 8); return d;
}
static_assert(sizeof(EdgeTerms) == sizeof(EdgeRec), "terms overlay the staged records");

DWX_DEV uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
#ifndef DWX_WALK_BATCH
#define DWX_WALK_BATCH 5
#endif
constexpr uint32_t WALK_BATCH = DWX_WALK_BATCH;   // staged terms read per step of a row walk
#ifndef DWX_LEARN_BATCH
#define DWX_LEARN_BATCH 4
#endif
constexpr uint32_t LEARN_BATCH = DWX_LEARN_BATCH; // staged 32-byte learning records read per step
#ifndef DWX_CHAIN_PAIRS
#define DWX_CHAIN_PAIRS 1   // two lanes per variable in learning sweeps over small TERMS tiles
#endif

// entry of the 8-byte terms table (build_terms8_kernel): the f64 product w * f with sign(hit) + 1
// in bits 0-1 and sign(miss) + 1 in bits 2-3 of its mantissa (always zero in such a product)
DWX_DEV double terms8_pick(unsigned long long u, uint32_t code) {
  const unsigned long long v = u & ~15ull;
  double wf; __builtin_memcpy(&wf, &v, 8);
  return code == 1u ? 0.0 : (code == 0u ? -wf : wf);
}
DWX_DEV double terms8_hit(unsigned long long u) { return terms8_pick(u, (uint32_t)u & 3u); }
DWX_DEV double terms8_miss(unsigned long long u) { return terms8_pick(u, ((uint32_t)u >> 2) & 3u); }

struct TileView {
  const uint32_t *rowptr;  // indexed by (row - row_bias)
  uint32_t row_bias;
  const EdgeRec *edges;    // indexed by (edge - edge_bias)
  uint32_t edge_bias;
  const float *w;          // W_ARRAY: staged weights, indexed like edges
  long long *agg;          // LDS gradient accumulators [2W] (learning, small W) or null
  double *pot;             // per-row potential scratch (row - row_bias), or null
  // W_PRESUM (giant_decide_kernel): the boolean variable's four potentials {free 1, free 0,
  // evidence 1, evidence 0} were summed by other workgroups; its gradient walk is left to them
  // too -- the decision goes here: {evidence value, free value, 1 | count_t << 1, -}
  const double *presum = nullptr;
  uint32_t *decision = nullptr;
};

DWX_DEV uint32_t edge_func(const EdgeRec &e) { return e.packed & EDGE_FUNC_MASK; }
DWX_DEV uint32_t edge_arity(const EdgeRec &e) { return (e.packed >> EDGE_ARITY_SHIFT) & EDGE_ARITY_MASK; }
DWX_DEV uint32_t edge_owner_lane(const EdgeRec &e) { return e.packed >> EDGE_OWNER_SHIFT; }
DWX_DEV float bits_to_float(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }
DWX_DEV uint32_t float_to_bits(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }

// The signs of one generic record (arity >= 2) in NS scenarios over NCHAIN chains: its first
// GEN_ARITY positions are loaded together (entries, then assignments on every chain), wider
// factors walk memory.  chain[j] = which of chains[] scenario j reads.
template <int NS, int NCHAIN>
DWX_DEV void record_signs(const KernelParams &P, const EdgeRec &er, uint32_t me,
                          const uint32_t *const (&chains)[NCHAIN], const int (&chain)[NS],
                          const uint32_t (&prop)[NS], double (&s)[NS]) {
  const uint32_t func = edge_func(er), ar = edge_arity(er);
  if (ar <= GEN_ARITY) {
    VifRec vf[GEN_ARITY];
    uint32_t val[NCHAIN][GEN_ARITY];
#pragma unroll
    for (uint32_t i = 0; i < GEN_ARITY; ++i) vf[i] = P.vifs[er.aux + umin(i, ar - 1u)];
#pragma unroll
    for (int c = 0; c < NCHAIN; ++c)
#pragma unroll
      for (uint32_t i = 0; i < GEN_ARITY; ++i) val[c][i] = chains[c][vf[i].vid];
    const VifsPreloaded<NS, NCHAIN> src{vf, val, chain};
    factor_signs_from<NS>(func, ar, src, me, prop, s);
  } else {
    const uint32_t *arr[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) arr[j] = chains[chain[j]];
    const uint32_t *const (&carr)[NS] = arr;
    const VifsInMemory<NS> src{P.vifs + er.aux, carr};
    factor_signs_from<NS>(func, ar, src, me, prop, s);
  }
}

// A cooperative walk over records [es, ee) (W_COOP / W_COOPB): the group's lanes stride over
// them, COOP_U records per lane and step, in four phases so that every phase's loads -- of all
// COOP_U records -- are in flight together: the records; their weights and the first GEN_ARITY
// factor->variable entries of the non-unary ones; those variables' assignments on every chain;
// then the arithmetic.  A record is evaluated in NS scenarios at once: scenario j = the owner
// takes prop[j] (hit[j]: that value "hits" a pre-signed record), everybody else its value on
// chain chain[j] of chains[].  fn(record, index, weight, term[NS]) gets sign * feature value
// per scenario (Factor::potential, src/factor.h:59-86).  Factors wider than GEN_ARITY walk
// memory as everywhere else.
template <int WMODE, int NS, int NCHAIN, class Fn>
DWX_DEV void coop_for_records(const KernelParams &P, const TileView &T, uint32_t es, uint32_t ee, uint32_t me,
                              const uint32_t *const (&chains)[NCHAIN], const int (&chain)[NS],
                              const uint32_t (&prop)[NS], const bool (&hit)[NS], Fn &&fn) {
  const uint32_t stride = Coop<WMODE>::stride();
  for (uint32_t e0 = es + Coop<WMODE>::lane(); e0 < ee; e0 += stride * COOP_U) {
    EdgeRec er[COOP_U];
    float w[COOP_U];
    VifRec vf[COOP_U][GEN_ARITY];
    uint32_t val[COOP_U][NCHAIN][GEN_ARITY];
#pragma unroll
    for (uint32_t u = 0; u < COOP_U; ++u) {
      const uint32_t e = e0 + u * stride;
      er[u] = T.edges[(e < ee ? e : es) - T.edge_bias];
    }
#pragma unroll
    for (uint32_t u = 0; u < COOP_U; ++u) {
      w[u] = P.w32[er[u].wid];
      const bool generic = !(er[u].packed & EDGE_PRESIGNED);
      const uint32_t ar = generic ? edge_arity(er[u]) : 1u, base = (generic && ar >= 2u) ? er[u].aux : 0u;
#pragma unroll
      for (uint32_t i = 0; i < GEN_ARITY; ++i) vf[u][i] = P.vifs[base + umin(i, ar - 1u)];
    }
#pragma unroll
    for (uint32_t u = 0; u < COOP_U; ++u)
#pragma unroll
      for (int c = 0; c < NCHAIN; ++c)
#pragma unroll
        for (uint32_t i = 0; i < GEN_ARITY; ++i) val[u][c][i] = chains[c][vf[u][i].vid];
#pragma unroll
    for (uint32_t u = 0; u < COOP_U; ++u) {
      const uint32_t e = e0 + u * stride;
      if (e >= ee) continue;
      double term[NS];
      if (er[u].packed & EDGE_PRESIGNED) {
#pragma unroll
        for (int j = 0; j < NS; ++j) term[j] = (double)(hit[j] ? er[u].fval : bits_to_float(er[u].aux));
      } else {
        const double fv = (er[u].packed & EDGE_F64_FLAG) ? P.edge_fval64[e] : (double)er[u].fval;
        const uint32_t func = edge_func(er[u]), ar = edge_arity(er[u]);
        double sg[NS];
        if (ar == 1u) {
#pragma unroll
          for (int j = 0; j < NS; ++j) sg[j] = unary_sign(func, prop[j] == er[u].aux);
        } else if (ar <= GEN_ARITY) {
          const VifsPreloaded<NS, NCHAIN> src{vf[u], val[u], chain};
          factor_signs_from<NS>(func, ar, src, me, prop, sg);
        } else {
          const uint32_t *arr[NS];
#pragma unroll
          for (int j = 0; j < NS; ++j) arr[j] = chains[chain[j]];
          const uint32_t *const (&carr)[NS] = arr;
          factor_signs<NS>(func, ar, er[u].aux, P.vifs, me, carr, prop, sg);
        }
#pragma unroll
        for (int j = 0; j < NS; ++j) term[j] = sg[j] * fv;
      }
      fn(er[u], e, (double)w[u], term);
    }
  }
}

// SIMPLE (a per-tile, workgroup-uniform property, TILE_SIMPLE): every record is a
// unary factor with an f32-exact feature value.  The SIMPLE variants below contain no
// global load, so nothing in the compute phase waits on vmcnt -- which retires in
// order and would otherwise also wait for the next tile's prefetch.
// sign * feature_value of one record for `proposal` (= Factor::potential,
// src/factor.h:59-86).  `hit`: boolean owner -> proposal == 1; categorical owner ->
// proposal == value of the record's row.  Pre-signed records need nothing else.
template <bool SIMPLE>
DWX_DEV double edge_term(const KernelParams &P, const EdgeRec &er, uint32_t idx,
                         const uint32_t *assign, uint32_t me, uint32_t proposal, bool hit) {
  if (SIMPLE || (er.packed & EDGE_PRESIGNED))
    return (double)(hit ? er.fval : bits_to_float(er.aux));
  const double fv = (er.packed & EDGE_F64_FLAG) ? P.edge_fval64[idx] : (double)er.fval;
  if (edge_arity(er) == 1u) return unary_sign(edge_func(er), proposal == er.aux) * fv;
  // (the factor's positions are loaded together, not one dependent load after the other)
  const uint32_t *const chains[1] = {assign};
  const int chain[1] = {0};
  const uint32_t prop[1] = {proposal};
  double sg[1];
  record_signs<1, 1>(P, er, me, chains, chain, prop, sg);
  return sg[0] * fv;
}

template <int WMODE>
DWX_DEV double edge_weight(const KernelParams &P, const TileView &T, const EdgeRec &er, uint32_t e) {
  if (WMODE == W_INRECORD || WMODE == W_TERMS) return (double)bits_to_float(er.wid);
  if (WMODE == W_ARRAY) return (double)T.w[e - T.edge_bias];
  return (double)P.w32[er.wid];
}

// FactorGraph::potential for one value row (src/factor_graph.h:127-145):
// pot = sum_i weight[wid_i] * (sign_i * feature_value_i), in row order; [es, ee) is the
// row's record range.
template <int WMODE, bool SIMPLE>
DWX_DEV double range_potential(const KernelParams &P, const TileView &T, uint32_t es, uint32_t ee,
                               const uint32_t *assign, uint32_t me, uint32_t proposal) {
  double pot = 0.0;
  if (Coop<WMODE>::on) {
    const uint32_t *const chains[1] = {assign};
    const int chain[1] = {0};
    const uint32_t prop[1] = {proposal};
    const bool hit[1] = {true};
    coop_for_records<WMODE, 1, 1>(P, T, es, ee, me, chains, chain, prop, hit,
                                  [&](const EdgeRec &, uint32_t, double w, const double (&term)[1]) { pot += w * term[0]; });
    return Coop<WMODE>::sum(pot);
  }
  if (WMODE == W_LREC) {   // (the row's own value is proposed: every record "hits")
    const LearnRec *recs = (const LearnRec *)T.edges;
    const bool evid = assign == P.assign_evid;
    for (uint32_t e = es; e < ee; e += LEARN_BATCH) {
      LearnRec r[LEARN_BATCH];
#pragma unroll
      for (uint32_t u = 0; u < LEARN_BATCH; ++u) r[u] = recs[umin(e + u, ee - 1) - T.edge_bias];
#pragma unroll
      for (uint32_t u = 0; u < LEARN_BATCH; ++u)
        pot += (e + u < ee) ? (double)r[u].w * (double)(evid ? r[u].se1 : r[u].sf1) : 0.0;
    }
    return pot;
  }
  if (WMODE == W_TERMS8) {
    const unsigned long long *tab = (const unsigned long long *)T.edges;
    if (ee - es == 1u) return pot + terms8_hit(tab[es - T.edge_bias]);
    for (uint32_t e = es; e < ee; e += WALK_BATCH) {
      unsigned long long u[WALK_BATCH];
#pragma unroll
      for (uint32_t k = 0; k < WALK_BATCH; ++k) u[k] = tab[umin(e + k, ee - 1) - T.edge_bias];
#pragma unroll
      for (uint32_t k = 0; k < WALK_BATCH; ++k) pot += (e + k < ee) ? terms8_hit(u[k]) : 0.0;
    }
    return pot;
  }
  if (WMODE == W_TERMS) {
    // WALK_BATCH LDS reads in flight per step instead of one dependent read per record; the
    // additions stay sequential and in row order (a slot past the end re-reads the last
    // record and adds +0.0, which changes nothing: a running sum is never -0.0)
    const EdgeTerms *terms = (const EdgeTerms *)T.edges;
    if (ee - es == 1u) return pot + terms[es - T.edge_bias].t1;   // (one read; the usual categorical row)
    for (uint32_t e = es; e < ee; e += WALK_BATCH) {
      double t1[WALK_BATCH];
#pragma unroll
      for (uint32_t u = 0; u < WALK_BATCH; ++u) t1[u] = terms[umin(e + u, ee - 1) - T.edge_bias].t1;
#pragma unroll
      for (uint32_t u = 0; u < WALK_BATCH; ++u) pot += (e + u < ee) ? t1[u] : 0.0;
    }
    return pot;
  }
  for (uint32_t e = es; e < ee; ++e) {
    const EdgeRec er = T.edges[e - T.edge_bias];
    const double w = edge_weight<WMODE>(P, T, er, e);
    pot += w * edge_term<SIMPLE>(P, er, e, assign, me, proposal, true);
  }
  return pot;
}

template <int WMODE, bool SIMPLE>
DWX_DEV double row_potential(const KernelParams &P, const TileView &T, uint32_t row,
                             const uint32_t *assign, uint32_t me, uint32_t proposal) {
  return range_potential<WMODE, SIMPLE>(P, T, T.rowptr[row - T.row_bias], T.rowptr[row + 1 - T.row_bias],
                                        assign, me, proposal);
}

// boolean variable: both proposals in one pass over the row (same sums, same order
// as two calls of FactorGraph::potential, src/gibbs_sampler.h:201-202)
// FIXED (all-unary graphs, compact records): pp - pn as ONE fixed-point sum of the records'
// t1 - t0 (pot_fix, factor_functions.h); returned as pp = that sum, pn = 0.
template <int WMODE, bool SIMPLE, bool FIXED = false>
DWX_DEV void bool_potentials(const KernelParams &P, const TileView &T, uint32_t row,
                             const uint32_t *assign, uint32_t me, double &pp, double &pn) {
  if (WMODE == W_FIXSUM) {   // sorted_sweep_kernel: pp - pn, summed in fixed point over the weight-sorted records
    pp = T.presum[0]; pn = 0.0;     // (no rows, no records: the view holds the sum only)
    return;
  }
  const uint32_t es = T.rowptr[row - T.row_bias], ee = T.rowptr[row + 1 - T.row_bias];
  pp = 0.0; pn = 0.0;
  static_assert(!FIXED || (SIMPLE && (WMODE == W_TERMS8 || WMODE == W_TERMS || WMODE == W_ARRAY)),
                "fixed-point potentials: staged all-unary tiles only");
  if (FIXED) {
    long long acc = 0;
    if (WMODE == W_TERMS8) {
      const unsigned long long *tab = (const unsigned long long *)T.edges;
      for (uint32_t e = es; e < ee; e += WALK_BATCH) {
        unsigned long long u[WALK_BATCH];
#pragma unroll
        for (uint32_t k = 0; k < WALK_BATCH; ++k) u[k] = tab[umin(e + k, ee - 1) - T.edge_bias];
#pragma unroll
        for (uint32_t k = 0; k < WALK_BATCH; ++k)
          acc += (e + k < ee) ? pot_fix(terms8_hit(u[k]) - terms8_miss(u[k])) : 0ll;
      }
    } else if (WMODE == W_TERMS) {
      const EdgeTerms *terms = (const EdgeTerms *)T.edges;
      for (uint32_t e = es; e < ee; e += WALK_BATCH) {
        EdgeTerms tt[WALK_BATCH];
#pragma unroll
        for (uint32_t u = 0; u < WALK_BATCH; ++u) tt[u] = terms[umin(e + u, ee - 1) - T.edge_bias];
#pragma unroll
        for (uint32_t u = 0; u < WALK_BATCH; ++u) acc += (e + u < ee) ? pot_fix(tt[u].t1 - tt[u].t0) : 0ll;
      }
    } else {
      for (uint32_t e = es; e < ee; ++e) {
        const EdgeRec er = T.edges[e - T.edge_bias];
        const double w = edge_weight<WMODE>(P, T, er, e);
        acc += pot_fix(w * (double)er.fval - w * (double)bits_to_float(er.aux));
      }
    }
    pp = pot_unfix(acc);
    return;
  }
  if (WMODE == W_PRESUM) {   // (inference: the evidence chain's sums; learning, free chain only: the free chain's)
    const bool evid = assign == P.assign_evid;
    pp = T.presum[evid ? 2 : 0]; pn = T.presum[evid ? 3 : 1];
    return;
  }

  if (Coop<WMODE>::on) {
    const uint32_t *const chains[1] = {assign};
    const int chain[2] = {0, 0};
    const uint32_t prop[2] = {1u, 0u};
    const bool hit[2] = {true, false};
    coop_for_records<WMODE, 2, 1>(P, T, es, ee, me, chains, chain, prop, hit,
                                  [&](const EdgeRec &, uint32_t, double w, const double (&term)[2]) {
                                    pp += w * term[0];
                                    pn += w * term[1];
                                  });
    pp = Coop<WMODE>::sum(pp); pn = Coop<WMODE>::sum(pn);
    return;
  }
  if (WMODE == W_TERMS8) {
    const unsigned long long *tab = (const unsigned long long *)T.edges;
    for (uint32_t e = es; e < ee; e += WALK_BATCH) {
      unsigned long long u[WALK_BATCH];
#pragma unroll
      for (uint32_t k = 0; k < WALK_BATCH; ++k) u[k] = tab[umin(e + k, ee - 1) - T.edge_bias];
#pragma unroll
      for (uint32_t k = 0; k < WALK_BATCH; ++k) {
        const bool in = e + k < ee;
        pp += in ? terms8_hit(u[k]) : 0.0;
        pn += in ? terms8_miss(u[k]) : 0.0;
      }
    }
    return;
  }
  if (WMODE == W_TERMS) {
    const EdgeTerms *terms = (const EdgeTerms *)T.edges;   // batched as in range_potential
    for (uint32_t e = es; e < ee; e += WALK_BATCH) {
      EdgeTerms tt[WALK_BATCH];
#pragma unroll
      for (uint32_t u = 0; u < WALK_BATCH; ++u) tt[u] = terms[umin(e + u, ee - 1) - T.edge_bias];
#pragma unroll
      for (uint32_t u = 0; u < WALK_BATCH; ++u) {
        const bool in = e + u < ee;
        pp += in ? tt[u].t1 : 0.0;
        pn += in ? tt[u].t0 : 0.0;
      }
    }
    return;
  }
  if (SIMPLE) {
    for (uint32_t e = es; e < ee; ++e) {
      const EdgeRec er = T.edges[e - T.edge_bias];
      const double w = edge_weight<WMODE>(P, T, er, e);
      pp += w * (double)er.fval;
      pn += w * (double)bits_to_float(er.aux);
    }
    return;
  }
  // Generic records, GEN_BATCH per step in three phases, so that the loads of a phase are all
  // in flight together: the records' first GEN_ARITY vif entries (clamped inside the factor;
  // a pre-signed record reads entry 0), then those variables' assignments, then the
  // arithmetic -- in row order, one product per record and proposal, as before.  A factor
  // wider than GEN_ARITY walks memory as it always did.
  const uint32_t prop[2] = {1u, 0u};
  for (uint32_t e0 = es; e0 < ee; e0 += GEN_BATCH) {
    EdgeRec er[GEN_BATCH];
    VifRec vf[GEN_BATCH][GEN_ARITY];
    uint32_t val[GEN_BATCH][1][GEN_ARITY];
#pragma unroll
    for (uint32_t u = 0; u < GEN_BATCH; ++u) {
      er[u] = T.edges[umin(e0 + u, ee - 1) - T.edge_bias];
      const bool generic = !(er[u].packed & EDGE_PRESIGNED);
      const uint32_t ar = generic ? edge_arity(er[u]) : 1u, base = (generic && ar >= 2u) ? er[u].aux : 0u;
#pragma unroll
      for (uint32_t i = 0; i < GEN_ARITY; ++i) vf[u][i] = P.vifs[base + umin(i, ar - 1u)];
    }
#pragma unroll
    for (uint32_t u = 0; u < GEN_BATCH; ++u)
#pragma unroll
      for (uint32_t i = 0; i < GEN_ARITY; ++i) val[u][0][i] = assign[vf[u][i].vid];
#pragma unroll
    for (uint32_t u = 0; u < GEN_BATCH; ++u) {
      const uint32_t e = e0 + u;
      if (e >= ee) continue;
      const double w = edge_weight<WMODE>(P, T, er[u], e);
      if (er[u].packed & EDGE_PRESIGNED) {
        pp += w * (double)er[u].fval;
        pn += w * (double)bits_to_float(er[u].aux);
        continue;
      }
      const double fv = (er[u].packed & EDGE_F64_FLAG) ? P.edge_fval64[e] : (double)er[u].fval;
      const uint32_t func = edge_func(er[u]), ar = edge_arity(er[u]);
      double sg[2];
      if (ar == 1u) {
        sg[0] = unary_sign(func, 1u == er[u].aux); sg[1] = unary_sign(func, 0u == er[u].aux);
      } else if (ar <= GEN_ARITY) {
        const int chain[2] = {0, 0};
        const VifsPreloaded<2, 1> src{vf[u], val[u], chain};
        factor_signs_from<2>(func, ar, src, me, prop, sg);
      } else {
        const uint32_t *const arr[2] = {assign, assign};
        factor_signs<2>(func, ar, er[u].aux, P.vifs, me, arr, prop, sg);
      }
      pp += w * (sg[0] * fv);
      pn += w * (sg[1] * fv);
    }
  }
}

// learning, generic path: the potentials of BOTH chains of a boolean variable in one walk
// (same sums, same order as two calls of bool_potentials)
template <int WMODE>
DWX_DEV void bool_potentials_both(const KernelParams &P, const TileView &T, uint32_t row, uint32_t me,
                                  double &ppf, double &pnf, double &ppe, double &pne) {
  const uint32_t es = T.rowptr[row - T.row_bias], ee = T.rowptr[row + 1 - T.row_bias];
  ppf = 0.0; pnf = 0.0; ppe = 0.0; pne = 0.0;
  if (WMODE == W_PRESUM) {
    ppf = T.presum[0]; pnf = T.presum[1]; ppe = T.presum[2]; pne = T.presum[3];
    return;
  }
  if (Coop<WMODE>::on) {
    const uint32_t *const chains[2] = {P.assign_free, P.assign_evid};
    const int chain[4] = {0, 0, 1, 1};
    const uint32_t prop[4] = {1u, 0u, 1u, 0u};
    const bool hit[4] = {true, false, true, false};
    coop_for_records<WMODE, 4, 2>(P, T, es, ee, me, chains, chain, prop, hit,
                                  [&](const EdgeRec &, uint32_t, double w, const double (&term)[4]) {
                                    ppf += w * term[0]; pnf += w * term[1];
                                    ppe += w * term[2]; pne += w * term[3];
                                  });
    ppf = Coop<WMODE>::sum(ppf); pnf = Coop<WMODE>::sum(pnf);
    ppe = Coop<WMODE>::sum(ppe); pne = Coop<WMODE>::sum(pne);
    return;
  }
  for (uint32_t e = es; e < ee; ++e) {
    const EdgeRec er = T.edges[e - T.edge_bias];
    const double w = edge_weight<WMODE>(P, T, er, e);
    if (er.packed & EDGE_PRESIGNED) {
      const double h = w * (double)er.fval, m = w * (double)bits_to_float(er.aux);
      ppf += h; pnf += m; ppe += h; pne += m;
    } else {
      const double fv = (er.packed & EDGE_F64_FLAG) ? P.edge_fval64[e] : (double)er.fval;
      const uint32_t prop[4] = {1u, 0u, 1u, 0u};
      double sg[4];
      if (edge_arity(er) == 1u) {
#pragma unroll
        for (int j = 0; j < 4; ++j) sg[j] = unary_sign(edge_func(er), prop[j] == er.aux);
      } else {
        const uint32_t *const chains[2] = {P.assign_free, P.assign_evid};
        const int chain[4] = {0, 0, 1, 1};
        record_signs<4, 2>(P, er, me, chains, chain, prop, sg);
      }
      ppf += w * (sg[0] * fv); pnf += w * (sg[1] * fv);
      ppe += w * (sg[2] * fv); pne += w * (sg[3] * fv);
    }
  }
}

#ifndef DWX_DRAW_GUARD
#define DWX_DRAW_GUARD 1e-4
#endif
// (tests/hipemu builds one library with DWX_DRAW_GUARD = 2: the f32 tier then never decides and every
// categorical draw of a small domain goes through the second, linear-space tier -- parity over it)
constexpr double DRAW_GUARD = DWX_DRAW_GUARD;   // >> every f32 error bound below
constexpr uint32_t SMALL_CARD = 8;    // domains up to this size are drawn out of registers

// src/gibbs_sampler.h:204-214: proposal 1 iff r * (1 + exp(pn - pp)) < 1.
// Fast path: the same quantity with an f32 exp (relative error < 1e-5 for |x| < 30); its
// verdict is taken only when it clears 1 by DRAW_GUARD, otherwise -- about one draw in
// 10^4 -- the exact f64 expression decides.  The result therefore ALWAYS equals the exact
// expression's; only the f64 exp is skipped.
DWX_DEV uint32_t bool_draw(double r, double pp, double pn) {
  const double x = pn - pp;
  if (x > -30.0 && x < 30.0) {
    const double q = r * (1.0 + (double)DWX_FAST_EXPF((float)x));
    if (q < 1.0 - DRAW_GUARD) return 1u;
    if (q > 1.0 + DRAW_GUARD) return 0u;
  }
  return (r * (1.0 + exp(x)) < 1.0) ? 1u : 0u;
}

// categorical draw, src/gibbs_sampler.h:217-246 (inverse CDF with ONE uniform):
//   sum = logadd over d of pot_d;  first d with  r - sum_{j<=d} exp(pot_j - sum) <= 0.
// Fast path (potentials buffered in LDS): normalise with max-subtracted f32 exps and pick
// the first d whose cumulative mass reaches r; accepted only if r is at least DRAW_GUARD
// away from both cumulative boundaries of that d (f32 error of a boundary < 1e-5, the
// reference's own logadd cut-off shifts it by < 1e-8); otherwise the exact sequence runs.
template <int WMODE, bool SIMPLE>
DWX_DEV uint32_t cat_draw(const KernelParams &P, const TileView &T, uint32_t row0, uint32_t card,
                          const uint32_t *assign, uint32_t me, double r) {
  if (SIMPLE && card <= SMALL_CARD) {
    // Small domains, all-unary tile (the usual case): potentials live in registers (fully
    // unrolled, no dynamic indexing), each row pointer is read once -- no LDS scratch
    // traffic.  (Not instantiated for the generic factor code: 8 inlined copies of it
    // would bloat the kernel far beyond the instruction cache.)
    double pot[SMALL_CARD];
    double m = -1e300;
    uint32_t es = T.rowptr[row0 - T.row_bias];
#pragma unroll
    for (uint32_t d = 0; d < SMALL_CARD; ++d) {
      pot[d] = -1e300;
      if (d < card) {
        const uint32_t ee = T.rowptr[row0 + d + 1 - T.row_bias];
        pot[d] = range_potential<WMODE, SIMPLE>(P, T, es, ee, assign, me, d);
        es = ee;
        m = pot[d] > m ? pot[d] : m;
      }
    }
    float ex[SMALL_CARD];
    float S = 0.f;
#pragma unroll
    for (uint32_t d = 0; d < SMALL_CARD; ++d) {
      const double z = pot[d] - m;
      ex[d] = (d < card && z > -30.0) ? DWX_FAST_EXPF((float)z) : 0.f;
      S += ex[d];
    }
    const double target = r * (double)S, guard = DRAW_GUARD * (double)S;
    float c = 0.f;
    bool decided = false, near = false;
    uint32_t pick = 0;
#pragma unroll
    for (uint32_t d = 0; d < SMALL_CARD; ++d) {
      const float lo = c;
      c += ex[d];
      if (!decided && !near && d < card && (double)c >= target) {
        if (target - (double)lo > guard && (double)c - target > guard) { decided = true; pick = d; }
        else near = true;
      }
    }
    if (decided) return pick;
#ifdef DWX_EXP_NOSLOW      // (timing experiment, results invalid: what the exact f64 sequence costs a chunk's critical path)
    return pick;
#endif
    // Second tier (round 4).  The exact sequence below is sixteen DEPENDENT f64 transcendentals
    // (eight logadds = exp + log1p each, eight exps): ~5 us for the whole wave whenever ONE of its
    // lanes is within DRAW_GUARD of a boundary -- one wave in ten; measured on config 4: a third of
    // the inference sweep and 5 us on the critical path of every mini-batch of a split learning sweep.
    // The same sequence in LINEAR space needs eight INDEPENDENT f64 exps: with e_d = exp(pot_d - m),
    // the reference's sum = logadd(... logadd(-100000, pot_0) ..., pot_{card-1}) is log(L) + m where
    // L follows  L <- hi + lo,  or  L <- hi  when lo / hi < exp(-18.42)  (the cut-off of logadd,
    // src/common.h:118-132; lo, hi = min, max of (L, e_d); the start value -100000 is dropped the
    // same way), and r -= exp(pot_d - sum) is r -= e_d / L: the first d with e_0 + .. + e_d >= r L.
    // Every quantity here carries ~1e-15 relative error like the reference's own; the verdict is
    // taken only 1e-11 L away from both cumulative boundaries and with every cut-off decision 1e-9
    // clear of its threshold -- otherwise (one draw in 10^10) the exact sequence still decides.
    // (|m| < 1000: the reference's sum lives in log space, its rounding error grows with its magnitude
    // -- 8 ulp(1000) ~ 1e-12 is what the 1e-11 guard covers; larger potentials take the exact path.)
    if (m > -1000.0 && m < 1000.0) {
      constexpr double CUT = 1.0006809757111146e-08;    // exp(-18.42)
      double e[SMALL_CARD];
#pragma unroll
      for (uint32_t d = 0; d < SMALL_CARD; ++d) e[d] = d < card ? exp(pot[d] - m) : 0.0;
      double L = 0.0;
      bool clear = true;
#pragma unroll
      for (uint32_t d = 0; d < SMALL_CARD; ++d) {
        if (d < card) {
          const double lo = L < e[d] ? L : e[d], hi = L < e[d] ? e[d] : L;
          const double thr = hi * CUT, gap = lo - thr;
          if ((gap < 0.0 ? -gap : gap) <= 1e-9 * thr && hi > 0.0) clear = false;
          L = gap < 0.0 ? hi : hi + lo;
        }
      }
      const double target2 = r * L, g2 = 1e-11 * L;
      double cum = 0.0;
      bool done2 = false, near2 = false;
      uint32_t pick2 = card - 1;
#pragma unroll
      for (uint32_t d = 0; d < SMALL_CARD; ++d) {
        const double lo2 = cum;
        cum += e[d];
        if (!done2 && !near2 && d < card && cum >= target2) {
          if (target2 - lo2 > g2 && cum - target2 > g2) { done2 = true; pick2 = d; }
          else near2 = true;
        }
      }
      if (clear && done2) return pick2;
    }
    // exact: the reference's sequence
    double sum = -100000.0;
#pragma unroll
    for (uint32_t d = 0; d < SMALL_CARD; ++d) if (d < card) sum = logadd(sum, pot[d]);
    uint32_t res = card - 1;   // the reference asserts here (:243); rounding can leave r > 0
    bool found = false;
#pragma unroll
    for (uint32_t d = 0; d < SMALL_CARD; ++d) {
      if (d < card && !found) {
        r -= exp(pot[d] - sum);
        if (r <= 0) { res = d; found = true; }
      }
    }
    return res;
  }
  if (T.pot) {
    double *pot = T.pot + (row0 - T.row_bias);
    double m = -1e300;
    for (uint32_t d = 0; d < card; ++d) {
      const double v = row_potential<WMODE, SIMPLE>(P, T, row0 + d, assign, me, d);
      pot[d] = v;
      m = v > m ? v : m;
    }
    float S = 0.f;
    for (uint32_t d = 0; d < card; ++d) {
      const double z = pot[d] - m;
      S += z > -30.0 ? DWX_FAST_EXPF((float)z) : 0.f;
    }
    const double target = r * (double)S, guard = DRAW_GUARD * (double)S;
    float c = 0.f;
    for (uint32_t d = 0; d < card; ++d) {
      const double z = pot[d] - m;
      const float lo = c;
      c += z > -30.0 ? DWX_FAST_EXPF((float)z) : 0.f;
      if ((double)c >= target) {
        if (target - (double)lo > guard && (double)c - target > guard) return d;
        break;   // too close to a boundary: let the exact sequence decide
      }
    }
    // exact: the reference's sequence on the buffered potentials
    double sum = -100000.0;
    for (uint32_t d = 0; d < card; ++d) sum = logadd(sum, pot[d]);
    for (uint32_t d = 0; d < card; ++d) {
      r -= exp(pot[d] - sum);
      if (r <= 0) return d;
    }
    return card - 1;  // the reference asserts here (:243); rounding can leave r > 0
  }
  // no scratch (oversized variable): recompute potentials instead of buffering them
  double sum = -100000.0;
  for (uint32_t d = 0; d < card; ++d)
    sum = logadd(sum, row_potential<WMODE, SIMPLE>(P, T, row0 + d, assign, me, d));
  for (uint32_t d = 0; d < card; ++d) {
    r -= exp(row_potential<WMODE, SIMPLE>(P, T, row0 + d, assign, me, d) - sum);
    if (r <= 0) return d;
  }
  return card - 1;
}

// sgd_on_factor (src/factor_graph.cc:243-260), gradient accumulated in fixed point:
// G[wid] += round(2^30 * t * (pot_free - pot_evid)),  T[wid] += round(2^30 * t).
// count_t = false for boolean variables of an un-split sweep: their update counts are
// static and were folded into T_static on the host (dwx_sampler_create).
// hit_value: the proposal that "hits" a pre-signed record of this row (1 for a boolean
// variable, the row's value for a categorical one).
// sgd_on_factor over records [es, ee), shared out over the lanes of a cooperating group
template <int WMODE>
DWX_DEV void coop_sgd_range(const KernelParams &P, const TileView &T, uint32_t es, uint32_t ee, uint32_t me,
                            uint32_t evid_value, uint32_t free_value, uint32_t hit_value, double t, bool count_t) {
  const uint32_t *const chains[2] = {P.assign_evid, P.assign_free};
  const int chain[2] = {0, 1};
  const uint32_t prop[2] = {evid_value, free_value};
  const bool hit[2] = {evid_value == hit_value, free_value == hit_value};
  coop_for_records<WMODE, 2, 2>(P, T, es, ee, me, chains, chain, prop, hit,
                                [&](const EdgeRec &er, uint32_t, double, const double (&term)[2]) {
    if (er.packed & EDGE_FIXED_FLAG) return;   // weights_isfixed (src/factor_graph.cc:247)
    const long long gi = llrint(FIX_SCALE * (t * (term[1] - term[0])));
    if (gi) atomicAdd((unsigned long long *)&P.grad[er.wid], (unsigned long long)gi);
    if (count_t) atomicAdd((unsigned long long *)&P.grad[P.num_weights + er.wid], (unsigned long long)llrint(FIX_SCALE * t));
  });
}

// W_COOP / W_COOPB: the lanes of the cooperating group share the row's records.
template <bool SIMPLE, int WMODE = W_GLOBAL>
DWX_DEV void sgd_row(const KernelParams &P, const TileView &T, uint32_t row, uint32_t me,
                     uint32_t evid_value, uint32_t free_value, uint32_t hit_value, double t,
                     const bool count_t) {
  const uint32_t es = T.rowptr[row - T.row_bias], ee = T.rowptr[row + 1 - T.row_bias];
  if (WMODE == W_FIXSUM) return;   // (never reached: those tiles take the pull gradient)
  if (WMODE == W_PRESUM) {   // boolean only (hit_value 1, t 1): the pieces' workgroups walk the row
    T.decision[0] = evid_value; T.decision[1] = free_value; T.decision[2] = 1u | (count_t ? 2u : 0u);
    return;
  }
  if (Coop<WMODE>::on) {
    coop_sgd_range<WMODE>(P, T, es, ee, me, evid_value, free_value, hit_value, t, count_t);
    return;
  }
  if (WMODE == W_LREC) {
    const LearnRec *recs = (const LearnRec *)T.edges;
    const bool evid_hits = evid_value == hit_value, free_hits = free_value == hit_value;
    for (uint32_t e = es; e < ee; ++e) {
      const LearnRec r = recs[e - T.edge_bias];
      if (r.packed & EDGE_FIXED_FLAG) continue;
      const double g = (double)(free_hits ? r.sf1 : r.sf0) - (double)(evid_hits ? r.se1 : r.se0);
      const long long gi = llrint(FIX_SCALE * (t * g));
      const long long ti = count_t ? llrint(FIX_SCALE * t) : 0;
      long long *dst = T.agg ? T.agg : P.grad;
      if (gi) atomicAdd((unsigned long long *)&dst[r.wid], (unsigned long long)gi);
      if (count_t) atomicAdd((unsigned long long *)&dst[P.num_weights + r.wid], (unsigned long long)ti);
    }
    return;
  }
  for (uint32_t e = es; e < ee; ++e) {
    const EdgeRec er = T.edges[e - T.edge_bias];
    if (er.packed & EDGE_FIXED_FLAG) continue;   // weights_isfixed (src/factor_graph.cc:247)
    double pot_evid, pot_free;
    if (SIMPLE || (er.packed & EDGE_PRESIGNED)) {
      pot_evid = edge_term<true>(P, er, e, P.assign_evid, me, evid_value, evid_value == hit_value);
      pot_free = edge_term<true>(P, er, e, P.assign_free, me, free_value, free_value == hit_value);
    } else {   // one walk over the factor for both evaluations
      const double fv = (er.packed & EDGE_F64_FLAG) ? P.edge_fval64[e] : (double)er.fval;
      const uint32_t prop[2] = {evid_value, free_value};
      double sg[2];
      if (edge_arity(er) == 1u) {
        sg[0] = unary_sign(edge_func(er), prop[0] == er.aux); sg[1] = unary_sign(edge_func(er), prop[1] == er.aux);
      } else {
        const uint32_t *const chains[2] = {P.assign_evid, P.assign_free};
        const int chain[2] = {0, 1};
        record_signs<2, 2>(P, er, me, chains, chain, prop, sg);
      }
      pot_evid = sg[0] * fv;
      pot_free = sg[1] * fv;
    }
    const double g = pot_free - pot_evid;
    const long long gi = llrint(FIX_SCALE * (t * g));
    const long long ti = count_t ? llrint(FIX_SCALE * t) : 0;
    if (T.agg) {   // workgroup-local accumulation in LDS (few, heavily shared weights)
      if (gi) atomicAdd((unsigned long long *)&T.agg[er.wid], (unsigned long long)gi);
      if (count_t) atomicAdd((unsigned long long *)&T.agg[P.num_weights + er.wid], (unsigned long long)ti);
    } else {
      if (gi) atomicAdd((unsigned long long *)&P.grad[er.wid], (unsigned long long)gi);
      if (count_t) atomicAdd((unsigned long long *)&P.grad[P.num_weights + er.wid], (unsigned long long)ti);
    }
  }
}

// ---------------------------------------------------------------- one variable
// Per-lane inputs of a variable, prefetched one tile ahead.
struct VarPre {
  uint32_t meta, orig, row0, init;
};

// independent loads only (no load depends on another: vmcnt retires in order, so a
// dependent load here would make the whole prefetch wait)
// NT: non-temporal (these words are read once per sweep; cached they evict the f32 weight
// table that the gathers re-use -- config 3: -3 % per sweep; the table-streaming inference
// build gathers nothing and is 9 % faster with plain loads)
template <bool LEARN, bool NT = true>
DWX_DEV VarPre load_var_pre(const KernelParams &P, uint32_t p) {
  VarPre v;
  if (NT) {
    v.meta = DWX_NT_LOAD(&P.v_meta[p]);
    v.orig = DWX_NT_LOAD(&P.v_orig[p]);
    v.row0 = DWX_NT_LOAD(&P.v_row[p]);
    v.init = LEARN ? DWX_NT_LOAD(&P.v_init[p]) : 0u;   // dense evidence value (assignment_dense)
  } else {
    v.meta = P.v_meta[p];
    v.orig = P.v_orig[p];
    v.row0 = P.v_row[p];
    v.init = LEARN ? P.v_init[p] : 0u;
  }
  return v;
}

// MULTI: P.n_sweeps consecutive inference sweeps of ONE variable of an all-unary tile in one go.
// No factor of such a graph reads another variable, so a variable's potentials do not change
// from sweep to sweep (the weights are fixed during inference): they are summed ONCE, exactly as
// a single sweep sums them, and sweep k draws with its own uniform -- Philox keyed (seed, global
// id, sweep + k), as n separate launches would.  The draws are the reference's own expressions
// (src/gibbs_sampler.h:204-214 / :217-246) with their sweep-invariant parts hoisted:
// r * (1 + exp(pn - pp)) < 1, and r -= exp(pot_d - sum) until r <= 0 -- what bool_draw and
// cat_draw return by construction (their f32 fast paths only ever shortcut to this verdict).
// Tallies are added once per value, the assignment is the last sweep's: bit for bit what
// n_sweeps calls of the single-sweep path leave behind (tests/test_multi_sweep.py).
// [k_lo, k_hi): the slice of the launch's sweeps this call draws (a tile with fewer variables
// than lanes cuts a long run of sweeps into slices, so that every lane draws: sweep8_kernel);
// the slice that holds the last sweep stores the assignment.
template <int WMODE, bool FIXED>
DWX_DEV void infer_variable_multi(const KernelParams &P, const TileView &T, uint32_t p, const VarPre pre,
                                  const uint32_t k_lo, const uint32_t k_hi, const bool store) {
  const uint32_t meta = pre.meta;
  if ((meta & VM_EVIDENCE) && !(P.flags & OPT_SAMPLE_EVIDENCE)) return;
  const uint32_t card = meta >> VM_CARD_SHIFT, row0 = pre.row0;
  const uint64_t vid = P.vid_offset + pre.orig;
  uint32_t prop = 0;
  if (!(meta & VM_CATEGORICAL)) {
    double pp, pn;
    bool_potentials<WMODE, true, FIXED>(P, T, row0, P.assign_evid, p, pp, pn);
    const double scale = 1.0 + exp(pn - pp);
    uint32_t count = 0;
    for (uint32_t k = k_lo; k < k_hi; ++k) {
      double A, B;
      philox_uniforms(P.seed, vid, P.sweep + k, A, B);
      prop = (A * scale < 1.0) ? 1u : 0u;
      count += prop;
    }
    if (count) atomicAdd(&P.tally[row0], count);
  } else if (card <= SMALL_CARD) {
    double e[SMALL_CARD];
    uint32_t es = T.rowptr[row0 - T.row_bias];
#pragma unroll
    for (uint32_t d = 0; d < SMALL_CARD; ++d) {
      e[d] = -1e300;
      if (d < card) {
        const uint32_t ee = T.rowptr[row0 + d + 1 - T.row_bias];
        e[d] = range_potential<WMODE, true>(P, T, es, ee, P.assign_evid, p, d);
        es = ee;
      }
    }
    double sum = -100000.0;
#pragma unroll
    for (uint32_t d = 0; d < SMALL_CARD; ++d) if (d < card) sum = logadd(sum, e[d]);
#pragma unroll
    for (uint32_t d = 0; d < SMALL_CARD; ++d) e[d] = d < card ? exp(e[d] - sum) : 0.0;
    // The reference stops at the first d with r <= 0.  Every e[d] is >= 0, so the running r
    // never rises (a rounded subtraction of a non-negative number is monotone): the values
    // r > 0 form a prefix, and the first d with r <= 0 is the NUMBER of d with r > 0 -- capped
    // at card - 1 (the reference asserts there; rounding can leave r > 0 after the last value,
    // and the padding slots d >= card subtract 0).  Same subtractions in the same order, no
    // early exit to track.  Counts: eight 8-bit counters in one 64-bit word, unpacked every 255
    // draws.
    static_assert(SMALL_CARD == 8, "eight 8-bit counters per 64-bit word");
    uint32_t cnt[SMALL_CARD];
#pragma unroll
    for (uint32_t d = 0; d < SMALL_CARD; ++d) cnt[d] = 0;
    for (uint32_t k0 = k_lo; k0 < k_hi; k0 += 255u) {
      const uint32_t k1 = umin(k_hi, k0 + 255u);
      unsigned long long packed = 0;
      for (uint32_t k = k0; k < k1; ++k) {
        double r, B;
        philox_uniforms(P.seed, vid, P.sweep + k, r, B);
        uint32_t res = 0;
#pragma unroll
        for (uint32_t d = 0; d < SMALL_CARD; ++d) {
          r -= e[d];
          res += r > 0 ? 1u : 0u;
        }
        res = umin(res, card - 1);
        packed += 1ull << (res * 8u);
        prop = res;
      }
#pragma unroll
      for (uint32_t d = 0; d < SMALL_CARD; ++d) cnt[d] += (uint32_t)(packed >> (d * 8u)) & 255u;
    }
#pragma unroll
    for (uint32_t d = 0; d < SMALL_CARD; ++d)
      if (d < card && cnt[d]) atomicAdd(&P.tally[row0 + d], cnt[d]);
  } else {
    // (bigger domains: cat_draw sums the potentials into the tile's LDS scratch on every call; two
    // slices of one variable write the same values there)
    for (uint32_t k = k_lo; k < k_hi; ++k) {
      double A, B;
      philox_uniforms(P.seed, vid, P.sweep + k, A, B);
      prop = cat_draw<WMODE, true>(P, T, row0, card, P.assign_evid, p, A);
      atomicAdd(&P.tally[row0 + prop], 1u);
    }
  }
  if (store) DWX_NT_STORE(prop, &P.assign_evid[p]);
}

// want_delta (learning, TILE_PULL tiles only): instead of scattering gradient atomics,
// return hit(free) - hit(evid) in {-1,0,+1} for a variable that triggers SGD (0 otherwise).
// NOCAT: the caller's graph has no categorical lane tile (sweep_kernel's TV_PAIR build): every
// categorical branch -- cat_draw, the noise-aware rows -- compiles out.
template <bool LEARN, int WMODE, bool SIMPLE, bool FIXED = false, bool NOCAT = false>
DWX_DEV int process_variable(const KernelParams &P, const TileView &T, uint32_t p,
                             const VarPre pre, double A, double B, const bool want_delta = false) {
  const uint32_t meta = pre.meta;
  const bool is_cat = !NOCAT && (meta & VM_CATEGORICAL);
  const bool is_evid = meta & VM_EVIDENCE;
  const uint32_t card = meta >> VM_CARD_SHIFT;
  const uint32_t row0 = pre.row0;
  // W_COOP: all 64 lanes of the wave run this function for the SAME variable; the potentials
  // are wave-wide sums (identical in every lane), so every lane takes the same decisions;
  // stores and tallies happen once, the gradient rows are shared out over the lanes
  constexpr bool COOP = Coop<WMODE>::on;
  const bool leader = !COOP || Coop<WMODE>::lane() == 0u;
  if (!LEARN) {
    // sample_single_variable (src/gibbs_sampler.h:151-169)
    if (is_evid && !(P.flags & OPT_SAMPLE_EVIDENCE)) return 0;
    uint32_t prop;
    if (!is_cat) {
      double pp, pn;
      bool_potentials<WMODE, SIMPLE, FIXED>(P, T, row0, P.assign_evid, p, pp, pn);
      prop = bool_draw(A, pp, pn);
      // single owner per row: a no-return atomic is a fire-and-forget increment the
      // wave never waits for (a load-add-store would stall on the load)
      if (prop && leader) atomicAdd(&P.tally[row0], 1u);
    } else {
      prop = cat_draw<WMODE, SIMPLE>(P, T, row0, card, P.assign_evid, p, A);
      if (leader) atomicAdd(&P.tally[row0 + prop], 1u);
    }
    // (a variable of an all-unary tile has no neighbours: nobody re-reads its assignment)
    if (SIMPLE) DWX_NT_STORE(prop, &P.assign_evid[p]); else if (leader) P.assign_evid[p] = prop;
    return 0;
  }
  // sample_sgd_single_variable (src/gibbs_sampler.h:127-149)
  const bool noise_aware = P.flags & OPT_NOISE_AWARE;
  const bool has_truth = meta & VM_TRUTHINESS;
  // free chain
  uint32_t p_free;
  double pp_f = 0.0, pn_f = 0.0;
  // (generic path: if the evidence chain will be drawn too, its potentials come from the same walk)
  const bool both = !SIMPLE && !is_cat && !(!noise_aware && is_evid) && !(noise_aware && has_truth);
  double pp_e = 0.0, pn_e = 0.0;
  if (!is_cat) {
    if (both) bool_potentials_both<WMODE>(P, T, row0, p, pp_f, pn_f, pp_e, pn_e);
    else bool_potentials<WMODE, SIMPLE, FIXED>(P, T, row0, P.assign_free, p, pp_f, pn_f);
    p_free = bool_draw(A, pp_f, pn_f);
  } else {
    p_free = cat_draw<WMODE, SIMPLE>(P, T, row0, card, P.assign_free, p, A);
  }
  if (SIMPLE) DWX_NT_STORE(p_free, &P.assign_free[p]); else if (leader) P.assign_free[p] = p_free;
  // evidence chain: sample_evid (src/gibbs_sampler.h:171-190)
  const uint32_t evid_value = pre.init;
  uint32_t p_evid;
  if (!noise_aware && is_evid) {
    p_evid = evid_value;
  } else if (noise_aware && has_truth) {
    double sum = 0;
    p_evid = 0;
    for (uint32_t i = 0; i < card; ++i) {
      sum += P.row_truth[row0 + i];
      if (sum >= B) { p_evid = i; break; }
    }
  } else if (!is_cat) {
    // unary factors do not read neighbours: both chains see the same potentials
    // (same records, same weights, same order => bit-identical sums)
    const double pp = SIMPLE ? pp_f : pp_e, pn = SIMPLE ? pn_f : pn_e;
    p_evid = bool_draw(B, pp, pn);
  } else {
    p_evid = cat_draw<WMODE, SIMPLE>(P, T, row0, card, P.assign_evid, p, B);
  }
  if (SIMPLE) DWX_NT_STORE(p_evid, &P.assign_evid[p]); else if (leader) P.assign_evid[p] = p_evid;
  // src/gibbs_sampler.h:144-146
  if (!(P.flags & OPT_LEARN_NON_EVIDENCE) &&
      ((!noise_aware && !is_evid) || (noise_aware && !has_truth)))
    return 0;
  // sgd_on_variable (src/factor_graph.cc:262-314)
  if (!is_cat) {
    // a pre-signed record's gradient is (free hits ? A : B) - (evid hits ? A : B): zero
    // for the whole row when both chains agree (update counts are static, T_static)
    const bool dyn_t = P.flags & OPT_DYNAMIC_T;
    if (SIMPLE && !dyn_t && p_free == evid_value) return 0;
    if (SIMPLE && want_delta) return (int)p_free - (int)evid_value;
    sgd_row<SIMPLE, WMODE>(P, T, row0, p, evid_value, p_free, 1u, 1.0, dyn_t);
    return 0;
  }
  for (uint32_t val = 0; val < card; ++val) {
    if (!noise_aware && val != evid_value) continue;
    double t = 1.0;
    if (noise_aware) {
      t = P.row_truth ? P.row_truth[row0 + val] : 0.0;
      if (is_linear_zero(t)) continue;
    }
    sgd_row<SIMPLE, WMODE>(P, T, row0 + val, p, val, p_free, val, t, true);
    if (val == p_free) continue;
    sgd_row<SIMPLE, WMODE>(P, T, row0 + p_free, p, val, p_free, p_free, t, true);
  }
  return 0;
}

// ---------------------------------------------------------------- learning, TILE_TERMS2

// the even bits of x, packed (lanes 2j and 2j + 1 of a chain-pair tile vote alike: 32 variables per wave)
DWX_DEV uint32_t compress_even_bits(unsigned long long x) {
  x &= 0x5555555555555555ull;
  x = (x | (x >> 1)) & 0x3333333333333333ull;
  x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
  x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
  x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
  x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
  return (uint32_t)x;
}

// sample_sgd_single_variable (src/gibbs_sampler.h:127-149) + sgd_on_variable
// (src/factor_graph.cc:262-275) for a boolean variable, everything out of LDS.
// pull_unary (TILE_PULL_UNARY): the pre-signed records' gradient is pulled (aux_kernels.h) from the
// returned hit(free) - hit(evid) in {-1, 0, +1} (0 for a variable that triggers no SGD); only the
// other records scatter.
template <class Rec>
DWX_DEV int learn_variable_terms2(const KernelParams &P, const uint32_t *rowptr, uint32_t row_bias,
                                  const Rec *recs, uint32_t edge_bias, long long *agg,
                                  uint32_t p, const VarPre pre, double A, double B, const bool pull_unary) {
  const bool is_evid = pre.meta & VM_EVIDENCE;
  const bool noise_aware = P.flags & OPT_NOISE_AWARE;
  const uint32_t es = rowptr[pre.row0 - row_bias], ee = rowptr[pre.row0 + 1 - row_bias];
  double ppf = 0.0, pnf = 0.0, ppe = 0.0, pne = 0.0;
  // LEARN_BATCH staged records per step (all LDS reads in flight); sums stay sequential and in
  // row order, a slot past the end adds +0.0 (cannot change a running sum)
  for (uint32_t e = es; e < ee; e += LEARN_BATCH) {
    Rec r[LEARN_BATCH];
#pragma unroll
    for (uint32_t u = 0; u < LEARN_BATCH; ++u) r[u] = recs[umin(e + u, ee - 1) - edge_bias];
#pragma unroll
    for (uint32_t u = 0; u < LEARN_BATCH; ++u) {
      const bool in = e + u < ee;
      const double w = (double)r[u].w;
      ppf += in ? w * (double)lrec_term(r[u], 0u, 1u) : 0.0; pnf += in ? w * (double)lrec_term(r[u], 0u, 0u) : 0.0;
      ppe += in ? w * (double)lrec_term(r[u], 1u, 1u) : 0.0; pne += in ? w * (double)lrec_term(r[u], 1u, 0u) : 0.0;
    }
  }
  const uint32_t p_free = bool_draw(A, ppf, pnf);
  P.assign_free[p] = p_free;
  const uint32_t evid_value = pre.init;
  // boolean variables carry no truthiness: sample_evid is "evidence value" or a Gibbs draw
  const uint32_t p_evid = (!noise_aware && is_evid) ? evid_value : bool_draw(B, ppe, pne);
  P.assign_evid[p] = p_evid;
  if (!(P.flags & OPT_LEARN_NON_EVIDENCE) && (noise_aware || !is_evid)) return 0;
  for (uint32_t e = es; e < ee; ++e) {
    const Rec r = recs[e - edge_bias];
    if (lrec_fixed(r)) continue;
    if (pull_unary && lrec_presigned(r)) continue;
    const double pot_free = (double)lrec_term(r, 0u, p_free);
    const double pot_evid = (double)lrec_term(r, 1u, evid_value);
    const long long gi = llrint(FIX_SCALE * (pot_free - pot_evid));
    long long *dst = agg ? agg : P.grad;
    const uint32_t wid = lrec_wid(r);
#ifndef DWX_EXP_NOSCATTER   // (experiment: what the pairwise records' gradient atomics cost -- profiles/r04/v3/)
    if (gi) atomicAdd((unsigned long long *)&dst[wid], (unsigned long long)gi);
#endif
    if (P.flags & OPT_DYNAMIC_T)
      atomicAdd((unsigned long long *)&dst[P.num_weights + wid], (unsigned long long)(long long)FIX_SCALE);
  }
  return (int)p_free - (int)evid_value;
}

// The same with TWO lanes per variable (tiles of at most 128 variables -- twelve and more records
// each -- would leave half of the workgroup idle in this phase): lane 2j sums and draws the free
// chain of variable j, lane 2j + 1 its evidence chain -- each sum in row order as before -- they
// swap the free sample, and each takes every other record of the gradient walk.
template <class Rec>
DWX_DEV int learn_variable_terms2_pair(const KernelParams &P, const uint32_t *rowptr, uint32_t row_bias,
                                       const Rec *recs, uint32_t edge_bias, long long *agg,
                                       uint32_t p, const VarPre pre, double A, double B, const uint32_t chain,
                                       const bool pull_unary) {
  const bool is_evid = pre.meta & VM_EVIDENCE;
  const bool noise_aware = P.flags & OPT_NOISE_AWARE;
  const uint32_t es = rowptr[pre.row0 - row_bias], ee = rowptr[pre.row0 + 1 - row_bias];
  double pp = 0.0, pn = 0.0;
  for (uint32_t e = es; e < ee; e += LEARN_BATCH) {
    Rec r[LEARN_BATCH];
#pragma unroll
    for (uint32_t u = 0; u < LEARN_BATCH; ++u) r[u] = recs[umin(e + u, ee - 1) - edge_bias];
#pragma unroll
    for (uint32_t u = 0; u < LEARN_BATCH; ++u) {
      const bool in = e + u < ee;
      const double w = (double)r[u].w;
      pp += in ? w * (double)lrec_term(r[u], chain, 1u) : 0.0;
      pn += in ? w * (double)lrec_term(r[u], chain, 0u) : 0.0;
    }
  }
  const uint32_t evid_value = pre.init;
  uint32_t mine;
  if (chain == 0u) {
    mine = bool_draw(A, pp, pn);
    P.assign_free[p] = mine;
  } else {
    mine = (!noise_aware && is_evid) ? evid_value : bool_draw(B, pp, pn);
    P.assign_evid[p] = mine;
  }
  const uint32_t theirs = DWX_PAIR_SWAP_U32(mine);
  const uint32_t p_free = chain == 0u ? mine : theirs;
  if (!(P.flags & OPT_LEARN_NON_EVIDENCE) && (noise_aware || !is_evid)) return 0;
  for (uint32_t e = es + chain; e < ee; e += 2u) {
    const Rec r = recs[e - edge_bias];
    if (lrec_fixed(r)) continue;
    if (pull_unary && lrec_presigned(r)) continue;
    const double pot_free = (double)lrec_term(r, 0u, p_free);
    const double pot_evid = (double)lrec_term(r, 1u, evid_value);
    const long long gi = llrint(FIX_SCALE * (pot_free - pot_evid));
    long long *dst = agg ? agg : P.grad;
    const uint32_t wid = lrec_wid(r);
#ifndef DWX_EXP_NOSCATTER   // (experiment: what the pairwise records' gradient atomics cost -- profiles/r04/v3/)
    if (gi) atomicAdd((unsigned long long *)&dst[wid], (unsigned long long)gi);
#endif
    if (P.flags & OPT_DYNAMIC_T)
      atomicAdd((unsigned long long *)&dst[P.num_weights + wid], (unsigned long long)(long long)FIX_SCALE);
  }
  return (int)p_free - (int)evid_value;   // (both lanes of the pair hold it)
}

// learning sweep over a boolean TILE_TERMS2 / TILE_TERMS3 tile of at most 128 variables: two
// lanes per variable (workgroup-uniform)
template <bool LEARN, int K, bool WIDE>
DWX_DEV bool chain_pair_tile(const TileDesc &d) {
  return K <= 6 && LEARN && WIDE && (d.flags & (TILE_TERMS2 | TILE_TERMS3)) && !(d.flags & TILE_CATEGORICAL) &&
         !(d.flags & TILE_OUTSIDE) && 2u * d.nv <= BLOCK_THREADS && DWX_CHAIN_PAIRS;
}

}  // namespace dwx
#endif  // DWX_TILE_WALK_H_
