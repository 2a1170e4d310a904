// sweep_kernels.h -- the Gibbs sweep as hand-written HIP for gfx950 (CDNA4).
//
// One workgroup (256 threads = 4 wave64) owns one TILE of consecutive variables of one
// colour.  The tile's value-row pointers and its 16-byte edge records are ONE
// contiguous HBM range each (variable-major layout, graph_compile.cc), staged into
// LDS with coalesced 16 B/lane loads; then one lane per variable walks its rows out
// of LDS in the reference's order (ascending (value, factor id)), gathers
// weight[wid] (L2 / Infinity-Cache resident) and neighbour assignments, draws with a
// counter-based Philox4x32-10 stream and writes its assignment + tally.  Variables
// of one launch form an independent set, so a launch is exactly a sequential Gibbs
// scan of those variables.  This path is gather/stream bound: no MFMA.
//
// Precision: potentials, the logistic / log-sum-exp draw and the SGD update are f64
// like the reference; the weight each factor is multiplied with is the f64 master
// weight rounded to f32 (the sampling copy: 4 MB for 1M weights, L2-resident).
//
// Reference functions restated here (paths relative to /root/reference):
//   sample_single_variable      src/gibbs_sampler.h:151-169
//   sample_sgd_single_variable  src/gibbs_sampler.h:127-149
//   sample_evid / draw_sample   src/gibbs_sampler.h:171-254
//   FactorGraph::potential      src/factor_graph.h:127-145
//   Factor::potential + signs   src/factor.h:59-299
//   sgd_on_variable / _factor   src/factor_graph.cc:243-314
//   update_weight               src/inference_result.h:66-85 (batched: apply_kernel)
//   logadd                      src/common.h:118-132
//
// The file is HIP source; tests/hipemu compiles the very same text for the host
// (fibers stand in for a workgroup) so the kernels run under ASan/UBSan in CI.
#ifndef DWX_SWEEP_KERNELS_H_
#define DWX_SWEEP_KERNELS_H_

#include "device_types.h"

#ifndef DWX_DEV
#define DWX_DEV __device__ __forceinline__
#endif

// Streamed-once loads / stores (per-variable words, row pointers, assignments of an all-unary
// graph): non-temporal, so that they do not evict the re-used f32 weight table from L2.
#ifndef DWX_NT_LOAD
#ifdef DWX_NO_NT_META
#define DWX_NT_LOAD(p) (*(p))
#define DWX_NT_STORE(v, p) (*(p) = (v))
#else
#define DWX_NT_LOAD(p) __builtin_nontemporal_load(p)
#define DWX_NT_STORE(v, p) __builtin_nontemporal_store((v), (p))
#endif
#endif

namespace dwx {

constexpr uint32_t kNoVar = 0xFFFFFFFFu;

// ---------------------------------------------------------------- RNG
// Philox4x32-10 (Salmon et al., SC'11).  key = seed, counter = (variable id, sweep).
DWX_DEV void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t &c0, uint32_t &c1, uint32_t &c2,
                           uint32_t &c3) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
}

// two uniforms in [0,1) with 53 random bits each
DWX_DEV void philox_uniforms(uint64_t seed, uint64_t vid, uint64_t sweep, double &A, double &B) {
  uint32_t c0 = (uint32_t)vid, c1 = (uint32_t)(vid >> 32), c2 = (uint32_t)sweep,
           c3 = (uint32_t)(sweep >> 32);
  philox4x32_10((uint32_t)seed, (uint32_t)(seed >> 32), c0, c1, c2, c3);
  uint64_t a = (uint64_t)c0 | ((uint64_t)c1 << 32);
  uint64_t b = (uint64_t)c2 | ((uint64_t)c3 << 32);
  A = (double)(a >> 11) * (1.0 / 9007199254740992.0);
  B = (double)(b >> 11) * (1.0 / 9007199254740992.0);
}

// ---------------------------------------------------------------- math
// src/common.h:118-132
DWX_DEV double logadd(double a, double b) {
  if (a < b) { double t = a; a = b; b = t; }
  else if (a <= b && b <= a) return 0.693147180559945 + a;
  double nd = b - a;
  if (nd < -18.42) return a;
  return a + log1p(exp(nd));
}

DWX_DEV bool is_linear_zero(double x) {
  return x <= LINEAR_ZERO_THRESHOLD && x >= -LINEAR_ZERO_THRESHOLD;
}

// ---------------------------------------------------------------- factor functions
// src/factor.h:94-100: the variable being sampled takes `proposal`, others their
// current assignment on the chain.
DWX_DEV bool vif_sat(const VifRec vf, uint32_t me, uint32_t proposal, const uint32_t *assign) {
  uint32_t val = (vf.vid == me) ? proposal : assign[vf.vid];
  return val == vf.equal_to;
}

// unary factor: the only predicate is on the sampled variable itself
DWX_DEV double unary_sign(uint32_t func, bool s) {
  switch (func) {
    case FUNC_AND: case FUNC_ISTRUE: case FUNC_OR: case FUNC_IMPLY_NATURAL:
      return s ? 1.0 : -1.0;
    case FUNC_EQUAL:
      return 1.0;
    default:  // AND_CATEGORICAL, IMPLY_MLN, LINEAR, RATIO (log2(1+s)), LOGICAL
      return s ? 1.0 : 0.0;
  }
}

// The two FactorToVariable entries of an EDGE_INLINE2 record, decoded from the record
// itself (TILE_INLINE2 tiles; no load).  `me` = device position of the record's owner.  A
// pre-signed record of such a tile decodes to (me, me): harmless, its terms come from the
// record's own fields.
DWX_DEV void decode_inline2(const EdgeRec &r, uint32_t me, VifRec &a, VifRec &b) {
  const bool pre = r.packed & EDGE_PRESIGNED;
  a.vid = (pre || (r.packed & INLINE2_A_IS_OWNER)) ? me : r.aux;
  b.vid = (pre || (r.packed & INLINE2_B_IS_OWNER)) ? me : r.aux;
  a.equal_to = (r.packed >> INLINE2_PRED_A_SHIFT) & INLINE2_PRED_MASK;
  b.equal_to = (r.packed >> INLINE2_PRED_B_SHIFT) & INLINE2_PRED_MASK;
}

// binary factor from its two satisfied bits (a = first predicate, b = second / head)
DWX_DEV double binary_sign(uint32_t func, bool a, bool b) {
  switch (func) {
    case FUNC_AND: case FUNC_ISTRUE: return (a && b) ? 1.0 : -1.0;
    case FUNC_AND_CATEGORICAL: return (a && b) ? 1.0 : 0.0;
    case FUNC_OR: return (a || b) ? 1.0 : -1.0;
    case FUNC_EQUAL: return (a == b) ? 1.0 : -1.0;
    case FUNC_IMPLY_NATURAL: return !a ? 0.0 : (b ? 1.0 : -1.0);
    case FUNC_IMPLY_MLN: return !a ? 1.0 : (b ? 1.0 : 0.0);
    case FUNC_LINEAR: case FUNC_LOGICAL: return (!a || b) ? 1.0 : 0.0;
    default: return (!a || b) ? 1.0 : 0.0;   // FUNC_RATIO: log2(1 + [!a || b])
  }
}

// sign functions of src/factor.h:112-299 (returned as double, before * feature_value)
DWX_DEV double factor_sign(uint32_t func, uint32_t arity, uint32_t aux, const VifRec *vifs,
                           const uint32_t *assign, uint32_t me, uint32_t proposal) {
  if (arity == 1) return unary_sign(func, proposal == aux);
  const VifRec *v = vifs + aux;
  switch (func) {
    case FUNC_AND: case FUNC_ISTRUE: {
      for (uint32_t i = 0; i < arity; ++i) if (!vif_sat(v[i], me, proposal, assign)) return -1.0;
      return 1.0;
    }
    case FUNC_AND_CATEGORICAL: {
      for (uint32_t i = 0; i < arity; ++i) if (!vif_sat(v[i], me, proposal, assign)) return 0.0;
      return 1.0;
    }
    case FUNC_OR: {
      for (uint32_t i = 0; i < arity; ++i) if (vif_sat(v[i], me, proposal, assign)) return 1.0;
      return -1.0;
    }
    case FUNC_EQUAL: {
      const bool first = vif_sat(v[0], me, proposal, assign);
      for (uint32_t i = 1; i < arity; ++i) if (vif_sat(v[i], me, proposal, assign) != first) return -1.0;
      return 1.0;
    }
    case FUNC_IMPLY_MLN: case FUNC_IMPLY_NATURAL: {
      bool body = true;
      for (uint32_t i = 0; i + 1 < arity; ++i) body &= vif_sat(v[i], me, proposal, assign);
      if (!body) return func == FUNC_IMPLY_MLN ? 1.0 : 0.0;
      const bool head = vif_sat(v[arity - 1], me, proposal, assign);
      return func == FUNC_IMPLY_MLN ? (head ? 1.0 : 0.0) : (head ? 1.0 : -1.0);
    }
    default: {  // LINEAR, RATIO, LOGICAL (src/factor.h:244-296)
      const bool head = vif_sat(v[arity - 1], me, proposal, assign);
      double res = (func == FUNC_RATIO) ? 1.0 : 0.0;
      for (uint32_t i = 0; i + 1 < arity; ++i) {
        const bool s = vif_sat(v[i], me, proposal, assign);
        res += ((!s) || head) ? 1.0 : 0.0;
      }
      if (func == FUNC_LINEAR) return res;
      if (func == FUNC_RATIO) return log2(res);
      return res > 0.0 ? 1.0 : 0.0;
    }
  }
}

// NS evaluations of one factor in ONE walk over its variables.  Scenario j: the sampled
// variable takes prop[j], every other variable its assignment on chain arr[j].  The generic
// path needs several per record -- both proposals of a boolean owner; both chains when
// learning; (evidence chain, evidence value) and (free chain, free sample) for the gradient --
// and walking once per evaluation loads every vif entry and neighbour assignment again.
// Same case analysis as factor_sign; s[j] = the sign in scenario j.
//
// Src says where position i's entry and a neighbour's value come from: memory (VifsInMemory:
// any arity, a loop) or registers filled by an earlier, batched load phase (VifsPreloaded:
// arity <= GEN_ARITY, loops unrolled so that every register index is static).
constexpr uint32_t GEN_ARITY = 3;   // positions of a factor the batched generic walk preloads
constexpr uint32_t PROP_OWN = 0xFFFFFFFFu, PROP_OTHER = 0xFFFFFFFEu;   // see factor_signs_from
#ifndef DWX_GEN_BATCH
#define DWX_GEN_BATCH 1
#endif
constexpr uint32_t GEN_BATCH = DWX_GEN_BATCH;   // records per step of the batched generic walk

template <int NS>
struct VifsInMemory {
  static constexpr uint32_t MAXA = 0;
  const VifRec *v;
  const uint32_t *const (&arr)[NS];
  DWX_DEV VifRec vif(uint32_t i) const { return v[i]; }
  DWX_DEV uint32_t value(int j, uint32_t, uint32_t vid) const { return arr[j][vid]; }
};
// (one chain per scenario pair is enough for the preloaded form: chain[j] selects the value row)
template <int NS, int NCHAIN>
struct VifsPreloaded {
  static constexpr uint32_t MAXA = GEN_ARITY;
  const VifRec (&vf)[GEN_ARITY];
  const uint32_t (&val)[NCHAIN][GEN_ARITY];
  const int (&chain)[NS];
  DWX_DEV VifRec vif(uint32_t i) const { return vf[i]; }
  DWX_DEV uint32_t value(int j, uint32_t i, uint32_t) const { return val[chain[j]][i]; }
};

template <int NS, class Src>
DWX_DEV void factor_signs_from(uint32_t func, uint32_t arity, const Src &src, uint32_t me,
                               const uint32_t (&prop)[NS], double (&s)[NS]) {
  constexpr uint32_t MAXA = Src::MAXA;
  const uint32_t n = MAXA ? MAXA : arity;   // (MAXA: constant trip count, positions past the arity skipped)
  // (PROP_OWN / PROP_OTHER: "the owner takes the value its own predicate names" / "any other
  // value" -- what the edge-parallel staging of a categorical tile asks, where a record's
  // proposal is its row's value)
  auto sat = [&](uint32_t i, bool (&a)[NS]) {
    const VifRec vf = src.vif(i);
    const bool mine = vf.vid == me;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const bool own = prop[j] == PROP_OWN || (prop[j] != PROP_OTHER && prop[j] == vf.equal_to);
      a[j] = mine ? own : src.value(j, i, vf.vid) == vf.equal_to;
    }
  };
  bool a[NS];
  switch (func) {
    case FUNC_AND: case FUNC_ISTRUE: case FUNC_AND_CATEGORICAL: {
      bool all[NS];
#pragma unroll
      for (int j = 0; j < NS; ++j) all[j] = true;
#pragma unroll
      for (uint32_t i = 0; i < n; ++i) {
        if (MAXA && i >= arity) continue;
        sat(i, a);
#pragma unroll
        for (int j = 0; j < NS; ++j) all[j] &= a[j];
      }
      const double no = func == FUNC_AND_CATEGORICAL ? 0.0 : -1.0;
#pragma unroll
      for (int j = 0; j < NS; ++j) s[j] = all[j] ? 1.0 : no;
      return;
    }
    case FUNC_OR: {
      bool any[NS];
#pragma unroll
      for (int j = 0; j < NS; ++j) any[j] = false;
#pragma unroll
      for (uint32_t i = 0; i < n; ++i) {
        if (MAXA && i >= arity) continue;
        sat(i, a);
#pragma unroll
        for (int j = 0; j < NS; ++j) any[j] |= a[j];
      }
#pragma unroll
      for (int j = 0; j < NS; ++j) s[j] = any[j] ? 1.0 : -1.0;
      return;
    }
    case FUNC_EQUAL: {
      bool first[NS], eq[NS];
      sat(0, first);
#pragma unroll
      for (int j = 0; j < NS; ++j) eq[j] = true;
#pragma unroll
      for (uint32_t i = 1; i < (MAXA ? MAXA : arity); ++i) {
        if (MAXA && i >= arity) continue;
        sat(i, a);
#pragma unroll
        for (int j = 0; j < NS; ++j) eq[j] &= a[j] == first[j];
      }
#pragma unroll
      for (int j = 0; j < NS; ++j) s[j] = eq[j] ? 1.0 : -1.0;
      return;
    }
    case FUNC_IMPLY_MLN: case FUNC_IMPLY_NATURAL: {
      bool body[NS], head[NS];
#pragma unroll
      for (int j = 0; j < NS; ++j) { body[j] = true; head[j] = false; }
#pragma unroll
      for (uint32_t i = 0; i < n; ++i) {
        if (MAXA && i >= arity) continue;
        sat(i, a);
        const bool is_head = i + 1 == arity;
#pragma unroll
        for (int j = 0; j < NS; ++j) { if (is_head) head[j] = a[j]; else body[j] &= a[j]; }
      }
#pragma unroll
      for (int j = 0; j < NS; ++j)
        s[j] = func == FUNC_IMPLY_MLN ? (!body[j] ? 1.0 : (head[j] ? 1.0 : 0.0))
                                      : (!body[j] ? 0.0 : (head[j] ? 1.0 : -1.0));
      return;
    }
    default: {  // LINEAR, RATIO, LOGICAL (src/factor.h:244-296)
      bool head[NS];
      if (MAXA) {
#pragma unroll
        for (int j = 0; j < NS; ++j) head[j] = false;
#pragma unroll
        for (uint32_t i = 0; i < MAXA; ++i) {
          if (i + 1 != arity) continue;
          sat(i, a);
#pragma unroll
          for (int j = 0; j < NS; ++j) head[j] = a[j];
        }
      } else {
        sat(arity - 1, head);
      }
      double r[NS];
#pragma unroll
      for (int j = 0; j < NS; ++j) r[j] = (func == FUNC_RATIO) ? 1.0 : 0.0;
#pragma unroll
      for (uint32_t i = 0; i < n; ++i) {
        if (i + 1 >= arity) continue;
        sat(i, a);
#pragma unroll
        for (int j = 0; j < NS; ++j) r[j] += ((!a[j]) || head[j]) ? 1.0 : 0.0;
      }
#pragma unroll
      for (int j = 0; j < NS; ++j)
        s[j] = func == FUNC_LINEAR ? r[j] : (func == FUNC_RATIO ? log2(r[j]) : (r[j] > 0.0 ? 1.0 : 0.0));
      return;
    }
  }
}

template <int NS>
DWX_DEV void factor_signs(uint32_t func, uint32_t arity, uint32_t aux, const VifRec *vifs, uint32_t me,
                          const uint32_t *const (&arr)[NS], const uint32_t (&prop)[NS], double (&s)[NS]) {
  if (arity == 1) {
#pragma unroll
    for (int j = 0; j < NS; ++j) s[j] = unary_sign(func, prop[j] == aux);
    return;
  }
  const VifsInMemory<NS> src{vifs + aux, arr};
  factor_signs_from<NS>(func, arity, src, me, prop, s);
}

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
enum { W_GLOBAL = 0, W_ARRAY = 1, W_INRECORD = 2, W_TERMS = 3, W_TERMS8 = 4, W_COOP = 5, W_COOPB = 6, W_PRESUM = 7, W_LREC = 8 };
constexpr uint32_t GIANT_PIECE = 8192;     // records per workgroup of a boolean oversized variable
#ifndef DWX_GIANT_THREADS
#define DWX_GIANT_THREADS 1024
#endif
constexpr uint32_t GIANT_THREADS = DWX_GIANT_THREADS;   // lanes per oversized variable (giant kernels)
constexpr uint32_t COOP_U = 4;             // records per lane and step of a cooperative walk

// sum over the 64 lanes of a wave, the same value (and the same association: the xor
// butterfly) in every lane
// Sums `acc` over the runs of equal `key` among the 64 lanes of a wave (equal keys sit in
// neighbouring lanes); every lane gets the sum from itself to the end of its run, `head` says
// whether it is the first lane of its run.  All 64 lanes call together.
#ifndef DWX_WAVE_SEG_SUM_I64
DWX_DEV long long wave_seg_sum_i64(uint32_t key, long long acc, bool &head) {
  const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
  for (uint32_t off = 1; off < 64u; off <<= 1) {
    const uint32_t ok = (uint32_t)__shfl_down((int)key, off, 64);
    const long long oa = __shfl_down(acc, off, 64);
    if (lane + off < 64u && ok == key) acc += oa;
  }
  const uint32_t pk = (uint32_t)__shfl_up((int)key, 1, 64);
  head = lane == 0u || pk != key;
  return acc;
}
#define DWX_WAVE_SEG_SUM_I64(key, acc, head) wave_seg_sum_i64(key, acc, head)
#endif
// the value of the neighbouring lane (lane ^ 1), both lanes of the pair calling together
#ifndef DWX_PAIR_SWAP_U32
#define DWX_PAIR_SWAP_U32(v) ((uint32_t)__shfl_xor((int)(v), 1, 64))
#endif
#ifndef DWX_WAVE_SUM_F64
DWX_DEV double wave_sum_f64(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
#define DWX_WAVE_SUM_F64(v) wave_sum_f64(v)
#endif
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
template <int WMODE, bool SIMPLE>
DWX_DEV void bool_potentials(const KernelParams &P, const TileView &T, uint32_t row,
                             const uint32_t *assign, uint32_t me, double &pp, double &pn) {
  const uint32_t es = T.rowptr[row - T.row_bias], ee = T.rowptr[row + 1 - T.row_bias];
  pp = 0.0; pn = 0.0;
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

// f32 exp for the guarded fast paths below (v_exp_f32; error ~1e-6 relative for |x| < 30)
#ifndef DWX_FAST_EXPF
#define DWX_FAST_EXPF(x) __expf(x)
#endif
constexpr double DRAW_GUARD = 1e-4;   // >> every f32 error bound below
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

// want_delta (learning, TILE_PULL tiles only): instead of scattering gradient atomics,
// return hit(free) - hit(evid) in {-1,0,+1} for a variable that triggers SGD (0 otherwise).
template <bool LEARN, int WMODE, bool SIMPLE>
DWX_DEV int process_variable(const KernelParams &P, const TileView &T, uint32_t p,
                             const VarPre pre, double A, double B, const bool want_delta = false) {
  const uint32_t meta = pre.meta;
  const bool is_cat = meta & VM_CATEGORICAL;
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
      bool_potentials<WMODE, SIMPLE>(P, T, row0, P.assign_evid, p, pp, pn);
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
    else bool_potentials<WMODE, SIMPLE>(P, T, row0, P.assign_free, p, pp_f, pn_f);
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

// sample_sgd_single_variable (src/gibbs_sampler.h:127-149) + sgd_on_variable
// (src/factor_graph.cc:262-275) for a boolean variable, everything out of LDS.
DWX_DEV void learn_variable_terms2(const KernelParams &P, const uint32_t *rowptr, uint32_t row_bias,
                                   const LearnRec *recs, uint32_t edge_bias, long long *agg,
                                   uint32_t p, const VarPre pre, double A, double B) {
  const bool is_evid = pre.meta & VM_EVIDENCE;
  const bool noise_aware = P.flags & OPT_NOISE_AWARE;
  const uint32_t es = rowptr[pre.row0 - row_bias], ee = rowptr[pre.row0 + 1 - row_bias];
  double ppf = 0.0, pnf = 0.0, ppe = 0.0, pne = 0.0;
  // LEARN_BATCH staged records per step (all LDS reads in flight); sums stay sequential and in
  // row order, a slot past the end adds +0.0 (cannot change a running sum)
  for (uint32_t e = es; e < ee; e += LEARN_BATCH) {
    LearnRec r[LEARN_BATCH];
#pragma unroll
    for (uint32_t u = 0; u < LEARN_BATCH; ++u) r[u] = recs[umin(e + u, ee - 1) - edge_bias];
#pragma unroll
    for (uint32_t u = 0; u < LEARN_BATCH; ++u) {
      const bool in = e + u < ee;
      const double w = (double)r[u].w;
      ppf += in ? w * (double)r[u].sf1 : 0.0; pnf += in ? w * (double)r[u].sf0 : 0.0;
      ppe += in ? w * (double)r[u].se1 : 0.0; pne += in ? w * (double)r[u].se0 : 0.0;
    }
  }
  const uint32_t p_free = bool_draw(A, ppf, pnf);
  P.assign_free[p] = p_free;
  const uint32_t evid_value = pre.init;
  // boolean variables carry no truthiness: sample_evid is "evidence value" or a Gibbs draw
  const uint32_t p_evid = (!noise_aware && is_evid) ? evid_value : bool_draw(B, ppe, pne);
  P.assign_evid[p] = p_evid;
  if (!(P.flags & OPT_LEARN_NON_EVIDENCE) && (noise_aware || !is_evid)) return;
  for (uint32_t e = es; e < ee; ++e) {
    const LearnRec r = recs[e - edge_bias];
    if (r.packed & EDGE_FIXED_FLAG) continue;
    const double pot_free = (double)(p_free ? r.sf1 : r.sf0);
    const double pot_evid = (double)(evid_value ? r.se1 : r.se0);
    const long long gi = llrint(FIX_SCALE * (pot_free - pot_evid));
    long long *dst = agg ? agg : P.grad;
    if (gi) atomicAdd((unsigned long long *)&dst[r.wid], (unsigned long long)gi);
    if (P.flags & OPT_DYNAMIC_T)
      atomicAdd((unsigned long long *)&dst[P.num_weights + r.wid], (unsigned long long)(long long)FIX_SCALE);
  }
}

// The same with TWO lanes per variable (tiles of at most 128 variables -- twelve and more records
// each -- would leave half of the workgroup idle in this phase): lane 2j sums and draws the free
// chain of variable j, lane 2j + 1 its evidence chain -- each sum in row order as before -- they
// swap the free sample, and each takes every other record of the gradient walk.
DWX_DEV void learn_variable_terms2_pair(const KernelParams &P, const uint32_t *rowptr, uint32_t row_bias,
                                        const LearnRec *recs, uint32_t edge_bias, long long *agg,
                                        uint32_t p, const VarPre pre, double A, double B, const uint32_t chain) {
  const bool is_evid = pre.meta & VM_EVIDENCE;
  const bool noise_aware = P.flags & OPT_NOISE_AWARE;
  const uint32_t es = rowptr[pre.row0 - row_bias], ee = rowptr[pre.row0 + 1 - row_bias];
  double pp = 0.0, pn = 0.0;
  for (uint32_t e = es; e < ee; e += LEARN_BATCH) {
    LearnRec r[LEARN_BATCH];
#pragma unroll
    for (uint32_t u = 0; u < LEARN_BATCH; ++u) r[u] = recs[umin(e + u, ee - 1) - edge_bias];
#pragma unroll
    for (uint32_t u = 0; u < LEARN_BATCH; ++u) {
      const bool in = e + u < ee;
      const double w = (double)r[u].w;
      pp += in ? w * (double)(chain ? r[u].se1 : r[u].sf1) : 0.0;
      pn += in ? w * (double)(chain ? r[u].se0 : r[u].sf0) : 0.0;
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
  if (!(P.flags & OPT_LEARN_NON_EVIDENCE) && (noise_aware || !is_evid)) return;
  for (uint32_t e = es + chain; e < ee; e += 2u) {
    const LearnRec r = recs[e - edge_bias];
    if (r.packed & EDGE_FIXED_FLAG) continue;
    const double pot_free = (double)(p_free ? r.sf1 : r.sf0);
    const double pot_evid = (double)(evid_value ? r.se1 : r.se0);
    const long long gi = llrint(FIX_SCALE * (pot_free - pot_evid));
    long long *dst = agg ? agg : P.grad;
    if (gi) atomicAdd((unsigned long long *)&dst[r.wid], (unsigned long long)gi);
    if (P.flags & OPT_DYNAMIC_T)
      atomicAdd((unsigned long long *)&dst[P.num_weights + r.wid], (unsigned long long)(long long)FIX_SCALE);
  }
}

// learning sweep over a boolean TILE_TERMS2 / TILE_TERMS3 tile of at most 128 variables: two
// lanes per variable (workgroup-uniform)
template <bool LEARN, int K, bool WIDE>
DWX_DEV bool chain_pair_tile(const TileDesc &d) {
  return K <= 6 && LEARN && WIDE && (d.flags & (TILE_TERMS2 | TILE_TERMS3)) && !(d.flags & TILE_CATEGORICAL) &&
         !(d.flags & TILE_OUTSIDE) && 2u * d.nv <= BLOCK_THREADS && DWX_CHAIN_PAIRS;
}

// ---------------------------------------------------------------- kernels
#ifndef DWX_DYN_LDS
#define DWX_DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#endif

// The tile's edge records: lane t takes records t, t + 256, ...  (16 B per lane,
// consecutive lanes -> consecutive records: one coalesced stream).  Read through a
// buffer descriptor of exactly the tile's range: the hardware bounds check returns
// zeros for lanes past the last record (no clamping arithmetic, no branch, no memory
// traffic), and the K loads differ only in their scalar offset, so they cost no
// per-load address VALU.  "nt": the stream is read once per sweep and must not evict
// the re-used f32 weight table from the XCD's L2.
#ifndef DWX_LOAD_TILE_RECORDS
typedef uint32_t dwx_u32x4 __attribute__((ext_vector_type(4)));
template <int K>
DWX_DEV void load_tile_records(const EdgeRec *base, uint32_t nedges, uint32_t t, EdgeRec (&rec)[K]) {
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)(nedges * sizeof(EdgeRec)), 0x00020000);
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const dwx_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(
        rsrc, (int)(t * sizeof(EdgeRec)), (int)(k * BLOCK_THREADS * sizeof(EdgeRec)), /*nt*/ 2);
    rec[k].wid = v.x; rec[k].aux = v.y; rec[k].packed = v.z; rec[k].fval = bits_to_float(v.w);
  }
}
#define DWX_LOAD_TILE_RECORDS(K, base, nedges, t, rec) load_tile_records<K>(base, nedges, t, rec)
#endif

// The same stream for the 8-byte records of an all-TILE_SIMPLE graph (buffer_load_dwordx2).
#ifndef DWX_LOAD_TILE_RECORDS8
typedef uint32_t dwx_u32x2 __attribute__((ext_vector_type(2)));
template <int K>
DWX_DEV void load_tile_records8(const EdgeRec8 *base, uint32_t nedges, uint32_t t, EdgeRec8 (&rec)[K]) {
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)(nedges * sizeof(EdgeRec8)), 0x00020000);
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const dwx_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(
        rsrc, (int)(t * sizeof(EdgeRec8)), (int)(k * BLOCK_THREADS * sizeof(EdgeRec8)), /*nt*/ 2);
    rec[k].key = v.x; rec[k].f = bits_to_float(v.y);
  }
}
#define DWX_LOAD_TILE_RECORDS8(K, base, nedges, t, rec) load_tile_records8<K>(base, nedges, t, rec)
#endif

// An 8-byte record back in its 16-byte pre-signed form (a zero-filled lane past the tile's
// end decodes to weight 0 and two -0.0f: staged, never read).
DWX_DEV float rec8_signed(uint32_t code, float f) {   // code = sign + 1
  const uint32_t b = float_to_bits(f);
  return bits_to_float(code == 1u ? 0u : (code == 0u ? (b ^ 0x80000000u) : b));
}
DWX_DEV EdgeRec expand_record(const EdgeRec8 &c) {
  EdgeRec r;
  r.wid = c.key & REC8_WID_MASK;
  r.fval = rec8_signed((c.key >> REC8_HIT_SHIFT) & 3u, c.f);
  r.aux = float_to_bits(rec8_signed((c.key >> REC8_MISS_SHIFT) & 3u, c.f));
  r.packed = EDGE_PRESIGNED | (1u << EDGE_ARITY_SHIFT) | ((c.key & REC8_FIXED) ? EDGE_FIXED_FLAG : 0u);
  return r;
}
DWX_DEV EdgeRec expand_record(const EdgeRec &r) { return r; }

// Everything a lane holds in registers for the tile it will stage next.
template <int K, class Rec = EdgeRec, int RP = (int)ROWPTR_UNROLL>
struct TilePrefetch {
  Rec rec[K];
  uint32_t rp[RP];     // row pointers t, t + 256, ... of the tile
  VarPre pre;
};

// The descriptor is workgroup-uniform: keep it in scalar registers.  Loading (vector
// registers, no wait) and scalarising (needs the data) are separate steps so that the
// load of the descriptor two tiles ahead can stay in flight across a whole tile.
#ifndef DWX_UNIFORM
#define DWX_UNIFORM(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#endif
#ifndef DWX_BALLOT
#define DWX_BALLOT(pred) ((unsigned long long)__ballot(pred))
#endif
DWX_DEV TileDesc scalarise(const TileDesc &v) {
  TileDesc d;
  d.v0 = DWX_UNIFORM(v.v0); d.nv = DWX_UNIFORM(v.nv);
  d.r0 = DWX_UNIFORM(v.r0); d.nrows = DWX_UNIFORM(v.nrows);
  d.e0 = DWX_UNIFORM(v.e0); d.nedges = DWX_UNIFORM(v.nedges);
  d.flags = DWX_UNIFORM(v.flags); d.pad1 = 0;
  return d;
}

// (TILE_GIANT / TILE_WIDE: one variable, processed by giant_kernel / wide_kernel)
DWX_DEV bool tile_fits(const KernelParams &, const TileDesc &d) {
  return !(d.flags & TILE_OUTSIDE);
}

// Branch-free on purpose: a predicated load compiles to a divergent branch with an
// s_waitcnt vmcnt(0) behind it, which serialises the loads.  Every lane therefore
// always loads -- out-of-range lanes get zero records from the buffer bounds check and
// re-load the tile's last row pointer / variable (the arrays carry one padding
// element) -- and all K + ROWPTR_UNROLL + 4 loads of a lane are in flight together.
template <bool LEARN, int K>
DWX_DEV void issue_record_loads(const KernelParams &P, const TileDesc &d, uint32_t t, EdgeRec (&rec)[K]) {
  // inference over an all-unary tile whose potential terms are already tabulated
  // (edge_terms, same 16-byte stride): stream those instead of the records
  const EdgeRec *stream = (!LEARN && P.edge_terms && (d.flags & (TILE_SIMPLE | TILE_INLINE2))) ? (const EdgeRec *)P.edge_terms : P.edges;
  DWX_LOAD_TILE_RECORDS(K, stream + d.e0, d.nedges, t, rec);
}
template <bool LEARN, int K>
DWX_DEV void issue_record_loads(const KernelParams &P, const TileDesc &d, uint32_t t, EdgeRec8 (&rec)[K]) {
  // (an inference sweep on the 8-byte terms table streams that instead: same stride)
  const EdgeRec8 *stream = (!LEARN && P.edge_terms) ? (const EdgeRec8 *)P.edge_terms : P.edges8;
  DWX_LOAD_TILE_RECORDS8(K, stream + d.e0, d.nedges, t, rec);
}

// pair: two lanes per variable (lanes 2j and 2j + 1 take variable j: chain_pair_tile)
template <bool LEARN, int K, bool NT = true, class Rec, int RP>
DWX_DEV void issue_tile_loads(const KernelParams &P, const TileDesc &d, uint32_t t,
                              TilePrefetch<K, Rec, RP> &f, const bool pair = false) {
  issue_record_loads<LEARN, K>(P, d, t, f.rec);
#pragma unroll
  for (uint32_t k = 0; k < (uint32_t)RP; ++k) {
    const uint32_t *rp = &P.row_ptr[d.r0 + umin(t + k * BLOCK_THREADS, d.nrows)];
    f.rp[k] = NT ? DWX_NT_LOAD(rp) : *rp;
  }
  f.pre = load_var_pre<LEARN, NT>(P, d.v0 + umin(pair ? t >> 1 : t, d.nv - 1));
}

// Edge-parallel evaluation of a tile's staged records whose factors have arity <= GEN_ARITY
// (TILE_TERMS3): the lane's K records in three batched phases -- all their first three
// factor->variable entries, then those variables' assignments on every chain, then the sign
// functions on registers -- in NS scenarios at once (as coop_for_records).  out(k, term[NS]).
template <int K, int NS, int NCHAIN, class Out>
DWX_DEV void stage_generic_records(const KernelParams &P, const TileDesc &d, const EdgeRec (&rec)[K],
                                   const uint32_t *const (&chains)[NCHAIN], const int (&chain)[NS],
                                   const uint32_t (&prop)[NS], const bool (&hit)[NS], Out &&out) {
  VifRec vf[K][GEN_ARITY];
  uint32_t val[K][NCHAIN][GEN_ARITY];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const bool generic = !(rec[k].packed & EDGE_PRESIGNED);
    const uint32_t ar = generic ? edge_arity(rec[k]) : 1u, base = (generic && ar >= 2u) ? rec[k].aux : 0u;
#pragma unroll
    for (uint32_t i = 0; i < GEN_ARITY; ++i) vf[k][i] = P.vifs[base + umin(i, ar - 1u)];
  }
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int c = 0; c < NCHAIN; ++c)
#pragma unroll
      for (uint32_t i = 0; i < GEN_ARITY; ++i) val[k][c][i] = chains[c][vf[k][i].vid];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const EdgeRec r = rec[k];
    double term[NS];
    if (r.packed & EDGE_PRESIGNED) {
#pragma unroll
      for (int j = 0; j < NS; ++j) term[j] = (double)(hit[j] ? r.fval : bits_to_float(r.aux));
    } else {
      const uint32_t me = d.v0 + edge_owner_lane(r);
      double sg[NS];
      const VifsPreloaded<NS, NCHAIN> src{vf[k], val[k], chain};
      factor_signs_from<NS>(edge_func(r), edge_arity(r), src, me, prop, sg);
#pragma unroll
      for (int j = 0; j < NS; ++j) term[j] = sg[j] * (double)r.fval;
    }
    out(k, term);
  }
}

// Persistent, software-pipelined sweep: workgroup b handles tiles b, b + gridDim.x, ...
// of the launch.  While a tile is processed out of LDS, the NEXT tile's edge records,
// row pointers and per-variable inputs are already in flight into registers, so the
// HBM latency of a tile hides behind the previous tile's arithmetic; the Philox draw
// of a tile is computed under the latency of its weight gathers.  K = records staged
// per lane (LDS holds K * 256 records).  Oversized variables are skipped here and
// handled by giant_kernel.
// WIDE (learning only): the graph has TILE_TERMS2 tiles; their records are staged as
// 32-byte LearnRec (LDS doubles, one workgroup per CU, so registers are plentiful).
template <bool LEARN, int K, bool WIDE = false>
// (the learning kernel's LDS footprint admits 2 workgroups per CU at K = 12: give the
// register allocator the matching budget instead of spilling at the 3-per-CU limit)
__global__ void __launch_bounds__(BLOCK_THREADS, WIDE ? (K <= 6 ? 3 : 1) : (LEARN ? (K <= 6 ? 3 : 2) : 3)) sweep_kernel(const KernelParams P) {
  DWX_DYN_LDS(dyn_lds);
  uint32_t *s_rowptr = (uint32_t *)dyn_lds;
  double *s_pot = (double *)(dyn_lds + P.lds_pot_off);
  EdgeRec *s_edges = (EdgeRec *)(dyn_lds + P.lds_edge_off);
  float *s_w = (float *)(dyn_lds + P.lds_w_off);
  long long *s_agg = (LEARN && P.lds_agg_off) ? (long long *)(dyn_lds + P.lds_agg_off) : nullptr;
  const uint32_t t = threadIdx.x;
  constexpr int WMODE = LEARN ? W_ARRAY : W_INRECORD;
  uint32_t tile = P.tile_begin + blockIdx.x;
  if (tile >= P.tile_end) return;
  const uint32_t stride = gridDim.x;
  TileDesc d = scalarise(P.tiles[tile]);
  uint32_t next = tile + stride;
  bool has_next = next < P.tile_end;
  TileDesc dn = scalarise(P.tiles[has_next ? next : tile]);   // one descriptor ahead
  TilePrefetch<K> f;
  issue_tile_loads<LEARN, K>(P, d, t, f, chain_pair_tile<LEARN, K, WIDE>(d));
  if (s_agg) {   // the first __syncthreads of the loop orders this before any use
    for (uint32_t i = t; i < 2 * P.num_weights; i += BLOCK_THREADS) s_agg[i] = 0;
  }
  for (;;) {
    const bool fits = tile_fits(P, d);   // workgroup-uniform
    const VarPre pre = f.pre;
    double A = 0.0, B = 0.0;
    // tabulated terms (see issue_tile_loads): nothing to gather, nothing to multiply
    const bool tabulated = !LEARN && P.edge_terms && (d.flags & (TILE_SIMPLE | TILE_INLINE2));   // workgroup-uniform
    // learning, pull-gradient tile (all-unary boolean, no gradient scatter): the compute
    // phase needs only the records' potential terms, exactly as an inference sweep does
    const bool pull = LEARN && tile_fits(P, d) && (d.flags & TILE_PULL) && !(P.flags & OPT_NO_PULL);   // uniform
    if (fits) {
      // gather the f32 sampling weight of every record this lane staged ...
      const EdgeRec (&rec)[K] = f.rec;
      float w[K];
      if (!tabulated) {
#pragma unroll
        for (int k = 0; k < K; ++k) w[k] = P.w32[rec[k].wid];
      } else {
#pragma unroll
        for (int k = 0; k < K; ++k) w[k] = 0.0f;
      }
      // ... and draw this lane's uniforms while the gathers are in flight
      philox_uniforms(P.seed, P.vid_offset + pre.orig, P.sweep, A, B);
      // unconditional LDS writes: slots past the tile's last record receive copies of
      // it and are never read
      // (TILE_TERMS2 staging keeps two vif records and the neighbour values per staged
      // record live: only instantiated for K <= 6; the host clears the flag for K = 12)
      if (K <= 6 && LEARN && WIDE && (d.flags & TILE_TERMS3)) {
        // arity <= 3: the four sign * feature value products of every record (free / evidence
        // chain x proposal 1 / 0) through the general sign functions on batched loads
        LearnRec *s_lrec = (LearnRec *)s_edges;
        const uint32_t *const chains[2] = {P.assign_free, P.assign_evid};
        const int chain[4] = {0, 0, 1, 1};
        const bool cat = d.flags & TILE_CATEGORICAL;   // (a record's proposal is its row's value)
        const uint32_t p1 = cat ? PROP_OWN : 1u, p0 = cat ? PROP_OTHER : 0u;
        const uint32_t prop[4] = {p1, p0, p1, p0};
        const bool hit[4] = {true, false, true, false};
        stage_generic_records<K, 4, 2>(P, d, rec, chains, chain, prop, hit, [&](int k, const double (&term)[4]) {
          LearnRec lr;
          lr.wid = rec[k].wid; lr.packed = rec[k].packed; lr.w = w[k]; lr.pad = 0;
          lr.sf1 = (float)term[0]; lr.sf0 = (float)term[1]; lr.se1 = (float)term[2]; lr.se0 = (float)term[3];
          s_lrec[t + k * BLOCK_THREADS] = lr;
        });
      } else if (K <= 6 && LEARN && WIDE && (d.flags & TILE_TERMS2)) {
        LearnRec *s_lrec = (LearnRec *)s_edges;
        VifRec va[K], vb[K];
        if (d.flags & TILE_INLINE2) {   // workgroup-uniform
#pragma unroll
          for (int k = 0; k < K; ++k) decode_inline2(rec[k], d.v0 + edge_owner_lane(rec[k]), va[k], vb[k]);
        } else {
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const bool bin = !(rec[k].packed & EDGE_PRESIGNED);
            const VifRec *vp = P.vifs + (bin ? rec[k].aux : 0u);
            va[k] = vp[0]; vb[k] = vp[1];
          }
        }
        uint32_t of[K], oe[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const uint32_t me = d.v0 + edge_owner_lane(rec[k]);
          const VifRec o = (va[k].vid == me) ? vb[k] : va[k];
          of[k] = P.assign_free[o.vid];
          oe[k] = P.assign_evid[o.vid];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const EdgeRec r = rec[k];
          LearnRec lr;
          lr.wid = r.wid; lr.packed = r.packed; lr.w = w[k]; lr.pad = 0;
          if (r.packed & EDGE_PRESIGNED) {
            lr.sf1 = lr.se1 = r.fval;
            lr.sf0 = lr.se0 = bits_to_float(r.aux);
          } else {
            const uint32_t me = d.v0 + edge_owner_lane(r);
            const bool a_me = va[k].vid == me, b_me = vb[k].vid == me;
            const bool a1 = va[k].equal_to == 1u, a0 = va[k].equal_to == 0u;
            const bool b1 = vb[k].equal_to == 1u, b0 = vb[k].equal_to == 0u;
            const bool af = of[k] == va[k].equal_to, bf = of[k] == vb[k].equal_to;
            const bool ae = oe[k] == va[k].equal_to, be = oe[k] == vb[k].equal_to;
            const uint32_t fn = edge_func(r);
            const double fv = (double)r.fval;
            lr.sf1 = (float)(binary_sign(fn, a_me ? a1 : af, b_me ? b1 : bf) * fv);
            lr.sf0 = (float)(binary_sign(fn, a_me ? a0 : af, b_me ? b0 : bf) * fv);
            lr.se1 = (float)(binary_sign(fn, a_me ? a1 : ae, b_me ? b1 : be) * fv);
            lr.se0 = (float)(binary_sign(fn, a_me ? a0 : ae, b_me ? b0 : be) * fv);
          }
          s_lrec[t + k * BLOCK_THREADS] = lr;
        }
      } else if (K <= 6 && !LEARN && (d.flags & TILE_TERMS3)) {
        // inference, arity <= 3: both proposals' terms of every record, edge-parallel
        EdgeTerms *s_terms = (EdgeTerms *)s_edges;
        const uint32_t *const chains[1] = {P.assign_evid};
        const int chain[2] = {0, 0};
        const bool cat = d.flags & TILE_CATEGORICAL;   // (a record's proposal is its row's value; t0 unused)
        const uint32_t prop[2] = {cat ? PROP_OWN : 1u, cat ? PROP_OTHER : 0u};
        const bool hit[2] = {true, false};
        stage_generic_records<K, 2, 1>(P, d, rec, chains, chain, prop, hit, [&](int k, const double (&term)[2]) {
          const double wv = (double)w[k];
          EdgeTerms tt;
          tt.t1 = wv * term[0];
          tt.t0 = wv * term[1];
          s_terms[t + k * BLOCK_THREADS] = tt;
        });
      } else if (K <= 6 && !LEARN && (d.flags & TILE_TERMS2)) {
        // inference, boolean tile with pre-signed and arity-2 records: evaluate every
        // record here.  Three batched phases so that a lane's K vif-pair loads, then its K
        // neighbour-assignment gathers, are all in flight together (inside the per-variable
        // loop they would be 2 dependent round trips per record, serialised).
        EdgeTerms *s_terms = (EdgeTerms *)s_edges;
        if (tabulated) {
          // the stream holds TabRec2 entries (build_terms_kernel): w*f tabulated, the other
          // endpoint inline -- one neighbour gather per record is all that is left
          uint32_t other[K];
#pragma unroll
          for (int k = 0; k < K; ++k) other[k] = P.assign_evid[rec[k].packed];   // TabRec2::other
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const EdgeRec r = rec[k];
            const double wf = u32x2_to_double(r.wid, r.aux);
            const uint32_t bits = float_to_bits(r.fval);
            EdgeTerms tt;
            if (bits & TAB2_UNARY) {
              tt.t1 = (bits & TAB2_C1) ? wf : 0.0;
              const uint32_t c0 = (bits >> TAB2_C0_SHIFT) & 3u;   // 0: -1, 1: 0, 2: +1
              tt.t0 = c0 == 1u ? 0.0 : (c0 == 2u ? wf : -wf);
            } else {
              const uint32_t pa = (bits >> INLINE2_PRED_A_SHIFT) & INLINE2_PRED_MASK;
              const uint32_t pb = (bits >> INLINE2_PRED_B_SHIFT) & INLINE2_PRED_MASK;
              const bool a_me = bits & INLINE2_A_IS_OWNER, b_me = bits & INLINE2_B_IS_OWNER;
              const bool a_o = other[k] == pa, b_o = other[k] == pb;
              const bool a1 = a_me ? (pa == 1u) : a_o, b1 = b_me ? (pb == 1u) : b_o;
              const bool a0 = a_me ? (pa == 0u) : a_o, b0 = b_me ? (pb == 0u) : b_o;
              const uint32_t fn = bits & EDGE_FUNC_MASK;
              tt.t1 = binary_sign(fn, a1, b1) * wf;
              tt.t0 = binary_sign(fn, a0, b0) * wf;
            }
            s_terms[t + k * BLOCK_THREADS] = tt;
          }
        } else {
        VifRec va[K], vb[K];
        if (d.flags & TILE_INLINE2) {   // workgroup-uniform
#pragma unroll
          for (int k = 0; k < K; ++k) decode_inline2(rec[k], d.v0 + edge_owner_lane(rec[k]), va[k], vb[k]);
        } else {
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const bool bin = !(rec[k].packed & EDGE_PRESIGNED);
            const VifRec *vp = P.vifs + (bin ? rec[k].aux : 0u);   // padded: always in bounds
            va[k] = vp[0]; vb[k] = vp[1];
          }
        }
        uint32_t other[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const uint32_t me = d.v0 + edge_owner_lane(rec[k]);
          // the neighbour: the position that is not the owner (if both are the owner, any)
          const VifRec o = (va[k].vid == me) ? vb[k] : va[k];
          other[k] = P.assign_evid[o.vid];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const EdgeRec r = rec[k];
          const double wv = (double)w[k];
          EdgeTerms tt;
          if (r.packed & EDGE_PRESIGNED) {
            tt.t1 = wv * (double)r.fval;
            tt.t0 = wv * (double)bits_to_float(r.aux);
          } else {
            const uint32_t me = d.v0 + edge_owner_lane(r);
            const bool a_me = va[k].vid == me, b_me = vb[k].vid == me;
            const bool a_o = other[k] == va[k].equal_to, b_o = other[k] == vb[k].equal_to;
            // satisfied bits under proposal x: own positions compare x with their predicate
            const bool a1 = a_me ? (va[k].equal_to == 1u) : a_o, b1 = b_me ? (vb[k].equal_to == 1u) : b_o;
            const bool a0 = a_me ? (va[k].equal_to == 0u) : a_o, b0 = b_me ? (vb[k].equal_to == 0u) : b_o;
            const uint32_t fn = edge_func(r);
            const double fv = (double)r.fval;
            tt.t1 = wv * (binary_sign(fn, a1, b1) * fv);
            tt.t0 = wv * (binary_sign(fn, a0, b0) * fv);
          }
          s_terms[t + k * BLOCK_THREADS] = tt;
        }
        }
      } else if (LEARN ? pull : (bool)(d.flags & TILE_SIMPLE)) {
        // inference, all-unary tile: do the per-record arithmetic here, edge-parallel
        // and straight-line, and stage the two potential terms instead of the record:
        // t1 = w * (sign(hit) * f), t0 = w * (sign(miss) * f), the sign already folded
        // into the record by the host.  Same products as FactorGraph::potential.
        EdgeTerms *s_terms = (EdgeTerms *)s_edges;
        if (tabulated) {
          // the stream already holds these very products (build_terms_kernel): copy through
#pragma unroll
          for (int k = 0; k < K; ++k) s_edges[t + k * BLOCK_THREADS] = rec[k];
        } else {
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const EdgeRec r = rec[k];
            const double wv = (double)w[k];
            EdgeTerms tt;
            tt.t1 = wv * (double)r.fval;                 // proposal hits
            tt.t0 = wv * (double)bits_to_float(r.aux);   // proposal misses
            s_terms[t + k * BLOCK_THREADS] = tt;
          }
        }
      } else {
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const uint32_t i = t + k * BLOCK_THREADS;
          EdgeRec r = rec[k];
          if (LEARN) { s_w[i] = w[k]; } else { r.wid = float_to_bits(w[k]); }
          s_edges[i] = r;
        }
      }
#pragma unroll
      for (uint32_t k = 0; k < ROWPTR_UNROLL; ++k) s_rowptr[t + k * BLOCK_THREADS] = f.rp[k];
      for (uint32_t i = t + ROWPTR_UNROLL * BLOCK_THREADS; i <= d.nrows; i += BLOCK_THREADS)
        s_rowptr[i] = P.row_ptr[d.r0 + i];
      __syncthreads();
    }
    // Prefetch, unconditionally and branch-free (a conditional prefetch makes the
    // compiler copy the freshly loaded registers right behind the loads, i.e. wait for
    // them): first the descriptor two tiles ahead (vector registers; scalarised only at
    // the bottom of the loop), then the next tile's records / row pointers / variable
    // inputs.  Past the last tile the "next tile" is an empty one at the launch's first
    // record: K loads of one cached line.
    const uint32_t nn = next + stride;
    const bool has_nn = has_next && nn < P.tile_end;
    const TileDesc raw_nn = P.tiles[has_nn ? nn : tile];
    TileDesc dl = dn;
    if (!has_next) { dl.nedges = 0; dl.nrows = 0; dl.nv = 1; }
    issue_tile_loads<LEARN, K>(P, dl, t, f, has_next && chain_pair_tile<LEARN, K, WIDE>(dl));
    // process the current tile out of LDS
    int delta = 0;
    if (fits && chain_pair_tile<LEARN, K, WIDE>(d)) {
      if (t < 2u * d.nv)
        learn_variable_terms2_pair(P, s_rowptr, d.r0, (const LearnRec *)s_edges, d.e0, s_agg, d.v0 + (t >> 1), pre, A, B, t & 1u);
    } else if (fits && t < d.nv) {
      TileView T{s_rowptr, d.r0, s_edges, d.e0, s_w, s_agg, P.lds_pot_off ? s_pot : nullptr};
      if (K <= 6 && LEARN && WIDE && (d.flags & TILE_TERMS3) && (d.flags & TILE_CATEGORICAL))
        process_variable<LEARN, W_LREC, false>(P, T, d.v0 + t, pre, A, B);
      else if (K <= 6 && LEARN && WIDE && (d.flags & (TILE_TERMS2 | TILE_TERMS3)))
        learn_variable_terms2(P, s_rowptr, d.r0, (const LearnRec *)s_edges, d.e0, s_agg, d.v0 + t, pre, A, B);
      else if (LEARN && pull)   // the staged records ARE terms: sgd_row is never reached (want_delta)
        delta = process_variable<LEARN, W_TERMS, true>(P, T, d.v0 + t, pre, A, B, true);
      else if ((d.flags & TILE_SIMPLE) || (K <= 6 && !LEARN && (d.flags & (TILE_TERMS2 | TILE_TERMS3))))
        delta = process_variable<LEARN, LEARN ? W_ARRAY : W_TERMS, true>(P, T, d.v0 + t, pre, A, B, false);
      else
        process_variable<LEARN, WMODE, false>(P, T, d.v0 + t, pre, A, B);
    }
    if (pull) {
      // every wave of the tile publishes its two ballots (also when all zero: the words
      // are rewritten each learning sweep, so nothing needs clearing)
      const unsigned long long nz = DWX_BALLOT(delta != 0), ng = DWX_BALLOT(delta < 0);
      if ((t & 63u) == 0) {
        unsigned long long *w = P.delta + ((size_t)tile * 4 + (t >> 6)) * 2;
        DWX_NT_STORE(nz, &w[0]); DWX_NT_STORE(ng, &w[1]);
      }
    }
    if (!has_next) break;
    __syncthreads();   // LDS is rewritten by the next iteration
    d = dn;
    dn = scalarise(raw_nn);
    tile = next; next = nn; has_next = has_nn;
  }
  if (s_agg) {   // one flush per persistent workgroup
    __syncthreads();
    for (uint32_t i = t; i < 2 * P.num_weights; i += BLOCK_THREADS) {
      const long long v = s_agg[i];
      if (v) atomicAdd((unsigned long long *)&P.grad[i], (unsigned long long)v);
    }
  }
}

// ---------------------------------------------------------------- all-unary graphs
// sweep8_kernel: the sweep of a graph whose EVERY tile is TILE_SIMPLE (all factors unary,
// f32-exact feature values).  Same tiles, same pipeline, same LDS image and the same
// per-variable code as sweep_kernel, but the record stream is P.edges8 (8 bytes per record:
// half the stream, half the prefetch registers) and none of the non-unary variants exists.
// Never launched with the terms table (that run streams 16-byte terms through sweep_kernel).
//
// What bounds it (config 3, 1 M weights): not HBM and not the L2 request rate but the CU's
// vector L1 (TCP): every record costs one uncoalesced 4-byte gather that misses L1 -- a tile
// of 2 560 records takes ~4.7 k cycles per CU when the table is L1-resident (1 000 weights)
// and ~9.8 k when every gather goes to L2 (1 M weights), whatever the occupancy and however
// early the gathers are issued (a two-tile-deep version of this loop, gathers of tile j+1
// and records of tile j+2 in flight under the compute of tile j, ran 5 % SLOWER).
// TAB (inference on unchanged weights, from the second consecutive sweep on): the stream is
// the 8-byte terms table of build_terms8_kernel -- per record the exact f64 product w * f with
// the two sign codes in its four lowest mantissa bits (the product of two f32 has at most 48
// significant bits: at least five trailing zeros) -- and no weight is gathered at all.
// RP: row pointers prefetched per lane.  Boolean tiles have 257; a categorical tile has up to
// rcap + 1 = 1537 -- with RP = 7 all of them ride the prefetch instead of being loaded and
// awaited while staging (config 4: 0.296 -> 0.246 ms); boolean graphs keep 2 (the extra
// loads cost config 3's repeated inference 15 %).
// (3 workgroups per CU also when learning: pull-gradient tiles stage 16-byte terms only)
#ifndef DWX_S8_INFER_WG
#define DWX_S8_INFER_WG 3
#endif
#ifndef DWX_S8_LEARN_WG
#define DWX_S8_LEARN_WG 4
#endif
template <bool LEARN, int K, bool TAB = false, int RP = (int)ROWPTR_UNROLL>
__global__ void __launch_bounds__(BLOCK_THREADS, TAB ? 4 : (LEARN ? DWX_S8_LEARN_WG : DWX_S8_INFER_WG)) sweep8_kernel(const KernelParams P) {
  static_assert(!(LEARN && TAB), "the terms table serves inference sweeps only");
  DWX_DYN_LDS(dyn_lds);
  uint32_t *s_rowptr = (uint32_t *)dyn_lds;
  double *s_pot = (double *)(dyn_lds + P.lds_pot_off);
  EdgeRec *s_edges = (EdgeRec *)(dyn_lds + P.lds_edge_off);
  float *s_w = (float *)(dyn_lds + P.lds_w_off);
  long long *s_agg = (LEARN && P.lds_agg_off) ? (long long *)(dyn_lds + P.lds_agg_off) : nullptr;
  const uint32_t t = threadIdx.x;
  uint32_t tile = P.tile_begin + blockIdx.x;
  if (tile >= P.tile_end) return;
  const uint32_t stride = gridDim.x;
  TileDesc d = scalarise(P.tiles[tile]);
  uint32_t next = tile + stride;
  bool has_next = next < P.tile_end;
  TileDesc dn = scalarise(P.tiles[has_next ? next : tile]);   // one descriptor ahead
  TilePrefetch<K, EdgeRec8, RP> f;
  issue_tile_loads<LEARN, K, !TAB>(P, d, t, f);
  if (s_agg) {   // the first __syncthreads of the loop orders this before any use
    for (uint32_t i = t; i < 2 * P.num_weights; i += BLOCK_THREADS) s_agg[i] = 0;
  }
  for (;;) {
    const bool fits = tile_fits(P, d);   // workgroup-uniform
    // learning, pull-gradient tile: the compute phase needs only the potential terms
    const bool pull = LEARN && fits && (d.flags & TILE_PULL) && !(P.flags & OPT_NO_PULL);   // uniform
    const VarPre pre = f.pre;
    double A = 0.0, B = 0.0;
    if (fits) {
      // the f32 sampling weight of every record this lane stages (a zero-filled lane past
      // the tile's end gathers w32[0]: one cached line) ...
      float w[K];
      if (!TAB) {
#pragma unroll
        for (int k = 0; k < K; ++k) w[k] = P.w32[f.rec[k].key & REC8_WID_MASK];
      }
      // ... and this lane's uniforms while the gathers are in flight
      philox_uniforms(P.seed, P.vid_offset + pre.orig, P.sweep, A, B);
      if (TAB) {
        // the table's entries go to LDS as they are (8 bytes per record: half the staging
        // area of the 16-byte modes, more workgroups per CU); the row walk decodes them
        EdgeRec8 *s_tab = (EdgeRec8 *)s_edges;
#pragma unroll
        for (int k = 0; k < K; ++k) s_tab[t + k * BLOCK_THREADS] = f.rec[k];
      } else if (!LEARN) {
        // the two potential terms of every record, edge-parallel and straight-line:
        // t1 = w * (sign(hit) * f), t0 = w * (sign(miss) * f) -- the products of
        // FactorGraph::potential (src/factor_graph.h:127-145)
        EdgeTerms *s_terms = (EdgeTerms *)s_edges;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const EdgeRec8 c = f.rec[k];
          const double wv = (double)w[k];
          EdgeTerms tt;
          tt.t1 = wv * (double)rec8_signed((c.key >> REC8_HIT_SHIFT) & 3u, c.f);
          tt.t0 = wv * (double)rec8_signed((c.key >> REC8_MISS_SHIFT) & 3u, c.f);
          s_terms[t + k * BLOCK_THREADS] = tt;
        }
      } else if (pull) {
        // learning, pull-gradient tile: the same terms in the terms table's own 8-byte form --
        // the exact f64 product w * f (two f32 factors: at most 48 significant bits, its lowest
        // mantissa bits are zero) with sign(hit) + 1 and sign(miss) + 1 in those bits; the row
        // walk rebuilds t1 = sign(hit) * (w f), t0 = sign(miss) * (w f) exactly (a sign flip is
        // exact).  Half the staging area: a fourth workgroup per CU (config 3: 0.671 -> 0.660 ms;
        // the inference sweep, which walks each row once, is 3 % faster with the 16-byte terms
        // above and gains nothing from a fourth workgroup).
        unsigned long long *s_tab = (unsigned long long *)s_edges;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const EdgeRec8 c = f.rec[k];
          const double wf = (double)w[k] * (double)c.f;
          unsigned long long u;
          __builtin_memcpy(&u, &wf, 8);
          s_tab[t + k * BLOCK_THREADS] = u | ((c.key >> REC8_HIT_SHIFT) & 15u);
        }
      } else {
        // learning with a gradient scatter: the records in their 16-byte form + f32 weights
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const uint32_t i = t + k * BLOCK_THREADS;
          s_w[i] = w[k];
          s_edges[i] = expand_record(f.rec[k]);
        }
      }
#pragma unroll
      for (uint32_t k = 0; k < (uint32_t)RP; ++k) s_rowptr[t + k * BLOCK_THREADS] = f.rp[k];
      for (uint32_t i = t + RP * BLOCK_THREADS; i <= d.nrows; i += BLOCK_THREADS)
        s_rowptr[i] = P.row_ptr[d.r0 + i];
      __syncthreads();
    }
    // prefetch, unconditionally and branch-free (see sweep_kernel): the descriptor two tiles
    // ahead, then the next tile's records / row pointers / variable inputs
    const uint32_t nn = next + stride;
    const bool has_nn = has_next && nn < P.tile_end;
    const TileDesc raw_nn = P.tiles[has_nn ? nn : tile];
    TileDesc dl = dn;
    if (!has_next) { dl.nedges = 0; dl.nrows = 0; dl.nv = 1; }
    issue_tile_loads<LEARN, K, !TAB>(P, dl, t, f);
    // the current tile out of LDS
    int delta = 0;
    if (fits && t < d.nv) {
      TileView T{s_rowptr, d.r0, s_edges, d.e0, s_w, s_agg, P.lds_pot_off ? s_pot : nullptr};
      if (LEARN && pull)   // the staged records ARE terms: sgd_row is never reached (want_delta)
        delta = process_variable<LEARN, W_TERMS8, true>(P, T, d.v0 + t, pre, A, B, true);
      else
        process_variable<LEARN, LEARN ? W_ARRAY : (TAB ? W_TERMS8 : W_TERMS), true>(P, T, d.v0 + t, pre, A, B, false);
    }
    if (pull) {
      const unsigned long long nz = DWX_BALLOT(delta != 0), ng = DWX_BALLOT(delta < 0);
      if ((t & 63u) == 0) {
        unsigned long long *wd = P.delta + ((size_t)tile * 4 + (t >> 6)) * 2;
        DWX_NT_STORE(nz, &wd[0]); DWX_NT_STORE(ng, &wd[1]);
      }
    }
    if (!has_next) break;
    __syncthreads();   // LDS is rewritten by the next iteration
    d = dn;
    dn = scalarise(raw_nn);
    tile = next; next = nn; has_next = has_nn;
  }
  if (s_agg) {   // one flush per persistent workgroup
    __syncthreads();
    for (uint32_t i = t; i < 2 * P.num_weights; i += BLOCK_THREADS) {
      const long long v = s_agg[i];
      if (v) atomicAdd((unsigned long long *)&P.grad[i], (unsigned long long)v);
    }
  }
}

// Variables too big for one tile (rows > rcap or edge records > ecap), e.g. the few
// very-high-degree variables of a power-law graph: ONE WORKGROUP of GIANT_THREADS lanes per such
// variable.  The lanes stride over the variable's records straight from HBM (COOP_U coalesced
// 16-byte loads per lane and step, then their weight gathers, all in flight together), keep
// partial potentials, and an LDS tree hands the totals to every lane (block_sum_all); from
// there all lanes run the code of the tiles (process_variable, W_COOPB) in lockstep: stores and
// tallies once, the gradient rows shared out.  (The partial sums re-associate the f64
// additions: a potential can differ from the sequential sum in its last bits; a decision flips
// only if r*(1+e^x) is within ~1e-16 of 1.)
template <bool LEARN>
__global__ void __launch_bounds__(GIANT_THREADS) giant_kernel(const KernelParams P, const uint32_t *giant_tiles,
                                                              uint32_t n) {
  if (blockIdx.x >= n) return;
  const TileDesc d = P.tiles[giant_tiles[blockIdx.x]];    // nv == 1 by construction
  const uint32_t p = d.v0;
  const VarPre vp = load_var_pre<LEARN>(P, p);
  double A, B;
  philox_uniforms(P.seed, P.vid_offset + vp.orig, P.sweep, A, B);
  const TileView T{P.row_ptr, 0u, P.edges, 0u, nullptr, nullptr, nullptr};
  process_variable<LEARN, W_COOPB, false>(P, T, p, vp, A, B);
}

// BOOLEAN oversized variables, several workgroups each.  One workgroup sits on one CU, and a CU
// turns around one scattered request every ~2.3 cycles (tools/gather_bench): a hub with 10^5
// records kept its workgroup busy for 0.3 ms while the rest of the chip idled, once per colour.
// So the row is cut into pieces of GIANT_PIECE records and the variable handled in three steps:
//   giant_pot_kernel     one workgroup per piece: the piece's share of the four potentials
//                        (free / evidence chain x proposal 1 / 0) -> partial[piece][4]
//   giant_decide_kernel  one lane per variable: adds its pieces' shares IN PIECE ORDER and runs
//                        process_variable<W_PRESUM> -- draws, stores, tallies; the gradient walk
//                        is only decided ({evidence value, free value}) -> decision[variable]
//   giant_grad_kernel    (learning) one workgroup per piece again: sgd_on_factor over the piece
// Categorical oversized variables (a row per value) keep the single workgroup of giant_kernel.
struct GiantPiece { uint32_t slot, e0, e1, pad; };   // slot: index into the boolean-giant list

template <bool LEARN>
__global__ void __launch_bounds__(GIANT_THREADS)
giant_pot_kernel(const KernelParams P, const uint32_t *bgiant_tiles, const GiantPiece *pieces, uint32_t piece0,
                 uint32_t n, double *partial) {
  if (blockIdx.x >= n) return;
  const GiantPiece pc = pieces[piece0 + blockIdx.x];
  const uint32_t p = P.tiles[bgiant_tiles[pc.slot]].v0;
  const TileView T{P.row_ptr, 0u, P.edges, 0u, nullptr, nullptr, nullptr};
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  if (LEARN) {
    const uint32_t *const chains[2] = {P.assign_free, P.assign_evid};
    const int chain[4] = {0, 0, 1, 1};
    const uint32_t prop[4] = {1u, 0u, 1u, 0u};
    const bool hit[4] = {true, false, true, false};
    coop_for_records<W_COOPB, 4, 2>(P, T, pc.e0, pc.e1, p, chains, chain, prop, hit,
                                    [&](const EdgeRec &, uint32_t, double w, const double (&term)[4]) {
#pragma unroll
                                      for (int j = 0; j < 4; ++j) acc[j] += w * term[j];
                                    });
  } else {
    const uint32_t *const chains[1] = {P.assign_evid};
    const int chain[2] = {0, 0};
    const uint32_t prop[2] = {1u, 0u};
    const bool hit[2] = {true, false};
    coop_for_records<W_COOPB, 2, 1>(P, T, pc.e0, pc.e1, p, chains, chain, prop, hit,
                                    [&](const EdgeRec &, uint32_t, double w, const double (&term)[2]) {
                                      acc[2] += w * term[0];
                                      acc[3] += w * term[1];
                                    });
  }
#pragma unroll
  for (int j = LEARN ? 0 : 2; j < 4; ++j) {
    const double v = block_sum_all(acc[j]);
    if (threadIdx.x == 0) partial[(size_t)(piece0 + blockIdx.x) * 4 + j] = v;
  }
}

template <bool LEARN>
__global__ void __launch_bounds__(BLOCK_THREADS)
giant_decide_kernel(const KernelParams P, const uint32_t *bgiant_tiles, const uint32_t *piece_off, uint32_t slot0,
                    uint32_t n, const double *partial, uint32_t *decision) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t slot = slot0 + i, p = P.tiles[bgiant_tiles[slot]].v0;
  double sums[4] = {0.0, 0.0, 0.0, 0.0};
  for (uint32_t k = piece_off[slot]; k < piece_off[slot + 1]; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) sums[j] += partial[(size_t)k * 4 + j];
  uint32_t *dec = decision + (size_t)slot * 4;
  dec[2] = 0u;
  const VarPre vp = load_var_pre<LEARN>(P, p);
  double A, B;
  philox_uniforms(P.seed, P.vid_offset + vp.orig, P.sweep, A, B);
  TileView T{P.row_ptr, 0u, P.edges, 0u, nullptr, nullptr, nullptr};
  T.presum = sums; T.decision = dec;
  process_variable<LEARN, W_PRESUM, false>(P, T, p, vp, A, B);
}

__global__ void __launch_bounds__(GIANT_THREADS)
giant_grad_kernel(const KernelParams P, const uint32_t *bgiant_tiles, const GiantPiece *pieces, uint32_t piece0,
                  uint32_t n, const uint32_t *decision) {
  if (blockIdx.x >= n) return;
  const GiantPiece pc = pieces[piece0 + blockIdx.x];
  const uint32_t *dec = decision + (size_t)pc.slot * 4;
  if (!(dec[2] & 1u)) return;      // (uniform over the workgroup) this variable does not learn now
  const uint32_t p = P.tiles[bgiant_tiles[pc.slot]].v0;
  const TileView T{P.row_ptr, 0u, P.edges, 0u, nullptr, nullptr, nullptr};
  coop_sgd_range<W_COOPB>(P, T, pc.e0, pc.e1, p, dec[0], dec[1], 1u, 1.0, (dec[2] & 2u) != 0);
}

// Degree bin between the lane-per-variable tiles and giant_kernel: a variable
// with hundreds to thousands of records -- it fits a tile, but one lane would walk its row
// record after record while the 255 other lanes of the workgroup wait at the barrier.  ONE WAVE
// per such variable (four per workgroup, no barrier anywhere): the 64 lanes stride over the
// row's records straight from HBM (coalesced 16-byte loads, all weight gathers of a step in
// flight), the partial potentials meet in a xor butterfly (wave_sum_f64: every lane holds the
// total), and from there all lanes run the very code of the tiles (process_variable) in
// lockstep -- every factor function, both variable types, the evidence chain, noise-aware
// truthiness -- with W_COOP doing stores and tallies once and sharing the gradient rows out.
// As in giant_kernel the potential is a re-associated f64 sum: it can differ from the
// sequential one in its last bits; a draw flips only within ~1e-16 of its threshold.
template <bool LEARN>
__global__ void __launch_bounds__(BLOCK_THREADS) wide_kernel(const KernelParams P, const uint32_t *wide_tiles,
                                                             uint32_t n) {
  const uint32_t idx = blockIdx.x * (BLOCK_THREADS / 64u) + (threadIdx.x >> 6);
  if (idx >= n) return;                                   // (the whole wave leaves)
  const TileDesc d = P.tiles[wide_tiles[idx]];            // nv == 1 by construction
  const uint32_t p = d.v0;
  const VarPre vp = load_var_pre<LEARN>(P, p);
  double A, B;
  philox_uniforms(P.seed, P.vid_offset + vp.orig, P.sweep, A, B);
  const TileView T{P.row_ptr, 0u, P.edges, 0u, nullptr, nullptr, nullptr};
  process_variable<LEARN, W_COOP, false>(P, T, p, vp, A, B);
}

// Pull-based weight gradient for TILE_PULL tiles (replaces their gradient atomics).
// inc_* is the incidence list of every (triggering boolean variable, non-fixed record)
// pair, SORTED BY WEIGHT: inc_wid[i], inc_slot[i] = tile * 256 + lane of the owning
// variable, inc_d[i] = sign(hit)*f - sign(miss)*f of the record (f32-exact).  The
// record's gradient is delta(owner) * inc_d, delta in {-1, 0, +1} read from the ballot
// bit-planes the sweep wrote (2 bits per variable: L2-resident).  A workgroup stages
// 256 * PULL_RUN entries' contributions in LDS (coalesced loads), then every lane sums its
// PULL_RUN consecutive entries and flushes one atomic per weight run -- neighbouring
// lanes hit neighbouring weights.  Integer sums: the result is independent of the order
// and identical to what the per-record atomics would have produced.
struct alignas(16) DeltaPair { unsigned long long nz, ng; };
struct alignas(16) U32x4 { uint32_t v[4]; };
struct alignas(16) F32x4 { float v[4]; };

// One lane owns PULL_RUN consecutive entries (a multiple of 4: 16-byte loads straight
// from HBM; a wave covers one contiguous 4 KiB span per array, every line is consumed
// fully across the lane's loads), gathers their owners' bits (one 16-byte L2 hit each,
// all in flight), sums per weight run in registers and flushes one atomic per run.
// No LDS, no barrier.
__global__ void __launch_bounds__(BLOCK_THREADS)
pull_grad_kernel(const uint32_t *inc_wid, const uint32_t *inc_slot, const float *inc_d,
                 uint32_t n, const unsigned long long *delta, long long *grad) {
  const uint32_t stride = gridDim.x * blockDim.x;
  const uint32_t n_runs = (n + PULL_RUN - 1) / PULL_RUN;   // arrays are padded to a full run
  const uint32_t lane = threadIdx.x & 63u;
  // (whole waves stay in the loop: the lanes meet in a wave-wide sum at its end)
  for (uint32_t r0 = (blockIdx.x * blockDim.x + threadIdx.x) - lane; r0 < n_runs; r0 += stride) {
    const bool valid = r0 + lane < n_runs;
    const uint32_t i0 = (valid ? r0 + lane : n_runs - 1) * PULL_RUN;
    uint32_t key[PULL_RUN], slot[PULL_RUN];
    float dd[PULL_RUN];
#pragma unroll
    for (uint32_t k = 0; k < PULL_RUN / 4; ++k) {
      const U32x4 a = ((const U32x4 *)(inc_wid + i0))[k];
      const U32x4 b = ((const U32x4 *)(inc_slot + i0))[k];
      const F32x4 c = ((const F32x4 *)(inc_d + i0))[k];
#pragma unroll
      for (int j = 0; j < 4; ++j) { key[4 * k + j] = a.v[j]; slot[4 * k + j] = b.v[j]; dd[4 * k + j] = c.v[j]; }
    }
    DeltaPair dp[PULL_RUN];
#pragma unroll
    for (uint32_t k = 0; k < PULL_RUN; ++k) dp[k] = ((const DeltaPair *)delta)[slot[k] >> 6];
    uint32_t cur = key[0];
    long long acc = 0;
#pragma unroll
    for (uint32_t k = 0; k < PULL_RUN; ++k) {
      const unsigned long long bit = 1ull << (slot[k] & 63u);
      long long v = 0;
      if (dp[k].nz & bit) {
        const long long q = llrint(FIX_SCALE * (double)dd[k]);
        v = (dp[k].ng & bit) ? -q : q;
      }
      if (key[k] != cur) {
        if (acc && valid) atomicAdd((unsigned long long *)&grad[cur], (unsigned long long)acc);
        cur = key[k];
        acc = v;
      } else {
        acc += v;
      }
    }
    // The lane's last weight run usually continues in the next lanes (a heavily tied weight
    // spans hundreds of lanes): one atomic per weight and WAVE instead of one per lane -- with
    // 10^3-10^4 weights the per-lane atomics queued up on a few thousand addresses.
    bool head;
    const long long total = DWX_WAVE_SEG_SUM_I64(valid ? cur : 0xFFFFFFFFu, valid ? acc : 0ll, head);
    if (head && total) atomicAdd((unsigned long long *)&grad[cur], (unsigned long long)total);
  }
}

// Block pull: the same sums as pull_grad_kernel without its random L2 gathers and with no
// atomics at all.  The owners of the incidence entries are cut into blocks of <= BP_TILES
// consecutive tiles whose ballot pairs (128 KiB) fit LDS; ell holds, per block and weight,
// BP_ROW * DEPTH entries (host: build_level; what does not fit a row goes through
// pull_grad_kernel).  Workgroup (block b, part p) copies b's ballots into LDS once, then
// streams its share of b's rows -- coalesced 16-byte loads, independent iterations, no
// barrier -- and stores one partial sum per weight; fold_partials_kernel adds the blocks'
// partials into grad.  Integer sums: the result equals pull_grad_kernel's.
// a 16-byte row, read once per sweep: non-temporal
#ifndef DWX_LOAD_ROW_NT
typedef uint32_t dwx_row_u32x4 __attribute__((ext_vector_type(4)));
DWX_DEV U32x4 load_row_nt(const U32x4 *p) {
  const dwx_row_u32x4 v = __builtin_nontemporal_load((const dwx_row_u32x4 *)p);
  U32x4 r;
  r.v[0] = v.x; r.v[1] = v.y; r.v[2] = v.z; r.v[3] = v.w;
  return r;
}
#else
DWX_DEV U32x4 load_row_nt(const U32x4 *p) { return *p; }
#endif

// UNIFORM: every record delta of the graph is the same (one feature value, one factor
// function -- the usual case): its step comes in as an argument instead of an LDS table.
template <int DEPTH, bool UNIFORM>
__global__ void __launch_bounds__(BP_THREADS)
pull_ell_kernel(const U32x4 *__restrict__ ell, const uint32_t *block_tile0, uint32_t parts, const long long *qtab,
                uint32_t n_deltas, uint32_t Wp, const unsigned long long *delta, long long *__restrict__ partial) {
  DWX_DYN_LDS(dyn_lds);
  DeltaPair *s_delta = (DeltaPair *)dyn_lds;
  constexpr uint32_t PAIRS = BP_TILES * 4;                 // ballot pairs per block
  // the deltas' fixed-point steps live in LDS too (a dependent global load inside the loop
  // would wait for every load issued before it: vmcnt retires in order)
  long long *s_q = (long long *)(dyn_lds + PAIRS * sizeof(DeltaPair));
  const uint32_t tid = threadIdx.x;
  const uint32_t b = blockIdx.x / parts, part = blockIdx.x % parts;
  const DeltaPair *src = (const DeltaPair *)delta + (size_t)block_tile0[b] * 4;   // (padded allocation)
  for (uint32_t i = tid; i < PAIRS; i += BP_THREADS) s_delta[i] = src[i];
  if (!UNIFORM)
    for (uint32_t i = tid; i < BP_DELTA_SLOTS; i += BP_THREADS) s_q[i] = i < n_deltas ? qtab[i] : 0;
  const long long q0 = qtab[0];
  const uint32_t *s_words = (const uint32_t *)s_delta;
  __syncthreads();
  // this part's weights: whole groups of BP_THREADS
  const uint32_t groups = Wp / BP_THREADS, per = (groups + parts - 1) / parts;
  const uint32_t g0 = part * per, g1 = g0 + per < groups ? g0 + per : groups;
  const U32x4 *__restrict__ rows = ell + (size_t)b * DEPTH * Wp;
  long long *__restrict__ out = partial + (size_t)b * Wp;
  // BP_UNROLL groups per step: all their row loads are issued before the first is used (the
  // compiler does not hoist them over the stores on its own); past the end the last group is
  // loaded again and not stored
  for (uint32_t g = g0; g < g1; g += BP_UNROLL) {
    U32x4 row[BP_UNROLL][DEPTH];
#pragma unroll
    for (uint32_t u = 0; u < BP_UNROLL; ++u) {
      const uint32_t w = umin(g + u, g1 - 1) * BP_THREADS + tid;
#pragma unroll
      for (int dd = 0; dd < DEPTH; ++dd) row[u][dd] = load_row_nt(&rows[(size_t)dd * Wp + w]);
    }
#pragma unroll
    for (uint32_t u = 0; u < BP_UNROLL; ++u) {
      long long acc = 0;
#pragma unroll
      for (int dd = 0; dd < DEPTH; ++dd) {
#pragma unroll
        for (uint32_t k = 0; k < BP_ROW; ++k) {
          // branch-free: an empty entry decodes to the block's last slot and adds zero.
          // Only the two 32-bit words that hold the owner's bits are read (a ballot pair is
          // {nz lo, nz hi, ng lo, ng hi}), and no step table when all deltas are equal.
          const uint32_t e = row[u][dd].v[k];
          const uint32_t slot = e & BP_SLOT_MASK;
          const uint32_t word = (slot >> 6) * 4u + ((slot >> 5) & 1u);
          const uint32_t nzw = s_words[word], ngw = s_words[word + 2u];
          const long long q = UNIFORM ? q0 : s_q[(e >> BP_SLOT_BITS) & (BP_DELTA_SLOTS - 1)];
          const uint32_t nz = (nzw >> (slot & 31u)) & (e != BP_EMPTY ? 1u : 0u), ng = (ngw >> (slot & 31u)) & 1u;
          const long long t = ng ? -q : q;
          acc += nz ? t : 0;
        }
      }
      if (g + u < g1) DWX_NT_STORE(acc, &out[(g + u) * BP_THREADS + tid]);
    }
  }
}

// grad[w] += sum over blocks of partial[block][w]
__global__ void __launch_bounds__(BLOCK_THREADS)
fold_partials_kernel(const long long *partial, uint32_t n_blocks, uint32_t Wp, uint32_t W, long long *grad) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < W; w += stride) {
    long long acc = 0;
    for (uint32_t b = 0; b < n_blocks; ++b) acc += partial[(size_t)b * Wp + w];
    if (acc) grad[w] += acc;
  }
}

// Batched InferenceResult::update_weight (src/inference_result.h:66-85): apply one
// mini-batch's accumulated gradient to every non-fixed weight that received updates, then
// clear the accumulators.  T = dynamic counts (atomics) + static counts (boolean variables,
// precomputed per chunk; null when the plan counts dynamically).
//
// The reference applies its T updates of a weight one after the other, each seeing the
// samples the previous ones already moved: over one batch the weight follows the flow
//   dw/dtau = -(G(w) + reg * T * w),  tau in [0, stepsize]
// and therefore never overshoots, however many factors share the weight.  One plain step
// w -= stepsize * (G + reg T w) does (it diverges once stepsize * curvature > 2).  So the
// batch is integrated instead: with G linearised around the current weight with slope h[w]
// (t_hess: the batch's Gershgorin curvature bound of this weight), the flow's
// end point is
//   w - s * (G + reg T w),   s = (1 - exp(-c stepsize)) / c,   c = h[w] + reg T.
// s -> stepsize for c stepsize << 1 (weights with few factors: the reference's own step, to
// first order w / (1 + reg stepsize)^T - stepsize G), s -> 1 / c for heavily tied weights
// (the flow has converged within the batch).  L1 (the reference adds reg * (w < 0) per
// update, not scaled by the step) keeps its form; only the gradient step saturates.
// Also refreshes the f32 sampling copy of each weight it changes.
DWX_DEV double saturating_step(double c, double stepsize) {
  return c > 0.0 ? -expm1(-c * stepsize) / c : stepsize;
}
__global__ void __launch_bounds__(BLOCK_THREADS)
apply_kernel(double *weights, float *w32, const uint8_t *w_fixed, long long *grad,
             const long long *t_static, const long long *t_hess, uint32_t W, double stepsize,
             double reg_param, int l2) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < W; i += stride) {
    const long long G = grad[i], Td = grad[W + i];
    if (G != 0 || Td != 0) { grad[i] = 0; grad[W + i] = 0; }
    const long long Tn = Td + (t_static ? t_static[i] : 0);
    if (w_fixed[i] || Tn == 0) continue;
    const double Tt = (double)Tn / FIX_SCALE, Gg = (double)G / FIX_SCALE;
    const double h = t_hess ? (double)t_hess[i] / H_SCALE : 0.0;
    double x = weights[i];
    if (l2) {
      x -= saturating_step(h + reg_param * Tt, stepsize) * (Gg + reg_param * Tt * x);
    } else {
      x += reg_param * Tt * (x < 0 ? 1.0 : 0.0);
      x -= saturating_step(h, stepsize) * Gg;
    }
    weights[i] = x;
    w32[i] = (float)x;
  }
}

// Inference with unchanged weights repeats the same products sweep after sweep: tabulate
// them once.  For every pre-signed (unary) record, exactly the two terms the staging pass of
// sweep_kernel computes -- f64 products of two f32 values, exact -- in a stream with the
// records' own 16-byte stride; other records get zeros (their tiles never read the table).
// Inference sweeps then stream the table and touch no weight: the 50 M random L2 requests
// per sweep that bound config 3's inference are gone.
__global__ void __launch_bounds__(BLOCK_THREADS)
build_terms_kernel(const TileDesc *tiles, uint32_t n_tiles, const EdgeRec *edges, const float *w32,
                   EdgeTerms *terms) {
  for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const TileDesc d = tiles[tile];
    const bool simple = d.flags & TILE_SIMPLE, inline2 = d.flags & TILE_INLINE2;
    if (!simple && !inline2) continue;        // such tiles never read the table
    for (uint32_t i = threadIdx.x; i < d.nedges; i += BLOCK_THREADS) {
      const EdgeRec r = edges[d.e0 + i];
      const double wv = (double)w32[r.wid];
      if (simple) {
        EdgeTerms tt;
        tt.t1 = wv * (double)r.fval;
        tt.t0 = wv * (double)bits_to_float(r.aux);
        terms[d.e0 + i] = tt;
      } else {
        TabRec2 tr;
        if (r.packed & EDGE_PRESIGNED) {
          // hit / miss values are s * f with s in {-1, 0, +1}: one product, two small codes
          const float hit = r.fval, miss = bits_to_float(r.aux);
          const float ref = hit != 0.0f ? hit : miss;
          tr.wf = wv * (double)ref;
          const uint32_t c0 = miss == 0.0f ? 1u : (miss == ref ? 2u : 0u);
          tr.bits = (r.packed & EDGE_FUNC_MASK) | TAB2_UNARY | (hit != 0.0f ? TAB2_C1 : 0u) | (c0 << TAB2_C0_SHIFT);
          tr.other = d.v0 + edge_owner_lane(r);
        } else {
          tr.wf = wv * (double)r.fval;
          tr.bits = r.packed & (EDGE_FUNC_MASK | (EDGE_ARITY_MASK << EDGE_ARITY_SHIFT));
          tr.other = r.aux;
        }
        ((TabRec2 *)terms)[d.e0 + i] = tr;
      }
    }
  }
}

// The terms table of an all-unary graph (compact records): 8 bytes per record, the exact
// product w * f (f64 of two f32) with sign(hit) + 1 and sign(miss) + 1 in its four lowest
// mantissa bits -- zero in every such product, so nothing is lost.
__global__ void __launch_bounds__(BLOCK_THREADS)
build_terms8_kernel(const EdgeRec8 *edges8, uint64_t n, const float *w32, unsigned long long *terms) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const EdgeRec8 c = edges8[i];
    const double wf = (double)w32[c.key & REC8_WID_MASK] * (double)c.f;
    unsigned long long u;
    __builtin_memcpy(&u, &wf, 8);
    terms[i] = u | ((c.key >> REC8_HIT_SHIFT) & 15u);   // hit code in bits 0-1, miss code in bits 2-3
  }
}

// f64 master weights -> f32 sampling copy (after dwx_set_weights)
__global__ void __launch_bounds__(BLOCK_THREADS)
refresh_w32_kernel(const double *weights, float *w32, uint32_t W) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < W; i += stride)
    w32[i] = (float)weights[i];
}

// Replica averaging (InferenceResult::average_weights + copy_weights_to,
// src/inference_result.cc:75-86): the caller summed the replicas' weights in place; divide
// by their number.  Fixed weights are put back verbatim (copy_weights_to skips them; a sum
// of n equal values divided by n need not round back for n = 3, 5, 6, 7).
__global__ void __launch_bounds__(BLOCK_THREADS)
average_weights_kernel(double *weights, float *w32, const uint8_t *w_fixed, const double *w_init,
                       uint32_t W, double n_replicas) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < W; i += stride) {
    const double w = w_fixed[i] ? w_init[i] : weights[i] / n_replicas;
    weights[i] = w;
    w32[i] = (float)w;
  }
}

// Halo exchange (multi-GPU, cross-shard factors): gather the listed variables' assignments of
// the selected chains into a contiguous buffer [chain][i] (what a peer receives), and the
// reverse for the ghosts.  pos = device positions; chains = bit 0 free, bit 1 evidence chain;
// the buffer holds the selected chains back to back.
// BITS per value: 32, 8 (every listed cardinality <= 256) or 1 (every listed variable boolean:
// a wave's 64 values are one ballot); block = 8-byte words per chain block.
template <int BITS>
__global__ void __launch_bounds__(BLOCK_THREADS)
halo_pack_kernel(const uint32_t *pos, uint32_t n, const uint32_t *assign_free, const uint32_t *assign_evid,
                 uint32_t chains, unsigned long long *buf, uint32_t block) {
  const uint32_t stride = gridDim.x * blockDim.x;
  const uint32_t n_round = (n + blockDim.x - 1) / blockDim.x * blockDim.x;   // (whole workgroups ballot together)
  unsigned long long *second = buf + ((chains & 1u) ? block : 0u);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
    const uint32_t p = pos[i < n ? i : n - 1];
    const uint32_t vf = (chains & 1u) ? assign_free[p] : 0u, ve = (chains & 2u) ? assign_evid[p] : 0u;
    if (BITS == 1) {
      const unsigned long long mf = DWX_BALLOT(i < n && (vf & 1u)), me = DWX_BALLOT(i < n && (ve & 1u));
      if ((threadIdx.x & 63u) == 0 && i < n) {
        if (chains & 1u) buf[i >> 6] = mf;
        if (chains & 2u) second[i >> 6] = me;
      }
    } else if (i < n) {
      if (BITS == 8) {
        if (chains & 1u) ((unsigned char *)buf)[i] = (unsigned char)vf;
        if (chains & 2u) ((unsigned char *)second)[i] = (unsigned char)ve;
      } else {
        if (chains & 1u) ((uint32_t *)buf)[i] = vf;
        if (chains & 2u) ((uint32_t *)second)[i] = ve;
      }
    }
  }
}
template <int BITS>
__global__ void __launch_bounds__(BLOCK_THREADS)
halo_unpack_kernel(const uint32_t *pos, uint32_t n, uint32_t *assign_free, uint32_t *assign_evid,
                   uint32_t chains, const unsigned long long *buf, uint32_t block) {
  const uint32_t stride = gridDim.x * blockDim.x;
  const unsigned long long *second = buf + ((chains & 1u) ? block : 0u);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint32_t p = pos[i];
    uint32_t vf = 0, ve = 0;
    if (BITS == 1) {
      if (chains & 1u) vf = (uint32_t)(buf[i >> 6] >> (i & 63u)) & 1u;
      if (chains & 2u) ve = (uint32_t)(second[i >> 6] >> (i & 63u)) & 1u;
    } else if (BITS == 8) {
      if (chains & 1u) vf = ((const unsigned char *)buf)[i];
      if (chains & 2u) ve = ((const unsigned char *)second)[i];
    } else {
      if (chains & 1u) vf = ((const uint32_t *)buf)[i];
      if (chains & 2u) ve = ((const uint32_t *)second)[i];
    }
    if (chains & 1u) assign_free[p] = vf;
    if (chains & 2u) assign_evid[p] = ve;
  }
}

// test hook: the raw Philox4x32-10 block function and the two uniforms drawn from it, on the
// device (Random123 known-answer vectors; tests/test_philox_kat.py)
__global__ void test_philox_kernel(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                   uint32_t *out, double *uni) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    philox4x32_10(k0, k1, c0, c1, c2, c3);
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
    // the same block through the sampler's own entry point: seed = key, counter = (vid, sweep)
    philox_uniforms((uint64_t)k0 | ((uint64_t)k1 << 32), (uint64_t)out[4] | ((uint64_t)out[5] << 32),
                    (uint64_t)out[6] | ((uint64_t)out[7] << 32), uni[0], uni[1]);
  }
}

// test hook: one factor function evaluated on the device (test/factor_test.cc)
__global__ void test_sign_kernel(uint32_t func, uint32_t arity, const VifRec *vifs,
                                 const uint32_t *assign, double *out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    if (arity == 1) out[0] = factor_sign(func, 1, 1u, vifs, assign, 0u, assign[0]);
    else out[0] = factor_sign(func, arity, 0u, vifs, assign, kNoVar, 0u);
  }
}

}  // namespace dwx
#endif  // DWX_SWEEP_KERNELS_H_
