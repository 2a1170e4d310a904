// factor_functions.h -- device-side arithmetic shared by every kernel: the Philox4x32-10 stream,
// logadd / the draws' math, and the ten factor functions (src/factor.h:59-299) in their
// one-evaluation and N-scenarios-in-one-walk forms.  Part of sweep_kernels.h (HIP source that
// tests/hipemu also compiles for the host).
#ifndef DWX_FACTOR_FUNCTIONS_H_
#define DWX_FACTOR_FUNCTIONS_H_

#include "device_intrinsics.h"



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

// Potentials of the boolean variables of an all-unary graph (compact records) are summed in
// FIXED POINT, 2^-32: pp - pn = sum over the row of fix(w * d), d = (sign(hit) - sign(miss)) * f
// (an exact f64 product of two f32-exact factors).  Integer sums do not depend on the order of
// the records, which is what lets the weight-sorted sweep (sorted_sweep_kernel) add a variable's
// terms in the order of their WEIGHTS; every kernel that can meet such a variable uses the same
// sum, and the oracle's schedule mode restates it (fixed-point mask), so parity stays exact.
// (The reference's own sums are -Ofast doubles in row order: neither is "the" value.)
// The conversion is the round-to-nearest of one f64 addition: exact for |p| < 2^19 (clamped;
// a single term of half a million is far past where exp() saturates).
constexpr double POT_FIX_SCALE = 4294967296.0;   // 2^32
constexpr double POT_FIX_CLAMP = 524288.0;       // 2^19
DWX_DEV long long pot_fix(double p) {
  p = p < -POT_FIX_CLAMP ? -POT_FIX_CLAMP : (p > POT_FIX_CLAMP ? POT_FIX_CLAMP : p);
  const double magic = 6755399441055744.0 / POT_FIX_SCALE;   // 1.5 * 2^52 * 2^-32
  const double sh = p + magic;
  long long a, b;
  __builtin_memcpy(&a, &sh, 8);
  __builtin_memcpy(&b, &magic, 8);
  return a - b;
}
DWX_DEV double pot_unfix(long long q) { return (double)q * (1.0 / POT_FIX_SCALE); }

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
// Straight-line (round 4): a per-lane `switch (func)` compiles to a cascade of exec-mask branches -- in the
// edge-parallel staging of sweep_kernel<LEARN> (6 records x 4 evaluations per lane) that cascade was 4 253 of
// the kernel's 6 915 instructions, all scalar.  A function of two bits IS its four-entry truth table: one byte
// per function id, two bits per (a, b) holding sign + 1:
//   AND, ISTRUE          (a && b) ? +1 : -1                     OR       (a || b) ? +1 : -1
//   AND_CATEGORICAL      (a && b) ? +1 :  0                     EQUAL    (a == b) ? +1 : -1
//   IMPLY_NATURAL        !a ? 0 : (b ? +1 : -1)                 IMPLY_MLN !a ? +1 : (b ? +1 : 0)
//   LINEAR, LOGICAL      (!a || b) ? +1 : 0                     RATIO    log2(1 + [!a || b]) = the same
// (src/factor.h:112-299 at arity 2; ids that are no function read as the last line, as the switch's default did)
constexpr uint32_t binary_truth_byte(int sign_ff, int sign_ft, int sign_tf, int sign_tt) {
  return (uint32_t)(sign_ff + 1) | (uint32_t)(sign_ft + 1) << 2 | (uint32_t)(sign_tf + 1) << 4 | (uint32_t)(sign_tt + 1) << 6;
}
constexpr uint64_t binary_truth_word(int first_func) {
  uint64_t word = 0;
  for (int i = 0; i < 8; ++i) {
    const uint32_t f = (uint32_t)(first_func + i);
    uint32_t byte = binary_truth_byte(1, 1, 0, 1);                       // LINEAR / LOGICAL / RATIO / IMPLY_MLN
    if (f == FUNC_AND || f == FUNC_ISTRUE) byte = binary_truth_byte(-1, -1, -1, 1);
    else if (f == FUNC_AND_CATEGORICAL) byte = binary_truth_byte(0, 0, 0, 1);
    else if (f == FUNC_OR) byte = binary_truth_byte(-1, 1, 1, 1);
    else if (f == FUNC_EQUAL) byte = binary_truth_byte(1, -1, -1, 1);
    else if (f == FUNC_IMPLY_NATURAL) byte = binary_truth_byte(0, 0, -1, 1);
    word |= (uint64_t)byte << (8 * i);
  }
  return word;
}
// the function's table (func < 16: EDGE_FUNC_MASK)
DWX_DEV uint32_t binary_truth(uint32_t func) {
  constexpr uint64_t LO = binary_truth_word(0), HI = binary_truth_word(8);
  const uint64_t word = (func & 8u) ? HI : LO;
  return (uint32_t)(word >> ((func & 7u) * 8u)) & 0xFFu;
}
// ... and its entry: sign + 1 in two bits / the sign -1, 0 or +1
DWX_DEV uint32_t binary_code2(uint32_t truth, bool a, bool b) {
  return (truth >> (((a ? 2u : 0u) | (b ? 1u : 0u)) * 2u)) & 3u;
}
DWX_DEV int binary_code(uint32_t truth, bool a, bool b) { return (int)binary_code2(truth, a, b) - 1; }
DWX_DEV double binary_sign(uint32_t func, bool a, bool b) { return (double)binary_code(binary_truth(func), a, b); }

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

}  // namespace dwx
#endif  // DWX_FACTOR_FUNCTIONS_H_
