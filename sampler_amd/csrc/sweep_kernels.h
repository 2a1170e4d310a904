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
//
// Layout of the device code: factor_functions.h (RNG, math, factor functions) <- tile_walk.h (row
// walks, draws, SGD of one variable) <- this file (staging + the sweep kernels of the three degree
// bins) ; aux_kernels.h (pull gradient, apply, tables, halo).
#ifndef DWX_SWEEP_KERNELS_H_
#define DWX_SWEEP_KERNELS_H_

#include <type_traits>

#include "tile_walk.h"

namespace dwx {

// ---------------------------------------------------------------- kernels
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
  if (d.flags & TILE_UNIT_ROWS) {   // workgroup-uniform: one record per row, nothing to load
#pragma unroll
    for (uint32_t k = 0; k < (uint32_t)RP; ++k) f.rp[k] = d.e0 + umin(t + k * BLOCK_THREADS, d.nrows);
  } else {
#pragma unroll
    for (uint32_t k = 0; k < (uint32_t)RP; ++k) {
      const uint32_t *rp = &P.row_ptr[d.r0 + umin(t + k * BLOCK_THREADS, d.nrows)];
      f.rp[k] = NT ? DWX_NT_LOAD(rp) : *rp;
    }
  }
  f.pre = load_var_pre<LEARN, NT>(P, d.v0 + umin(pair ? t >> 1 : t, d.nv - 1));
}

// Edge-parallel evaluation of a tile's staged records whose factors have arity <= GEN_ARITY
// (TILE_TERMS3): the lane's K records in three batched phases -- all their first three
// factor->variable entries, then those variables' assignments on every chain, then the sign
// functions on registers -- in NS scenarios at once (as coop_for_records).  out(k, term[NS]).
// SBATCH records per batch: their loads are all in flight together (registers against
// memory-level parallelism; 0: all K).  Measured (config 3c / categorical 4b, K = 6): inference
// 0.934 / 0.391 ms in one batch, 0.916 / 0.349 in two (155 VGPRs and no spills instead of 167 + 16
// spilled: config 3b's inference gains 5 % too); learning 2.92 / 1.93 against 3.04 / 1.74, and both
// forms in one kernel cost more than either -- inference takes two batches, learning one.
template <int K, int NS, int NCHAIN, int SBATCH, class Out>
DWX_DEV void stage_generic_records(const KernelParams &P, const TileDesc &d, const EdgeRec (&rec)[K],
                                   const uint32_t *const (&chains)[NCHAIN], const int (&chain)[NS],
                                   const uint32_t (&prop)[NS], const bool (&hit)[NS], Out &&out) {
  constexpr int SB = (SBATCH > 0 && SBATCH < K) ? SBATCH : K;
#pragma unroll
  for (int k0 = 0; k0 < K; k0 += SB) {
    VifRec vf[SB][GEN_ARITY];
    uint32_t val[SB][NCHAIN][GEN_ARITY];
#pragma unroll
    for (int b = 0; b < SB; ++b) {
      const int k = k0 + b < K ? k0 + b : K - 1;
      const bool generic = !(rec[k].packed & EDGE_PRESIGNED);
      const uint32_t ar = generic ? edge_arity(rec[k]) : 1u, base = (generic && ar >= 2u) ? rec[k].aux : 0u;
#pragma unroll
      for (uint32_t i = 0; i < GEN_ARITY; ++i) vf[b][i] = P.vifs[base + umin(i, ar - 1u)];
    }
#pragma unroll
    for (int b = 0; b < SB; ++b)
#pragma unroll
      for (int c = 0; c < NCHAIN; ++c)
#pragma unroll
        for (uint32_t i = 0; i < GEN_ARITY; ++i) val[b][c][i] = chains[c][vf[b][i].vid];
#pragma unroll
    for (int b = 0; b < SB; ++b) {
      const int k = k0 + b;
      if (k >= K) break;
      const EdgeRec r = rec[k];
      double term[NS];
      if (r.packed & EDGE_PRESIGNED) {
#pragma unroll
        for (int j = 0; j < NS; ++j) term[j] = (double)(hit[j] ? r.fval : bits_to_float(r.aux));
      } else {
        const uint32_t me = d.v0 + edge_owner_lane(r);
        double sg[NS];
        const VifsPreloaded<NS, NCHAIN> src{vf[b], val[b], chain};
        factor_signs_from<NS>(edge_func(r), edge_arity(r), src, me, prop, sg);
#pragma unroll
        for (int j = 0; j < NS; ++j) term[j] = sg[j] * (double)r.fval;
      }
      out(k, term);
    }
  }
}

// The per-workgroup gradient accumulators of a learning launch on a graph with few weights (LDS:
// int64[2W], W <= LDS_AGG_MAX_W), flushed once per persistent workgroup with atomics that skip
// zero sums.  (Round 4 measured what that flush costs a chunk of config 4's split sweep -- 407
// workgroups x 16 sums onto 16 addresses: 1.2 us of 18, compiled out; a row of sums per workgroup
// + a summing update kernel instead was 3 us per chunk SLOWER in wall time: not kept.)
DWX_DEV void flush_accumulators(const KernelParams &P, const long long *s_agg, uint32_t t) {
  if (!s_agg) return;
  __syncthreads();
#ifndef DWX_EXP_NOFLUSH   // (timing experiment)
  for (uint32_t i = t; i < 2 * P.num_weights; i += BLOCK_THREADS) {
    const long long v = s_agg[i];
    if (v) atomicAdd((unsigned long long *)&P.grad[i], (unsigned long long)v);
  }
#endif
}

// Persistent, software-pipelined sweep: workgroup b handles tiles b, b + gridDim.x, ...
// of the launch.  While a tile is processed out of LDS, the NEXT tile's edge records,
// row pointers and per-variable inputs are already in flight into registers, so the
// HBM latency of a tile hides behind the previous tile's arithmetic; the Philox draw
// of a tile is computed under the latency of its weight gathers.  K = records staged
// per lane (LDS holds K * 256 records).  Oversized variables are skipped here and
// handled by giant_kernel.
// WIDE (learning only): the graph has TILE_TERMS2 / TILE_TERMS3 tiles; their records are staged as
// 32-byte LearnRec (the LDS region doubles: 48 KiB per workgroup at K = 6, three workgroups per CU
// -- measured against two and four: 2.33 / 2.75 / 4.8 ms on config 3b).
// TV: the tile variants this instantiation contains (round 4).  One body with workgroup-uniform
// branches over EVERY variant was 384 KB of machine code at <LEARN, 6, WIDE> -- the generic
// factor walk, the arity <= 3 staging through the general sign functions, the vif-pair loads -- and
// the register allocation of the one path a graph runs paid for all of them (34 VGPRs and 268 SGPRs
// spilled on config 3b's path, which needs none of those).  The host picks the smallest build that
// holds the graph's tile classes (dwx_api.cc: tile_variants): TV_PAIR for all-boolean graphs whose tiles
// are all pre-signed unary and/or inline arity-2 records (config 3b / 5b), TV_ALL otherwise.
constexpr uint32_t TV_SIMPLE = 1u, TV_TERMS2_INLINE = 2u, TV_TERMS2_VIFS = 4u, TV_TERMS3 = 8u, TV_GENERIC = 16u;
constexpr uint32_t TV_CATEGORICAL = 32u;      // some lane tile holds categorical variables
constexpr uint32_t TV_ALL = 63u, TV_PAIR = TV_SIMPLE | TV_TERMS2_INLINE;
#ifndef DWX_PAIR_WG
#define DWX_PAIR_WG 4
#endif
// the staged learning record of a TILE_TERMS2 tile, in either form (tile_walk.h); codes: sign + 1
DWX_DEV void store_learn_rec(LearnRec *dst, const EdgeRec &r, float w, bool presigned, float miss,
                             uint32_t cu1, uint32_t cu0, uint32_t ce1, uint32_t ce0) {
  // sign in {-1, 0, +1} times an f32: that f32, its negation or a zero -- in f32 what
  // (float)(sign * (double)f) is, bit for bit
  LearnRec lr;
  lr.wid = r.wid; lr.packed = r.packed; lr.w = w; lr.pad = 0;
  lr.sf1 = presigned ? r.fval : (float)((int)cu1 - 1) * r.fval; lr.sf0 = presigned ? miss : (float)((int)cu0 - 1) * r.fval;
  lr.se1 = presigned ? r.fval : (float)((int)ce1 - 1) * r.fval; lr.se0 = presigned ? miss : (float)((int)ce0 - 1) * r.fval;
  *dst = lr;
}
DWX_DEV void store_learn_rec(LearnRec16 *dst, const EdgeRec &r, float w, bool presigned, float miss,
                             uint32_t cu1, uint32_t cu0, uint32_t ce1, uint32_t ce0) {
  LearnRec16 lr;
  lr.wf = r.wid | (presigned ? LR16_PRESIGNED : 0u) | ((r.packed & EDGE_FIXED_FLAG) ? LR16_FIXED : 0u);
  lr.w = w; lr.a = r.fval;
  lr.b = presigned ? miss : bits_to_float(cu1 | cu0 << 2 | ce1 << 4 | ce0 << 6);
  *dst = lr;
}
template <bool LEARN, int K, bool WIDE = false, uint32_t TV = TV_ALL>
// (the learning kernel's LDS footprint admits 2 workgroups per CU at K = 12: give the
// register allocator the matching budget instead of spilling at the 3-per-CU limit)
// (the TV_PAIR learning build stages 16-byte records: its LDS footprint admits DWX_PAIR_WG workgroups per CU)
__global__ void __launch_bounds__(BLOCK_THREADS, WIDE ? (K <= 6 ? ((TV & TV_TERMS3) ? 3 : DWX_PAIR_WG) : 1) : (LEARN ? (K <= 6 ? 3 : 2) : 3)) sweep_kernel(const KernelParams P) {
  DWX_DYN_LDS(dyn_lds);
  uint32_t *s_rowptr = (uint32_t *)dyn_lds;
  double *s_pot = (double *)(dyn_lds + P.lds_pot_off);
  EdgeRec *s_edges = (EdgeRec *)(dyn_lds + P.lds_edge_off);
  float *s_w = (float *)(dyn_lds + P.lds_w_off);
  long long *s_agg = (LEARN && P.lds_agg_off) ? (long long *)(dyn_lds + P.lds_agg_off) : nullptr;
  const uint32_t t = threadIdx.x;
  constexpr int WMODE = LEARN ? W_ARRAY : W_INRECORD;
  // staged learning records of TILE_TERMS2 tiles: 16 bytes in a build without arity-3 staging (tile_walk.h)
  using LRec = typename std::conditional<(TV & TV_TERMS3) != 0, LearnRec, LearnRec16>::type;
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
    // ... and a TERMS tile whose unary records are pulled the same way (the others scatter)
    const bool pull_unary = K <= 6 && LEARN && WIDE && fits && (d.flags & TILE_PULL_UNARY) && !(P.flags & OPT_NO_PULL);
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
      if ((TV & TV_TERMS3) && K <= 6 && LEARN && WIDE && (d.flags & TILE_TERMS3)) {
        // arity <= 3: the four sign * feature value products of every record (free / evidence
        // chain x proposal 1 / 0) through the general sign functions on batched loads
        LearnRec *s_lrec = (LearnRec *)s_edges;
        const uint32_t *const chains[2] = {P.assign_free, P.assign_evid};
        const int chain[4] = {0, 0, 1, 1};
        const bool cat = d.flags & TILE_CATEGORICAL;   // (a record's proposal is its row's value)
        const uint32_t p1 = cat ? PROP_OWN : 1u, p0 = cat ? PROP_OTHER : 0u;
        const uint32_t prop[4] = {p1, p0, p1, p0};
        const bool hit[4] = {true, false, true, false};
        auto put = [&](int k, const double (&term)[4]) {
          LearnRec lr;
          lr.wid = rec[k].wid; lr.packed = rec[k].packed; lr.w = w[k]; lr.pad = 0;
          lr.sf1 = (float)term[0]; lr.sf0 = (float)term[1]; lr.se1 = (float)term[2]; lr.se0 = (float)term[3];
          s_lrec[t + k * BLOCK_THREADS] = lr;
        };
        stage_generic_records<K, 4, 2, 0>(P, d, rec, chains, chain, prop, hit, put);
      } else if ((TV & (TV_TERMS2_INLINE | TV_TERMS2_VIFS)) && K <= 6 && LEARN && WIDE && (d.flags & TILE_TERMS2)) {
        LRec *s_lrec = (LRec *)s_edges;
        VifRec va[K], vb[K];
        if (!(TV & TV_TERMS2_VIFS) || (d.flags & TILE_INLINE2)) {   // workgroup-uniform
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
          // straight-line (no per-lane branch: the function is its truth table, factor_functions.h; a
          // pre-signed record decodes to harmless operands and takes its own two fields at the end)
          const EdgeRec r = rec[k];
          const bool presigned = r.packed & EDGE_PRESIGNED;
          const uint32_t me = d.v0 + edge_owner_lane(r);
          const bool a_me = va[k].vid == me, b_me = vb[k].vid == me;
          const bool a1 = va[k].equal_to == 1u, a0 = va[k].equal_to == 0u;
          const bool b1 = vb[k].equal_to == 1u, b0 = vb[k].equal_to == 0u;
          const bool af = of[k] == va[k].equal_to, bf = of[k] == vb[k].equal_to;
          const bool ae = oe[k] == va[k].equal_to, be = oe[k] == vb[k].equal_to;
          const uint32_t truth = binary_truth(edge_func(r));
          // sign + 1 of the four evaluations: free / evidence chain x proposal 1 / 0
          const uint32_t cu1 = binary_code2(truth, a_me ? a1 : af, b_me ? b1 : bf);
          const uint32_t cu0 = binary_code2(truth, a_me ? a0 : af, b_me ? b0 : bf);
          const uint32_t ce1 = binary_code2(truth, a_me ? a1 : ae, b_me ? b1 : be);
          const uint32_t ce0 = binary_code2(truth, a_me ? a0 : ae, b_me ? b0 : be);
          const float miss = bits_to_float(r.aux);
          store_learn_rec(&s_lrec[t + k * BLOCK_THREADS], r, w[k], presigned, miss, cu1, cu0, ce1, ce0);
        }
      } else if ((TV & TV_TERMS3) && K <= 6 && !LEARN && (d.flags & TILE_TERMS3)) {
        // inference, arity <= 3: both proposals' terms of every record, edge-parallel
        EdgeTerms *s_terms = (EdgeTerms *)s_edges;
        const uint32_t *const chains[1] = {P.assign_evid};
        const int chain[2] = {0, 0};
        const bool cat = d.flags & TILE_CATEGORICAL;   // (a record's proposal is its row's value; t0 unused)
        const uint32_t prop[2] = {cat ? PROP_OWN : 1u, cat ? PROP_OTHER : 0u};
        const bool hit[2] = {true, false};
        stage_generic_records<K, 2, 1, (K + 1) / 2>(P, d, rec, chains, chain, prop, hit, [&](int k, const double (&term)[2]) {
          const double wv = (double)w[k];
          EdgeTerms tt;
          tt.t1 = wv * term[0];
          tt.t0 = wv * term[1];
          s_terms[t + k * BLOCK_THREADS] = tt;
        });
      } else if ((TV & (TV_TERMS2_INLINE | TV_TERMS2_VIFS)) && K <= 6 && !LEARN && (d.flags & TILE_TERMS2)) {
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
            // (straight-line; a unary entry's garbage operands are never selected)
            const bool unary = bits & TAB2_UNARY;
            const uint32_t c0 = (bits >> TAB2_C0_SHIFT) & 3u;   // 0: -1, 1: 0, 2: +1
            const uint32_t pa = (bits >> INLINE2_PRED_A_SHIFT) & INLINE2_PRED_MASK;
            const uint32_t pb = (bits >> INLINE2_PRED_B_SHIFT) & INLINE2_PRED_MASK;
            const bool a_me = bits & INLINE2_A_IS_OWNER, b_me = bits & INLINE2_B_IS_OWNER;
            const bool a_o = other[k] == pa, b_o = other[k] == pb;
            const bool a1 = a_me ? (pa == 1u) : a_o, b1 = b_me ? (pb == 1u) : b_o;
            const bool a0 = a_me ? (pa == 0u) : a_o, b0 = b_me ? (pb == 0u) : b_o;
            const uint32_t truth = binary_truth(bits & EDGE_FUNC_MASK);
            const double s1 = (double)binary_code(truth, a1, b1) * wf, s0 = (double)binary_code(truth, a0, b0) * wf;
            tt.t1 = unary ? ((bits & TAB2_C1) ? wf : 0.0) : s1;
            tt.t0 = unary ? (c0 == 1u ? 0.0 : (c0 == 2u ? wf : -wf)) : s0;
            s_terms[t + k * BLOCK_THREADS] = tt;
          }
        } else {
        VifRec va[K], vb[K];
        if (!(TV & TV_TERMS2_VIFS) || (d.flags & TILE_INLINE2)) {   // workgroup-uniform
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
          // (straight-line, as in the learning staging above: sign * f in f32 is exact)
          const bool presigned = r.packed & EDGE_PRESIGNED;
          const uint32_t me = d.v0 + edge_owner_lane(r);
          const bool a_me = va[k].vid == me, b_me = vb[k].vid == me;
          const bool a_o = other[k] == va[k].equal_to, b_o = other[k] == vb[k].equal_to;
          // satisfied bits under proposal x: own positions compare x with their predicate
          const bool a1 = a_me ? (va[k].equal_to == 1u) : a_o, b1 = b_me ? (vb[k].equal_to == 1u) : b_o;
          const bool a0 = a_me ? (va[k].equal_to == 0u) : a_o, b0 = b_me ? (vb[k].equal_to == 0u) : b_o;
          const uint32_t truth = binary_truth(edge_func(r));
          const float u1 = (float)binary_code(truth, a1, b1) * r.fval, u0 = (float)binary_code(truth, a0, b0) * r.fval;
          tt.t1 = wv * (double)(presigned ? r.fval : u1);
          tt.t0 = wv * (double)(presigned ? bits_to_float(r.aux) : u0);
          s_terms[t + k * BLOCK_THREADS] = tt;
        }
        }
      } else if ((TV & TV_SIMPLE) && (LEARN ? pull : (bool)(d.flags & TILE_SIMPLE))) {
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
        s_rowptr[i] = (d.flags & TILE_UNIT_ROWS) ? d.e0 + i : P.row_ptr[d.r0 + i];
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
    if ((TV & (TV_TERMS2_INLINE | TV_TERMS2_VIFS | TV_TERMS3)) && fits && chain_pair_tile<LEARN, K, WIDE>(d)) {
      if (t < 2u * d.nv)
        delta = learn_variable_terms2_pair(P, s_rowptr, d.r0, (const LRec *)s_edges, d.e0, s_agg, d.v0 + (t >> 1), pre, A, B,
                                           t & 1u, pull_unary);
    } else if (fits && t < d.nv) {
      TileView T{s_rowptr, d.r0, s_edges, d.e0, s_w, s_agg, P.lds_pot_off ? s_pot : nullptr};
      constexpr uint32_t TV_T23 = TV_TERMS2_INLINE | TV_TERMS2_VIFS | TV_TERMS3;
      constexpr bool NOCAT = !(TV & TV_CATEGORICAL);
      if ((TV & TV_TERMS3) && K <= 6 && LEARN && WIDE && (d.flags & TILE_TERMS3) && (d.flags & TILE_CATEGORICAL))
        process_variable<LEARN, W_LREC, false>(P, T, d.v0 + t, pre, A, B);
      else if ((TV & TV_T23) && K <= 6 && LEARN && WIDE && (d.flags & (TILE_TERMS2 | TILE_TERMS3)))
        delta = learn_variable_terms2(P, s_rowptr, d.r0, (const LRec *)s_edges, d.e0, s_agg, d.v0 + t, pre, A, B, pull_unary);
      else if ((TV & TV_SIMPLE) && LEARN && pull)   // the staged records ARE terms: sgd_row is never reached (want_delta)
        delta = process_variable<LEARN, W_TERMS, true, false, NOCAT>(P, T, d.v0 + t, pre, A, B, true);
      else if (((TV & TV_SIMPLE) && (d.flags & TILE_SIMPLE)) || ((TV & TV_T23) && K <= 6 && !LEARN && (d.flags & (TILE_TERMS2 | TILE_TERMS3))))
        delta = process_variable<LEARN, LEARN ? W_ARRAY : W_TERMS, true, false, NOCAT>(P, T, d.v0 + t, pre, A, B, false);
      else if (TV & TV_GENERIC)
        process_variable<LEARN, WMODE, false>(P, T, d.v0 + t, pre, A, B);
    }
    if (pull || pull_unary) {
      // every wave of the tile publishes its two ballots (also when all zero: the words
      // are rewritten each learning sweep, so nothing needs clearing)
      const unsigned long long nz = DWX_BALLOT(delta != 0), ng = DWX_BALLOT(delta < 0);
      if (pull_unary && chain_pair_tile<LEARN, K, WIDE>(d)) {
        // two lanes per variable: a wave holds 32 variables, half a ballot word
        const uint32_t nz32 = compress_even_bits(nz), ng32 = compress_even_bits(ng);
        if ((t & 63u) == 0) {
          uint32_t *w = (uint32_t *)(P.delta + ((size_t)tile * 4 + (t >> 7)) * 2);
          const uint32_t half = (t >> 6) & 1u;
          DWX_NT_STORE(nz32, &w[half]); DWX_NT_STORE(ng32, &w[2 + half]);
        }
      } else if ((t & 63u) == 0) {
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
  flush_accumulators(P, s_agg, t);
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
// Learning: the register budget of three workgroups per CU (150 VGPRs, nothing spilled).  With the
// budget of four (128) the K = 12 build spilled 15 VGPRs to scratch for 1.6 % on config 3 -- a
// sweep the weight-sorted super-tiles now take; this kernel serves what is left (fewer than
// 4096 weights, the tiles at the edges of a mini-batch, categorical rows).
#ifndef DWX_S8_INFER_WG
#define DWX_S8_INFER_WG 3
#endif
#ifndef DWX_S8_LEARN_WG
#define DWX_S8_LEARN_WG 3
#endif
// MULTI (inference, gathering build only): P.n_sweeps sweeps per launch -- the tile is staged once
// and every variable is drawn n_sweeps times (infer_variable_multi, tile_walk.h).
// LW (sweep8_merged_kernel, persist_kernels.h): the weight gathers read lw32, the workgroup's own LDS copy
// of the f32 weights, instead of P.w32.
// ONE (sweep8_merged_kernel when the launch has a workgroup per tile -- a mini-batch of a split sweep): no next
// tile, so no descriptor ahead, no prefetch of a tile that does not exist, no loop.
// Pre: work of the caller's that needs no tile data, run once the first tile's loads are in flight (the merged
// kernel's weight update: its own loads then return under the record stream's instead of before it).
struct NoPrologue { DWX_DEV void operator()() const {} };
template <bool LEARN, int K, bool TAB = false, int RP = (int)ROWPTR_UNROLL, bool MULTI = false, bool LW = false,
          bool ONE = false, class Pre = NoPrologue>
DWX_DEV void sweep8_body(const KernelParams &P, const float *lw32, Pre pre_fn = Pre()) {
  static_assert(!(LEARN && TAB), "the terms table serves inference sweeps only");
  static_assert(!MULTI || (!LEARN && !TAB), "several sweeps per launch: the gathering inference build");
  static_assert(!LW || (LEARN && !TAB && !MULTI), "LDS weights: the learning build");
  DWX_DYN_LDS(dyn_lds);
  uint32_t *s_rowptr = (uint32_t *)dyn_lds;
  double *s_pot = (double *)(dyn_lds + P.lds_pot_off);
  EdgeRec *s_edges = (EdgeRec *)(dyn_lds + P.lds_edge_off);
  float *s_w = (float *)(dyn_lds + P.lds_w_off);
  long long *s_agg = (LEARN && P.lds_agg_off) ? (long long *)(dyn_lds + P.lds_agg_off) : nullptr;
  const uint32_t t = threadIdx.x;
  uint32_t tile = P.tile_begin + blockIdx.x;
  if (tile >= P.tile_end) { pre_fn(); return; }   // (workgroup-uniform)
  const uint32_t stride = gridDim.x;
  TileDesc d = scalarise(P.tiles[tile]);
  uint32_t next = tile + stride;
  bool has_next = !ONE && next < P.tile_end;
  TileDesc dn = ONE ? d : scalarise(P.tiles[has_next ? next : tile]);   // one descriptor ahead
  TilePrefetch<K, EdgeRec8, RP> f;
  issue_tile_loads<LEARN, K, !TAB>(P, d, t, f);
  pre_fn();
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
        for (int k = 0; k < K; ++k) w[k] = (LW ? lw32 : P.w32)[f.rec[k].key & REC8_WID_MASK];
      }
      // ... and this lane's uniforms while the gathers are in flight
      if (!MULTI) philox_uniforms(P.seed, P.vid_offset + pre.orig, P.sweep, A, B);
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
        s_rowptr[i] = (d.flags & TILE_UNIT_ROWS) ? d.e0 + i : P.row_ptr[d.r0 + i];
      __syncthreads();
    }
    // prefetch, unconditionally and branch-free (see sweep_kernel): the descriptor two tiles
    // ahead, then the next tile's records / row pointers / variable inputs
    const uint32_t nn = next + stride;
    const bool has_nn = has_next && nn < P.tile_end;
    TileDesc raw_nn = d;
    if (!ONE) {
      raw_nn = P.tiles[has_nn ? nn : tile];
      TileDesc dl = dn;
      if (!has_next) { dl.nedges = 0; dl.nrows = 0; dl.nv = 1; }
      issue_tile_loads<LEARN, K, !TAB>(P, dl, t, f);
    }
    // the current tile out of LDS
    int delta = 0;
    if (MULTI) {
      // A tile with fewer variables than lanes (eight-value rows: 192 of 256) would leave lanes
      // idle for thousands of draws: a long run is cut into MULTI_SLICES slices per variable and
      // the (variable, slice) items are dealt out over all lanes -- the potentials are summed
      // once per item, the draws of a slice are its own sweeps' (same uniforms, same verdicts),
      // tallies add up, the slice with the last sweep stores the assignment.
      if (fits) {
        TileView T{s_rowptr, d.r0, s_edges, d.e0, s_w, s_agg, P.lds_pot_off ? s_pot : nullptr};
        const uint32_t n = P.n_sweeps;
        const uint32_t S = (n >= MULTI_SLICE_MIN_SWEEPS && d.nv < BLOCK_THREADS) ? MULTI_SLICES : 1u;   // uniform
        if (S == 1u) {   // (its own call: the loop over the sweeps keeps scalar bounds)
          if (t < d.nv) infer_variable_multi<W_TERMS, true>(P, T, d.v0 + t, pre, 0u, n, true);
        } else for (uint32_t item = t; item < S * d.nv; item += BLOCK_THREADS) {
          const uint32_t sl = item / d.nv, var = item - sl * d.nv;
          const VarPre vp = var == t ? pre : load_var_pre<false, false>(P, d.v0 + var);
          const uint32_t k_lo = (uint32_t)((uint64_t)n * sl / S), k_hi = (uint32_t)((uint64_t)n * (sl + 1) / S);
          infer_variable_multi<W_TERMS, true>(P, T, d.v0 + var, vp, k_lo, k_hi, sl + 1 == S);
        }
      }
    } else if (fits && t < d.nv) {
      TileView T{s_rowptr, d.r0, s_edges, d.e0, s_w, s_agg, P.lds_pot_off ? s_pot : nullptr};
      if (LEARN && pull)   // the staged records ARE terms: sgd_row is never reached (want_delta)
        delta = process_variable<LEARN, W_TERMS8, true, true>(P, T, d.v0 + t, pre, A, B, true);
      else   // (FIXED: boolean variables of a compact-record graph sum their potentials in fixed point)
        process_variable<LEARN, LEARN ? W_ARRAY : (TAB ? W_TERMS8 : W_TERMS), true, true>(P, T, d.v0 + t, pre, A, B, false);
    }
    if (pull) {
      const unsigned long long nz = DWX_BALLOT(delta != 0), ng = DWX_BALLOT(delta < 0);
      if ((t & 63u) == 0) {
        unsigned long long *wd = P.delta + ((size_t)tile * 4 + (t >> 6)) * 2;
        DWX_NT_STORE(nz, &wd[0]); DWX_NT_STORE(ng, &wd[1]);
      }
    }
    if (ONE || !has_next) break;
    __syncthreads();   // LDS is rewritten by the next iteration
    d = dn;
    dn = scalarise(raw_nn);
    tile = next; next = nn; has_next = has_nn;
  }
  flush_accumulators(P, s_agg, t);
}
template <bool LEARN, int K, bool TAB = false, int RP = (int)ROWPTR_UNROLL, bool MULTI = false>
__global__ void __launch_bounds__(BLOCK_THREADS, TAB ? 4 : (LEARN ? DWX_S8_LEARN_WG : DWX_S8_INFER_WG)) sweep8_kernel(const KernelParams P) {
  sweep8_body<LEARN, K, TAB, RP, MULTI, false>(P, nullptr);
}

// ---------------------------------------------------------------- weight-sorted super-tiles
// What bounded sweep8_kernel on a graph with a million weights was one L2 request per record:
// the weight gathers of a variable-major record stream are random.  Here ONE WORKGROUP of
// SORT_THREADS = 1024 lanes -- the CU's whole LDS, one workgroup per CU -- takes a super-tile: up to
// SUPER_NV_MAX = 16 384 consecutive boolean variables of an all-unary graph (<= 64 tiles), and
// streams the super-tile's records in the order of their WEIGHT IDS (the second, sorted copy of
// graph_compile.cc; SORT_K = 20 coalesced 8-byte loads in flight per lane, the next step's records
// under this step's gathers): the 64 lanes of a wave-instruction gather ascending, neighbouring
// weights, a fraction of a 128-byte line per lane instead of a line each (config 3: 164 k records
// over a 4 MB table = 0.2 L2 requests per record; tools/sorted_bench.hip: 0.285 ms per 10^8 records
// against 0.609 ms for the same loop over unsorted records).  A record adds w * d, d = (sign(hit) -
// sign(miss)) * f, to its owner's potential difference pp - pn in LDS -- a 64-bit fixed-point
// atomic add (pot_fix: integer sums are order-independent, so the result does not depend on
// the sorting and equals what sweep8_kernel's row walks compute; the oracle restates it).
// Then the workgroup walks the super-tile's tiles, 256 lanes per tile and four tiles per pass,
// the per-variable words of the next pass in flight under the current one: the tiles' own
// process_variable (W_FIXSUM: the potential is given) draws, stores, tallies, and -- learning --
// publishes the wave ballots of the pull gradient exactly where sweep8_kernel puts them.
// LDS: 128 KiB of sums + the tile starts + the table of distinct d values (<= 1024; none when UNI).
// Replaces, for these variables, FactorGraph::potential's loop (src/factor_graph.h:127-145) and
// draw_sample (src/gibbs_sampler.h:198-215).

// UNI: the table of distinct d holds ONE value besides entry 0 (every record the same function and
// feature value -- config 3: d = 2): it rides in a scalar register, the per-record ds_read_b64 of the
// table goes away (round 4; the index is still what tells a zero-filled lane past the end: di == 0).
template <bool LEARN, bool UNI = false>
__global__ void __launch_bounds__(SORT_THREADS, SORT_WG_PER_CU)
sorted_sweep_kernel(const KernelParams P, const SuperTile *supers, uint32_t n_supers, const SortRec8 *recs,
                    const double *dvals, uint32_t n_dvals) {
  DWX_DYN_LDS(dyn_lds);
  unsigned long long *s_acc = (unsigned long long *)dyn_lds;            // [SUPER_NV_MAX]
  uint32_t *s_tv = (uint32_t *)(dyn_lds + SUPER_NV_MAX * sizeof(long long));   // [SORT_TV_SLOTS] first variable of every tile (+ end)
  double *s_d = (double *)(dyn_lds + SUPER_NV_MAX * sizeof(long long) + SORT_TV_SLOTS * sizeof(uint32_t));   // [n_dvals]
  const uint32_t t = threadIdx.x;
  if (blockIdx.x >= n_supers) return;
  SuperTile S = supers[blockIdx.x];
  S.tile0 = DWX_UNIFORM(S.tile0); S.ntiles = DWX_UNIFORM(S.ntiles); S.v0 = DWX_UNIFORM(S.v0); S.nv = DWX_UNIFORM(S.nv);
  S.lo = DWX_UNIFORM(S.lo); S.hi = DWX_UNIFORM(S.hi); S.nrec = DWX_UNIFORM(S.nrec);
  const SortRec8 *base = recs + (((uint64_t)S.hi << 32) | S.lo);
  SortRec8 rec[SORT_K];
  DWX_LOAD_SORTED_RECORDS(SORT_K, base, S.nrec, 0u, t, rec);
  const double d_uni = UNI ? dvals[1] : 0.0;    // (uniform address: a scalar load)
  if (!UNI) for (uint32_t i = t; i < n_dvals; i += SORT_THREADS) s_d[i] = dvals[i];
  for (uint32_t i = t; i < S.ntiles; i += SORT_THREADS) s_tv[i] = P.tiles[S.tile0 + i].v0;
  if (t == 0) s_tv[S.ntiles] = S.v0 + S.nv;
  for (uint32_t i = t; i < S.nv; i += SORT_THREADS) s_acc[i] = 0ull;
  __syncthreads();
  constexpr uint32_t STEP = SORT_K * SORT_THREADS;
  for (uint32_t first = 0; first < S.nrec; first += STEP) {
    float w[SORT_K];
#pragma unroll
    for (int k = 0; k < SORT_K; ++k) w[k] = P.w32[rec[k].wid];
    SortRec8 cur[SORT_K];
#pragma unroll
    for (int k = 0; k < SORT_K; ++k) cur[k] = rec[k];
    // (the next step's records under this step's gathers; past the end the descriptor zero-fills)
    DWX_LOAD_SORTED_RECORDS(SORT_K, base, S.nrec, first + STEP, t, rec);
#pragma unroll
    for (int k = 0; k < SORT_K; ++k) {
      const uint32_t di = cur[k].od >> SORT_OWNER_BITS;     // 0: a lane past the end
      const long long q = pot_fix((double)w[k] * (UNI ? d_uni : s_d[di]));
      if (di) atomicAdd(&s_acc[cur[k].od & SORT_OWNER_MASK], (unsigned long long)q);
    }
  }
  // the draws: BLOCK_THREADS lanes per tile, SORT_THREADS / BLOCK_THREADS tiles at a time; the
  // per-variable words of the next pass are in flight under the current one (branch-free loads,
  // clamped inside the super-tile)
  constexpr uint32_t TPB = SORT_THREADS / BLOCK_THREADS;
  const uint32_t lane = t & (BLOCK_THREADS - 1), sub = t / BLOCK_THREADS;
  struct Pass { bool have; uint32_t tile, p; bool live; VarPre pre; };
  auto issue = [&](uint32_t j0) {
    Pass ps;
    ps.have = j0 + sub < S.ntiles;
    const uint32_t j = ps.have ? j0 + sub : 0u;
    const uint32_t tv0 = s_tv[j], tnv = s_tv[j + 1] - tv0;
    ps.tile = S.tile0 + j;
    ps.live = ps.have && lane < tnv;
    ps.p = tv0 + (lane < tnv ? lane : tnv - 1u);
    ps.pre = load_var_pre<LEARN>(P, ps.p);
    return ps;
  };
  Pass nxt = issue(0u);
  __syncthreads();     // every record's term is in its owner's sum
  for (uint32_t j0 = 0; j0 < S.ntiles; j0 += TPB) {
    const Pass ps = nxt;
    nxt = issue(j0 + TPB < S.ntiles ? j0 + TPB : j0);
    int delta = 0;
    if (ps.live) {
      double A, B;
      philox_uniforms(P.seed, P.vid_offset + ps.pre.orig, P.sweep, A, B);
      const double x = pot_unfix((long long)s_acc[ps.p - S.v0]);
      TileView T{nullptr, 0u, nullptr, 0u, nullptr, nullptr, nullptr};
      T.presum = &x;
      delta = process_variable<LEARN, W_FIXSUM, true>(P, T, ps.p, ps.pre, A, B, true);
    }
    if (LEARN) {
      const unsigned long long nz = DWX_BALLOT(delta != 0), ng = DWX_BALLOT(delta < 0);
      if (ps.have && (t & 63u) == 0) {
        unsigned long long *wd = P.delta + ((size_t)ps.tile * 4 + (lane >> 6)) * 2;
        DWX_NT_STORE(nz, &wd[0]); DWX_NT_STORE(ng, &wd[1]);
      }
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

}  // namespace dwx

#include "aux_kernels.h"
#include "persist_kernels.h"

#endif  // DWX_SWEEP_KERNELS_H_
