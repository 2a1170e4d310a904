// aux_kernels.h -- the kernels around the sweep: pull gradient (list and block form), apply
// (update_weight), terms tables, weight refresh / averaging, halo pack / unpack, test hooks.
// Part of sweep_kernels.h.
#ifndef DWX_AUX_KERNELS_H_
#define DWX_AUX_KERNELS_H_

#include "tile_walk.h"

namespace dwx {

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
struct alignas(16) F32x4 { float v[4]; };

// One lane owns PULL_RUN consecutive entries (a multiple of 4: 16-byte loads straight
// from HBM; a wave covers one contiguous 4 KiB span per array, every line is consumed
// fully across the lane's loads), gathers their owners' bits (one 16-byte L2 hit each,
// all in flight), sums per weight run in registers and flushes one atomic per run.
// No LDS, no barrier.
// SEG: weights own long stretches of the list (host: entries per weight >= 8) -- the lanes' last
// runs are summed per weight across the wave before the atomic; else every lane flushes its own.
template <bool SEG>
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
    if (SEG) {
      bool head;
      const long long total = DWX_WAVE_SEG_SUM_I64(valid ? cur : 0xFFFFFFFFu, valid ? acc : 0ll, head);
      if (head && total) atomicAdd((unsigned long long *)&grad[cur], (unsigned long long)total);
    } else if (acc && valid) {
      atomicAdd((unsigned long long *)&grad[cur], (unsigned long long)acc);
    }
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
  // UNIFORM: one BYTE per (block, weight) -- the count of +1 / -1 steps; fold_partials_kernel multiplies
  // (an eighth of the partial sums' traffic: 160 MB written and read again per sweep on config 3)
  signed char *__restrict__ out8 = (signed char *)partial + (size_t)b * Wp;
  // BP_UNROLL groups per step: all their row loads are issued before the first is used (the
  // compiler does not hoist them over the stores on its own); past the end the last group is
  // loaded again and not stored
  for (uint32_t g = g0; g < g1; g += BP_UNROLL) {
    U32x4 row[BP_UNROLL][DEPTH];
#pragma unroll
    for (uint32_t u = 0; u < BP_UNROLL; ++u) {
      const uint32_t w = umin(g + u, g1 - 1) * BP_THREADS + tid;
#pragma unroll
      for (int dd = 0; dd < DEPTH; ++dd) row[u][dd] = DWX_LOAD_ROW_NT(&rows[(size_t)dd * Wp + w]);
    }
#pragma unroll
    for (uint32_t u = 0; u < BP_UNROLL; ++u) {
      long long acc = 0;
      int cnt = 0;   // UNIFORM: the sum in units of the one step (at most BP_ROW * DEPTH entries: a byte)
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
          if (UNIFORM) {
            cnt += nz ? (ng ? -1 : 1) : 0;
          } else {
            const long long t = ng ? -q : q;
            acc += nz ? t : 0;
          }
        }
      }
      if (g + u < g1) {
        if (UNIFORM) DWX_NT_STORE((signed char)cnt, &out8[(g + u) * BP_THREADS + tid]);
        else DWX_NT_STORE(acc, &out[(g + u) * BP_THREADS + tid]);
      }
    }
  }
}

// grad[w] += sum over blocks of partial[block][w]  (UNIFORM: byte counts of the one step q0)
template <bool UNIFORM>
__global__ void __launch_bounds__(BLOCK_THREADS)
fold_partials_kernel(const long long *partial, uint32_t n_blocks, uint32_t Wp, uint32_t W, long long *grad,
                     const long long *qtab) {
  const uint32_t stride = gridDim.x * blockDim.x;
  const long long q0 = qtab[0];
  for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < W; w += stride) {
    long long acc = 0;
    if (UNIFORM) {
      int cnt = 0;
      for (uint32_t b = 0; b < n_blocks; ++b) cnt += ((const signed char *)partial)[(size_t)b * Wp + w];
      acc = (long long)cnt * q0;
    } else {
      for (uint32_t b = 0; b < n_blocks; ++b) acc += partial[(size_t)b * Wp + w];
    }
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
// (the flow has converged within the batch).
// Also refreshes the f32 sampling copy of each weight it changes.
DWX_DEV double saturating_step(double c, double stepsize) {
  return c > 0.0 ? -expm1(-c * stepsize) / c : stepsize;
}
// h is an upper BOUND of the batch's curvature; the reference's visits feel the true one,
// somewhere in [0, h].  While no step can overshoot anyway (s (h + r) <= 1: monotone approach
// whatever the truth) the flow is integrated with the middle of that interval, CURV_MID h --
// half the worst-case lag behind the reference in a transient -- and the step is capped at
// 1 / (h + r), the converged flow under the bound.  r = the exact part (L2 pull, reg T).
#ifndef DWX_CURV_MID
#define DWX_CURV_MID 0.5
#endif
DWX_DEV double batch_step(double h, double r, double stepsize) {
  const double s = saturating_step(DWX_CURV_MID * h + r, stepsize), cb = h + r;
  return s * cb > 1.0 ? 1.0 / cb : s;
}
// L1 (src/inference_result.h:76-78): every visit adds reg_param -- NOT scaled by the step --
// while the weight is negative, then takes the gradient step.  Over a batch of T visits that is
// the flow of a piecewise-linear field,
//   w < 0:   dw/dtau = P - G - h (w - w0),   P = reg_param T / stepsize
//   w >= 0:  dw/dtau =     - G - h (w - w0),                     tau in [0, stepsize]
// integrated piece by piece: the push STOPS at the zero crossing.  Where the two fields point
// at each other across zero (the gradient wants the weight negative, the push is stronger) the
// reference's weight rides a sawtooth around zero -- w <- w - d + reg_param [w < 0] with
// d = stepsize G(0) / T per visit, a rotation whose values fill [-d, reg_param - d) evenly; a
// batch that ends there ends at that sawtooth's mean, reg_param / 2 - d.
DWX_DEV double l1_flow(double w0, double G, double h, double T, double stepsize, double reg_param) {
  if (!(stepsize > 0.0)) return w0;
  const double P = reg_param * T / stepsize, hm = DWX_CURV_MID * h;
  double w = w0, left = stepsize;
  for (int piece = 0; piece < 3; ++piece) {
    const double aP = -G - hm * (w - w0), aN = aP + P;  // the two fields at w
    double a;
    if (w < 0.0 || (w == 0.0 && aN <= 0.0)) a = aN;
    else if (w > 0.0 || aP > 0.0) a = aP;
    else return 0.5 * reg_param + aP * stepsize / T;    // sliding: aN > 0 >= aP, d = -aP stepsize / T
    double phi = hm > 0.0 ? -expm1(-hm * left) / hm : left;
    if (phi * h > 1.0) phi = 1.0 / h;                   // as batch_step: never past the bound's fixed point
    const double end = w + a * phi;                     // dw/dtau = a - hm (w(tau) - w)
    if (w == 0.0 || (w < 0.0) == (end <= 0.0)) return end;   // no crossing: monotone towards its fixed point
    const double q = -w / a;                            // phi(tau_c): the flow reaches zero
    left -= hm > 0.0 ? -log1p(-hm * q) / hm : q;
    w = 0.0;
    if (!(left > 0.0)) return 0.0;
  }
  return w;
}
// Where the batch's curvature is negligible (h stepsize small: weights with few visits per batch --
// the untied weights of a DeepDive graph) the flow's "the push stops at zero" is NOT what the
// reference does: a visit's push is a jump of the whole reg_param (w = -0.001, one visit, no
// gradient: the reference lands on 0.009 and stays).  There the reference's own recurrence
//   w += reg_param [w < 0];  w -= stepsize g          (src/inference_result.h:76-78)
// is followed exactly over the batch's n = T visits with the batch's mean gradient per visit, in
// closed form: rise by reg - d per visit while negative, fall by d per visit while not, and once
// both have happened w <- w - d + reg [w < 0] is a rotation of x = w + d on [0, reg).  A batch that
// spans a whole period of that rotation ends at its mean (the visits' own noise decides the phase),
// a shorter one at the rotation's exact end point.  (T is real: truthiness-weighted.)
DWX_DEV double l1_visits(double w, double G, double T, double stepsize, double reg_param) {
  double n = T;
  const double d = stepsize * G / T, up = reg_param - d;
  if (w < 0.0) {
    if (!(up > 0.0)) return w + n * up;          // the gradient outweighs the push: it keeps falling
    const double k = ceil(-w / up);              // visits until it is >= 0
    if (k >= n) return w + n * up;
    w += k * up; n -= k;                         // in [0, up)
  }
  if (!(d > 0.0)) return w - n * d;              // nothing pulls it below zero again: no more pushes
  const double k = floor(w / d) + 1.0;           // visits until it is negative
  if (k > n) return w - n * d;
  w -= k * d; n -= k;                            // in [-d, 0)
  if (!(up > 0.0)) return w + n * up;
  if (n * d >= reg_param) return 0.5 * reg_param - d;
  double x = fmod(w + d - n * d, reg_param);
  if (x < 0.0) x += reg_param;
  return x - d;
}
// (curvature "negligible": the flow's step differs from the plain one by less than 1/32 of itself)
DWX_DEV double l1_update(double w0, double G, double h, double T, double stepsize, double reg_param) {
  return DWX_CURV_MID * h * stepsize <= 0.0625 ? l1_visits(w0, G, T, stepsize, reg_param)
                                               : l1_flow(w0, G, h, T, stepsize, reg_param);
}
// InferenceResult::update_weight (src/inference_result.h:66-85) for one weight over one batch:
// G, Td = the batch's fixed-point gradient sum and dynamic update count; returns the new value
// (x itself for a fixed weight or one nobody visited).
DWX_DEV double apply_value(double x, bool fixed, const long long *t_static, const long long *t_hess, uint32_t i,
                           long long G, long long Td, double stepsize, double reg_param, int l2) {
  const long long Tn = Td + (t_static ? t_static[i] : 0);
  if (fixed || Tn == 0) return x;
  const double Tt = (double)Tn / FIX_SCALE, Gg = (double)G / FIX_SCALE;
  const double h = t_hess ? (double)t_hess[i] / H_SCALE : 0.0;
  if (l2) return x - batch_step(h, reg_param * Tt, stepsize) * (Gg + reg_param * Tt * x);
  return l1_update(x, Gg, h, Tt, stepsize, reg_param);
}
DWX_DEV void apply_one(double *weights, float *w32, const uint8_t *w_fixed, const long long *t_static,
                       const long long *t_hess, uint32_t i, long long G, long long Td, double stepsize,
                       double reg_param, int l2) {
  const long long Tn = Td + (t_static ? t_static[i] : 0);
  if (w_fixed[i] || Tn == 0) return;
  const double x = apply_value(weights[i], false, t_static, t_hess, i, G, Td, stepsize, reg_param, l2);
  weights[i] = x;
  w32[i] = (float)x;
}
// w_in (the last update of a sweep whose mini-batches ran as merged launches, persist_kernels.h): the
// weights before this update live in another buffer -- every weight is written, touched or not;
// also_zero: a second gradient buffer to clear (the one the last merged launch read).
__global__ void __launch_bounds__(BLOCK_THREADS)
apply_kernel(double *weights, float *w32, const uint8_t *w_fixed, long long *grad,
             const long long *t_static, const long long *t_hess, uint32_t W, double stepsize,
             double reg_param, int l2, const double *w_in = nullptr, long long *also_zero = nullptr) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < W; i += stride) {
    const long long G = grad[i], Td = grad[W + i];
    if (G != 0 || Td != 0) { grad[i] = 0; grad[W + i] = 0; }
    if (also_zero) { also_zero[i] = 0; also_zero[W + i] = 0; }
    if (w_in) {
      const double x = apply_value(w_in[i], w_fixed[i] != 0, t_static, t_hess, i, G, Td, stepsize, reg_param, l2);
      weights[i] = x;
      w32[i] = (float)x;
    } else {
      apply_one(weights, w32, w_fixed, t_static, t_hess, i, G, Td, stepsize, reg_param, l2);
    }
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

// Multi-GPU, all-boolean all-unary graphs: every contribution to a gradient sum is a multiple of
// 2^shift (dwx_graph_info.grad_shift), so the sums travel through the all-reduce as 32-bit counts
// -- half the bytes -- and come back shifted.  `bad` is raised if a sum is not such a multiple or
// does not fit (it never is by construction; dwx_wait reports it).
__global__ void __launch_bounds__(BLOCK_THREADS)
grad_pack32_kernel(const long long *grad, int *out, uint32_t W, uint32_t shift, uint32_t *bad) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < W; i += stride) {
    const long long g = grad[i], v = g >> shift;
    if ((long long)((unsigned long long)v << shift) != g || v != (long long)(int)v) atomicAdd(bad, 1u);
    out[i] = (int)v;
  }
}
__global__ void __launch_bounds__(BLOCK_THREADS)
grad_unpack32_kernel(const int *in, long long *grad, uint32_t W, uint32_t shift) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < W; i += stride)
    grad[i] = (long long)((unsigned long long)(long long)in[i] << shift);
}

// The same as 16-bit counts, two per 32-bit word (RCCL has no 16-bit integer type): word i =
// c[2i] + 65536 * c[2i + 1] as ARITHMETIC (a negative low count borrows from the high one), so that
// the all-reduce's plain 32-bit sums of the words ARE the packed sums of the counts as long as every
// summed count stays inside (-2^15, 2^15) -- what the ranks agreed on before choosing this form
// (sum over ranks of max_records_per_weight x grad_unit_max < 2^15).  A quarter of the int64 bytes.
__global__ void __launch_bounds__(BLOCK_THREADS)
grad_pack16_kernel(const long long *grad, int *out, uint32_t W, uint32_t shift, uint32_t *bad) {
  const uint32_t stride = gridDim.x * blockDim.x, words = (W + 1u) / 2u;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < words; i += stride) {
    const long long g0 = grad[2 * i], g1 = 2 * i + 1 < W ? grad[2 * i + 1] : 0;
    const long long v0 = g0 >> shift, v1 = g1 >> shift;
    if ((long long)((unsigned long long)v0 << shift) != g0 || (long long)((unsigned long long)v1 << shift) != g1 ||
        v0 <= -32768 || v0 >= 32768 || v1 <= -32768 || v1 >= 32768)
      atomicAdd(bad, 1u);
    out[i] = (int)(v0 + v1 * 65536);
  }
}
__global__ void __launch_bounds__(BLOCK_THREADS)
grad_unpack16_kernel(const int *in, long long *grad, uint32_t W, uint32_t shift) {
  const uint32_t stride = gridDim.x * blockDim.x, words = (W + 1u) / 2u;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < words; i += stride) {
    const int x = in[i];
    const int lo = (int)(short)(x & 0xFFFF);        // the low count, sign-extended
    const int hi = (x - lo) >> 16;                  // exact: x - lo is a multiple of 65536
    grad[2 * i] = (long long)((unsigned long long)(long long)lo << shift);
    if (2 * i + 1 < W) grad[2 * i + 1] = (long long)((unsigned long long)(long long)hi << shift);
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
#endif  // DWX_AUX_KERNELS_H_
