// persist_kernels.h -- a split learning sweep of a few-weights graph as ONE persistent launch.
// OPT-IN (DWX_PERSIST=1 at dwx_sampler_create): built, exact, measured -- and slower than the launches
// it replaces, as the CDNA guide's price list predicts (a grid barrier costs more than a kernel boundary).
//
// A graph whose weights are tied to millions of factors each (config 4 with learning: 8 weights x
// 2.5 M evidence factors) learns in up to 64 mini-batches per sweep (inside a batch
// all draws see the weights of its start -- the reference updates after every factor visit,
// src/inference_result.h:66-85).  As separate launches a mini-batch is a sweep kernel + an update
// kernel, 18 + 5 us, 64 times.  Here ONE launch of at most one workgroup per CU runs all chunks:
//   * a workgroup walks its tiles of chunk c (tile t0 + b, t0 + b + grid, ...: the tiles' own staging
//     and process_variable, gradient sums in LDS), with the loads of its NEXT tile -- the first one
//     of chunk c + 1 when this chunk is done -- already in flight: records, row pointers and
//     per-variable words do not depend on the weights;
//   * it stores its 2W sums as ONE ROW (write-through stores, rows[c & 1][b]), drains, signals one
//     arrival on a counter, polls until all `grid` workgroups of the chunk have arrived, acquires;
//   * EVERY workgroup then adds up all rows and applies the update to its own LDS copy of the weights
//     (integer sums, the same update in every workgroup: identical weights everywhere without a
//     broadcast), and goes on with chunk c + 1, whose weight gathers read that LDS copy.
// Workgroup 0 writes the weights back at the end.  Results are bit for bit those of the chunk-by-chunk
// launches (same sums, same update function, same Philox counters): tests/test_gpu_parity.py.
// Every spin is bounded: a workgroup that gives up raises a flag all others leave on, and dwx_wait
// reports it (seen with three workgroups per CU asked for: not all resident).  The hand-off follows the
// CDNA guide's Guideline 16, recipe R1 (device_intrinsics.h).
// Measured on config 4 with learning (profiles/r04/v5/): 2.02 ms per sweep with one workgroup per CU
// (31 us per chunk: two tiles one after the other on most CUs, row store + drain, a 256-way fan-in on
// one counter + the poll, the acquire, 256 rows read back by every workgroup), 2.49 ms with two per
// CU, against 1.59 ms for the 128 plain launches -- hence off.
#ifndef DWX_PERSIST_KERNELS_H_
#define DWX_PERSIST_KERNELS_H_

namespace dwx {

template <int K, int RP>
__global__ void __launch_bounds__(BLOCK_THREADS, 1) persist_learn8_kernel(const KernelParams P, const PersistArgs A) {
  DWX_DYN_LDS(dyn_lds);
  uint32_t *s_rowptr = (uint32_t *)dyn_lds;
  double *s_pot = (double *)(dyn_lds + P.lds_pot_off);
  EdgeRec *s_edges = (EdgeRec *)(dyn_lds + P.lds_edge_off);
  float *s_w = (float *)(dyn_lds + P.lds_w_off);
  long long *s_agg = (long long *)(dyn_lds + P.lds_agg_off);
  double *s_w64 = (double *)(dyn_lds + A.lds_w64_off);
  float *s_lw32 = (float *)(s_w64 + P.num_weights);
  long long *s_red = (long long *)(dyn_lds + A.lds_red_off);       // [BLOCK_THREADS]
  uint32_t *s_flag = (uint32_t *)(s_red + BLOCK_THREADS);
  const uint32_t t = threadIdx.x, b = blockIdx.x, G = A.grid, W = P.num_weights, n2 = 2u * W;
  for (uint32_t i = t; i < W; i += BLOCK_THREADS) { s_w64[i] = A.weights[i]; s_lw32[i] = A.w32[i]; }
  // lanes of the in-launch update: component j of the 2W sums, row lane rl (n2c: 2W rounded up to a power of two)
  uint32_t n2c = 1;
  while (n2c < n2) n2c <<= 1;
  const uint32_t jc = t & (n2c - 1u), rl = t / n2c, RL = BLOCK_THREADS / n2c;
  TileDesc d{};
  TilePrefetch<K, EdgeRec8, RP> f;
  bool pre_ok = false;          // f / d hold the loads of tile pre_tile (issued before the last barrier)
  uint32_t pre_tile = 0;
  for (uint32_t c = 0; c < A.n_chunks; ++c) {
    const uint32_t t0 = A.chunk_tiles[2 * c], t1 = A.chunk_tiles[2 * c + 1];
    for (uint32_t i = t; i < n2; i += BLOCK_THREADS) s_agg[i] = 0;
    uint32_t tile = t0 + b;
    bool have = tile < t1;
    if (have && !(pre_ok && pre_tile == tile)) {
      d = scalarise(P.tiles[tile]);
      issue_tile_loads<true, K, true>(P, d, t, f);
    }
    pre_ok = false;
    __syncthreads();     // accumulators cleared, the weights of the last update in place
    while (have) {
      // the next tile of this workgroup: in this chunk, else its first one of the next chunk
      uint32_t nt = tile + G;
      bool hn = nt < t1, cross = false;
      if (!hn && c + 1 < A.n_chunks) {
        nt = A.chunk_tiles[2 * c + 2] + b;
        hn = nt < A.chunk_tiles[2 * c + 3];
        cross = hn;
      }
      const TileDesc raw_n = P.tiles[hn ? nt : tile];      // (vector registers: no wait yet)
      const VarPre pre = f.pre;
      double Au = 0.0, Bu = 0.0;
      {
        float w[K];
#pragma unroll
        for (int k = 0; k < K; ++k) w[k] = s_lw32[f.rec[k].key & REC8_WID_MASK];
        philox_uniforms(P.seed, P.vid_offset + pre.orig, P.sweep, Au, Bu);
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const uint32_t i = t + k * BLOCK_THREADS;
          s_w[i] = w[k];
          s_edges[i] = expand_record(f.rec[k]);
        }
#pragma unroll
        for (uint32_t k = 0; k < (uint32_t)RP; ++k) s_rowptr[t + k * BLOCK_THREADS] = f.rp[k];
        for (uint32_t i = t + RP * BLOCK_THREADS; i <= d.nrows; i += BLOCK_THREADS)
          s_rowptr[i] = (d.flags & TILE_UNIT_ROWS) ? d.e0 + i : P.row_ptr[d.r0 + i];
      }
      __syncthreads();
      // the next tile's loads under this tile's draws
      TileDesc dn = scalarise(raw_n);
      if (!hn) { dn.nedges = 0; dn.nrows = 0; dn.nv = 1; }
      issue_tile_loads<true, K, true>(P, dn, t, f);
      if (t < d.nv) {
        TileView T{s_rowptr, d.r0, s_edges, d.e0, s_w, s_agg, P.lds_pot_off ? s_pot : nullptr};
        process_variable<true, W_ARRAY, true, true>(P, T, d.v0 + t, pre, Au, Bu, false);
      }
      __syncthreads();   // LDS is rewritten by the next tile / read by the row store
      d = dn;
      if (cross) { pre_ok = true; pre_tile = nt; have = false; }
      else { tile = nt; have = hn; }
    }
    // this workgroup's sums of chunk c: one row, whole 128-byte lines per store instruction, write-through
    {
      long long *row = A.rows + ((size_t)(c & 1u) * G + b) * A.row_stride;
      if (t < 64u)
        for (uint32_t i = t; i < A.row_stride; i += 64u) DWX_AGENT_STORE_I64(&row[i], i < n2 ? s_agg[i] : 0LL);
      DWX_DRAIN_VMEM();
      __syncthreads();
      if (t == 0) {
        (void)DWX_AGENT_ADD_U32(&A.bar[0], 1u);
        const uint32_t target = G * (c + 1u);
        uint32_t ok = 1u, spins = 0;
        while (DWX_AGENT_LOAD_U32(&A.bar[0]) < target) {
          if (DWX_AGENT_LOAD_U32(&A.bar[1]) != 0u || ++spins > A.spin_limit) {
            DWX_AGENT_STORE_U32(&A.bar[1], 1u);     // (sticky: every workgroup leaves at its next poll)
            ok = 0u;
            break;
          }
          DWX_SLEEP();
        }
        *s_flag = ok;
        DWX_ACQUIRE_AGENT();
      }
      __syncthreads();
      if (*s_flag == 0u) return;         // (workgroup-uniform: every wave of the workgroup leaves)
    }
    // the update of chunk c, by every workgroup for itself: sum the rows, apply to the LDS weights
    {
      long long acc = 0;
      const long long *rows = A.rows + (size_t)(c & 1u) * G * A.row_stride;
      if (jc < n2)
        for (uint32_t r = rl; r < G; r += RL) acc += rows[(size_t)r * A.row_stride + jc];
      s_red[t] = acc;
      __syncthreads();
      for (uint32_t half = RL / 2u; half >= 1u; half >>= 1) {
        if (rl < half) s_red[t] += s_red[t + half * n2c];
        __syncthreads();
      }
      if (t < W) {
        const long long *ts = A.t_static + (size_t)c * n2;
        const double x = apply_value(s_w64[t], A.w_fixed[t] != 0, ts, ts + W, t, s_red[t], s_red[W + t], A.stepsize,
                                     A.reg_param, A.l2);
        s_w64[t] = x;
        s_lw32[t] = (float)x;
      }
      // (the next chunk's first __syncthreads orders these writes before any gather)
    }
  }
  __syncthreads();
  if (b == 0)
    for (uint32_t i = t; i < W; i += BLOCK_THREADS) { A.weights[i] = s_w64[i]; A.w32[i] = s_lw32[i]; }
}

// ---------------------------------------------------------------- one launch per mini-batch
// What the persistent launch above could not buy, the other way round (round 4): a mini-batch of a split
// sweep stays a launch of its own -- a kernel boundary is cheaper than a grid barrier -- but there is ONE
// launch per mini-batch instead of two: the update of mini-batch c - 1 (apply_kernel's arithmetic,
// apply_value) runs as the PROLOGUE of mini-batch c's sweep kernel, by every workgroup for itself, from the
// previous launch's gradient sums (complete: the launch has ended) into an LDS copy of the f32 weights that
// the tile loop gathers from.  Workgroup 0 also writes the new weights out (to the OTHER of two buffers:
// its neighbours may still be reading the old ones) and zeroes the buffer the next mini-batch will add
// into (three gradient buffers in turn).  A dependent launch costs ~4 us here whatever it does
// (profiles/r04/v5): 64 of them less per sweep of config 4.  Bit for bit the two-launch path.
// The update runs once the tile's own loads are in flight (they need no weight): its few loads return under the
// record stream's instead of ahead of it.  ONE: the launch has a workgroup per tile (sweep8_body).
template <int K, int RP, bool ONE = false>
__global__ void __launch_bounds__(BLOCK_THREADS, DWX_S8_LEARN_WG) sweep8_merged_kernel(const KernelParams P, const MergeArgs M) {
  DWX_DYN_LDS(dyn_lds);
  float *s_lw32 = (float *)(dyn_lds + M.lds_lw32_off);
  auto update = [&]() {
    const uint32_t W = P.num_weights;
    for (uint32_t i = threadIdx.x; i < W; i += BLOCK_THREADS) {
      double x = M.w_src[i];
      if (M.prev_grad) {
        x = apply_value(x, M.w_fixed[i] != 0, M.t_static, M.t_static ? M.t_static + W : nullptr, i, M.prev_grad[i],
                        M.prev_grad[W + i], M.stepsize, M.reg_param, M.l2);
        if (blockIdx.x == 0) { M.w_dst[i] = x; M.w32_dst[i] = (float)x; }
      }
      s_lw32[i] = (float)x;
    }
    if (blockIdx.x == 0 && M.zero)
      for (uint32_t i = threadIdx.x; i < 2u * W; i += BLOCK_THREADS) M.zero[i] = 0;
    __syncthreads();
  };
  sweep8_body<true, K, false, RP, false, true, ONE, decltype(update)>(P, s_lw32, update);
}

}  // namespace dwx
#endif  // DWX_PERSIST_KERNELS_H_
