// dwx_api.cc -- implementation of the C ABI in include/dwx.h.
//
// Compiled as HIP for gfx950 into libdwx.so (Makefile).  The runtime layer is
// rt_hip.h; when DWX_EMU is defined (tests/hipemu only) the same file is compiled by
// g++ against a host emulation of that layer so the kernels run under sanitizers.
#include "../../include/dwx.h"

#include <algorithm>
#include <array>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "device_build.h"
#include "graph_compile.h"
#include "host_parallel.h"

#include "sweep_kernels.h"   // (device_intrinsics.h: the HIP runtime shim rt_hip.h comes with it)

// a finer cut must lower the batches' curvature estimate to this fraction, else the plan stops
// (dwx_sgd_plan; multi-GPU drivers apply the same rule to the global estimate)
constexpr double PLAN_MIN_GAIN = 0.8;

#ifndef DWX_PULL_GRID
#define DWX_PULL_GRID 16u   // pull_grad_kernel workgroups per CU (grid-stride beyond)
#endif

using namespace dwx;

namespace {
thread_local std::string g_err;

int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}

template <class Fn>
int guarded(Fn &&fn) {
  try {
    fn();
    return DWX_OK;
  } catch (const std::bad_alloc &) {
    return fail(DWX_E_NOMEM, "out of host memory");
  } catch (const std::invalid_argument &e) {
    return fail(DWX_E_INVALID, e.what());
  } catch (const std::exception &e) {
    return fail(DWX_E_DEVICE, e.what());
  }
}

// `pad` extra zeroed elements behind the data (the kernels' clamped, branch-free loads
// may touch one element past an empty tile)
template <class T>
T *upload(const std::vector<T> &v, rt::stream_t s, size_t pad = 0) {
  T *d = (T *)rt::dmalloc((v.size() + pad) * sizeof(T));
  rt::h2d(d, v.data(), v.size() * sizeof(T), s);
  if (pad) rt::dmemset(d + v.size(), 0, pad * sizeof(T), s);
  return d;
}

template <class T>
T *upload(const RawArray<T> &v, rt::stream_t s, size_t pad = 0) {
  T *d = (T *)rt::dmalloc((v.size() + pad) * sizeof(T));
  rt::h2d(d, v.data(), v.size() * sizeof(T), s);
  if (pad) rt::dmemset(d + v.size(), 0, pad * sizeof(T), s);
  return d;
}

template <class T>
T *upload_raw(const T *h, size_t n, rt::stream_t s) {
  T *d = (T *)rt::dmalloc(n * sizeof(T));
  rt::h2d(d, h, n * sizeof(T), s);
  return d;
}

struct TimedSpan {
  rt::event_t a, b, c;   // a..b: sweep kernels of all colours; b..c: pull_grad_kernel (learning)
  int kind;
  uint32_t launches;
  bool has_pull;
  bool new_sweep;   // the span opens a sweep (inference: always; learning: its first chunk)
  uint32_t n_sweeps = 1;   // ... or several (dwx_sample_n_async on an all-unary graph)
};
}  // namespace

struct dwx_graph {
  std::shared_ptr<CompiledGraph> cg;
};

struct dwx_sampler {
  std::shared_ptr<CompiledGraph> cg;
  dwx_options opts;
  int device = 0;
  rt::stream_t stream = nullptr;
  // The three degree bins of a launch are independent sets of one colour: the wave-per-variable and
  // workgroup-per-variable kernels run beside the tiles' sweep kernel on two side streams, forked
  // from and joined into `stream` with events (a hub graph's chunk: 180 -> 85 us).
  rt::stream_t side[2] = {nullptr, nullptr};
  rt::event_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
  // device buffers
  uint32_t *d_v_meta = nullptr, *d_v_orig = nullptr, *d_v_row = nullptr, *d_v_init = nullptr;
  uint32_t *d_row_ptr = nullptr;
  TileDesc *d_tiles = nullptr;
  uint32_t *d_giant = nullptr, *d_wide = nullptr;
  // oversized variables: categorical ones -> giant_kernel (one workgroup each, d_giant);
  // boolean ones -> pieces of GIANT_PIECE records over several workgroups (giant_pot_kernel,
  // giant_decide_kernel, giant_grad_kernel)
  std::vector<uint32_t> cgiant_tiles, bgiant_tiles, bgiant_piece_off;
  uint32_t *d_bgiant = nullptr, *d_bgiant_piece_off = nullptr, *d_bgiant_decision = nullptr;
  GiantPiece *d_bgiant_pieces = nullptr;
  double *d_bgiant_partial = nullptr;
  int stage_k = 12;
  // learning-sweep plan (dwx_sgd_plan)
  struct Chunk { uint32_t launch, t0, t1; };
  std::vector<Chunk> plan_chunks;
  uint32_t plan_batches = 1;
  double plan_eta = 0.0;
  bool plan_valid = false;
  std::map<uint32_t, double> row_sum_cache;   // batches -> R
  std::vector<uint64_t> sgd_work;             // [tiles + 1] prefix sums of SGD-visited records
  uint64_t sgd_work_max_launch = 0;           // ... of the colour launch with the most
  bool wide_learn = false;   // the graph has TILE_TERMS2 tiles: 32-byte staged records when learning
  bool tv_pair = false;      // every lane tile is pre-signed unary and / or inline arity-2: sweep_kernel<.., TV_PAIR>
  unsigned persistent_blocks[2] = {1, 1};
  bool rec8 = false;                    // the graph streams 8-byte records (CompiledGraph::edges8)
  bool rp_cat = false;                  // ... with every row pointer of a categorical tile prefetched (K = 6 builds)
  unsigned persistent_blocks8[2] = {1, 1};
  // every tile that fits is TILE_PULL: a learning sweep with the pull gradient stages 16-byte
  // terms only (no f32 weight array behind the records) -- smaller LDS, one more workgroup per CU
  bool all_pull = false;
  size_t lds_learn_pull = 0;
  unsigned persistent_blocks_pull = 1;   // REC8 learning kernel at lds_learn_pull
  size_t lds_tab = 0;                    // inference on the 8-byte terms table: 8 bytes staged per record
  unsigned persistent_blocks_tab = 1;
  EdgeRec8 *d_edges8 = nullptr;
  // weight-sorted super-tiles (sorted_sweep_kernel)
  SortRec8 *d_sorted = nullptr;
  SuperTile *d_supers = nullptr;
  double *d_sort_dvals = nullptr;
  uint32_t n_sort_dvals = 0;
  size_t lds_sorted = 0;
  bool sorted_learn = false;            // ... also in learning sweeps (the super-tiles' tiles pull their gradient)
  bool no_sort_uni = false;             // DWX_NO_SORT_UNI: never the single-d build of sorted_sweep_kernel (A/B)
  bool sorted_fallback_said = false;    // the "more than 8 runs: tile sweep instead" note was printed
  double *d_row_truth = nullptr, *d_edge_fval64 = nullptr;
  EdgeRec *d_edges = nullptr;
  VifRec *d_vifs = nullptr;
  uint32_t *d_assign_free = nullptr, *d_assign_evid = nullptr, *d_tally = nullptr;
  double *d_weights = nullptr;
  float *d_w32 = nullptr;
  double *d_w_init = nullptr;   // initial weights, uploaded on the first replica averaging
  // tabulated potential terms of the pre-signed records (inference with unchanged weights)
  EdgeTerms *d_terms = nullptr;
  int terms_state = 0;          // 0 weights changed since / 1 one inference sweep ran on them / 2 table valid
  bool has_simple_tiles = false;
  // pull-based gradient (TILE_PULL tiles)
  unsigned long long *d_delta = nullptr;
  // One incidence list + static update counts per plan level (number of batches): sorted by
  // (chunk, weight), every chunk padded to whole runs; t_static holds one row of W counts per
  // chunk (rows >= chunks: a multi-GPU driver sizes it for the slowest rank).  Level 1 (the
  // un-split sweep) is built at create, the others when a plan first needs them.
  struct Level {
    std::vector<Chunk> chunks;
    bool fast = false;                 // lists + static counts present (else: atomics, dynamic T)
    uint32_t *d_inc_wid = nullptr, *d_inc_slot = nullptr;
    float *d_inc_d = nullptr;
    std::vector<uint32_t> inc_begin, inc_end;   // per chunk, entries (multiples of PULL_RUN)
    long long *d_t_static = nullptr;   // [rows][T[W] | h[W]]: static update counts, curvature bounds
    uint32_t rows = 0;
    double c_max = 0.0, t_max = 0.0;   // largest h and largest static T of any (chunk, weight)
    // block pull (pull_ell_kernel; graphs with many weights): per group of the tables (the
    // un-split sweep, or each chunk of a split one) entry rows per (variable block, weight);
    // d_inc_* then only holds what did not fit a row.  blocks == 0: the group pulls its list.
    struct BlockTable {
      U32x4 *d_ell = nullptr;
      uint32_t *d_tile0 = nullptr;
      uint32_t blocks = 0, depth = 0, parts = 0;
    };
    std::vector<BlockTable> bp;           // [groups] or empty
    long long *d_bp_qtab = nullptr, *d_bp_partial = nullptr;   // shared: deltas; [max blocks][Wp] sums
    uint32_t bp_deltas = 0, bp_wp = 0;
    // a split plan's own weight-sorted layout: super-tiles cut ALONG its chunks, as many per chunk
    // as workgroups are resident, so that every chunk's launch of sorted_sweep_kernel is as wide as
    // the chip (the graph's default super-tiles of 64 tiles leave a quarter-sweep chunk 95 workgroups
    // for 256 CUs).  Empty: the chunks use the default layout (whole super-tiles inside a chunk).
    std::vector<SuperTile> sorted_supers;
    SuperTile *d_supers = nullptr;
    SortRec8 *d_sorted = nullptr;
    uint32_t *d_chunk_tiles = nullptr; // [2 * chunks]: the chunks' tile ranges (persist_learn8_kernel)
    bool layout_done = false;          // ensure_level_layout ran (whatever it decided)
    uint64_t sweeps = 0;               // learning sweeps run at this level (plan_layouts == 0: the layout comes after 2048)
    // a split sweep of this level as ONE graph launch (dwx_sample_sgd_async): the instantiated graph,
    // patched in place every sweep; use_graph < 0: replay failed once, the level stays on plain launches
    rt::graph_exec_t gexec = nullptr;
    int use_graph = 0;
    ~Level() {
      rt::graph_exec_destroy(gexec);
      rt::dfree(d_supers); rt::dfree(d_sorted); rt::dfree(d_chunk_tiles);
      rt::dfree(d_inc_wid); rt::dfree(d_inc_slot); rt::dfree(d_inc_d); rt::dfree(d_t_static);
      for (auto &t : bp) { rt::dfree(t.d_ell); rt::dfree(t.d_tile0); }
      rt::dfree(d_bp_qtab); rt::dfree(d_bp_partial);
    }
  };
  std::map<uint32_t, std::unique_ptr<Level>> levels;
  Level *plan_level = nullptr;        // level of the current plan
  bool plan_force_dynamic = false;    // dwx_sgd_plan_force_dynamic (multi-GPU agreement)
  uint32_t cur_chunk = 0;             // last chunk handed to dwx_sgd_accumulate_async
  uint8_t *d_w_fixed = nullptr;
  long long *d_grad = nullptr;
  // a split learning sweep of an all-unary graph with few weights as ONE persistent launch
  // (persist_learn8_kernel; opt-in, DWX_PERSIST=1: measured slower than the chunk-by-chunk launches):
  // grid (<= one workgroup per CU), its rows [2][grid][row_stride], the barrier words, LDS layout
  uint32_t persist_grid = 0, persist_row_stride = 0;
  long long *d_persist_rows = nullptr;
  uint32_t *d_persist_bar = nullptr;
  size_t lds_persist = 0;
  uint32_t persist_w64_off = 0, persist_red_off = 0;
  bool persist_check_pending = false;
  uint64_t persist_launches = 0;
  // ... and with one launch per mini-batch (sweep8_merged_kernel: the update as the next sweep kernel's
  // prologue): three gradient buffers ([0] = d_grad) and two weight buffers in turn
  bool merge_ok = false;
  long long *d_gbuf[3] = {nullptr, nullptr, nullptr};
  double *d_wbuf64[2] = {nullptr, nullptr};
  float *d_wbuf32[2] = {nullptr, nullptr};
  size_t lds_merge = 0;
  uint32_t merge_lw32_off = 0;
  uint64_t merged_sweeps = 0;
  int *d_grad32 = nullptr;              // dwx_grad_pack32_async: the gradient sums as 32-bit counts
  uint32_t *d_pack_bad = nullptr;       // ... and its "not a multiple / does not fit" counter
  bool pack_check_pending = false;
  KernelParams base{};
  size_t lds_bytes[2] = {0, 0};  // [0] inference kernel, [1] learning kernel
  uint64_t sweep = 0;
  uint64_t infer_sweeps = 0;  // inference sweeps since the last clear_tallies
  // kernel timing
  bool timing = false;
  uint64_t graph_launches = 0;        // split learning sweeps handed over as one graph launch
  std::vector<TimedSpan> spans;
  double t_ms[3] = {0, 0, 0};   // [0] inference sweep kernels, [1] learning sweep kernels, [2] pull_grad
  uint64_t t_launches[3] = {0, 0, 0}, t_sweeps[3] = {0, 0, 0};

  ~dwx_sampler() {
    for (auto &sp : spans) { rt::event_destroy(sp.a); rt::event_destroy(sp.b); rt::event_destroy(sp.c); }
    rt::dfree(d_v_meta); rt::dfree(d_v_orig); rt::dfree(d_v_row); rt::dfree(d_v_init);
    rt::dfree(d_row_ptr); rt::dfree(d_tiles); rt::dfree(d_giant); rt::dfree(d_wide); rt::dfree(d_bgiant); rt::dfree(d_bgiant_piece_off);
    rt::dfree(d_bgiant_decision); rt::dfree(d_bgiant_pieces); rt::dfree(d_bgiant_partial); rt::dfree(d_row_truth); rt::dfree(d_edge_fval64);
    rt::dfree(d_sorted); rt::dfree(d_supers); rt::dfree(d_sort_dvals);
    rt::dfree(d_grad32); rt::dfree(d_pack_bad);
    rt::dfree(d_edges); rt::dfree(d_edges8); rt::dfree(d_vifs); rt::dfree(d_assign_free); rt::dfree(d_assign_evid);
    rt::dfree(d_tally); rt::dfree(d_weights); rt::dfree(d_w32); rt::dfree(d_w_init); rt::dfree(d_terms); rt::dfree(d_delta);
    rt::dfree(d_w_fixed); rt::dfree(d_grad); rt::dfree(d_persist_rows); rt::dfree(d_persist_bar);
    rt::dfree(d_gbuf[1]); rt::dfree(d_gbuf[2]); rt::dfree(d_wbuf64[0]); rt::dfree(d_wbuf64[1]); rt::dfree(d_wbuf32[0]); rt::dfree(d_wbuf32[1]);
    for (int i = 0; i < 2; ++i) { if (side[i]) rt::stream_destroy(side[i]); if (ev_join[i]) rt::event_destroy(ev_join[i]); }
    if (ev_fork) rt::event_destroy(ev_fork);
    if (stream) rt::stream_destroy(stream);
  }
};

// One side of a halo exchange with one peer: the device positions of the variables whose
// assignments travel, and the buffer they travel in ([chain][i], both chains' worth).
struct dwx_halo {
  dwx_sampler *s = nullptr;
  uint32_t n = 0;
  uint32_t bits = 32;       // per value: 1 (all boolean), 8 (cardinalities <= 256) or 32
  uint32_t block = 0;       // 8-byte words of one chain's block
  uint32_t *d_pos = nullptr;
  unsigned long long *d_buf = nullptr;
  ~dwx_halo() { rt::dfree(d_pos); rt::dfree(d_buf); }
};

namespace {
// launch the sweep kernel (+ the oversized-variable kernel) over tiles [t0, t1) of launch l
// multi (inference of a graph without degree-binned variables): P.n_sweeps sweeps per launch
template <bool LEARN>
uint32_t launch_tiles(dwx_sampler *s, KernelParams &P, size_t l, uint32_t t0, uint32_t t1, const bool multi = false) {
  const CompiledGraph &c = *s->cg;
  if (t1 <= t0) return 0;
  uint32_t launches = 0;
  P.tile_begin = t0;
  P.tile_end = t1;
  // the oversized and mid-degree variables among these tiles (the lists are sorted by tile index)
  const auto &cgt = s->cgiant_tiles;
  const uint32_t cg0 = (uint32_t)(std::lower_bound(cgt.begin(), cgt.end(), t0) - cgt.begin());
  const uint32_t cg1 = (uint32_t)(std::lower_bound(cgt.begin(), cgt.end(), t1) - cgt.begin());
  const auto &bgt = s->bgiant_tiles;
  const uint32_t bg0 = (uint32_t)(std::lower_bound(bgt.begin(), bgt.end(), t0) - bgt.begin());
  const uint32_t bg1 = (uint32_t)(std::lower_bound(bgt.begin(), bgt.end(), t1) - bgt.begin());
  const uint32_t *wb = c.wide_tiles.data() + c.launch_wide[l], *we = c.wide_tiles.data() + c.launch_wide[l + 1];
  const uint32_t w0 = (uint32_t)(std::lower_bound(wb, we, t0) - c.wide_tiles.data());
  const uint32_t w1 = (uint32_t)(std::lower_bound(wb, we, t1) - c.wide_tiles.data());
  const bool any_giant = cg1 > cg0 || bg1 > bg0, any_wide = w1 > w0;
  // fork: the side streams start where `stream` stands now, beside the tiles' kernel
  // (learning launches only: an inference launch is too short -- the joins cost more than they hide)
  const bool fork = LEARN && s->ev_fork && (any_giant || any_wide);
  rt::stream_t st_giant = fork ? s->side[0] : s->stream, st_wide = fork ? s->side[1] : s->stream;
  if (fork) {
    rt::event_record(s->ev_fork, s->stream);
    if (any_giant) rt::stream_wait_event(st_giant, s->ev_fork);
    if (any_wide) rt::stream_wait_event(st_wide, s->ev_fork);
  }
  // the lane-bin tiles of [a, b): the persistent tile sweep
  auto launch_lane_tiles = [&](uint32_t a, uint32_t b) {
  if (b <= a) return;
  P.tile_begin = a;
  P.tile_end = b;
  const uint32_t t0 = a, t1 = b;
  // persistent grid: as many workgroups as stay resident, each striding over tiles
  // all-unary graph: 8-byte record stream (a run on the terms table streams those instead)
  const bool rec8 = s->rec8;
  const bool tab8 = rec8 && !LEARN && P.edge_terms;   // inference on the 8-byte terms table
  const bool slim = LEARN && rec8 && s->all_pull && !(P.flags & OPT_NO_PULL);
  const unsigned grid = std::min<unsigned>(t1 - t0, tab8 ? s->persistent_blocks_tab : slim ? s->persistent_blocks_pull :
                                           (rec8 ? s->persistent_blocks8 : s->persistent_blocks)[LEARN ? 1 : 0]);
  // (all-unary graphs stage 8-byte terms whenever the compute phase needs no record)
  const size_t lds = tab8 ? s->lds_tab : slim ? s->lds_learn_pull : s->lds_bytes[LEARN ? 1 : 0];
  constexpr int RPC = (int)ROWPTR_UNROLL_CAT;

  if constexpr (!LEARN) {
    if (multi) {   // (only asked for on compact-record graphs, never on the terms table)
      constexpr int RP = (int)ROWPTR_UNROLL;
      if (s->rp_cat) rt::launch(sweep8_kernel<false, 6, false, RPC, true>, grid, BLOCK_THREADS, lds, s->stream, P);
      else switch (s->stage_k) {
        case 3: rt::launch(sweep8_kernel<false, 3, false, RP, true>, grid, BLOCK_THREADS, lds, s->stream, P); break;
        case 6: rt::launch(sweep8_kernel<false, 6, false, RP, true>, grid, BLOCK_THREADS, lds, s->stream, P); break;
        default: rt::launch(sweep8_kernel<false, 12, false, RP, true>, grid, BLOCK_THREADS, lds, s->stream, P); break;
      }
      ++launches;
      return;
    }
  }
  if (rec8 && s->rp_cat) {
    if (tab8) rt::launch(sweep8_kernel<false, 6, true, RPC>, grid, BLOCK_THREADS, lds, s->stream, P);
    else rt::launch(sweep8_kernel<LEARN, 6, false, RPC>, grid, BLOCK_THREADS, lds, s->stream, P);
  } else if (tab8) {
    switch (s->stage_k) {
      case 3: rt::launch(sweep8_kernel<false, 3, true>, grid, BLOCK_THREADS, lds, s->stream, P); break;
      case 6: rt::launch(sweep8_kernel<false, 6, true>, grid, BLOCK_THREADS, lds, s->stream, P); break;
      default: rt::launch(sweep8_kernel<false, 12, true>, grid, BLOCK_THREADS, lds, s->stream, P); break;
    }
  } else if (rec8) {
    switch (s->stage_k) {
      case 3: rt::launch(sweep8_kernel<LEARN, 3>, grid, BLOCK_THREADS, lds, s->stream, P); break;
      case 6: rt::launch(sweep8_kernel<LEARN, 6>, grid, BLOCK_THREADS, lds, s->stream, P); break;
      default: rt::launch(sweep8_kernel<LEARN, 12>, grid, BLOCK_THREADS, lds, s->stream, P); break;
    }
  } else if (s->tv_pair) {
    // every tile pre-signed unary and / or inline arity-2 records (config 3b / 5b): the small build
    rt::launch(sweep_kernel<LEARN, 6, LEARN, TV_PAIR>, grid, BLOCK_THREADS, lds, s->stream, P);
  } else if (LEARN && s->wide_learn) {
    switch (s->stage_k) {
      case 3: rt::launch(sweep_kernel<LEARN, 3, LEARN>, grid, BLOCK_THREADS, lds, s->stream, P); break;
      case 6: rt::launch(sweep_kernel<LEARN, 6, LEARN>, grid, BLOCK_THREADS, lds, s->stream, P); break;
      default: rt::launch(sweep_kernel<LEARN, 12, LEARN>, grid, BLOCK_THREADS, lds, s->stream, P); break;
    }
  } else {
    switch (s->stage_k) {
      case 3: rt::launch(sweep_kernel<LEARN, 3>, grid, BLOCK_THREADS, lds, s->stream, P); break;
      case 6: rt::launch(sweep_kernel<LEARN, 6>, grid, BLOCK_THREADS, lds, s->stream, P); break;
      default: rt::launch(sweep_kernel<LEARN, 12>, grid, BLOCK_THREADS, lds, s->stream, P); break;
    }
  }
  ++launches;
  };
  // Weight-sorted super-tiles lying wholly inside [t0, t1): sorted_sweep_kernel, one workgroup
  // each (not on the terms table: repeated inference gathers nothing; not when this sweep
  // scatters its gradient or counts updates dynamically -- the kernel only publishes ballots);
  // the tiles before, between and behind them: the tile sweep.
  bool used_sorted = false;
  if (s->d_supers && !P.edge_terms && !multi && (!LEARN || (s->sorted_learn && !(P.flags & (OPT_NO_PULL | OPT_DYNAMIC_T))))) {
    // (a chunk of a split learning sweep: the plan level's own layout, cut along the chunks)
    const dwx_sampler::Level *lv = (LEARN && s->plan_level && s->plan_level->d_supers) ? s->plan_level : nullptr;
    const std::vector<SuperTile> &sv = lv ? lv->sorted_supers : c.supers;
    const SuperTile *d_sv = lv ? lv->d_supers : s->d_supers;
    const SortRec8 *d_sr = lv ? lv->d_sorted : s->d_sorted;
    size_t a = std::lower_bound(sv.begin(), sv.end(), t0, [](const SuperTile &x, uint32_t t) { return x.tile0 < t; }) - sv.begin();
    size_t b = a;
    while (b < sv.size() && sv[b].tile0 + sv[b].ntiles <= t1) ++b;
    // gap-free runs of super-tiles
    struct Run { size_t a, b; };
    std::vector<Run> runs;
    for (size_t i = a; i < b;) {
      size_t j = i + 1;
      while (j < b && sv[j].tile0 == sv[j - 1].tile0 + sv[j - 1].ntiles) ++j;
      runs.push_back({i, j});
      i = j;
    }
    if (runs.size() > 8 && !s->sorted_fallback_said) {
      // (this used to be silent) ineligible tiles -- wave / workgroup bins, categorical rows --
      // interleave with the eligible ones: the launch falls back to the tile sweep, said once per sampler
      s->sorted_fallback_said = true;
      fprintf(stderr, "dwx: a launch's weight-sorted super-tiles form %zu gap-free runs (more than 8): its tiles take "
                      "the plain tile sweep (slower weight gathers)\n", runs.size());
    }
    if (!runs.empty() && runs.size() <= 8) {
      used_sorted = true;
      uint32_t cursor = t0;
      for (const Run &r : runs) {
        launch_lane_tiles(cursor, sv[r.a].tile0);
        // (one distinct d: the UNI build keeps it in a scalar register; DWX_NO_SORT_UNI: A/B knob)
        if (s->n_sort_dvals == 2 && !s->no_sort_uni)
          rt::launch(sorted_sweep_kernel<LEARN, true>, (unsigned)(r.b - r.a), SORT_THREADS, s->lds_sorted, s->stream, P,
                     (const SuperTile *)(d_sv + r.a), (uint32_t)(r.b - r.a), d_sr,
                     (const double *)s->d_sort_dvals, s->n_sort_dvals);
        else
          rt::launch(sorted_sweep_kernel<LEARN, false>, (unsigned)(r.b - r.a), SORT_THREADS, s->lds_sorted, s->stream, P,
                     (const SuperTile *)(d_sv + r.a), (uint32_t)(r.b - r.a), d_sr,
                     (const double *)s->d_sort_dvals, s->n_sort_dvals);
        ++launches;
        cursor = sv[r.b - 1].tile0 + sv[r.b - 1].ntiles;
      }
      launch_lane_tiles(cursor, t1);
    }
  }
  if (!used_sorted) launch_lane_tiles(t0, t1);
  P.tile_begin = t0;
  P.tile_end = t1;
  // categorical oversized variables: a workgroup each
  if (cg1 > cg0) {
    rt::launch(giant_kernel<LEARN>, cg1 - cg0, GIANT_THREADS, 0, st_giant, P,
               (const uint32_t *)(s->d_giant + cg0), cg1 - cg0);
    ++launches;
  }
  // boolean: the pieces' partial potentials, the decisions, (learning) the pieces' gradients
  if (bg1 > bg0) {
    const uint32_t p0 = s->bgiant_piece_off[bg0], np = s->bgiant_piece_off[bg1] - p0;
    rt::launch(giant_pot_kernel<LEARN>, np, GIANT_THREADS, 0, st_giant, P, (const uint32_t *)s->d_bgiant,
               (const GiantPiece *)s->d_bgiant_pieces, p0, np, s->d_bgiant_partial);
    rt::launch(giant_decide_kernel<LEARN>, (bg1 - bg0 + BLOCK_THREADS - 1) / BLOCK_THREADS, BLOCK_THREADS, 0, st_giant, P,
               (const uint32_t *)s->d_bgiant, (const uint32_t *)s->d_bgiant_piece_off, bg0, bg1 - bg0,
               (const double *)s->d_bgiant_partial, s->d_bgiant_decision);
    launches += 2;
    if (LEARN) {
      rt::launch(giant_grad_kernel, np, GIANT_THREADS, 0, st_giant, P, (const uint32_t *)s->d_bgiant,
                 (const GiantPiece *)s->d_bgiant_pieces, p0, np, (const uint32_t *)s->d_bgiant_decision);
      ++launches;
    }
  }
  // mid-degree variables among these tiles: a wave each
  if (w1 > w0) {
    const uint32_t per_block = BLOCK_THREADS / 64u;
    rt::launch(wide_kernel<LEARN>, (w1 - w0 + per_block - 1) / per_block, BLOCK_THREADS, 0, st_wide, P,
               (const uint32_t *)(s->d_wide + w0), w1 - w0);
    ++launches;
  }
  // join: whatever follows on `stream` (the next colour, the pull gradient, an update) waits for all three
  if (fork) {
    if (any_giant) { rt::event_record(s->ev_join[0], st_giant); rt::stream_wait_event(s->stream, s->ev_join[0]); }
    if (any_wide) { rt::event_record(s->ev_join[1], st_wide); rt::stream_wait_event(s->stream, s->ev_join[1]); }
  }
  return launches;
}

// n inference sweeps of a graph whose sweeps can share a launch: compact records (all-unary, so
// no variable reads another and the potentials of a variable are the same in every sweep while
// the weights rest) and no degree-binned variable (wide / giant tiles run kernels of their own).
// ONE launch of the gathering tile sweep stages every tile once and draws n times per variable
// (sweep8_kernel<..., MULTI>, infer_variable_multi): neither the terms table nor the weight-sorted
// copy is needed -- a record is read once per n sweeps.
bool multi_sweep_graph(const dwx_sampler *s) {
  return s->rec8 && s->cgiant_tiles.empty() && s->bgiant_tiles.empty() && s->cg->wide_tiles.empty() &&
         !getenv("DWX_NO_MULTI_SWEEP");
}

void enqueue_inference_multi(dwx_sampler *s, uint32_t n) {
  const CompiledGraph &c = *s->cg;
  rt::set_device(s->device);
  KernelParams P = s->base;
  P.sweep = s->sweep;
  P.n_sweeps = n;
  P.edge_terms = nullptr;
  TimedSpan sp{};
  if (s->timing) {
    sp.a = rt::event_create(); sp.b = rt::event_create(); sp.c = rt::event_create(); sp.kind = 0;
    rt::event_record(sp.a, s->stream);
  }
  uint32_t launches = 0;
  for (size_t l = 0; l + 1 < c.launch_off.size(); ++l) {
    const uint32_t t1 = s->opts.sample_evidence ? c.launch_tile[l + 1] : c.launch_query_tile_end[l];
    launches += launch_tiles<false>(s, P, l, c.launch_tile[l], t1, true);
  }
  if (s->timing) {
    rt::event_record(sp.b, s->stream);
    rt::event_record(sp.c, s->stream);
    sp.launches = launches; sp.has_pull = false; sp.new_sweep = true; sp.n_sweeps = n;
    s->spans.push_back(sp);
  }
  s->sweep += n;
}

// one inference sweep (GibbsSampler::sample)
void enqueue_inference(dwx_sampler *s) {
  const CompiledGraph &c = *s->cg;
  rt::set_device(s->device);
  KernelParams P = s->base;
  P.sweep = s->sweep;
  // The second consecutive inference sweep on the same weights tabulates the pre-signed
  // records' potential terms (one extra pass, about the cost of a sweep); from then on
  // sweeps stream the table and gather no weights.  (Not on the first one: learning and
  // inference sweeps may alternate, and then the table would be rebuilt for a single use.)
  if (s->has_simple_tiles && s->terms_state == 1 && c.NIdx) {
    const unsigned grid = std::min<unsigned>((unsigned)c.tiles.size(), 256u * 16u);
    if (s->rec8) {   // all-unary graph: the 8-byte table
      if (!s->d_terms) s->d_terms = (EdgeTerms *)rt::dmalloc(c.NIdx * 8);
      rt::launch(build_terms8_kernel, grid, BLOCK_THREADS, 0, s->stream, (const EdgeRec8 *)s->d_edges8,
                 (uint64_t)c.NIdx, (const float *)s->d_w32, (unsigned long long *)s->d_terms);
    } else {
      if (!s->d_terms) s->d_terms = (EdgeTerms *)rt::dmalloc(c.NIdx * sizeof(EdgeTerms));
      rt::launch(build_terms_kernel, grid, BLOCK_THREADS, 0, s->stream, (const TileDesc *)s->d_tiles,
                 (uint32_t)c.tiles.size(), (const EdgeRec *)s->d_edges, (const float *)s->d_w32, s->d_terms);
    }
    s->terms_state = 2;
  }
  P.edge_terms = s->terms_state == 2 ? s->d_terms : nullptr;
  if (s->terms_state == 0) s->terms_state = 1;
  TimedSpan sp{};
  if (s->timing) {
    sp.a = rt::event_create(); sp.b = rt::event_create(); sp.c = rt::event_create(); sp.kind = 0;
    rt::event_record(sp.a, s->stream);
  }
  uint32_t launches = 0;
  for (size_t l = 0; l + 1 < c.launch_off.size(); ++l) {
    // only the query variables' tiles unless --sample_evidence (src/gibbs_sampler.h:157)
    const uint32_t t1 = s->opts.sample_evidence ? c.launch_tile[l + 1] : c.launch_query_tile_end[l];
    launches += launch_tiles<false>(s, P, l, c.launch_tile[l], t1);
  }
  if (s->timing) {
    rt::event_record(sp.b, s->stream);
    rt::event_record(sp.c, s->stream);
    sp.launches = launches; sp.has_pull = false; sp.new_sweep = true;
    s->spans.push_back(sp);
  }
  ++s->sweep;
}

// Tile boundaries of colour launch l cut into at most nb runs of (nearly) equal SGD work
// (records visited by sgd_on_variable; query variables sample but do not learn, so a cut
// by tile count would put all the learning of an evidence-after-query launch in half of
// the batches).  Returns strictly increasing boundaries from the launch's first tile to
// its last.
std::vector<uint32_t> cut_launch(const dwx_sampler *s, size_t l, uint32_t nb) {
  const CompiledGraph &c = *s->cg;
  const uint32_t t0 = c.launch_tile[l], t1 = c.launch_tile[l + 1];
  std::vector<uint32_t> cut{t0};
  if (t1 == t0) return cut;
  const uint64_t w0 = s->sgd_work[t0], total = s->sgd_work[t1] - w0;
  // nb pieces for the colour with the MOST work; a colour with less gets as many as keep its
  // pieces no bigger than that colour's (a greedy colouring leaves most of the work in the first
  // colours: cutting the small late colours nb-fold too only multiplies dependent launches --
  // 20 colours x 64 batches x 6 kernels on the power-law test graph)
  if (nb > 1 && s->sgd_work_max_launch > 0)
    nb = (uint32_t)std::max<uint64_t>(1, (total * nb + s->sgd_work_max_launch - 1) / s->sgd_work_max_launch);
  nb = std::max(1u, std::min(nb, t1 - t0));
  for (uint32_t b = 1; b < nb; ++b) {
    uint32_t t;
    if (total == 0) {
      t = t0 + (uint32_t)((uint64_t)(t1 - t0) * b / nb);
    } else {
      const uint64_t target = w0 + (total / nb) * b + (total % nb) * b / nb;
      t = (uint32_t)(std::lower_bound(s->sgd_work.begin() + t0, s->sgd_work.begin() + t1, target) -
                     s->sgd_work.begin());
    }
    if (t > cut.back() && t < t1) cut.push_back(t);
  }
  cut.push_back(t1);
  return cut;
}

bool triggers_sgd(const dwx_options &o, uint32_t meta) {
  return o.learn_non_evidence || (!o.noise_aware && (meta & VM_EVIDENCE)) ||
         (o.noise_aware && (meta & VM_TRUTHINESS));
}

// How far one record can move its owner's potential between two of the owner's values, the
// building block of every curvature bound below: |sign(hit) - sign(miss)| * |f| for a unary
// factor (src/factor.h:112-299 at arity 1; boolean owner: hit = "1 satisfies the predicate",
// miss = "0 does"; categorical owner: the record sits in the row of the value that satisfies
// it), 2 |f| (arity - 1) for wider ones (a sign spans [-1, 1]; LINEAR counts up to arity - 1).
// Fixed weights do not move: 0.  The oracle restates this (orc_sched_accumulate).
double host_unary_sign(uint32_t func, bool sat) {
  switch (func) {
    case FUNC_AND: case FUNC_ISTRUE: case FUNC_OR: case FUNC_IMPLY_NATURAL: return sat ? 1.0 : -1.0;
    case FUNC_EQUAL: return 1.0;
    default: return sat ? 1.0 : 0.0;
  }
}
double record_delta(const CompiledGraph &c, uint32_t e, bool owner_is_cat) {
  const EdgeRec &r = c.edges[e];
  if (r.packed & EDGE_FIXED_FLAG) return 0.0;
  if (r.packed & EDGE_PRESIGNED) {
    float miss;
    std::memcpy(&miss, &r.aux, 4);
    return std::fabs((double)r.fval - (double)miss);
  }
  const double f = (r.packed & EDGE_F64_FLAG) ? c.edge_fval64[e] : (double)r.fval;
  const uint32_t ar = (r.packed & EDGE_INLINE2) ? 2u : ((r.packed >> EDGE_ARITY_SHIFT) & EDGE_ARITY_MASK);
  if (ar <= 1u) {
    const uint32_t fn = r.packed & EDGE_FUNC_MASK;
    const double hit = host_unary_sign(fn, owner_is_cat || r.aux == 1u);
    const double miss = host_unary_sign(fn, !owner_is_cat && r.aux == 0u);
    return std::fabs(hit - miss) * std::fabs(f);
  }
  return 2.0 * std::fabs(f) * (double)(ar - 1u);
}

// Gershgorin bound of one variable's share of the batch curvature, per record: the variable's
// SGD gradient over the weights has Jacobian kappa * d d^T for a boolean variable (logistic:
// kappa = p (1 - p) <= 1/4, d = the records' deltas) and J^T (diag(p) - p p^T) J for a
// categorical one (softmax over its value rows; absolute row sums of diag(p) - p p^T are
// 2 p (1 - p) <= 1/2).  Row sum of |H| for weight w: sum over w's records r of
//   1/4 * d_r * (sum of the row's deltas)              boolean
//   1/2 * d_r * (largest sum of deltas of a value row) categorical.
// fn(e, bound) is called for every record with a non-zero bound.
template <class Fn>
void for_each_record_bound(const CompiledGraph &c, uint32_t p, Fn &&fn) {
  const uint32_t m = c.v_meta[p];
  const bool cat = m & VM_CATEGORICAL;
  double S = 0.0;
  for (uint32_t r = c.v_row[p]; r < c.v_row[p + 1]; ++r) {
    double sr = 0.0;
    for (uint32_t e = c.row_ptr[r]; e < c.row_ptr[r + 1]; ++e) sr += record_delta(c, e, cat);
    S = cat ? std::max(S, sr) : S + sr;
  }
  if (S == 0.0) return;
  const double kappa = cat ? 0.5 : 0.25;
  for (uint32_t e = c.row_ptr[c.v_row[p]]; e < c.row_ptr[c.v_row[p + 1]]; ++e) {
    const double d = record_delta(c, e, cat);
    if (d != 0.0) fn(e, kappa * d * S);
  }
}

// A plan level's own weight-sorted layout (Level::sorted_supers): for a split plan cut ALONG its
// chunks, as many super-tiles per chunk as workgroups are resident; for an un-split sweep, where a
// launch holds query AND evidence tiles, one run per launch -- the default layout cuts those apart
// (an inference sweep launches over the query part only), so a learning launch over it ends two
// runs with a round of small super-tiles each (config 3: learning sweep 0.288 -> 0.269 ms with one).
// Worth it while a chunk still makes super-tiles of 8 tiles and more.  Costs 0.3 s of host time
// and a second copy of the records: dwx_options.plan_layouts says when it is built.
void ensure_level_layout(dwx_sampler *s, dwx_sampler::Level *L, uint32_t batches) {
  if (L->layout_done) return;
  L->layout_done = true;
  const CompiledGraph &c = *s->cg;
  bool mixed_launch = false;
  for (size_t l = 0; l + 1 < c.launch_tile.size(); ++l)
    mixed_launch = mixed_launch || (c.launch_query_tile_end[l] > c.launch_tile[l] && c.launch_query_tile_end[l] < c.launch_tile[l + 1]);
  if (!(batches > 1 || mixed_launch) || c.supers.empty() || !s->sorted_learn || L->chunks.empty()) return;
  uint64_t tiles_in_chunks = 0;
  std::vector<std::pair<uint32_t, uint32_t>> ranges;
  for (const auto &ch : L->chunks) { ranges.push_back({ch.t0, ch.t1}); tiles_in_chunks += ch.t1 - ch.t0; }
  if (tiles_in_chunks / L->chunks.size() < 8ull * c.sorted_slots) return;
  rt::set_device(s->device);
  SortedLayout lay;
  const bool on_device = devb::available() && !getenv("DWX_HOST_BUILD");
  build_sorted_layout(c, ranges, c.sorted_per_super, c.sorted_slots, batches == 1, host_threads(), lay, on_device);
  if (lay.supers.empty()) return;
  SortRec8 *d_sorted = nullptr;
  SuperTile *d_supers = nullptr;
  if (on_device) {
    // (the records sit at their super-tiles' planned offsets whatever the order of the descriptors)
    d_supers = upload(lay.supers, s->stream);
    d_sorted = (SortRec8 *)rt::dmalloc((lay.n + 1) * sizeof(SortRec8));
    rt::dmemset(d_sorted + lay.n, 0, sizeof(SortRec8), s->stream);
    devb::build_sorted_records(s->d_tiles, s->d_edges, s->d_edges8, d_supers, (uint32_t)lay.supers.size(),
                               c.sort_dbits.data(), (uint32_t)c.sort_dbits.size(), lay.n, d_sorted, (void *)s->stream);
    rt::dfree(d_supers);
  }
  // (launch_tiles looks super-tiles up by their first tile: ascending, like the default layout)
  std::sort(lay.supers.begin(), lay.supers.end(), [](const SuperTile &a, const SuperTile &b) { return a.tile0 < b.tile0; });
  if (!on_device) d_sorted = upload(lay.recs, s->stream, 1);
  d_supers = upload(lay.supers, s->stream);
  rt::stream_sync(s->stream);   // (the host copies die with this scope; the sweeps queued so far used the default layout)
  L->sorted_supers.swap(lay.supers);
  L->d_sorted = d_sorted;
  L->d_supers = d_supers;
}

// Everything a plan level needs to run its chunks without per-record atomics: per chunk,
// (a) the static update counts of its boolean variables -- a boolean variable that triggers
// SGD visits every factor of its row once with t = 1 (src/factor_graph.cc:265-273),
// independent of the samples drawn -- and (b) the pull-gradient incidence list of its
// TILE_PULL tiles: every (triggering variable, non-fixed record) pair, sorted by weight
// (two stable parallel counting sorts: by weight, then by chunk).
dwx_sampler::Level *build_level(dwx_sampler *s, uint32_t batches) {
  auto it = s->levels.find(batches);
  if (it != s->levels.end()) return it->second.get();
  rt::set_device(s->device);
  const CompiledGraph &c = *s->cg;
  const dwx_options &o = s->opts;
  const uint32_t nth = host_threads();
  std::unique_ptr<dwx_sampler::Level> L(new dwx_sampler::Level());
  // The pieces of a split sweep are visited ROUND-ROBIN over the colour launches, piece b of n
  // at position (b + 1/2) / n of the sweep: any order of independent sets is a Gibbs scan, and
  // the reference's scan in id order meets the colours interleaved, not one after the other --
  // with heavily tied weights the weights at the end of a sweep are those the LAST visits want,
  // and a colour can be a biased sample of the model (the observed middle variables of a chain
  // a - b - c: tests/golden tied_chain; the reference itself run on the colour-major relabelling
  // of that graph learns 0.87 .. 1.0 in its first epoch against 0.34 .. 0.52 in id order).
  {
    struct Piece { double at; dwx_sampler::Chunk ch; };
    std::vector<Piece> pieces;
    for (size_t l = 0; l + 1 < c.launch_off.size(); ++l) {
      const std::vector<uint32_t> cut = cut_launch(s, l, batches);
      const size_t n = cut.size() - 1;
      for (size_t b = 0; b < n; ++b)
        pieces.push_back({((double)b + 0.5) / (double)n, {(uint32_t)l, cut[b], cut[b + 1]}});
    }
    std::stable_sort(pieces.begin(), pieces.end(), [](const Piece &a, const Piece &b) { return a.at < b.at; });
    for (const Piece &pc : pieces) L->chunks.push_back(pc.ch);
  }
  // (plan_layouts 0, the default: with the level wherever the records are sorted on the device -- it
  // costs milliseconds there; a host build pays 0.3 s per level and waits for 2048 sweeps)
  // DWX_TIMING=1: wall time of the level's build steps on stderr (each ends on a stream sync)
  const bool lv_timing = getenv("DWX_TIMING") != nullptr;
  auto lv_t0 = std::chrono::steady_clock::now();
  auto lv_step = [&](const char *what) {
    if (!lv_timing) return;
    rt::stream_sync(s->stream);
    const auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[dwx level B=%u] %-28s %.3f s\n", batches, what, std::chrono::duration<double>(t - lv_t0).count());
    lv_t0 = t;
  };
  lv_step("chunks");
  if (s->opts.plan_layouts == 1 || (s->opts.plan_layouts == 0 && devb::available() && !getenv("DWX_HOST_BUILD")))
    ensure_level_layout(s, L.get(), batches);
  lv_step("weight-sorted layout");
  const uint32_t nc = (uint32_t)L->chunks.size();
  // beyond this the per-chunk tables cost more than they save: such plans keep the
  // per-record atomics and dynamic counts
  uint32_t table_chunks = 4 * MAX_PLAN_BATCHES;
  if (const char *e = getenv("DWX_PLAN_TABLE_CHUNKS")) table_chunks = (uint32_t)std::max(1L, atol(e));   // test hook
  L->fast = nc >= 1 && nc <= table_chunks && (uint64_t)nc * c.W * 16 <= ((uint64_t)2 << 30);
  if (batches == 1) L->fast = true;
  if (L->fast && c.W) {
    // groups of the tables: the chunks of a split sweep (an update follows each of them); ONE
    // group for an un-split sweep, whatever its number of colour launches (one update at the end)
    struct Group { uint32_t t0, t1; };
    std::vector<Group> groups;
    if (batches == 1) groups.push_back({0u, (uint32_t)c.tiles.size()});
    else for (const auto &ch : L->chunks) groups.push_back({ch.t0, ch.t1});
    const uint32_t nc = (uint32_t)groups.size();
    std::vector<uint32_t> chunk_of(c.tiles.size(), 0);
    for (uint32_t k = 0; k < nc; ++k)
      for (uint32_t t = groups[k].t0; t < groups[k].t1; ++t) chunk_of[t] = k;
    // (a) static tables, one row pair per group (rows are disjoint: groups in parallel): the
    // update counts T of boolean variables and the curvature bounds h of all variables
    // (apply_kernel's saturating step); row k = [T[W] | h[W]]
    L->rows = nc;
    const bool tables_on_device = devb::available() && !getenv("DWX_HOST_BUILD");
    if (tables_on_device) {
      // (a) on the device: a lane per variable over the uploaded records, integer atomics (device_build.hip)
      std::vector<uint32_t> group_of(c.tiles.size(), 0xFFFFFFFFu);
      for (uint32_t k = 0; k < nc; ++k)
        for (uint32_t t = groups[k].t0; t < groups[k].t1; ++t) group_of[t] = k;
      L->d_t_static = (long long *)rt::dmalloc((size_t)nc * 2 * c.W * 8);
      long long tm = 0, hm = 0;
      devb::build_static_tables(s->d_tiles, (uint32_t)c.tiles.size(), group_of.data(), s->d_v_meta, s->d_v_row, s->d_row_ptr,
                                s->d_edges, s->d_edge_fval64, (const uint8_t *)s->d_w_fixed, (uint32_t)c.W, nc,
                                o.learn_non_evidence != 0, o.noise_aware != 0, L->d_t_static, &tm, &hm, (void *)s->stream);
      L->c_max = (double)hm / H_SCALE; L->t_max = (double)tm / FIX_SCALE;
      lv_step("static tables (device)");
    } else {
    RawArray<long long> ts((size_t)nc * 2 * c.W);
    parallel_ranges((uint64_t)nc * 2 * c.W, nth, [&](uint64_t b, uint64_t e) { std::fill(ts.data() + b, ts.data() + e, 0LL); });
    const long long one = (long long)FIX_SCALE;
    // Few groups (the un-split level has ONE: a single thread used to walk all of config 5's 10^9
    // records here): the variables of a group are shared out over the threads instead, which add
    // into the group's rows atomically -- integer sums, the same tables in any order.
    auto group_rows = [&](uint64_t k, uint32_t pb, uint32_t pe, auto &&add) {
      long long *row = ts.data() + k * 2 * c.W, *hrow = row + c.W;
      for (uint32_t p = pb; p < pe; ++p) {
        const uint32_t m = c.v_meta[p];
        if (!triggers_sgd(o, m)) continue;
        for_each_record_bound(c, p, [&](uint32_t e, double bound) {
          add(&hrow[c.edges[e].wid], (long long)std::llrint(H_SCALE * bound));
        });
        if (m & VM_CATEGORICAL) continue;   // (their counts depend on the samples: dynamic)
        for (uint32_t e = c.row_ptr[c.v_row[p]]; e < c.row_ptr[c.v_row[p] + 1]; ++e)
          if (!c.w_fixed[c.edges[e].wid]) add(&row[c.edges[e].wid], one);
      }
    };
    if (nc >= nth || nth <= 1) {
      parallel_ranges(nc, std::min(nth, nc), [&](uint64_t kb, uint64_t ke) {
        for (uint64_t k = kb; k < ke; ++k)
          group_rows(k, c.tile_v[groups[k].t0], c.tile_v[groups[k].t1], [](long long *a, long long v) { *a += v; });
      }, 2);
    } else {
      for (uint64_t k = 0; k < nc; ++k) {
        const uint32_t p0 = c.tile_v[groups[k].t0], p1 = c.tile_v[groups[k].t1];
        parallel_ranges(p1 - p0, nth, [&](uint64_t b, uint64_t e) {
          group_rows(k, p0 + (uint32_t)b, p0 + (uint32_t)e,
                     [](long long *a, long long v) { __atomic_fetch_add(a, v, __ATOMIC_RELAXED); });
        });
      }
    }
    {
      const uint32_t T = std::max(1u, std::min(nth, 16u));
      std::vector<double> hm(T, 0.0), tm(T, 0.0);
      parallel_parts((uint64_t)nc * c.W, T, [&](uint32_t t, uint64_t b, uint64_t e) {
        long long h = 0, tt = 0;
        for (uint64_t i = b; i < e; ++i) {
          const uint64_t k = i / c.W, w = i % c.W;
          tt = std::max(tt, ts[k * 2 * c.W + w]);
          h = std::max(h, ts[(k * 2 + 1) * c.W + w]);
        }
        hm[t] = (double)h / H_SCALE; tm[t] = (double)tt / FIX_SCALE;
      }, 0);
      for (uint32_t t = 0; t < T; ++t) { L->c_max = std::max(L->c_max, hm[t]); L->t_max = std::max(L->t_max, tm[t]); }
    }
    L->d_t_static = upload_raw(ts.data(), (size_t)nc * 2 * c.W, s->stream);
    rt::stream_sync(s->stream);      // (ts dies with this scope)
    }
    // (b) incidence list
    if (tables_on_device) {
      // on the device (device_build.hip): ordered emit per tile, one stable radix sort by (chunk, weight),
      // the block tables filled by a lane per weight -- the structures below, bit for bit
      uint64_t min_w = 131072, bp_tiles = BP_TILES;
      if (const char *e = getenv("DWX_BLOCK_PULL_MIN_W")) min_w = (uint64_t)std::max(0L, atol(e));          // test hooks
      if (const char *e = getenv("DWX_BLOCK_PULL_TILES")) bp_tiles = (uint64_t)std::min<long>(BP_TILES, std::max(1L, atol(e)));
      std::vector<uint32_t> tile_info(c.tiles.size(), 0xFFFFFFFFu);
      for (size_t ti = 0; ti < c.tiles.size(); ++ti) {
        const TileDesc &td = c.tiles[ti];
        const bool unary_only = !(td.flags & TILE_PULL) && (td.flags & TILE_PULL_UNARY) && c.ecap <= 6 * BLOCK_THREADS;
        if ((!(td.flags & TILE_PULL) && !unary_only) || (td.flags & TILE_OUTSIDE)) continue;
        tile_info[ti] = chunk_of[ti] | ((unary_only ? 2u : 1u) << 30);
      }
      devb::Incidence inc;
      devb::build_incidence(s->d_tiles, c.tiles.data(), (uint32_t)c.tiles.size(), tile_info.data(), s->d_v_meta, s->d_v_row,
                            s->d_row_ptr, s->d_edges, o.learn_non_evidence != 0, o.noise_aware != 0, (uint32_t)c.W, nc, min_w,
                            (uint32_t)bp_tiles, inc, (void *)s->stream);
      L->d_inc_wid = inc.d_inc_wid; L->d_inc_slot = inc.d_inc_slot; L->d_inc_d = inc.d_inc_d;
      L->inc_begin = inc.inc_begin; L->inc_end = inc.inc_end;
      lv_step("incidence list (device)");
      if (!inc.bp.empty()) {
        const bool bp_timing = getenv("DWX_TIMING") != nullptr;
        L->bp.resize(nc);
        for (uint32_t k = 0; k < nc; ++k) {
          if (!inc.bp[k].blocks) continue;
          dwx_sampler::Level::BlockTable &bt = L->bp[k];
          bt.d_ell = inc.bp[k].d_ell;
          bt.d_tile0 = upload(inc.bp[k].tile0, s->stream);
          bt.blocks = inc.bp[k].blocks; bt.depth = inc.bp[k].depth;
          bt.parts = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(rt::cu_count() / bt.blocks, inc.wp / BP_THREADS));
          if (bp_timing)
            fprintf(stderr, "[dwx block pull] group %u: %u blocks x depth %u, %zu deltas, %llu of %llu entries on the list (device build)\n",
                    k, bt.blocks, bt.depth, inc.dvals.size(), (unsigned long long)inc.bp[k].on_list, (unsigned long long)inc.bp[k].total);
        }
        std::vector<long long> qtab(inc.dvals.size());
        for (size_t i = 0; i < inc.dvals.size(); ++i) {
          float d; std::memcpy(&d, &inc.dvals[i], 4);
          qtab[i] = std::llrint(FIX_SCALE * (double)d);
        }
        L->d_bp_qtab = upload(qtab, s->stream);
        L->d_bp_partial = (long long *)rt::dmalloc(inc.max_blocks * inc.wp * 8);
        L->bp_deltas = (uint32_t)inc.dvals.size(); L->bp_wp = (uint32_t)inc.wp;
        rt::allow_dynamic_lds(pull_ell_kernel<1, false>, BP_LDS_BYTES);
        rt::allow_dynamic_lds(pull_ell_kernel<2, false>, BP_LDS_BYTES);
        rt::allow_dynamic_lds(pull_ell_kernel<1, true>, BP_LDS_BYTES);
        rt::allow_dynamic_lds(pull_ell_kernel<2, true>, BP_LDS_BYTES);
      }
      rt::stream_sync(s->stream);
    } else {
    struct Inc { uint32_t wid, slot; float d; uint32_t chunk; };
    RawArray<Inc> by_w, by_c;
    std::vector<uint64_t> w_start, c_start;
    parallel_group_by_key<Inc>(
        c.tiles.size(), nth, c.W, [](const Inc &r) { return (uint64_t)r.wid; },
        [&](uint64_t tb, uint64_t te, auto &&emit) {
          for (uint64_t ti = tb; ti < te; ++ti) {
            const TileDesc &td = c.tiles[ti];
            // (TILE_PULL_UNARY: only the tile's pre-signed records; not in a K = 12 build, whose
            // kernels walk such tiles generically)
            const bool unary_only = !(td.flags & TILE_PULL) && (td.flags & TILE_PULL_UNARY) && c.ecap <= 6 * BLOCK_THREADS;
            if (!(td.flags & TILE_PULL) && !unary_only) continue;
            if (td.flags & TILE_OUTSIDE) continue;   // giant_kernel / wide_kernel: atomics
            for (uint32_t l = 0; l < td.nv; ++l) {
              const uint32_t p = td.v0 + l, m = c.v_meta[p];
              if (!(o.learn_non_evidence || (!o.noise_aware && (m & VM_EVIDENCE)))) continue;
              for (uint32_t e = c.row_ptr[c.v_row[p]]; e < c.row_ptr[c.v_row[p] + 1]; ++e) {
                const EdgeRec &r = c.edges[e];
                if (r.packed & EDGE_FIXED_FLAG) continue;
                if (unary_only && !(r.packed & EDGE_PRESIGNED)) continue;
                float miss;
                std::memcpy(&miss, &r.aux, 4);
                const float dd = r.fval - miss;    // exact: |hit| == |miss| or one of them is 0
                if (dd == 0.0f) continue;
                emit(Inc{r.wid, (uint32_t)(ti * BLOCK_THREADS + l), dd, chunk_of[ti]});
              }
            }
          }
        },
        by_w, w_start, 64);
    const uint64_t n_inc = by_w.size();
    if (n_inc + (uint64_t)nc * PULL_RUN >= 0xFFFFFFFFull)
      throw std::invalid_argument("incidence list exceeds 2^32-1 entries");
    // the groups' lists: by weight inside a group
    const RawArray<Inc> *src = &by_w;
    if (nc > 1) {
      parallel_group_by_key<Inc>(
          n_inc, nth, nc, [](const Inc &r) { return (uint64_t)r.chunk; },
          [&](uint64_t b, uint64_t e, auto &&emit) { for (uint64_t i = b; i < e; ++i) emit(by_w[i]); },
          by_c, c_start);
      by_w.clear();
      src = &by_c;
    } else {
      c_start = {0, n_inc};
    }
    // Block pull (graphs with many weights): per group, rows of BP_ROW * depth entries per
    // (variable block, weight) for pull_ell_kernel; an entry beyond its row stays on the list.
    RawArray<Inc> kept;                    // the lists of all groups after the tables took their entries
    std::vector<uint64_t> kept_start;
    {
      uint64_t min_w = 131072, bp_tiles = BP_TILES;   // (below: the list pull is as fast, tools/tied_sweep.py)
      if (const char *e = getenv("DWX_BLOCK_PULL_MIN_W")) min_w = (uint64_t)std::max(0L, atol(e));          // test hooks
      if (const char *e = getenv("DWX_BLOCK_PULL_TILES")) bp_tiles = (uint64_t)std::min<long>(BP_TILES, std::max(1L, atol(e)));
      const bool bp_timing = getenv("DWX_TIMING") != nullptr;
      // the distinct record deltas (few: feature values repeat), as fixed-point steps
      std::vector<uint32_t> dvals;
      bool enabled = n_inc && c.W >= min_w;
      if (enabled) {
        const uint32_t T = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(nth, n_inc / 65536));
        std::vector<std::vector<uint32_t>> local(T);
        parallel_parts(n_inc, T, [&](uint32_t t, uint64_t b, uint64_t e) {
          std::vector<uint32_t> &v = local[t];
          uint32_t last = 0; bool have = false;
          for (uint64_t i = b; i < e && v.size() <= 4 * BP_MAX_DELTAS; ++i) {
            uint32_t bits; std::memcpy(&bits, &(*src)[i].d, 4);
            if (have && bits == last) continue;
            last = bits; have = true;
            v.push_back(bits);
            if (v.size() % 4096 == 0) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); }
          }
          std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end());
        }, 0);
        for (auto &v : local) dvals.insert(dvals.end(), v.begin(), v.end());
        std::sort(dvals.begin(), dvals.end()); dvals.erase(std::unique(dvals.begin(), dvals.end()), dvals.end());
        enabled = dvals.size() <= BP_MAX_DELTAS;
      }
      if (enabled) {
        const uint64_t Wp = (c.W + BP_THREADS - 1) / BP_THREADS * BP_THREADS;
        L->bp.resize(nc);
        uint64_t max_blocks = 0;
        std::vector<uint64_t> ov_count(nc, 0);
        std::vector<std::vector<uint64_t>> ov_start(nc);   // per group, per weight: entries left on the list
        std::vector<RawArray<Inc>> ov(nc);
        std::vector<uint8_t> has(c.tiles.size(), 0);
        std::vector<uint32_t> block_of(c.tiles.size(), 0);
        for (uint32_t k = 0; k < nc; ++k) {
          const Inc *ent = src->data() + c_start[k];
          const uint64_t n = c_start[k + 1] - c_start[k];
          if (!n) continue;
          // (1) blocks: runs of <= bp_tiles consecutive tiles, started at tiles that own entries
          std::fill(has.begin(), has.end(), 0);
          parallel_ranges(n, nth, [&](uint64_t b, uint64_t e) {
            for (uint64_t i = b; i < e; ++i) {   // (test first: the flags are shared by all threads)
              uint8_t &h = has[ent[i].slot / BLOCK_THREADS];
              if (!h) h = 1;
            }
          });
          std::vector<uint32_t> tile0;
          for (uint32_t ti = 0; ti < c.tiles.size(); ++ti) {
            if (!has[ti]) continue;
            if (tile0.empty() || ti >= tile0.back() + bp_tiles) tile0.push_back(ti);
            block_of[ti] = (uint32_t)tile0.size() - 1;
          }
          const uint64_t nvb = tile0.size();
          const double lambda = (double)n / ((double)c.W * (double)nvb);
          if (lambda < 0.5) continue;   // nearly empty rows: this group keeps its list
          const uint32_t depth = lambda > 3.2 ? 2u : 1u, cap = BP_ROW * depth;
          // (2) where each weight's entries start inside the group (they are sorted by weight)
          std::vector<uint64_t> w_at(c.W + 1, n);
          parallel_ranges(n, nth, [&](uint64_t b, uint64_t e) {
            for (uint64_t i = b; i < e; ++i) {
              const uint32_t w = ent[i].wid, prev = i ? ent[i - 1].wid + 1 : 0;
              if (!i || ent[i - 1].wid != w) for (uint32_t x = prev; x <= w; ++x) w_at[x] = i;
            }
          });
          // (3) the table; entries of one weight come in tile order: blocks ascending
          RawArray<U32x4> ell(nvb * depth * Wp);
          parallel_ranges(ell.size(), nth, [&](uint64_t b, uint64_t e) { std::memset((void *)(ell.data() + b), 0xFF, (e - b) * sizeof(U32x4)); });
          std::vector<uint64_t> &ovs = ov_start[k];
          ovs.assign(c.W + 1, 0);
          parallel_ranges(c.W, nth, [&](uint64_t wb, uint64_t we) {
            for (uint64_t w = wb; w < we; ++w) {
              uint32_t cur = ~0u, kk = 0;
              uint64_t o = 0;
              for (uint64_t i = w_at[w]; i < w_at[w + 1]; ++i) {
                const Inc &r = ent[i];
                const uint32_t ti = r.slot / BLOCK_THREADS, vb = block_of[ti];
                if (vb != cur) { cur = vb; kk = 0; }
                if (kk < cap) {
                  uint32_t bits; std::memcpy(&bits, &r.d, 4);
                  const uint32_t di = (uint32_t)(std::lower_bound(dvals.begin(), dvals.end(), bits) - dvals.begin());
                  ell[((uint64_t)vb * depth + kk / BP_ROW) * Wp + w].v[kk % BP_ROW] =
                      ((ti - tile0[vb]) * BLOCK_THREADS + r.slot % BLOCK_THREADS) | (di << BP_SLOT_BITS);
                } else {
                  ++o;
                }
                ++kk;
              }
              ovs[w + 1] = o;
            }
          });
          for (uint64_t w = 0; w < c.W; ++w) ovs[w + 1] += ovs[w];
          ov_count[k] = ovs[c.W];
          // the entries the rows could not take, in list order (same walk, now copying)
          ov[k].reset(ov_count[k]);
          if (ov_count[k]) {
            Inc *dst = ov[k].data();
            parallel_ranges(c.W, nth, [&](uint64_t wb, uint64_t we) {
              for (uint64_t w = wb; w < we; ++w) {
                uint32_t cur = ~0u, kk = 0;
                uint64_t o = ovs[w];
                for (uint64_t i = w_at[w]; i < w_at[w + 1]; ++i) {
                  const uint32_t vb = block_of[ent[i].slot / BLOCK_THREADS];
                  if (vb != cur) { cur = vb; kk = 0; }
                  if (kk >= cap) dst[o++] = ent[i];
                  ++kk;
                }
              }
            });
          }
          dwx_sampler::Level::BlockTable &bt = L->bp[k];
          bt.d_ell = upload_raw(ell.data(), ell.size(), s->stream);
          bt.d_tile0 = upload(tile0, s->stream);
          bt.blocks = (uint32_t)nvb; bt.depth = depth;
          // parts per block: one workgroup per CU over all blocks (it owns the CU's whole LDS)
          bt.parts = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(rt::cu_count() / nvb, Wp / BP_THREADS));
          max_blocks = std::max(max_blocks, nvb);
          rt::stream_sync(s->stream);   // ell dies with this scope
          if (bp_timing)
            fprintf(stderr, "[dwx block pull] group %u: %llu blocks x depth %u, %zu deltas, %llu of %llu entries on the list\n",
                    k, (unsigned long long)nvb, depth, dvals.size(), (unsigned long long)ov_count[k], (unsigned long long)n);
        }
        if (max_blocks) {
          // the lists shrink to what the tables left (same order: by weight inside a group)
          kept_start.assign(nc + 1, 0);
          for (uint32_t k = 0; k < nc; ++k)
            kept_start[k + 1] = kept_start[k] + (L->bp[k].blocks ? ov_count[k] : c_start[k + 1] - c_start[k]);
          kept.reset(kept_start[nc]);
          for (uint32_t k = 0; k < nc; ++k) {
            const Inc *ent = src->data() + c_start[k];
            const uint64_t n = c_start[k + 1] - c_start[k];
            Inc *dst = kept.data() + kept_start[k];
            if (!L->bp[k].blocks) std::copy(ent, ent + n, dst);
            else std::copy(ov[k].data(), ov[k].data() + ov[k].size(), dst);
          }
          std::vector<long long> qtab(dvals.size());
          for (size_t i = 0; i < dvals.size(); ++i) {
            float d; std::memcpy(&d, &dvals[i], 4);
            qtab[i] = std::llrint(FIX_SCALE * (double)d);
          }
          L->d_bp_qtab = upload(qtab, s->stream);
          L->d_bp_partial = (long long *)rt::dmalloc(max_blocks * Wp * 8);
          L->bp_deltas = (uint32_t)dvals.size(); L->bp_wp = (uint32_t)Wp;
          rt::allow_dynamic_lds(pull_ell_kernel<1, false>, BP_LDS_BYTES);
          rt::allow_dynamic_lds(pull_ell_kernel<2, false>, BP_LDS_BYTES);
          rt::allow_dynamic_lds(pull_ell_kernel<1, true>, BP_LDS_BYTES);
          rt::allow_dynamic_lds(pull_ell_kernel<2, true>, BP_LDS_BYTES);
          rt::stream_sync(s->stream);
          by_w.clear(); by_c.clear();
          src = &kept;
          c_start = kept_start;
        } else {
          L->bp.clear();
        }
      }
    }
    // columns; every chunk padded to whole runs with neutral entries (its last weight, zero
    // contribution), so the kernel's 16-byte loads never leave the chunk
    std::vector<uint64_t> off(nc + 1, 0);
    for (uint32_t k = 0; k < nc; ++k)
      off[k + 1] = off[k] + (c_start[k + 1] - c_start[k] + PULL_RUN - 1) / PULL_RUN * PULL_RUN;
    const uint64_t padded = off[nc];
    L->inc_begin.resize(nc); L->inc_end.resize(nc);
    if (padded) {
      RawArray<uint32_t> iw(padded), is(padded);
      RawArray<float> id(padded);
      for (uint32_t k = 0; k < nc; ++k) {
        L->inc_begin[k] = (uint32_t)off[k]; L->inc_end[k] = (uint32_t)off[k + 1];
        const uint64_t n = c_start[k + 1] - c_start[k], base = c_start[k];
        parallel_ranges(off[k + 1] - off[k], nth, [&](uint64_t b, uint64_t e) {
          for (uint64_t i = b; i < e; ++i) {
            const Inc &r = (*src)[base + std::min(i, n - 1)];
            iw[off[k] + i] = r.wid; is[off[k] + i] = r.slot; id[off[k] + i] = i < n ? r.d : 0.0f;
          }
        });
      }
      L->d_inc_wid = upload_raw(iw.data(), padded, s->stream);
      L->d_inc_slot = upload_raw(is.data(), padded, s->stream);
      L->d_inc_d = upload_raw(id.data(), padded, s->stream);
      rt::stream_sync(s->stream);   // the columns die with this scope
    }
    }
    rt::stream_sync(s->stream);
  }
  dwx_sampler::Level *out = L.get();
  s->levels[batches] = std::move(L);
  return out;
}

// Curvature of one SGD mini-batch in weight space, when every colour launch is cut into
// `batches` runs of tiles.  A variable v that triggers SGD contributes kappa_v d_v d_v^T to
// the batch Hessian bound, d_v[w] = sum of |d_r| over its non-fixed records with weight w,
// d_r = the change of the record's potential between proposals (|sign(hit) f - sign(miss) f|
// for pre-signed records, a 2|f|(arity-1) bound otherwise), kappa = 1/4 (logistic) for
// boolean, 1/2 for categorical variables.  The largest eigenvalue lambda of that matrix
// decides whether ONE batched step tracks the reference's sequential updates
// (stepsize * lambda < 2).  Estimated per chunk by two power iterations from the all-ones
// vector (the matrix is entrywise non-negative: the iteration converges from below to its
// Perron value) combined with the largest diagonal entry (exact for a batch dominated by
// one heavily tied weight), times a safety factor; the result is the max over chunks.
double row_sum_bound(dwx_sampler *s, uint32_t batches) {
  auto it = s->row_sum_cache.find(batches);
  if (it != s->row_sum_cache.end()) return it->second;
  const CompiledGraph &c = *s->cg;
  std::vector<double> x(c.W, 0.0), y(c.W, 0.0), diag(c.W, 0.0);
  devb::CurvatureScratch dev_sc;      // (device vectors of the batches estimated on the device, kept over the chunks)
  std::vector<uint8_t> seen(c.W, 0);
  std::vector<uint32_t> touched;
  double lam_max = 0.0;
  auto rec_d = [&](uint32_t e, bool cat) -> double { return record_delta(c, e, cat); };
  auto triggers = [&](uint32_t m) {
    return s->opts.learn_non_evidence || (!s->opts.noise_aware && (m & VM_EVIDENCE)) ||
           (s->opts.noise_aware && (m & VM_TRUTHINESS));
  };
  for (size_t l = 0; l + 1 < c.launch_off.size(); ++l) {
    const std::vector<uint32_t> cut = cut_launch(s, l, batches);
    for (size_t b = 0; b + 1 < cut.size(); ++b) {
      const uint32_t ta = cut[b], tb = cut[b + 1];
      if (s->sgd_work[tb] == s->sgd_work[ta]) continue;   // nothing learns in this run
      const uint32_t p0 = c.tile_v[ta], p1 = c.tile_v[tb];
      // y = H x, three times: x0 = 1 (y = row sums), then two normalised power steps
      double lam = 0.0, dmax = 0.0;
      if ((uint64_t)(p1 - p0) * 8 >= c.W && p1 - p0 >= 65536 && devb::available() && !getenv("DWX_HOST_BUILD")) {
        // ... on the device: the arithmetic of the branch below, term for term (device_build.hip)
        rt::set_device(s->device);
        const double v = devb::batch_curvature(p0, p1, s->d_v_meta, s->d_v_row, s->d_row_ptr, s->d_edges, s->d_edge_fval64,
                                               s->opts.learn_non_evidence != 0, s->opts.noise_aware != 0, (uint32_t)c.W, dev_sc,
                                               (void *)s->stream);
        lam = v; dmax = 0.0;
      } else if ((uint64_t)(p1 - p0) * 8 >= c.W && p1 - p0 >= 65536) {
        // a batch that touches a good part of the weight table: variable ranges in parallel on ALL host
        // threads, ONE shared y (and diagonal) in 64-bit FIXED POINT -- integer atomic adds: the sums
        // do not depend on the order, lambda is reproducible -- and dense passes over the table.
        // (Round 4: this was one private f64 vector per thread, hence 16 threads at most and a merge
        // pass over 16 W doubles per iteration: 6 s of config 5's sampler create at W = 10 M.)
        const uint32_t T = host_threads();
        // bound of one term kappa d dot (x <= 1 entrywise: dot <= the variable's sum of deltas) and of
        // a weight's number of terms, for the scale
        std::vector<double> umax(T, 0.0), dmx(T, 0.0);
        std::vector<uint64_t> nrec(T, 0);
        parallel_parts(p1 - p0, T, [&](uint32_t t, uint64_t b, uint64_t e) {
          double u = 0.0, dm = 0.0;
          uint64_t n = 0;
          for (uint32_t p = p0 + (uint32_t)b; p < p0 + (uint32_t)e; ++p) {
            const uint32_t m = c.v_meta[p];
            if (!triggers(m)) continue;
            const bool cat = (m & VM_CATEGORICAL) != 0;
            const uint32_t e0 = c.row_ptr[c.v_row[p]], e1 = c.row_ptr[c.v_row[p + 1]];
            double S = 0.0, dv = 0.0;
            for (uint32_t k = e0; k < e1; ++k) { const double d = rec_d(k, cat); S += d; dv = std::max(dv, d); }
            u = std::max(u, (cat ? 0.5 : 0.25) * dv * S);
            dm = std::max(dm, (cat ? 0.5 : 0.25) * dv * dv);
            n += e1 - e0;
          }
          umax[t] = u; dmx[t] = dm; nrec[t] = n;
        }, 0);
        double U = 0.0, D2 = 0.0;
        uint64_t R = 0;
        for (uint32_t t = 0; t < T; ++t) { U = std::max(U, umax[t]); D2 = std::max(D2, dmx[t]); R += nrec[t]; }
        if (U > 0.0 && R > 0) {
        const double scale = std::ldexp(1.0, 62) / ((double)R * U), dscale = std::ldexp(1.0, 62) / ((double)R * D2);
        RawArray<long long> yfix(c.W), dfix(c.W);
        parallel_ranges(c.W, T, [&](uint64_t b, uint64_t e) { for (uint64_t w = b; w < e; ++w) { yfix[w] = 0; dfix[w] = 0; } });
        std::fill(x.begin(), x.end(), 1.0);
        for (int iter = 0; iter < 3; ++iter) {
          parallel_parts(p1 - p0, T, [&](uint32_t, uint64_t b, uint64_t e) {
            for (uint32_t p = p0 + (uint32_t)b; p < p0 + (uint32_t)e; ++p) {
              const uint32_t m = c.v_meta[p];
              if (!triggers(m)) continue;
              const uint32_t e0 = c.row_ptr[c.v_row[p]], e1 = c.row_ptr[c.v_row[p + 1]];
              const bool cat = (m & VM_CATEGORICAL) != 0;
              const double kappa = cat ? 0.5 : 0.25;
              double dot = 0.0;
              for (uint32_t k = e0; k < e1; ++k) dot += rec_d(k, cat) * x[c.edges[k].wid];
              for (uint32_t k = e0; k < e1; ++k) {
                const double d = rec_d(k, cat);
                if (d == 0.0) continue;
                __atomic_fetch_add(&yfix[c.edges[k].wid], (long long)std::llrint(scale * (kappa * d * dot)), __ATOMIC_RELAXED);
                if (iter == 0) __atomic_fetch_add(&dfix[c.edges[k].wid], (long long)std::llrint(dscale * (kappa * d * d)), __ATOMIC_RELAXED);
              }
            }
          }, 0);
          std::vector<double> part(3 * T, 0.0);
          parallel_parts(c.W, T, [&](uint32_t t, uint64_t b, uint64_t e) {
            double xy = 0.0, xx = 0.0, yy = 0.0;
            for (uint64_t w = b; w < e; ++w) {
              const double yw = (double)yfix[w] / scale;
              y[w] = yw;
              xy += x[w] * yw; xx += x[w] * x[w]; yy += yw * yw;
            }
            part[3 * t] = xy; part[3 * t + 1] = xx; part[3 * t + 2] = yy;
          }, 0);
          double xy = 0.0, xx = 0.0, yy = 0.0;
          for (uint32_t t = 0; t < T; ++t) { xy += part[3 * t]; xx += part[3 * t + 1]; yy += part[3 * t + 2]; }
          if (xx > 0) lam = std::max(lam, xy / xx);
          const double norm = yy > 0 ? 1.0 / std::sqrt(yy) : 0.0;
          parallel_ranges(c.W, T, [&](uint64_t b, uint64_t e) { for (uint64_t w = b; w < e; ++w) { x[w] = y[w] * norm; y[w] = 0.0; yfix[w] = 0; } });
        }
        std::vector<double> dpart(T, 0.0);
        parallel_parts(c.W, T, [&](uint32_t t, uint64_t b, uint64_t e) {
          long long mx = 0;
          for (uint64_t w = b; w < e; ++w) { mx = std::max(mx, dfix[w]); x[w] = 0.0; }
          dpart[t] = (double)mx / dscale;
        }, 0);
        for (uint32_t t = 0; t < T; ++t) dmax = std::max(dmax, dpart[t]);
        } else {
          std::fill(x.begin(), x.end(), 0.0);
        }
      } else {
      for (int iter = 0; iter < 3; ++iter) {
        for (uint32_t p = p0; p < p1; ++p) {
          const uint32_t m = c.v_meta[p];
          if (!triggers(m)) continue;
          const uint32_t e0 = c.row_ptr[c.v_row[p]], e1 = c.row_ptr[c.v_row[p + 1]];
          const double kappa = (m & VM_CATEGORICAL) ? 0.5 : 0.25;
          double dot = 0.0;
          for (uint32_t e = e0; e < e1; ++e) {
            const double d = rec_d(e, (m & VM_CATEGORICAL) != 0);
            if (d == 0.0) continue;
            const uint32_t w = c.edges[e].wid;
            if (!seen[w]) { seen[w] = 1; touched.push_back(w); x[w] = 1.0; }
            dot += d * x[w];
          }
          for (uint32_t e = e0; e < e1; ++e) {
            const double d = rec_d(e, (m & VM_CATEGORICAL) != 0);
            if (d == 0.0) continue;
            y[c.edges[e].wid] += kappa * d * dot;
            if (iter == 0) diag[c.edges[e].wid] += kappa * d * d;   // (lower bound of H_ww)
          }
        }
        double xy = 0.0, xx = 0.0, yy = 0.0;
        for (uint32_t w : touched) { xy += x[w] * y[w]; xx += x[w] * x[w]; yy += y[w] * y[w]; }
        if (xx > 0) lam = std::max(lam, xy / xx);
        const double norm = yy > 0 ? 1.0 / std::sqrt(yy) : 0.0;
        for (uint32_t w : touched) { x[w] = y[w] * norm; y[w] = 0.0; }
      }
      for (uint32_t w : touched) { dmax = std::max(dmax, diag[w]); diag[w] = 0.0; seen[w] = 0; x[w] = 0.0; }
      touched.clear();
      }
      lam_max = std::max(lam_max, 1.1 * std::max(lam, dmax));
    }
  }
  if (getenv("DWX_TIMING")) fprintf(stderr, "[dwx curvature] batches=%u lambda=%.6g\n", batches, lam_max);
  s->row_sum_cache[batches] = lam_max;
  return lam_max;
}

void make_plan(dwx_sampler *s, double stepsize, uint32_t force_batches) {
  const CompiledGraph &c = *s->cg;
  uint32_t max_tiles = 1;
  for (size_t l = 0; l + 1 < c.launch_off.size(); ++l)
    max_tiles = std::max(max_tiles, c.launch_tile[l + 1] - c.launch_tile[l]);
  // No plan cuts finer than MAX_PLAN_BATCHES.  Stability does not depend on the cut: every
  // weight's step saturates at the inverse of its own curvature bound (apply_kernel), so a batch
  // can never overshoot.  The cut decides how closely the batches follow the reference's
  // sequential updates: inside a batch all draws see the weights of its start.
  max_tiles = std::min(max_tiles, MAX_PLAN_BATCHES);
  const double cap = s->opts.step_cap;
  uint32_t B = 1;
  const double eta = stepsize;
  if (force_batches) {
    // A forced count is the caller's GLOBAL plan (multi-GPU: every rank must run the same number
    // of batches, apply after each chunk and put the same vectors through the collectives): it is
    // NOT clamped by this shard's own tile count -- a shard with fewer tiles than batches simply
    // has fewer chunks and idles through the rest (a one-tile shard used to fall
    // back to plan_batches = 1, skip collectives its peers entered and apply whole-sweep counts).
    B = std::min(force_batches, MAX_PLAN_BATCHES);
  } else if (cap > 0 && stepsize > 0) {
    // Double the cut while the batches' curvature is above the cap AND cutting still lowers it:
    // a batch can be no finer than one variable, so a hub with 10^5 factors sets a floor no cut
    // gets under (its own kappa |d|^2) -- past it more batches only cost dependent launches
    // (the saturating step keeps the update stable there, section 3.5).
    while (B < max_tiles && stepsize * row_sum_bound(s, B) > cap) {
      // (looking two doublings ahead: one cut may fall on an unlucky boundary)
      if (row_sum_bound(s, 2 * B) > PLAN_MIN_GAIN * row_sum_bound(s, B) &&
          (4 * B > max_tiles || row_sum_bound(s, 4 * B) > PLAN_MIN_GAIN * PLAN_MIN_GAIN * row_sum_bound(s, B)))
        break;
      B *= 2;
    }
    B = std::min(B, max_tiles);
  }
  s->plan_batches = B;
  s->plan_eta = eta;
  // The step decays from sweep to sweep, so a run that starts at B batches walks down through
  // B/2, B/4, ...  Those levels are built when a plan first NEEDS them (round 4: at config 5's size a
  // level costs seconds of host time, and a `-l 10` run that starts at 4 batches never reaches 2 --
  // building them all up front was 15 s of its first learning epoch).  A caller that wants the
  // one-off work out of a timed region plans the coarser counts itself (bench.py: dist.py's
  // prepare), or sets DWX_EAGER_LEVELS.
  if (B > 1 && !s->levels.count(B) && getenv("DWX_EAGER_LEVELS")) {
    for (uint32_t b = B / 2; b >= 2; b /= 2) {
      if (cap > 0) (void)row_sum_bound(s, b);
      (void)build_level(s, b);
    }
  }
  s->plan_level = build_level(s, B);
  s->plan_chunks = s->plan_level->chunks;
  s->cur_chunk = 0;
  s->plan_force_dynamic = false;
  s->plan_valid = true;
}

// The smallest step any weight takes under the current plan: stepsize saturated by the largest
// curvature bound (+ regularisation pull) of the plan's tables -- what `effective_stepsize`
// reports (== stepsize for weights with few factors; ~1/h for heavily tied ones).
double plan_min_step(dwx_sampler *s) {
  const dwx_sampler::Level *L = s->plan_level;
  if (!L || L->c_max <= 0.0 || s->plan_eta <= 0.0) return s->plan_eta;
  const double r = s->opts.regularization == 1 ? s->opts.reg_param * L->t_max : 0.0;
  const double c = DWX_CURV_MID * L->c_max + r, st = -std::expm1(-c * s->plan_eta) / c;
  return st * (L->c_max + r) > 1.0 ? 1.0 / (L->c_max + r) : st;   // batch_step (aux_kernels.h)
}

uint64_t layout_after_sweeps() {
  uint64_t n = 2048;
  if (const char *e = getenv("DWX_LAYOUT_AFTER_SWEEPS")) n = (uint64_t)std::max(1L, atol(e));   // test hook
  return n;
}

// accumulate the gradient of one chunk of the plan (sampling both chains on the way)
void enqueue_learn_chunk(dwx_sampler *s, uint32_t chunk) {
  rt::set_device(s->device);
  if (chunk >= s->plan_chunks.size()) return;   // ranks with fewer chunks idle through the rest
  const dwx_sampler::Chunk &ch = s->plan_chunks[chunk];
  KernelParams P = s->base;
  P.sweep = s->sweep;
  // a level that has run this many sweeps is worth a layout of its own (dwx_options.plan_layouts
  // == 0: the run pays 0.3 s once where it has already spent as much on the default layout)
  if (chunk == 0 && s->plan_level && ++s->plan_level->sweeps == layout_after_sweeps() && s->opts.plan_layouts == 0)
    ensure_level_layout(s, s->plan_level, s->plan_batches);
  const dwx_sampler::Level &L = *s->plan_level;
  const bool split = s->plan_batches > 1;
  // a split sweep without per-chunk tables falls back to per-record atomics and counts
  const bool fast = L.fast && !(split && s->plan_force_dynamic);
  if (split && !fast) P.flags |= OPT_DYNAMIC_T | OPT_NO_PULL;
  TimedSpan sp{};
  const bool timing = s->timing;
  if (timing) {
    sp.a = rt::event_create(); sp.b = rt::event_create(); sp.c = rt::event_create(); sp.kind = 1;
    rt::event_record(sp.a, s->stream);
  }
  const uint32_t launches = launch_tiles<true>(s, P, ch.launch, ch.t0, ch.t1);
  if (timing) rt::event_record(sp.b, s->stream);
  bool pulled = false;
  // the pull-based gradient of the TILE_PULL tiles: un-split sweeps once, after the last
  // chunk (= colour launch), over the whole list; split sweeps per chunk over its part
  uint32_t pb = 0, pe = 0;
  if (fast && !L.inc_end.empty()) {
    if (!split) { if (chunk + 1 == s->plan_chunks.size()) { pb = L.inc_begin.front(); pe = L.inc_end.back(); } }
    else { pb = L.inc_begin[chunk]; pe = L.inc_end[chunk]; }
  }
  // block pull of the same group as the list: the whole sweep (un-split, after the last colour
  // launch) or this chunk; what did not fit the block tables' rows stays on the weight-sorted list
  // (config 3: 2.8 % of the entries) for pull_grad_kernel.  (The list walked on a side stream
  // beside pull_ell_kernel, fold_partials_kernel waiting for both: bench step 0.5523 -> 0.5507 ms,
  // pull 0.1127 -> 0.1108 -- inside the noise, not kept.)
  const dwx_sampler::Level::BlockTable *bt = nullptr;
  if (fast && !L.bp.empty() && (split || chunk + 1 == s->plan_chunks.size())) {
    bt = &L.bp[split ? chunk : 0];
    if (!bt->blocks) bt = nullptr;
  }
  auto launch_list = [&](rt::stream_t st) {
    const uint32_t n = pe - pb;
    const unsigned chunk_sz = BLOCK_THREADS * PULL_RUN;
    const unsigned grid = std::min<unsigned>((n + chunk_sz - 1) / chunk_sz, 256u * DWX_PULL_GRID);
    auto go = [&](auto kernel) {
      rt::launch(kernel, grid, BLOCK_THREADS, 0, st, (const uint32_t *)(L.d_inc_wid + pb),
                 (const uint32_t *)(L.d_inc_slot + pb), (const float *)(L.d_inc_d + pb), n,
                 (const unsigned long long *)s->d_delta, s->d_grad);
    };
    // (a weight's entries span lanes only when it has many: what the block tables left over of a
    // million weights does not)
    if ((uint64_t)n >= 8 * s->cg->W) go(pull_grad_kernel<true>); else go(pull_grad_kernel<false>);
    pulled = true;
  };
  if (bt) {
    const unsigned grid = bt->blocks * bt->parts;
    auto go = [&](auto kernel) {
      rt::launch(kernel, grid, BP_THREADS, BP_LDS_BYTES, s->stream, (const U32x4 *)bt->d_ell,
                 (const uint32_t *)bt->d_tile0, bt->parts, (const long long *)L.d_bp_qtab, L.bp_deltas, L.bp_wp,
                 (const unsigned long long *)s->d_delta, L.d_bp_partial);
    };
    const bool uni = L.bp_deltas == 1;
    if (bt->depth == 2) { if (uni) go(pull_ell_kernel<2, true>); else go(pull_ell_kernel<2, false>); }
    else { if (uni) go(pull_ell_kernel<1, true>); else go(pull_ell_kernel<1, false>); }
    const uint32_t W = (uint32_t)s->cg->W;
    const unsigned fgrid = std::min<unsigned>((W + BLOCK_THREADS - 1) / BLOCK_THREADS, 4096u);
    if (uni)
      rt::launch(fold_partials_kernel<true>, fgrid, BLOCK_THREADS, 0, s->stream, (const long long *)L.d_bp_partial,
                 bt->blocks, L.bp_wp, W, s->d_grad, (const long long *)L.d_bp_qtab);
    else
      rt::launch(fold_partials_kernel<false>, fgrid, BLOCK_THREADS, 0, s->stream, (const long long *)L.d_bp_partial,
                 bt->blocks, L.bp_wp, W, s->d_grad, (const long long *)L.d_bp_qtab);
    pulled = true;
  }
  if (pe > pb) launch_list(s->stream);
  if (timing) {
    rt::event_record(sp.c, s->stream);
    sp.launches = launches; sp.has_pull = pulled; sp.new_sweep = chunk == 0;
    s->spans.push_back(sp);
  }
}

void enqueue_apply(dwx_sampler *s) {
  rt::set_device(s->device);
  s->terms_state = 0;   // the weights change
  const uint32_t W = (uint32_t)s->cg->W;
  if (!W) return;
  const unsigned grid = std::min<unsigned>((W + BLOCK_THREADS - 1) / BLOCK_THREADS, 2048u);
  // static tables of what was just accumulated: row pair 0 of an un-split sweep, the chunk's of
  // a split one.  On the atomics fallback (a split plan without tables, or one a multi-GPU
  // driver forced to count dynamically) the counts are in the grad buffer and the curvature
  // bounds are those of the WHOLE sweep (level 1: >= any chunk's, so the step only saturates
  // earlier)
  const dwx_sampler::Level *L = s->plan_level;
  const bool split = s->plan_batches > 1;
  const long long *ts = nullptr, *hs = nullptr;
  if (L && L->fast && L->d_t_static && !(split && s->plan_force_dynamic)) {
    const uint32_t row = split ? s->cur_chunk : 0u;
    if (row < L->rows) { ts = L->d_t_static + (size_t)row * 2 * W; hs = ts + W; }
  } else {
    auto it = s->levels.find(1);
    if (it != s->levels.end() && it->second->d_t_static) hs = it->second->d_t_static + W;
  }
  rt::launch(apply_kernel, grid, BLOCK_THREADS, 0, s->stream, s->d_weights, s->d_w32,
             (const uint8_t *)s->d_w_fixed, s->d_grad, ts, hs, W, s->plan_eta, s->opts.reg_param,
             (int)(s->opts.regularization == 1), (const double *)nullptr, (long long *)nullptr);
}

// A split learning sweep of an all-unary graph with few weights (LDS gradient accumulators), one launch
// per mini-batch: the update of mini-batch c - 1 is the prologue of mini-batch c's sweep kernel
// (sweep8_merged_kernel, persist_kernels.h); three gradient buffers and two weight buffers in turn, the
// last update by apply_kernel into the sampler's own arrays.  false: the sweep does not qualify.
bool enqueue_merged_sweep(dwx_sampler *s) {
  dwx_sampler::Level *L = s->plan_level;
  const uint32_t n = (uint32_t)s->plan_chunks.size();
  const CompiledGraph &c = *s->cg;
  const uint32_t W = (uint32_t)c.W;
  if (!s->merge_ok || !L || s->plan_batches < 2 || n < 2 || !L->fast || !L->d_t_static || L->rows < n || s->plan_force_dynamic)
    return false;
  rt::set_device(s->device);
  if (!s->d_gbuf[1]) {
    for (int k = 1; k < 3; ++k) {
      s->d_gbuf[k] = (long long *)rt::dmalloc((size_t)W * 16);
      rt::dmemset(s->d_gbuf[k], 0, (size_t)W * 16, s->stream);
    }
    for (int k = 0; k < 2; ++k) { s->d_wbuf64[k] = (double *)rt::dmalloc((size_t)W * 8); s->d_wbuf32[k] = (float *)rt::dmalloc((size_t)W * 4 + 4); }
  }
  s->d_gbuf[0] = s->d_grad;
  if (s->plan_level) ++s->plan_level->sweeps;
  const unsigned cap = s->persistent_blocks8[1];
  // The leading tiles of the first mini-batch in which NOTHING learns -- the query part of the colour launch:
  // half of config 4's tiles, 190 us of bandwidth work at the head of 63 latency-bound launches of 16 us -- run
  // beside the mini-batches instead of ahead of them: on a side stream, one workgroup per CU (the mini-batches'
  // workgroups keep the other two slots of every CU).  They read the sweep's starting weights -- the sampler's
  // own arrays, which no merged launch writes; the closing update waits for them -- add nothing to any gradient
  // (flush_accumulators skips zeros) and, on an all-unary graph, nobody reads their samples: the same results.
  uint32_t quiet_end = s->plan_chunks[0].t0;
  if (s->side[0] && s->ev_fork && s->ev_join[0] && !getenv("DWX_NO_QUIET_ASIDE")) {
    const dwx_sampler::Chunk &c0 = s->plan_chunks[0];
    while (quiet_end < c0.t1 && quiet_end + 1 < s->sgd_work.size() && s->sgd_work[quiet_end + 1] == s->sgd_work[quiet_end]) ++quiet_end;
    if (quiet_end - c0.t0 < 4 * rt::grid_barrier_blocks()) quiet_end = c0.t0;   // (too few to be worth a launch)
  }
  const bool aside = quiet_end > s->plan_chunks[0].t0;
  if (aside) {
    KernelParams Q = s->base;
    Q.sweep = s->sweep;
    Q.tile_begin = s->plan_chunks[0].t0; Q.tile_end = quiet_end;
    rt::event_record(s->ev_fork, s->stream);
    rt::stream_wait_event(s->side[0], s->ev_fork);
    unsigned side_wgs = std::max(1u, rt::grid_barrier_blocks());
    if (const char *e = getenv("DWX_QUIET_GRID")) side_wgs = (unsigned)std::max(1L, atol(e));   // (tuning knob)
    const unsigned grid = std::min<unsigned>(quiet_end - Q.tile_begin, side_wgs);
    const size_t lds = s->lds_bytes[1];
    if (s->rp_cat) rt::launch(sweep8_kernel<true, 6, false, (int)ROWPTR_UNROLL_CAT>, grid, BLOCK_THREADS, lds, s->side[0], Q);
    else switch (s->stage_k) {
      case 3: rt::launch(sweep8_kernel<true, 3>, grid, BLOCK_THREADS, lds, s->side[0], Q); break;
      case 6: rt::launch(sweep8_kernel<true, 6>, grid, BLOCK_THREADS, lds, s->side[0], Q); break;
      default: rt::launch(sweep8_kernel<true, 12>, grid, BLOCK_THREADS, lds, s->side[0], Q); break;
    }
    rt::event_record(s->ev_join[0], s->side[0]);
  }
  for (uint32_t ci = 0; ci < n; ++ci) {
    dwx_sampler::Chunk ch = s->plan_chunks[ci];
    if (ci == 0) ch.t0 = quiet_end;
    s->cur_chunk = ci;
    KernelParams P = s->base;
    P.sweep = s->sweep;
    P.grad = s->d_gbuf[ci % 3];
    P.tile_begin = ch.t0; P.tile_end = ch.t1;
    MergeArgs M{};
    M.prev_grad = ci ? s->d_gbuf[(ci - 1) % 3] : nullptr;
    M.zero = s->d_gbuf[(ci + 1) % 3];
    M.w_src = ci <= 1 ? s->d_weights : s->d_wbuf64[(ci - 2) % 2];
    M.w_dst = ci ? s->d_wbuf64[(ci - 1) % 2] : nullptr;
    M.w32_dst = ci ? s->d_wbuf32[(ci - 1) % 2] : nullptr;
    M.w_fixed = s->d_w_fixed;
    M.t_static = ci ? L->d_t_static + (size_t)(ci - 1) * 2 * W : nullptr;
    M.stepsize = s->plan_eta; M.reg_param = s->opts.reg_param;
    M.l2 = s->opts.regularization == 1 ? 1 : 0;
    M.lds_lw32_off = s->merge_lw32_off;
    TimedSpan sp{};
    if (s->timing) {
      sp.a = rt::event_create(); sp.b = rt::event_create(); sp.c = rt::event_create(); sp.kind = 1;
      rt::event_record(sp.a, s->stream);
    }
    if (ch.t1 > ch.t0 && ch.t1 - ch.t0 <= cap && !getenv("DWX_NO_ONE_TILE")) {
      // a workgroup per tile (the usual shape of a mini-batch): the build without the next-tile machinery
      const unsigned grid = ch.t1 - ch.t0;
      if (s->rp_cat) rt::launch(sweep8_merged_kernel<6, (int)ROWPTR_UNROLL_CAT, true>, grid, BLOCK_THREADS, s->lds_merge, s->stream, P, M);
      else switch (s->stage_k) {
        case 3: rt::launch(sweep8_merged_kernel<3, (int)ROWPTR_UNROLL, true>, grid, BLOCK_THREADS, s->lds_merge, s->stream, P, M); break;
        case 6: rt::launch(sweep8_merged_kernel<6, (int)ROWPTR_UNROLL, true>, grid, BLOCK_THREADS, s->lds_merge, s->stream, P, M); break;
        default: rt::launch(sweep8_merged_kernel<12, (int)ROWPTR_UNROLL, true>, grid, BLOCK_THREADS, s->lds_merge, s->stream, P, M); break;
      }
    } else if (ch.t1 > ch.t0) {
      const unsigned grid = std::min<unsigned>(ch.t1 - ch.t0, cap);
      if (s->rp_cat) rt::launch(sweep8_merged_kernel<6, (int)ROWPTR_UNROLL_CAT>, grid, BLOCK_THREADS, s->lds_merge, s->stream, P, M);
      else switch (s->stage_k) {
        case 3: rt::launch(sweep8_merged_kernel<3, (int)ROWPTR_UNROLL>, grid, BLOCK_THREADS, s->lds_merge, s->stream, P, M); break;
        case 6: rt::launch(sweep8_merged_kernel<6, (int)ROWPTR_UNROLL>, grid, BLOCK_THREADS, s->lds_merge, s->stream, P, M); break;
        default: rt::launch(sweep8_merged_kernel<12, (int)ROWPTR_UNROLL>, grid, BLOCK_THREADS, s->lds_merge, s->stream, P, M); break;
      }
    } else {
      // (an empty chunk: the update alone, through a one-workgroup launch over no tile)
      P.tile_begin = P.tile_end = 0;
      if (s->rp_cat) rt::launch(sweep8_merged_kernel<6, (int)ROWPTR_UNROLL_CAT>, 1u, BLOCK_THREADS, s->lds_merge, s->stream, P, M);
      else switch (s->stage_k) {
        case 3: rt::launch(sweep8_merged_kernel<3, (int)ROWPTR_UNROLL>, 1u, BLOCK_THREADS, s->lds_merge, s->stream, P, M); break;
        case 6: rt::launch(sweep8_merged_kernel<6, (int)ROWPTR_UNROLL>, 1u, BLOCK_THREADS, s->lds_merge, s->stream, P, M); break;
        default: rt::launch(sweep8_merged_kernel<12, (int)ROWPTR_UNROLL>, 1u, BLOCK_THREADS, s->lds_merge, s->stream, P, M); break;
      }
    }
    if (s->timing) {
      rt::event_record(sp.b, s->stream);
      rt::event_record(sp.c, s->stream);
      sp.launches = 1; sp.has_pull = false; sp.new_sweep = ci == 0;
      s->spans.push_back(sp);
    }
  }
  // the last mini-batch's update: from the weights the last merged launch wrote, into the sampler's own arrays
  if (aside) rt::stream_wait_event(s->stream, s->ev_join[0]);   // (the quiet tiles still read those arrays)
  {
    const uint32_t last = n - 1;
    const unsigned grid = std::min<unsigned>((W + BLOCK_THREADS - 1) / BLOCK_THREADS, 2048u);
    const long long *ts = L->d_t_static + (size_t)last * 2 * W;
    const double *w_in = last == 0 ? (const double *)nullptr : (const double *)s->d_wbuf64[(last - 1) % 2];
    rt::launch(apply_kernel, grid, BLOCK_THREADS, 0, s->stream, s->d_weights, s->d_w32, (const uint8_t *)s->d_w_fixed,
               s->d_gbuf[last % 3], ts, ts + W, W, s->plan_eta, s->opts.reg_param, (int)(s->opts.regularization == 1), w_in,
               last ? s->d_gbuf[(last - 1) % 3] : (long long *)nullptr);
  }
  s->terms_state = 0;
  ++s->merged_sweeps;
  return true;
}

// A split learning sweep of a few-weights, all-unary graph as ONE persistent launch
// (persist_kernels.h): every chunk's draws, gradient rows, the barrier and the update inside it.
// false: this sweep does not qualify (the caller enqueues it chunk by chunk).
bool enqueue_persistent_sweep(dwx_sampler *s) {
  dwx_sampler::Level *L = s->plan_level;
  const uint32_t n = (uint32_t)s->plan_chunks.size();
  if (!s->persist_grid || !L || s->plan_batches < 2 || n < PERSIST_MIN_CHUNKS || !L->fast || !L->d_t_static ||
      L->rows < n || s->plan_force_dynamic)
    return false;
  rt::set_device(s->device);
  uint32_t most = 0;
  for (const auto &ch : s->plan_chunks) most = std::max(most, ch.t1 - ch.t0);
  if (!most) return false;
  if (!L->d_chunk_tiles) {
    std::vector<uint32_t> ct;
    for (const auto &ch : L->chunks) { ct.push_back(ch.t0); ct.push_back(ch.t1); }
    L->d_chunk_tiles = upload(ct, s->stream);
    rt::stream_sync(s->stream);      // (the host copy dies with this scope)
  }
  ++L->sweeps;
  KernelParams P = s->base;
  P.sweep = s->sweep;
  PersistArgs A{};
  A.chunk_tiles = L->d_chunk_tiles;
  A.n_chunks = n;
  A.grid = std::min(most, s->persist_grid);
  A.rows = s->d_persist_rows;
  A.row_stride = s->persist_row_stride;
  A.bar = s->d_persist_bar;
  A.t_static = L->d_t_static;
  A.weights = s->d_weights; A.w32 = s->d_w32; A.w_fixed = s->d_w_fixed;
  A.stepsize = s->plan_eta; A.reg_param = s->opts.reg_param;
  A.l2 = s->opts.regularization == 1 ? 1 : 0;
  A.spin_limit = 4u << 20;
  A.lds_w64_off = s->persist_w64_off; A.lds_red_off = s->persist_red_off;
  TimedSpan sp{};
  if (s->timing) {
    sp.a = rt::event_create(); sp.b = rt::event_create(); sp.c = rt::event_create(); sp.kind = 1;
    rt::event_record(sp.a, s->stream);
  }
  rt::dmemset(s->d_persist_bar, 0, 16, s->stream);
  if (s->rp_cat) rt::launch(persist_learn8_kernel<6, (int)ROWPTR_UNROLL_CAT>, A.grid, BLOCK_THREADS, s->lds_persist, s->stream, P, A);
  else switch (s->stage_k) {
    case 3: rt::launch(persist_learn8_kernel<3, (int)ROWPTR_UNROLL>, A.grid, BLOCK_THREADS, s->lds_persist, s->stream, P, A); break;
    case 6: rt::launch(persist_learn8_kernel<6, (int)ROWPTR_UNROLL>, A.grid, BLOCK_THREADS, s->lds_persist, s->stream, P, A); break;
    default: rt::launch(persist_learn8_kernel<12, (int)ROWPTR_UNROLL>, A.grid, BLOCK_THREADS, s->lds_persist, s->stream, P, A); break;
  }
  if (s->timing) {
    rt::event_record(sp.b, s->stream);
    rt::event_record(sp.c, s->stream);
    sp.launches = 1; sp.has_pull = false; sp.new_sweep = true;
    s->spans.push_back(sp);
  }
  s->terms_state = 0;               // the weights change
  s->cur_chunk = n - 1;
  s->persist_check_pending = true;
  ++s->persist_launches;
  return true;
}

void drain_spans(dwx_sampler *s) {
  if (s->spans.empty()) return;
  rt::stream_sync(s->stream);
  for (auto &sp : s->spans) {
    s->t_ms[sp.kind] += rt::event_elapsed_ms(sp.a, sp.b);
    s->t_launches[sp.kind] += sp.launches;
    s->t_sweeps[sp.kind] += sp.new_sweep ? sp.n_sweeps : 0;   // (a learning sweep is one or more chunks)
    if (sp.has_pull) {
      s->t_ms[2] += rt::event_elapsed_ms(sp.b, sp.c);
      s->t_launches[2] += 1;
    }
    if (sp.kind == 1 && sp.new_sweep) s->t_sweeps[2] += 1;
    rt::event_destroy(sp.a); rt::event_destroy(sp.b); rt::event_destroy(sp.c);
  }
  s->spans.clear();
}
}  // namespace

extern "C" {

const char *dwx_last_error(void) { return g_err.c_str(); }
int dwx_version(void) { return DWX_VERSION; }

void dwx_default_options(dwx_options *o) {
  std::memset(o, 0, sizeof *o);
  o->regularization = 1;  // l2 (src/cmd_parser.cc:163-166)
  o->reg_param = 0.01;    // src/cmd_parser.cc:162
  o->step_cap = 1.5;   // stepsize * lambda of one SGD mini-batch (the hard limit is 2)
  o->seed = 0x5eed5eedULL;
}

// ------------------------------------------------------------------ graph
int dwx_graph_create(const dwx_graph_desc *desc, const dwx_compile_opts *opts, dwx_graph **out) {
  if (!desc || !out) return fail(DWX_E_INVALID, "null argument");
  dwx_compile_opts o{};
  if (opts) o = *opts;
  // the weight-sorted copy of the records is built on the sampler's device where the library can
  // (device_build.hip; DWX_HOST_BUILD=1: the host builder, which is also its checker)
  o.defer_sorted_records = devb::available() && !getenv("DWX_HOST_BUILD");
  bool limit = false;
  try {
    auto cg = std::make_shared<CompiledGraph>();
    compile_graph(*desc, o, *cg, &limit);
    *out = new dwx_graph{cg};
    return DWX_OK;
  } catch (const std::bad_alloc &) {
    return fail(DWX_E_NOMEM, "out of host memory while compiling the graph");
  } catch (const std::exception &e) {
    return fail(limit ? DWX_E_LIMIT : DWX_E_INVALID, e.what());
  }
}

void dwx_graph_destroy(dwx_graph *g) { delete g; }

int dwx_graph_get_info(const dwx_graph *g, dwx_graph_info *out) {
  if (!g || !out) return fail(DWX_E_INVALID, "null argument");
  const CompiledGraph &c = *g->cg;
  std::memset(out, 0, sizeof *out);
  out->num_owned_variables = c.Vo;
  out->num_variables = c.V; out->num_factors = c.F; out->num_edges = c.E; out->num_weights = c.W;
  out->num_values = c.R; out->num_index_entries = c.NIdx; out->num_vif_entries = c.NVif;
  out->num_colors = c.n_colors; out->num_launches = c.launch_off.size() - 1;
  out->num_tiles = c.tile_v.size() - 1; out->num_giant_tiles = c.n_giant_tiles;
  out->num_wide_tiles = c.n_wide_tiles;
  out->num_staged_tiles = c.ecap <= 6 * BLOCK_THREADS ? c.n_terms2_tiles : 0;
  out->max_cardinality = c.max_card; out->device_bytes = c.device_bytes();
  out->num_query_variables = c.n_query;
  out->has_categorical = c.has_categorical; out->order_is_identity = c.order_is_identity;
  out->num_super_tiles = c.supers.size(); out->num_sorted_records = c.n_sorted;
  out->grad_shift = c.grad_shift; out->grad_unit_max = c.grad_unit_max; out->max_records_per_weight = c.max_records_per_weight;
  return DWX_OK;
}

int dwx_graph_get_schedule(const dwx_graph *g, uint64_t *order, uint64_t *launch_off) {
  if (!g || !order || !launch_off) return fail(DWX_E_INVALID, "null argument");
  const CompiledGraph &c = *g->cg;
  for (uint64_t p = 0; p < c.Vo; ++p) order[p] = c.perm[p];
  std::copy(c.launch_off.begin(), c.launch_off.end(), launch_off);
  return DWX_OK;
}

int dwx_graph_get_fixed_point_mask(const dwx_graph *g, uint8_t *mask) {
  if (!g || !mask) return fail(DWX_E_INVALID, "null argument");
  const CompiledGraph &c = *g->cg;
  std::fill(mask, mask + c.V, (uint8_t)0);
  if (c.edges8.size() == 0) return DWX_OK;      // compact records only (sweep8_kernel, sorted_sweep_kernel)
  for (const TileDesc &t : c.tiles) {
    if (t.flags & (TILE_OUTSIDE | TILE_CATEGORICAL)) continue;   // wave / workgroup bins, value rows: f64 sums
    for (uint32_t l = 0; l < t.nv; ++l) mask[c.perm[t.v0 + l]] = 1;
  }
  return DWX_OK;
}

int dwx_graph_get_values(const dwx_graph *g, uint64_t *var_val_base, uint64_t *value_sparse) {
  if (!g) return fail(DWX_E_INVALID, "null argument");
  const CompiledGraph &c = *g->cg;
  // (by all host threads: 1.6 GB at config 5's size, and the first touch of the caller's arrays)
  auto copy = [](const std::vector<uint64_t> &src, uint64_t *dst) {
    parallel_ranges(src.size(), host_threads(), [&](uint64_t b, uint64_t e) { std::memcpy(dst + b, src.data() + b, (e - b) * 8); });
  };
  if (var_val_base) copy(c.ref_var_val_base, var_val_base);
  if (value_sparse) copy(c.value_sparse, value_sparse);
  return DWX_OK;
}

int dwx_graph_get_positions(const dwx_graph *g, const uint64_t *vids, uint64_t n, uint64_t *out) {
  if (!g || (n && (!vids || !out))) return fail(DWX_E_INVALID, "null argument");
  const CompiledGraph &c = *g->cg;
  for (uint64_t i = 0; i < n; ++i) {
    if (vids[i] >= c.V) return fail(DWX_E_INVALID, "variable id out of range");
    out[i] = c.pos[vids[i]];
  }
  return DWX_OK;
}

int dwx_graph_get_index(const dwx_graph *g, uint64_t *index_base, uint64_t *index_len,
                        uint64_t *factor_index) {
  if (!g) return fail(DWX_E_INVALID, "null argument");
  const CompiledGraph &c = *g->cg;
  for (uint64_t r = 0; r < c.R; ++r) {
    if (index_base) index_base[r] = c.ref_row_ptr[r];
    if (index_len) index_len[r] = c.ref_row_ptr[r + 1] - c.ref_row_ptr[r];
  }
  if (factor_index) for (uint64_t i = 0; i < c.NIdx; ++i) factor_index[i] = c.ref_fidx[i];
  return DWX_OK;
}

// ------------------------------------------------------------------ sampler
int dwx_sampler_create(const dwx_graph *g, const dwx_options *opts, dwx_sampler **out) {
  if (!g || !opts || !out) return fail(DWX_E_INVALID, "null argument");
  std::unique_ptr<dwx_sampler> s(new dwx_sampler());
  int rc = guarded([&]() {
    rt::init_device(opts->device);
    rt::oom_hook() = [](int dev) { devb::release_scratch(dev, false); };
    s->cg = g->cg;
    s->opts = *opts;
    s->device = opts->device;
    const CompiledGraph &c = *s->cg;
    s->stream = rt::stream_create();
    rt::stream_t st = s->stream;
    if (!getenv("DWX_NO_SIDE_STREAMS")) {   // (A/B knob)
      for (int i = 0; i < 2; ++i) { s->side[i] = rt::stream_create(); s->ev_join[i] = rt::event_create_ordering(); }
      s->ev_fork = rt::event_create_ordering();
    }
    const bool timing = getenv("DWX_TIMING") != nullptr;   // wall time of every phase on stderr
    auto t_phase = std::chrono::steady_clock::now();
    auto phase = [&](const char *what) {
      if (!timing) return;
      rt::stream_sync(st);
      auto t = std::chrono::steady_clock::now();
      fprintf(stderr, "[dwx sampler_create] %-24s %.3f s\n", what, std::chrono::duration<double>(t - t_phase).count());
      t_phase = t;
    };
    s->d_v_meta = upload(c.v_meta, st, 1);
    s->d_v_orig = upload(c.perm, st, 1);
    s->d_v_row = upload(c.v_row, st);
    s->d_v_init = upload(c.v_init, st, 1);
    s->d_row_ptr = upload(c.row_ptr, st, 1);
    {
      // edge-parallel evaluation of binary factors is compiled for K <= 6 only (register
      // budget): with 3072-record tiles such tiles take the generic path
      std::vector<TileDesc> tiles = c.tiles;
      if (c.ecap > 6 * BLOCK_THREADS) for (auto &t : tiles) t.flags &= ~(TILE_TERMS2 | TILE_TERMS3 | TILE_PULL_UNARY);
      s->d_tiles = upload(tiles, st);
    }
    for (uint32_t ti : c.giant_tiles) ((c.tiles[ti].flags & TILE_CATEGORICAL) ? s->cgiant_tiles : s->bgiant_tiles).push_back(ti);
    s->d_giant = upload(s->cgiant_tiles, st);
    s->d_wide = upload(c.wide_tiles, st);
    {
      std::vector<GiantPiece> pieces;
      s->bgiant_piece_off.assign(1, 0u);
      for (uint32_t slot = 0; slot < s->bgiant_tiles.size(); ++slot) {
        const TileDesc &td = c.tiles[s->bgiant_tiles[slot]];
        for (uint32_t e = td.e0; e < td.e0 + td.nedges; e += GIANT_PIECE)
          pieces.push_back(GiantPiece{slot, e, std::min(e + GIANT_PIECE, td.e0 + td.nedges), 0u});
        if (!td.nedges) pieces.push_back(GiantPiece{slot, td.e0, td.e0, 0u});
        s->bgiant_piece_off.push_back((uint32_t)pieces.size());
      }
      s->d_bgiant = upload(s->bgiant_tiles, st);
      s->d_bgiant_piece_off = upload(s->bgiant_piece_off, st);
      s->d_bgiant_pieces = upload(pieces, st);
      s->d_bgiant_partial = (double *)rt::dmalloc(std::max<size_t>(1, pieces.size()) * 4 * 8);
      rt::dmemset(s->d_bgiant_partial, 0, std::max<size_t>(1, pieces.size()) * 4 * 8, st);
      s->d_bgiant_decision = (uint32_t *)rt::dmalloc(std::max<size_t>(1, s->bgiant_tiles.size()) * 16);
    }
    if (!c.row_truth.empty()) s->d_row_truth = upload(c.row_truth, st);
    if (!c.edge_fval64.empty()) s->d_edge_fval64 = upload(c.edge_fval64, st);
    s->d_edges = upload(c.edges, st, 1);
    s->rec8 = c.edges8.size() != 0;
    if (s->rec8) s->d_edges8 = upload(c.edges8, st, 1);
    if (!c.supers.empty()) {
      s->d_supers = upload(c.supers, st);
      if (c.sorted_deferred) {
        // planned on the host, built here: emit + radix sort by (super-tile, weight id) on the device
        s->d_sorted = (SortRec8 *)rt::dmalloc((c.n_sorted + 1) * sizeof(SortRec8));
        rt::dmemset(s->d_sorted + c.n_sorted, 0, sizeof(SortRec8), st);
        devb::build_sorted_records(s->d_tiles, s->d_edges, s->d_edges8, s->d_supers, (uint32_t)c.supers.size(),
                                   c.sort_dbits.data(), (uint32_t)c.sort_dbits.size(), c.n_sorted, s->d_sorted, (void *)st);
      } else {
        s->d_sorted = upload(c.sorted_recs, st, 1);
      }
      s->d_sort_dvals = upload(c.sort_dvals, st);
      s->n_sort_dvals = (uint32_t)c.sort_dvals.size();
      s->lds_sorted = (size_t)SUPER_NV_MAX * 8 + SORT_TV_SLOTS * 4 + (size_t)s->n_sort_dvals * 8;
      s->sorted_learn = c.W > LDS_AGG_MAX_W;     // == their tiles are TILE_PULL
      rt::allow_dynamic_lds(sorted_sweep_kernel<false, false>, s->lds_sorted);
      rt::allow_dynamic_lds(sorted_sweep_kernel<true, false>, s->lds_sorted);
      rt::allow_dynamic_lds(sorted_sweep_kernel<false, true>, s->lds_sorted);
      rt::allow_dynamic_lds(sorted_sweep_kernel<true, true>, s->lds_sorted);
      s->no_sort_uni = getenv("DWX_NO_SORT_UNI") != nullptr;
    }
    {
      // the batched walks load a factor's first entries branch-free (a unary record reads entry
      // 0 and ignores it): the array is never empty, entry 0 names a real variable
      std::vector<VifRec> vifs = c.vifs;
      if (vifs.size() < 4) vifs.resize(4, VifRec{0u, 0u});
      s->d_vifs = upload(vifs, st, 4);
    }
    // InferenceResult init (src/inference_result.cc:24-42): both chains start at the
    // evidence value or 0, tallies zero, weights at their initial values
    std::vector<uint32_t> a0(c.V);
    for (uint64_t p = 0; p < c.V; ++p) a0[p] = (c.v_meta[p] & VM_EVIDENCE) ? c.v_init[p] : 0u;
    s->d_assign_free = upload(a0, st);
    s->d_assign_evid = upload(a0, st);
    s->d_tally = (uint32_t *)rt::dmalloc((c.R + 1) * 4);
    rt::dmemset(s->d_tally, 0, (c.R + 1) * 4, st);
    s->d_weights = upload(c.w_init, st);
    phase("upload graph");
    {
      std::vector<float> w32(c.W);
      for (uint64_t i = 0; i < c.W; ++i) w32[i] = (float)c.w_init[i];
      s->d_w32 = upload(w32, st, 1);
      // (+ one variable block of padding: pull_ell_kernel copies whole blocks)
      const size_t n_delta = c.tiles.size() * 8 + 2 + (size_t)BP_TILES * 8;
      s->d_delta = (unsigned long long *)rt::dmalloc(n_delta * 8);
      rt::dmemset(s->d_delta, 0, n_delta * 8, st);
    }
    s->d_w_fixed = upload(c.w_fixed, st);
    s->d_grad = (long long *)rt::dmalloc(c.W * 16);
    rt::dmemset(s->d_grad, 0, c.W * 16, st);

    KernelParams &P = s->base;
    P.v_meta = s->d_v_meta; P.v_orig = s->d_v_orig; P.v_row = s->d_v_row; P.v_init = s->d_v_init;
    P.row_ptr = s->d_row_ptr; P.row_truth = s->d_row_truth; P.edges = s->d_edges; P.edges8 = s->d_edges8; P.edge_terms = nullptr;
    for (const TileDesc &td : c.tiles) if (td.flags & (TILE_SIMPLE | TILE_INLINE2)) { s->has_simple_tiles = true; break; }
    P.edge_fval64 = s->d_edge_fval64; P.vifs = s->d_vifs; P.tiles = s->d_tiles;
    P.assign_free = s->d_assign_free; P.assign_evid = s->d_assign_evid; P.tally = s->d_tally;
    P.w32 = s->d_w32; P.grad = s->d_grad; P.delta = s->d_delta;
    P.seed = opts->seed; P.sweep = 0; P.vid_offset = opts->var_id_offset; P.tile_begin = 0; P.num_weights = (uint32_t)c.W;
    P.flags = (opts->sample_evidence ? OPT_SAMPLE_EVIDENCE : 0) |
              (opts->learn_non_evidence ? OPT_LEARN_NON_EVIDENCE : 0) |
              (opts->noise_aware ? OPT_NOISE_AWARE : 0) |
              (c.has_f64_fval ? OPT_HAS_F64_FVAL : 0) | (c.has_truthiness ? OPT_HAS_TRUTHINESS : 0);
    P.ecap = c.ecap; P.rcap = c.rcap;
    // dynamic LDS layout: [row pointers | potentials scratch (categorical) | K*256 edge
    // records | K*256 f32 weights (learning kernel only)]; K = records staged per lane
    if (c.ecap > MAX_ECAP) throw std::invalid_argument("tile_edges exceeds the staging capacity");
    s->stage_k = c.ecap <= 3 * BLOCK_THREADS ? 3 : (c.ecap <= 6 * BLOCK_THREADS ? 6 : 12);
    const size_t slots = (size_t)s->stage_k * BLOCK_THREADS;
    s->rp_cat = c.edges8.size() != 0 && c.has_categorical && s->stage_k == 6;
    const size_t nrp = std::max<size_t>(c.rcap + 1, (s->rp_cat ? ROWPTR_UNROLL_CAT : ROWPTR_UNROLL) * BLOCK_THREADS);
    size_t off = (nrp * 4 + 15) & ~(size_t)15;
    P.lds_pot_off = c.has_categorical ? (uint32_t)off : 0u;
    if (c.has_categorical) off += (size_t)c.rcap * 8;
    off = (off + 15) & ~(size_t)15;
    P.lds_edge_off = (uint32_t)off;
    s->wide_learn = c.n_terms2_tiles > 0 && s->stage_k <= 6;
    off += slots * sizeof(EdgeRec);
    P.lds_w_off = (uint32_t)off;     // f32 weights right behind the 16-byte records
    s->lds_bytes[0] = off;
    {
      // the tile classes this graph holds: sweep_kernel's smallest build that contains them
      uint32_t tv = 0;
      for (const TileDesc &td : c.tiles) {
        if (td.flags & TILE_OUTSIDE) continue;
        if (td.flags & TILE_CATEGORICAL) tv |= TV_CATEGORICAL;
        if (td.flags & TILE_TERMS3) tv |= TV_TERMS3;
        else if (td.flags & TILE_TERMS2) tv |= (td.flags & TILE_INLINE2) ? TV_TERMS2_INLINE : TV_TERMS2_VIFS;
        else if (td.flags & TILE_SIMPLE) tv |= TV_SIMPLE;
        else tv |= TV_GENERIC;
      }
      // (TV_PAIR stages 16-byte learning records: the weight id shares its word with two flags)
      s->tv_pair = s->wide_learn && s->stage_k == 6 && (tv & TV_TERMS2_INLINE) && !(tv & ~TV_PAIR) &&
                   c.W <= LR16_WID_MASK && !getenv("DWX_NO_TILE_VARIANTS");
    }
    // learning: 16-byte records + f32 weights (20 B per slot); with TILE_TERMS2 tiles the same region
    // alternatively holds 32-byte LearnRecs (which carry their weight) -- 16-byte ones in the TV_PAIR build
    s->lds_bytes[1] = P.lds_edge_off + slots * (s->wide_learn && !s->tv_pair ? 32 : 20);
    P.lds_agg_off = 0;
    P.n_sweeps = 1;
    if (c.W > 0 && c.W <= LDS_AGG_MAX_W) {
      P.lds_agg_off = (uint32_t)((s->lds_bytes[1] + 15) & ~(size_t)15);
      s->lds_bytes[1] = P.lds_agg_off + (size_t)c.W * 16;
    }
    if (s->lds_bytes[1] > 160 * 1024) throw std::invalid_argument("tile does not fit the 160 KiB LDS");
    auto prepare = [&](auto infer, auto learn) {
      rt::allow_dynamic_lds(infer, s->lds_bytes[0]);
      rt::allow_dynamic_lds(learn, s->lds_bytes[1]);
      s->persistent_blocks[0] = rt::resident_blocks(infer, BLOCK_THREADS, s->lds_bytes[0]);
      s->persistent_blocks[1] = rt::resident_blocks(learn, BLOCK_THREADS, s->lds_bytes[1]);
    };
    if (s->tv_pair) {
      prepare(sweep_kernel<false, 6, false, TV_PAIR>, sweep_kernel<true, 6, true, TV_PAIR>);
    } else if (s->wide_learn) {
      switch (s->stage_k) {
        case 3: prepare(sweep_kernel<false, 3>, sweep_kernel<true, 3, true>); break;
        case 6: prepare(sweep_kernel<false, 6>, sweep_kernel<true, 6, true>); break;
        default: prepare(sweep_kernel<false, 12>, sweep_kernel<true, 12, true>); break;
      }
    } else {
      switch (s->stage_k) {
        case 3: prepare(sweep_kernel<false, 3>, sweep_kernel<true, 3>); break;
        case 6: prepare(sweep_kernel<false, 6>, sweep_kernel<true, 6>); break;
        default: prepare(sweep_kernel<false, 12>, sweep_kernel<true, 12>); break;
      }
    }
    if (s->rec8) {
      s->all_pull = P.lds_agg_off == 0;   // (TILE_PULL implies it; a graph of oversized variables only has no such tile)
      for (const TileDesc &td : c.tiles)
        if (!(td.flags & TILE_OUTSIDE) && !(td.flags & TILE_PULL)) { s->all_pull = false; break; }
      s->lds_tab = P.lds_edge_off + (size_t)s->stage_k * BLOCK_THREADS * 8;
      s->lds_learn_pull = s->lds_tab;
      auto prepare8 = [&](auto infer, auto learn) {
        rt::allow_dynamic_lds(infer, s->lds_bytes[0]);
        rt::allow_dynamic_lds(learn, s->lds_bytes[1]);
        s->persistent_blocks8[0] = rt::resident_blocks(infer, BLOCK_THREADS, s->lds_bytes[0]);
        s->persistent_blocks8[1] = rt::resident_blocks(learn, BLOCK_THREADS, s->lds_bytes[1]);
        s->persistent_blocks_pull = rt::resident_blocks(learn, BLOCK_THREADS, s->lds_learn_pull);
      };
      auto prepare_tab = [&](auto tab) {
        rt::allow_dynamic_lds(tab, s->lds_tab);
        s->persistent_blocks_tab = rt::resident_blocks(tab, BLOCK_THREADS, s->lds_tab);
      };
      if (s->rp_cat) {
        constexpr int RPC = (int)ROWPTR_UNROLL_CAT;
        prepare8(sweep8_kernel<false, 6, false, RPC>, sweep8_kernel<true, 6, false, RPC>);
        prepare_tab(sweep8_kernel<false, 6, true, RPC>);
      } else
      switch (s->stage_k) {
        case 3: prepare8(sweep8_kernel<false, 3>, sweep8_kernel<true, 3>); prepare_tab(sweep8_kernel<false, 3, true>); break;
        case 6: prepare8(sweep8_kernel<false, 6>, sweep8_kernel<true, 6>); prepare_tab(sweep8_kernel<false, 6, true>); break;
        default: prepare8(sweep8_kernel<false, 12>, sweep8_kernel<true, 12>); prepare_tab(sweep8_kernel<false, 12, true>); break;
      }
    }
    if (P.lds_agg_off && s->rec8 && s->cgiant_tiles.empty() && s->bgiant_tiles.empty() && c.wide_tiles.empty() &&
        !getenv("DWX_NO_MERGED_APPLY")) {
      // split learning sweeps, one launch per mini-batch: the update as the next sweep kernel's prologue
      s->merge_lw32_off = (uint32_t)((s->lds_bytes[1] + 15) & ~(size_t)15);
      s->lds_merge = s->merge_lw32_off + (size_t)c.W * 4 + 16;
      s->merge_ok = s->lds_merge <= 160 * 1024;
      if (s->merge_ok) {
        if (s->rp_cat) {
          rt::allow_dynamic_lds(sweep8_merged_kernel<6, (int)ROWPTR_UNROLL_CAT>, s->lds_merge);
          rt::allow_dynamic_lds(sweep8_merged_kernel<6, (int)ROWPTR_UNROLL_CAT, true>, s->lds_merge);
        } else switch (s->stage_k) {
          case 3: rt::allow_dynamic_lds(sweep8_merged_kernel<3, (int)ROWPTR_UNROLL>, s->lds_merge);
                  rt::allow_dynamic_lds(sweep8_merged_kernel<3, (int)ROWPTR_UNROLL, true>, s->lds_merge); break;
          case 6: rt::allow_dynamic_lds(sweep8_merged_kernel<6, (int)ROWPTR_UNROLL>, s->lds_merge);
                  rt::allow_dynamic_lds(sweep8_merged_kernel<6, (int)ROWPTR_UNROLL, true>, s->lds_merge); break;
          default: rt::allow_dynamic_lds(sweep8_merged_kernel<12, (int)ROWPTR_UNROLL>, s->lds_merge);
                   rt::allow_dynamic_lds(sweep8_merged_kernel<12, (int)ROWPTR_UNROLL, true>, s->lds_merge); break;
        }
      }
    }
    if (P.lds_agg_off && c.W <= PERSIST_MAX_W && s->rec8 && s->cgiant_tiles.empty() && s->bgiant_tiles.empty() &&
        c.wide_tiles.empty() && getenv("DWX_PERSIST") && !getenv("DWX_NO_PERSIST")) {
      // a split learning sweep as one persistent launch (opt-in): at most one workgroup per CU, all resident
      s->persist_grid = rt::grid_barrier_blocks();
      if (const char *e = getenv("DWX_PERSIST_WG_PER_CU")) s->persist_grid *= (uint32_t)std::max(1L, std::min(3L, atol(e)));   // (experiment)
      s->persist_row_stride = (uint32_t)((2 * c.W + 15) / 16 * 16);
      s->d_persist_rows = (long long *)rt::dmalloc((size_t)2 * s->persist_grid * s->persist_row_stride * 8);
      s->d_persist_bar = (uint32_t *)rt::dmalloc(16);
      s->persist_w64_off = (uint32_t)((s->lds_bytes[1] + 15) & ~(size_t)15);
      s->persist_red_off = (uint32_t)((s->persist_w64_off + c.W * 12 + 15) & ~(size_t)15);
      s->lds_persist = s->persist_red_off + (size_t)BLOCK_THREADS * 8 + 16;
      if (s->lds_persist > 160 * 1024) s->persist_grid = 0;
      else if (s->rp_cat) rt::allow_dynamic_lds(persist_learn8_kernel<6, (int)ROWPTR_UNROLL_CAT>, s->lds_persist);
      else switch (s->stage_k) {
        case 3: rt::allow_dynamic_lds(persist_learn8_kernel<3, (int)ROWPTR_UNROLL>, s->lds_persist); break;
        case 6: rt::allow_dynamic_lds(persist_learn8_kernel<6, (int)ROWPTR_UNROLL>, s->lds_persist); break;
        default: rt::allow_dynamic_lds(persist_learn8_kernel<12, (int)ROWPTR_UNROLL>, s->lds_persist); break;
      }
    }
    rt::stream_sync(st);
    // the un-split sweep's curvature estimate is needed by the first dwx_sgd_plan: pay
    // for it here (one host pass over the records), not inside the first learning sweep
    {
      // SGD work per tile (records sgd_on_variable visits, src/factor_graph.cc:262-314):
      // what the mini-batch plan balances its cuts on
      std::vector<uint64_t> work(c.tiles.size() + 1, 0);
      parallel_ranges(c.tiles.size(), host_threads(), [&](uint64_t tb, uint64_t te) {
        for (uint64_t ti = tb; ti < te; ++ti) {
          const TileDesc &td = c.tiles[ti];
          uint64_t n = 0;
          for (uint32_t p = td.v0; p < td.v0 + td.nv; ++p) {
            const uint32_t m = c.v_meta[p];
            const bool trig = opts->learn_non_evidence || (!opts->noise_aware && (m & VM_EVIDENCE)) ||
                              (opts->noise_aware && (m & VM_TRUTHINESS));
            if (!trig) continue;
            for (uint32_t e = c.row_ptr[c.v_row[p]]; e < c.row_ptr[c.v_row[p + 1]]; ++e)
              n += !(c.edges[e].packed & EDGE_FIXED_FLAG);
          }
          work[ti + 1] = n;
        }
      }, 64);
      for (size_t i = 0; i < c.tiles.size(); ++i) work[i + 1] += work[i];
      s->sgd_work.swap(work);
      for (size_t l = 0; l + 1 < c.launch_off.size(); ++l)
        s->sgd_work_max_launch = std::max(s->sgd_work_max_launch,
                                          s->sgd_work[c.launch_tile[l + 1]] - s->sgd_work[c.launch_tile[l]]);
    }
    phase("kernel setup");
    (void)build_level(s.get(), 1);
    phase("static counts + gradient incidence list");
    if (opts->step_cap > 0) (void)row_sum_bound(s.get(), 1);
    phase("curvature estimate");
    devb::release_scratch(s->device, true);   // (the builds' cached temporaries: kept unless memory is tight)
    rt::stream_sync(st);
    rt::stage_trim();                          // (the uploads' pinned staging chunks)
  });
  if (rc != DWX_OK) return rc;
  *out = s.release();
  return DWX_OK;
}

int dwx_device_count(int32_t *count) {
  if (!count) return fail(DWX_E_INVALID, "null argument");
  return guarded([&]() { *count = rt::device_count(); });
}

int dwx_buffer_copy(dwx_sampler *s, void *dst, const void *src, uint64_t nbytes, int to_device) {
  if (!s || (nbytes && (!dst || !src))) return fail(DWX_E_INVALID, "null argument");
  return guarded([&]() {
    rt::set_device(s->device);
    if (to_device) rt::h2d(dst, src, nbytes, s->stream);
    else rt::d2h(dst, src, nbytes, s->stream);
    rt::stream_sync(s->stream);
  });
}

int dwx_device_init(int32_t device) {
  return guarded([&]() {
    rt::init_device(device);
    rt::oom_hook() = [](int dev) { devb::release_scratch(dev, false); };
    rt::dfree(rt::dmalloc(16));   // forces the context
  });
}

void dwx_sampler_destroy(dwx_sampler *s) {
  if (!s) return;
  try {
    rt::set_device(s->device);
    rt::stream_sync(s->stream);
  } catch (...) {
  }
  const int device = s->device;
  delete s;
  devb::release_scratch(device, false);
}


int dwx_sample_async(dwx_sampler *s) {
  if (!s) return fail(DWX_E_INVALID, "null sampler");
  return guarded([&]() {
    enqueue_inference(s);
    ++s->infer_sweeps;
  });
}

int dwx_sample_n_async(dwx_sampler *s, uint32_t n_sweeps) {
  if (!s) return fail(DWX_E_INVALID, "null sampler");
  return guarded([&]() {
    if (n_sweeps > 1 && multi_sweep_graph(s)) {
      enqueue_inference_multi(s, n_sweeps);
    } else {
      for (uint32_t k = 0; k < n_sweeps; ++k) enqueue_inference(s);
    }
    s->infer_sweeps += n_sweeps;
  });
}

int dwx_sgd_plan(dwx_sampler *s, double stepsize, uint32_t force_batches, uint32_t *batches,
                 uint32_t *n_chunks, double *effective_stepsize) {
  if (!s) return fail(DWX_E_INVALID, "null sampler");
  return guarded([&]() {
    make_plan(s, stepsize, force_batches);
    if (batches) *batches = s->plan_batches;
    if (n_chunks) *n_chunks = (uint32_t)s->plan_chunks.size();
    if (effective_stepsize) *effective_stepsize = plan_min_step(s);
  });
}

int dwx_sgd_curvature(dwx_sampler *s, uint32_t batches, double *lambda) {
  if (!s || !lambda) return fail(DWX_E_INVALID, "null argument");
  if (batches == 0) return fail(DWX_E_INVALID, "batches must be >= 1");
  return guarded([&]() { *lambda = row_sum_bound(s, batches); });
}

int dwx_sgd_plan_rows(dwx_sampler *s, uint32_t n_rows) {
  if (!s) return fail(DWX_E_INVALID, "null sampler");
  if (!s->plan_valid || !s->plan_level) return fail(DWX_E_INVALID, "no plan: call dwx_sgd_plan first");
  return guarded([&]() {
    dwx_sampler::Level &L = *s->plan_level;
    const uint64_t W = s->cg->W;
    if (!L.fast || s->plan_batches <= 1 || n_rows <= L.rows || !W) return;
    rt::set_device(s->device);
    long long *grown = (long long *)rt::dmalloc((size_t)n_rows * 2 * W * 8);
    rt::dmemset(grown, 0, (size_t)n_rows * 2 * W * 8, s->stream);
    rt::d2d(grown, L.d_t_static, (size_t)L.rows * 2 * W * 8, s->stream);
    rt::stream_sync(s->stream);
    rt::dfree(L.d_t_static);
    L.d_t_static = grown;
    L.rows = n_rows;
  });
}

int dwx_sgd_plan_force_dynamic(dwx_sampler *s, int on) {
  if (!s) return fail(DWX_E_INVALID, "null sampler");
  if (!s->plan_valid || !s->plan_level) return fail(DWX_E_INVALID, "no plan: call dwx_sgd_plan first");
  s->plan_force_dynamic = on != 0;
  return DWX_OK;
}

int dwx_sgd_get_chunks(dwx_sampler *s, uint64_t *chunk_range) {
  if (!s || !chunk_range) return fail(DWX_E_INVALID, "null argument");
  if (!s->plan_valid) return fail(DWX_E_INVALID, "no plan: call dwx_sgd_plan first");
  const CompiledGraph &c = *s->cg;
  size_t i = 0;
  for (const auto &ch : s->plan_chunks) { chunk_range[i++] = c.tile_v[ch.t0]; chunk_range[i++] = c.tile_v[ch.t1]; }
  return DWX_OK;
}

int dwx_sgd_accumulate_async(dwx_sampler *s, uint32_t chunk) {
  if (!s) return fail(DWX_E_INVALID, "null sampler");
  if (!s->plan_valid) return fail(DWX_E_INVALID, "no plan: call dwx_sgd_plan first");
  return guarded([&]() {
    s->cur_chunk = chunk;
    enqueue_learn_chunk(s, chunk);
  });
}

int dwx_sgd_apply_async(dwx_sampler *s) {
  if (!s) return fail(DWX_E_INVALID, "null sampler");
  if (!s->plan_valid) return fail(DWX_E_INVALID, "no plan: call dwx_sgd_plan first");
  return guarded([&]() { enqueue_apply(s); });
}

int dwx_sgd_finish(dwx_sampler *s) {
  if (!s) return fail(DWX_E_INVALID, "null sampler");
  ++s->sweep;
  s->plan_valid = false;
  return DWX_OK;
}

int dwx_sample_sgd_async(dwx_sampler *s, double stepsize) {
  if (!s) return fail(DWX_E_INVALID, "null sampler");
  return guarded([&]() {
    make_plan(s, stepsize, 0);
    const size_t n = s->plan_chunks.size();
    auto enqueue_sweep = [&]() {
      for (size_t c = 0; c < n; ++c) {
        s->cur_chunk = (uint32_t)c;
        enqueue_learn_chunk(s, (uint32_t)c);
        if (s->plan_batches > 1 || c + 1 == n) enqueue_apply(s);
      }
    };
    // Graph replay (DWX_GRAPH=n in the environment: sweeps of at least n mini-batches; default
    // off).  A split sweep is a chain of 2-4 short dependent launches per mini-batch; from the
    // level's second sweep on (the first one has done every lazy allocation) it can be captured,
    // patched into the level's instantiated graph -- only the sweep counter and the step differ
    // from sweep to sweep -- and launched as ONE graph.  Measured on config 4's learning sweep
    // (64 mini-batches, 128 launches): 1.610 -> 1.603 ms, i.e. nothing: the chain is bound by the
    // latency of the chunk kernels themselves (17.5-22 us each for 3 us of work at full width,
    // profiles/r03/kernel_stats_cfg4learn.csv), not by the host's launches -- hence off.  Never
    // while kernels are being timed (the events would land inside the graph), nor on the sweep
    // that builds the level's own layout (host work and uploads).
    dwx_sampler::Level *L = s->plan_level;
    const char *genv = getenv("DWX_GRAPH");
    const int graph_min_chunks = genv ? atoi(genv) : 0;
    const bool replay = L && graph_min_chunks > 0 && s->plan_batches > 1 && n >= (size_t)graph_min_chunks &&
                        !s->timing && L->use_graph >= 0 && L->sweeps >= 1 &&
                        !(s->opts.plan_layouts == 0 && L->sweeps + 1 == layout_after_sweeps());
    if (enqueue_persistent_sweep(s)) {
      // (few weights, many mini-batches: the whole split sweep is one persistent launch; opt-in)
    } else if (!replay && enqueue_merged_sweep(s)) {
      // (few weights, split sweep: one launch per mini-batch, the update as the next one's prologue)
    } else if (!replay) {
      enqueue_sweep();
    } else {
      rt::set_device(s->device);
      rt::graph_t g = nullptr;
      rt::capture_begin(s->stream);
      try {
        enqueue_sweep();
        g = rt::capture_end(s->stream);
      } catch (...) {
        rt::capture_abandon(s->stream);
        L->use_graph = -1;
        throw;
      }
      bool launched = false;
      try {
        if (L->gexec && !rt::graph_exec_update(L->gexec, g)) { rt::graph_exec_destroy(L->gexec); L->gexec = nullptr; }
        if (!L->gexec) L->gexec = rt::graph_instantiate(g);
        rt::graph_launch(L->gexec, s->stream);
        launched = true;
        L->use_graph = 1;
        ++s->graph_launches;
      } catch (...) {
        // nothing of the captured sweep has run: this level goes back to plain launches for good
        L->use_graph = -1;
        rt::graph_exec_destroy(L->gexec);
        L->gexec = nullptr;
      }
      rt::graph_destroy(g);
      if (!launched) { --L->sweeps; enqueue_sweep(); }
    }
    ++s->sweep;
    s->plan_valid = false;
  });
}

int dwx_grad_pack_async(dwx_sampler *s, uint32_t shift, uint32_t bits, void **dev32, uint64_t *n_words) {
  if (!s || !dev32 || !n_words) return fail(DWX_E_INVALID, "null argument");
  if (shift == 0 || shift > 62) return fail(DWX_E_INVALID, "gradient shift out of range (dwx_graph_info.grad_shift)");
  if (bits != 32 && bits != 16) return fail(DWX_E_INVALID, "gradient counts travel as 32- or 16-bit integers");
  return guarded([&]() {
    rt::set_device(s->device);
    const uint32_t W = (uint32_t)s->cg->W;
    if (!s->d_grad32) {
      s->d_grad32 = (int *)rt::dmalloc((size_t)W * 4 + 4);
      s->d_pack_bad = (uint32_t *)rt::dmalloc(4);
      rt::dmemset(s->d_pack_bad, 0, 4, s->stream);
    }
    const uint32_t words = bits == 32 ? W : (W + 1u) / 2u;
    const unsigned grid = std::max(1u, std::min((words + BLOCK_THREADS - 1) / BLOCK_THREADS, 1024u));
    if (bits == 32)
      rt::launch(grad_pack32_kernel, grid, BLOCK_THREADS, 0, s->stream, (const long long *)s->d_grad, s->d_grad32, W, shift,
                 s->d_pack_bad);
    else
      rt::launch(grad_pack16_kernel, grid, BLOCK_THREADS, 0, s->stream, (const long long *)s->d_grad, s->d_grad32, W, shift,
                 s->d_pack_bad);
    s->pack_check_pending = true;
    *dev32 = s->d_grad32;
    *n_words = words;
  });
}

int dwx_grad_unpack_async(dwx_sampler *s, uint32_t shift, uint32_t bits) {
  if (!s) return fail(DWX_E_INVALID, "null sampler");
  if (!s->d_grad32) return fail(DWX_E_INVALID, "dwx_grad_unpack_async without dwx_grad_pack_async");
  if (bits != 32 && bits != 16) return fail(DWX_E_INVALID, "gradient counts travel as 32- or 16-bit integers");
  return guarded([&]() {
    rt::set_device(s->device);
    const uint32_t W = (uint32_t)s->cg->W;
    const uint32_t words = bits == 32 ? W : (W + 1u) / 2u;
    const unsigned grid = std::max(1u, std::min((words + BLOCK_THREADS - 1) / BLOCK_THREADS, 1024u));
    if (bits == 32)
      rt::launch(grad_unpack32_kernel, grid, BLOCK_THREADS, 0, s->stream, (const int *)s->d_grad32, s->d_grad, W, shift);
    else
      rt::launch(grad_unpack16_kernel, grid, BLOCK_THREADS, 0, s->stream, (const int *)s->d_grad32, s->d_grad, W, shift);
  });
}

int dwx_grad_pack32_async(dwx_sampler *s, uint32_t shift, void **dev32, uint64_t *n) {
  return dwx_grad_pack_async(s, shift, 32, dev32, n);
}
int dwx_grad_unpack32_async(dwx_sampler *s, uint32_t shift) { return dwx_grad_unpack_async(s, shift, 32); }

int dwx_wait(dwx_sampler *s) {
  if (!s) return fail(DWX_E_INVALID, "null sampler");
  return guarded([&]() {
    rt::set_device(s->device);
    rt::stream_sync(s->stream);
    if (s->persist_check_pending) {   // (a persistent split sweep whose workgroups were not all resident gave up)
      uint32_t bar[4] = {0, 0, 0, 0};
      rt::d2h(bar, s->d_persist_bar, 16, s->stream);
      rt::stream_sync(s->stream);
      s->persist_check_pending = false;
      if (bar[1]) {
        s->persist_grid = 0;          // (never again on this sampler)
        throw std::runtime_error("a persistent learning sweep gave up at its grid barrier (its workgroups were not all "
                                 "resident): the weights and chains of that sweep are invalid; DWX_NO_PERSIST=1 disables it");
      }
    }
    if (s->pack_check_pending) {      // (the 32-bit gradient counts: exact by construction, verified here)
      uint32_t bad = 0;
      rt::d2h(&bad, s->d_pack_bad, 4, s->stream);
      rt::stream_sync(s->stream);
      s->pack_check_pending = false;
      if (bad) throw std::runtime_error("dwx_grad_pack_async: " + std::to_string(bad) +
                                        " gradient sums were not multiples of 2^shift or did not fit the count width");
    }
  });
}

int dwx_get_weights(dwx_sampler *s, double *out) {
  if (!s || !out) return fail(DWX_E_INVALID, "null argument");
  return guarded([&]() {
    rt::set_device(s->device);
    rt::d2h(out, s->d_weights, s->cg->W * 8, s->stream);
    rt::stream_sync(s->stream);
  });
}

int dwx_set_weights(dwx_sampler *s, const double *in) {
  if (!s || !in) return fail(DWX_E_INVALID, "null argument");
  return guarded([&]() {
    rt::set_device(s->device);
    const uint32_t W = (uint32_t)s->cg->W;
    s->terms_state = 0;
    rt::h2d(s->d_weights, in, (size_t)W * 8, s->stream);
    if (W) {
      const unsigned grid = std::min<unsigned>((W + BLOCK_THREADS - 1) / BLOCK_THREADS, 2048u);
      rt::launch(refresh_w32_kernel, grid, BLOCK_THREADS, 0, s->stream,
                 (const double *)s->d_weights, s->d_w32, W);
    }
    rt::stream_sync(s->stream);
  });
}

int dwx_average_weights_async(dwx_sampler *s, uint32_t n_replicas) {
  if (!s) return fail(DWX_E_INVALID, "null sampler");
  if (n_replicas == 0) return fail(DWX_E_INVALID, "n_replicas must be >= 1");
  return guarded([&]() {
    rt::set_device(s->device);
    const uint32_t W = (uint32_t)s->cg->W;
    if (!W) return;
    s->terms_state = 0;
    if (!s->d_w_init) {
      s->d_w_init = upload(s->cg->w_init, s->stream);
      rt::stream_sync(s->stream);
    }
    const unsigned grid = std::min<unsigned>((W + BLOCK_THREADS - 1) / BLOCK_THREADS, 2048u);
    rt::launch(average_weights_kernel, grid, BLOCK_THREADS, 0, s->stream, s->d_weights, s->d_w32,
               (const uint8_t *)s->d_w_fixed, (const double *)s->d_w_init, W, (double)n_replicas);
  });
}

int dwx_clear_tallies(dwx_sampler *s) {
  if (!s) return fail(DWX_E_INVALID, "null sampler");
  return guarded([&]() {
    rt::set_device(s->device);
    rt::dmemset(s->d_tally, 0, s->cg->R * 4, s->stream);
    s->infer_sweeps = 0;
  });
}

int dwx_get_tallies(dwx_sampler *s, uint64_t *tallies, uint64_t *nsamples) {
  if (!s) return fail(DWX_E_INVALID, "null sampler");
  return guarded([&]() {
    const CompiledGraph &c = *s->cg;
    rt::set_device(s->device);
    if (tallies) {
      RawArray<uint32_t> t(c.R);
      rt::d2h(t.data(), s->d_tally, c.R * 4, s->stream);
      rt::stream_sync(s->stream);
      parallel_ranges(c.Vo, host_threads(), [&](uint64_t pb, uint64_t pe) {
        for (uint64_t p = pb; p < pe; ++p) {
          const uint64_t rb = c.ref_var_val_base[c.perm[p]];
          for (uint32_t r = c.v_row[p]; r < c.v_row[p + 1]; ++r) tallies[rb + (r - c.v_row[p])] = t[r];
        }
      });
    }
    if (nsamples)  // agg_nsamples: +1 per inference sweep for every sampled variable
      parallel_ranges(c.V, host_threads(), [&](uint64_t vb, uint64_t ve) {
        for (uint64_t v = vb; v < ve; ++v)
          nsamples[v] = (v < c.Vo && (!c.var_is_evid[v] || s->opts.sample_evidence)) ? s->infer_sweeps : 0;
      });
  });
}

int dwx_get_assignments(dwx_sampler *s, int chain, uint64_t *out) {
  if (!s || !out) return fail(DWX_E_INVALID, "null argument");
  return guarded([&]() {
    const CompiledGraph &c = *s->cg;
    rt::set_device(s->device);
    std::vector<uint32_t> a(c.V);
    rt::d2h(a.data(), chain == 0 ? s->d_assign_free : s->d_assign_evid, c.V * 4, s->stream);
    rt::stream_sync(s->stream);
    for (uint64_t p = 0; p < c.V; ++p) out[c.perm[p]] = a[p];
  });
}

int dwx_set_assignments(dwx_sampler *s, int chain, const uint64_t *in) {
  if (!s || !in) return fail(DWX_E_INVALID, "null argument");
  return guarded([&]() {
    const CompiledGraph &c = *s->cg;
    rt::set_device(s->device);
    std::vector<uint32_t> a(c.V);
    for (uint64_t p = 0; p < c.V; ++p) a[p] = (uint32_t)in[c.perm[p]];
    rt::h2d(chain == 0 ? s->d_assign_free : s->d_assign_evid, a.data(), c.V * 4, s->stream);
    rt::stream_sync(s->stream);
  });
}

int dwx_get_sweep(dwx_sampler *s, uint64_t *out) {
  if (!s || !out) return fail(DWX_E_INVALID, "null argument");
  *out = s->sweep;
  return DWX_OK;
}

int dwx_set_sweep(dwx_sampler *s, uint64_t sweep) {
  if (!s) return fail(DWX_E_INVALID, "null sampler");
  s->sweep = sweep;
  return DWX_OK;
}

int dwx_device_buffer(dwx_sampler *s, int which, void **dev_ptr, uint64_t *nbytes) {
  if (!s || !dev_ptr || !nbytes) return fail(DWX_E_INVALID, "null argument");
  const CompiledGraph &c = *s->cg;
  switch (which) {
    case DWX_BUF_WEIGHTS: *dev_ptr = s->d_weights; *nbytes = c.W * 8; break;
    case DWX_BUF_GRAD: *dev_ptr = s->d_grad; *nbytes = c.W * 16; break;
    case DWX_BUF_ASSIGN_FREE: *dev_ptr = s->d_assign_free; *nbytes = c.V * 4; break;
    case DWX_BUF_ASSIGN_EVID: *dev_ptr = s->d_assign_evid; *nbytes = c.V * 4; break;
    case DWX_BUF_TALLIES: *dev_ptr = s->d_tally; *nbytes = c.R * 4; break;
    case DWX_BUF_TSTATIC: {
      auto it = s->levels.find(1);
      *dev_ptr = it == s->levels.end() ? nullptr : it->second->d_t_static;
      *nbytes = c.W * 16;
      break;
    }
    case DWX_BUF_TSTATIC_PLAN:
      if (!s->plan_valid || !s->plan_level) return fail(DWX_E_INVALID, "no plan: call dwx_sgd_plan first");
      {
        const bool tables = s->plan_level->fast && !(s->plan_batches > 1 && s->plan_force_dynamic);
        *dev_ptr = tables ? s->plan_level->d_t_static : nullptr;
        *nbytes = tables ? (uint64_t)s->plan_level->rows * c.W * 16 : 0;
      }
      break;
    case DWX_BUF_SORTED_RECORDS:     // (test hook: the device-built copy against the host builder's)
      *dev_ptr = s->d_sorted; *nbytes = s->d_sorted ? c.n_sorted * sizeof(SortRec8) : 0;
      break;
    case DWX_BUF_SORTED_RECORDS_PLAN:
      if (!s->plan_level) return fail(DWX_E_INVALID, "no plan: call dwx_sgd_plan first");
      *dev_ptr = s->plan_level->d_sorted;
      {
        uint64_t n = 0;
        for (const SuperTile &st : s->plan_level->sorted_supers) n += st.nrec;
        *nbytes = s->plan_level->d_sorted ? n * sizeof(SortRec8) : 0;
      }
      break;
    default: return fail(DWX_E_INVALID, "unknown buffer id");
  }
  return DWX_OK;
}

int dwx_halo_create(dwx_sampler *s, const uint64_t *vids, uint64_t n, dwx_halo **out) {
  if (!s || !out || (n && !vids)) return fail(DWX_E_INVALID, "null argument");
  if (n >= 0xFFFFFFFFull) return fail(DWX_E_LIMIT, "halo list exceeds 2^32-1 entries");
  std::unique_ptr<dwx_halo> h(new dwx_halo());
  int rc = guarded([&]() {
    const CompiledGraph &c = *s->cg;
    rt::set_device(s->device);
    std::vector<uint32_t> pos(n);
    uint32_t max_card = 0;
    for (uint64_t i = 0; i < n; ++i) {
      if (vids[i] >= c.V) throw std::invalid_argument("halo list: variable id out of range");
      pos[i] = c.pos[vids[i]];
      max_card = std::max(max_card, c.v_meta[pos[i]] >> VM_CARD_SHIFT);
    }
    h->s = s; h->n = (uint32_t)n;
    // (both ends list the same variables, so they choose the same width)
    h->bits = max_card <= 2 ? 1u : (max_card <= 256 ? 8u : 32u);
    h->block = (uint32_t)(((uint64_t)n * h->bits + 63) / 64);
    h->d_pos = upload(pos, s->stream);
    const size_t bytes = std::max<size_t>((size_t)2 * h->block * 8, 8);
    h->d_buf = (unsigned long long *)rt::dmalloc(bytes);
    rt::dmemset(h->d_buf, 0, bytes, s->stream);
    rt::stream_sync(s->stream);
  });
  if (rc != DWX_OK) return rc;
  *out = h.release();
  return DWX_OK;
}

void dwx_halo_destroy(dwx_halo *h) { delete h; }

int dwx_halo_buffer(dwx_halo *h, void **dev_ptr, uint64_t *nbytes) {
  if (!h || !dev_ptr || !nbytes) return fail(DWX_E_INVALID, "null argument");
  *dev_ptr = h->d_buf;
  *nbytes = (uint64_t)2 * h->block * 8;
  return DWX_OK;
}

int dwx_halo_message_bytes(dwx_halo *h, int chains, uint64_t *nbytes) {
  if (!h || !nbytes) return fail(DWX_E_INVALID, "null argument");
  if (chains < 1 || chains > 3) return fail(DWX_E_INVALID, "chains: bit 0 = free chain, bit 1 = evidence chain");
  *nbytes = (uint64_t)(chains == 3 ? 2 : 1) * h->block * 8;
  return DWX_OK;
}

namespace {
int halo_move(dwx_halo *h, int chains, bool pack) {
  if (!h) return fail(DWX_E_INVALID, "null halo list");
  if (chains < 1 || chains > 3) return fail(DWX_E_INVALID, "chains: bit 0 = free chain, bit 1 = evidence chain");
  if (!h->n) return DWX_OK;
  return guarded([&]() {
    dwx_sampler *s = h->s;
    rt::set_device(s->device);
    const unsigned grid = std::min<unsigned>((h->n + BLOCK_THREADS - 1) / BLOCK_THREADS, 4096u);
    auto go = [&](auto pk, auto unpk) {
      if (pack)
        rt::launch(pk, grid, BLOCK_THREADS, 0, s->stream, (const uint32_t *)h->d_pos, h->n,
                   (const uint32_t *)s->d_assign_free, (const uint32_t *)s->d_assign_evid, (uint32_t)chains,
                   h->d_buf, h->block);
      else
        rt::launch(unpk, grid, BLOCK_THREADS, 0, s->stream, (const uint32_t *)h->d_pos, h->n,
                   s->d_assign_free, s->d_assign_evid, (uint32_t)chains, (const unsigned long long *)h->d_buf, h->block);
    };
    if (h->bits == 1) go(halo_pack_kernel<1>, halo_unpack_kernel<1>);
    else if (h->bits == 8) go(halo_pack_kernel<8>, halo_unpack_kernel<8>);
    else go(halo_pack_kernel<32>, halo_unpack_kernel<32>);
  });
}
}  // namespace

int dwx_halo_pack_async(dwx_halo *h, int chains) { return halo_move(h, chains, true); }
int dwx_halo_unpack_async(dwx_halo *h, int chains) { return halo_move(h, chains, false); }

int dwx_stream(dwx_sampler *s, void **stream) {
  if (!s || !stream) return fail(DWX_E_INVALID, "null argument");
  *stream = (void *)s->stream;
  return DWX_OK;
}

int dwx_kernel_time_reset(dwx_sampler *s, int enable) {
  if (!s) return fail(DWX_E_INVALID, "null sampler");
  return guarded([&]() {
    rt::set_device(s->device);
    drain_spans(s);
    for (int k = 0; k < 3; ++k) { s->t_ms[k] = 0; s->t_launches[k] = 0; s->t_sweeps[k] = 0; }
    s->timing = enable != 0;
  });
}

int dwx_kernel_time(dwx_sampler *s, int kind, double *ms, uint64_t *launches, uint64_t *sweeps) {
  if (!s || kind < 0 || kind > 5) return fail(DWX_E_INVALID, "bad argument");
  if (kind == 5) {   // split learning sweeps run with one (merged) launch per mini-batch since the sampler was created
    if (ms) *ms = 0.0;
    if (launches) *launches = s->merged_sweeps;
    if (sweeps) *sweeps = s->merged_sweeps;
    return DWX_OK;
  }
  if (kind == 4) {   // split learning sweeps run as ONE persistent launch since the sampler was created (no device time)
    if (ms) *ms = 0.0;
    if (launches) *launches = s->persist_launches;
    if (sweeps) *sweeps = s->persist_launches;
    return DWX_OK;
  }
  if (kind == 3) {   // graph replays of split learning sweeps since the sampler was created (no device time)
    if (ms) *ms = 0.0;
    if (launches) *launches = s->graph_launches;
    if (sweeps) *sweeps = s->graph_launches;
    return DWX_OK;
  }
  return guarded([&]() {
    rt::set_device(s->device);
    drain_spans(s);
    if (ms) *ms = s->t_ms[kind];
    if (launches) *launches = s->t_launches[kind];
    if (sweeps) *sweeps = s->t_sweeps[kind];
  });
}

int dwx_test_philox(int device, const uint32_t key[2], const uint32_t ctr[4], uint32_t out[4], double uniforms[2]) {
  if (!key || !ctr || !out || !uniforms) return fail(DWX_E_INVALID, "null argument");
  return guarded([&]() {
    rt::init_device(device);
    rt::stream_t st = rt::stream_create();
    uint32_t *dout = (uint32_t *)rt::dmalloc(8 * 4);
    double *duni = (double *)rt::dmalloc(16);
    rt::h2d(dout + 4, ctr, 16, st);     // (the counter again, for the uniforms' entry point)
    rt::launch(test_philox_kernel, 1, 64, 0, st, key[0], key[1], ctr[0], ctr[1], ctr[2], ctr[3], dout, duni);
    rt::d2h(out, dout, 16, st);
    rt::d2h(uniforms, duni, 16, st);
    rt::stream_sync(st);
    rt::dfree(dout); rt::dfree(duni);
    rt::stream_destroy(st);
  });
}

int dwx_test_factor_sign(int device, int func, uint64_t arity, const uint8_t *sat, double *out) {
  if (!sat || !out || arity == 0 || arity > 4096) return fail(DWX_E_INVALID, "bad argument");
  return guarded([&]() {
    rt::init_device(device);
    rt::stream_t st = rt::stream_create();
    std::vector<VifRec> vifs(arity);
    std::vector<uint32_t> assign(arity);
    for (uint64_t i = 0; i < arity; ++i) { vifs[i] = VifRec{(uint32_t)i, 1u}; assign[i] = sat[i]; }
    VifRec *dv = upload(vifs, st);
    uint32_t *da = upload(assign, st);
    double *dout = (double *)rt::dmalloc(8);
    rt::launch(test_sign_kernel, 1, 64, 0, st, (uint32_t)func, (uint32_t)arity,
               (const VifRec *)dv, (const uint32_t *)da, dout);
    rt::d2h(out, dout, 8, st);
    rt::stream_sync(st);
    rt::dfree(dv); rt::dfree(da); rt::dfree(dout);
    rt::stream_destroy(st);
  });
}

}  // extern "C"
