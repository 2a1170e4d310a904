// dw_multi.cc -- see dw_multi.h.  Plain C++17 above the C ABI; the only device library it
// touches itself is the communicator's (RCCL, in dw_rccl.cc).
#include "dw_multi.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <map>
#include <stdexcept>
#include <thread>

namespace dw {

// ------------------------------------------------------------------ host agreement
void HostAgree::barrier() {
  std::unique_lock<std::mutex> lk(m_);
  if (aborted_) throw std::runtime_error("another rank failed");
  const uint64_t gen = gen_;
  if (++waiting_ == n_) {
    waiting_ = 0;
    ++gen_;
    cv_.notify_all();
    return;
  }
  cv_.wait(lk, [&]() { return gen_ != gen || aborted_; });
  if (aborted_) throw std::runtime_error("another rank failed");
}
void HostAgree::abort() {
  std::lock_guard<std::mutex> lk(m_);
  aborted_ = true;
  cv_.notify_all();
}
double HostAgree::max_f64(int rank, double v) {
  d_[rank] = v;
  barrier();
  const double r = *std::max_element(d_.begin(), d_.end());
  barrier();   // nobody overwrites d_ before everybody has read it
  return r;
}
uint64_t HostAgree::max_u64(int rank, uint64_t v) {
  u_[rank] = v;
  barrier();
  const uint64_t r = *std::max_element(u_.begin(), u_.end());
  barrier();
  return r;
}

namespace {
struct Check {
  void operator()(int rc) const {
    if (rc != DWX_OK) throw std::runtime_error(std::string("dwx: ") + dwx_last_error());
  }
};
double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ------------------------------------------------------------------ HostComm (tests)
// Host-staged collectives: every rank copies its buffer to the host (dwx_buffer_copy waits for
// the rank's stream first), the ranks meet at a barrier, every rank sums / picks what it needs
// and copies back.  Correct for any placement of the ranks, including all on one device.
class HostComm : public Comm {
 public:
  HostComm(int n, HostAgree *agree) : n_(n), agree_(agree), stage_(n), mail_(n) {}
  const char *name() const override { return "host-staged (test stand-in)"; }
  void allreduce_sum_i64(int rank, dwx_sampler *s, void *dev, uint64_t n) override { reduce<int64_t>(rank, s, dev, n); }
  void allreduce_sum_f64(int rank, dwx_sampler *s, void *dev, uint64_t n) override { reduce<double>(rank, s, dev, n); }
  void allreduce_sum_u32(int rank, dwx_sampler *s, void *dev, uint64_t n) override { reduce<uint32_t>(rank, s, dev, n); }
  void exchange(int rank, dwx_sampler *s, const std::vector<Xfer> &sends, const std::vector<Xfer> &recvs) override {
    Check ok;
    auto &out = mail_[rank];
    out.clear();
    for (const Xfer &x : sends) {
      std::vector<uint8_t> &b = out[x.peer];
      b.resize(x.nbytes);
      ok(dwx_buffer_copy(s, b.data(), x.dev, x.nbytes, 0));
    }
    agree_->barrier();
    for (const Xfer &x : recvs) {
      const std::vector<uint8_t> &b = mail_[x.peer].at(rank);
      if (b.size() != x.nbytes) throw std::runtime_error("halo exchange: send and receive sizes differ");
      ok(dwx_buffer_copy(s, x.dev, b.data(), x.nbytes, 1));
    }
    agree_->barrier();
  }

 private:
  template <class T>
  void reduce(int rank, dwx_sampler *s, void *dev, uint64_t n) {
    Check ok;
    std::vector<uint8_t> &mine = stage_[rank];
    mine.resize(n * sizeof(T));
    ok(dwx_buffer_copy(s, mine.data(), dev, n * sizeof(T), 0));
    agree_->barrier();
    // every rank adds the ranks' contributions in the same order (0, 1, ...): the f64 sums are
    // bit-identical everywhere, as an all-reduce's are
    std::vector<T> sum(n);
    std::memcpy(sum.data(), stage_[0].data(), n * sizeof(T));
    for (int r = 1; r < n_; ++r) {
      const T *p = (const T *)stage_[r].data();
      for (uint64_t i = 0; i < n; ++i) sum[i] += p[i];
    }
    agree_->barrier();
    ok(dwx_buffer_copy(s, dev, sum.data(), n * sizeof(T), 1));
  }
  int n_;
  HostAgree *agree_;
  std::vector<std::vector<uint8_t>> stage_;
  std::vector<std::map<int, std::vector<uint8_t>>> mail_;   // [sender][receiver]
};
}  // namespace

std::unique_ptr<Comm> make_host_comm(int n_ranks, HostAgree *agree) {
  return std::unique_ptr<Comm>(new HostComm(n_ranks, agree));
}

// ------------------------------------------------------------------ shards
void shard_range(uint64_t total, int rank, int world, uint64_t &begin, uint64_t &end) {
  const uint64_t per = (total + world - 1) / world;
  begin = std::min(total, per * (uint64_t)rank);
  end = std::min(total, begin + per);
}

dwx_graph_desc ShardGraph::desc() const {
  dwx_graph_desc d = g.desc();
  d.num_ghost_variables = n_ghost;
  return d;
}

// the variable side of a shard (out.begin / end / ghosts are set): owned variables first, then
// the ghosts; the domain blocks of the variables present, renumbered; the weights (global)
void fill_shard_variables(const LoadedGraph &w, ShardGraph &out) {
  const uint64_t begin = out.begin, end = out.end, n_owned = end - begin;
  const std::vector<uint64_t> &gh = out.ghosts;
  auto owned = [&](uint64_t v) { return v >= begin && v < end; };
  auto local_id = [&](uint64_t v) -> uint64_t {
    if (owned(v)) return v - begin;
    return n_owned + (uint64_t)(std::lower_bound(gh.begin(), gh.end(), v) - gh.begin());
  };
  LoadedGraph &g = out.g;
  const uint64_t nv = n_owned + gh.size();
  g.n_variables = nv; g.n_weights = w.n_weights;
  g.var_role.resize(nv); g.var_init_value.resize(nv); g.var_dtype.resize(nv); g.var_cardinality.resize(nv);
  g.n_evidence = g.n_query = 0;
  for (uint64_t i = 0; i < nv; ++i) {
    const uint64_t v = i < n_owned ? begin + i : gh[i - n_owned];
    g.var_role[i] = w.var_role[v]; g.var_init_value[i] = w.var_init_value[v];
    g.var_dtype[i] = w.var_dtype[v]; g.var_cardinality[i] = w.var_cardinality[v];
    if (i < n_owned) (w.var_role[v] >= 1 ? g.n_evidence : g.n_query)++;
  }
  g.dom_vid.clear(); g.dom_offset.assign(1, 0); g.dom_value.clear(); g.dom_truthiness.clear();
  for (size_t b = 0; b < w.dom_vid.size(); ++b) {
    const uint64_t v = w.dom_vid[b];
    if (!owned(v) && !std::binary_search(gh.begin(), gh.end(), v)) continue;
    g.dom_vid.push_back(local_id(v));
    for (uint64_t i = w.dom_offset[b]; i < w.dom_offset[b + 1]; ++i) {
      g.dom_value.push_back(w.dom_value[i]);
      g.dom_truthiness.push_back(w.dom_truthiness[i]);
    }
    g.dom_offset.push_back(g.dom_value.size());
  }
  g.w_initial_value = w.w_initial_value;
  g.w_is_fixed = w.w_is_fixed;
}

void make_shard(const LoadedGraph &w, uint64_t begin, uint64_t end, ShardGraph &out) {
  const uint32_t nth = dwx::host_threads();
  const uint64_t F = w.n_factors, n_owned = end - begin;
  auto owned = [&](uint64_t v) { return v >= begin && v < end; };
  // Two passes over the factor list in blocks of 64 k factors, both parallel, with nothing of
  // the size of the list kept in between (eight rank threads cut their shards side by side: at
  // config 5's 10^9 factors three F-sized arrays each were 136 GB, and the prefix sums between
  // the passes were serial): pass 1 counts the factors that touch an owned variable and their
  // edges per block and collects the ghosts, pass 2 (below) re-tests and copies.
  auto keeps = [&](uint64_t f) {
    for (uint64_t e = w.fac_edge_offset[f]; e < w.fac_edge_offset[f + 1]; ++e)
      if (owned(w.edge_vid[e])) return true;
    return false;
  };
  const uint64_t BLK = 1u << 16, nblk = (F + BLK - 1) / BLK;
  std::vector<uint64_t> blk_f(nblk + 1, 0), blk_e(nblk + 1, 0);
  const uint32_t T = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(nth, nblk));
  std::vector<std::vector<uint64_t>> gh_part(T);
  dwx::parallel_parts(nblk, T, [&](uint32_t t, uint64_t bb, uint64_t be) {
    std::vector<uint64_t> &gp = gh_part[t];
    size_t sorted_upto = 0;
    for (uint64_t b = bb; b < be; ++b) {
      uint64_t nfk = 0, nek = 0;
      for (uint64_t f = b * BLK; f < std::min(F, (b + 1) * BLK); ++f) {
        if (!keeps(f)) continue;
        ++nfk; nek += w.fac_edge_offset[f + 1] - w.fac_edge_offset[f];
        for (uint64_t e = w.fac_edge_offset[f]; e < w.fac_edge_offset[f + 1]; ++e)
          if (!owned(w.edge_vid[e])) gp.push_back(w.edge_vid[e]);
      }
      blk_f[b + 1] = nfk; blk_e[b + 1] = nek;
      if (gp.size() - sorted_upto > (1u << 20)) {    // (a ghost is named by every factor that reads it)
        std::sort(gp.begin(), gp.end());
        gp.erase(std::unique(gp.begin(), gp.end()), gp.end());
        sorted_upto = gp.size();
      }
    }
    std::sort(gp.begin(), gp.end());
    gp.erase(std::unique(gp.begin(), gp.end()), gp.end());
  }, 2);
  for (uint64_t b = 0; b < nblk; ++b) { blk_f[b + 1] += blk_f[b]; blk_e[b + 1] += blk_e[b]; }
  const uint64_t nf = blk_f[nblk], ne = blk_e[nblk];
  // ghosts: remote endpoints of kept factors, ascending
  std::vector<uint64_t> gh;
  for (auto &gp : gh_part) { gh.insert(gh.end(), gp.begin(), gp.end()); std::vector<uint64_t>().swap(gp); }
  std::sort(gh.begin(), gh.end());
  gh.erase(std::unique(gh.begin(), gh.end()), gh.end());
  out.ghosts = gh;
  out.n_ghost = gh.size();
  out.begin = begin; out.end = end;
  auto local_id = [&](uint64_t v) -> uint64_t {
    if (owned(v)) return v - begin;
    return n_owned + (uint64_t)(std::lower_bound(gh.begin(), gh.end(), v) - gh.begin());
  };
  fill_shard_variables(w, out);
  LoadedGraph &g = out.g;
  g.n_factors = nf; g.n_edges = ne;
  g.fac_func.reset(nf); g.fac_edge_offset.reset(nf + 1); g.fac_weight_id.reset(nf); g.fac_feature_value.reset(nf);
  g.edge_vid.reset(ne); g.edge_equal_to.reset(ne);
  dwx::parallel_ranges(nblk, nth, [&](uint64_t bb, uint64_t be) {
    for (uint64_t b = bb; b < be; ++b) {
      uint64_t nfid = blk_f[b], o = blk_e[b];
      for (uint64_t f = b * BLK; f < std::min(F, (b + 1) * BLK); ++f) {
        if (!keeps(f)) continue;
        g.fac_func[nfid] = w.fac_func[f];
        g.fac_edge_offset[nfid] = o;
        g.fac_weight_id[nfid] = w.fac_weight_id[f];
        g.fac_feature_value[nfid] = w.fac_feature_value[f];
        for (uint64_t e = w.fac_edge_offset[f]; e < w.fac_edge_offset[f + 1]; ++e, ++o) {
          g.edge_vid[o] = local_id(w.edge_vid[e]);
          g.edge_equal_to[o] = w.edge_equal_to[e];
        }
        ++nfid;
      }
    }
  }, 2);
  g.fac_edge_offset[nf] = ne;
}

// ------------------------------------------------------------------ the ranks
namespace {
struct Shared {          // what the rank threads share (one process)
  const CmdLine *args = nullptr;
  const LoadedGraph *whole = nullptr;      // shards: meta counts, variables, domains, weights only
  std::vector<ShardGraph> *shards = nullptr;   // [rank], cut while loading (load_factors_sharded)
  int world = 1;
  std::vector<int> devices;
  HostAgree *agree = nullptr;
  Comm *comm = nullptr;
  bool replicas = false;
  dwx_graph *replica_graph = nullptr;          // replicas: compiled once, shared
  std::vector<const std::vector<uint64_t> *> ghosts;   // [rank] -> its ghost list (shards)
  std::vector<std::pair<uint64_t, uint64_t>> bounds;   // [rank] -> [begin, end)
  // results for the dump (filled by every rank, read by rank 0 after a barrier)
  struct Result {
    const LoadedGraph *g = nullptr;
    uint64_t id_offset = 0, n_owned = 0;
    std::vector<uint64_t> tallies, nsamples, base, sparse;
  };
  std::vector<Result> results;
  std::vector<double> weights;                 // rank 0's after learning
  std::string first_error;
  std::mutex err_mutex;
};

class Rank {
 public:
  Rank(Shared &sh, int rank) : sh_(sh), rank_(rank), args_(*sh.args) {}
  ~Rank() {
    for (auto &p : send_) dwx_halo_destroy(p.h);
    for (auto &p : recv_) dwx_halo_destroy(p.h);
    dwx_sampler_destroy(s_);
    if (!sh_.replicas) dwx_graph_destroy(graph_);
  }
  void run();

 private:
  struct Peer { int peer; dwx_halo *h; void *buf; uint64_t n; };
  void setup();
  void setup_halo();
  void halo(int chains);
  void learn_shards();
  void learn_replicas();
  void inference();
  void collect();
  // the mini-batch plan, agreed once per batch count (mirror of dwx_sgd_plan on the GLOBAL
  // curvature: a weight's curvature adds up over the shards)
  double global_curvature(uint32_t batches);
  void plan(double stepsize, uint32_t &batches, uint32_t &n_chunks, double &min_step);
  void test_fault(uint64_t epoch);
  bool root() const { return rank_ == 0; }
  bool progress() const { return root() && !args_.should_be_quiet; }

  Shared &sh_;
  int rank_;
  const CmdLine &args_;
  Check ok;
  ShardGraph shard_;
  dwx_graph *graph_ = nullptr;
  dwx_sampler *s_ = nullptr;
  dwx_graph_info info_{};
  uint64_t W_ = 0;
  void *d_grad_ = nullptr, *d_weights_ = nullptr, *d_tallies_ = nullptr;
  bool has_categorical_ = false;
  bool narrow_ = false;           // the gradient all-reduce travels as 32- or 16-bit counts
  uint32_t narrow_shift_ = 0, narrow_bits_ = 32;
  std::vector<Peer> send_, recv_;
  std::map<uint32_t, double> lam_;
  std::map<uint32_t, uint32_t> level_chunks_;
  std::map<uint32_t, bool> level_dynamic_;
  uint32_t max_batches_ = 0;
  bool dynamic_now_ = false;
};

void Rank::setup() {
  const LoadedGraph &whole = *sh_.whole;
  dwx_options o;
  dwx_default_options(&o);
  o.device = sh_.devices[rank_];
  o.sample_evidence = args_.should_sample_evidence;
  o.learn_non_evidence = args_.should_learn_non_evidence;
  o.noise_aware = args_.is_noise_aware;
  o.regularization = args_.regularization_l1 ? 0 : 1;
  o.reg_param = args_.reg_param;
  o.step_cap = args_.step_cap;
  o.plan_layouts = args_.plan_layouts;
  (void)dwx_device_init(o.device);
  if (sh_.replicas) {
    graph_ = sh_.replica_graph;
    o.seed = args_.seed + (uint64_t)rank_;      // every replica its own chains
    sh_.results[rank_].g = &whole;
    sh_.results[rank_].n_owned = whole.n_variables;
  } else {
    uint64_t b, e;
    shard_range(whole.n_variables, rank_, sh_.world, b, e);
    if (e <= b)
      throw std::runtime_error("--gpus " + std::to_string(sh_.world) + ": rank " + std::to_string(rank_) +
                               " would own no variable (" + std::to_string(whole.n_variables) +
                               " variables in blocks of " + std::to_string((whole.n_variables + sh_.world - 1) / sh_.world) +
                               "): use fewer GPUs");
    shard_ = std::move((*sh_.shards)[rank_]);   // cut while loading
    sh_.bounds[rank_] = {b, e};
    sh_.ghosts[rank_] = &shard_.ghosts;
    dwx_graph_desc desc = shard_.desc();
    dwx_compile_opts co = compile_opts_for(args_);
    co.n_threads = std::max(1u, dwx::host_threads() / (uint32_t)sh_.world);   // the ranks compile side by side
    ok(dwx_graph_create(&desc, &co, &graph_));
    o.seed = args_.seed;                         // one Philox key: counters use global ids
    o.var_id_offset = b;
    sh_.results[rank_].g = &shard_.g;
    sh_.results[rank_].id_offset = b;
    sh_.results[rank_].n_owned = e - b;
  }
  ok(dwx_graph_get_info(graph_, &info_));
  ok(dwx_sampler_create(graph_, &o, &s_));
  W_ = whole.n_weights;
  uint64_t nb = 0;
  ok(dwx_device_buffer(s_, DWX_BUF_GRAD, &d_grad_, &nb));
  ok(dwx_device_buffer(s_, DWX_BUF_WEIGHTS, &d_weights_, &nb));
  ok(dwx_device_buffer(s_, DWX_BUF_TALLIES, &d_tallies_, &nb));
  if (!sh_.replicas) {
    // what every rank must decide alike (a block without categorical variables next to one
    // with them must not reduce W elements against the other's 2 W)
    has_categorical_ = sh_.agree->max_u64(rank_, info_.has_categorical) != 0;
    // The gradient sums as 32-bit counts (half the bytes of every all-reduce) where every rank's
    // graph allows it (dwx_graph_info.grad_shift: all-boolean all-unary blocks): the smallest shift
    // any rank knows, and the bound covers the sum over all ranks.
    {
      const uint64_t sh = info_.grad_shift;
      const uint64_t m = sh_.agree->max_u64(rank_, sh ? 64 - sh : 64);          // (64 - min shift; 64: some rank knows nothing)
      // (the largest contribution in fixed-point units: ranks may know different shifts, the
      // count bound is taken at the common one)
      const uint64_t qmax = sh_.agree->max_u64(rank_, info_.grad_unit_max << sh);
      const uint64_t recs = sh_.agree->max_u64(rank_, info_.max_records_per_weight) * (uint64_t)sh_.world;
      narrow_shift_ = (uint32_t)(64 - m);
      const uint64_t unit = narrow_shift_ > 0 ? qmax >> narrow_shift_ : 0;
      // (the override is agreed on: ranks that disagreed would put different types through one collective)
      const bool no_narrow = sh_.agree->max_u64(rank_, getenv("DWX_NO_NARROW_ALLREDUCE") ? 1 : 0) != 0;
      const bool no_16 = sh_.agree->max_u64(rank_, getenv("DWX_NO_16BIT_ALLREDUCE") ? 1 : 0) != 0;
      narrow_ = !has_categorical_ && narrow_shift_ > 0 && unit > 0 && recs * unit < (1ull << 31) && sh_.world > 1 && !no_narrow;
      // ... and as 16-bit counts, two per word, where even the sum over all ranks stays below 2^15
      // (config 5a: a weight has ~10^3 records over all shards): a quarter of the int64 bytes
      narrow_bits_ = (narrow_ && recs * unit < (1ull << 15) && !no_16) ? 16u : 32u;
      if (root() && !args_.should_be_quiet)
        std::cout << "Gradient all-reduce: "
                  << (narrow_ ? std::to_string(narrow_bits_) + "-bit counts, shift " + std::to_string(narrow_shift_) : std::string("int64 sums"))
                  << std::endl;
    }
    // each rank counted its own block's boolean updates and curvature bounds: sum [T | h] once
    void *ts = nullptr;
    ok(dwx_device_buffer(s_, DWX_BUF_TSTATIC, &ts, &nb));
    if (W_) sh_.comm->allreduce_sum_i64(rank_, s_, ts, 2 * W_);
    setup_halo();
    halo(3);   // ghosts start from their owners' state
  }
}

void Rank::setup_halo() {
  sh_.agree->barrier();   // every rank's ghost list and bounds are published
  const uint64_t b = sh_.bounds[rank_].first, e = sh_.bounds[rank_].second, n_owned = e - b;
  for (int k = 0; k < sh_.world; ++k) {
    if (k == rank_) continue;
    // my owned variables that rank k ghosts (ascending global id on both sides)
    const std::vector<uint64_t> &theirs = *sh_.ghosts[k];
    auto lo = std::lower_bound(theirs.begin(), theirs.end(), b), hi = std::lower_bound(theirs.begin(), theirs.end(), e);
    if (hi > lo) {
      std::vector<uint64_t> ids;
      for (auto it = lo; it != hi; ++it) ids.push_back(*it - b);
      Peer p{k, nullptr, nullptr, ids.size()};
      ok(dwx_halo_create(s_, ids.data(), ids.size(), &p.h));
      uint64_t nb = 0;
      ok(dwx_halo_buffer(p.h, &p.buf, &nb));
      send_.push_back(p);
    }
    // my ghosts that rank k owns
    const std::vector<uint64_t> &mine = shard_.ghosts;
    const uint64_t kb = sh_.bounds[k].first, ke = sh_.bounds[k].second;
    auto mlo = std::lower_bound(mine.begin(), mine.end(), kb), mhi = std::lower_bound(mine.begin(), mine.end(), ke);
    if (mhi > mlo) {
      std::vector<uint64_t> ids;
      for (auto it = mlo; it != mhi; ++it) ids.push_back(n_owned + (uint64_t)(it - mine.begin()));
      Peer p{k, nullptr, nullptr, ids.size()};
      ok(dwx_halo_create(s_, ids.data(), ids.size(), &p.h));
      uint64_t nb = 0;
      ok(dwx_halo_buffer(p.h, &p.buf, &nb));
      recv_.push_back(p);
    }
  }
  sh_.agree->barrier();   // (the ghost lists may go away only after everybody has read them)
}

// chains: bit 0 free, bit 1 evidence.  Gather -> ONE grouped send/recv -> scatter, all on the
// sampler's stream (one bit per boolean boundary variable and chain: dwx_halo_message_bytes).
void Rank::halo(int chains) {
  if (sh_.replicas || sh_.world == 1) return;
  auto bytes = [&](const Peer &p) {
    uint64_t nb = 0;
    ok(dwx_halo_message_bytes(p.h, chains, &nb));
    return nb;
  };
  std::vector<Xfer> sends, recvs;
  for (auto &p : send_) { ok(dwx_halo_pack_async(p.h, chains)); sends.push_back({p.peer, p.buf, bytes(p)}); }
  for (auto &p : recv_) recvs.push_back({p.peer, p.buf, bytes(p)});
  sh_.comm->exchange(rank_, s_, sends, recvs);    // (every rank takes part, also with empty lists)
  for (auto &p : recv_) ok(dwx_halo_unpack_async(p.h, chains));
}

// Test hook (tests/test_dw_multi.py): DWX_DW_TEST_FAULT="rank:epoch[:block]" -- that rank throws at
// the start of that learning epoch; with "block" every other rank then sits in a call that
// never returns (what a collective without its dead peer does), so that only the watchdog of
// gibbs_multi can end the run.
// Compiled only into the tests' host harness (-DDWX_TEST_HOOKS, tests/hipemu/Makefile): the product
// binary does not read the variable.
void Rank::test_fault(uint64_t epoch) {
#ifdef DWX_TEST_HOOKS
  const char *e = getenv("DWX_DW_TEST_FAULT");
  if (!e) return;
  int r = -1; unsigned long long ep = 0; char mode[16] = {0};
  if (sscanf(e, "%d:%llu:%15s", &r, &ep, mode) < 2 || ep != epoch) return;
  if (r == rank_) throw std::runtime_error("injected fault (DWX_DW_TEST_FAULT)");
  if (!strcmp(mode, "block")) for (;;) std::this_thread::sleep_for(std::chrono::seconds(1));
#else
  (void)epoch;
#endif
}

double Rank::global_curvature(uint32_t batches) {
  auto it = lam_.find(batches);
  if (it != lam_.end()) return it->second;
  double lam = 0;
  ok(dwx_sgd_curvature(s_, batches, &lam));
  lam = (double)sh_.world * sh_.agree->max_f64(rank_, lam);
  lam_[batches] = lam;
  return lam;
}

void Rank::plan(double stepsize, uint32_t &batches, uint32_t &n_chunks, double &min_step) {
  const double cap = args_.step_cap;
  batches = 1;
  if (cap > 0 && stepsize > 0) {
    if (!max_batches_) {
      // no rank can cut finer than its tile count; the search ends at the largest (<= 64)
      max_batches_ = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(64, sh_.agree->max_u64(rank_, info_.num_tiles)));
    }
    // (the rule of dwx_sgd_plan on the global estimate: cut finer while it is above the cap and
    // a finer cut still lowers it by a fifth)
    while (batches < max_batches_ && stepsize * global_curvature(batches) > cap) {
      if (global_curvature(2 * batches) > 0.8 * global_curvature(batches) &&
          (4 * batches > max_batches_ || global_curvature(4 * batches) > 0.64 * global_curvature(batches)))
        break;
      batches *= 2;
    }
  }
  if (!level_chunks_.count(batches)) {
    // first use of this batch count: agree on the slowest rank's chunk count and share the
    // static tables.  (Only this level: every rank derives the same plan for every step, so the
    // coarser levels the decaying step walks down to are agreed on when -- and if -- they are
    // first used; building them all in the first epoch cost 24 s at config 5's size.)
    for (uint32_t b = batches, once = 1; once; once = 0) {
      if (!level_chunks_.count(b)) {
        if (b > 1 && cap > 0) (void)global_curvature(b);
        uint32_t got_b = 0, n_mine = 0;
        ok(dwx_sgd_plan(s_, stepsize, b, &got_b, &n_mine, nullptr));
        // (a forced count is honoured whatever this shard's size; a rank that ran another plan
        // than its peers would enter other collectives than they do)
        if (sh_.agree->max_u64(rank_, got_b != b ? 1 : 0) != 0)
          throw std::runtime_error("the ranks could not agree on " + std::to_string(b) + " mini-batches per sweep");
        level_chunks_[b] = (uint32_t)sh_.agree->max_u64(rank_, n_mine);
        if (b > 1) {
          void *tp = nullptr;
          uint64_t nb = 0;
          ok(dwx_device_buffer(s_, DWX_BUF_TSTATIC_PLAN, &tp, &nb));
          // if ANY rank has no per-chunk tables for this level, all count dynamically
          const bool dyn = sh_.agree->max_u64(rank_, nb == 0 ? 1 : 0) != 0;
          level_dynamic_[b] = dyn;
          if (!dyn && W_) {
            ok(dwx_sgd_plan_rows(s_, level_chunks_[b]));
            ok(dwx_device_buffer(s_, DWX_BUF_TSTATIC_PLAN, &tp, &nb));
            sh_.comm->allreduce_sum_i64(rank_, s_, tp, nb / 8);
            ok(dwx_wait(s_));
          }
        }
      }
      if (b == 1) break;
    }
  }
  uint32_t got = 0, n_mine = 0;
  ok(dwx_sgd_plan(s_, stepsize, batches, &got, &n_mine, &min_step));
  dynamic_now_ = batches > 1 && level_dynamic_[batches];
  if (dynamic_now_) ok(dwx_sgd_plan_force_dynamic(s_, 1));
  n_chunks = level_chunks_[batches];
}

// DimmWitted::learn (src/dimmwitted.cc:162-207) over variable-block shards
void Rank::learn_shards() {
  const uint64_t V = sh_.whole->n_variables;
  std::vector<double> weights(W_), prev(sh_.whole->w_initial_value);
  double t_total = now(), stepsize = args_.stepsize;
  for (uint64_t e = 0; e < args_.n_learning_epoch; ++e) {
    if (progress()) std::cout << std::setprecision(3) << "LEARNING EPOCH " << e << "~" << e << "...." << std::flush;
    const double t0 = now();
    uint32_t batches = 1, n_chunks = 1;
    double min_step = stepsize;
    test_fault(e);
    plan(stepsize, batches, n_chunks, min_step);
    for (uint32_t c = 0; c < n_chunks; ++c) {
      ok(dwx_sgd_accumulate_async(s_, c));        // ranks with fewer chunks idle through the rest
      if (batches > 1 || c + 1 == n_chunks) {
        // [G | T]: the counts T travel only when somebody counts dynamically (categorical
        // variables; split plans without tables)
        if (W_ && narrow_ && !dynamic_now_) {
          // (two's complement: the unsigned 32-bit sum of the ranks' signed counts is their signed sum)
          void *d32 = nullptr;
          uint64_t n32 = 0;
          ok(dwx_grad_pack_async(s_, narrow_shift_, narrow_bits_, &d32, &n32));
          sh_.comm->allreduce_sum_u32(rank_, s_, d32, n32);
          ok(dwx_grad_unpack_async(s_, narrow_shift_, narrow_bits_));
        } else if (W_) {
          sh_.comm->allreduce_sum_i64(rank_, s_, d_grad_, (has_categorical_ || dynamic_now_) ? 2 * W_ : W_);
        }
        ok(dwx_sgd_apply_async(s_));
      }
    }
    ok(dwx_sgd_finish(s_));
    halo(3);
    if (!args_.should_be_quiet) {   // timed epochs: everybody waits (the quiet run stays queued)
      ok(dwx_wait(s_));
      sh_.agree->barrier();
    }
    if (progress()) {
      const double elapsed = now() - t0;
      ok(dwx_get_weights(s_, weights.data()));
      double lmax = -INFINITY, l2 = 0.0;
      for (uint64_t j = 0; j < W_; ++j) {
        const double diff = fabs(weights[j] - prev[j]);
        l2 += diff * diff;
        if (lmax < diff) lmax = diff;
      }
      std::cout << std::setprecision(3) << "" << elapsed << " sec." << "," << V / elapsed << " vars/sec."
                << ",stepsize=" << stepsize << ",lmax=" << lmax / stepsize << ",l2=" << sqrt(l2) / stepsize;
      if (batches > 1 || min_step < 0.999 * stepsize) std::cout << ",batches=" << batches << ",min_step=" << min_step;
      std::cout << std::endl << std::setprecision(6);
      prev = weights;
    }
    stepsize *= args_.decay;
  }
  ok(dwx_wait(s_));
  sh_.agree->barrier();
  if (root()) std::cout << std::setprecision(6) << "TOTAL LEARNING TIME: " << now() - t_total << " sec." << std::endl;
}

// the reference's replica loop: learn on every copy, then merge + average + copy back
// (src/dimmwitted.cc:162-216); n requested epochs = ceil(n / replicas) rounds (:280-282)
void Rank::learn_replicas() {
  const uint64_t V = sh_.whole->n_variables, n = (uint64_t)sh_.world;
  const uint64_t rounds = (args_.n_learning_epoch + n - 1) / n;
  std::vector<double> weights(W_), prev(sh_.whole->w_initial_value);
  double t_total = now(), stepsize = args_.stepsize;
  for (uint64_t e = 0; e < rounds; ++e) {
    if (progress())
      std::cout << std::setprecision(3) << "LEARNING EPOCH " << e * n << "~" << (e + 1) * n - 1 << "...." << std::flush;
    const double t0 = now();
    ok(dwx_sample_sgd_async(s_, stepsize));
    if (W_) sh_.comm->allreduce_sum_f64(rank_, s_, d_weights_, W_);
    ok(dwx_average_weights_async(s_, (uint32_t)n));
    if (!args_.should_be_quiet) {
      ok(dwx_wait(s_));
      sh_.agree->barrier();
    }
    if (progress()) {
      const double elapsed = now() - t0;
      ok(dwx_get_weights(s_, weights.data()));
      double lmax = -INFINITY, l2 = 0.0;
      for (uint64_t j = 0; j < W_; ++j) {
        const double diff = fabs(weights[j] - prev[j]);
        l2 += diff * diff;
        if (lmax < diff) lmax = diff;
      }
      std::cout << std::setprecision(3) << "" << elapsed << " sec." << "," << (V * n) / elapsed << " vars/sec."
                << ",stepsize=" << stepsize << ",lmax=" << lmax / stepsize << ",l2=" << sqrt(l2) / stepsize
                << std::endl << std::setprecision(6);
      prev = weights;
    }
    stepsize *= args_.decay;
  }
  ok(dwx_wait(s_));
  sh_.agree->barrier();
  if (root()) std::cout << std::setprecision(6) << "TOTAL LEARNING TIME: " << now() - t_total << " sec." << std::endl;
}

// DimmWitted::inference (src/dimmwitted.cc:121-160)
void Rank::inference() {
  const uint64_t V = sh_.whole->n_variables, n = (uint64_t)sh_.world;
  const uint64_t rounds = sh_.replicas ? (args_.n_inference_epoch + n - 1) / n : args_.n_inference_epoch;
  const uint64_t per_round = sh_.replicas ? n : 1;
  double t_total = now();
  ok(dwx_clear_tallies(s_));
  // quiet and nothing to exchange between the sweeps (replicas; shards without a ghost anywhere):
  // all rounds in one call -- one launch on an all-unary graph (dwx_sample_n_async)
  bool exchange = false;
  if (!sh_.replicas)
    for (int k = 0; k < sh_.world; ++k) exchange = exchange || !sh_.ghosts[k]->empty();
  uint64_t e0 = 0;
  if (args_.should_be_quiet && !exchange)
    for (; e0 < rounds; e0 += std::min<uint64_t>(rounds - e0, 1u << 20))
      ok(dwx_sample_n_async(s_, (uint32_t)std::min<uint64_t>(rounds - e0, 1u << 20)));
  for (uint64_t e = e0; e < rounds; ++e) {
    if (progress())
      std::cout << std::setprecision(3) << "INFERENCE EPOCH " << e * per_round << "~" << (e + 1) * per_round - 1 << "...." << std::flush;
    const double t0 = now();
    ok(dwx_sample_async(s_));
    halo(2);
    if (!args_.should_be_quiet) {
      ok(dwx_wait(s_));
      sh_.agree->barrier();
    }
    if (progress()) {
      const double elapsed = now() - t0;
      std::cout << std::setprecision(3) << "" << elapsed << " sec." << "," << (V * per_round) / elapsed << " vars/sec"
                << std::endl << std::setprecision(6);
    }
  }
  // replicas: aggregate_marginals_from (src/inference_result.cc:113-127) -- sum the copies' tallies
  if (sh_.replicas && rounds && info_.num_values) sh_.comm->allreduce_sum_u32(rank_, s_, d_tallies_, info_.num_values);
  ok(dwx_wait(s_));
  sh_.agree->barrier();
  if (root()) std::cout << std::setprecision(6) << "TOTAL INFERENCE TIME: " << now() - t_total << " sec." << std::endl;
}

void Rank::collect() {
  Shared::Result &r = sh_.results[rank_];
  if (sh_.replicas && !root()) return;     // every replica holds the summed tallies: one copy is enough
  const uint64_t nv = r.g->n_variables;
  r.tallies.resize(info_.num_values); r.nsamples.resize(nv); r.base.resize(nv); r.sparse.resize(info_.num_values);
  ok(dwx_get_tallies(s_, r.tallies.data(), r.nsamples.data()));
  ok(dwx_graph_get_values(graph_, r.base.data(), r.sparse.data()));
  if (sh_.replicas)
    for (auto &x : r.nsamples) x *= (uint64_t)sh_.world;
}

void Rank::run() {
  setup();
  sh_.agree->barrier();
  if (sh_.replicas) learn_replicas(); else learn_shards();
  if (root()) {
    sh_.weights.resize(W_);
    ok(dwx_get_weights(s_, sh_.weights.data()));
  }
  sh_.agree->barrier();      // rank 0 dumps the weights while the others wait here
  if (root()) {
    // dump_weights (src/dimmwitted.cc:245-258): before inference starts
    if (!args_.should_be_quiet) {
      std::cout << "LEARNING SNIPPETS (QUERY WEIGHTS):" << std::endl;
      for (uint64_t j = 0; j < W_ && j < 10; ++j) std::cout << "   " << j << " " << sh_.weights[j] << std::endl;
      std::cout << "   ..." << std::endl;
    }
    const std::string fn = args_.output_folder + "/inference_result.out.weights.text";
    std::cout << "DUMPING... TEXT    : " << fn << std::endl;
    std::ofstream f(fn);
    if (!f) throw std::runtime_error("cannot write " + fn);
    dump_weights_in_text(f, sh_.weights);
  }
  inference();
  if (args_.n_inference_epoch > 0) collect();
  sh_.agree->barrier();
}
}  // namespace

// ------------------------------------------------------------------ entry
int gibbs_multi(const CmdLine &args) {
  int exit_code = 0;
  dwx_graph *replica_graph = nullptr;
  try {
    const bool replicas = args.gpus < 1;           // -c N without --gpus: the reference's n_datacopy
    int n = replicas ? (int)args.n_datacopy : args.gpus;
    int32_t have = 0;
    (void)dwx_device_count(&have);
    std::vector<int> devices = args.devices;
    if (devices.empty()) {
      if (args.comm == "host") {
        // the test communicator may stack its ranks on the devices there are
        for (int r = 0; r < n; ++r) devices.push_back(have > 0 ? r % have : 0);
      } else if (replicas && have >= 1 && n > have) {
        // n_datacopy describes NUMA copies of a CPU run; existing command lines keep working on
        // the GPUs there are (the epoch counts follow the copies actually made)
        std::cerr << "dw: -c " << n << " but " << have << " GPU(s) visible: " << have << " replica(s)" << std::endl;
        n = have;
        for (int r = 0; r < n; ++r) devices.push_back(r);
      } else {
        if (n > have)
          throw std::runtime_error("--gpus " + std::to_string(n) + " but " + std::to_string(have) +
                                   " HIP device(s) visible (the dwx sampler has no CPU fallback)");
        for (int r = 0; r < n; ++r) devices.push_back(r);
      }
    }
    if ((int)devices.size() != n) throw std::runtime_error("--devices must list one device per rank");
    // (DWX_DW_FORCE_MULTI: run the rank machinery and the communicator with ONE rank -- the RCCL
    // calls of a box with a single GPU; tests)
    if (n <= 1 && !getenv("DWX_DW_FORCE_MULTI")) {
      CmdLine one = args;
      one.gpus = 0; one.n_datacopy = 1;
      if (!devices.empty()) one.device = devices[0];
      return gibbs(one);
    }
    if (!args.should_be_quiet) {
      std::cout << std::endl;
      std::cout << "#################MACHINE CONFIG#################" << std::endl;
      std::cout << "# # HIP devices      : ";
      for (int d : devices) std::cout << d << " ";
      std::cout << "(" << (replicas ? "replicas of the whole graph, -c" : "variable-block shards, --gpus") << ")" << std::endl;
      std::cout << "################################################" << std::endl;
      std::cout << std::endl;
      std::cout << args << std::endl;
    }
    LoadedGraph whole;
    read_meta(args.fg_file, whole);
    std::cout << "Factor graph to load:\t#V=" << whole.n_variables << " #F=" << whole.n_factors
              << " #W=" << whole.n_weights << " #E=" << whole.n_edges << " #Val=0" << std::endl;
    std::cout << "\tinitializing factor graph..." << std::endl;
    std::cout << "\tloading factor graph..." << std::endl;
    load_variables(args.variable_file, whole);
    load_weights(args.weight_file, whole);
    load_domains(args.domain_file, whole);
    // shards: every factor record is decoded once and lands in the shards that own one of its
    // variables -- the whole graph's factor columns never exist (replicas need them: -c N)
    std::vector<ShardGraph> shards;
    if (replicas) {
      load_factors(args.factor_file, whole);
    } else {
      for (int r = 0; r < n; ++r) {
        uint64_t b, e;
        shard_range(whole.n_variables, r, n, b, e);
        if (e <= b)
          throw std::runtime_error("--gpus " + std::to_string(n) + ": rank " + std::to_string(r) +
                                   " would own no variable (" + std::to_string(whole.n_variables) +
                                   " variables in blocks of " + std::to_string((whole.n_variables + n - 1) / n) +
                                   "): use fewer GPUs");
      }
      load_factors_sharded(args.factor_file, whole, n, shards);
      if (getenv("DWX_DW_VERIFY_SHARDS")) {
        // (test hook: the sharded loader against load_factors + make_shard, column by column)
        LoadedGraph all;
        read_meta(args.fg_file, all);
        load_variables(args.variable_file, all);
        load_weights(args.weight_file, all);
        load_domains(args.domain_file, all);
        load_factors(args.factor_file, all);
        for (int r = 0; r < n; ++r) {
          ShardGraph ref;
          make_shard(all, shards[r].begin, shards[r].end, ref);
          const LoadedGraph &a = shards[r].g, &b = ref.g;
          auto same = [](const auto &x, const auto &y, uint64_t k) {
            return x.size() >= k && y.size() >= k && std::equal(x.data(), x.data() + k, y.data());
          };
          const bool ok = shards[r].ghosts == ref.ghosts && shards[r].n_ghost == ref.n_ghost &&
              a.n_variables == b.n_variables && a.n_factors == b.n_factors && a.n_edges == b.n_edges &&
              a.n_weights == b.n_weights && a.n_evidence == b.n_evidence && a.n_query == b.n_query &&
              a.var_role == b.var_role && a.var_init_value == b.var_init_value && a.var_dtype == b.var_dtype &&
              a.var_cardinality == b.var_cardinality && a.dom_vid == b.dom_vid && a.dom_offset == b.dom_offset &&
              a.dom_value == b.dom_value && a.dom_truthiness == b.dom_truthiness &&
              a.w_initial_value == b.w_initial_value && a.w_is_fixed == b.w_is_fixed &&
              same(a.fac_func, b.fac_func, a.n_factors) && same(a.fac_edge_offset, b.fac_edge_offset, a.n_factors + 1) &&
              same(a.fac_weight_id, b.fac_weight_id, a.n_factors) &&
              std::memcmp(a.fac_feature_value.data(), b.fac_feature_value.data(), 8 * a.n_factors) == 0 &&
              same(a.edge_vid, b.edge_vid, a.n_edges) && same(a.edge_equal_to, b.edge_equal_to, a.n_edges);
          if (!ok) throw std::runtime_error("DWX_DW_VERIFY_SHARDS: shard " + std::to_string(r) + " differs from make_shard");
        }
        std::cout << "DWX_DW_VERIFY_SHARDS: " << n << " shards equal make_shard" << std::endl;
      }
    }
    HostAgree agree(n);
    std::unique_ptr<Comm> comm = args.comm == "host" ? make_host_comm(n, &agree) : make_rccl_comm(devices);
    Shared sh;
    sh.args = &args; sh.whole = &whole; sh.world = n; sh.devices = devices; sh.agree = &agree; sh.comm = comm.get();
    sh.replicas = replicas; sh.shards = &shards;
    sh.ghosts.assign(n, nullptr); sh.bounds.assign(n, {0, 0}); sh.results.resize(n);
    uint64_t num_values = 0;
    for (uint64_t v = 0; v < whole.n_variables; ++v) num_values += whole.var_dtype[v] == 0 ? 1 : whole.var_cardinality[v];
    if (replicas) {
      dwx_graph_desc desc = whole.desc();
      Check ok;
      dwx_compile_opts co = compile_opts_for(args);
      ok(dwx_graph_create(&desc, &co, &replica_graph));
      sh.replica_graph = replica_graph;
    }
    auto print_size = [&](const char *what) {
      std::cout << what << "#V=" << whole.n_variables << "(#Vqry=" << whole.n_query << " #Vevd=" << whole.n_evidence
                << ") #F=" << whole.n_factors << " #W=" << whole.n_weights << " #E=" << whole.n_edges
                << " #Val=" << num_values << std::endl;
    };
    print_size("Factor graph loaded:\t");
    print_size("Factor graph indexed:\t");
    if (!args.should_be_quiet)
      std::cout << "Ranks: " << n << " x " << (replicas ? "replica" : "shard") << ", collectives: " << comm->name() << std::endl;

    std::vector<std::unique_ptr<Rank>> ranks;
    for (int r = 0; r < n; ++r) ranks.emplace_back(new Rank(sh, r));
    std::vector<std::thread> th;
    std::atomic<int> done{0};
    std::atomic<bool> failed{false};
    std::chrono::steady_clock::time_point t_fail;
    for (int r = 0; r < n; ++r)
      th.emplace_back([&, r]() {
        try {
          ranks[r]->run();
        } catch (const std::exception &e) {
          {
            std::lock_guard<std::mutex> lk(sh.err_mutex);
            if (sh.first_error.empty() && std::string(e.what()) != "another rank failed")
              sh.first_error = "rank " + std::to_string(r) + ": " + e.what();
            if (!failed.load()) { t_fail = std::chrono::steady_clock::now(); failed.store(true); }
          }
          // release the others: the host-side agreements throw, the device-side collectives abort
          agree.abort();
          comm->abort();
        }
        ++done;
      });
    // A rank that failed takes the run down: the others are woken out of host barriers and
    // collectives (above); one that still sits in a runtime call that never returns -- a
    // collective its dead peer will not join -- must not keep `dw` alive: five seconds after
    // the first failure the process reports it and exits 1 (no re-exec, no retry).
    while (done.load() < n) {
      std::this_thread::sleep_for(std::chrono::milliseconds(20));
      if (failed.load() && std::chrono::steady_clock::now() - t_fail > std::chrono::seconds(5)) {
        std::string msg;
        { std::lock_guard<std::mutex> lk(sh.err_mutex); msg = sh.first_error; }
        std::cerr << "dw: " << (msg.empty() ? "a rank failed" : msg) << " (the other ranks did not stop within 5 s: exiting)" << std::endl;
        std::cout.flush();
        std::_Exit(1);
      }
    }
    for (auto &t : th) t.join();
    if (!sh.first_error.empty()) throw std::runtime_error(sh.first_error);

    // aggregate_results_and_dump (src/dimmwitted.cc:260-277), only if -i > 0: the ranks' blocks
    // in rank order = ascending variable id
    if (args.n_inference_epoch > 0) {
      const std::string fn = args.output_folder + "/inference_result.out.text";
      if (!args.should_be_quiet) {
        std::cout << "INFERENCE SNIPPETS (QUERY VARIABLES):" << std::endl;
        size_t ct = 0;
        const Shared::Result &r0 = sh.results[0];
        for (uint64_t v = 0; v < r0.n_owned && ct < 10; ++v) {
          if (r0.g->var_role[v] >= 1 && !args.should_sample_evidence) continue;
          ++ct;
          std::cout << "   " << v + r0.id_offset << "  NSAMPLE=" << r0.nsamples[v] << std::endl;
          const uint64_t k = r0.g->var_dtype[v] == 0 ? 1 : r0.g->var_cardinality[v];
          for (uint64_t j = 0; j < k; ++j)
            std::cout << "      @ " << (r0.g->var_dtype[v] == 0 ? 1 : r0.sparse[r0.base[v] + j]) << " -> EXP="
                      << 1.0 * r0.tallies[r0.base[v] + j] / r0.nsamples[v] << std::endl;
        }
        std::cout << "   ..." << std::endl;
      }
      std::cout << "DUMPING... TEXT    : " << fn << std::endl;
      std::ofstream f(fn);
      if (!f) throw std::runtime_error("cannot write " + fn);
      for (int r = 0; r < (replicas ? 1 : n); ++r) {
        const Shared::Result &res = sh.results[r];
        dump_marginals_in_text(f, *res.g, args.should_sample_evidence, res.base.data(), res.sparse.data(), res.tallies.data(),
                               res.nsamples.data(), res.id_offset, res.n_owned);
      }
    }
    quick_exit_if_done(0);     // (the result files are written: see dw_cli.cc)
    ranks.clear();   // samplers and shard graphs go before the shared replica graph
  } catch (const std::exception &e) {
    std::cerr << "dw: " << e.what() << std::endl;
    exit_code = 1;
  }
  dwx_graph_destroy(replica_graph);
  return exit_code;
}

}  // namespace dw
