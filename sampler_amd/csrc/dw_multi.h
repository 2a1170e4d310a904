// dw_multi.h -- `dw gibbs` over several GPUs of one node, in C++ above the C ABI.
//
// One host thread per GPU ("rank"), each driving its own dwx_sampler on its own device and
// stream; the ranks meet in collectives of a Comm:
//   RcclComm  (dw_rccl.cc)   rccl.h over xGMI: ncclAllReduce on the samplers' streams, grouped
//                            ncclSend/ncclRecv for the halo -- the product path
//   HostComm  (dw_multi.cc)  host-staged sums through dwx_buffer_copy: a TEST stand-in
//                            (`--comm host`; lets two ranks share one GPU, or none at all under
//                            tests/hipemu), never chosen on its own
// Two decompositions:
//   shards   (--gpus N)  the graph cut into N contiguous variable blocks, weights global;
//            per learning mini-batch ONE all-reduce(sum) of the int64 gradient vector
//            (replaces the reference's per-epoch weight averaging, src/dimmwitted.cc:209-216,
//            and its dormant merge_gradients_from, src/inference_result.cc:57-62), every rank
//            applies the identical update; factors crossing a block boundary live on both
//            sides and their remote variables (ghosts) are refreshed by a halo exchange after
//            every sweep; tallies stay sharded until the dump.
//   replicas (-c N)      the reference's own n_datacopy mode (src/dimmwitted.cc:97-119): every
//            rank holds the whole graph and its own chains; weights are summed and averaged
//            after every learning round (:199-216), ceil(n / N) rounds (:280-282), tallies are
//            summed at the end (:264-265).
#ifndef DWX_DW_MULTI_H_
#define DWX_DW_MULTI_H_

#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "dw_cli.h"

namespace dw {

// Host-side agreement between the rank threads (setup and planning only: a handful of
// scalars per batch count, never per sweep).  abort() releases everybody with an exception
// when one rank failed.
class HostAgree {
 public:
  explicit HostAgree(int n) : n_(n), d_(n), u_(n) {}
  int size() const { return n_; }
  void barrier();
  double max_f64(int rank, double v);
  uint64_t max_u64(int rank, uint64_t v);
  void abort();

 private:
  int n_, waiting_ = 0;
  uint64_t gen_ = 0;
  bool aborted_ = false;
  std::mutex m_;
  std::condition_variable cv_;
  std::vector<double> d_;
  std::vector<uint64_t> u_;
};

struct Xfer { int peer; void *dev; uint64_t nbytes; };

// Device-side collectives, called by every rank's thread with its own buffer and stream.
class Comm {
 public:
  virtual ~Comm() {}
  virtual const char *name() const = 0;
  virtual void allreduce_sum_i64(int rank, dwx_sampler *s, void *dev, uint64_t n) = 0;
  virtual void allreduce_sum_f64(int rank, dwx_sampler *s, void *dev, uint64_t n) = 0;
  virtual void allreduce_sum_u32(int rank, dwx_sampler *s, void *dev, uint64_t n) = 0;
  // one grouped exchange: every send has a matching recv on the peer, same byte count
  virtual void exchange(int rank, dwx_sampler *s, const std::vector<Xfer> &sends,
                        const std::vector<Xfer> &recvs) = 0;
  // a rank failed: make every collective that is queued or will be entered return (with an
  // error) instead of waiting for the rank that will never join it.  Any thread, any time, once.
  virtual void abort() {}
};

std::unique_ptr<Comm> make_host_comm(int n_ranks, HostAgree *agree);
// dw_rccl.cc (product) / tests/hipemu/dw_rccl_stub.cc (no RCCL under emulation: throws)
std::unique_ptr<Comm> make_rccl_comm(const std::vector<int> &devices);

// The block [begin, end) of a loaded graph as a graph of its own: owned variables first
// (local id = global id - begin), then the ghosts in ascending global id; every factor that
// touches an owned variable.
struct ShardGraph {
  LoadedGraph g;
  uint64_t n_ghost = 0, begin = 0, end = 0;
  std::vector<uint64_t> ghosts;   // global ids
  dwx_graph_desc desc() const;
};
void make_shard(const LoadedGraph &whole, uint64_t begin, uint64_t end, ShardGraph &out);
// the variable side of a shard whose begin / end / ghosts are set (variables, domains, weights)
void fill_shard_variables(const LoadedGraph &whole, ShardGraph &out);
// The factor files decoded ONCE, straight into the shards of all `world` ranks (dw_cli.cc, next
// to load_factors: same format, cuts and errors): `whole` holds the meta counts, the variables,
// domains and weights -- and never the factor columns.  Result == make_shard of every block.
void load_factors_sharded(const std::vector<std::string> &files, const LoadedGraph &whole, int world,
                          std::vector<ShardGraph> &shards);
void shard_range(uint64_t total, int rank, int world, uint64_t &begin, uint64_t &end);

// `dw gibbs --gpus N` / `-c N`: returns the process exit code
int gibbs_multi(const CmdLine &args);

}  // namespace dw
#endif
