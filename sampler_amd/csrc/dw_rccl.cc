// dw_rccl.cc -- the product communicator of `dw gibbs --gpus N`: RCCL (rccl.h) over xGMI.
// One communicator per rank, created together (ncclCommInitAll: the ranks are host threads of
// one process, one GPU each); every collective is enqueued on the calling rank's sampler stream,
// so kernels and collectives are ordered on the device without host synchronisation.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <stdexcept>
#include <string>

#include "dw_multi.h"

namespace dw {
namespace {
void nccl_ok(ncclResult_t r, const char *what) {
  if (r != ncclSuccess) throw std::runtime_error(std::string("RCCL error in ") + what + ": " + ncclGetErrorString(r));
}
#define DW_NCCL(x) nccl_ok((x), #x)

class RcclComm : public Comm {
 public:
  explicit RcclComm(const std::vector<int> &devices) : devices_(devices), comms_(devices.size()) {
    for (size_t i = 0; i < devices.size(); ++i)
      for (size_t j = i + 1; j < devices.size(); ++j)
        if (devices[i] == devices[j])
          throw std::runtime_error("RCCL needs one GPU per rank (device " + std::to_string(devices[i]) +
                                   " listed twice); --comm host stacks ranks on one device for tests");
    DW_NCCL(ncclCommInitAll(comms_.data(), (int)devices.size(), devices.data()));
  }
  ~RcclComm() override {
    if (aborted_.load()) return;          // (ncclCommAbort freed them)
    for (ncclComm_t c : comms_) if (c) (void)ncclCommDestroy(c);
  }
  // ncclCommAbort makes the operations in flight on the communicator fail instead of waiting for
  // a peer that died: the surviving ranks' next dwx_wait / dwx_get_weights returns, their threads
  // end, `dw` exits 1 with the first error (the watchdog in gibbs_multi covers a runtime that
  // does not come back even then).
  void abort() override {
    if (aborted_.exchange(true)) return;
    for (size_t i = 0; i < comms_.size(); ++i)
      if (comms_[i]) { (void)hipSetDevice(devices_[i]); (void)ncclCommAbort(comms_[i]); }
  }
  const char *name() const override { return "RCCL"; }
  void allreduce_sum_i64(int rank, dwx_sampler *s, void *dev, uint64_t n) override { reduce(rank, s, dev, n, ncclInt64); }
  void allreduce_sum_f64(int rank, dwx_sampler *s, void *dev, uint64_t n) override { reduce(rank, s, dev, n, ncclFloat64); }
  void allreduce_sum_u32(int rank, dwx_sampler *s, void *dev, uint64_t n) override { reduce(rank, s, dev, n, ncclUint32); }
  void exchange(int rank, dwx_sampler *s, const std::vector<Xfer> &sends, const std::vector<Xfer> &recvs) override {
    if (sends.empty() && recvs.empty()) return;
    hipStream_t st = stream_of(rank, s);
    DW_NCCL(ncclGroupStart());
    for (const Xfer &x : sends) DW_NCCL(ncclSend(x.dev, x.nbytes, ncclUint8, x.peer, comms_[rank], st));
    for (const Xfer &x : recvs) DW_NCCL(ncclRecv(x.dev, x.nbytes, ncclUint8, x.peer, comms_[rank], st));
    DW_NCCL(ncclGroupEnd());
  }

 private:
  hipStream_t stream_of(int rank, dwx_sampler *s) {
    void *st = nullptr;
    if (dwx_stream(s, &st) != DWX_OK) throw std::runtime_error(std::string("dwx: ") + dwx_last_error());
    if (hipSetDevice(devices_[rank]) != hipSuccess) throw std::runtime_error("hipSetDevice failed");
    return (hipStream_t)st;
  }
  void reduce(int rank, dwx_sampler *s, void *dev, uint64_t n, ncclDataType_t t) {
    hipStream_t st = stream_of(rank, s);
    DW_NCCL(ncclAllReduce(dev, dev, n, t, ncclSum, comms_[rank], st));
  }
  std::vector<int> devices_;
  std::vector<ncclComm_t> comms_;
  std::atomic<bool> aborted_{false};
};
}  // namespace

std::unique_ptr<Comm> make_rccl_comm(const std::vector<int> &devices) {
  return std::unique_ptr<Comm>(new RcclComm(devices));
}
}  // namespace dw
