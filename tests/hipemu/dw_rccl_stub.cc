// TEST-ONLY: dw_emu (the `dw` host program over the emulated library) has no device and no
// RCCL; its multi-rank runs use the host-staged communicator (--comm host).
#include <stdexcept>

#include "dw_multi.h"

namespace dw {
std::unique_ptr<Comm> make_rccl_comm(const std::vector<int> &) {
  throw std::runtime_error("this test build has no RCCL: run with --comm host");
}
}  // namespace dw
