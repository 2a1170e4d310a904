// device_build_stub.cc -- TEST-ONLY: the host harness has no device to build on; the host builders
// of graph_compile.cc / dwx_api.cc run instead (they are also the checker of the device build).
#include <stdexcept>

#include "device_build.h"

namespace dwx {
namespace devb {
bool available() { return false; }
void release_scratch(int, bool) {}
uint64_t scratch_bytes(int) { return 0; }
void build_sorted_records(const TileDesc *, const EdgeRec *, const EdgeRec8 *, const SuperTile *, uint32_t,
                          const uint32_t *, uint32_t, uint64_t, SortRec8 *, void *) {
  throw std::runtime_error("device build is not available in the host harness");
}
void build_static_tables(const TileDesc *, uint32_t, const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *,
                         const EdgeRec *, const double *, const uint8_t *, uint32_t, uint32_t, bool, bool, long long *,
                         long long *, long long *, void *) {
  throw std::runtime_error("device build is not available in the host harness");
}
void build_incidence(const TileDesc *, const TileDesc *, uint32_t, const uint32_t *, const uint32_t *, const uint32_t *,
                     const uint32_t *, const EdgeRec *, bool, bool, uint32_t, uint32_t, uint64_t, uint32_t, Incidence &, void *) {
  throw std::runtime_error("device build is not available in the host harness");
}
CurvatureScratch::~CurvatureScratch() {}
double batch_curvature(uint32_t, uint32_t, const uint32_t *, const uint32_t *, const uint32_t *, const EdgeRec *, const double *,
                       bool, bool, uint32_t, CurvatureScratch &, void *) {
  throw std::runtime_error("device build is not available in the host harness");
}
}  // namespace devb
}  // namespace dwx
