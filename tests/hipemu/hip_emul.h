// hip_emul.h -- TEST-ONLY host emulation of the handful of HIP constructs the dwx
// kernels use, so that sampler_amd/csrc/sweep_kernels.h (the product's kernel source,
// unmodified) can be compiled by g++ and run under ASan/UBSan without a GPU.
//
// This is NOT a CPU backend of the product: nothing under sampler_amd/ includes or
// loads it, the product library (libdwx.so) is always the hipcc/gfx950 build and
// fails loudly without a device.  A workgroup is emulated by ucontext fibers (one per
// thread); __syncthreads() yields to a round-robin scheduler, which is a correct
// barrier for non-divergent barriers.  LDS is a freshly poisoned buffer per block.
#ifndef DWX_HIP_EMUL_H_
#define DWX_HIP_EMUL_H_

#include <stdint.h>

#include <cmath>
#include <cstddef>
#include <functional>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace emu {
extern dim3 threadIdx_, blockIdx_, blockDim_, gridDim_;
extern unsigned char *g_lds;
void syncthreads();
// wave64 ballot; must be reached by every thread of the workgroup
unsigned long long ballot(bool pred);
// sum over the 64 lanes of the calling thread's wave in the xor-butterfly association of the
// device code (wave_sum_f64); only the lanes of ONE wave have to reach it together
double wave_sum(double v);
// per-lane sum of `acc` from the lane to the end of its run of equal keys, and whether the lane
// heads its run; the 64 lanes of a wave have to reach it together
long long wave_seg_sum(unsigned key, long long acc, bool &head);
// the value of lane ^ 1; the two lanes of a pair have to reach it together
unsigned pair_swap(unsigned v);
// run `body` once per (block, thread) of the grid, blocks sequentially
void run_grid(unsigned grid, unsigned block, size_t lds_bytes, const std::function<void()> &body);
}  // namespace emu

#define threadIdx (::emu::threadIdx_)
#define blockIdx (::emu::blockIdx_)
#define blockDim (::emu::blockDim_)
#define gridDim (::emu::gridDim_)

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define DWX_DEV inline
#define DWX_DYN_LDS(name) unsigned char *name = ::emu::g_lds
#define DWX_FAST_EXPF(x) expf(x)
#define __shared__ static
#define DWX_BALLOT(pred) (::emu::ballot(pred))
#define DWX_WAVE_SUM_F64(v) (::emu::wave_sum(v))
#define DWX_PAIR_SWAP_U32(v) (::emu::pair_swap(v))
#define DWX_WAVE_SEG_SUM_I64(key, acc, head) (::emu::wave_seg_sum(key, acc, head))
#define DWX_UNIFORM(x) (x)
#define DWX_NT_LOAD(p) (*(p))
#define DWX_LOAD_ROW_NT(p) (*(p))
#define DWX_NT_STORE(v, p) (*(p) = (v))

// in-launch hand-off words (the harness runs a grid's blocks one after another: a kernel with a grid
// barrier is launched with ONE block here, rt::grid_barrier_blocks())
#define DWX_AGENT_STORE_I64(p, v) (*(p) = (long long)(v))
#define DWX_AGENT_STORE_U32(p, v) (*(p) = (uint32_t)(v))
#define DWX_AGENT_LOAD_U32(p) (*(p))
#define DWX_AGENT_ADD_U32(p, v) (atomicAdd((unsigned *)(p), (unsigned)(v)))
#define DWX_DRAIN_VMEM() ((void)0)
#define DWX_ACQUIRE_AGENT() ((void)0)
#define DWX_SLEEP() ((void)0)

inline void __syncthreads() { ::emu::syncthreads(); }

inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) {
  unsigned long long old = *p;
  *p = old + v;
  return old;
}

inline unsigned atomicAdd(unsigned *p, unsigned v) {
  unsigned old = *p;
  *p = old + v;
  return old;
}

using std::exp;

using std::log1p;
using std::log2;
using std::pow;
using std::llrint;

// host stand-in for the buffer-descriptor record stream (zero-fills past the end, like
// the hardware bounds check)
#include "device_types.h"
template <int K>
inline void emu_load_tile_records(const dwx::EdgeRec *base, uint32_t nedges, uint32_t t,
                                  dwx::EdgeRec (&rec)[K]) {
  for (int k = 0; k < K; ++k) {
    const uint32_t i = t + k * dwx::BLOCK_THREADS;
    rec[k] = i < nedges ? base[i] : dwx::EdgeRec{0u, 0u, 0u, 0.0f};
  }
}
#define DWX_LOAD_TILE_RECORDS(K, base, nedges, t, rec) emu_load_tile_records<K>(base, nedges, t, rec)
template <int K>
inline void emu_load_tile_records8(const dwx::EdgeRec8 *base, uint32_t nedges, uint32_t t,
                                   dwx::EdgeRec8 (&rec)[K]) {
  for (int k = 0; k < K; ++k) {
    const uint32_t i = t + k * dwx::BLOCK_THREADS;
    rec[k] = i < nedges ? base[i] : dwx::EdgeRec8{0u, 0.0f};
  }
}
#define DWX_LOAD_TILE_RECORDS8(K, base, nedges, t, rec) emu_load_tile_records8<K>(base, nedges, t, rec)
template <int K>
inline void emu_load_sorted_records(const dwx::SortRec8 *base, uint32_t nrec, uint32_t first, uint32_t t,
                                    dwx::SortRec8 (&rec)[K]) {
  for (int k = 0; k < K; ++k) {
    const uint64_t i = (uint64_t)first + t + (uint64_t)k * dwx::SORT_THREADS;
    rec[k] = i < nrec ? base[i] : dwx::SortRec8{0u, 0u};
  }
}
#define DWX_LOAD_SORTED_RECORDS(K, base, nrec, first, t, rec) emu_load_sorted_records<K>(base, nrec, first, t, rec)

#endif
