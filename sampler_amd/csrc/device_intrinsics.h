// device_intrinsics.h -- the gfx950 forms of the handful of hardware primitives the kernels are
// written against: address-space qualifiers, non-temporal accesses, buffer-descriptor record
// streams, wave ballots / shuffles, the dynamic-LDS symbol, v_exp_f32.
//
// This file is the ONE seam of the test harness: tests/hipemu (a sanitizer harness that compiles
// the kernel sources for the host, never a backend of the product) defines DWX_EMU and provides
// the same names from tests/hipemu/hip_emul.h + rt_emu.h instead.  No other product header tests
// for the emulation; everything else spelled #ifndef DWX_... in the sources is a compile-time
// tuning constant (tools/variant.sh builds A/B variants with -D).
#ifndef DWX_DEVICE_INTRINSICS_H_
#define DWX_DEVICE_INTRINSICS_H_

#include "device_types.h"

#ifdef DWX_EMU
#include "rt_emu.h"      // (pulls in hip_emul.h: every name below, for the host)
#else
#include "rt_hip.h"

#define DWX_DEV __device__ __forceinline__
#define DWX_DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
// workgroup-uniform values into scalar registers; wave64 ballot
#define DWX_UNIFORM(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#define DWX_BALLOT(pred) ((unsigned long long)__ballot(pred))
// f32 exp for the guarded fast draws (v_exp_f32; error ~1e-6 relative for |x| < 30)
#define DWX_FAST_EXPF(x) __expf(x)

// Streamed-once loads / stores (per-variable words, row pointers, assignments of an all-unary
// graph): non-temporal, so that they do not evict the re-used f32 weight table from L2.
#ifdef DWX_NO_NT_META     // (tuning experiment)
#define DWX_NT_LOAD(p) (*(p))
#define DWX_NT_STORE(v, p) (*(p) = (v))
#else
#define DWX_NT_LOAD(p) __builtin_nontemporal_load(p)
#define DWX_NT_STORE(v, p) __builtin_nontemporal_store((v), (p))
#endif

// In-launch hand-off between workgroups (persist_learn8_kernel's chunk barrier; the CDNA guide's
// Guideline 16, recipe R1): payload stored WRITE-THROUGH by agent-scope relaxed atomic stores (global
// address space: global_store ... sc1), every storing wave drains, ONE lane signals with an agent-scope
// atomic add; the consumer polls ONE word relaxed, then ONE agent-scope acquire (+ its wait) before
// the workgroup's barrier and the plain loads.
#define DWX_GLOBAL_PTR(T, p) ((__attribute__((address_space(1))) T *)(p))
#define DWX_AGENT_STORE_I64(p, v) __hip_atomic_store(DWX_GLOBAL_PTR(long long, p), (long long)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define DWX_AGENT_STORE_U32(p, v) __hip_atomic_store(DWX_GLOBAL_PTR(uint32_t, p), (uint32_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define DWX_AGENT_LOAD_U32(p) __hip_atomic_load(DWX_GLOBAL_PTR(uint32_t, p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define DWX_AGENT_ADD_U32(p, v) __hip_atomic_fetch_add(DWX_GLOBAL_PTR(uint32_t, p), (uint32_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define DWX_DRAIN_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define DWX_ACQUIRE_AGENT() do { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } while (0)
#define DWX_SLEEP() __builtin_amdgcn_s_sleep(2)

namespace dwx {
typedef uint32_t dwx_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t dwx_u32x2 __attribute__((ext_vector_type(2)));

// a 16-byte row of the block-pull tables, read once per sweep: non-temporal
DWX_DEV U32x4 load_row_nt(const U32x4 *p) {
  const dwx_u32x4 v = __builtin_nontemporal_load((const dwx_u32x4 *)p);
  U32x4 r;
  r.v[0] = v.x; r.v[1] = v.y; r.v[2] = v.z; r.v[3] = v.w;
  return r;
}

// The tile's edge records: lane t takes records t, t + 256, ...  (16 B per lane,
// consecutive lanes -> consecutive records: one coalesced stream).  Read through a
// buffer descriptor of exactly the tile's range: the hardware bounds check returns
// zeros for lanes past the last record (no clamping arithmetic, no branch, no memory
// traffic), and the K loads differ only in their scalar offset, so they cost no
// per-load address VALU.  "nt": the stream is read once per sweep and must not evict
// the re-used f32 weight table from the XCD's L2.
template <int K>
DWX_DEV void load_tile_records(const EdgeRec *base, uint32_t nedges, uint32_t t, EdgeRec (&rec)[K]) {
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)(nedges * sizeof(EdgeRec)), 0x00020000);
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const dwx_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(
        rsrc, (int)(t * sizeof(EdgeRec)), (int)(k * BLOCK_THREADS * sizeof(EdgeRec)), /*nt*/ 2);
    const uint32_t fw = v.w;
    float f;
    __builtin_memcpy(&f, &fw, 4);
    rec[k].wid = v.x; rec[k].aux = v.y; rec[k].packed = v.z; rec[k].fval = f;
  }
}
// The same stream for the 8-byte records of an all-TILE_SIMPLE graph (buffer_load_dwordx2).
template <int K>
DWX_DEV void load_tile_records8(const EdgeRec8 *base, uint32_t nedges, uint32_t t, EdgeRec8 (&rec)[K]) {
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)(nedges * sizeof(EdgeRec8)), 0x00020000);
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const dwx_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(
        rsrc, (int)(t * sizeof(EdgeRec8)), (int)(k * BLOCK_THREADS * sizeof(EdgeRec8)), /*nt*/ 2);
    const uint32_t fw = v.y;
    float f;
    __builtin_memcpy(&f, &fw, 4);
    rec[k].key = v.x; rec[k].f = f;
  }
}
// ... and for the weight-sorted records of a super-tile: records first + t, first + t + SORT_THREADS, ...
template <int K>
DWX_DEV void load_sorted_records(const SortRec8 *base, uint32_t nrec, uint32_t first, uint32_t t, SortRec8 (&rec)[K]) {
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)(nrec * sizeof(SortRec8)), 0x00020000);
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const dwx_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(
        rsrc, (int)(t * sizeof(SortRec8)), (int)((first + k * SORT_THREADS) * sizeof(SortRec8)), /*nt*/ 2);
    rec[k].wid = v.x; rec[k].od = v.y;
  }
}

// Sums `acc` over the runs of equal `key` among the 64 lanes of a wave (equal keys sit in
// neighbouring lanes); every lane gets the sum from itself to the end of its run, `head` says
// whether it is the first lane of its run.  All 64 lanes call together.
DWX_DEV long long wave_seg_sum_i64(uint32_t key, long long acc, bool &head) {
  const uint32_t lane = threadIdx.x & 63u;
  // (no two neighbouring lanes share a key -- lightly tied weights: every lane heads its own run)
  const uint32_t nk = (uint32_t)__shfl_down((int)key, 1, 64);
  if (__ballot(lane < 63u && nk == key) == 0ull) { head = true; return acc; }
#pragma unroll
  for (uint32_t off = 1; off < 64u; off <<= 1) {
    const uint32_t ok = (uint32_t)__shfl_down((int)key, off, 64);
    const long long oa = __shfl_down(acc, off, 64);
    if (lane + off < 64u && ok == key) acc += oa;
  }
  const uint32_t pk = (uint32_t)__shfl_up((int)key, 1, 64);
  head = lane == 0u || pk != key;
  return acc;
}
// sum over the 64 lanes of a wave, the same value (and the same association: the xor
// butterfly) in every lane
DWX_DEV double wave_sum_f64(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
}  // namespace dwx
#define DWX_LOAD_TILE_RECORDS(K, base, nedges, t, rec) load_tile_records<K>(base, nedges, t, rec)
#define DWX_LOAD_TILE_RECORDS8(K, base, nedges, t, rec) load_tile_records8<K>(base, nedges, t, rec)
#define DWX_LOAD_SORTED_RECORDS(K, base, nrec, first, t, rec) load_sorted_records<K>(base, nrec, first, t, rec)
#define DWX_LOAD_ROW_NT(p) load_row_nt(p)
#define DWX_WAVE_SEG_SUM_I64(key, acc, head) wave_seg_sum_i64(key, acc, head)
#define DWX_WAVE_SUM_F64(v) wave_sum_f64(v)
// the value of the neighbouring lane (lane ^ 1), both lanes of the pair calling together
#define DWX_PAIR_SWAP_U32(v) ((uint32_t)__shfl_xor((int)(v), 1, 64))
#endif   // DWX_EMU

#endif  // DWX_DEVICE_INTRINSICS_H_
