// hip_emul.cc -- fiber scheduler of the TEST-ONLY HIP emulation (see hip_emul.h).
#include "hip_emul.h"

#include <ucontext.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <vector>

#if defined(__SANITIZE_ADDRESS__)
#include <sanitizer/common_interface_defs.h>
#define EMU_ASAN 1
#else
#define EMU_ASAN 0
#endif

namespace emu {

dim3 threadIdx_, blockIdx_, blockDim_, gridDim_;
unsigned char *g_lds = nullptr;

namespace {
constexpr size_t kStack = 256 * 1024;
struct Fiber {
  ucontext_t ctx;
  char *stack = nullptr;
  bool done = false;
};
ucontext_t g_sched;
std::vector<Fiber> g_fibers;
Fiber *g_cur = nullptr;
const std::function<void()> *g_body = nullptr;
unsigned g_bar_arrived = 0, g_bar_gen = 0;
#if EMU_ASAN
void *g_sched_fake = nullptr;
const void *g_sched_bottom = nullptr;
size_t g_sched_size = 0;
#endif

void to_sched(bool dying) {
#if EMU_ASAN
  void *fake = nullptr;
  __sanitizer_start_switch_fiber(dying ? nullptr : &fake, g_sched_bottom, g_sched_size);
#endif
  Fiber *self = g_cur;
  swapcontext(&self->ctx, &g_sched);
#if EMU_ASAN
  __sanitizer_finish_switch_fiber(fake, &g_sched_bottom, &g_sched_size);
#endif
  (void)dying;
}

void entry() {
#if EMU_ASAN
  __sanitizer_finish_switch_fiber(nullptr, &g_sched_bottom, &g_sched_size);
#endif
  (*g_body)();
  g_cur->done = true;
  to_sched(true);
}
}  // namespace

// A counting barrier: a fiber waits until every LIVE fiber of the workgroup has arrived (the
// scheduler opens it; fibers that returned do not count).  Yields for other reasons -- a lane
// waiting for its wave or its pair partner -- therefore never pass for an arrival.
void syncthreads() {
  const unsigned my_gen = g_bar_gen;
  ++g_bar_arrived;
  while (g_bar_gen == my_gen) to_sched(false);
}

unsigned long long ballot(bool pred) {
  static unsigned char preds[1024];
  preds[threadIdx_.x] = pred ? 1 : 0;
  syncthreads();
  const unsigned w0 = threadIdx_.x & ~63u;
  unsigned long long m = 0;
  for (unsigned i = 0; i < 64 && w0 + i < blockDim_.x; ++i)
    if (preds[w0 + i]) m |= 1ull << i;
  syncthreads();
  return m;
}

// Wave-level exchange without a workgroup barrier: a lane deposits its value and yields until
// the 64th lane of its wave has arrived (the waves of a workgroup may be in different places).
double wave_sum(double v) {
  static double slot[16][64], result[16];
  static unsigned arrived[16], gen[16];
  const unsigned w = threadIdx_.x >> 6, l = threadIdx_.x & 63u;
  const unsigned lanes = blockDim_.x - w * 64 < 64 ? blockDim_.x - w * 64 : 64;
  slot[w][l] = v;
  const unsigned my_gen = gen[w];
  if (++arrived[w] == lanes) {
    double x[64];
    for (unsigned i = 0; i < 64; ++i) x[i] = i < lanes ? slot[w][i] : 0.0;
    for (unsigned m = 32; m >= 1; m >>= 1) {      // the device's butterfly, lane 0's view
      double y[64];
      for (unsigned i = 0; i < 64; ++i) y[i] = x[i] + x[i ^ m];
      for (unsigned i = 0; i < 64; ++i) x[i] = y[i];
    }
    result[w] = x[0];
    arrived[w] = 0;
    ++gen[w];
  }
  while (gen[w] == my_gen) to_sched(false);
  return result[w];
}

long long wave_seg_sum(unsigned key, long long acc, bool &head) {
  static unsigned keys[16][64], arrived[16], gen[16];
  static long long accs[16][64], sums[2][16][64];
  static bool heads[2][16][64];
  const unsigned w = threadIdx_.x >> 6, l = threadIdx_.x & 63u;
  const unsigned lanes = blockDim_.x - w * 64 < 64 ? blockDim_.x - w * 64 : 64;
  keys[w][l] = key; accs[w][l] = acc;
  const unsigned my_gen = gen[w];
  if (++arrived[w] == lanes) {
    for (unsigned i = 0; i < lanes; ++i) {
      long long t = 0;
      for (unsigned j = i; j < lanes && keys[w][j] == keys[w][i]; ++j) t += accs[w][j];
      sums[my_gen & 1u][w][i] = t;
      heads[my_gen & 1u][w][i] = i == 0 || keys[w][i - 1] != keys[w][i];
    }
    arrived[w] = 0;
    ++gen[w];
  }
  while (gen[w] == my_gen) to_sched(false);
  head = heads[my_gen & 1u][w][l];
  return sums[my_gen & 1u][w][l];
}

// A pair exchange (the device's __shfl_xor(v, 1)): deposit, yield until the partner has arrived.
// Two slots per lane, by generation: the partner may run ahead into its next exchange.
unsigned pair_swap(unsigned v) {
  static unsigned slot[2][1024], arrived[512], gen[512];
  const unsigned t = threadIdx_.x, pr = t >> 1;
  const unsigned my_gen = gen[pr];
  slot[my_gen & 1u][t] = v;
  if (++arrived[pr] == 2) {
    arrived[pr] = 0;
    ++gen[pr];
  }
  while (gen[pr] == my_gen) to_sched(false);
  return slot[my_gen & 1u][t ^ 1u];
}

void run_grid(unsigned grid, unsigned block, size_t lds_bytes, const std::function<void()> &body) {
  // one launch at a time: the emulated "device" state above is global, and the multi-rank host
  // drivers (dw gibbs --gpus N: one host thread per rank) launch from several threads
  static std::mutex one_launch;
  std::lock_guard<std::mutex> hold(one_launch);
  if (g_fibers.size() < block) {
    size_t old = g_fibers.size();
    g_fibers.resize(block);
    for (size_t i = old; i < block; ++i) g_fibers[i].stack = (char *)malloc(kStack);
  }
  gridDim_ = dim3(grid);
  blockDim_ = dim3(block);
  g_body = &body;
  unsigned char *lds = (unsigned char *)malloc(lds_bytes ? lds_bytes : 16);
  for (unsigned b = 0; b < grid; ++b) {
    memset(lds, 0xA5, lds_bytes);  // LDS is uninitialised on the device
    g_lds = lds;
    blockIdx_ = dim3(b);
    for (unsigned t = 0; t < block; ++t) {
      Fiber &f = g_fibers[t];
      f.done = false;
      getcontext(&f.ctx);
      f.ctx.uc_stack.ss_sp = f.stack;
      f.ctx.uc_stack.ss_size = kStack;
      f.ctx.uc_link = nullptr;
      makecontext(&f.ctx, (void (*)())entry, 0);
    }
    unsigned alive = block;
    g_bar_arrived = 0;
    while (alive) {
      for (unsigned t = 0; t < block; ++t) {
        Fiber &f = g_fibers[t];
        if (f.done) continue;
        threadIdx_ = dim3(t);
        g_cur = &f;
#if EMU_ASAN
        void *fake = nullptr;
        __sanitizer_start_switch_fiber(&fake, f.stack, kStack);
#endif
        swapcontext(&g_sched, &f.ctx);
#if EMU_ASAN
        __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#endif
        if (f.done) --alive;
        if (g_bar_arrived && g_bar_arrived >= alive) { g_bar_arrived = 0; ++g_bar_gen; }
      }
    }
  }
  free(lds);
  g_lds = nullptr;
  g_body = nullptr;
}

}  // namespace emu
