// host_wave.h -- the 64-lane host "wave" of the emulators (TEST INFRASTRUCTURE ONLY): one fiber (ucontext) per lane on ONE thread,
// every cross-lane operation a barrier at which the lane yields to the next one.  Lanes run round-robin from barrier to barrier: all
// lanes of a wave pass the same sequence of barriers, which is all a barrier promises.
#ifndef TRACY_AMD_TESTS_HOST_WAVE_H
#define TRACY_AMD_TESTS_HOST_WAVE_H
#include <ucontext.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <type_traits>
#include <vector>

namespace {
struct WaveShared {
  struct Barrier {
    WaveShared* w;
    void arrive_and_wait() { w->yield(); }
  } bar{this};
  int32_t xchg[64];
  std::vector<char> lds;
  // the 64 fibers
  static constexpr size_t kStack = size_t(1) << 19;
  ucontext_t sched{}, fib[64];
  static std::vector<char>& stack_pool() { static std::vector<char> p(64 * kStack); return p; }  // one wave at a time: allocated (and faulted in) once
  std::function<void(uint32_t)> body;
  bool done[64];
  int cur = -1;
  WaveShared() = default;
  WaveShared(const WaveShared&) = delete;
  void yield() { swapcontext(&fib[cur], &sched); }
  static void trampoline(unsigned lo, unsigned hi) {
    WaveShared* w = reinterpret_cast<WaveShared*>(((uintptr_t)hi << 32) | (uintptr_t)lo);
    const int l = w->cur;
    w->body((uint32_t)l);
    w->done[l] = true;  // returning resumes uc_link = the scheduler
  }
  template <class F>
  void run(F&& f) {
    body = std::forward<F>(f);
    std::vector<char>& stacks = stack_pool();
    const uintptr_t self = reinterpret_cast<uintptr_t>(this);
    for (int l = 0; l < 64; ++l) {
      done[l] = false;
      getcontext(&fib[l]);
      fib[l].uc_stack.ss_sp = stacks.data() + (size_t)l * kStack;
      fib[l].uc_stack.ss_size = kStack;
      fib[l].uc_link = &sched;
      makecontext(&fib[l], reinterpret_cast<void (*)()>(&WaveShared::trampoline), 2, (unsigned)(self & 0xffffffffu), (unsigned)(self >> 32));
    }
    for (bool any = true; any;) {
      any = false;
      for (int l = 0; l < 64; ++l) {
        if (done[l]) continue;
        cur = l;
        swapcontext(&sched, &fib[l]);
        any = any || !done[l];
      }
    }
  }
};

struct HostWave {
  uint32_t lane_;
  WaveShared* sh;
  uint32_t lane() const { return lane_; }
  int32_t shift_up(int32_t x) {  // lane L receives lane L-1's value (DPP wave_shr:1); lane 0 gets 0
    sh->xchg[lane_] = x;
    sh->bar.arrive_and_wait();
    int32_t r = lane_ ? sh->xchg[lane_ - 1] : 0;
    sh->bar.arrive_and_wait();
    return r;
  }
  int32_t shift_up_row(int32_t x) {  // the same inside a row of sixteen lanes (DPP row_shr:1): the first lane of a row gets 0
    sh->xchg[lane_] = x;
    sh->bar.arrive_and_wait();
    int32_t r = (lane_ & 15u) ? sh->xchg[lane_ - 1] : 0;
    sh->bar.arrive_and_wait();
    return r;
  }
  int32_t shift_up_or(int32_t x, int32_t first) {  // as shift_up, lane 0 keeps `first` (the DPP `old` operand)
    const int32_t r = shift_up(x);
    return lane_ ? r : first;
  }
  int32_t rot16(int32_t x) {  // lane j of a row of 16 receives lane j - 1 of the same row, lane 0 lane 15 (DPP row_ror:1)
    sh->xchg[lane_] = x;
    sh->bar.arrive_and_wait();
    const int32_t r = sh->xchg[(lane_ & ~15u) | ((lane_ - 1u) & 15u)];
    sh->bar.arrive_and_wait();
    return r;
  }
  int32_t rot(int32_t x, std::integral_constant<int, 16>) { return rot16(x); }
  int32_t rot(int32_t x, std::integral_constant<int, 4>) {  // the same inside a quad (DPP quad_perm:[3,0,1,2])
    sh->xchg[lane_] = x;
    sh->bar.arrive_and_wait();
    const int32_t r = sh->xchg[(lane_ & ~3u) | ((lane_ - 1u) & 3u)];
    sh->bar.arrive_and_wait();
    return r;
  }
  uint64_t ballot(bool p) {
    sh->xchg[lane_] = p ? 1 : 0;
    sh->bar.arrive_and_wait();
    uint64_t m = 0;
    for (int l = 0; l < 64; ++l) m |= (uint64_t)(sh->xchg[l] & 1) << l;
    sh->bar.arrive_and_wait();
    return m;
  }
  uint32_t bcast(uint32_t x, uint32_t src) {
    sh->xchg[lane_] = (int32_t)x;
    sh->bar.arrive_and_wait();
    const uint32_t r = (uint32_t)sh->xchg[src & 63];
    sh->bar.arrive_and_wait();
    return r;
  }
  // reductions / scan over the 64 lanes (decompose_wave.h)
  template <class F>
  uint32_t reduce(uint32_t x, F f) {
    sh->xchg[lane_] = (int32_t)x;
    sh->bar.arrive_and_wait();
    uint32_t r = (uint32_t)sh->xchg[0];
    for (int l = 1; l < 64; ++l) r = f(r, (uint32_t)sh->xchg[l]);
    sh->bar.arrive_and_wait();
    return r;
  }
  uint32_t sum(uint32_t x) { return reduce(x, [](uint32_t a, uint32_t b) { return a + b; }); }
  uint32_t umin(uint32_t x) { return reduce(x, [](uint32_t a, uint32_t b) { return a < b ? a : b; }); }
  uint32_t umax(uint32_t x) { return reduce(x, [](uint32_t a, uint32_t b) { return a > b ? a : b; }); }
  uint32_t excl_sum(uint32_t x) {  // sum of the lanes below
    sh->xchg[lane_] = (int32_t)x;
    sh->bar.arrive_and_wait();
    uint32_t r = 0;
    for (uint32_t l = 0; l < lane_; ++l) r += (uint32_t)sh->xchg[l];
    sh->bar.arrive_and_wait();
    return r;
  }
  void sync() { sh->bar.arrive_and_wait(); }
  void sync_global() { sh->bar.arrive_and_wait(); }
  char* lds() { return sh->lds.data(); }
};
}  // namespace
#endif
