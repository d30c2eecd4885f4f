// capi.hip -- the extern "C" boundary (include/tracy_hip.h): argument checking, staging, bucketing of
// pairs by strip height K, workspace chunking, kernel launches.  No compute happens on the host and
// there is no CPU fallback: every entry point needs a gfx950 device.
#include <hip/hip_runtime.h>
#include <atomic>
#include <sched.h>
#include <malloc.h>
#include <map>
#include <chrono>

#include <thread>

#include <algorithm>
#include <cctype>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <memory>
#include <vector>

#include "../../include/tracy_hip.h"
#include "capi_internal.h"
#include "launch.h"

using namespace tracyhip;

namespace tracyhip {

static thread_local std::string g_last_error;

int set_error(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

hipError_t DevBuf::ensure(size_t bytes) {
  if (bytes <= cap) return hipSuccess;
  if (p) {
    hipError_t e = hipFree(p);
    p = nullptr;
    cap = 0;
    if (e != hipSuccess) return e;
  }
  size_t want = bytes + bytes / 8 + 256;
  hipError_t e = hipMalloc(&p, want);
  if (e != hipSuccess) {  // retry with the exact size before giving up
    (void)hipGetLastError();
    want = bytes;
    e = hipMalloc(&p, want);
  }
  if (e == hipSuccess) cap = want;
  else p = nullptr;
  return e;
}
void DevBuf::release() {
  if (p) (void)hipFree(p);
  p = nullptr;
  cap = 0;
}
hipError_t PinBuf::ensure(size_t bytes) {
  if (bytes <= cap) return hipSuccess;
  if (p) (void)hipHostFree(p);
  p = nullptr;
  cap = 0;
  size_t want = bytes + bytes / 4 + 256;
  hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
  if (e == hipSuccess) cap = want;
  else p = nullptr;
  return e;
}
void PinBuf::release() {
  if (p) (void)hipHostFree(p);
  p = nullptr;
  cap = 0;
}

// smallest P*K over the strip heights compiled for the mode; ties go to the taller strip
int choose_k(uint32_t m, int mode, bool needle) {
  static const int ks_all[] = {16, 15, 12, 8, 4};
  static const int ks_needle[] = {16, 8, 4};
  static const int ks_prof[] = {8, 4};
  const int* ks = (mode == MODE_PROF) ? ks_prof : needle ? ks_needle : ks_all;
  const int nk = (mode == MODE_PROF) ? 2 : needle ? 3 : 5;
  int best = ks[0];
  uint64_t best_cost = ~0ull;
  for (int i = 0; i < nk; ++i) {
    const uint64_t cost = (uint64_t)num_passes(m ? m : 1, ks[i]) * ks[i];
    if (cost < best_cost) { best_cost = cost; best = ks[i]; }
  }
  return best;
}

uint64_t seqset_extent(const tracyhip_seqset& s) {
  uint64_t ext = 0;
  const uint64_t mult = (s.kind == TRACYHIP_SEQ_PROFILE) ? 6 : 1;
  for (uint32_t i = 0; i < s.count; ++i) ext = std::max<uint64_t>(ext, s.offset[i] + mult * s.length[i]);
  return ext;
}

}  // namespace tracyhip

// ---- small device helpers ------------------------------------------------------------------------
__global__ void encode_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n, uint8_t* __restrict__ special) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const uint32_t c = dp_code(in[i]);
    out[i] = (uint8_t)c;
    if (c >= 4u) special[i >> 8] = 1;  // N, '-' / other: rare (same value from every writer)
  }
}

// profile x profile: is row 4 ('N') zero over a whole profile?  (NaN counts as non-zero.)  One wave per sequence.  colclass (a2 set
// only): the class of every column (dp_kernels.h column_class), stored at the index of the column's row-0 element.
struct Row4Desc { uint64_t off; uint32_t len, pad; };
__global__ __launch_bounds__(64) void row4_zero_kernel(const Row4Desc* __restrict__ d, const float* __restrict__ data, uint8_t* __restrict__ out,
                                                       uint8_t* __restrict__ colclass) {
  const Row4Desc s = d[blockIdx.x];
  bool nz = false;
  for (uint32_t j = threadIdx.x; j < s.len; j += 64) {
    nz |= !(data[s.off + 4ull * s.len + j] == 0.0f);
    if (colclass) colclass[s.off + j] = (uint8_t)tracyhip::column_class(data + s.off, s.len, j);
  }
  const unsigned long long any = __ballot(nz);
  if (threadIdx.x == 0) out[blockIdx.x] = any ? 0 : 1;
}

#define HIP_TRY(expr)                                                                               \
  do {                                                                                              \
    hipError_t _e = (expr);                                                                         \
    if (_e != hipSuccess)                                                                           \
      return set_error(_e == hipErrorOutOfMemory ? TRACYHIP_ERR_OOM : TRACYHIP_ERR_HIP, "%s failed: %s (%s:%d)", \
                       #expr, hipGetErrorString(_e), __FILE__, __LINE__);                           \
  } while (0)

namespace tracyhip {

static hipError_t get_event(tracyhip_ctx* ctx, hipEvent_t* e) {
  if (!ctx->free_events.empty()) { *e = ctx->free_events.back(); ctx->free_events.pop_back(); return hipSuccess; }
  return hipEventCreate(e);
}
namespace {
struct HostProfile {
  bool on = getenv("TRACYHIP_HOST_TIMERS") != nullptr;
  std::mutex m;
  std::map<std::string, std::pair<double, uint64_t>> acc;
  ~HostProfile() {
    if (!on) return;
    for (auto const& kv : acc) fprintf(stderr, "host %-28s %10.3f ms %8llu x\n", kv.first.c_str(), kv.second.first, (unsigned long long)kv.second.second);
  }
};
HostProfile& host_profile() { static HostProfile p; return p; }
inline uint64_t now_ns() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace
// ---- the persistent workers of parallel_for (capi_internal.h) ----
namespace {
struct HostPool {
  std::mutex use;  // one job at a time
  std::mutex m;
  std::condition_variable cv, done_cv;
  const std::function<void(uint32_t)>* job = nullptr;
  std::atomic<uint32_t> next{0};  // the next of the job's kHostThreads slices
  uint64_t gen = 0;
  uint32_t pending = 0;
  uint32_t nworkers = 0;  // threads besides the caller: TRACYHIP_HOST_THREADS - 1, or what the process may run on, at most kHostThreads - 1
  bool started = false;
  void slices(const std::function<void(uint32_t)>& f) {
    for (uint32_t i = next.fetch_add(1); i < kHostThreads; i = next.fetch_add(1)) f(i);
  }
  void worker() {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(uint32_t)>* j;
      {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return gen != seen; });
        seen = gen;
        j = job;
      }
      slices(*j);
      {
        std::lock_guard<std::mutex> lk(m);
        if (--pending == 0) done_cv.notify_one();
      }
    }
  }
  static uint32_t share() {
    uint32_t want = kHostThreads;
    if (const char* e = getenv("TRACYHIP_HOST_THREADS")) { const int v = atoi(e); if (v >= 1) want = (uint32_t)v; }
    else {
      cpu_set_t set;
      CPU_ZERO(&set);
      if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int c = CPU_COUNT(&set); if (c >= 1) want = (uint32_t)c; }
      // ... and a container's CPU quota (cgroup v2 cpu.max "quota period"; the command line's usable_threads() reads the same file)
      if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32] = {0};
        unsigned long period = 0;
        if (fscanf(f, "%31s %lu", quota, &period) == 2 && period > 0 && strcmp(quota, "max") != 0) {
          const unsigned long q = strtoul(quota, nullptr, 10) / period;
          if (q >= 1 && q < want) want = (uint32_t)q;
        }
        fclose(f);
      }
    }
    return want > kHostThreads ? kHostThreads : want;
  }
  bool run(const std::function<void(uint32_t)>& f) {
    std::unique_lock<std::mutex> u(use, std::try_to_lock);
    if (!u.owns_lock()) return false;
    if (!started) {
      // A job is always cut into kHostThreads slices (the callers keep per-slice partial results); how many threads work them off
      // is the process's share of the host: N ranks on one node must not start 8 N threads on cores they were not given.
      const uint32_t want = share();
      nworkers = want - 1;
      for (uint32_t t = 0; t < nworkers; ++t) std::thread([this] { worker(); }).detach();
      started = true;
    }
    {
      std::lock_guard<std::mutex> lk(m);
      job = &f;
      next.store(0);
      pending = nworkers;
      ++gen;
    }
    if (nworkers) cv.notify_all();
    slices(f);
    std::unique_lock<std::mutex> lk(m);
    done_cv.wait(lk, [&] { return pending == 0; });
    return true;
  }
};
}  // namespace
uint32_t host_pool_threads() {
  static const uint32_t n = HostPool::share();
  return n;
}
bool host_pool_run(const std::function<void(uint32_t)>& job) {
  static HostPool* pool = new HostPool;  // never destroyed: its detached workers may outlive static destructors
  return pool->run(job);
}

HostScope::HostScope(const char* l) : label(l), t0(host_profile().on ? now_ns() : 0) {}
HostScope::~HostScope() {
  HostProfile& p = host_profile();
  if (!p.on) return;
  const double ms = (double)(now_ns() - t0) * 1e-6;
  std::lock_guard<std::mutex> lk(p.m);
  auto& e = p.acc[label];
  e.first += ms;
  e.second += 1;
}

// ---- options (CtxKnobs) ----
namespace {
struct KnobField { const char* name; bool CtxKnobs::*flag; };
const KnobField kKnobFlags[] = {
    {"no_stream", &CtxKnobs::no_stream}, {"no_narrow", &CtxKnobs::no_narrow}, {"no_compact", &CtxKnobs::no_compact},
    {"no_screen", &CtxKnobs::no_screen}, {"no_band", &CtxKnobs::no_band}, {"no_band16", &CtxKnobs::no_band16},
    {"no_front", &CtxKnobs::no_front}, {"no_prefix", &CtxKnobs::no_prefix}, {"no_vote", &CtxKnobs::no_vote},
    {"no_origin", &CtxKnobs::no_origin}, {"no_subwindow", &CtxKnobs::no_subwindow}, {"no_prelim_origin", &CtxKnobs::no_prelim_origin},
    {"no_cq", &CtxKnobs::no_cq}, {"no_fused_walk", &CtxKnobs::no_fused_walk}, {"no_cont16", &CtxKnobs::no_cont16}, {"no_decomp_wave", &CtxKnobs::no_decomp_wave}, {"no_af_split", &CtxKnobs::no_af_split}, {"no_front_lists", &CtxKnobs::no_front_lists}, {"no_origin_band", &CtxKnobs::no_origin_band}, {"no_quads", &CtxKnobs::no_quads}, {"no_fork", &CtxKnobs::no_fork}, {"sweeps_alone", &CtxKnobs::sweeps_alone}, {"verbose", &CtxKnobs::verbose}};
bool same_name(const char* a, const char* b) {
  for (; *a && *b; ++a, ++b)
    if (std::tolower((unsigned char)*a) != std::tolower((unsigned char)*b)) return false;
  return *a == *b;
}
}  // namespace
bool knobs_set(CtxKnobs& k, const char* name, const char* value) {
  if (!name || !value) return false;
  if (same_name(name, "band_w")) {
    char* end = nullptr;
    const long v = strtol(value, &end, 10);
    if (end == value) return false;
    k.band_w = v < 0 ? -1 : (int32_t)std::min<long>(4096, v);  // (widths the band forms cannot hold leave the pair on the whole matrix)
    return true;
  }
  if (same_name(name, "quad_tier_min")) {
    const long v = atol(value);
    if (v < 0) return false;
    k.quad_tier_min = (uint32_t)std::min<long>(v, 0x7fffffffl);
    return true;
  }
  if (same_name(name, "front_list_min")) {
    const long v = atol(value);
    if (v < 0) return false;
    k.front_list_min = (uint32_t)std::min<long>(v, 0x7fffffffl);
    return true;
  }
  if (same_name(name, "ckpt_b")) {
    const long v = atol(value);
    if (v < 32 || v > 1024) return false;
    k.ckpt_b = (uint32_t)v;
    return true;
  }
  for (const KnobField& f : kKnobFlags)
    if (same_name(name, f.name)) {
      // (the environment form is "set = on", whatever the value, as it has always been; "0" through the API switches off)
      k.*(f.flag) = !(value[0] == '0' && value[1] == 0);
      return true;
    }
  return false;
}
void knobs_from_env(CtxKnobs& k) {
  for (const KnobField& f : kKnobFlags) {
    std::string env = "TRACYHIP_";
    for (const char* c = f.name; *c; ++c) env.push_back((char)std::toupper((unsigned char)*c));
    if (getenv(env.c_str())) k.*(f.flag) = true;
  }
  if (getenv("TRACYHIP_HOST_TIMERS")) k.verbose = true;
  if (const char* e = getenv("TRACYHIP_BAND_W")) knobs_set(k, "band_w", e);
  if (const char* e = getenv("TRACYHIP_CKPT_B")) knobs_set(k, "ckpt_b", e);
  if (const char* e = getenv("TRACYHIP_QUAD_TIER_MIN")) knobs_set(k, "quad_tier_min", e);
  if (const char* e = getenv("TRACYHIP_FRONT_LIST_MIN")) knobs_set(k, "front_list_min", e);
}
std::string knobs_describe(const CtxKnobs& k) {
  std::string s;
  for (const KnobField& f : kKnobFlags) { s += f.name; s += k.*(f.flag) ? "=1\n" : "=0\n"; }
  s += "band_w=" + std::to_string(k.band_w) + "\n";
  s += "ckpt_b=" + std::to_string(k.ckpt_b) + "\n";
  s += "quad_tier_min=" + std::to_string(k.quad_tier_min) + "\n";
  s += "front_list_min=" + std::to_string(k.front_list_min) + "\n";
  s += "host_threads=" + std::to_string(host_pool_threads()) + "\n";
  return s;
}

int timing_begin(tracyhip_ctx* ctx, int which, uint64_t cells, uint64_t bytes) {
  if (!ctx->timing) return TRACYHIP_OK;
  tracyhip_ctx::Pending p{which, nullptr, nullptr, cells, bytes};
  HIP_TRY(get_event(ctx, &p.e0));
  HIP_TRY(get_event(ctx, &p.e1));
  HIP_TRY(hipEventRecord(p.e0, ctx->stream));
  ctx->pending.push_back(p);
  return TRACYHIP_OK;
}
int timing_end(tracyhip_ctx* ctx) {
  if (!ctx->timing || ctx->pending.empty()) return TRACYHIP_OK;
  HIP_TRY(hipEventRecord(ctx->pending.back().e1, ctx->stream));
  return TRACYHIP_OK;
}
int timing_collect(tracyhip_ctx* ctx) {
  for (auto& p : ctx->pending) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
      ctx->acc[p.which].ms += ms;
      ctx->acc[p.which].launches += 1;
      ctx->acc[p.which].cells += p.cells;
      ctx->acc[p.which].bytes += p.bytes;
    } else {
      (void)hipGetLastError();
    }
    ctx->free_events.push_back(p.e0);
    ctx->free_events.push_back(p.e1);
  }
  ctx->pending.clear();
  return TRACYHIP_OK;
}

int ctx_begin(tracyhip_ctx* ctx) {
  if (!ctx) return set_error(TRACYHIP_ERR_ARG, "null context");
  // a synchronous call on a context with queued asynchronous work runs after it (the worker thread itself passes through)
  if (ctx->async && std::this_thread::get_id() != ctx->async->worker_id) {
    std::unique_lock<std::mutex> lk(ctx->async->m);
    ctx->async->cv.wait(lk, [&] { return ctx->async->q.empty() && !ctx->async->busy; });
  }
  HIP_TRY(hipSetDevice(ctx->device));
  return TRACYHIP_OK;
}

int async_submit(tracyhip_ctx* ctx, std::function<int()> fn) {
  if (!ctx) return set_error(TRACYHIP_ERR_ARG, "null context");
  if (!ctx->async) {
    auto* a = new tracyhip_ctx::AsyncState();
    ctx->async = a;
    a->worker = std::thread([a]() {
      for (;;) {
        std::function<int()> job;
        {
          std::unique_lock<std::mutex> lk(a->m);
          a->cv.wait(lk, [&] { return a->stop || !a->q.empty(); });
          if (a->q.empty()) return;  // stop requested and nothing left
          job = std::move(a->q.front());
          a->q.pop_front();
          a->busy = true;
        }
        const int rc = job();
        {
          std::lock_guard<std::mutex> lk(a->m);
          if (rc != TRACYHIP_OK && a->rc == TRACYHIP_OK) { a->rc = rc; a->msg = tracyhip_last_error(); }
          a->busy = false;
        }
        a->cv.notify_all();
      }
    });
    a->worker_id = a->worker.get_id();
  }
  {
    std::lock_guard<std::mutex> lk(ctx->async->m);
    ctx->async->q.push_back(std::move(fn));
  }
  ctx->async->cv.notify_all();
  return TRACYHIP_OK;
}

int async_drain(tracyhip_ctx* ctx) {
  if (!ctx || !ctx->async) return TRACYHIP_OK;
  auto* a = ctx->async;
  std::unique_lock<std::mutex> lk(a->m);
  a->cv.wait(lk, [&] { return a->q.empty() && !a->busy; });
  const int rc = a->rc;
  if (rc != TRACYHIP_OK) set_error(rc, "%s", a->msg.c_str());
  a->rc = TRACYHIP_OK;
  a->msg.clear();
  return rc;
}

int stage_in(tracyhip_ctx* ctx, DevBuf& buf, const void* src, uint64_t bytes, int mem, const void** dev) {
  if (mem == TRACYHIP_MEM_DEVICE || bytes == 0) {
    *dev = src;
    return TRACYHIP_OK;
  }
  HIP_TRY(buf.ensure(bytes));
  HIP_TRY(hipMemcpyAsync(buf.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  *dev = buf.p;
  return TRACYHIP_OK;
}

int check_params(const tracyhip_params* prm, uint64_t max_mn) {
  if (!prm) return set_error(TRACYHIP_ERR_ARG, "null params");
  auto ab = [](int32_t x) { return (int64_t)(x < 0 ? -(int64_t)x : x); };
  // (beyond |1000| the table-driven tracebacks keep unshifted entries and the pipelines leave their 16-bit / banded forms: kWideScore)
  if (ab(prm->match) > 30000 || ab(prm->mismatch) > 30000)
    return set_error(TRACYHIP_ERR_RANGE, "|match|,|mismatch| must be <= 30000 (int16 query profile)");
  const int64_t c = ab(prm->go) + ab(prm->ge) + std::max(ab(prm->match), ab(prm->mismatch));
  if ((int64_t)(max_mn + 2) * c + 1000000 >= (1ll << 26))
    return set_error(TRACYHIP_ERR_RANGE, "(m+n) * cost exceeds the exact range of the x32 int32 kernels");
  return TRACYHIP_OK;
}

// ---- the DP driver shared by gotoh/needle score/align -----------------------------------------------
static int64_t iabs64(int32_t x) { return x < 0 ? -(int64_t)x : (int64_t)x; }
int32_t sub_limit(const tracyhip_params* prm) { return (int32_t)std::max(iabs64(prm->match), iabs64(prm->mismatch)); }

// 16-bit score kernel: every real DP value must fit int16 with room below for the sentinel.  Q bounds the absolute value of
// a substitution score: max(|match|, |mismatch|) for strings and normalised profiles (the a-priori call, Q = 0), the
// device-reported maximum when a launch has seen a larger query-profile entry (range_verdict).
bool narrow_ok(const tracyhip_params* prm, uint32_t maxm, int K, int64_t Q) {
  // free end gaps on the first/last row only, strictly negative extension, one pass of the strip height
  if (!prm->hfree || prm->vfree || prm->go > 0 || prm->ge >= 0 || num_passes(maxm ? maxm : 1, K) != 1) return false;
  Q = std::max<int64_t>(Q, sub_limit(prm));
  const int64_t rows = (int64_t)num_passes(maxm ? maxm : 1, K) * 64 * K;
  const int64_t low = iabs64(prm->go) + rows * iabs64(prm->ge) + 2 * (iabs64(prm->go) + iabs64(prm->ge)) + 2 * Q;
  const int64_t high = rows * Q;
  return (low < -(int64_t)kNegInf16 - iabs64(prm->ge) - 64) && (high < 30000);
}

// profile x profile score kernel with 16-bit cells (any AlignConfig, any number of passes).  Every H, E, F of the matrix is at least
// the score of the path "one vertical gap, then one horizontal gap" minus one more gap open, and at most min(m, n) substitution
// scores; the sentinel (kNegInf16) has to lose against that and must not wrap when a gap cost is added to it once.
bool arith16_ok(const tracyhip_params* prm, uint64_t max_mn, int64_t Q) {
  if (prm->go > 0 || prm->ge > 0) return false;
  Q = std::max<int64_t>(Q, sub_limit(prm));
  const int64_t low = 3 * iabs64(prm->go) + ((int64_t)max_mn + 2) * iabs64(prm->ge) + Q;
  const int64_t high = ((int64_t)max_mn / 2 + 1) * Q;
  return low < -(int64_t)kNegInf16 - 1000 && iabs64(prm->go) + iabs64(prm->ge) < 10000 && high < 30000;
}

// origin-tracking sweep: string x string only, so substitution scores are match / mismatch exactly
bool origin_ok(const tracyhip_params* prm, uint32_t maxm, uint32_t maxn, int K) {
  if (!prm->hfree || prm->vfree || prm->go > 0 || prm->ge >= 0 || num_passes(maxm ? maxm : 1, K) != 1) return false;
  if ((uint64_t)maxn + 64 >= (1u << kOriginBits)) return false;
  const int64_t rows = 64 * (int64_t)K;
  const int64_t low = iabs64(prm->go) + rows * iabs64(prm->ge) + 2 * (iabs64(prm->go) + iabs64(prm->ge)) + iabs64(prm->mismatch) + iabs64(prm->match);
  const int64_t high = rows * sub_limit(prm);
  // 14-bit score field [-8192, 8191]; the sentinel kNegInfOrigin must stay below every real value and above the field's floor
  return (low < -(int64_t)kNegInfOrigin - iabs64(prm->ge) - 64) && (-(int64_t)kNegInfOrigin + iabs64(prm->go) + iabs64(prm->ge) < 8000) && (high < 8000);
}

// What the launches of one run reported about their substitution scores (DpArgs::err).  TRACYHIP_OK: every value range the
// host assumed before the launch held.  kWiden: a 16-bit kernel ran outside its proven range -- its results are discarded
// and the caller repeats the work on the int32 kernels.  TRACYHIP_ERR_RANGE: not even those hold the values exactly.
int range_verdict(const tracyhip_params* prm, const int32_t* herr, const std::vector<std::pair<uint32_t, int>>& narrow_launches,
                  uint64_t max_mn, int value_shift) {
  if (herr[0] & 1) return set_error(TRACYHIP_ERR_RANGE, "a query-profile score does not fit the int16 table (profile values too large)");
  if (herr[0] & 2) return set_error(TRACYHIP_ERR_RANGE, "traceback left the matrix (degenerate scoring parameters)");
  int64_t Q = sub_limit(prm);
  bool seen = false;
  if (herr[1] > Q) { Q = herr[1]; seen = true; }
  if (herr[2] || herr[3]) {
    float fa, fb;
    std::memcpy(&fa, &herr[2], 4);
    std::memcpy(&fb, &herr[3], 4);
    if (!(fa < 3.0e38f) || !(fb < 3.0e38f)) return set_error(TRACYHIP_ERR_RANGE, "a profile holds NaN or infinite values");
    const double bound = (double)std::max(fa, 1.001f) * (double)std::max(fb, 1.001f) * (double)sub_limit(prm) * 1.0001 + 1.0;
    if (bound > 1.0e9) return set_error(TRACYHIP_ERR_RANGE, "profile values too large for exact int32 scoring");
    Q = std::max<int64_t>(Q, (int64_t)bound + 1);
    seen = true;
  }
  if (!seen) return TRACYHIP_OK;
  for (auto const& nl : narrow_launches)  // (K = 0 marks a launch of the 16-bit profile x profile kernel: first = its largest m + n)
    if (nl.second == 0 ? !arith16_ok(prm, nl.first, Q) : !narrow_ok(prm, nl.first, nl.second, Q)) return kWiden;
  const int64_t c = iabs64(prm->go) + iabs64(prm->ge) + Q;
  if ((int64_t)(max_mn + 2) * c + 1000000 >= (1ll << (31 - value_shift)))
    return set_error(TRACYHIP_ERR_RANGE, "un-normalised profile: (m+n) * (gap cost + largest substitution score %lld) exceeds the exact range of the int32 kernels",
                     (long long)Q);
  return TRACYHIP_OK;
}

int run_dp(tracyhip_ctx* ctx, const DpProblem& pb, const tracyhip_params* prm, bool needle, bool trace,
           int32_t* d_scores, uint8_t* d_ops, const uint64_t* d_ops_off, uint32_t* d_ops_len, int stage, DpCkpt* ck) {
  const uint32_t np = (uint32_t)pb.desc.size();
  if (np == 0) return TRACYHIP_OK;
  if (sub_limit(prm) > kWideScore && ((trace && pb.mode == MODE_CQ) || stage == DP_BAND))
    return set_error(TRACYHIP_ERR_ARG, "run_dp: scoring beyond |%d| takes the byte-compare / whole-matrix kernels", kWideScore);
  hipStream_t st = ctx->stream;
  TRACYHIP_HOST_SCOPE(hs_all, "run_dp");

  // order: strip height, then longest first (long problems start early, short ones fill the tail)
  auto hs_plan = std::make_unique<HostScope>("run_dp.plan");
  std::vector<uint32_t> order(np);
  for (uint32_t i = 0; i < np; ++i) order[i] = i;
  auto before = [&](uint32_t x, uint32_t y) {
    if (pb.k[x] != pb.k[y]) return pb.k[x] > pb.k[y];
    const uint32_t fx = pb.desc[x].flags & PAIR_ROW4_ZERO, fy = pb.desc[y].flags & PAIR_ROW4_ZERO;
    if (fx != fy) return fx > fy;  // profile x profile: 16-term pairs and 25-term pairs go to different launches
    return (uint64_t)pb.desc[x].m * pb.desc[x].n > (uint64_t)pb.desc[y].m * pb.desc[y].n;
  };
  // (the order only balances the tail of a launch: batches of one strip height whose sizes lie within 25 % of each other --
  // the final alignments of a `tracy align` batch -- keep the caller's order and save the host the sort between two kernels)
  bool similar = true;
  {
    uint64_t lo = ~0ull, hi = 0;
    for (uint32_t i = 0; i < np && similar; ++i) {
      const uint64_t c = (uint64_t)pb.desc[i].m * pb.desc[i].n;
      lo = std::min(lo, c); hi = std::max(hi, c);
      similar = pb.k[i] == pb.k[0] && (pb.desc[i].flags & PAIR_ROW4_ZERO) == (pb.desc[0].flags & PAIR_ROW4_ZERO);
    }
    similar = similar && hi <= lo + lo / 4;
  }
  if (!similar && !std::is_sorted(order.begin(), order.end(), before)) std::stable_sort(order.begin(), order.end(), before);

  // workspace plan: chunks of consecutive (sorted) pairs whose traceback words fit the limit
  const uint64_t word_bytes = needle ? 4 : 8;
  uint64_t limit = ctx->ws_limit;
  if (limit == 0 && trace && stage == DP_PLAIN) {  // only the full-matrix traceback needs a workspace plan
    size_t fr = 0, tot = 0;
    HIP_TRY(hipMemGetInfo(&fr, &tot));
    limit = (uint64_t)(fr * 0.70 / ctx->mem_share) + ctx->d_bits.cap;  // this context's share of what is free now plus what it already holds
  } else if (limit == 0) {
    limit = ~0ull;
  }
  HIP_TRY(ctx->h_desc.ensure(sizeof(PairDesc) * (size_t)np));
  PairDesc* hd = static_cast<PairDesc*>(ctx->h_desc.p);
  struct Chunk { uint32_t lo, hi; uint64_t words, scratch; };
  std::vector<Chunk> chunks;
  if (!(trace && stage == DP_PLAIN) && np >= (1u << 16)) {
    // no traceback words: one chunk, and the only running total is the boundary-row scratch of multi-pass pairs.  Long lists
    // (an all-pairs job: 36 MB of descriptors) are laid out by a few threads: sizes, a scan over the threads' totals, the copy.
    constexpr uint32_t NT = 8;
    uint64_t tsum[NT + 1] = {0};
    auto range = [&](uint32_t t) { return std::make_pair((uint32_t)((uint64_t)np * t / NT), (uint32_t)((uint64_t)np * (t + 1) / NT)); };
    auto scr_of = [&](uint32_t j) -> uint64_t {
      const PairDesc& d = pb.desc[order[j]];
      return (d.m && d.n && num_passes(d.m, pb.k[order[j]]) > 1) ? (uint64_t)d.n + 2 : 0;
    };
    {
      std::vector<std::thread> th;
      for (uint32_t t = 0; t < NT; ++t)
        th.emplace_back([&, t]() { uint64_t a = 0; for (uint32_t j = range(t).first; j < range(t).second; ++j) a += scr_of(j); tsum[t + 1] = a; });
      for (auto& x : th) x.join();
    }
    for (uint32_t t = 0; t < NT; ++t) tsum[t + 1] += tsum[t];
    {
      std::vector<std::thread> th;
      for (uint32_t t = 0; t < NT; ++t)
        th.emplace_back([&, t]() {
          uint64_t a = tsum[t];
          for (uint32_t j = range(t).first; j < range(t).second; ++j) {
            PairDesc d = pb.desc[order[j]];
            d.bits_off = 0;
            d.scratch_off = a;
            a += scr_of(j);
            hd[j] = d;
          }
        });
      for (auto& x : th) x.join();
    }
    chunks.push_back(Chunk{0, np, 0, tsum[NT]});
  } else {
    Chunk c{0, 0, 0, 0};
    for (uint32_t j = 0; j < np; ++j) {
      PairDesc d = pb.desc[order[j]];
      const int K = pb.k[order[j]];
      const uint32_t P = (d.m && d.n) ? num_passes(d.m, K) : 0;
      const uint64_t words = (trace && stage == DP_PLAIN) ? (uint64_t)P * steps_per_pass(d.n) * 64 : 0;
      const uint64_t scr = (P > 1) ? (uint64_t)d.n + 2 : 0;
      if (words * word_bytes > limit)
        return set_error(TRACYHIP_ERR_OOM, "one pair needs %llu bytes of traceback planes, workspace limit is %llu",
                         (unsigned long long)(words * word_bytes), (unsigned long long)limit);
      if (c.hi > c.lo && (c.words + words) * word_bytes > limit) {
        chunks.push_back(c);
        c = Chunk{j, j, 0, 0};
      }
      d.bits_off = c.words;
      d.scratch_off = c.scratch;
      c.words += words;
      c.scratch += scr;
      c.hi = j + 1;
      hd[j] = d;
    }
    chunks.push_back(c);
  }
  hs_plan.reset();  // the planning part ends here
  uint64_t max_words = 0, max_scr = 0;
  for (const Chunk& c : chunks) { max_words = std::max(max_words, c.words); max_scr = std::max(max_scr, c.scratch); }
  HIP_TRY(ctx->d_desc.ensure(sizeof(PairDesc) * (size_t)np));
  HIP_TRY(hipMemcpyAsync(ctx->d_desc.p, hd, sizeof(PairDesc) * (size_t)np, hipMemcpyHostToDevice, st));
  if (trace && stage == DP_PLAIN) HIP_TRY(ctx->d_bits.ensure(max_words * word_bytes));
  if (stage == DP_BAND) {
    uint32_t maxrun = 0;
    for (const Chunk& c : chunks) maxrun = std::max(maxrun, c.hi - c.lo);
    HIP_TRY(ctx->d_band.ensure((size_t)maxrun * ck->B * 64 * 8));
  }
  if (max_scr) HIP_TRY(ctx->d_scratch.ensure(max_scr * 8));
  HIP_TRY(ctx->d_err.ensure(kErrBytes));
  HIP_TRY(hipMemsetAsync(ctx->d_err.p, 0, sizeof(int32_t) * kErrWords, st));
  std::vector<std::pair<uint32_t, int>> narrow_launches;  // (tallest problem, K) of every 16-bit launch, for range_verdict
  uint64_t band_cells_credited = 0;                       // m * n of the band launches (replaced by the swept cells after the sync)
  uint64_t max_mn = 0;
  for (uint32_t j = 0; j < np; ++j) max_mn = std::max<uint64_t>(max_mn, (uint64_t)pb.desc[j].m + pb.desc[j].n);

  DpArgs a{};
  a.a1 = pb.d_a1;
  a.a2 = pb.d_a2;
  a.bits = static_cast<uint64_t*>(ctx->d_bits.p);
  a.bits32 = static_cast<uint32_t*>(ctx->d_bits.p);
  a.scratch = static_cast<int32_t*>(ctx->d_scratch.p);
  a.scores = d_scores;
  a.err = static_cast<int32_t*>(ctx->d_err.p);
  a.match = prm->match; a.mismatch = prm->mismatch; a.go = prm->go; a.ge = prm->ge;
  a.hfree = prm->hfree; a.vfree = prm->vfree;
  a.qlimit = sub_limit(prm);
  // Gotoh tracebacks: the sweep's workgroup walks its pair itself (TRACYHIP_NO_FUSED_WALK=1: the separate walk launch)
  const bool fused_walk = trace && stage == DP_PLAIN && !needle && !ctx->knobs.no_fused_walk;
  if (fused_walk) { a.walk_ops = d_ops; a.walk_ops_off = d_ops_off; a.walk_ops_len = d_ops_len; }
  a.screen = ctx->knobs.no_screen ? 0 : 1;
  a.colcode = pb.d_colclass;
  if (pb.mode == MODE_QP && pb.d_a2 == ctx->codes() && !ctx->knobs.no_compact) a.special_blocks = ctx->special_blocks();
  if (pb.mode == MODE_CQ && !ctx->knobs.no_compact) a.special_blocks = pb.d_special;
  if (stage == DP_BAND && ctx->timing) {
    a.swept = reinterpret_cast<unsigned long long*>(static_cast<int32_t*>(ctx->d_err.p) + kErrSweptWord);
    HIP_TRY(hipMemsetAsync(a.swept, 0, sizeof(unsigned long long), st));
  }
  if (ck) a.ends = ck->d_ends;
  if (ck && stage == DP_CKPT) { a.votes = ck->d_votes; a.vote_nt = ck->vote_nt; }
  if (ck) { a.ckpt = ck->d_ckpt; a.lastrow = ck->d_lastrow; a.ckpt_B = ck->B; a.band = static_cast<uint64_t*>(ctx->d_band.p); a.ckpt_narrow = ck->narrow ? 1 : 0; }
  const PairDesc* dd = static_cast<const PairDesc*>(ctx->d_desc.p);

  for (const Chunk& c : chunks) {
    uint32_t j = c.lo;
    while (j < c.hi) {  // one launch per run of equal K (and, profile x profile, equal term count)
      uint32_t e = j;
      const int K = pb.k[order[j]];
      const uint32_t row4 = hd[j].flags & PAIR_ROW4_ZERO;
      while (e < c.hi && pb.k[order[e]] == K && (hd[e].flags & PAIR_ROW4_ZERO) == row4) ++e;
      a.pairs = dd + j;
      int trc;
      if (ctx->timing) {
        uint64_t cells = 0, bytes = 0;
        for (uint32_t q = j; q < e; ++q) {
          const PairDesc& d = hd[q];
          uint64_t mn = (uint64_t)(stage == DP_PREFIX ? std::min<uint32_t>(d.m, (uint32_t)kPrefixLanes * K) : d.m) * d.n;
          if (trace && stage == DP_PLAIN && (d.flags & PAIR_BANDED)) {  // multi-pass band form: the cells its passes sweep, not the matrix
            mn = 0;
            for (uint32_t base = 0; base < d.m; base += 64u * K) {
              const uint32_t rows_here = std::min<uint32_t>(d.m - base, 64u * K);
              const int64_t c_lo = std::max<int64_t>(1, (int64_t)base + 1 + band_dmin(d)), c_hi = std::min<int64_t>(d.n, (int64_t)(base + rows_here) + band_dmax(d));
              if (c_hi >= c_lo) mn += (uint64_t)rows_here * (uint64_t)(c_hi - c_lo + 1);
            }
          }
          cells += mn;
          bytes += (trace ? mn / 2 : 0) + (pb.a1_profile ? 24ull * d.m : d.m) + (pb.a2_profile ? 24ull * d.n : d.n) + 4;
        }
        if (stage == DP_BAND) { bytes = 0; band_cells_credited += cells; }  // (bytes: the bands live in a per-pair buffer, no matrix-sized traffic)
        if ((trc = timing_begin(ctx, stage == DP_BAND ? TRACYHIP_TIMER_BAND : stage == DP_PREFIX ? TRACYHIP_TIMER_PREFIX : stage == DP_ORIGIN ? TRACYHIP_TIMER_ORIGIN
                                     : trace ? TRACYHIP_TIMER_TRACE : TRACYHIP_TIMER_SCORE, cells, bytes))) return trc;
      }
      bool narrow = false;
      if (!needle && !trace && (pb.mode == MODE_QP || pb.mode == MODE_CHAR) && !ctx->knobs.no_narrow) {
        uint32_t maxm = 0;
        for (uint32_t q = j; q < e; ++q) maxm = std::max(maxm, hd[q].m);
        narrow = narrow_ok(prm, maxm, K);
      }
      if (stage == DP_PREFIX || (stage == DP_CKPT && ck->narrow) || (stage == DP_PLAIN && narrow)) {
        uint32_t maxm = 0;
        for (uint32_t q = j; q < e; ++q) maxm = std::max(maxm, hd[q].m);
        narrow_launches.emplace_back(maxm, K);
      }
      if (stage == DP_PREFIX) {
        HIP_TRY(launch_gotoh_prefix(K, a, e - j, st));
      } else if (stage == DP_ORIGIN) {
        HIP_TRY(launch_gotoh_origin(K, pb.mode == MODE_CQ ? 1 : pb.mode == MODE_QP ? 2 : 0, pb.cq_codes, a, e - j, st));
      } else if (stage == DP_CKPT) {
        // one representation for the whole batch: the caller checks narrow_ok for the largest problem
        narrow = ck->narrow;
        HIP_TRY(launch_gotoh_ckpt(pb.mode, K, narrow, a, e - j, st));
      }
      else if (stage == DP_BAND) {
        WalkArgs wa{};
        wa.pairs = dd + j; wa.ops = d_ops; wa.ops_off = d_ops_off; wa.ops_len = d_ops_len; wa.err = a.err; wa.npairs = e - j; wa.K = K;
        HIP_TRY(launch_band_trace(pb.mode, K, a, wa, e - j, st));
      } else if (!needle && pb.mode == MODE_PROF) {
        bool a16 = false;
        if (!trace && !ctx->knobs.no_narrow) {
          uint64_t mn = 0;
          for (uint32_t q = j; q < e; ++q) mn = std::max<uint64_t>(mn, (uint64_t)hd[q].m + hd[q].n);
          if ((a16 = arith16_ok(prm, mn, 0))) narrow_launches.emplace_back((uint32_t)mn, 0);
        }
        HIP_TRY(launch_gotoh_prof(K, trace, row4 != 0, a16, a, e - j, st));
      }
      else
        HIP_TRY(needle ? launch_needle(pb.mode, K, trace, a, e - j, st)
                       : launch_gotoh(pb.mode, K, trace, narrow, a, e - j, st, trace && pb.mode == MODE_QP && sub_limit(prm) > kWideScore));
      if ((trc = timing_end(ctx))) return trc;
      if (trace && stage == DP_PLAIN && !fused_walk) {
        WalkArgs wa{};
        wa.pairs = dd + j;
        wa.bits = a.bits;
        wa.ops = d_ops;
        wa.ops_off = d_ops_off;
        wa.ops_len = d_ops_len;
        wa.err = a.err;
        wa.npairs = e - j;
        wa.K = K;
        if ((trc = timing_begin(ctx, TRACYHIP_TIMER_WALK, 0, 0))) return trc;
        HIP_TRY(needle ? launch_needle_walk(wa, a.bits32, st) : launch_gotoh_walk(wa, st));
        if ((trc = timing_end(ctx))) return trc;
      }
      j = e;
    }
  }
  int32_t herr[kErrWords] = {};
  unsigned long long h_swept = 0;
  HIP_TRY(hipMemcpyAsync(herr, ctx->d_err.p, sizeof(herr), hipMemcpyDeviceToHost, st));
  if (a.swept) HIP_TRY(hipMemcpyAsync(&h_swept, a.swept, sizeof(h_swept), hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx_sync(ctx));
  timing_collect(ctx);
  if (a.swept) {  // the band timer reports the cells it really evaluated, not the matrices it stands in for
    ctx->acc[TRACYHIP_TIMER_BAND].cells -= std::min<uint64_t>(ctx->acc[TRACYHIP_TIMER_BAND].cells, band_cells_credited);
    ctx->acc[TRACYHIP_TIMER_BAND].cells += h_swept;
  }
  // the packed score field of the origin-tracking sweep is sized for substitution scores of normalised profiles
  if (stage == DP_ORIGIN && pb.mode == MODE_QP && herr[1] > sub_limit(prm)) return kWiden;
  const int verdict = range_verdict(prm, herr, narrow_launches, max_mn, trace ? (needle ? 2 : kTagShift) : 0);
  if (verdict == kWiden && stage == DP_PLAIN) {  // the 16-bit score kernel met an un-normalised profile: same work on the int32 kernel
    const bool keep = ctx->knobs.no_narrow;
    ctx->knobs.no_narrow = true;
    const int rc = run_dp(ctx, pb, prm, needle, trace, d_scores, d_ops, d_ops_off, d_ops_len, stage, ck);
    ctx->knobs.no_narrow = keep;
    return rc;
  }
  return verdict;  // DP_CKPT / DP_PREFIX: kWiden goes to the pipeline, which restarts its orientation stage on the int32 kernels
}

// ---- band kernels (band16.h) ----------------------------------------------------------------------------------------
int band16_pick_k(int32_t dmin, int32_t dmax) { return b16_pick_k(dmin, dmax); }

bool origin16_ok(const tracyhip_params* prm, uint32_t maxm, uint32_t maxn) {
  if (!prm->hfree || prm->vfree) return false;
  return b16_origin_ok(prm->match, prm->mismatch, prm->go, prm->ge, maxm, maxn);
}

int build_b16_tables(tracyhip_ctx* ctx, DevBuf& buf, const void* d_a1, bool strings, std::vector<B16TableDesc>& desc, const tracyhip_params* prm) {
  const uint32_t ns = (uint32_t)desc.size();
  if (ns == 0) return TRACYHIP_OK;
  uint64_t tot = 0;
  for (auto& d : desc) {
    d.stride = b16_table_stride(d.m);
    d.out_off = tot;
    tot += (uint64_t)kB16Codes * d.stride;
  }
  hipStream_t st = ctx->stream;
  HIP_TRY(buf.ensure(tot * sizeof(int16_t) + 64));
  HIP_TRY(ctx->d_b16desc.ensure(sizeof(B16TableDesc) * (size_t)ns));
  // (staging blocks of their own, four in rotation: no host wait for an upload, and the pipelines synchronise several times between
  // one build and the fourth after it)
  PinBuf& stage = ctx->h_b16desc[ctx->b16_round++ & 3u];
  HIP_TRY(stage.ensure(sizeof(B16TableDesc) * (size_t)ns));
  std::memcpy(stage.p, desc.data(), sizeof(B16TableDesc) * (size_t)ns);
  HIP_TRY(hipMemcpyAsync(ctx->d_b16desc.p, stage.p, sizeof(B16TableDesc) * (size_t)ns, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx->d_err.ensure(kErrBytes));
  int trc;
  if ((trc = timing_begin(ctx, TRACYHIP_TIMER_MISC, 0, tot * 2))) return trc;
  HIP_TRY(launch_b16_tables(static_cast<const B16TableDesc*>(ctx->d_b16desc.p), ns, d_a1, strings, prm->match, prm->mismatch, sub_limit(prm), kTagShift,
                            static_cast<int16_t*>(buf.p), static_cast<int32_t*>(ctx->d_err.p), st));
  if ((trc = timing_end(ctx))) return trc;
  return TRACYHIP_OK;  // (no host wait: the staging block is the tables' own)
}

int run_band16(tracyhip_ctx* ctx, Band16Job& job, const tracyhip_params* prm, int32_t* d_scores, uint32_t* d_ends, uint8_t* d_ops,
               const uint64_t* d_ops_off, uint32_t* d_ops_len) {
  const uint32_t nall = (uint32_t)job.desc.size();
  if (nall == 0) return TRACYHIP_OK;
  TRACYHIP_HOST_SCOPE(hs_all, "run_band16");
  hipStream_t st = ctx->stream;
  auto hs_plan = std::make_unique<HostScope>("run_band16.plan");
  uint64_t limit = ctx->ws_limit;
  // Order: strip height (12, 8, 4), the caller's order within one (the pairs of a pipeline stage are of a size).  Laid out by a
  // few threads: per-thread counts per strip height, a scan, the fill -- descriptors go straight into the pinned staging block.
  constexpr int NB = 3;
  auto bucket_of = [](int K) { return K == 12 ? 0 : K == 8 ? 1 : K == 4 ? 2 : -1; };
  static const int bucket_k[NB] = {12, 8, 4};
  struct Part { uint32_t n[NB]; uint64_t bytes[NB]; uint32_t nmax[NB]; uint64_t cells[NB], tbytes[NB]; bool bad; };  // (cells / tbytes: what the timers credit)
  Part part[kHostThreads] = {};
  std::vector<uint64_t> wb(nall);  // bytes of traceback words per pair (kind 0)
  parallel_for(nall, [&](uint32_t lo, uint32_t hi, uint32_t tid) {
    Part& pt = part[tid];
    for (uint32_t i = lo; i < hi; ++i) {
      const int K = job.k[i];
      if (K == 0) { wb[i] = 0; continue; }
      const PairDesc& d = job.desc[i];
      const int b = bucket_of(K);
      if (b < 0 || d.m == 0 || d.n == 0 || b16_window(K, band_dmin(d), band_dmax(d)) > b16_max_window(K)) { pt.bad = true; continue; }
      const uint64_t wds = (job.kind == 0 || ctx->timing) ? b16_words(d.m, d.n, K, band_dmin(d), band_dmax(d)) : 0;
      const uint64_t bytes = job.kind == 0 ? ((b16_store_words(d.m, d.n, K, band_dmin(d), band_dmax(d)) * b16_word_bytes(K) + 15u) & ~15ull) : 0;
      wb[i] = bytes;
      pt.n[b] += 1;
      pt.bytes[b] += bytes;
      pt.nmax[b] = std::max(pt.nmax[b], d.n);
      pt.cells[b] += wds * (uint64_t)K;
      pt.tbytes[b] += (job.kind == 0 ? wds * b16_word_bytes(K) : 0) + 12ull * d.m + d.n + 4;
    }
  });
  uint32_t bn[NB] = {0, 0, 0}, bnmax[NB] = {0, 0, 0};
  uint64_t total_bytes = 0, bcells[NB] = {0, 0, 0}, btbytes[NB] = {0, 0, 0};
  for (uint32_t t = 0; t < kHostThreads; ++t) {
    if (part[t].bad) { return set_error(TRACYHIP_ERR_ARG, "run_band16: pair outside the band kernels' domain"); }
    for (int b = 0; b < NB; ++b) {
      bn[b] += part[t].n[b]; total_bytes += part[t].bytes[b];
      bnmax[b] = std::max(bnmax[b], part[t].nmax[b]); bcells[b] += part[t].cells[b]; btbytes[b] += part[t].tbytes[b];
    }
  }
  const uint32_t np = bn[0] + bn[1] + bn[2];
  if (np == 0) { return TRACYHIP_OK; }
  if (limit == 0) {
    if (job.kind != 0 || total_bytes + 64 <= ctx->d_bits.cap) limit = ~0ull;  // (the words fit what is there: no driver call)
    else {
      size_t fr = 0, tot = 0;
      HIP_TRY(hipMemGetInfo(&fr, &tot));
      limit = (uint64_t)(fr * 0.70 / ctx->mem_share) + ctx->d_bits.cap;
    }
  }
  HIP_TRY(ctx->h_desc.ensure(sizeof(PairDesc) * (size_t)np));
  PairDesc* hd = static_cast<PairDesc*>(ctx->h_desc.p);
  std::vector<int> hk(np);
  struct Chunk { uint32_t lo, hi; uint64_t bytes; };
  std::vector<Chunk> chunks;
  uint64_t max_mn = 0;
  if (total_bytes <= limit) {
    // one chunk: positions and word offsets from the scan over (strip height, thread)
    uint32_t pos0[NB][kHostThreads];
    uint64_t off0[NB][kHostThreads];
    uint32_t p = 0;
    uint64_t o = 0;
    for (int b = 0; b < NB; ++b)
      for (uint32_t t = 0; t < kHostThreads; ++t) { pos0[b][t] = p; off0[b][t] = o; p += part[t].n[b]; o += part[t].bytes[b]; }
    uint64_t tmax[kHostThreads] = {};
    parallel_for(nall, [&](uint32_t lo, uint32_t hi, uint32_t tid) {
      uint32_t pp[NB];
      uint64_t oo[NB];
      for (int b = 0; b < NB; ++b) { pp[b] = pos0[b][tid]; oo[b] = off0[b][tid]; }
      uint64_t mx = 0;
      for (uint32_t i = lo; i < hi; ++i) {
        const int K = job.k[i];
        if (K == 0) continue;
        const int b = bucket_of(K);
        PairDesc d = job.desc[i];
        d.bits_off = oo[b];
        oo[b] += wb[i];
        hd[pp[b]] = d;
        hk[pp[b]] = K;
        ++pp[b];
        mx = std::max<uint64_t>(mx, (uint64_t)d.m + d.n);
      }
      tmax[tid] = mx;
    });
    for (uint32_t t = 0; t < kHostThreads; ++t) max_mn = std::max(max_mn, tmax[t]);
    chunks.push_back(Chunk{0, np, total_bytes});
  } else {
    // the words do not fit the workspace at once: chunks of consecutive pairs (serial; rare)
    uint32_t pos = 0;
    Chunk c{0, 0, 0};
    for (int b = 0; b < NB; ++b)
      for (uint32_t i = 0; i < nall; ++i) {
        if (job.k[i] != bucket_k[b]) continue;
        if (wb[i] > limit) { return set_error(TRACYHIP_ERR_OOM, "one pair needs %llu bytes of traceback words, workspace limit is %llu", (unsigned long long)wb[i], (unsigned long long)limit); }
        if (c.hi > c.lo && c.bytes + wb[i] > limit) { chunks.push_back(c); c = Chunk{pos, pos, 0}; }
        PairDesc d = job.desc[i];
        d.bits_off = c.bytes;
        c.bytes += wb[i];
        hd[pos] = d;
        hk[pos] = bucket_k[b];
        c.hi = ++pos;
        max_mn = std::max<uint64_t>(max_mn, (uint64_t)d.m + d.n);
      }
    chunks.push_back(c);
  }
  uint64_t max_bytes = 0;
  for (const Chunk& ch : chunks) max_bytes = std::max(max_bytes, ch.bytes);
  hs_plan.reset();  // the planning part ends here
  HIP_TRY(ctx->d_desc.ensure(sizeof(PairDesc) * (size_t)np));
  HIP_TRY(hipMemcpyAsync(ctx->d_desc.p, hd, sizeof(PairDesc) * (size_t)np, hipMemcpyHostToDevice, st));
  if (job.kind == 0) HIP_TRY(ctx->d_bits.ensure(max_bytes + 64));
  HIP_TRY(ctx->d_err.ensure(kErrBytes));
  // (the error words are NOT cleared here: the table kernel of this stage may have reported into them; the caller cleared them)
  Band16Args a{};
  a.qp = job.d_qp; a.codes = job.d_codes; a.bits = static_cast<uint8_t*>(ctx->d_bits.p); a.scores = d_scores; a.ends = d_ends;
  a.err = static_cast<int32_t*>(ctx->d_err.p); a.go = prm->go; a.ge = prm->ge; a.hfree = prm->hfree;
  a.ops = d_ops; a.ops_off = d_ops_off; a.ops_len = d_ops_len;
  const PairDesc* dd = static_cast<const PairDesc*>(ctx->d_desc.p);
  if (chunks.size() == 1) {
    // the whole job at once: per strip height the pairs [first, first + count) and the sums of the planning pass (no walk over the list)
    uint32_t first[NB];
    for (int b = 0, q = 0; b < NB; ++b) { first[b] = (uint32_t)q; q += (int)bn[b]; }
    const int used = (bn[0] != 0) + (bn[1] != 0) + (bn[2] != 0);
    const uint32_t nmax_all = std::max(bnmax[0], std::max(bnmax[1], bnmax[2]));
    const uint32_t cap_all = (nmax_all + 7u) & ~3u;
    int trc;
    if (np <= 24576u && used > 1 && 4ull * cap_all + b16_table_bytes(12) <= 64u * 1024u) {  // few waves, several heights: band16_multi_kernel
      Band16Args ak[3] = {a, a, a};  // 12, 8, 4
      for (int b = 0; b < NB; ++b) { ak[b].pairs = dd + first[b]; ak[b].npairs = bn[b]; ak[b].code_cap = cap_all; }
      if ((trc = timing_begin(ctx, job.kind == 0 ? TRACYHIP_TIMER_TRACE : TRACYHIP_TIMER_ORIGIN, bcells[0] + bcells[1] + bcells[2], btbytes[0] + btbytes[1] + btbytes[2]))) return trc;
      HIP_TRY(launch_band16_multi(job.kind, ak[0], ak[1], ak[2], st));
      if ((trc = timing_end(ctx))) return trc;
    } else {
      for (int b = 0; b < NB; ++b) {
        if (bn[b] == 0) continue;
        a.pairs = dd + first[b];
        a.npairs = bn[b];
        a.code_cap = (bnmax[b] + 7u) & ~3u;
        if (4ull * a.code_cap + b16_table_bytes(bucket_k[b]) > 64u * 1024u) return set_error(TRACYHIP_ERR_RANGE, "run_band16: reference of %u columns does not fit the staging area", bnmax[b]);
        if ((trc = timing_begin(ctx, job.kind == 0 ? TRACYHIP_TIMER_TRACE : TRACYHIP_TIMER_ORIGIN, bcells[b], btbytes[b]))) return trc;
        HIP_TRY(launch_band16(bucket_k[b], job.kind, a, st));
        if ((trc = timing_end(ctx))) return trc;
      }
    }
  } else
  for (const Chunk& ch : chunks) {
    uint32_t j = ch.lo;
    // a chunk of few waves with more than one strip height: one launch for all of them (band16_multi_kernel)
    if (ch.hi - ch.lo <= 24576u && hk[ch.lo] != hk[ch.hi - 1]) {
      Band16Args ak[3] = {a, a, a};  // 12, 8, 4
      uint32_t nmax = 0;
      uint64_t cells = 0, bytes = 0;
      for (int b = 0; b < 3; ++b) { ak[b].pairs = dd + ch.lo; ak[b].npairs = 0; }
      for (uint32_t e = ch.lo; e < ch.hi; ++e) {
        const int K = hk[e], b = bucket_of(K);
        if (ak[b].npairs == 0) ak[b].pairs = dd + e;
        ak[b].npairs += 1;
        nmax = std::max(nmax, hd[e].n);
        if (ctx->timing) {
          const uint64_t wds = b16_words(hd[e].m, hd[e].n, K, band_dmin(hd[e]), band_dmax(hd[e]));
          cells += wds * (uint64_t)K;
          bytes += (job.kind == 0 ? wds * b16_word_bytes(K) : 0) + 12ull * hd[e].m + hd[e].n + 4;
        }
      }
      const uint32_t cap = (nmax + 7u) & ~3u;
      if (4ull * cap + b16_table_bytes(12) <= 64u * 1024u) {
        for (int b = 0; b < 3; ++b) ak[b].code_cap = cap;
        int trc;
        if ((trc = timing_begin(ctx, job.kind == 0 ? TRACYHIP_TIMER_TRACE : TRACYHIP_TIMER_ORIGIN, cells, bytes))) return trc;
        HIP_TRY(launch_band16_multi(job.kind, ak[0], ak[1], ak[2], st));
        if ((trc = timing_end(ctx))) return trc;
        continue;
      }
    }
    while (j < ch.hi) {
      uint32_t e = j;
      const int K = hk[j];
      uint32_t nmax = 0;
      uint64_t cells = 0, bytes = 0;
      while (e < ch.hi && hk[e] == K) {
        nmax = std::max(nmax, hd[e].n);
        if (ctx->timing) {
          const uint64_t wds = b16_words(hd[e].m, hd[e].n, K, band_dmin(hd[e]), band_dmax(hd[e]));
          cells += wds * (uint64_t)K;
          bytes += (job.kind == 0 ? wds * b16_word_bytes(K) : 0) + 12ull * hd[e].m + hd[e].n + 4;
        }
        ++e;
      }
      a.pairs = dd + j;
      a.npairs = e - j;
      a.code_cap = (nmax + 7u) & ~3u;
      if (4ull * a.code_cap + b16_table_bytes(K) > 64u * 1024u) return set_error(TRACYHIP_ERR_RANGE, "run_band16: reference of %u columns does not fit the staging area", nmax);
      int trc;
      if ((trc = timing_begin(ctx, job.kind == 0 ? TRACYHIP_TIMER_TRACE : TRACYHIP_TIMER_ORIGIN, cells, bytes))) return trc;
      HIP_TRY(launch_band16(K, job.kind, a, st));
      if ((trc = timing_end(ctx))) return trc;
      j = e;
    }
  }
  int32_t herr[kErrWords] = {};
  HIP_TRY(hipMemcpyAsync(herr, ctx->d_err.p, sizeof(herr), hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx_sync(ctx));
  timing_collect(ctx);
  if (herr[0] & 1) return set_error(TRACYHIP_ERR_RANGE, "a query-profile score does not fit the int16 table (profile values too large)");
  if (herr[1] > sub_limit(prm)) return kWiden;  // un-normalised profile: the band kernels' fields are sized for normalised ones
  return TRACYHIP_OK;  // (bit 1 -- a walk left its band -- is the caller's to resolve: such pairs report ops_len 0)
}

int run_prefix_keep_cq(tracyhip_ctx* ctx, const void* d_a1, const void* d_a2, const uint8_t* d_special, const std::vector<PairDesc>& pre,
                       const tracyhip_params* prm, int32_t* d_lastrow) {
  hipStream_t st = ctx->stream;
  const size_t np = pre.size();
  HIP_TRY(ctx->d_err.ensure(kErrBytes));
  HIP_TRY(hipMemsetAsync(ctx->d_err.p, 0, sizeof(int32_t) * kErrWords, st));
  if (np == 0) return TRACYHIP_OK;
  // (a staging block of its own: run_front fills h_desc while this copy may still be queued)
  HIP_TRY(ctx->h_pre.ensure(sizeof(PairDesc) * np));
  PairDesc* hd = static_cast<PairDesc*>(ctx->h_pre.p);
  uint64_t cells[kHostThreads] = {}, bytes[kHostThreads] = {};
  parallel_for((uint32_t)np, [&](uint32_t lo, uint32_t hi, uint32_t tid) {
    uint64_t c = 0, b = 0;
    for (uint32_t i = lo; i < hi; ++i) {
      hd[i] = pre[i];
      c += (uint64_t)std::min<uint32_t>(pre[i].m, kFrontRows) * pre[i].n;
      b += (uint64_t)kFrontRows + pre[i].n + 4ull * pre[i].n;
    }
    cells[tid] = c; bytes[tid] = b;
  });
  HIP_TRY(ctx->d_pre.ensure(sizeof(PairDesc) * np));
  HIP_TRY(hipMemcpyAsync(ctx->d_pre.p, hd, sizeof(PairDesc) * np, hipMemcpyHostToDevice, st));
  DpArgs a{};
  a.pairs = static_cast<const PairDesc*>(ctx->d_pre.p);
  a.a1 = d_a1; a.a2 = d_a2; a.scores = nullptr; a.err = static_cast<int32_t*>(ctx->d_err.p);
  a.match = prm->match; a.mismatch = prm->mismatch; a.go = prm->go; a.ge = prm->ge; a.hfree = prm->hfree; a.vfree = prm->vfree;
  a.qlimit = sub_limit(prm);
  a.special_blocks = d_special;
  a.lastrow = d_lastrow;
  uint64_t tc = 0, tb = 0;
  for (uint32_t t = 0; t < kHostThreads; ++t) { tc += cells[t]; tb += bytes[t]; }
  int trc;
  if ((trc = timing_begin(ctx, TRACYHIP_TIMER_SCORE, tc, tb))) return trc;
  HIP_TRY(launch_gotoh_front_prefix_cq(a, (uint32_t)np, st));
  if ((trc = timing_end(ctx))) return trc;
  return TRACYHIP_OK;
}

// one tier of run_front: place, band below row R on strips of KB rows and the diagonals c* +- halfw, certify
static int run_front_once(tracyhip_ctx* ctx, const std::vector<FrontDesc>& fd, const int16_t* d_qp, const uint32_t* d_row, const tracyhip_params* prm,
                          FrontResult& out, const uint8_t* d_codes, bool keep_err, int KB, int32_t halfw) {
  hipStream_t st = ctx->stream;
  const size_t nf = fd.size();
  out.fo.assign(nf, FrontOut{});
  out.score.assign(nf, 0);
  out.ce.assign(nf, 0);
  if (nf == 0) return TRACYHIP_OK;
  const size_t per = sizeof(FrontDesc) + sizeof(PairDesc) + sizeof(FrontOut) + sizeof(int32_t) + 2 * sizeof(uint32_t);
  HIP_TRY(ctx->d_front.ensure(per * nf + 64));
  PairDesc* d_pairs = static_cast<PairDesc*>(ctx->d_front.p);
  FrontDesc* d_fd = reinterpret_cast<FrontDesc*>(d_pairs + nf);
  FrontOut* d_fo = reinterpret_cast<FrontOut*>(d_fd + nf);
  int32_t* d_fs = reinterpret_cast<int32_t*>(d_fo + nf);
  uint32_t* d_fe = reinterpret_cast<uint32_t*>(d_fs + nf);
  HIP_TRY(ctx->h_desc.ensure(sizeof(FrontDesc) * nf));
  std::memcpy(ctx->h_desc.p, fd.data(), sizeof(FrontDesc) * nf);
  HIP_TRY(hipMemcpyAsync(d_fd, ctx->h_desc.p, sizeof(FrontDesc) * nf, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx->d_err.ensure(kErrBytes));
  if (!keep_err) HIP_TRY(hipMemsetAsync(ctx->d_err.p, 0, sizeof(int32_t) * kErrWords, st));
  uint32_t max_rest = 0;
  uint64_t cells = 0, bytes = 0;
  {
    uint32_t mr[kHostThreads] = {};
    uint64_t cs[kHostThreads] = {}, bs[kHostThreads] = {};
    parallel_for((uint32_t)nf, [&](uint32_t lo, uint32_t hi, uint32_t tid) {
      uint32_t m0 = 0;
      uint64_t c0 = 0, b0 = 0;
      for (uint32_t i = lo; i < hi; ++i) {
        const FrontDesc& f = fd[i];
        m0 = std::max(m0, f.m_rest);
        c0 += (uint64_t)b16_strips(f.m_rest, KB) * KB * (uint64_t)(KB + 2 * halfw);
        b0 += 12ull * f.m_rest + f.m_rest + 2ull * halfw + 4ull * (2 * halfw + KB) + 8ull * f.n;  // tables, codes, the kept row (twice: place, certify)
      }
      mr[tid] = m0; cs[tid] = c0; bs[tid] = b0;
    });
    for (uint32_t t = 0; t < kHostThreads; ++t) { max_rest = std::max(max_rest, mr[t]); cells += cs[t]; bytes += bs[t]; }
  }
  Band16Args a{};
  a.pairs = d_pairs; a.npairs = (uint32_t)nf; a.qp = d_qp; a.codes = d_codes ? d_codes : ctx->codes(); a.scores = d_fs; a.ends = d_fe;
  a.err = static_cast<int32_t*>(ctx->d_err.p); a.go = prm->go; a.ge = prm->ge; a.hfree = 1; a.row = d_row;
  a.code_cap = (max_rest + 2u * (uint32_t)halfw + 16u) & ~3u;  // front_place_body: a sub-window is at most m_rest + 2 halfw + 2 columns
  if (4ull * a.code_cap + b16_table_bytes(KB) + 32ull * kB16RowCap > 64u * 1024u)
    return set_error(TRACYHIP_ERR_RANGE, "run_front: traces of %u rows do not fit the staging area", max_rest);
  int trc;
  if ((trc = timing_begin(ctx, TRACYHIP_TIMER_FRONT, cells, bytes))) return trc;
  HIP_TRY(launch_front_place(d_fd, (uint32_t)nf, d_row, prm->go + prm->ge, halfw, d_pairs, d_fo, st));
  HIP_TRY(launch_band16_cont(KB, a, st, !ctx->knobs.no_cont16));  // (run_front's callers are in the 16-bit domain: the prefix rows above were swept there)
  HIP_TRY(launch_front_certify(d_fd, (uint32_t)nf, d_row, prm->go, prm->ge, halfw, d_fs, d_fe, d_fo, st));
  if ((trc = timing_end(ctx))) return trc;
  std::vector<uint32_t> h_fe(2 * nf);
  int32_t herr[kErrWords] = {};
  HIP_TRY(hipMemcpyAsync(out.fo.data(), d_fo, sizeof(FrontOut) * nf, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(out.score.data(), d_fs, sizeof(int32_t) * nf, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(h_fe.data(), d_fe, sizeof(uint32_t) * 2 * nf, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(herr, ctx->d_err.p, sizeof(herr), hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx_sync(ctx));
  timing_collect(ctx);
  if (herr[0] & 1) return set_error(TRACYHIP_ERR_RANGE, "a query-profile score does not fit the int16 table (profile values too large)");
  for (size_t i = 0; i < nf; ++i) out.ce[i] = h_fe[2 * i + 1] ? h_fe[2 * i + 1] + out.fo[i].shift : 0u;
  return TRACYHIP_OK;
}

// Two tiers: strips of 8 rows on the diagonals c* +- 60 first (a pair that lost less than ~250 against its row maxima certifies
// there, and a step of 8 cells costs two thirds of a step of 12), then -- for what did not certify -- the widest band one period
// holds (kFrontK = 12, kFrontHalfW = 90).  What fails both is the caller's to sweep in full.
int run_front(tracyhip_ctx* ctx, const std::vector<FrontDesc>& fd, const int16_t* d_qp, const uint32_t* d_row, const tracyhip_params* prm,
              FrontResult& out, const uint8_t* d_codes, bool keep_err) {
  int rc;
  if ((rc = run_front_once(ctx, fd, d_qp, d_row, prm, out, d_codes, keep_err, 8, 60))) return rc;
  std::vector<FrontDesc> again;
  std::vector<uint32_t> idx;
  for (size_t i = 0; i < fd.size(); ++i)
    if (!(out.fo[i].ok && out.ce[i])) { FrontDesc f = fd[i]; f.out = (uint32_t)again.size(); again.push_back(f); idx.push_back((uint32_t)i); }
  if (again.empty()) return TRACYHIP_OK;
  FrontResult wide;
  if ((rc = run_front_once(ctx, again, d_qp, d_row, prm, wide, d_codes, false, kFrontK, kFrontHalfW))) return rc;
  for (size_t q = 0; q < idx.size(); ++q) { out.fo[idx[q]] = wide.fo[q]; out.score[idx[q]] = wide.score[q]; out.ce[idx[q]] = wide.ce[q]; }
  return TRACYHIP_OK;
}

// validate a pair list and turn it into device-side descriptors + staged payloads
int build_problem(tracyhip_ctx* ctx, const tracyhip_pairs* pairs, int mem, bool needle, DpProblem& pb, uint64_t* max_mn) {
  if (!pairs) return set_error(TRACYHIP_ERR_ARG, "null pairs");
  const tracyhip_seqset& s1 = pairs->a1;
  const tracyhip_seqset& s2 = pairs->a2;
  if (pairs->npairs && (!s1.offset || !s1.length || !s2.offset || !s2.length))
    return set_error(TRACYHIP_ERR_ARG, "null offset/length arrays");
  if (s1.kind == TRACYHIP_SEQ_CHAR && s2.kind == TRACYHIP_SEQ_CHAR) pb.mode = MODE_CHAR;
  else if (s1.kind == TRACYHIP_SEQ_PROFILE && s2.kind == TRACYHIP_SEQ_CHAR) pb.mode = MODE_QP;
  else if (s1.kind == TRACYHIP_SEQ_PROFILE && s2.kind == TRACYHIP_SEQ_PROFILE) pb.mode = MODE_PROF;
  else return set_error(TRACYHIP_ERR_ARG, "unsupported sequence kinds (a1 CHAR with a2 PROFILE)");
  if (needle && pb.mode == MODE_QP)
    return set_error(TRACYHIP_ERR_ARG, "needle takes two strings or two profiles (needle.h:12-14)");
  pb.a1_profile = s1.kind == TRACYHIP_SEQ_PROFILE;
  pb.a2_profile = s2.kind == TRACYHIP_SEQ_PROFILE;
  const uint64_t e1 = seqset_extent(s1), e2 = seqset_extent(s2);
  if ((e1 && !s1.data) || (e2 && !s2.data)) return set_error(TRACYHIP_ERR_ARG, "null sequence data");
  int rc;
  if ((rc = stage_in(ctx, ctx->d_in1, s1.data, e1 * (pb.a1_profile ? 4 : 1), mem, &pb.d_a1))) return rc;
  if ((rc = stage_in(ctx, ctx->d_in2, s2.data, e2 * (pb.a2_profile ? 4 : 1), mem, &pb.d_a2))) return rc;
  pb.d_a2_chars = pb.d_a2;
  if (pb.mode == MODE_QP && e2) {  // reference characters -> profile-row codes (align.h:121-136)
    HIP_TRY(ctx->ensure_codes(e2, ctx->stream));
    hipLaunchKernelGGL(encode_kernel, dim3((unsigned)((e2 + 255) / 256)), dim3(256), 0, ctx->stream,
                       static_cast<const uint8_t*>(pb.d_a2), ctx->codes(), e2, ctx->special_blocks());
    HIP_TRY(hipGetLastError());
    pb.d_a2 = ctx->codes();
  }
  // profile x profile (Gotoh): classify the sequences once, so that pairs whose two profiles have an all-zero row 4 -- trace
  // profiles always do, profile.h:37-38 -- run the 16-term kernel and the others the 25-term one
  std::vector<uint8_t> z1, z2;
  if (pb.mode == MODE_PROF && !needle && pairs->npairs) {
    const uint32_t n1 = s1.count, n2 = s2.count;
    std::vector<Row4Desc> hd(n1 + n2);
    for (uint32_t i = 0; i < n1; ++i) hd[i] = Row4Desc{s1.offset[i], s1.length[i], 0};
    for (uint32_t i = 0; i < n2; ++i) hd[n1 + i] = Row4Desc{s2.offset[i], s2.length[i], 0};
    HIP_TRY(ctx->d_tmp[0].ensure(sizeof(Row4Desc) * hd.size() + hd.size()));
    Row4Desc* dd = static_cast<Row4Desc*>(ctx->d_tmp[0].p);
    uint8_t* dz = reinterpret_cast<uint8_t*>(dd + hd.size());
    HIP_TRY(hipMemcpyAsync(dd, hd.data(), sizeof(Row4Desc) * hd.size(), hipMemcpyHostToDevice, ctx->stream));
    // column classes of the a2 set for the screened substitution score (one byte per float of the set: indexed like row 0)
    uint8_t* colclass = nullptr;
    if (n2 && e2 && !ctx->knobs.no_screen) {
      HIP_TRY(ctx->ensure_codes(e2, ctx->stream));
      colclass = ctx->codes();
      pb.d_colclass = colclass;
    }
    if (n1) hipLaunchKernelGGL(row4_zero_kernel, dim3(n1), dim3(64), 0, ctx->stream, dd, static_cast<const float*>(pb.d_a1), dz, (uint8_t*)nullptr);
    if (n2) hipLaunchKernelGGL(row4_zero_kernel, dim3(n2), dim3(64), 0, ctx->stream, dd + n1, static_cast<const float*>(pb.d_a2), dz + n1, colclass);
    HIP_TRY(hipGetLastError());
    std::vector<uint8_t> hz(n1 + n2);
    HIP_TRY(hipMemcpyAsync(hz.data(), dz, hz.size(), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx_sync(ctx));
    z1.assign(hz.begin(), hz.begin() + n1);
    z2.assign(hz.begin() + n1, hz.end());
  }
  pb.desc.resize(pairs->npairs);
  pb.k.resize(pairs->npairs);
  *max_mn = 0;
  // strip height per a1 sequence (not per pair: an all-pairs list names every sequence a thousand times)
  std::vector<int> kseq(s1.count);
  for (uint32_t i = 0; i < s1.count; ++i) kseq[i] = choose_k(s1.length[i], pb.mode, needle);
  // long pair lists are filled by a few threads (the list of an all-pairs job is 36 MB of descriptors)
  const uint32_t npairs = pairs->npairs;
  const uint32_t nthr = npairs >= (1u << 16) ? 8u : 1u;
  std::vector<uint64_t> tmax(nthr, 0);
  std::vector<uint32_t> tbad(nthr, ~0u);
  auto fill = [&](uint32_t tid) {
    const uint32_t lo = (uint32_t)((uint64_t)npairs * tid / nthr), hi = (uint32_t)((uint64_t)npairs * (tid + 1) / nthr);
    uint64_t mx = 0;
    for (uint32_t i = lo; i < hi; ++i) {
      const uint32_t i1 = pairs->a1_index ? pairs->a1_index[i] : i;
      const uint32_t i2 = pairs->a2_index ? pairs->a2_index[i] : i;
      if (i1 >= s1.count || i2 >= s2.count) { tbad[tid] = i; return; }
      PairDesc d{};
      d.a1_off = s1.offset[i1];
      d.a2_off = s2.offset[i2];
      d.m = s1.length[i1];
      d.n = s2.length[i2];
      d.a1_stride = d.m;
      d.a2_stride = d.n;
      d.out = i;
      if (!z1.empty() && z1[i1] && z2[i2]) d.flags |= PAIR_ROW4_ZERO;
      pb.desc[i] = d;
      pb.k[i] = kseq[i1];
      mx = std::max<uint64_t>(mx, (uint64_t)d.m + d.n);
    }
    tmax[tid] = mx;
  };
  if (nthr == 1) fill(0);
  else {
    std::vector<std::thread> th;
    for (uint32_t t = 0; t < nthr; ++t) th.emplace_back(fill, t);
    for (auto& t : th) t.join();
  }
  for (uint32_t t = 0; t < nthr; ++t) {
    if (tbad[t] != ~0u) return set_error(TRACYHIP_ERR_ARG, "pair %u indexes past the sequence sets", tbad[t]);
    *max_mn = std::max(*max_mn, tmax[t]);
  }
  return TRACYHIP_OK;
}

}  // namespace tracyhip

// ====================================================================================================
namespace tracyhip {
// Checkpointed 16-bit score sweeps of `full` and prefix bounds of `pre` in ONE launch (strand by certificate with the
// orientation voted beforehand, pipeline.hip).  Profile x code pairs of one strip height K; scores land at PairDesc::out.
int run_ckpt_prefix(tracyhip_ctx* ctx, const void* d_a1, const void* d_a2, const std::vector<PairDesc>& full, const std::vector<int>& fullk,
                    const std::vector<PairDesc>& pre, const tracyhip_params* prm, int32_t* d_scores, DpCkpt* ck, bool front_shape) {
  hipStream_t st = ctx->stream;
  const size_t nf = full.size(), np = pre.size();
  if (nf + np == 0) return TRACYHIP_OK;
  auto hs1 = std::make_unique<HostScope>("run_ckpt_prefix.plan");
  HIP_TRY(ctx->h_desc.ensure(sizeof(PairDesc) * (nf + np)));
  PairDesc* hd = static_cast<PairDesc*>(ctx->h_desc.p);
  // the sweeps by strip height (one launch each; the prefix workgroups ride with the first), longest first inside a launch, as
  // run_dp orders them
  std::vector<uint32_t> order(nf);
  for (uint32_t i = 0; i < nf; ++i) order[i] = i;
  auto before = [&](uint32_t x, uint32_t y) {
    if (fullk[x] != fullk[y]) return fullk[x] > fullk[y];
    return (uint64_t)full[x].m * full[x].n > (uint64_t)full[y].m * full[y].n;
  };
  if (!std::is_sorted(order.begin(), order.end(), before)) std::stable_sort(order.begin(), order.end(), before);
  for (size_t i = 0; i < nf; ++i) hd[i] = full[order[i]];
  for (size_t i = 0; i < np; ++i) hd[nf + i] = pre[i];
  HIP_TRY(ctx->d_desc.ensure(sizeof(PairDesc) * (nf + np)));
  HIP_TRY(hipMemcpyAsync(ctx->d_desc.p, hd, sizeof(PairDesc) * (nf + np), hipMemcpyHostToDevice, st));
  HIP_TRY(ctx->d_err.ensure(kErrBytes));
  HIP_TRY(hipMemsetAsync(ctx->d_err.p, 0, sizeof(int32_t) * kErrWords, st));
  hs1.reset();
  DpArgs a{};
  a.a1 = d_a1; a.a2 = d_a2; a.scores = d_scores; a.err = static_cast<int32_t*>(ctx->d_err.p);
  a.match = prm->match; a.mismatch = prm->mismatch; a.go = prm->go; a.ge = prm->ge; a.hfree = prm->hfree; a.vfree = prm->vfree;
  a.qlimit = sub_limit(prm);
  if (d_a2 == ctx->codes() && !ctx->knobs.no_compact) a.special_blocks = ctx->special_blocks();
  a.ckpt = ck->d_ckpt; a.lastrow = ck->d_lastrow; a.ckpt_B = ck->B; a.ckpt_narrow = 1;
  DpArgs ap = a;
  ap.pairs = static_cast<const PairDesc*>(ctx->d_desc.p) + nf;
  std::vector<std::pair<uint32_t, int>> narrow_launches;
  uint64_t max_mn = 0;
  int trc;
  size_t lo = 0;
  bool pre_done = np == 0;
  while (lo < nf || !pre_done) {
    const int K = lo < nf ? fullk[order[lo]] : (front_shape ? 15 : fullk.empty() ? 15 : fullk[0]);
    size_t hi = lo;
    while (hi < nf && fullk[order[hi]] == K) ++hi;
    const uint32_t npre = pre_done ? 0u : (uint32_t)np;
    const uint32_t prows = front_shape ? kFrontRows : (uint32_t)kPrefixLanes * (uint32_t)K;
    uint32_t maxm = 0;
    if (ctx->timing) {
      uint64_t cells = 0, bytes = 0;
      for (size_t i = lo; i < hi; ++i) { cells += (uint64_t)hd[i].m * hd[i].n; bytes += 24ull * hd[i].m + hd[i].n + 4; }
      for (size_t i = 0; i < npre; ++i) cells += (uint64_t)std::min<uint32_t>(hd[nf + i].m, prows) * hd[nf + i].n;
      if ((trc = timing_begin(ctx, TRACYHIP_TIMER_SCORE, cells, bytes))) return trc;
    }
    for (size_t i = lo; i < hi; ++i) { maxm = std::max(maxm, hd[i].m); max_mn = std::max<uint64_t>(max_mn, (uint64_t)hd[i].m + hd[i].n); }
    for (size_t i = 0; i < npre; ++i) max_mn = std::max<uint64_t>(max_mn, (uint64_t)hd[nf + i].m + hd[nf + i].n);
    DpArgs af = a;
    af.pairs = static_cast<const PairDesc*>(ctx->d_desc.p) + lo;
    if (front_shape) HIP_TRY(launch_gotoh_ckpt_front(K, af, (uint32_t)(hi - lo), ap, npre, st));
    else HIP_TRY(launch_gotoh_ckpt_prefix(K, af, (uint32_t)(hi - lo), ap, npre, st));
    if ((trc = timing_end(ctx))) return trc;
    if (hi > lo) narrow_launches.emplace_back(maxm, K);
    if (npre) narrow_launches.emplace_back(prows, front_shape ? kFrontPrefixK : K);  // (the prefix is a sweep of its own rows)
    pre_done = true;
    lo = hi;
  }
  int32_t herr[kErrWords] = {};
  HIP_TRY(hipMemcpyAsync(herr, ctx->d_err.p, sizeof(herr), hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx_sync(ctx));
  timing_collect(ctx);
  return range_verdict(prm, herr, narrow_launches, max_mn, 0);  // kWiden: the pipeline restarts on the int32 kernels
}
}  // namespace tracyhip

extern "C" {

const char* tracyhip_last_error(void) { return g_last_error.c_str(); }
const char* tracyhip_version(void) { return "tracy_amd 0.1 (gfx950)"; }

int tracyhip_device_count(int* count) {
  if (!count) return set_error(TRACYHIP_ERR_ARG, "null count");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    *count = 0;
    return set_error(TRACYHIP_ERR_NODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *count = n;
  return TRACYHIP_OK;
}

int tracyhip_create(int device, tracyhip_ctx** out) {
  if (!out) return set_error(TRACYHIP_ERR_ARG, "null out pointer");
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return set_error(TRACYHIP_ERR_NODEVICE, "no HIP device visible (this library has no CPU fallback)");
  }
  if (device < 0 || device >= n) return set_error(TRACYHIP_ERR_ARG, "device %d out of range [0,%d)", device, n);
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return set_error(TRACYHIP_ERR_NODEVICE, "device %d is %s; the kernels are built for gfx950 only", device, prop.gcnArchName);
  HIP_TRY(hipSetDevice(device));
  // The pipelines build megabytes of descriptors between two launches (72 bytes per trace and stage) in vectors that live for one
  // stage.  Above glibc's mmap threshold every such vector is mapped and unmapped anew -- a page fault per 4 KB, tens of milliseconds
  // per 100 000-trace step, and whether the allocator does it varies from process to process.  tracyhip_tune_host_allocator() keeps
  // blocks of up to 32 MB on the heap and the heap's top untrimmed; it is process-wide and therefore the application's to call
  // (the CLI and bench.py do; TRACYHIP_MALLOPT=1 does it here).  The stream-ordered pipelines build no such vectors.
  static const bool malloc_tuned = []() {  // opt-in (process-wide): tracyhip_tune_host_allocator()
    const char* e = getenv("TRACYHIP_MALLOPT");
    if (e && atoi(e) != 0) tracyhip_tune_host_allocator();
    return true;
  }();
  (void)malloc_tuned;
  tracyhip_ctx* c = new tracyhip_ctx();
  c->device = device;
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete c;
    return set_error(TRACYHIP_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
  }
  c->own_stream = c->stream;
  c->b16_fork_ok = c->b16_fork.create() == hipSuccess;  // (without them the band stages queue their launches in a row)
  if (!c->b16_fork_ok) c->b16_fork.destroy();
  knobs_from_env(c->knobs);
  *out = c;
  return TRACYHIP_OK;
}

int tracyhip_destroy(tracyhip_ctx* c) {
  if (!c) return TRACYHIP_OK;
  if (c->async) {  // finish queued calls, then stop the worker
    (void)async_drain(c);
    { std::lock_guard<std::mutex> lk(c->async->m); c->async->stop = true; }
    c->async->cv.notify_all();
    c->async->worker.join();
    delete c->async;
    c->async = nullptr;
  }
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  for (auto* l : c->lanes) tracyhip_destroy(l);
  c->lanes.clear();
  c->release_all();
  for (auto& p : c->pending) { (void)hipEventDestroy(p.e0); (void)hipEventDestroy(p.e1); }
  for (auto e : c->free_events) (void)hipEventDestroy(e);
  c->b16_fork.destroy();
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
  return TRACYHIP_OK;
}

int tracyhip_set_option(tracyhip_ctx* c, const char* name, const char* value) {
  if (!c || !name || !value) return set_error(TRACYHIP_ERR_ARG, "null context / name / value");
  if (!knobs_set(c->knobs, name, value)) return set_error(TRACYHIP_ERR_ARG, "unknown option or bad value: %s=%s", name, value);
  for (auto* l : c->lanes) l->knobs = c->knobs;
  return TRACYHIP_OK;
}
int tracyhip_describe(tracyhip_ctx* c, char* buf, size_t cap) {
  if (!c) return set_error(TRACYHIP_ERR_ARG, "null context");
  const std::string s = knobs_describe(c->knobs) + "device=" + std::to_string(c->device) + "\nlanes=" + std::to_string(c->lanes.size() + 1) +
                        "\nworkspace_limit=" + std::to_string(c->ws_limit) + "\n";
  if (buf && cap) {
    const size_t n = std::min(cap - 1, s.size());
    std::memcpy(buf, s.data(), n);
    buf[n] = 0;
  }
  return (int)s.size() + 1;
}
int tracyhip_tune_host_allocator(void) {
  static const bool once = []() {
    mallopt(M_MMAP_THRESHOLD, 32 << 20);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    mallopt(M_TOP_PAD, 64 << 20);
    return true;
  }();
  (void)once;
  return TRACYHIP_OK;
}
int tracyhip_last_call_stats(tracyhip_ctx* c, tracyhip_call_stats* out) {
  if (!c || !out) return set_error(TRACYHIP_ERR_ARG, "null context / out");
  *out = c->stats;
  for (auto* l : c->lanes) {  // (a call split over lanes: every lane counted its chunk)
    if (l->stats.traces == 0) continue;  // a lane without a chunk (small batches run on one) says nothing about how the call went
    const uint32_t* a = reinterpret_cast<const uint32_t*>(&l->stats);
    uint32_t* o = reinterpret_cast<uint32_t*>(out);
    for (size_t i = 0; i < sizeof(tracyhip_call_stats) / sizeof(uint32_t); ++i) {
      if (i == offsetof(tracyhip_call_stats, traces) / sizeof(uint32_t)) continue;  // (the call's count is the context's)
      if (i != offsetof(tracyhip_call_stats, stream_ordered) / sizeof(uint32_t)) o[i] += a[i];
      else o[i] = o[i] && a[i];
    }
  }
  return TRACYHIP_OK;
}

int tracyhip_set_stream(tracyhip_ctx* c, void* s) {
  if (!c) return set_error(TRACYHIP_ERR_ARG, "null context");
  c->stream = s ? static_cast<hipStream_t>(s) : c->own_stream;
  return TRACYHIP_OK;
}

int tracyhip_set_workspace_limit(tracyhip_ctx* c, uint64_t bytes) {
  if (!c) return set_error(TRACYHIP_ERR_ARG, "null context");
  c->ws_limit = bytes;
  c->ws_cache_budget = 0;  // (stream.hip workspace_budget: a kept answer belongs to the setting it was made under)
  for (auto* l : c->lanes) { l->ws_limit = bytes; l->ws_cache_budget = 0; }
  return TRACYHIP_OK;
}

int tracyhip_set_lanes(tracyhip_ctx* c, uint32_t n) {
  if (!c) return set_error(TRACYHIP_ERR_ARG, "null context");
  if (n < 1 || n > 8) return set_error(TRACYHIP_ERR_ARG, "lanes must be in [1, 8]");
  int rc = ctx_begin(c);
  if (rc) return rc;
  const size_t want = n - 1;  // the context itself is the first lane
  while (c->lanes.size() > want) { tracyhip_destroy(c->lanes.back()); c->lanes.pop_back(); }
  while (c->lanes.size() < want) {
    tracyhip_ctx* l = nullptr;
    if ((rc = tracyhip_create(c->device, &l))) return rc;
    l->ws_limit = c->ws_limit;
    l->timing = c->timing;
    l->knobs = c->knobs;
    c->lanes.push_back(l);
  }
  return TRACYHIP_OK;
}

int tracyhip_timing_enable(tracyhip_ctx* c, int on) {
  if (!c) return set_error(TRACYHIP_ERR_ARG, "null context");
  c->timing = on != 0;
  for (auto* l : c->lanes) l->timing = c->timing;
  return TRACYHIP_OK;
}
int tracyhip_timing_reset(tracyhip_ctx* c) {
  if (!c) return set_error(TRACYHIP_ERR_ARG, "null context");
  for (auto& a : c->acc) a = tracyhip_kernel_timing{};
  for (auto* l : c->lanes) tracyhip_timing_reset(l);
  return TRACYHIP_OK;
}
int tracyhip_timing_get(tracyhip_ctx* c, int which, tracyhip_kernel_timing* out) {
  if (!c || !out || which < 0 || which >= TRACYHIP_TIMER_COUNT) return set_error(TRACYHIP_ERR_ARG, "bad timing query");
  *out = c->acc[which];
  for (auto* l : c->lanes) {  // kernels of different lanes overlap on the device: their durations add up here
    out->ms += l->acc[which].ms; out->launches += l->acc[which].launches;
    out->cells += l->acc[which].cells; out->bytes += l->acc[which].bytes;
  }
  return TRACYHIP_OK;
}

int tracyhip_synchronize(tracyhip_ctx* c) {
  if (!c) return set_error(TRACYHIP_ERR_ARG, "null context");
  const int arc = async_drain(c);  // queued *_async calls first; their first error is what this call reports
  std::string amsg;
  if (arc) amsg = tracyhip_last_error();
  int rc = ctx_begin(c);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (arc) return set_error(arc, "%s", amsg.c_str());
  return TRACYHIP_OK;
}

int tracyhip_gotoh_score_async(tracyhip_ctx* ctx, const tracyhip_pairs* pairs, const tracyhip_params* prm, int mem, int32_t* scores) {
  if (!ctx || !pairs || !prm) return set_error(TRACYHIP_ERR_ARG, "null context / pairs / params");
  const tracyhip_pairs p = *pairs;
  const tracyhip_params q = *prm;
  return async_submit(ctx, [=]() { return tracyhip_gotoh_score(ctx, &p, &q, mem, scores); });
}
int tracyhip_gotoh_align_async(tracyhip_ctx* ctx, const tracyhip_pairs* pairs, const tracyhip_params* prm, int mem, int32_t* scores,
                               uint8_t* ops, const uint64_t* ops_offset, uint32_t* ops_len) {
  if (!ctx || !pairs || !prm) return set_error(TRACYHIP_ERR_ARG, "null context / pairs / params");
  const tracyhip_pairs p = *pairs;
  const tracyhip_params q = *prm;
  return async_submit(ctx, [=]() { return tracyhip_gotoh_align(ctx, &p, &q, mem, scores, ops, ops_offset, ops_len); });
}
int tracyhip_align_traces_async(tracyhip_ctx* ctx, const tracyhip_align_job* job, const tracyhip_params* prm, int mem,
                                const tracyhip_align_result* out) {
  if (!ctx || !job || !prm || !out) return set_error(TRACYHIP_ERR_ARG, "null context / job / params / result");
  const tracyhip_align_job j = *job;
  const tracyhip_params q = *prm;
  const tracyhip_align_result o = *out;
  return async_submit(ctx, [=]() { return tracyhip_align_traces(ctx, &j, &q, mem, &o); });
}
int tracyhip_decompose_traces_async(tracyhip_ctx* ctx, const tracyhip_decompose_job* job, const tracyhip_params* prm, int mem,
                                    const tracyhip_decompose_result* out) {
  if (!ctx || !job || !prm || !out) return set_error(TRACYHIP_ERR_ARG, "null context / job / params / result");
  const tracyhip_decompose_job j = *job;
  const tracyhip_params q = *prm;
  const tracyhip_decompose_result o = *out;
  return async_submit(ctx, [=]() { return tracyhip_decompose_traces(ctx, &j, &q, mem, &o); });
}

static int dp_entry(tracyhip_ctx* ctx, const tracyhip_pairs* pairs, const tracyhip_params* prm, int mem, bool needle,
                    bool trace, int32_t* scores, uint8_t* ops, const uint64_t* ops_offset, uint32_t* ops_len) {
  int rc = ctx_begin(ctx);
  if (rc) return rc;
  if (mem != TRACYHIP_MEM_HOST && mem != TRACYHIP_MEM_DEVICE) return set_error(TRACYHIP_ERR_ARG, "bad mem kind");
  if (!pairs) return set_error(TRACYHIP_ERR_ARG, "null pairs");
  if (!trace && !scores && pairs->npairs) return set_error(TRACYHIP_ERR_ARG, "null scores");
  if (trace && pairs->npairs && (!ops || !ops_offset || !ops_len)) return set_error(TRACYHIP_ERR_ARG, "null ops/ops_offset/ops_len");
  DpProblem pb;
  DpProblemLease lease(ctx, pb);
  uint64_t max_mn = 0;
  if ((rc = build_problem(ctx, pairs, mem, needle, pb, &max_mn))) return rc;
  if ((rc = check_params(prm, max_mn))) return rc;
  const uint32_t np = pairs->npairs;
  if (np == 0) return TRACYHIP_OK;
  hipStream_t st = ctx->stream;

  int32_t* d_scores = scores;
  uint8_t* d_ops = ops;
  uint32_t* d_len = ops_len;
  uint64_t ops_total = 0;
  if (trace)
    for (uint32_t i = 0; i < np; ++i) ops_total = std::max<uint64_t>(ops_total, ops_offset[i] + pb.desc[i].m + pb.desc[i].n);
  if (mem == TRACYHIP_MEM_HOST) {
    if (scores) { HIP_TRY(ctx->d_scores.ensure(sizeof(int32_t) * (size_t)np)); d_scores = static_cast<int32_t*>(ctx->d_scores.p); }
    if (trace) {
      HIP_TRY(ctx->d_ops.ensure(ops_total ? ops_total : 1));
      HIP_TRY(ctx->d_ops_len.ensure(sizeof(uint32_t) * (size_t)np));
      d_ops = static_cast<uint8_t*>(ctx->d_ops.p);
      d_len = static_cast<uint32_t*>(ctx->d_ops_len.p);
    }
  }
  const uint64_t* d_off = nullptr;
  if (trace) {
    HIP_TRY(ctx->h_off.ensure(sizeof(uint64_t) * (size_t)np));
    std::memcpy(ctx->h_off.p, ops_offset, sizeof(uint64_t) * (size_t)np);
    HIP_TRY(ctx->d_ops_off.ensure(sizeof(uint64_t) * (size_t)np));
    HIP_TRY(hipMemcpyAsync(ctx->d_ops_off.p, ctx->h_off.p, sizeof(uint64_t) * (size_t)np, hipMemcpyHostToDevice, st));
    d_off = static_cast<const uint64_t*>(ctx->d_ops_off.p);
  }
  if ((rc = run_dp(ctx, pb, prm, needle, trace, d_scores, d_ops, d_off, d_len))) return rc;
  if (mem == TRACYHIP_MEM_HOST) {
    if (scores) HIP_TRY(hipMemcpyAsync(scores, d_scores, sizeof(int32_t) * (size_t)np, hipMemcpyDeviceToHost, st));
    if (trace) {
      if (ops_total) HIP_TRY(hipMemcpyAsync(ops, d_ops, ops_total, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(ops_len, d_len, sizeof(uint32_t) * (size_t)np, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(ctx_sync(ctx));
  }
  return TRACYHIP_OK;
}

// case-sensitive column codes of a string (MODE_CQ, dp_kernels.h cq_code)
__global__ void cq_rows_check_kernel(const uint8_t* __restrict__ in, uint64_t n, int32_t* flag) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && !tracyhip::cq_row_char(in[i])) atomicOr(flag, 1);
}
__global__ void encode_cq_codes_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (uint8_t)tracyhip::cq_code(in[i]);
}

int tracyhip_gotoh_banded(tracyhip_ctx* ctx, const tracyhip_pairs* pairs, const tracyhip_params* prm, const int32_t* band_lo, const int32_t* band_hi,
                          int mem, int32_t* scores, uint8_t* ops, const uint64_t* ops_offset, uint32_t* ops_len, uint32_t* ends) {
  int rc = ctx_begin(ctx);
  if (rc) return rc;
  if (mem != TRACYHIP_MEM_HOST && mem != TRACYHIP_MEM_DEVICE) return set_error(TRACYHIP_ERR_ARG, "bad mem kind");
  if (!pairs || !prm || !band_lo || !band_hi) return set_error(TRACYHIP_ERR_ARG, "null pairs / params / band");
  const bool origin = ends != nullptr;
  const uint32_t np = pairs->npairs;
  if (np == 0) return TRACYHIP_OK;
  if (!origin && (!ops || !ops_offset || !ops_len)) return set_error(TRACYHIP_ERR_ARG, "null ops/ops_offset/ops_len");
  if (prm->vfree || prm->go > 0 || prm->ge >= 0) return set_error(TRACYHIP_ERR_ARG, "the band kernels take AlignConfig<.,false>, go <= 0, ge < 0");
  if (origin && !prm->hfree) return set_error(TRACYHIP_ERR_ARG, "the origin-tracking sweep takes AlignConfig<true,false>");
  if (sub_limit(prm) > kWideScore) return set_error(TRACYHIP_ERR_RANGE, "the band kernels take |match|, |mismatch| <= %d (tracyhip_gotoh_align has no such limit)", kWideScore);
  if (pairs->a1.kind == TRACYHIP_SEQ_PROFILE && pairs->a2.kind == TRACYHIP_SEQ_PROFILE) return set_error(TRACYHIP_ERR_ARG, "the band kernels take string or profile rows against a string");
  DpProblem pb;
  DpProblemLease lease(ctx, pb);
  uint64_t max_mn = 0;
  if ((rc = build_problem(ctx, pairs, mem, false, pb, &max_mn))) return rc;  // MODE_QP: a2 encoded to codes 0..5
  if ((rc = check_params(prm, max_mn))) return rc;
  hipStream_t st = ctx->stream;
  const bool strings = pb.mode == MODE_CHAR;
  const uint8_t* d_codes = static_cast<const uint8_t*>(pb.d_a2);
  if (strings) {
    const uint64_t e2 = seqset_extent(pairs->a2);
    HIP_TRY(ctx->ensure_codes(e2 ? e2 : 1, st));
    if (e2) hipLaunchKernelGGL(encode_cq_codes_kernel, dim3((unsigned)((e2 + 255) / 256)), dim3(256), 0, st, static_cast<const uint8_t*>(pb.d_a2), ctx->codes(), e2);
    HIP_TRY(hipGetLastError());
    d_codes = ctx->codes();
  }
  // one table per a1 sequence
  const tracyhip_seqset& s1 = pairs->a1;
  if (strings) {
    // The string tables score a row byte against the column codes A C G T N only (b16_table_row): a row holding any other byte
    // (lower case, IUPAC, '-') would mismatch an identical column byte, where gotoh.h compares bytes (align.h:96-101) -- refused here,
    // tracyhip_gotoh_align takes such strings (the pipelines run the same test before they use this form: cq_rows_kernel)
    const uint64_t e1 = seqset_extent(s1);
    HIP_TRY(ctx->d_err.ensure(kErrBytes));
    int32_t* d_flag = static_cast<int32_t*>(ctx->d_err.p) + kErrVerdictWord;
    HIP_TRY(hipMemsetAsync(d_flag, 0, sizeof(int32_t), st));
    if (e1) hipLaunchKernelGGL(cq_rows_check_kernel, dim3((unsigned)((e1 + 255) / 256)), dim3(256), 0, st, static_cast<const uint8_t*>(pb.d_a1), e1, d_flag);
    HIP_TRY(hipGetLastError());
    int32_t h_flag = 0;
    HIP_TRY(hipMemcpyAsync(&h_flag, d_flag, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx_sync(ctx));
    if (h_flag) return set_error(TRACYHIP_ERR_ARG, "tracyhip_gotoh_banded: string rows must hold A C G T N only (the band kernels score through a five-letter table); use tracyhip_gotoh_align");
  }
  std::vector<B16TableDesc> td(s1.count);
  for (uint32_t i = 0; i < s1.count; ++i) td[i] = B16TableDesc{s1.offset[i], 0, s1.length[i], s1.length[i], 0, 0};
  HIP_TRY(ctx->d_err.ensure(kErrBytes));
  HIP_TRY(hipMemsetAsync(ctx->d_err.p, 0, sizeof(int32_t) * kErrWords, st));
  if ((rc = build_b16_tables(ctx, ctx->d_b16tab[0], pb.d_a1, strings, td, prm))) return rc;
  Band16Job job;
  job.kind = origin ? 1 : 0;
  job.d_qp = static_cast<const int16_t*>(ctx->d_b16tab[0].p);
  job.d_codes = d_codes;
  job.desc.resize(np); job.k.resize(np);
  uint64_t ops_total = 0;
  for (uint32_t i = 0; i < np; ++i) {
    PairDesc d = pb.desc[i];
    if (d.m == 0 || d.n == 0) return set_error(TRACYHIP_ERR_ARG, "pair %u: empty sequence", i);
    const uint32_t i1 = pairs->a1_index ? pairs->a1_index[i] : i;
    const int K = band16_pick_k(band_lo[i], band_hi[i]);
    if (K == 0) return set_error(TRACYHIP_ERR_RANGE, "pair %u: band of %lld diagonals is wider than the band kernels sweep (<= %u)", i,
                                 (long long)band_hi[i] - band_lo[i] + 1, b16_max_window(12) - 12 + 1);
    if (origin && !origin16_ok(prm, d.m, d.n)) return set_error(TRACYHIP_ERR_RANGE, "pair %u outside the origin-tracking sweep's packed fields", i);
    d.a1_off = td[i1].out_off; d.a1_stride = td[i1].stride; d.ckpt_off = band_pack(band_lo[i], band_hi[i]);
    job.desc[i] = d; job.k[i] = K;
    if (!origin) ops_total = std::max<uint64_t>(ops_total, ops_offset[i] + d.m + d.n);
  }
  int32_t* d_scores = scores;
  uint8_t* d_ops = ops;
  uint32_t* d_len = ops_len;
  uint32_t* d_ends = ends;
  if (mem == TRACYHIP_MEM_HOST) {
    if (scores) { HIP_TRY(ctx->d_scores.ensure(sizeof(int32_t) * (size_t)np)); d_scores = static_cast<int32_t*>(ctx->d_scores.p); }
    if (origin) { HIP_TRY(ctx->d_ends.ensure(sizeof(uint32_t) * 2 * (size_t)np)); d_ends = static_cast<uint32_t*>(ctx->d_ends.p); }
    else {
      HIP_TRY(ctx->d_ops.ensure(ops_total ? ops_total : 1));
      HIP_TRY(ctx->d_ops_len.ensure(sizeof(uint32_t) * (size_t)np));
      d_ops = static_cast<uint8_t*>(ctx->d_ops.p);
      d_len = static_cast<uint32_t*>(ctx->d_ops_len.p);
    }
  }
  const uint64_t* d_off = nullptr;
  if (!origin) {
    HIP_TRY(ctx->h_off.ensure(sizeof(uint64_t) * (size_t)np));
    std::memcpy(ctx->h_off.p, ops_offset, sizeof(uint64_t) * (size_t)np);
    HIP_TRY(ctx->d_ops_off.ensure(sizeof(uint64_t) * (size_t)np));
    HIP_TRY(hipMemcpyAsync(ctx->d_ops_off.p, ctx->h_off.p, sizeof(uint64_t) * (size_t)np, hipMemcpyHostToDevice, st));
    d_off = static_cast<const uint64_t*>(ctx->d_ops_off.p);
  }
  rc = run_band16(ctx, job, prm, d_scores, d_ends, d_ops, d_off, d_len);
  if (rc == kWiden) rc = set_error(TRACYHIP_ERR_RANGE, "profile values outside the range of the band kernels");
  if (rc) return rc;
  if (mem == TRACYHIP_MEM_HOST) {
    if (scores) HIP_TRY(hipMemcpyAsync(scores, d_scores, sizeof(int32_t) * (size_t)np, hipMemcpyDeviceToHost, st));
    if (origin) HIP_TRY(hipMemcpyAsync(ends, d_ends, sizeof(uint32_t) * 2 * (size_t)np, hipMemcpyDeviceToHost, st));
    else {
      if (ops_total) HIP_TRY(hipMemcpyAsync(ops, d_ops, ops_total, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(ops_len, d_len, sizeof(uint32_t) * (size_t)np, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(ctx_sync(ctx));
  }
  return TRACYHIP_OK;
}

int tracyhip_gotoh_score(tracyhip_ctx* ctx, const tracyhip_pairs* pairs, const tracyhip_params* prm, int mem, int32_t* scores) {
  return dp_entry(ctx, pairs, prm, mem, false, false, scores, nullptr, nullptr, nullptr);
}
int tracyhip_gotoh_align(tracyhip_ctx* ctx, const tracyhip_pairs* pairs, const tracyhip_params* prm, int mem,
                         int32_t* scores, uint8_t* ops, const uint64_t* ops_offset, uint32_t* ops_len) {
  return dp_entry(ctx, pairs, prm, mem, false, true, scores, ops, ops_offset, ops_len);
}
int tracyhip_needle_score(tracyhip_ctx* ctx, const tracyhip_pairs* pairs, const tracyhip_params* prm, int mem, int32_t* scores) {
  return dp_entry(ctx, pairs, prm, mem, true, false, scores, nullptr, nullptr, nullptr);
}
int tracyhip_needle_align(tracyhip_ctx* ctx, const tracyhip_pairs* pairs, const tracyhip_params* prm, int mem,
                          int32_t* scores, uint8_t* ops, const uint64_t* ops_offset, uint32_t* ops_len) {
  return dp_entry(ctx, pairs, prm, mem, true, true, scores, ops, ops_offset, ops_len);
}

int tracyhip_alignment_rows(tracyhip_ctx* ctx, const tracyhip_pairs* pairs, int mem, const uint8_t* ops,
                            const uint64_t* ops_offset, const uint32_t* ops_len, uint8_t* rows0, uint8_t* rows1) {
  int rc = ctx_begin(ctx);
  if (rc) return rc;
  if (!pairs) return set_error(TRACYHIP_ERR_ARG, "null pairs");
  const uint32_t np = pairs->npairs;
  if (np == 0) return TRACYHIP_OK;
  if (!ops || !ops_offset || !ops_len || !rows0 || !rows1) return set_error(TRACYHIP_ERR_ARG, "null ops/rows");
  // kinds are taken as given here (no query-profile encoding): build descriptors by hand
  const tracyhip_seqset& s1 = pairs->a1;
  const tracyhip_seqset& s2 = pairs->a2;
  const bool p1 = s1.kind == TRACYHIP_SEQ_PROFILE, p2 = s2.kind == TRACYHIP_SEQ_PROFILE;
  hipStream_t st = ctx->stream;
  const void *d_a1, *d_a2;
  if ((rc = stage_in(ctx, ctx->d_in1, s1.data, seqset_extent(s1) * (p1 ? 4 : 1), mem, &d_a1))) return rc;
  if ((rc = stage_in(ctx, ctx->d_in2, s2.data, seqset_extent(s2) * (p2 ? 4 : 1), mem, &d_a2))) return rc;
  HIP_TRY(ctx->h_desc.ensure(sizeof(PairDesc) * (size_t)np));
  PairDesc* hd = static_cast<PairDesc*>(ctx->h_desc.p);
  uint64_t total = 0;
  for (uint32_t i = 0; i < np; ++i) {
    const uint32_t i1 = pairs->a1_index ? pairs->a1_index[i] : i;
    const uint32_t i2 = pairs->a2_index ? pairs->a2_index[i] : i;
    if (i1 >= s1.count || i2 >= s2.count) return set_error(TRACYHIP_ERR_ARG, "pair %u indexes past the sequence sets", i);
    PairDesc d{};
    d.a1_off = s1.offset[i1]; d.a2_off = s2.offset[i2];
    d.m = s1.length[i1]; d.n = s2.length[i2];
    d.a1_stride = d.m; d.a2_stride = d.n; d.out = i;
    hd[i] = d;
    if (mem == TRACYHIP_MEM_HOST) {  // host buffers only need to cover the strings themselves
      if (ops_len[i] > d.m + d.n) return set_error(TRACYHIP_ERR_ARG, "ops_len[%u] exceeds m+n", i);
      total = std::max<uint64_t>(total, ops_offset[i] + ops_len[i]);
    }
  }
  HIP_TRY(ctx->d_desc.ensure(sizeof(PairDesc) * (size_t)np));
  HIP_TRY(hipMemcpyAsync(ctx->d_desc.p, hd, sizeof(PairDesc) * (size_t)np, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx->h_off.ensure(sizeof(uint64_t) * (size_t)np));
  std::memcpy(ctx->h_off.p, ops_offset, sizeof(uint64_t) * (size_t)np);
  HIP_TRY(ctx->d_ops_off.ensure(sizeof(uint64_t) * (size_t)np));
  HIP_TRY(hipMemcpyAsync(ctx->d_ops_off.p, ctx->h_off.p, sizeof(uint64_t) * (size_t)np, hipMemcpyHostToDevice, st));
  RowsArgs ra{};
  ra.pairs = static_cast<const PairDesc*>(ctx->d_desc.p);
  ra.a1 = d_a1; ra.a2 = d_a2;
  ra.a1_profile = p1; ra.a2_profile = p2;
  ra.a2_onehot = (p1 && !p2);  // gotoh(profile, _createProfile(string)): row 1 shows consensus chars of the one-hot profile
  ra.ops_off = static_cast<const uint64_t*>(ctx->d_ops_off.p);
  ra.npairs = np;
  if (mem == TRACYHIP_MEM_HOST) {
    HIP_TRY(ctx->d_ops.ensure(total ? total : 1));
    HIP_TRY(ctx->d_ops_len.ensure(sizeof(uint32_t) * (size_t)np));
    HIP_TRY(ctx->d_rows0.ensure(total ? total : 1));
    HIP_TRY(ctx->d_rows1.ensure(total ? total : 1));
    if (total) HIP_TRY(hipMemcpyAsync(ctx->d_ops.p, ops, total, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(ctx->d_ops_len.p, ops_len, sizeof(uint32_t) * (size_t)np, hipMemcpyHostToDevice, st));
    ra.ops = static_cast<const uint8_t*>(ctx->d_ops.p);
    ra.ops_len = static_cast<const uint32_t*>(ctx->d_ops_len.p);
    ra.rows0 = static_cast<uint8_t*>(ctx->d_rows0.p);
    ra.rows1 = static_cast<uint8_t*>(ctx->d_rows1.p);
  } else {
    ra.ops = ops; ra.ops_len = ops_len; ra.rows0 = rows0; ra.rows1 = rows1;
  }
  HIP_TRY(launch_alignment_rows(ra, st));
  if (mem == TRACYHIP_MEM_HOST && total) {
    HIP_TRY(hipMemcpyAsync(rows0, ra.rows0, total, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(rows1, ra.rows1, total, hipMemcpyDeviceToHost, st));
  }
  HIP_TRY(ctx_sync(ctx));
  return TRACYHIP_OK;
}

}  // extern "C"
