// capi_internal.h -- context and helpers shared by the C-ABI translation units.
#ifndef TRACY_AMD_CAPI_INTERNAL_H
#define TRACY_AMD_CAPI_INTERNAL_H

#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/tracy_hip.h"
#include "dp_kernels.h"
#include "band16_launch.h"
#include "front.h"

namespace tracyhip {

struct DevBuf {  // grow-only device buffer
  void* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes);
  void release();
};
struct PinBuf {  // grow-only pinned host buffer (descriptor uploads)
  void* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes);
  void release();
};

int set_error(int code, const char* fmt, ...);

// Every switch of the library in one place.  Read from the environment ONCE, when a context is created (TRACYHIP_<NAME IN CAPITALS>),
// changed afterwards only through tracyhip_set_option(ctx, "<name>", "<value>"); tracyhip_describe() prints them.  Every one selects
// another exact path (A/B measurements, tests of the fallback tiers); none changes a result.  Lanes inherit their context's.
struct CtxKnobs {
  bool no_stream = false;         // pipelines planned by the host between launches (the pre-round-4 form) instead of stream-ordered
  bool no_narrow = false;         // int32 score kernels instead of the 16-bit sweeps
  bool no_compact = false;        // every 16-bit sweep on the six-code table
  bool no_screen = false;         // profile x profile scores by the full float chain only
  bool no_band = false;           // no checkpointed score pass / band traceback: whole-matrix tracebacks
  bool no_band16 = false;         // no band kernels (band16.h)
  bool no_front = false;          // no pruned sweeps (front.h)
  bool no_prefix = false;         // no prefix bounds (strand by certificate, pruned sweeps)
  bool no_vote = false;           // no k-mer orientation vote
  bool no_origin = false;         // allele-vs-window alignments by full traceback instead of the origin-tracking sweep
  bool no_subwindow = false;      // origin sweeps over the whole window
  bool no_prelim_origin = false;  // preliminary alignment of `tracy align` by band traceback instead of its two ends
  bool no_cq = false;             // string x string by byte compare instead of the query-profile table
  bool no_fused_walk = false;     // a separate walk launch after a traceback sweep
  bool no_quads = false;          // stream-ordered pipelines: narrow bands on sixteen lanes per pair like the rest (band16.h b16_narrow_ok)
  bool no_fork = false;           // stream-ordered pipelines: the launches of a band stage in a row on the call's stream instead of side by side
  bool sweeps_alone = false;      // MEASUREMENT mode of the stream-ordered orientation stage: the voted strand's chain finishes BEFORE the other strand's full sweeps
                                  // start and its prefix cells are credited to the front timer -- TRACYHIP_TIMER_SCORE then times the full sweeps on a device
                                  // of their own (bench.py roofline.dominant_kernel_alone_frac); slower, same results
  bool no_origin_band = false;    // `tracy decompose`, gotoh(allele, slice): the band d1 +- (g + 1) of every co-optimal path instead of the g + 3 diagonals of the walked one
  bool no_front_lists = false;    // pruned sweeps: later tiers skip what an earlier one certified in place instead of running over a list of the rest
  bool no_af_split = false;       // allelicFraction by the one-launch kernel (tp / cls in LDS, every grid point screened) instead of prepare + search
  bool no_decomp_wave = false;    // decomposeAlleles by the step-wise kernel only (decompose_kernels.h) instead of the one-wave body with its working set in LDS (decompose_wave.h)
  bool no_cont16 = false;         // the band below a kept prefix row (front.h) on the tagged int32 recurrence instead of the 16-bit cells
  bool verbose = false;           // one line per pipeline stage on stderr: how many pairs took which tier (TRACYHIP_HOST_TIMERS sets it too)
  int32_t band_w = -1;            // half width of the certified band of the final alignments: -1 = from the preliminary alignment (default),
                                  // 0 = whole matrices, else [1, 4096]
  uint32_t ckpt_b = 256;
  uint32_t front_list_min = 1024; // stream-ordered pipelines: units from which the later tiers of a pruned sweep (and the allele prefixes) run over device-side lists
  uint32_t quad_tier_min = 32768;  // stream-ordered pipelines: units (traces, or alleles) from which the pruned sweeps get their narrow quad tier          // steps between wavefront checkpoints [32, 1024]
};
void knobs_from_env(CtxKnobs& k);
// name without the TRACYHIP_ prefix, any case; false: no such option / bad value
bool knobs_set(CtxKnobs& k, const char* name, const char* value);
std::string knobs_describe(const CtxKnobs& k);

// Reference codes (MODE_QP a2) live in ctx->d_codes with tracyhip::kCodePad spare bytes on both sides: the sweep kernels
// prefetch the column two steps ahead of every lane without clamping it, so idle lanes read up to 64 + 3 bytes
// before the first / behind the last base of a sequence (any byte value is a valid table row selector).
constexpr size_t kCodePad = 128;
int choose_k(uint32_t m, int mode, bool needle = false);
uint64_t seqset_extent(const tracyhip_seqset& s);

// a validated batch: descriptors in caller order + device pointers of the payloads
struct DpProblem {
  int mode = MODE_CHAR;
  bool a1_profile = false, a2_profile = false;
  const void* d_a1 = nullptr;
  const void* d_a2 = nullptr;        // MODE_QP: encoded codes
  const void* d_a2_chars = nullptr;  // the raw a2 payload on the device
  const uint8_t* d_special = nullptr;  // MODE_CQ 16-bit sweeps: block map of the a2 codes (DpArgs::special_blocks), or null
  int cq_codes = 6;                  // MODE_CQ: what the a2 columns of the batch hold -- 4: A C G T only, 5: with N, 6: anything (origin sweeps size their table by it)
  const uint8_t* d_colclass = nullptr;  // profile x profile: column classes of the a2 set (build_problem), or null
  std::vector<PairDesc> desc;
  std::vector<int> k;
};

}  // namespace tracyhip

struct tracyhip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t own_stream = nullptr;
  uint64_t ws_limit = 0;
  tracyhip::DevBuf d_desc, d_bits, d_scratch, d_in1, d_in2, d_codes, d_scores, d_ops, d_ops_off, d_ops_len, d_err,
      d_rows0, d_rows1;
  tracyhip::DevBuf d_special;
  tracyhip::DevBuf d_ends;  // tracy align: ends of the preliminary alignments + scratch of that stage (orient_and_align)
  std::vector<tracyhip::PairDesc> cache_desc;  // descriptor / strip-height vectors of the generic DP entry points, kept between
  std::vector<int> cache_k;                    // calls (an all-pairs list is 36 MB: allocating it afresh costs 6 ms of page faults)
  std::vector<tracyhip::PairDesc> cache_full, cache_pre, cache_b16;  // the same for the orientation stage's sweep / prefix lists and the band jobs
  std::vector<tracyhip::FrontDesc> cache_fd;
  std::vector<int> cache_fullk, cache_b16k;
  tracyhip::DevBuf d_tmp[8];
  tracyhip::DevBuf d_pipe[64];
  tracyhip::DevBuf d_ckpt, d_lastrow, d_band;  // pipeline intermediates (align_traces / decompose)
  tracyhip::DevBuf d_b16tab[4], d_b16desc;     // substitution tables of the band kernels (band16.h), their descriptors
  tracyhip::DevBuf d_front;                    // descriptors / pairs / results of the pruned orientation sweep (front.h)
  tracyhip::DevBuf d_pre;                      // descriptors of its prefix launch over string x code pairs (run_prefix_keep_cq)
  tracyhip::B16Fork b16_fork;                  // side streams of the band stages (stream.hip band_stage), created with the context
  bool b16_fork_ok = false;
  tracyhip::DevBuf d_stream;                   // everything the stream-ordered pipelines keep on the device between their stages (stream.hip)
  hipError_t ensure_codes(size_t bytes, hipStream_t st) {
    hipError_t e = d_codes.ensure(bytes + 2 * tracyhip::kCodePad);
    if (e != hipSuccess) return e;
    // pads hold code 0 ('A'): an ordinary column for every kernel
    if ((e = hipMemsetAsync(d_codes.p, 0, tracyhip::kCodePad, st)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(static_cast<uint8_t*>(d_codes.p) + tracyhip::kCodePad + bytes, 0, tracyhip::kCodePad, st)) != hipSuccess) return e;
    // one byte per 256 code bytes: set by the encoders where a block holds an N or '-' / other code (DpArgs::special_blocks)
    if ((e = d_special.ensure((bytes >> 8) + 2)) != hipSuccess) return e;
    return hipMemsetAsync(d_special.p, 0, (bytes >> 8) + 2, st);
  }
  uint8_t* special_blocks() const { return static_cast<uint8_t*>(d_special.p); }
  uint8_t* codes() const { return static_cast<uint8_t*>(d_codes.p) + tracyhip::kCodePad; }
  tracyhip::DevBuf d_aftab;                    // allelicFraction grid enumeration (trace independent)
  bool aftab_ready = false;
  uint64_t ws_cache_budget = 0, ws_cache_held = 0;  // stream.hip workspace_budget: the last answer and what the context held then
  uint32_t ws_cache_share = 0;
  tracyhip::DevBuf d_afscratch;                // af_prepare_kernel -> af_search_kernel: tp, class bytes, headers
  tracyhip::DevBuf d_declut, d_dectodo;        // decompose_wave.h: the (primary, secondary) class table; per trace "left to decompose_kernel"
  bool declut_ready = false;
  tracyhip::PinBuf h_desc, h_off, h_tmp, h_res, h_b16desc[4], h_pre;
  uint32_t b16_round = 0;
  // kernel timing
  struct Pending { int which; hipEvent_t e0, e1; uint64_t cells, bytes; };
  // lanes: further contexts (own stream, own buffers) the batch pipelines split a call over, one host thread each
  // (tracyhip_set_lanes); this context is the first lane, empty = the pipelines run on it alone
  std::vector<tracyhip_ctx*> lanes;
  uint32_t mem_share = 1;  // contexts planning workspace on this device at the same time (lanes of one call): each takes its share of what is free
  bool timing = false;
  tracyhip::CtxKnobs knobs;  // (the fields below it used to be: no_narrow, no_compact, no_screen)
  tracyhip_call_stats stats = {};  // tiers of the last pipeline call (tracyhip_last_call_stats)
  std::vector<Pending> pending;
  std::vector<hipEvent_t> free_events;
  tracyhip_kernel_timing acc[TRACYHIP_TIMER_COUNT] = {};
  // asynchronous calls (tracyhip_*_async): executed in issue order by one worker thread per context
  struct AsyncState {
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::function<int()>> q;
    std::thread worker;
    std::thread::id worker_id;
    bool busy = false, stop = false;
    int rc = TRACYHIP_OK;      // first error since the last synchronize
    std::string msg;
  };
  AsyncState* async = nullptr;
  void release_all() {
    tracyhip::DevBuf* all[] = {&d_desc, &d_bits, &d_scratch, &d_in1, &d_in2, &d_codes, &d_scores, &d_ops,
                               &d_ops_off, &d_ops_len, &d_err, &d_rows0, &d_rows1};
    for (auto* b : all) b->release();
    for (auto& b : d_tmp) b.release();
    for (auto& b : d_pipe) b.release();
    d_ckpt.release(); d_lastrow.release(); d_band.release(); d_special.release(); d_ends.release();
    d_aftab.release(); aftab_ready = false;
    d_declut.release(); d_dectodo.release(); declut_ready = false;
    d_afscratch.release();
    for (auto& b : d_b16tab) b.release();
    d_b16desc.release();
    d_front.release();
    d_pre.release();
    d_stream.release();
    h_desc.release();
    h_off.release();
    h_tmp.release();
    h_res.release();
    h_pre.release();
    for (auto& b : h_b16desc) b.release();
  }
};

namespace tracyhip {
// The per-trace host loops of the pipelines (descriptors, bands, verdicts of 10^5 traces between two launches) on a few threads:
// fn(lo, hi, tid) over [0, n) in contiguous slices; small n runs inline.
constexpr uint32_t kHostThreads = 8;
// Workers that stay alive between calls (starting seven threads costs ~0.3 ms, and a decompose call of 10^5 traces runs forty of
// these loops between its launches).  One job at a time: a second caller (another lane) gets `false` and starts its own threads.
bool host_pool_run(const std::function<void(uint32_t)>& job);  // job(slice) for slice = 0 .. kHostThreads - 1, worked off by the pool + the caller
uint32_t host_pool_threads();  // threads a job runs on: TRACYHIP_HOST_THREADS, or the cores the process may use, at most kHostThreads
template <class Fn>
void parallel_for(uint32_t n, Fn fn) {
  if (n < 16384u) { fn(0u, n, 0u); return; }
  auto lo = [&](uint32_t t) { return (uint32_t)((uint64_t)n * t / kHostThreads); };
  if (host_pool_run([&](uint32_t t) { fn(lo(t), lo(t + 1), t); })) return;
  // the pool is busy with another lane's loop: threads of our own, no more than the process's share
  std::atomic<uint32_t> next(0);
  auto work = [&]() { for (uint32_t t = next.fetch_add(1); t < kHostThreads; t = next.fetch_add(1)) fn(lo(t), lo(t + 1), t); };
  std::vector<std::thread> th;
  const uint32_t nth = host_pool_threads();
  th.reserve(nth);
  for (uint32_t t = 1; t < nth; ++t) th.emplace_back(work);
  work();
  for (auto& x : th) x.join();
}
// A DpProblem borrows the context's descriptor vectors for its lifetime and hands them back, whatever the way out: batches of
// 10^5 pairs are megabytes of descriptors per stage, and allocating them afresh costs a millisecond of page faults each time.
// (One lease at a time per context: the stages of a pipeline build their problems one after the other.)
struct DpProblemLease {
  tracyhip_ctx* c;
  DpProblem& p;
  DpProblemLease(tracyhip_ctx* c_, DpProblem& p_) : c(c_), p(p_) {
    p.desc.swap(c->cache_desc); p.k.swap(c->cache_k);
    p.desc.clear(); p.k.clear();
  }
  ~DpProblemLease() { p.desc.swap(c->cache_desc); p.k.swap(c->cache_k); }
  DpProblemLease(const DpProblemLease&) = delete;
  DpProblemLease& operator=(const DpProblemLease&) = delete;
};
// the same for a band job (Band16Job, below): entries whose strip height is 0 are never read, so stale descriptors may stay in them
template <class Job>
struct Band16Lease {
  tracyhip_ctx* c;
  Job& j;
  Band16Lease(tracyhip_ctx* c_, Job& j_) : c(c_), j(j_) { j.desc.swap(c->cache_b16); j.k.swap(c->cache_b16k); }
  ~Band16Lease() { j.desc.swap(c->cache_b16); j.k.swap(c->cache_b16k); }
  Band16Lease(const Band16Lease&) = delete;
  Band16Lease& operator=(const Band16Lease&) = delete;
};
// Host-side profile of the pipelines (TRACYHIP_HOST_TIMERS=1): wall time of the labelled scopes, summed per label and printed to
// stderr when the process ends.  For finding where the host keeps the GPU waiting between two launches; off = one branch.
struct HostScope {
  const char* label;
  uint64_t t0;
  explicit HostScope(const char* l);
  ~HostScope();
};
#define TRACYHIP_HOST_SCOPE(name, label) tracyhip::HostScope name(label)
int timing_begin(tracyhip_ctx* ctx, int which, uint64_t cells, uint64_t bytes);  // record start event
int timing_end(tracyhip_ctx* ctx);                                              // record stop event
int timing_collect(tracyhip_ctx* ctx);                                          // after a stream sync
int ctx_begin(tracyhip_ctx* ctx);
// hipStreamSynchronize(ctx->stream), counted in tracyhip_call_stats::host_syncs
inline hipError_t ctx_sync(tracyhip_ctx* ctx) { ctx->stats.host_syncs += 1; return hipStreamSynchronize(ctx->stream); }
int async_submit(tracyhip_ctx* ctx, std::function<int()> fn);  // queue a call on the context's worker thread
int async_drain(tracyhip_ctx* ctx);                           // wait for the queue; returns (and clears) the first error
int stage_in(tracyhip_ctx* ctx, DevBuf& buf, const void* src, uint64_t bytes, int mem, const void** dev);
int check_params(const tracyhip_params* prm, uint64_t max_mn);
// stage: DP_PLAIN = score-only or full-matrix traceback; DP_CKPT = score-only pass that also writes wavefront
// checkpoints + last-row values (PairDesc::ckpt_off / lastrow_off set by the caller); DP_BAND = band traceback
// from those checkpoints (trace must be true); DP_PREFIX = prefix bound of the semiglobal score (rows 1 .. kPrefixLanes*K of
// profile x code pairs, 16-bit domain): d_scores receives max_j max(H, F) of that row.
// DP_ORIGIN = origin-tracking sweep of string x string pairs (DpCkpt::d_ends receives {lead, end} per pair, d_scores H(m,n)).
enum { DP_PLAIN = 0, DP_CKPT = 1, DP_BAND = 2, DP_PREFIX = 3, DP_ORIGIN = 4 };
struct DpCkpt {
  int32_t* d_ckpt = nullptr;
  int32_t* d_lastrow = nullptr;
  uint32_t B = 256;
  bool narrow = false;  // set by the DP_CKPT stage (16-bit kernel used), read by the DP_BAND stage
  uint32_t* d_ends = nullptr;  // DP_ORIGIN: two entries per pair, indexed by PairDesc::out
  const uint32_t* d_votes = nullptr;  // DP_CKPT of both orientations: orientation votes (DpArgs::votes), or null
  uint32_t vote_nt = 0;
};
// origin-tracking sweep: one pass of strip height K, columns and scores inside the packed fields (dp_lane.h origin_step)
bool origin_ok(const tracyhip_params* prm, uint32_t maxm, uint32_t maxn, int K);
int run_ckpt_prefix(tracyhip_ctx* ctx, const void* d_a1, const void* d_a2, const std::vector<PairDesc>& full, const std::vector<int>& fullk,
                    const std::vector<PairDesc>& pre, const tracyhip_params* prm, int32_t* d_scores, DpCkpt* ck, bool front_shape = false);
int run_dp(tracyhip_ctx* ctx, const DpProblem& pb, const tracyhip_params* prm, bool needle, bool trace,
           int32_t* d_scores, uint8_t* d_ops, const uint64_t* d_ops_off, uint32_t* d_ops_len, int stage = DP_PLAIN,
           DpCkpt* ck = nullptr);
bool narrow_ok(const tracyhip_params* prm, uint32_t maxm, int K, int64_t Q = 0);
int32_t sub_limit(const tracyhip_params* prm);
// largest |match| / |mismatch| whose table entries x 32 (tagged tracebacks, band kernels) fit int16; wider scorings take slower forms
constexpr int32_t kWideScore = 1000;
// device error block (DpArgs::err): kErrWords words owned by the DP launches + one verdict word of the pipelines' reference check
constexpr int kErrVerdictWord = kErrWords;
constexpr int kErrSweptWord = kErrWords + 2;  // 64-bit counter of the band traceback (DpArgs::swept), 8-byte aligned
constexpr size_t kErrBytes = sizeof(int32_t) * (kErrWords + 4);
// internal status (never returned through the C ABI): a 16-bit launch ran outside its proven value range; repeat on int32
constexpr int kWiden = 1;
int range_verdict(const tracyhip_params* prm, const int32_t* herr, const std::vector<std::pair<uint32_t, int>>& narrow_launches,
                  uint64_t max_mn, int value_shift);
int build_problem(tracyhip_ctx* ctx, const tracyhip_pairs* pairs, int mem, bool needle, DpProblem& pb, uint64_t* max_mn);

// ---- band kernels (band16.h): Gotoh on a diagonal band, four pairs per wave ----
// a batch for them: descriptors whose a1_off / a1_stride point into the substitution tables d_qp (build_b16_tables), a2_off into
// d_codes (codes 0..5), ckpt_off = band_pack(dmin, dmax); k[i] = strip height of pair i (band16_pick_k)
struct Band16Job {
  int kind = 0;                 // 0: traceback words + walk (scores, ops); 1: origin-tracking sweep (scores, ends)
  const int16_t* d_qp = nullptr;
  const uint8_t* d_codes = nullptr;
  std::vector<PairDesc> desc;
  std::vector<int> k;           // 4 / 8 / 12; 0: the pair is not part of the job (callers fill both vectors for every trace, in order)
};
// smallest strip height whose lanes are done with a strip before the next one is due (K + width <= 15 (K + 1)); 0: the band is too wide
int band16_pick_k(int32_t dmin, int32_t dmax);
// value ranges of the origin-tracking sweep (packed 14-bit score field, 13-bit origin) for m rows / n columns
bool origin16_ok(const tracyhip_params* prm, uint32_t maxm, uint32_t maxn);
// substitution tables of `desc.size()` sequences (b16_table_kernel): entries are scores << kTagShift; desc[i].out_off / stride are
// filled in here, the buffer is (re)sized.  strings: a1 holds bytes, else float profiles.
int build_b16_tables(tracyhip_ctx* ctx, DevBuf& buf, const void* d_a1, bool strings, std::vector<B16TableDesc>& desc, const tracyhip_params* prm);
int run_band16(tracyhip_ctx* ctx, Band16Job& job, const tracyhip_params* prm, int32_t* d_scores, uint32_t* d_ends, uint8_t* d_ops,
               const uint64_t* d_ops_off, uint32_t* d_ops_len);
// The pruned orientation sweep (front.h) of fd.size() pairs whose prefix rows are in d_row (PAIR_KEEP_ROW): band placed, band swept
// below the kept row (strip height K, band of 2 halfw + 1 diagonals), certificate.  FrontDesc::out must be the pair's index in fd.
// Host results per pair: fo (ok: the score is gotohScore), score, c_e as a window column (0: none).
struct FrontResult {
  std::vector<FrontOut> fo;
  std::vector<int32_t> score;
  std::vector<uint32_t> ce;
};
constexpr int kFrontK = 12;
constexpr int32_t kFrontHalfW = 90;  // 2 * 90 + 12 <= 15 * 13: the widest band one period of the K = 12 strips holds
// d_codes: the codes the band kernels read (null: the context's); keep_err: the error words hold the flags of a launch that has
// not been read yet (run_prefix_keep_cq) -- they are not cleared, and reported with this call's
int run_front(tracyhip_ctx* ctx, const std::vector<FrontDesc>& fd, const int16_t* d_qp, const uint32_t* d_row, const tracyhip_params* prm,
              FrontResult& out, const uint8_t* d_codes = nullptr, bool keep_err = false);
// The prefix rows of the pruned sweep for string x code pairs (gotoh(allele, window), indigo.h:359): rows 1 .. kFrontRows of every
// pair over all its columns, row kFrontRows kept at d_lastrow + PairDesc::lastrow_off (PAIR_KEEP_ROW).  Queued, not waited for: the
// error words are cleared before the launch and read by the run_front call that follows.
int run_prefix_keep_cq(tracyhip_ctx* ctx, const void* d_a1, const void* d_a2, const uint8_t* d_special, const std::vector<PairDesc>& pre,
                       const tracyhip_params* prm, int32_t* d_lastrow);
}  // namespace tracyhip
#endif
