// pipeline.hip -- tracyhip_align_traces: the hot section of `tracy align` (sage.h:191-311) for a batch
// of traces, all DP on the device.  Host work between stages is limited to reading back a few bytes
// per trace (orientation scores, trimmed-slice geometry) to plan the next launch.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "../../include/tracy_hip.h"
#include "capi_internal.h"
#include "launch.h"
#include "pipe_internal.h"
#include "pipe_kernels.h"

using namespace tracyhip;

#define HIP_TRY(expr)                                                                               \
  do {                                                                                              \
    hipError_t _e = (expr);                                                                         \
    if (_e != hipSuccess)                                                                           \
      return set_error(_e == hipErrorOutOfMemory ? TRACYHIP_ERR_OOM : TRACYHIP_ERR_HIP, "%s failed: %s (%s:%d)", \
                       #expr, hipGetErrorString(_e), __FILE__, __LINE__);                           \
  } while (0)

namespace {
void reset_call_stats(tracyhip_ctx* ctx, uint32_t ntraces) {
  ctx->stats = tracyhip_call_stats{};
  ctx->stats.traces = ntraces;
  for (auto* l : ctx->lanes) l->stats = tracyhip_call_stats{};
}
struct StageClock {  // TRACYHIP_HOST_TIMERS: wall time from one mark to the next, by label
  HostScope* cur = nullptr;
  void mark(const char* label) { delete cur; cur = new HostScope(label); }
  ~StageClock() { delete cur; }
};

template <class T>
int copy_out(tracyhip_ctx* ctx, int mem, T* user, const T* dev, size_t count) {
  if (!user || count == 0 || user == dev) return TRACYHIP_OK;
  HIP_TRY(hipMemcpyAsync(user, dev, sizeof(T) * count, mem == TRACYHIP_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, ctx->stream));
  return TRACYHIP_OK;
}

// =====================================================================================================
// Orientation + preliminary alignment of trimmed traces against their reference windows: the part `tracy align`
// (sage.h:223-258) and `tracy decompose` (indigo.h:235-302, FASTA / indexed reference) have in common.
// =====================================================================================================
struct OrientIn {
  uint32_t nt;
  const void* d_prof;       // device: profile payload (float)
  const uint64_t* a1_off;   // host [nt]: first column of the trimmed view (float index)
  const uint32_t* mf;       // host [nt]: row stride of the profile (full trace length)
  const uint32_t* mt;       // host [nt]: columns of the trimmed view
  const uint64_t* a2_off;   // host [nt]: reference window in the code buffer
  const uint32_t* rn;       // host [nt]: its length
  const uint8_t* oriented;  // host [nt] or null: orientation given by the caller (no scores, no decision)
  bool exact;               // both orientation scores exact (no strand by certificate)
  int32_t* d_verr;          // device: verdict word of the reference validation, or null (checked with the first read-back)
  uint8_t* d_ops;           // device outputs of the preliminary alignment (push order)
  const uint64_t* d_ops_off;
  uint32_t* d_ops_len;
  int32_t* d_score;         // device [nt] or null
  bool ends_only;           // the caller reads nothing of the preliminary alignment but trimReferenceSlice's two ends (`tracy align`):
                            // where it can be certified, the band traceback gives way to an origin-tracking sweep (OrientOut::d_ends)
  // substitution tables of the FULL profiles (build_b16_tables), or null: with them the preliminary alignment runs on the band kernels
  // (band16.h) -- the diagonals its score allows around the end of the alignment on row m -- as an origin-tracking sweep (ends_only)
  // or as a traceback whose string is completed with the free end-gap columns outside the sub-window
  const int16_t* d_qp = nullptr;
  const B16TableDesc* td = nullptr;  // host [nt]
  const uint32_t* row0 = nullptr;    // host [nt]: first row of the trimmed view in its table
};
struct OrientOut {
  std::vector<int32_t> sc2;     // [2 nt] forward / reverse scores (the loser's may be a certified upper bound)
  std::vector<uint8_t> fwd, rc; // rs.forward; "read the window as its reverse complement"
  const uint32_t* d_ends = nullptr;  // device [2 nt] when set: {leading 'h' columns, last column that is not a trailing 'h'} of the
                                     // preliminary alignment instead of its ops (OrientIn::ends_only)
  std::vector<uint32_t> gap;         // with d_ends: per trace, the most gap columns the preliminary alignment's score allows
                                     // ((top - S*) / |ge|): how far from a perfect match the trace is
};

constexpr int kNoEnds = 2;  // orient_and_align_impl: the ends path met a pair outside the origin-tracking sweep's range

// One run of the orientation stage + preliminary alignment over a batch (orient_and_align below repeats it on wider kernels when a
// launch reports values outside its proven range): what the stages share lives here, each stage is a method.
struct OrientRun {
  tracyhip_ctx* ctx;
  const tracyhip_params& p;
  const OrientIn& in;
  OrientOut& o;
  const bool force_wide, no_ends;
  hipStream_t st;
  const CtxKnobs& kn;
  const uint32_t nt;
  const void* d_prof;
  const uint32_t *mf, *mt, *rn;
  int32_t* d_verr;
  int32_t h_verr = 0;
  bool verr_fetched = false;
  int32_t* d_sc2 = nullptr;
  bool use_band = false, b16 = false, ends_path = false, tb16_path = false, given = false;
  bool use_prefix = false, use_front = false, use_vote = false;
  int norient = 2, K0 = 0;
  DpCkpt ck;
  std::vector<uint64_t> ck_off, lr_off;
  std::vector<int32_t>& h_sc2;
  std::vector<uint8_t>&h_fwd, &h_rc;  // rs.forward (decides how rs.pos moves in trimReferenceSlice); "read the window as its reverse complement"
  std::vector<uint8_t> elig;
  std::vector<int8_t> front_strand;  // the strand whose score and c_e the pruned sweep certified
  std::vector<uint32_t> front_ce;
  StageClock sco;

  OrientRun(tracyhip_ctx* c, const tracyhip_params& p_, const OrientIn& in_, OrientOut& o_, bool wide, bool noends)
      : ctx(c), p(p_), in(in_), o(o_), force_wide(wide), no_ends(noends), st(c->stream), kn(c->knobs), nt(in_.nt), d_prof(in_.d_prof), mf(in_.mf),
        mt(in_.mt), rn(in_.rn), d_verr(in_.d_verr), h_sc2(o_.sc2), h_fwd(o_.fwd), h_rc(o_.rc) {}

  PairDesc stage1_desc(uint32_t t, int orient) const {  // orient 0 = forward, 1 = reverse complement
    PairDesc d{};
    d.a1_off = in.a1_off[t];
    d.a1_stride = mf[t];
    d.m = mt[t];
    d.a2_off = in.a2_off[t];
    d.n = rn[t];
    d.a2_stride = rn[t];
    d.out = (uint32_t)orient * nt + t;
    d.flags = orient ? PAIR_A2_REVCOMP : 0;
    d.ckpt_off = ck_off[(size_t)orient * nt + t];
    d.lastrow_off = lr_off[(size_t)orient * nt + t];
    return d;
  }
  int run_stage1(std::vector<std::pair<uint32_t, int>> const& what, int stage) {
    DpProblem pb;
    DpProblemLease lease(ctx, pb);
    pb.mode = MODE_QP;
    pb.a1_profile = true;
    pb.d_a1 = d_prof;
    pb.d_a2 = ctx->codes();
    pb.desc.reserve(what.size());
    pb.k.reserve(what.size());
    for (auto const& w : what) {
      pb.desc.push_back(stage1_desc(w.first, w.second));
      pb.k.push_back(choose_k(mt[w.first], MODE_QP));
    }
    return run_dp(ctx, pb, &p, false, false, d_sc2, nullptr, nullptr, nullptr, stage, stage == DP_CKPT ? &ck : nullptr);
  }
  int fetch_scores() {
    HIP_TRY(hipMemcpyAsync(h_sc2.data(), d_sc2, sizeof(int32_t) * 2 * (size_t)nt, hipMemcpyDeviceToHost, st));
    if (d_verr && !verr_fetched) HIP_TRY(hipMemcpyAsync(&h_verr, d_verr, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx_sync(ctx));
    verr_fetched = true;
    if (h_verr & 4) return set_error(TRACYHIP_ERR_ARG, "reference windows must be upper-case [ACGTN] (loadSingleFasta, fasta.h:54-95)");
    return TRACYHIP_OK;
  }

  // which forms the batch takes, and their workspace
  int prepare() {
    o.d_ends = nullptr;
    // ---- 1. orientation scores: gotohScore(trim, fwd) / gotohScore(trim, rev)  (sage.h:239-240) ----
    // When every trimmed profile fits one pass of its strip height, the score pass also leaves wavefront
    // checkpoints and the last-row values, and stage 2 recomputes only the bands its path crosses
    // (band traceback) instead of storing the whole traceback matrix.
    HIP_TRY(ctx->d_tmp[0].ensure(sizeof(int32_t) * 2 * (size_t)nt));
    d_sc2 = static_cast<int32_t*>(ctx->d_tmp[0].p);
    use_band = !kn.no_band && p.ge < 0 && p.go <= 0 && sub_limit(&p) <= kWideScore;  // hfree = 1, vfree = 0 here
    ck.B = kn.ckpt_b;  // (developer knob, 256)
    // ends_path: the preliminary alignment is only trimmed from (OrientIn::ends_only).  The sweep's score S* and the end c_e of
    // the alignment on row m (row_m_end_kernel) bound where an optimal path can lie -- at most g = (Q m - S*) / |ge| horizontal
    // gap columns, so it starts no earlier than column c_e - m - g -- and an origin-tracking sweep over that sub-window (about a
    // tenth of a 10 kb window) delivers the two ends trimReferenceSlice reads: no wavefront checkpoints, no band traceback.
    // (The argument is the one of the allele alignments of `tracy decompose`)
    b16 = in.d_qp != nullptr && in.td != nullptr && in.row0 != nullptr && p.ge < 0 && p.go <= 0 && sub_limit(&p) <= kWideScore && !kn.no_band16;
    const bool cert_base = !no_ends && use_band && !force_wide && !kn.no_narrow && !kn.no_prelim_origin;
    ends_path = in.ends_only && cert_base;
    tb16_path = !in.ends_only && b16 && cert_base;  // traceback on the band kernels (the string is an output: `tracy decompose`)
    {
      uint32_t maxmt = 0;
      for (uint32_t t = 0; t < nt; ++t) maxmt = std::max(maxmt, mt[t]);
      ends_path = ends_path && nt && narrow_ok(&p, maxmt, 16);
      tb16_path = tb16_path && nt && narrow_ok(&p, maxmt, 16);
      for (uint32_t t = 0; t < nt && tb16_path; ++t) tb16_path = mt[t] && rn[t];
      for (uint32_t t = 0; t < nt && ends_path; ++t)
      {
        // the sub-window is at most m + g + 2 columns with g <= (Q m - S*) / |ge| and S* >= go + m ge (the all-gap path)
        const uint64_t cap = 2ull * mt[t] + ((uint64_t)sub_limit(&p) * mt[t] + (uint64_t)(-(int64_t)p.go)) / (uint64_t)(-(int64_t)p.ge) + 3;
        ends_path = mt[t] && rn[t] && origin_ok(&p, mt[t], (uint32_t)std::min<uint64_t>(rn[t], cap), choose_k(mt[t], MODE_QP));
      }
    }
    if (ends_path) ck.B = 0x7fffffffu;  // row m only (tb16_path keeps the wavefront checkpoints: pairs whose band is too wide for the band kernels
                                         // -- a heterozygous trace scores far below its row maxima -- take the band traceback from them)
    // in.oriented: the references are already oriented by the caller (k-mer seeding): one score pass, no decision
    given = in.oriented != nullptr;
    norient = given ? 1 : 2;
    ck_off.assign(2 * (size_t)nt, 0); lr_off.assign(2 * (size_t)nt, 0);
    {
      uint64_t ck_tot = 0, lr_tot = 0;
      for (uint32_t t = 0; t < nt && use_band; ++t) {
        const int K = choose_k(mt[t], MODE_QP);
        if (mt[t] == 0 || rn[t] == 0 || num_passes(mt[t], K) != 1) { use_band = false; break; }
        const uint32_t lanes_used = (mt[t] + K - 1) / K;
        const uint64_t J = ((uint64_t)rn[t] + lanes_used - 1) / ck.B;
        for (int o = 0; o < norient; ++o) {
          ck_off[(size_t)o * nt + t] = ck_tot;
          lr_off[(size_t)o * nt + t] = lr_tot;
          ck_tot += J * ckpt_fields(K) * 64;  // (the 16-bit query-profile sweep packs its records into K + 1 of these fields)
          lr_tot += 2ull * ((uint64_t)rn[t] + 1);
        }
      }
      if (use_band) {
        const uint64_t band_bytes = ends_path ? 0 : (uint64_t)nt * ck.B * 64 * 8;  // (no band traceback on the ends path)
        const uint64_t need = (ck_tot + lr_tot) * 4 + band_bytes;
        const bool have = ctx->d_ckpt.cap >= ck_tot * 4 + 64 && ctx->d_lastrow.cap >= lr_tot * 4 + 64 && ctx->d_band.cap >= band_bytes;
        size_t fr = 0, tot = 0;
        if (!have) HIP_TRY(hipMemGetInfo(&fr, &tot));  // (a driver call: skipped when the grow-only buffers already fit)
        if (!have && need > (uint64_t)(fr * 0.8 / ctx->mem_share) + ctx->d_ckpt.cap + ctx->d_lastrow.cap + ctx->d_band.cap) use_band = false;
        else {
          HIP_TRY(ctx->d_ckpt.ensure(ck_tot * 4 + 64));
          HIP_TRY(ctx->d_lastrow.ensure(lr_tot * 4 + 64));
          uint32_t maxmt = 0;
          for (uint32_t t = 0; t < nt; ++t) maxmt = std::max(maxmt, mt[t]);
          ck.narrow = !force_wide && !kn.no_narrow && narrow_ok(&p, maxmt, 16);  // conservative: the tallest strip
          ck.d_ckpt = static_cast<int32_t*>(ctx->d_ckpt.p);
          ck.d_lastrow = static_cast<int32_t*>(ctx->d_lastrow.p);
        }
      }
    }
    ends_path = ends_path && use_band && ck.narrow;
    tb16_path = tb16_path && use_band && ck.narrow;
    h_sc2.assign(2 * (size_t)nt, 0);
    h_fwd.assign(nt, 0);
    h_rc.assign(nt, 0);
    // Strand by certificate: a cheap prefix pass (rows 1 .. 8K of both orientations, eight pairs per wave) bounds each
    // orientation's score from above; the orientation with the larger bound is scored in full, and if the other one's bound
    // stays below that score the strand is decided without ever sweeping the loser over all rows (its score array then
    // holds the bound).  Traces whose bounds are close, or whose certificate fails, get both full passes: the decision is
    // always the reference's `gsFwd > gsRev`.
    use_prefix = !given && use_band && ck.narrow && !in.exact && !kn.no_prefix;
    // The pruned sweep of the voted strand (front.h): its prefix rows are swept over the whole window like the other strand's, the
    // rows below them only on a band around the best column of the prefix -- and the result is taken when its certificate holds.
    // `tracy align` reads the preliminary alignment by its two ends, `tracy decompose` takes its traceback from the band kernels (S*, c_e
    // are all they need); a pair of the latter whose band fails gets the full sweep of its strand after all (checkpoints for the band
    // traceback).  Exact results either way (TRACYHIP_NO_FRONT=1: off).
    use_front = (ends_path || tb16_path) && b16 && !given && use_band && ck.narrow && !kn.no_front && !kn.no_prefix && !kn.no_vote;
    // traces no taller than the prefix (8K rows) have nothing left to bound: they get both full passes
    elig.assign(nt, 0);
    if (use_prefix || use_front) {
      uint32_t ne = 0;
      for (uint32_t t = 0; t < nt; ++t) ne += elig[t] = mt[t] > (uint32_t)kPrefixLanes * choose_k(mt[t], MODE_QP);
      if (ne == 0) use_prefix = use_front = false;
    }
    // With one strip height for the whole batch the strand to sweep first is voted from shared k-mers before any DP runs
    // (kmer_vote_kernel), and the prefix bounds of the other strand ride in the same launch as the full sweeps, where their
    // short workgroups fill the tail.  Undecided votes get both full sweeps.  TRACYHIP_NO_VOTE=1: the two-stage form below.
    use_vote = use_prefix && !kn.no_vote;
    K0 = choose_k(mt[0], MODE_QP);
    for (uint32_t t = 0; t < nt && use_vote; ++t)
      if (choose_k(mt[t], MODE_QP) != K0) use_vote = false;  // (the pruned sweep takes any mix: its prefixes have one shape, its full sweeps
                                                             // one launch per strip height)
    front_strand.assign(nt, -1);
    front_ce.assign(nt, 0);
    return TRACYHIP_OK;
  }

  // the pruned sweep of the voted strand (front.h) + the other strand in full or by its prefix bound
  int orient_front() {
    int rc;
    sco.mark("o.a votes+rowmax descs/launch/readback");
    const uint32_t R = kFrontRows;  // every prefix of this branch has the 16 x 8 shape
    const size_t need = (sizeof(VoteDesc) + sizeof(RowMaxDesc) + 4 * sizeof(uint32_t)) * (size_t)nt;
    HIP_TRY(ctx->d_tmp[7].ensure(need));
    VoteDesc* d_vd = static_cast<VoteDesc*>(ctx->d_tmp[7].p);
    RowMaxDesc* d_rm = reinterpret_cast<RowMaxDesc*>(d_vd + nt);
    int32_t* d_ub = reinterpret_cast<int32_t*>(d_rm + nt);
    uint32_t* d_votes = reinterpret_cast<uint32_t*>(d_ub + nt);
    int32_t* d_ub1 = reinterpret_cast<int32_t*>(d_votes + 2 * (size_t)nt);  // rows clamped at -1 (front.h, second certificate)
    // both descriptor lists in one pinned block (laid out like the device block: one copy), the bounds and the votes back in one
    HIP_TRY(ctx->h_res.ensure(need));
    VoteDesc* hv = static_cast<VoteDesc*>(ctx->h_res.p);
    RowMaxDesc* hrm = reinterpret_cast<RowMaxDesc*>(hv + nt);
    parallel_for(nt, [&](uint32_t lo, uint32_t hi, uint32_t) {
      for (uint32_t t = lo; t < hi; ++t) {
        hv[t] = VoteDesc{in.a1_off[t], in.a2_off[t], mf[t], mt[t], rn[t], 0u};
        hrm[t] = RowMaxDesc{in.a1_off[t], mf[t], mt[t], R};
      }
    });
    HIP_TRY(hipMemcpyAsync(d_vd, hv, (sizeof(VoteDesc) + sizeof(RowMaxDesc)) * (size_t)nt, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(kmer_vote_kernel, dim3(nt), dim3(64), 0, st, d_vd, static_cast<const float*>(d_prof), ctx->codes(), d_votes);
    hipLaunchKernelGGL(rowmax_rest_kernel, dim3(nt), dim3(64), 0, st, d_rm, static_cast<const float*>(d_prof), (float)p.match, (float)p.mismatch, d_ub, d_ub1);
    HIP_TRY(hipGetLastError());
    const int32_t* h_ub = reinterpret_cast<const int32_t*>(hrm + nt);  // (the tail of the pinned block: [bounds nt][votes 2 nt][bounds, rows clamped at -1, nt])
    const uint32_t* h_votes = reinterpret_cast<const uint32_t*>(h_ub + nt);
    const int32_t* h_ub1 = reinterpret_cast<const int32_t*>(h_votes + 2 * (size_t)nt);
    HIP_TRY(hipMemcpyAsync(const_cast<int32_t*>(h_ub), d_ub, sizeof(uint32_t) * 4 * (size_t)nt, hipMemcpyDeviceToHost, st));
    if (d_verr && !verr_fetched) HIP_TRY(hipMemcpyAsync(&h_verr, d_verr, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx_sync(ctx));
    if (d_verr) {
      verr_fetched = true;
      if (h_verr & 4) return set_error(TRACYHIP_ERR_ARG, "reference windows must be upper-case [ACGTN] (loadSingleFasta, fasta.h:54-95)");
    }
    sco.mark("o.b build front descs");
    // the lists of the combined launch, laid out by a few threads in trace order (class per trace, a scan, the fill); the vectors
    // are the context's (megabytes per call: no fresh pages)
    std::vector<int8_t> guess(nt, 0), both(nt, 1), cls(nt, 0);  // cls: 0 = pruned sweep, 1 = both strands in full, 2 = voted strand in full + prefix of the other
    struct VecLease {
      tracyhip_ctx* c;
      std::vector<PairDesc> fullv, prev;
      std::vector<int> fullk;
      std::vector<FrontDesc> fd;
      explicit VecLease(tracyhip_ctx* c_) : c(c_) { fullv.swap(c->cache_full); prev.swap(c->cache_pre); fullk.swap(c->cache_fullk); fd.swap(c->cache_fd); }
      ~VecLease() { fullv.swap(c->cache_full); prev.swap(c->cache_pre); fullk.swap(c->cache_fullk); fd.swap(c->cache_fd); }
    } vl(ctx);
    std::vector<PairDesc>&fullv = vl.fullv, &prev = vl.prev;
    std::vector<int>& fullk = vl.fullk;
    std::vector<FrontDesc>& fd = vl.fd;
    std::vector<uint32_t> ft;
    struct Cnt { uint32_t full, pre, fr; };
    Cnt cnt[kHostThreads] = {};
    const bool exact = in.exact;
    parallel_for(nt, [&](uint32_t lo, uint32_t hi, uint32_t tid) {
      Cnt c{0, 0, 0};
      for (uint32_t t = lo; t < hi; ++t) {
        const uint32_t vf = h_votes[2 * t], vr = h_votes[2 * t + 1];
        guess[t] = vf >= vr ? 0 : 1;
        const uint32_t hi_v = vf >= vr ? vf : vr, lo_v = vf >= vr ? vr : vf;
        both[t] = (mt[t] > R && hi_v >= 32 && hi_v >= 2 * lo_v) ? 0 : 1;  // a clear majority of shared k-mers, or both sweeps
        const bool front = !both[t] && mt[t] - R > 2u * (uint32_t)kFrontK && rn[t] >= 1 &&
                           origin16_ok(&p, mt[t], mt[t] - R + 2u * (uint32_t)kFrontHalfW + 16u);
        if (front) { cls[t] = 0; c.fr += 1; c.pre += exact ? 1 : 2; c.full += exact ? 1 : 0; }
        else if (exact || both[t]) { cls[t] = 1; c.full += 2; }
        else { cls[t] = 2; c.full += 1; c.pre += 1; }
      }
      cnt[tid] = c;
    });
    Cnt at[kHostThreads + 1] = {};
    for (uint32_t i = 0; i < kHostThreads; ++i) at[i + 1] = Cnt{at[i].full + cnt[i].full, at[i].pre + cnt[i].pre, at[i].fr + cnt[i].fr};
    const Cnt tot = at[kHostThreads];
    fullv.resize(tot.full); fullk.resize(tot.full); prev.resize(tot.pre); fd.resize(tot.fr); ft.resize(tot.fr);
    parallel_for(nt, [&](uint32_t lo, uint32_t hi, uint32_t tid) {
      Cnt w = at[tid];  // (a small batch runs as one slice: tid 0, whose offsets are zero)
      for (uint32_t t = lo; t < hi; ++t) {
        const int g = (int)guess[t];
        const int Kt = choose_k(mt[t], MODE_QP);
        if (cls[t] == 0) {
          PairDesc d = stage1_desc(t, g);
          d.flags |= PAIR_KEEP_ROW;
          prev[w.pre++] = d;
          if (exact) { fullv[w.full] = stage1_desc(t, 1 - g); fullk[w.full++] = Kt; }
          else prev[w.pre++] = stage1_desc(t, 1 - g);
          FrontDesc f{};
          f.row_off = d.lastrow_off;
          f.a2_off = in.a2_off[t];
          f.tab_off = in.td[t].out_off + in.row0[t] + R;
          f.tab_stride = in.td[t].stride;
          f.m_rest = mt[t] - R;
          f.n = rn[t];
          f.flags = g ? PAIR_A2_REVCOMP : 0u;
          f.out = w.fr;
          f.R = R;
          f.rest = h_ub[t];
          f.tight = (p.ge <= -2 && h_ub1[t] <= h_ub[t]) ? (uint32_t)(h_ub[t] - h_ub1[t]) + 1u : 0u;
          fd[w.fr] = f;
          ft[w.fr++] = t;
        } else if (cls[t] == 1) {
          fullv[w.full] = stage1_desc(t, g); fullk[w.full++] = Kt;
          fullv[w.full] = stage1_desc(t, 1 - g); fullk[w.full++] = Kt;
        } else {
          fullv[w.full] = stage1_desc(t, g); fullk[w.full++] = Kt;
          prev[w.pre++] = stage1_desc(t, 1 - g);
        }
      }
    });
    sco.mark("o.c run_ckpt_prefix (plan+launch+wait)");
    DpCkpt ckv = ck;
    // (row m only, no wavefront checkpoints: the band kernels take the traceback from S*, c_e alone, and the strand swept in full is the
    // one the vote does NOT pick -- 7 GB of writes per 100 000 traces of `tracy decompose` that nothing read; a pair that needs the
    // band traceback from checkpoints after all is swept once more, below)
    ckv.B = 0x7fffffffu;
    if ((rc = run_ckpt_prefix(ctx, d_prof, ctx->codes(), fullv, fullk, prev, &p, d_sc2, &ckv, true))) return rc;
    sco.mark("o.d run_front");
    FrontResult fres;
    if ((rc = run_front(ctx, fd, in.d_qp, reinterpret_cast<const uint32_t*>(ck.d_lastrow), &p, fres))) return rc;
    sco.mark("o.e fetch+merge");
    if ((rc = fetch_scores())) return rc;
    // merge the scores of a repeat launch (which overwrites d_sc2 at the repeated entries only) into the host copy
    auto repeat_full = [&](std::vector<std::pair<uint32_t, int>> const& what) -> int {
      if (what.empty()) return TRACYHIP_OK;
      int rr;
      if ((rr = run_stage1(what, DP_CKPT))) return rr;
      std::vector<int32_t> got(2 * (size_t)nt);
      HIP_TRY(hipMemcpyAsync(got.data(), d_sc2, sizeof(int32_t) * 2 * (size_t)nt, hipMemcpyDeviceToHost, st));
      HIP_TRY(ctx_sync(ctx));
      for (auto const& r : what) h_sc2[(size_t)r.second * nt + r.first] = got[(size_t)r.second * nt + r.first];
      return TRACYHIP_OK;
    };
    std::vector<std::pair<uint32_t, int>> retry;
    for (size_t i = 0; i < ft.size(); ++i) {
      const uint32_t t = ft[i];
      const int g = (int)guess[t];
      if (fres.fo[i].ok && fres.ce[i]) {
        h_sc2[(size_t)g * nt + t] = fres.score[i];
        front_strand[t] = (int8_t)g;
        front_ce[t] = fres.ce[i];
      } else {
        retry.emplace_back(t, g);
      }
    }
    ctx->stats.pruned += (uint32_t)ft.size(); ctx->stats.pruned_uncertified += (uint32_t)retry.size();
    if (ctx->knobs.verbose) fprintf(stderr, "pruned orientation sweep: %zu of %u traces, %zu not certified\n", ft.size(), nt, retry.size());
    if ((rc = repeat_full(retry))) return rc;
    if (!in.exact) {  // the other strand of a clear vote holds its prefix maximum: decided by its bound, or swept in full
      retry.clear();
      for (uint32_t t = 0; t < nt; ++t) {
        if (both[t]) continue;
        const size_t w = (size_t)guess[t] * nt + t, l = (size_t)(1 - guess[t]) * nt + t;
        const int64_t bound_l = (int64_t)h_sc2[l] + h_ub[t];
        const bool certified = guess[t] == 0 ? bound_l < (int64_t)h_sc2[w] : bound_l <= (int64_t)h_sc2[w];
        if (certified) h_sc2[l] = (int32_t)std::min<int64_t>(bound_l, 0x7fffffff);
        else retry.emplace_back(t, 1 - guess[t]);
      }
      if ((rc = repeat_full(retry))) return rc;
    }
    return TRACYHIP_OK;
  }

  // the strand voted from shared k-mers swept in full, prefix bounds of the other strand in the same launch
  int orient_vote() {
    int rc;
    std::vector<VoteDesc> hv(nt);
    std::vector<RowMaxDesc> hrm(nt);
    for (uint32_t t = 0; t < nt; ++t) {
      hv[t] = VoteDesc{in.a1_off[t], in.a2_off[t], mf[t], mt[t], rn[t], 0u};
      hrm[t] = RowMaxDesc{in.a1_off[t], mf[t], mt[t], (uint32_t)kPrefixLanes * K0};
    }
    const size_t need = (sizeof(VoteDesc) + sizeof(RowMaxDesc) + 3 * sizeof(uint32_t)) * (size_t)nt;
    HIP_TRY(ctx->d_tmp[7].ensure(need));
    VoteDesc* d_vd = static_cast<VoteDesc*>(ctx->d_tmp[7].p);
    RowMaxDesc* d_rm = reinterpret_cast<RowMaxDesc*>(d_vd + nt);
    int32_t* d_ub = reinterpret_cast<int32_t*>(d_rm + nt);
    uint32_t* d_votes = reinterpret_cast<uint32_t*>(d_ub + nt);
    HIP_TRY(hipMemcpyAsync(d_vd, hv.data(), sizeof(VoteDesc) * (size_t)nt, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_rm, hrm.data(), sizeof(RowMaxDesc) * (size_t)nt, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(kmer_vote_kernel, dim3(nt), dim3(64), 0, st, d_vd, static_cast<const float*>(d_prof), ctx->codes(), d_votes);
    hipLaunchKernelGGL(rowmax_rest_kernel, dim3(nt), dim3(64), 0, st, d_rm, static_cast<const float*>(d_prof), (float)p.match, (float)p.mismatch, d_ub);
    HIP_TRY(hipGetLastError());
    std::vector<int32_t> h_ub(nt);
    std::vector<uint32_t> h_votes(2 * (size_t)nt);
    HIP_TRY(hipMemcpyAsync(h_ub.data(), d_ub, sizeof(int32_t) * (size_t)nt, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h_votes.data(), d_votes, sizeof(uint32_t) * 2 * (size_t)nt, hipMemcpyDeviceToHost, st));
    if (d_verr && !verr_fetched) HIP_TRY(hipMemcpyAsync(&h_verr, d_verr, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx_sync(ctx));
    if (d_verr) {
      verr_fetched = true;
      if (h_verr & 4) return set_error(TRACYHIP_ERR_ARG, "reference windows must be upper-case [ACGTN] (loadSingleFasta, fasta.h:54-95)");
    }
    std::vector<int8_t> guess(nt, 0), both(nt, 1);
    std::vector<PairDesc> fullv, prev;
    fullv.reserve(nt + nt / 8);
    prev.reserve(nt);
    for (uint32_t t = 0; t < nt; ++t) {
      const uint32_t vf = h_votes[2 * t], vr = h_votes[2 * t + 1];
      guess[t] = vf >= vr ? 0 : 1;
      const uint32_t hi = vf >= vr ? vf : vr, lo = vf >= vr ? vr : vf;
      both[t] = (elig[t] && hi >= 32 && hi >= 2 * lo) ? 0 : 1;  // a clear majority of shared k-mers, or both sweeps
      fullv.push_back(stage1_desc(t, (int)guess[t]));
      if (both[t]) fullv.push_back(stage1_desc(t, 1 - guess[t]));
      else prev.push_back(stage1_desc(t, 1 - guess[t]));
    }
    DpCkpt ckv = ck;
    if ((rc = run_ckpt_prefix(ctx, d_prof, ctx->codes(), fullv, std::vector<int>(fullv.size(), K0), prev, &p, d_sc2, &ckv, false))) return rc;
    if ((rc = fetch_scores())) return rc;
    std::vector<std::pair<uint32_t, int>> retry;
    for (uint32_t t = 0; t < nt; ++t) {
      if (both[t]) continue;
      const size_t w = (size_t)guess[t] * nt + t, l = (size_t)(1 - guess[t]) * nt + t;
      const int64_t bound_l = (int64_t)h_sc2[l] + h_ub[t];
      const bool certified = guess[t] == 0 ? bound_l < (int64_t)h_sc2[w] : bound_l <= (int64_t)h_sc2[w];
      if (certified) h_sc2[l] = (int32_t)std::min<int64_t>(bound_l, 0x7fffffff);
      else retry.emplace_back(t, 1 - guess[t]);
    }
    if (!retry.empty()) {
      std::vector<int32_t> keep = h_sc2;
      if ((rc = run_stage1(retry, DP_CKPT))) return rc;
      std::vector<int32_t> got(2 * (size_t)nt);
      HIP_TRY(hipMemcpyAsync(got.data(), d_sc2, sizeof(int32_t) * 2 * (size_t)nt, hipMemcpyDeviceToHost, st));
      HIP_TRY(ctx_sync(ctx));
      h_sc2 = keep;
      for (auto const& r : retry) h_sc2[(size_t)r.second * nt + r.first] = got[(size_t)r.second * nt + r.first];
    }
    return TRACYHIP_OK;
  }

  // the two-stage form: prefix bounds of both strands, then the likely winner (no_vote)
  int orient_prefix() {
    int rc;
    std::vector<std::pair<uint32_t, int>> all2;
    for (int o = 0; o < 2; ++o)
      for (uint32_t t = 0; t < nt; ++t)
        if (elig[t]) all2.emplace_back(t, o);
    if ((rc = run_stage1(all2, DP_PREFIX))) return rc;
    std::vector<RowMaxDesc> hrm(nt);
    for (uint32_t t = 0; t < nt; ++t)
      hrm[t] = RowMaxDesc{in.a1_off[t], mf[t], mt[t], (uint32_t)kPrefixLanes * choose_k(mt[t], MODE_QP)};
    HIP_TRY(ctx->d_tmp[7].ensure(sizeof(RowMaxDesc) * (size_t)nt + sizeof(int32_t) * (size_t)nt));
    RowMaxDesc* d_rm = static_cast<RowMaxDesc*>(ctx->d_tmp[7].p);
    int32_t* d_ub = reinterpret_cast<int32_t*>(d_rm + nt);
    HIP_TRY(hipMemcpyAsync(d_rm, hrm.data(), sizeof(RowMaxDesc) * (size_t)nt, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(rowmax_rest_kernel, dim3(nt), dim3(64), 0, st, d_rm, static_cast<const float*>(d_prof), (float)p.match, (float)p.mismatch, d_ub);
    HIP_TRY(hipGetLastError());
    std::vector<int32_t> h_ub(nt);
    HIP_TRY(hipMemcpyAsync(h_ub.data(), d_ub, sizeof(int32_t) * (size_t)nt, hipMemcpyDeviceToHost, st));
    if ((rc = fetch_scores())) return rc;
    std::vector<int64_t> bound(2 * (size_t)nt);
    for (uint32_t t = 0; t < nt; ++t)
      for (int o = 0; o < 2; ++o) bound[(size_t)o * nt + t] = (int64_t)h_sc2[(size_t)o * nt + t] + h_ub[t];
    // full passes: the likely winner of every trace; both orientations where the bounds are within 10 % of each other
    std::vector<std::pair<uint32_t, int>> full;
    std::vector<int8_t> guess(nt), both(nt, 0);
    for (uint32_t t = 0; t < nt; ++t) {
      const int64_t bf = bound[t], br = bound[nt + t];
      guess[t] = bf >= br ? 0 : 1;
      const int64_t bw = guess[t] ? br : bf, bl = guess[t] ? bf : br;
      both[t] = (!elig[t] || bw <= 0 || bl * 10 > bw * 9) ? 1 : 0;
      full.emplace_back(t, (int)guess[t]);
      if (both[t]) full.emplace_back(t, 1 - guess[t]);
    }
    if ((rc = run_stage1(full, DP_CKPT))) return rc;
    std::vector<int32_t> pref = h_sc2;
    if ((rc = fetch_scores())) return rc;
    std::vector<std::pair<uint32_t, int>> retry;
    for (uint32_t t = 0; t < nt; ++t) {
      if (both[t]) continue;
      const size_t w = (size_t)guess[t] * nt + t, l = (size_t)(1 - guess[t]) * nt + t;
      // guess forward: forward iff gsFwd > gsRev, certified by bound(rev) < gsFwd; guess reverse: certified by bound(fwd) <= gsRev
      const bool certified = guess[t] == 0 ? bound[l] < (int64_t)h_sc2[w] : bound[l] <= (int64_t)h_sc2[w];
      if (certified) h_sc2[l] = (int32_t)std::min<int64_t>(bound[l], 0x7fffffff);
      else retry.emplace_back(t, 1 - guess[t]);
    }
    if (!retry.empty()) {
      std::vector<int32_t> keep = h_sc2;
      // d_sc2 is overwritten only at the retried entries; merge them into the host copy
      if ((rc = run_stage1(retry, DP_CKPT))) return rc;
      std::vector<int32_t> got(2 * (size_t)nt);
      HIP_TRY(hipMemcpyAsync(got.data(), d_sc2, sizeof(int32_t) * 2 * (size_t)nt, hipMemcpyDeviceToHost, st));
      HIP_TRY(ctx_sync(ctx));
      h_sc2 = keep;
      for (auto const& r : retry) h_sc2[(size_t)r.second * nt + r.first] = got[(size_t)r.second * nt + r.first];
    }
    (void)pref;
    return TRACYHIP_OK;
  }

  // both orientations swept in full (exact gsFwd / gsRev), or the one the caller gave
  int orient_full() {
    int rc;
    std::vector<std::pair<uint32_t, int>> all;
    for (int o = 0; o < norient; ++o)
      for (uint32_t t = 0; t < nt; ++t) all.emplace_back(t, o);
    // Both orientations swept in full (exact gsFwd / gsRev).  Only the winner's checkpoints are ever read, so the sweeps
    // of the strand an orientation vote marks as the likely loser write none (vote_skips_checkpoints: the kernel reads the
    // votes itself, no host round trip); a likely loser that wins after all is swept once more, with checkpoints.
    // (Also on the ends path, where only row m is kept: the sweep of the likely loser is 2 % faster without its row-m stores,
    // which is more than the vote costs.)
    const bool vote_ckpt = !given && use_band && ck.narrow && !kn.no_vote;
    std::vector<uint32_t> h_votes;
    if (vote_ckpt) {
      std::vector<VoteDesc> hv(nt);
      for (uint32_t t = 0; t < nt; ++t) hv[t] = VoteDesc{in.a1_off[t], in.a2_off[t], mf[t], mt[t], rn[t], 0u};
      HIP_TRY(ctx->d_tmp[7].ensure((sizeof(VoteDesc) + 2 * sizeof(uint32_t)) * (size_t)nt));
      VoteDesc* d_vd = static_cast<VoteDesc*>(ctx->d_tmp[7].p);
      uint32_t* d_votes = reinterpret_cast<uint32_t*>(d_vd + nt);
      HIP_TRY(ctx->h_res.ensure(sizeof(VoteDesc) * (size_t)nt));  // pinned staging (free until the results go out): no host wait
      std::memcpy(ctx->h_res.p, hv.data(), sizeof(VoteDesc) * (size_t)nt);
      HIP_TRY(hipMemcpyAsync(d_vd, ctx->h_res.p, sizeof(VoteDesc) * (size_t)nt, hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(kmer_vote_kernel, dim3(nt), dim3(64), 0, st, d_vd, static_cast<const float*>(d_prof), ctx->codes(), d_votes);
      HIP_TRY(hipGetLastError());
      ck.d_votes = d_votes;
      ck.vote_nt = nt;
      h_votes.resize(2 * (size_t)nt);
    }
    if ((rc = run_stage1(all, use_band ? DP_CKPT : DP_PLAIN))) return rc;
    if (vote_ckpt) HIP_TRY(hipMemcpyAsync(h_votes.data(), ck.d_votes, sizeof(uint32_t) * 2 * (size_t)nt, hipMemcpyDeviceToHost, st));
    ck.d_votes = nullptr;
    if ((rc = fetch_scores())) return rc;
    if (vote_ckpt) {
      std::vector<std::pair<uint32_t, int>> retry;
      for (uint32_t t = 0; t < nt; ++t) {
        const int w = h_sc2[t] > h_sc2[nt + t] ? 0 : 1;  // forward iff gsFwd > gsRev (sage.h:247)
        if (vote_skips_checkpoints(h_votes[2 * t], h_votes[2 * t + 1], (uint32_t)w)) retry.emplace_back(t, w);
      }
      if (!retry.empty() && (rc = run_stage1(retry, DP_CKPT))) return rc;  // same scores, now with checkpoints
    }
    return TRACYHIP_OK;
  }

  // forward iff gsFwd > gsRev (sage.h:247)
  void decide() {
    sco.mark("o.f stage2 setup");
    for (uint32_t t = 0; t < nt; ++t) {
      if (given) { h_fwd[t] = in.oriented[t] ? 1 : 0; h_rc[t] = 0; }
      else { h_fwd[t] = h_sc2[t] > h_sc2[nt + t] ? 1 : 0; h_rc[t] = !h_fwd[t]; }  // forward iff gsFwd > gsRev (sage.h:247)
    }

  }

  // ---- 2. preliminary alignment gotoh(trim, oriented reference) (sage.h:258 / indigo.h:302) ----
  int prelim() {
    int rc;
    {
      DpProblem pb;
      DpProblemLease lease(ctx, pb);
      pb.mode = MODE_QP;
      pb.a1_profile = true;
      pb.d_a1 = d_prof;
      pb.d_a2 = ctx->codes();
      pb.desc.resize(nt);
      pb.k.resize(nt);
      parallel_for(nt, [&](uint32_t lo, uint32_t hi, uint32_t) {
        for (uint32_t t = lo; t < hi; ++t) {
          PairDesc d{};
          d.a1_off = in.a1_off[t];
          d.a1_stride = mf[t];
          d.m = mt[t];
          d.a2_off = in.a2_off[t];
          d.n = rn[t];
          d.a2_stride = rn[t];
          d.out = t;
          d.flags = h_rc[t] ? PAIR_A2_REVCOMP : 0;
          const size_t o = h_rc[t] ? (size_t)nt + t : t;  // the winning orientation's checkpoints
          d.ckpt_off = ck_off[o];
          d.lastrow_off = lr_off[o];
          pb.desc[t] = d;
          pb.k[t] = choose_k(d.m, MODE_QP);
        }
      });
      if (ends_path || tb16_path) {
        // c_e from the winner's row m, the sub-window from S* and c_e; over it the origin-tracking sweep delivers the two ends (ends_path)
        // or the band kernels the traceback (tb16_path)
        HIP_TRY(ctx->d_ends.ensure((sizeof(uint32_t) * 5 + sizeof(RowEndDesc) + sizeof(RowMaxDesc)) * (size_t)nt));
        uint32_t* d_ends = static_cast<uint32_t*>(ctx->d_ends.p);
        uint32_t* d_ce = d_ends + 2 * (size_t)nt;
        uint32_t* d_shift = d_ce + nt;
        int32_t* d_top = reinterpret_cast<int32_t*>(d_shift + nt);
        RowEndDesc* d_re = reinterpret_cast<RowEndDesc*>(d_top + nt);
        RowMaxDesc* d_rm = reinterpret_cast<RowMaxDesc*>(d_re + nt);
        // both descriptor lists in one pinned block laid out like the device block (one copy); c_e and top come back into its tail
        HIP_TRY(ctx->h_res.ensure((sizeof(RowEndDesc) + sizeof(RowMaxDesc) + 2 * sizeof(uint32_t)) * (size_t)nt));
        RowEndDesc* hre = static_cast<RowEndDesc*>(ctx->h_res.p);
        RowMaxDesc* hrm = reinterpret_cast<RowMaxDesc*>(hre + nt);
        auto from_front = [&](uint32_t t) { return front_strand[t] >= 0 && (front_strand[t] != 0) == (h_rc[t] != 0); };  // (else the winner was swept in full)
        parallel_for(nt, [&](uint32_t lo, uint32_t hi, uint32_t) {
          for (uint32_t t = lo; t < hi; ++t) {
            hre[t] = RowEndDesc{pb.desc[t].lastrow_off, from_front(t) ? 0u : rn[t], 0};
            hrm[t] = RowMaxDesc{in.a1_off[t], mf[t], mt[t], 0u};
          }
        });
        HIP_TRY(hipMemcpyAsync(d_re, hre, (sizeof(RowEndDesc) + sizeof(RowMaxDesc)) * (size_t)nt, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(row_m_end_kernel, dim3(nt), dim3(64), 0, st, static_cast<const RowEndDesc*>(d_re), static_cast<const int32_t*>(ck.d_lastrow),
                           p.go + p.ge, d_ce);
        // what the diagonal steps of ANY path can add up to at most: every row gives at most max(0, its best table entry)
        hipLaunchKernelGGL(rowmax_rest_kernel, dim3(nt), dim3(64), 0, st, static_cast<const RowMaxDesc*>(d_rm), static_cast<const float*>(d_prof),
                           (float)p.match, (float)p.mismatch, d_top);
        HIP_TRY(hipGetLastError());
        std::vector<uint32_t> shift(nt, 0);
        uint32_t* h_ce = reinterpret_cast<uint32_t*>(hrm + nt);
        const int32_t* h_top = reinterpret_cast<const int32_t*>(h_ce + nt);
        HIP_TRY(hipMemcpyAsync(h_ce, d_ce, sizeof(uint32_t) * (size_t)nt, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(const_cast<int32_t*>(h_top), d_top, sizeof(int32_t) * (size_t)nt, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx_sync(ctx));  // (also: hre, hrm have been read)
        for (uint32_t t = 0; t < nt; ++t)
          if (from_front(t)) h_ce[t] = front_ce[t];
        // A path from (0, lead) to (m, c_e) collects at most top = sum over the rows of max(0, best entry of the row's table
        // column) on its diagonal steps, nothing positive on its vertical ones (go <= 0, ge < 0), and loses at least |ge| per
        // horizontal gap column: S* <= top - |ge| g.  (top is computed from the profile as it is -- normalised or not -- and is
        // what best * m overestimates: a profile column that is not one-hot cannot score `match`.)
        // Going back from (m, c_e), a path with at most g gap steps stays on the diagonals c_e - m - g .. c_e - m + g: the band the
        // band kernels sweep (band16.h), where the band fits them; other pairs take the origin-tracking sweep over the whole
        // sub-window (ends_path) or the whole matrix (tb16_path).
        sco.mark("o.g stage2 band plan");
        const int64_t age = -(int64_t)p.ge;
        std::vector<int32_t> h_pre(nt);
        o.gap.assign(nt, 0);
        Band16Job j16;
        Band16Lease<Band16Job> j16_lease(ctx, j16);
        j16.kind = ends_path ? 1 : 0;
        j16.d_qp = in.d_qp;
        j16.d_codes = ctx->codes();
        j16.desc.resize(nt);
        j16.k.assign(nt, 0);
        DpProblem rest;
        rest.mode = pb.mode; rest.a1_profile = pb.a1_profile; rest.d_a1 = pb.d_a1; rest.d_a2 = pb.d_a2;
        std::vector<PairDesc> wholes(tb16_path ? nt : 0);
        parallel_for(nt, [&](uint32_t lo, uint32_t hi, uint32_t) {
          for (uint32_t t = lo; t < hi; ++t) {
            const PairDesc whole = pb.desc[t];
            if (tb16_path) wholes[t] = whole;
            PairDesc& d = pb.desc[t];
            h_pre[t] = h_rc[t] ? h_sc2[nt + t] : h_sc2[t];
            const int64_t ce = h_ce[t];
            if (ce <= 0) {
              // H(m, c) == E(m, c) in every column: the reference's traceback (gotoh.h:143-167) runs along row m to column 0 and
              // up column 0 -- n 'h', then m 'v' -- so both ends are 0 (a junk trace: the all-gap path is optimal).  Column 1 alone
              // reproduces that: H(m, 1) == E(m, 1), opened from H(m, 0), whose origin is 0.
              if (ends_path) {
                d.a2_off += h_rc[t] ? (uint64_t)(d.n - 1u) : 0ull;
                d.n = 1;
                d.a2_stride = 1;
              }
              continue;
            }
            const int64_t loss = (int64_t)h_top[t] - (int64_t)h_pre[t];
            const int64_t g = loss > 0 ? loss / age : 0;
            o.gap[t] = (uint32_t)std::min<int64_t>(g, 0x7fffffff);
            int64_t a = ce - (int64_t)d.m - g - 2;
            if (a < 0) a = 0;
            shift[t] = (uint32_t)a;
            d.a2_off += h_rc[t] ? (uint64_t)(d.n - (uint32_t)ce) : (uint64_t)a;  // reverse view: column c is byte n - c
            d.n = (uint32_t)(ce - a);
            d.a2_stride = d.n;
            int K = 0;
            int32_t dlo = 0, dhi = 0;
            if (b16 && g < (1 << 20)) {
              const int32_t d1 = (int32_t)d.n - (int32_t)d.m;
              dlo = d1 - (int32_t)g - 1;
              dhi = d1 + (int32_t)g + 1;
              K = band16_pick_k(dlo, dhi);
              if (K && ends_path && !origin16_ok(&p, d.m, d.n)) K = 0;
              if (K && 4ull * ((d.n + 7u) & ~3u) + b16_table_bytes(K) > 60u * 1024u) K = 0;  // (the codes of four pairs are staged in LDS)
            }
            if (K) {
              PairDesc q = d;
              q.a1_off = in.td[t].out_off + in.row0[t];
              q.a1_stride = in.td[t].stride;
              q.ckpt_off = band_pack(dlo, dhi);
              q.lastrow_off = ends_path ? 0ull : ((uint64_t)(whole.n - (uint32_t)ce) | ((uint64_t)(uint32_t)a << 32));  // 'h' right / left of the sub-window
              j16.desc[t] = q;
              j16.k[t] = K;
            }
          }
        });
        for (uint32_t t = 0; t < nt; ++t)  // what the band kernels do not take: the origin-tracking sweep over the sub-window / the band traceback
          if (j16.k[t] == 0) { rest.desc.push_back(ends_path ? pb.desc[t] : wholes[t]); rest.k.push_back(pb.k[t]); }
        if (ends_path) {
          bool fits = true;  // (the pre-check used an upper bound of the sub-window; windows cut at c_e can only be shorter)
          for (size_t q = 0; q < rest.desc.size() && fits; ++q) fits = origin_ok(&p, rest.desc[q].m, rest.desc[q].n, rest.k[q]);
          if (!fits) return kNoEnds;  // (cannot happen while the pre-check's bound holds; the caller repeats the stage with the band traceback)
          HIP_TRY(hipMemcpyAsync(d_shift, shift.data(), sizeof(uint32_t) * (size_t)nt, hipMemcpyHostToDevice, st));
          DpCkpt oc;
          oc.d_ends = d_ends;
          if (rest.desc.size() < nt) {
            HIP_TRY(hipMemsetAsync(ctx->d_err.p, 0, sizeof(int32_t) * kErrWords, st));
            if ((rc = run_band16(ctx, j16, &p, nullptr, d_ends, nullptr, nullptr, nullptr))) return rc;
          }
          if ((rc = run_dp(ctx, rest, &p, false, false, nullptr, nullptr, nullptr, nullptr, DP_ORIGIN, &oc))) return rc;  // (kWiden: the caller restarts wide)
          hipLaunchKernelGGL(ends_shift_kernel, dim3((nt + 255) / 256), dim3(256), 0, st, d_ends, static_cast<const uint32_t*>(d_shift), nt);
          HIP_TRY(hipGetLastError());
          o.d_ends = d_ends;
        } else {
          sco.mark("o.h stage2 run_band16 + check");
          // traceback on the band; a pair whose banded score is not S* (or whose walk left the band: no ops) is repeated with the rest
          if (rest.desc.size() < nt) {
            HIP_TRY(ctx->d_tmp[7].ensure(sizeof(int32_t) * (size_t)nt));
            int32_t* d_sb = static_cast<int32_t*>(ctx->d_tmp[7].p);
            HIP_TRY(hipMemsetAsync(ctx->d_err.p, 0, sizeof(int32_t) * kErrWords, st));
            if ((rc = run_band16(ctx, j16, &p, d_sb, nullptr, in.d_ops, in.d_ops_off, in.d_ops_len))) return rc;
            std::vector<int32_t> h_sb(nt);
            std::vector<uint32_t> h_ol(nt);
            HIP_TRY(hipMemcpyAsync(h_sb.data(), d_sb, sizeof(int32_t) * (size_t)nt, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(h_ol.data(), in.d_ops_len, sizeof(uint32_t) * (size_t)nt, hipMemcpyDeviceToHost, st));
            HIP_TRY(ctx_sync(ctx));
            uint32_t nfail = 0;
            const size_t nb16 = nt - rest.desc.size();
            for (uint32_t t = 0; t < nt; ++t)
              if (j16.k[t] && (h_sb[t] != h_pre[t] || h_ol[t] == 0)) { rest.desc.push_back(wholes[t]); rest.k.push_back(pb.k[t]); ++nfail; }
            ctx->stats.prelim_banded += (uint32_t)nb16; ctx->stats.prelim_repeated += nfail;
            if (ctx->knobs.verbose) fprintf(stderr, "preliminary alignment: %zu of %u on the band, %u repeated\n", nb16, nt, nfail);
          }
          if (use_front) {  // the orientation stage of the pruned sweep leaves no wavefront checkpoints: sweep the pair's strand before its band traceback
            std::vector<std::pair<uint32_t, int>> resweep;
            for (auto const& d : rest.desc) resweep.emplace_back(d.out, h_rc[d.out] ? 1 : 0);
            if (!resweep.empty() && (rc = run_stage1(resweep, DP_CKPT))) return rc;
          }
          if ((rc = run_dp(ctx, rest, &p, false, true, nullptr, in.d_ops, in.d_ops_off, in.d_ops_len, DP_BAND, &ck))) return rc;
        }
        if (in.d_score) HIP_TRY(hipMemcpy(in.d_score, h_pre.data(), sizeof(int32_t) * (size_t)nt, hipMemcpyHostToDevice));
      } else if (use_band) {
        // the preliminary score equals the winning orientation score (same DP): no score array needed from the band pass
        if ((rc = run_dp(ctx, pb, &p, false, true, nullptr, in.d_ops,
                         in.d_ops_off, in.d_ops_len, DP_BAND, &ck)))
          return rc;
        std::vector<int32_t> h_pre(nt);
        for (uint32_t t = 0; t < nt; ++t) h_pre[t] = h_rc[t] ? h_sc2[nt + t] : h_sc2[t];
        if (in.d_score) HIP_TRY(hipMemcpy(in.d_score, h_pre.data(), sizeof(int32_t) * (size_t)nt, hipMemcpyHostToDevice));
      } else if ((rc = run_dp(ctx, pb, &p, false, true, in.d_score, in.d_ops,
                              in.d_ops_off, in.d_ops_len)))
        return rc;
    }

    return TRACYHIP_OK;
  }
};

int orient_and_align_impl(tracyhip_ctx* ctx, const tracyhip_params& p, const OrientIn& in, OrientOut& o, bool force_wide, bool no_ends = false) {
  OrientRun r(ctx, p, in, o, force_wide, no_ends);
  int rc = r.prepare();
  if (rc) return rc;
  // ---- 1. orientation scores: gotohScore(trim, fwd) / gotohScore(trim, rev)  (sage.h:239-240) ----
  if (r.use_front) rc = r.orient_front();
  else if (r.use_vote) rc = r.orient_vote();
  else if (r.use_prefix) rc = r.orient_prefix();
  else rc = r.orient_full();
  if (rc) return rc;
  r.decide();
  return r.prelim();
}

// The 16-bit sweeps assume substitution scores of normalised profiles (|q| <= max(|match|, |mismatch|)).  A launch that meets a
// larger entry reports it; when the 16-bit range no longer holds (kWiden) the whole stage is repeated on the int32 kernels.
int orient_and_align(tracyhip_ctx* ctx, const tracyhip_params& p, const OrientIn& in, OrientOut& o) {
  int rc = orient_and_align_impl(ctx, p, in, o, false);
  if (rc == kNoEnds) rc = orient_and_align_impl(ctx, p, in, o, false, true);
  if (rc == kWiden) rc = orient_and_align_impl(ctx, p, in, o, true);
  if (rc == kWiden) rc = set_error(TRACYHIP_ERR_RANGE, "profile values outside the range of the score kernels");
  return rc;
}

}  // namespace

// what every form of the call checks first; *empty: nothing to do
static int align_check_args(tracyhip_ctx* ctx, const tracyhip_align_job* job, const tracyhip_params* prm, int mem, const tracyhip_align_result* out, bool* empty) {
  int rc = ctx_begin(ctx);
  if (rc) return rc;
  *empty = false;
  if (!job || !out || !prm) return set_error(TRACYHIP_ERR_ARG, "null job/result/params");
  if (mem != TRACYHIP_MEM_HOST && mem != TRACYHIP_MEM_DEVICE) return set_error(TRACYHIP_ERR_ARG, "bad mem kind");
  const uint32_t nt = job->ntraces;
  if (nt == 0) { *empty = true; return TRACYHIP_OK; }
  const tracyhip_seqset& sp = job->profiles;
  const tracyhip_seqset& sr = job->refs;
  if (sp.kind != TRACYHIP_SEQ_PROFILE || sr.kind != TRACYHIP_SEQ_CHAR) return set_error(TRACYHIP_ERR_ARG, "profiles must be PROFILE, refs CHAR");
  if (!sp.offset || !sp.length || !sr.offset || !sr.length || sp.count < nt) return set_error(TRACYHIP_ERR_ARG, "bad sequence sets");
  if (!out->score_fwd || !out->score_rev || !out->forward || !out->slice_begin || !out->slice_len || !out->ref_pos ||
      !out->score_final || !out->ops || !out->ops_offset || !out->ops_len)
    return set_error(TRACYHIP_ERR_ARG, "null result array");
  return TRACYHIP_OK;
}

// one context, no lanes: stream-ordered where the batch has the shape for it (stream.hip), else planned by the host
static int align_traces_one(tracyhip_ctx* ctx, const tracyhip_align_job* job, const tracyhip_params* prm, int mem, const tracyhip_align_result* out) {
  bool empty = false;
  int rc = align_check_args(ctx, job, prm, mem, out, &empty);
  if (rc || empty) return rc;
  rc = stream_align(ctx, job, prm, mem, out);
  if (rc != kStreamNo) return rc;
  return align_traces_legacy(ctx, job, prm, mem, out);
}

// sage.h:191-311 for one context, planned by the host between launches: what the stages share, one method per stage
struct AlignRun {
  tracyhip_ctx* ctx;
  const tracyhip_align_job* job;
  const tracyhip_params* prm;
  const int mem;
  const tracyhip_align_result* out;
  const uint32_t nt;
  const tracyhip_seqset& sp;
  const tracyhip_seqset& sr;
  hipStream_t st;
  tracyhip_params p;
  const void *d_prof = nullptr, *d_ref = nullptr;
  int32_t* d_verr = nullptr;
  std::vector<uint32_t> mf, mt, tl, rn, ridx;
  std::vector<B16TableDesc> td;  // substitution tables of the full profiles (band kernels)
  bool b16 = false;
  OrientOut oo;
  std::vector<TrimOut> h_trim;
  void *d_final_sc = nullptr, *d_ops = nullptr, *d_olen = nullptr;
  uint64_t ops_total = 0;

  AlignRun(tracyhip_ctx* c, const tracyhip_align_job* j, const tracyhip_params* q, int m, const tracyhip_align_result* o)
      : ctx(c), job(j), prm(q), mem(m), out(o), nt(j->ntraces), sp(j->profiles), sr(j->refs), st(c->stream), p(*q) {
    p.hfree = 1;  // AlignConfig<true,false> semiglobal (sage.h:165)
    p.vfree = 0;
  }
  // device result arrays (user's in DEVICE mode, ours in HOST mode)
  int dev_arr(DevBuf& b, void* user, size_t bytes, void** dptr) {
    if (mem == TRACYHIP_MEM_DEVICE) { *dptr = user; return TRACYHIP_OK; }
    HIP_TRY(b.ensure(bytes));
    *dptr = b.p;
    return TRACYHIP_OK;
  }

  int setup() {
    int rc;
    // ---- stage payloads, encode the references once ----
    const uint64_t ep = seqset_extent(sp), er = seqset_extent(sr);
    if ((rc = stage_in(ctx, ctx->d_in1, sp.data, ep * 4, mem, &d_prof))) return rc;
    if ((rc = stage_in(ctx, ctx->d_in2, sr.data, er, mem, &d_ref))) return rc;
    // The validation verdict (second word of d_err; run_dp owns the first) is read back together with the orientation
    // scores: no host round trip between the encode and the first score pass.
    HIP_TRY(ctx->d_err.ensure(kErrBytes));
    HIP_TRY(hipMemsetAsync(ctx->d_err.p, 0, kErrBytes, st));
    d_verr = static_cast<int32_t*>(ctx->d_err.p) + kErrVerdictWord;
    HIP_TRY(ctx->ensure_codes(er ? er : 1, st));
    if (er) {
      // Windows oriented by the caller (indexed genome) are never reverse-complemented here, and every other letter scores as
      // the all-zero profile column it is in the reference (getReferenceSlice upper-cases only; align.h:121-136): no check.
      hipLaunchKernelGGL(encode_codes_kernel, dim3((unsigned)((er + 4095) / 4096)), dim3(256), 0, st, static_cast<const uint8_t*>(d_ref),
                         ctx->codes(), er, ctx->special_blocks(), job->oriented ? (int32_t*)nullptr : d_verr);
      HIP_TRY(hipGetLastError());
    }
    // ---- geometry per trace ----
    mf.resize(nt); mt.resize(nt); tl.resize(nt); rn.resize(nt); ridx.resize(nt);
    uint64_t max_mn = 0;
    for (uint32_t t = 0; t < nt; ++t) {
      ridx[t] = job->ref_index ? job->ref_index[t] : t;
      if (ridx[t] >= sr.count) return set_error(TRACYHIP_ERR_ARG, "ref_index[%u] out of range", t);
      mf[t] = sp.length[t];
      rn[t] = sr.length[ridx[t]];
      uint32_t l = job->trim_left, r = job->trim_right;
      if ((uint64_t)l + r >= mf[t]) { l = 0; r = 0; }  // createProfile, profile.h:24-27
      tl[t] = l;
      mt[t] = mf[t] - (l + r);
      max_mn = std::max<uint64_t>(max_mn, (uint64_t)mf[t] + rn[t]);
    }
    if ((rc = check_params(&p, max_mn))) return rc;

    return TRACYHIP_OK;
  }

  int orient() {
    int rc;
    // ---- 1.-2. orientation (sage.h:239-247) + preliminary alignment (sage.h:258): orient_and_align ----
    std::vector<uint64_t> off1(nt);
    uint64_t tot1 = 0;
    for (uint32_t t = 0; t < nt; ++t) { off1[t] = tot1; tot1 += (uint64_t)mt[t] + rn[t]; }
    HIP_TRY(ctx->d_tmp[1].ensure(tot1 ? tot1 : 1));                       // ops of the preliminary alignment
    HIP_TRY(ctx->d_tmp[2].ensure(sizeof(uint64_t) * (size_t)nt));          // their offsets
    HIP_TRY(ctx->d_tmp[3].ensure(sizeof(uint32_t) * (size_t)nt));          // their lengths
    HIP_TRY(ctx->d_tmp[4].ensure(sizeof(int32_t) * (size_t)nt));           // preliminary scores
    HIP_TRY(ctx->h_tmp.ensure(sizeof(uint64_t) * (size_t)nt + (size_t)nt * 8));
    std::memcpy(ctx->h_tmp.p, off1.data(), sizeof(uint64_t) * (size_t)nt);
    HIP_TRY(hipMemcpyAsync(ctx->d_tmp[2].p, ctx->h_tmp.p, sizeof(uint64_t) * (size_t)nt, hipMemcpyHostToDevice, st));
    std::vector<uint64_t> a1o(nt), a2o(nt);
    for (uint32_t t = 0; t < nt; ++t) { a1o[t] = sp.offset[t] + tl[t]; a2o[t] = sr.offset[ridx[t]]; }
    // substitution tables of the full profiles for the band kernels (band16.h): the preliminary alignment (rows tl .. tl + mt) and the
    // final one (all rows) read them
    b16 = p.ge < 0 && p.go <= 0 && sub_limit(&p) <= kWideScore && !ctx->knobs.no_band16;
    if (b16) {
      td.resize(nt);
      for (uint32_t t = 0; t < nt; ++t) td[t] = B16TableDesc{sp.offset[t], 0, mf[t], mf[t], 0, 0};
      if ((rc = build_b16_tables(ctx, ctx->d_b16tab[2], d_prof, false, td, &p))) return rc;
    }
    OrientIn oi{};
    oi.nt = nt; oi.d_prof = d_prof; oi.a1_off = a1o.data(); oi.mf = mf.data(); oi.mt = mt.data(); oi.a2_off = a2o.data(); oi.rn = rn.data();
    oi.oriented = job->oriented; oi.exact = job->strand_by_certificate == 0; oi.d_verr = d_verr; oi.ends_only = true;
    if (b16) { oi.d_qp = static_cast<const int16_t*>(ctx->d_b16tab[2].p); oi.td = td.data(); oi.row0 = tl.data(); }
    oi.d_ops = static_cast<uint8_t*>(ctx->d_tmp[1].p); oi.d_ops_off = static_cast<const uint64_t*>(ctx->d_tmp[2].p);
    oi.d_ops_len = static_cast<uint32_t*>(ctx->d_tmp[3].p); oi.d_score = static_cast<int32_t*>(ctx->d_tmp[4].p);
    if ((rc = orient_and_align(ctx, p, oi, oo))) return rc;

    return TRACYHIP_OK;
  }

  int trim() {
    // ---- 3. trimReferenceSlice (sage.h:259) ----
    HIP_TRY(ctx->d_tmp[5].ensure(sizeof(TrimOut) * (size_t)nt));
    HIP_TRY(ctx->d_tmp[6].ensure(sizeof(uint32_t) * (size_t)nt + (size_t)nt));
    uint32_t* d_rn = static_cast<uint32_t*>(ctx->d_tmp[6].p);
    uint8_t* d_fwd = reinterpret_cast<uint8_t*>(d_rn + nt);
    {
      uint8_t* hp = static_cast<uint8_t*>(ctx->h_tmp.p) + sizeof(uint64_t) * (size_t)nt;
      std::memcpy(hp, rn.data(), sizeof(uint32_t) * (size_t)nt);
      std::memcpy(hp + sizeof(uint32_t) * (size_t)nt, oo.fwd.data(), nt);
      HIP_TRY(hipMemcpyAsync(d_rn, hp, sizeof(uint32_t) * (size_t)nt + nt, hipMemcpyHostToDevice, st));
    }
    if (oo.d_ends)  // the two ends of the preliminary alignment (origin-tracking sweep) instead of its ops
      hipLaunchKernelGGL(trim_from_ends_kernel, dim3((nt + 255) / 256), dim3(256), 0, st, oo.d_ends, static_cast<const uint32_t*>(d_rn),
                         static_cast<const uint8_t*>(d_fwd), (uint32_t)job->trim_left, (uint32_t)job->trim_right, nt, static_cast<TrimOut*>(ctx->d_tmp[5].p));
    else
      hipLaunchKernelGGL(trim_kernel, dim3(nt), dim3(64), 0, st, static_cast<const uint8_t*>(ctx->d_tmp[1].p),
                         static_cast<const uint64_t*>(ctx->d_tmp[2].p), static_cast<const uint32_t*>(ctx->d_tmp[3].p), d_rn, d_fwd,
                         job->trim_left, job->trim_right, nt, static_cast<TrimOut*>(ctx->d_tmp[5].p));
    HIP_TRY(hipGetLastError());
    h_trim.resize(nt);
    HIP_TRY(hipMemcpyAsync(h_trim.data(), ctx->d_tmp[5].p, sizeof(TrimOut) * (size_t)nt, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx_sync(ctx));

    return TRACYHIP_OK;
  }

  int final_alignment() {
    int rc;
    // ---- 4. final alignment gotoh(full profile, profile of the trimmed slice) (sage.h:260, 311) ----
    for (uint32_t t = 0; t < nt; ++t) ops_total = std::max<uint64_t>(ops_total, out->ops_offset[t] + mf[t] + h_trim[t].len);
    if ((rc = dev_arr(ctx->d_scores, out->score_final, sizeof(int32_t) * (size_t)nt, &d_final_sc))) return rc;
    if ((rc = dev_arr(ctx->d_ops, out->ops, ops_total ? ops_total : 1, &d_ops))) return rc;
    if ((rc = dev_arr(ctx->d_ops_len, out->ops_len, sizeof(uint32_t) * (size_t)nt, &d_olen))) return rc;
    HIP_TRY(ctx->h_off.ensure(sizeof(uint64_t) * (size_t)nt));
    std::memcpy(ctx->h_off.p, out->ops_offset, sizeof(uint64_t) * (size_t)nt);
    HIP_TRY(ctx->d_ops_off.ensure(sizeof(uint64_t) * (size_t)nt));
    HIP_TRY(hipMemcpyAsync(ctx->d_ops_off.p, ctx->h_off.p, sizeof(uint64_t) * (size_t)nt, hipMemcpyHostToDevice, st));
    {
      DpProblem pb;
      DpProblemLease lease(ctx, pb);
      pb.mode = MODE_QP;
      pb.a1_profile = true;
      pb.d_a1 = d_prof;
      pb.d_a2 = ctx->codes();
      pb.desc.resize(nt);
      pb.k.resize(nt);
      // Certified diagonal band: the slice was cut to the aligned region, so the path of the final
      // alignment runs along the diagonal.  With top = the most the diagonal steps of ANY path can add up to (sum of the row
      // maxima of the profile) a path that leaves the diagonals [-W - (m-n)+, W + (n-m)+] makes more than W interior gap steps
      // and scores at most top - |ge| (W + 1).  The traceback DP runs on the band only (PAIR_BANDED: short strips, every pass
      // sweeps the columns its rows can reach); if its score S_b beats that bound, S_b is the optimum, every optimal path and
      // every tie the traceback tests lies inside, and scores, bits and path are those of the whole matrix.  Pairs that do not
      // certify are repeated on the whole matrix.  W = 48 by default (TRACYHIP_BAND_W=<W>; 0 = whole matrices): the traceback
      // launch takes 3.5 instead of 4.5 ms per 10 000 traces.  The traceback words keep the whole-matrix layout (four passes of
      // n + 63 steps, of which a pass writes a third); a compact layout would shrink the workspace, not the work.
      // Without the variable every pair gets the width its preliminary alignment suggests: the gap columns that alignment's score
      // allowed (OrientOut::gap) + 48 for what the trimmed ends add, within [32, 96]; 48 where that is not known.  (A pair that does not
      // certify costs a launch of its own at the end of the step -- 0.7 ms for a single pair -- so the width errs on the wide side.)
      const bool band_env = ctx->knobs.band_w >= 0;
      // (developer knob band_w, tracyhip_set_option: clamped to [0, 4096] -- widths the band forms cannot hold simply leave the pair on
      // the whole matrix)
      const int32_t bandW = (p.ge < 0 && p.go <= 0 && sub_limit(&p) <= kWideScore) ? (band_env ? ctx->knobs.band_w : 48) : 0;
      std::vector<int32_t> band_of(nt, bandW);
      if (!band_env && bandW > 0 && oo.gap.size() == nt)
        for (uint32_t t = 0; t < nt; ++t) band_of[t] = (int32_t)std::min<uint32_t>(96u, std::max<uint32_t>(32u, oo.gap[t] + 48u));
      constexpr int kBandK = 4;
      std::vector<uint8_t> banded(nt, 0);  // 1: multi-pass form of the whole-matrix kernel (PAIR_BANDED), 2: band kernels
      uint32_t nbanded = 0;
      // band kernels (band16.h) where the band fits them: four pairs per wave, only the band's cells swept and stored
      Band16Job j16;
      j16.kind = 0;
      j16.d_qp = static_cast<const int16_t*>(ctx->d_b16tab[2].p);
      j16.d_codes = ctx->codes();
      std::vector<PairDesc> whole(nt);  // every pair as a whole-matrix problem (what a pair that does not certify is repeated as)
      pb.desc.clear();
      pb.k.clear();
      for (uint32_t t = 0; t < nt; ++t) {
        PairDesc d{};
        d.a1_off = sp.offset[t];
        d.a1_stride = mf[t];
        d.m = mf[t];
        d.n = h_trim[t].len;
        d.a2_stride = d.n;
        // oriented slice [ri, ri+len): forward reads it in place, reverse reads original
        // [n-ri-len, n-ri) backwards with complemented codes
        d.a2_off = sr.offset[ridx[t]] + (oo.rc[t] ? rn[t] - h_trim[t].ri - h_trim[t].len : h_trim[t].ri);
        d.flags = oo.rc[t] ? PAIR_A2_REVCOMP : 0;
        d.out = t;
        whole[t] = d;
        int kt = choose_k(d.m, MODE_QP);
        if (b16 && bandW > 0 && d.m && d.n && 4ull * ((d.n + 7u) & ~3u) + b16_table_bytes(12) <= 60u * 1024u) {
          const int64_t over = (int64_t)d.n - (int64_t)d.m, aover = over < 0 ? -over : over;
          int64_t bw = band_of[t];
          const int64_t fit = ((int64_t)b16_max_window(12) - 12 - aover) / 2;  // the widest band the kernels sweep
          if (bw > fit && fit >= 24) bw = fit;
          const int32_t dlo = (int32_t)(-bw - (over < 0 ? -over : 0)), dhi = (int32_t)(bw + (over > 0 ? over : 0));
          const int K = band16_pick_k(dlo, dhi);
          if (K) {
            band_of[t] = (int32_t)bw;
            PairDesc q = d;
            q.a1_off = td[t].out_off; q.a1_stride = td[t].stride; q.ckpt_off = band_pack(dlo, dhi); q.lastrow_off = 0;
            j16.desc.push_back(q); j16.k.push_back(K);
            banded[t] = 2;
            ++nbanded;
            continue;
          }
        }
        if (bandW > 0 && d.m && d.n) {
          const int64_t bw = band_of[t];
          const int64_t over = (int64_t)d.n - (int64_t)d.m;
          const int64_t width = 2 * bw + (over < 0 ? -over : over);        // diagonals of the band
          const int64_t rows_pass = 64 * kBandK;
          // worth it when a pass sweeps well under half of the columns and there are passes to speak of
          if ((int64_t)d.m >= 3 * rows_pass && 2 * (rows_pass + width) < (int64_t)d.n) {
            d.flags |= PAIR_BANDED;
            d.ckpt_off = band_pack((int32_t)(-bw - (over < 0 ? -over : 0)), (int32_t)(bw + (over > 0 ? over : 0)));
            kt = kBandK;
            banded[t] = 1;
            ++nbanded;
          }
        }
        pb.desc.push_back(d);
        pb.k.push_back(kt);
      }
      // The multi-pass form keeps the whole-matrix word layout on strips of four rows -- about four times the traceback words of the
      // plain form.  It stays within the limit run_dp plans with (the caller's workspace limit, or this context's share of the free
      // memory): a pair whose banded words exceed it, and the whole batch if the sum does, go back to whole matrices.
      {
        uint64_t limit = ctx->ws_limit;
        if (limit == 0) {
          size_t fr = 0, tot = 0;
          HIP_TRY(hipMemGetInfo(&fr, &tot));
          limit = (uint64_t)(fr * 0.70 / ctx->mem_share) + ctx->d_bits.cap;
        }
        uint64_t bytes = 0;
        bool unband_all = false;
        for (PairDesc& d : pb.desc) {
          if (!(d.flags & PAIR_BANDED)) continue;
          const uint64_t b = (uint64_t)num_passes(d.m, kBandK) * steps_per_pass(d.n) * 64u * 8u;
          if (b > limit) { d.flags &= ~PAIR_BANDED; d.ckpt_off = 0; banded[d.out] = 0; --nbanded; continue; }
          bytes += b;
        }
        if (bytes > limit) unband_all = true;
        for (size_t q = 0; q < pb.desc.size(); ++q) {
          PairDesc& d = pb.desc[q];
          if (unband_all && (d.flags & PAIR_BANDED)) { d.flags &= ~PAIR_BANDED; d.ckpt_off = 0; banded[d.out] = 0; --nbanded; }
          if (!(d.flags & PAIR_BANDED)) pb.k[q] = choose_k(d.m, MODE_QP);
        }
      }
      int32_t* d_top = nullptr;
      if (nbanded) {  // the bound's top, per trace, while the DP runs
        DevBuf& b = ctx->d_tmp[6];
        HIP_TRY(b.ensure((sizeof(RowMaxDesc) + sizeof(int32_t)) * (size_t)nt));
        RowMaxDesc* d_rm = static_cast<RowMaxDesc*>(b.p);
        d_top = reinterpret_cast<int32_t*>(d_rm + nt);
        HIP_TRY(ctx->h_tmp.ensure(sizeof(RowMaxDesc) * (size_t)nt));  // (pinned, free at this point of the call: no host wait)
        RowMaxDesc* hrm = static_cast<RowMaxDesc*>(ctx->h_tmp.p);
        for (uint32_t t = 0; t < nt; ++t) hrm[t] = RowMaxDesc{sp.offset[t], mf[t], mf[t], 0u};
        HIP_TRY(hipMemcpyAsync(d_rm, hrm, sizeof(RowMaxDesc) * (size_t)nt, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(rowmax_rest_kernel, dim3(nt), dim3(64), 0, st, static_cast<const RowMaxDesc*>(d_rm), static_cast<const float*>(d_prof),
                           (float)p.match, (float)p.mismatch, d_top);
        HIP_TRY(hipGetLastError());
      }
      if (!j16.desc.empty()) {
        HIP_TRY(hipMemsetAsync(ctx->d_err.p, 0, sizeof(int32_t) * kErrWords, st));
        rc = run_band16(ctx, j16, &p, static_cast<int32_t*>(d_final_sc), nullptr, static_cast<uint8_t*>(d_ops), static_cast<const uint64_t*>(ctx->d_ops_off.p),
                        static_cast<uint32_t*>(d_olen));
        if (rc == kWiden) rc = set_error(TRACYHIP_ERR_RANGE, "profile values outside the range of the traceback kernels");
        if (rc) return rc;
      }
      if ((rc = run_dp(ctx, pb, &p, false, true, static_cast<int32_t*>(d_final_sc), static_cast<uint8_t*>(d_ops),
                       static_cast<const uint64_t*>(ctx->d_ops_off.p), static_cast<uint32_t*>(d_olen))))
        return rc;
      if (nbanded) {
        std::vector<int32_t> h_top(nt), h_sb(nt);
        std::vector<uint32_t> h_ol(nt);
        HIP_TRY(hipMemcpyAsync(h_top.data(), d_top, sizeof(int32_t) * (size_t)nt, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(h_sb.data(), d_final_sc, sizeof(int32_t) * (size_t)nt, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(h_ol.data(), d_olen, sizeof(uint32_t) * (size_t)nt, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx_sync(ctx));
        std::vector<PairDesc> again;
        std::vector<int> again_k;
        for (uint32_t t = 0; t < nt; ++t) {
          const int64_t lose = (int64_t)(-(int64_t)p.ge) * ((int64_t)band_of[t] + 1);
          if (!banded[t] || ((int64_t)h_sb[t] > (int64_t)h_top[t] - lose && h_ol[t] != 0)) continue;
          again.push_back(whole[t]);
          again_k.push_back(choose_k(whole[t].m, MODE_QP));
        }
        ctx->stats.final_banded += nbanded; ctx->stats.final_repeated += (uint32_t)again.size();
        if (ctx->knobs.verbose) {
          int64_t wsum = 0, lsum = 0, lmax = 0, xmax = -1000000;
          for (uint32_t t = 0; t < nt; ++t) {
            wsum += band_of[t]; const int64_t l = (int64_t)h_top[t] - h_sb[t]; lsum += l; lmax = std::max(lmax, l);
            if (oo.gap.size() == nt) xmax = std::max<int64_t>(xmax, l / (-(int64_t)p.ge) - (int64_t)oo.gap[t]);
          }
          fprintf(stderr, "band: %u of %u pairs banded (%zu on the band kernels), %zu repeated; mean W %.1f, mean top - S_b %.1f, max %lld; max needed W - gap of the trimmed alignment %lld\n", nbanded, nt,
                  j16.desc.size(), again.size(), (double)wsum / nt, (double)lsum / nt, (long long)lmax, (long long)xmax);
        }
        if (!again.empty()) {
          pb.desc.swap(again);
          pb.k.swap(again_k);
          if ((rc = run_dp(ctx, pb, &p, false, true, static_cast<int32_t*>(d_final_sc), static_cast<uint8_t*>(d_ops),
                           static_cast<const uint64_t*>(ctx->d_ops_off.p), static_cast<uint32_t*>(d_olen))))
            return rc;
        }
      }
    }

    return TRACYHIP_OK;
  }

  int results() {
    int rc;
    // ---- results ----
    // host-decided arrays go out from one pinned staging block (copies from pageable memory are staged by the runtime and
    // cost a host round trip each)
    HIP_TRY(ctx->h_res.ensure((size_t)nt * (5 * sizeof(uint32_t) + 1)));
    uint32_t* r32 = static_cast<uint32_t*>(ctx->h_res.p);
    uint8_t* r8 = reinterpret_cast<uint8_t*>(r32 + 5 * (size_t)nt);
    if (job->oriented != nullptr) std::copy(oo.sc2.begin(), oo.sc2.begin() + nt, oo.sc2.begin() + nt);  // one orientation: both arrays report its score
    for (uint32_t t = 0; t < nt; ++t) {
      r32[t] = (uint32_t)oo.sc2[t];
      r32[(size_t)nt + t] = (uint32_t)oo.sc2[(size_t)nt + t];
      r32[2 * (size_t)nt + t] = h_trim[t].ri;
      r32[3 * (size_t)nt + t] = h_trim[t].len;
      r32[4 * (size_t)nt + t] = h_trim[t].pos;
      r8[t] = oo.fwd[t];
    }
    const hipMemcpyKind up = (mem == TRACYHIP_MEM_HOST) ? hipMemcpyHostToHost : hipMemcpyHostToDevice;
    HIP_TRY(hipMemcpyAsync(out->score_fwd, r32, sizeof(int32_t) * (size_t)nt, up, st));
    HIP_TRY(hipMemcpyAsync(out->score_rev, r32 + nt, sizeof(int32_t) * (size_t)nt, up, st));
    HIP_TRY(hipMemcpyAsync(out->slice_begin, r32 + 2 * (size_t)nt, sizeof(uint32_t) * (size_t)nt, up, st));
    HIP_TRY(hipMemcpyAsync(out->slice_len, r32 + 3 * (size_t)nt, sizeof(uint32_t) * (size_t)nt, up, st));
    HIP_TRY(hipMemcpyAsync(out->ref_pos, r32 + 4 * (size_t)nt, sizeof(uint32_t) * (size_t)nt, up, st));
    HIP_TRY(hipMemcpyAsync(out->forward, r8, nt, up, st));
    if (out->score_prelim) {
      if ((rc = copy_out(ctx, mem, out->score_prelim, static_cast<const int32_t*>(ctx->d_tmp[4].p), nt))) return rc;
    }
    if (mem == TRACYHIP_MEM_HOST) {
      if ((rc = copy_out(ctx, mem, out->score_final, static_cast<const int32_t*>(d_final_sc), nt))) return rc;
      if ((rc = copy_out(ctx, mem, out->ops, static_cast<const uint8_t*>(d_ops), ops_total))) return rc;
      if ((rc = copy_out(ctx, mem, out->ops_len, static_cast<const uint32_t*>(d_olen), nt))) return rc;
    }
    HIP_TRY(ctx_sync(ctx));
    return TRACYHIP_OK;
  }
};

int tracyhip::align_traces_legacy(tracyhip_ctx* ctx, const tracyhip_align_job* job, const tracyhip_params* prm, int mem,
                                  const tracyhip_align_result* out) {
  bool empty = false;
  int rc = align_check_args(ctx, job, prm, mem, out, &empty);
  if (rc || empty) return rc;
  AlignRun r(ctx, job, prm, mem, out);
  if ((rc = r.setup())) return rc;
  if ((rc = r.orient())) return rc;             // 1.-2. orientation (sage.h:239-247) + preliminary alignment (sage.h:258)
  if ((rc = r.trim())) return rc;               // 3. trimReferenceSlice (sage.h:259)
  if ((rc = r.final_alignment())) return rc;    // 4. gotoh(full profile, trimmed slice) (sage.h:260, 311)
  return r.results();
}

// ---- lanes: one call, several chunks in flight (tracyhip_set_lanes) ----------------------------------
namespace {

// a contiguous run [lo, lo + k) of a sequence set as a set of its own: offsets rebased to the run's first element
// so that host-staged payloads travel once, with the chunk that uses them
struct SubSet {
  tracyhip_seqset s;
  std::vector<uint64_t> off;
};
void sub_seqset(const tracyhip_seqset& full, uint32_t lo, uint32_t k, size_t elem_bytes, SubSet& o) {
  uint64_t base = k ? ~0ull : 0ull;
  for (uint32_t i = 0; i < k; ++i) base = std::min<uint64_t>(base, full.offset[lo + i]);
  o.off.resize(k);
  for (uint32_t i = 0; i < k; ++i) o.off[i] = full.offset[lo + i] - base;
  o.s = full;
  o.s.data = full.data ? static_cast<const char*>(full.data) + base * elem_bytes : nullptr;
  o.s.offset = o.off.data();
  o.s.length = full.length + lo;
  o.s.count = k;
}
// result regions addressed by a host offset array: same idea (the chunk sees its own window of the buffer)
struct SubOffsets {
  std::vector<uint64_t> off;
  uint64_t base = 0;
};
void sub_offsets(const uint64_t* full, uint32_t lo, uint32_t k, SubOffsets& o) {
  o.base = k ? full[lo] : 0;
  o.off.resize(k);
  for (uint32_t i = 0; i < k; ++i) o.off[i] = full[lo + i] - o.base;
}
bool nondecreasing(const uint64_t* a, uint32_t n) {
  for (uint32_t i = 1; i < n; ++i)
    if (a[i] < a[i - 1]) return false;
  return true;
}
template <class T>
T* shifted(T* p, uint64_t by) { return p ? p + by : nullptr; }

// run fn(context, chunk index, lo, k) for every chunk of [0, nt): chunk 0 on the calling thread, the others on a thread of
// their own (each sets its context's device); first error wins.  Contexts may sit on one device (lanes) or on several
// (a device group).
template <class Fn>
int run_chunks(const std::vector<tracyhip_ctx*>& ctxs, uint32_t nt, Fn fn) {
  const uint32_t L = (uint32_t)ctxs.size();
  std::vector<int> rcs(L, TRACYHIP_OK);
  std::vector<std::string> msgs(L);
  std::vector<std::thread> th;
  auto bounds = [&](uint32_t c, uint32_t& lo, uint32_t& hi) { lo = (uint32_t)((uint64_t)nt * c / L); hi = (uint32_t)((uint64_t)nt * (c + 1) / L); };
  for (uint32_t c = 1; c < L; ++c) {
    uint32_t lo, hi;
    bounds(c, lo, hi);
    th.emplace_back([&, c, lo, hi]() {
      rcs[c] = fn(ctxs[c], c, lo, hi - lo);
      if (rcs[c] != TRACYHIP_OK) msgs[c] = tracyhip_last_error();  // the message lives in that thread
    });
  }
  {
    uint32_t lo, hi;
    bounds(0, lo, hi);
    rcs[0] = fn(ctxs[0], 0, lo, hi - lo);
    if (rcs[0] != TRACYHIP_OK) msgs[0] = tracyhip_last_error();
  }
  for (auto& t : th) t.join();
  for (uint32_t c = 0; c < L; ++c)
    if (rcs[c] != TRACYHIP_OK) return set_error(rcs[c], "%s", msgs[c].c_str());
  return TRACYHIP_OK;
}
template <class Fn>
int run_lanes(tracyhip_ctx* ctx, uint32_t nt, Fn fn) {
  HIP_TRY(ctx_sync(ctx));  // inputs the caller enqueued on the context's stream
  std::vector<tracyhip_ctx*> ctxs{ctx};
  ctxs.insert(ctxs.end(), ctx->lanes.begin(), ctx->lanes.end());
  for (auto* c : ctxs) c->mem_share = (uint32_t)ctxs.size();  // the lanes plan their workspaces concurrently, on one device
  const int rc = run_chunks(ctxs, nt, fn);
  for (auto* c : ctxs) c->mem_share = 1;
  return rc;
}
constexpr uint32_t kMinLaneChunk = 64;  // below this a chunk cannot fill the device anyway

}  // namespace

namespace {
bool align_splittable(const tracyhip_align_job* job, const tracyhip_align_result* out, const tracyhip_params* prm, uint32_t parts) {
  return parts >= 2 && job && out && prm && job->ntraces >= parts * kMinLaneChunk && job->profiles.offset && job->profiles.length &&
         job->refs.offset && job->refs.length && job->profiles.count >= job->ntraces && out->ops_offset &&
         (job->ref_index || job->refs.count >= job->ntraces) && nondecreasing(out->ops_offset, job->ntraces);
}
// traces [lo, lo + k) of an align job as a job of its own (sequence sets and result regions rebased to the chunk)
struct AlignChunk {
  tracyhip_align_job j;
  tracyhip_align_result o;
  SubSet sp, sr;
  SubOffsets so;
  AlignChunk(const tracyhip_align_job* job, const tracyhip_align_result* out, uint32_t lo, uint32_t k) : j(*job), o(*out) {
    j.ntraces = k;
    sub_seqset(job->profiles, lo, k, sizeof(float), sp);
    j.profiles = sp.s;
    if (job->ref_index) j.ref_index = job->ref_index + lo;  // shared references: the whole set goes with every chunk
    else { sub_seqset(job->refs, lo, k, 1, sr); j.refs = sr.s; }
    j.oriented = shifted(job->oriented, lo);
    sub_offsets(out->ops_offset, lo, k, so);
    o.score_fwd = shifted(out->score_fwd, lo); o.score_rev = shifted(out->score_rev, lo); o.forward = shifted(out->forward, lo);
    o.score_prelim = shifted(out->score_prelim, lo); o.slice_begin = shifted(out->slice_begin, lo);
    o.slice_len = shifted(out->slice_len, lo); o.ref_pos = shifted(out->ref_pos, lo); o.score_final = shifted(out->score_final, lo);
    o.ops = shifted(out->ops, so.base); o.ops_offset = so.off.data(); o.ops_len = shifted(out->ops_len, lo);
  }
};
}  // namespace

extern "C" int tracyhip_align_traces(tracyhip_ctx* ctx, const tracyhip_align_job* job, const tracyhip_params* prm, int mem,
                                     const tracyhip_align_result* out) {
  if (!ctx) return set_error(TRACYHIP_ERR_ARG, "null context");
  reset_call_stats(ctx, job ? job->ntraces : 0);
  const uint32_t L = (uint32_t)ctx->lanes.size() + 1;
  if (!align_splittable(job, out, prm, L)) return align_traces_one(ctx, job, prm, mem, out);
  int rc = ctx_begin(ctx);
  if (rc) return rc;
  return run_lanes(ctx, job->ntraces, [&](tracyhip_ctx* lane, uint32_t, uint32_t lo, uint32_t k) -> int {
    if (k == 0) return TRACYHIP_OK;
    if (lane != ctx) lane->stats.traces = k;  // (this lane took part: tracyhip_last_call_stats)
    AlignChunk c(job, out, lo, k);
    return align_traces_one(lane, &c.j, prm, mem, &c.o);
  });
}

// =====================================================================================================
// tracyhip_decompose_traces: the hot section of `tracy decompose` (indigo.h:190-388), FASTA reference.
// =====================================================================================================
#include "decompose_launch.h"

namespace {

template <class T>
int upload(tracyhip_ctx* ctx, DevBuf& b, const std::vector<T>& v, const T** out) {
  HIP_TRY(b.ensure(sizeof(T) * std::max<size_t>(v.size(), 1)));
  if (!v.empty()) { HIP_TRY(hipMemcpyAsync(b.p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice, ctx->stream)); HIP_TRY(ctx_sync(ctx)); }
  *out = static_cast<const T*>(b.p);
  return TRACYHIP_OK;
}

struct DevOut {  // a result array: the user's (DEVICE) or a staging buffer (HOST) copied back at the end
  void* dev = nullptr;
  void* user = nullptr;
  size_t bytes = 0;
};

}  // namespace

static int decompose_check_args(tracyhip_ctx* ctx, const tracyhip_decompose_job* job, const tracyhip_params* prm, int mem, const tracyhip_decompose_result* out,
                                bool* empty) {
  int rc = ctx_begin(ctx);
  if (rc) return rc;
  *empty = false;
  if (!job || !out || !prm) return set_error(TRACYHIP_ERR_ARG, "null job/result/params");
  if (mem != TRACYHIP_MEM_HOST && mem != TRACYHIP_MEM_DEVICE) return set_error(TRACYHIP_ERR_ARG, "bad mem kind");
  const uint32_t nt = job->ntraces;
  if (nt == 0) { *empty = true; return TRACYHIP_OK; }
  const tracyhip_seqset& sp = job->profiles;
  const tracyhip_seqset& sr = job->refs;
  const tracyhip_basecalls& bc = job->bc;
  const tracyhip_decomp_params& dp = job->dprm;
  if (sp.kind != TRACYHIP_SEQ_PROFILE || sr.kind != TRACYHIP_SEQ_CHAR || sp.count < nt || bc.ntraces != nt)
    return set_error(TRACYHIP_ERR_ARG, "bad sequence sets");
  if (!bc.primary || !bc.secondary || !bc.bc_offset || !bc.bc_len || (!bc.peaks && (!bc.signal || !bc.signal_offset || !bc.nsamples || !bc.bcpos)))
    return set_error(TRACYHIP_ERR_ARG, "null basecall arrays");  // (the peak table, or signal + bcpos to build it from)
  if (dp.maxindel < 1 || dp.maxindel > kMaxIndelGlobal) return set_error(TRACYHIP_ERR_RANGE, "maxindel must be in [1, %d]", kMaxIndelGlobal);
  if (!out->bp || !out->status || !out->score_fwd || !out->score_rev || !out->forward || !out->score_trim || !out->dcp_indel ||
      !out->dcp_err || !out->dcp_offset || !out->dstatus || !out->secdecomp || !out->fractions)
    return set_error(TRACYHIP_ERR_ARG, "null result array");
  for (int k = 0; k < 3; ++k)
    if (!out->score[k] || !out->ops[k] || !out->ops_offset[k] || !out->ops_len[k]) return set_error(TRACYHIP_ERR_ARG, "null allele alignment arrays");
  for (int k = 0; k < 2; ++k)
    if (!out->slice_begin[k] || !out->slice_len[k] || !out->ref_pos[k]) return set_error(TRACYHIP_ERR_ARG, "null allele slice arrays");
  return TRACYHIP_OK;
}

static int decompose_traces_one(tracyhip_ctx* ctx, const tracyhip_decompose_job* job, const tracyhip_params* prm, int mem,
                               const tracyhip_decompose_result* out) {
  bool empty = false;
  int rc = decompose_check_args(ctx, job, prm, mem, out, &empty);
  if (rc || empty) return rc;
  rc = stream_decompose(ctx, job, prm, mem, out);
  if (rc != kStreamNo) return rc;
  return decompose_traces_legacy(ctx, job, prm, mem, out);
}

// indigo.h:190-388 for one context, planned by the host between launches: what the stages share, one method per stage
struct DecomposeRun {
  tracyhip_ctx* ctx;
  const tracyhip_decompose_job* job;
  const tracyhip_params* prm;
  const int mem;
  const tracyhip_decompose_result* out;
  const uint32_t nt;
  const tracyhip_seqset& sp;
  const tracyhip_seqset& sr;
  const tracyhip_basecalls& bc;
  const tracyhip_decomp_params& dp;
  const tracyhip_seqset& srp;  // wildtype-trace reference (indigo.h:249-289), or data == null
  hipStream_t st;
  tracyhip_params p, pglobal;
  const uint32_t TL, TR;
  const bool wildtype, given, shared_stages;
  StageClock stage_clock, sc6;
  std::vector<uint32_t> mf, mt, tl, rn, ridx, sl, soff;  // sl / soff: trimmedSeq(length, offset)
  uint64_t max_mn = 0, ep = 0, er = 0, sext = 0, bext = 0, dext = 0, tot1 = 0;
  uint32_t maxbc = 0, maxcol = 0;
  int nbuf = 0;
  DevBuf &b_sc2, &b_ops1, &b_len1, &b_r0, &b_r1, &b_hst, &b_cq1, &b_cq2, &b_cqf, &b_opsA, &b_lenA, &b_trimA, &b_rnfw, &b_ends;
  const void *d_prof = nullptr, *d_ref = nullptr, *d_sig = nullptr, *d_pos = nullptr, *d_refprof = nullptr, *d_peaks = nullptr;
  std::vector<DevOut> outs;
  void *d_pri = nullptr, *d_sec = nullptr, *d_bp = nullptr, *d_sd = nullptr, *d_fr = nullptr, *d_di = nullptr, *d_de = nullptr, *d_dst = nullptr, *d_strim = nullptr;
  std::vector<int32_t> h_sc2;
  std::vector<uint8_t> h_fwd, h_rc;  // rs.forward / "read the window as its reverse complement"
  std::vector<uint64_t> off1, offA;
  const uint64_t *d_off1 = nullptr, *d_offA = nullptr;
  std::vector<PairDesc> desc_trim;
  const uint32_t* d_len1 = nullptr;
  uint8_t *d_cq_ref = nullptr, *d_cq_sd = nullptr;
  bool try_cq = false, use_cq = false;
  int32_t h_cq_flag = 1;
  int cq_codes = 6;
  std::vector<int32_t> h_hst, h_strim, h_status;
  std::vector<uint32_t> h_len1;
  std::vector<TrimOut> h_trimA[2];
  std::vector<B16TableDesc> td_pri;  // substitution tables of the primary alleles (band kernels), kept for allele 1 vs allele 2
  void *d_scoreK[3] = {}, *d_opsK[3] = {}, *d_lenK[3] = {};

  // what the stages of one allele k (0: primary, 1: secDecompose) hand each other (indigo.h:355-365)
  struct Allele {
    const void* seq = nullptr;
    DpProblem pb;
    DpProblemLease lease;
    bool b16 = false, use_origin = false, subwin = false;
    std::vector<B16TableDesc> td;
    std::vector<int32_t> h_s1;      // S* of gotoh(seq, window) where the certifying sweep ran
    std::vector<int64_t> gap_of;    // its gap-step budget; -1: not known
    std::vector<uint32_t> shift;    // columns of the window left of the sub-window the origin sweep runs on
    uint32_t* d_shift = nullptr;
    explicit Allele(tracyhip_ctx* c) : lease(c, pb) {}
  };

  DecomposeRun(tracyhip_ctx* c, const tracyhip_decompose_job* j, const tracyhip_params* q, int m, const tracyhip_decompose_result* o)
      : ctx(c), job(j), prm(q), mem(m), out(o), nt(j->ntraces), sp(j->profiles), sr(j->refs), bc(j->bc), dp(j->dprm), srp(j->ref_profiles), st(c->stream),
        p(*q), pglobal(*q), TL((uint32_t)j->dprm.trim_left), TR((uint32_t)j->dprm.trim_right), wildtype(j->ref_profiles.data != nullptr),
        given(j->oriented != nullptr), shared_stages(j->ref_profiles.data == nullptr), b_sc2(c->d_pipe[0]), b_ops1(c->d_pipe[1]), b_len1(c->d_pipe[2]),
        b_r0(c->d_pipe[3]), b_r1(c->d_pipe[4]), b_hst(c->d_pipe[5]), b_cq1(c->d_pipe[6]), b_cq2(c->d_pipe[7]), b_cqf(c->d_pipe[8]), b_opsA(c->d_pipe[9]),
        b_lenA(c->d_pipe[10]), b_trimA(c->d_pipe[11]), b_rnfw(c->d_pipe[12]), b_ends(c->d_pipe[13]) {
    nbuf = 14;
    p.hfree = 1;  // AlignConfig<true,false> semiglobal (indigo.h:164)
    p.vfree = 0;
    pglobal.hfree = 0;  // AlignConfig<false,false> (indigo.h:381)
    pglobal.vfree = 0;
  }
  DevBuf& buf() { return ctx->d_pipe[nbuf++]; }
  // a result array: the user's (DEVICE) or a staging buffer (HOST) copied back at the end
  int io(void* user, size_t bytes, bool upload_first, void** dptr) {
    if (mem == TRACYHIP_MEM_DEVICE) { *dptr = user; return TRACYHIP_OK; }
    DevBuf& b = buf();
    HIP_TRY(b.ensure(bytes ? bytes : 1));
    if (upload_first && bytes) HIP_TRY(hipMemcpyAsync(b.p, user, bytes, hipMemcpyHostToDevice, st));
    *dptr = b.p;
    outs.push_back(DevOut{b.p, user, bytes});
    return TRACYHIP_OK;
  }
  PairDesc qp_desc(uint32_t t, bool trimmed) const {
    PairDesc d{};
    d.a1_off = sp.offset[t] + (trimmed ? tl[t] : 0);
    d.a1_stride = mf[t];
    d.m = trimmed ? mt[t] : mf[t];
    d.a2_off = sr.offset[ridx[t]];
    d.n = rn[t];
    d.a2_stride = rn[t];
    d.out = t;
    return d;
  }

  int setup() {
    int rc;
    stage_clock.mark("decompose.0_setup");
    // ---- geometry ----
    mf.resize(nt); mt.resize(nt); tl.resize(nt); rn.resize(nt); ridx.resize(nt); sl.resize(nt); soff.resize(nt);
    for (uint32_t t = 0; t < nt; ++t) {
      ridx[t] = job->ref_index ? job->ref_index[t] : t;
      if (ridx[t] >= sr.count) return set_error(TRACYHIP_ERR_ARG, "ref_index[%u] out of range", t);
      mf[t] = sp.length[t];
      rn[t] = sr.length[ridx[t]];
      if (bc.bc_len[t] != mf[t]) return set_error(TRACYHIP_ERR_ARG, "trace %u: profile has %u columns but %u basecalls", t, mf[t], bc.bc_len[t]);
      if (bc.bc_len[t] >= 2u * kMaxIndelGlobal) return set_error(TRACYHIP_ERR_RANGE, "trace %u has %u basecalls; the scan tables hold < %d", t, bc.bc_len[t], 2 * kMaxIndelGlobal);
      uint32_t l = TL, r = TR;
      if ((uint64_t)l + r >= mf[t]) { l = 0; r = 0; }  // createProfile, profile.h:24-27
      tl[t] = l;
      mt[t] = mf[t] - (l + r);
      if ((uint64_t)(uint32_t)(TL + TR + 1) >= (uint64_t)mf[t]) { soff[t] = 0; sl[t] = mf[t]; }  // trimmedSeq, abif.h:68-75
      else { soff[t] = TL; sl[t] = mf[t] - TL - TR; }
      max_mn = std::max<uint64_t>(max_mn, (uint64_t)mf[t] + rn[t]);
      maxbc = std::max(maxbc, mf[t]);
      maxcol = std::max(maxcol, mt[t]);
    }
    if ((rc = check_params(&p, max_mn))) return rc;

    // ---- payloads ----
    ep = seqset_extent(sp); er = seqset_extent(sr);
    for (uint32_t t = 0; t < nt; ++t) {
      if (!bc.peaks) sext = std::max<uint64_t>(sext, bc.signal_offset[t] + 4ull * bc.nsamples[t]);
      bext = std::max<uint64_t>(bext, bc.bc_offset[t] + bc.bc_len[t]);
    }
    // wildtype-trace reference (indigo.h:249-289): the alignment of the trimmed trace runs against the wildtype PROFILE
    // (already oriented by the caller), everything after it against its primary basecalls (refs)
    if (wildtype) {
      if (!job->oriented) return set_error(TRACYHIP_ERR_ARG, "ref_profiles needs `oriented` (the caller picks the strand)");
      if (srp.kind != TRACYHIP_SEQ_PROFILE || srp.count != sr.count || !srp.offset || !srp.length)
        return set_error(TRACYHIP_ERR_ARG, "ref_profiles must be a PROFILE set parallel to refs");
      for (uint32_t i = 0; i < sr.count; ++i)
        if (srp.length[i] != sr.length[i]) return set_error(TRACYHIP_ERR_ARG, "ref_profiles[%u] and refs[%u] differ in length", i, i);
    }
    if ((rc = stage_in(ctx, buf(), sp.data, ep * 4, mem, &d_prof))) return rc;
    if ((rc = stage_in(ctx, buf(), sr.data, er, mem, &d_ref))) return rc;
    if (wildtype && (rc = stage_in(ctx, buf(), srp.data, seqset_extent(srp) * 4, mem, &d_refprof))) return rc;
    // the chromatogram is read at the basecalls' peak positions only: the caller's peak table, or signal + bcpos to build it from
    if (bc.peaks) { if ((rc = stage_in(ctx, buf(), bc.peaks, bext * 16, mem, &d_peaks))) return rc; }
    else {
      if ((rc = stage_in(ctx, buf(), bc.signal, sext * 4, mem, &d_sig))) return rc;
      if ((rc = stage_in(ctx, buf(), bc.bcpos, bext * 4, mem, &d_pos))) return rc;
    }
    for (uint32_t t = 0; t < nt; ++t) dext = std::max<uint64_t>(dext, out->dcp_offset[t] + 2ull * dp.maxindel + 2);
    if ((rc = io(bc.primary, bext, true, &d_pri))) return rc;
    if ((rc = io(bc.secondary, bext, true, &d_sec))) return rc;
    if ((rc = io(out->bp, sizeof(BreakpointOut) * (size_t)nt, false, &d_bp))) return rc;
    if ((rc = io(out->secdecomp, bext, false, &d_sd))) return rc;
    if ((rc = io(out->fractions, sizeof(double) * 2 * (size_t)nt, false, &d_fr))) return rc;
    if ((rc = io(out->dcp_indel, dext * 4, false, &d_di))) return rc;
    if ((rc = io(out->dcp_err, dext * 4, false, &d_de))) return rc;
    if ((rc = io(out->dstatus, sizeof(DecompOut) * (size_t)nt, false, &d_dst))) return rc;
    if ((rc = io(out->score_trim, sizeof(int32_t) * (size_t)nt, false, &d_strim))) return rc;

    // references: validate + encode
    HIP_TRY(ctx->d_err.ensure(kErrBytes));
    HIP_TRY(hipMemsetAsync(ctx->d_err.p, 0, kErrBytes, st));
    HIP_TRY(ctx->ensure_codes(er ? er : 1, st));
    if (er) {
      int32_t* d_verr = static_cast<int32_t*>(ctx->d_err.p) + kErrVerdictWord;
      // (as in tracyhip_align_traces: caller-oriented windows are taken as they are, no validation)
      hipLaunchKernelGGL(encode_codes_kernel, dim3((unsigned)((er + 4095) / 4096)), dim3(256), 0, st, static_cast<const uint8_t*>(d_ref),
                         ctx->codes(), er, ctx->special_blocks(), job->oriented ? (int32_t*)nullptr : d_verr);
      HIP_TRY(hipGetLastError());
      int32_t herr = 0;
      HIP_TRY(hipMemcpyAsync(&herr, d_verr, sizeof(int32_t), hipMemcpyDeviceToHost, st));
      HIP_TRY(ctx_sync(ctx));
      if ((herr & 4) && !wildtype) return set_error(TRACYHIP_ERR_ARG, "reference windows must be upper-case [ACGTN] (loadSingleFasta, fasta.h:54-95)");
    }

    return TRACYHIP_OK;
  }

  int orientation() {
    int rc;
    stage_clock.mark("decompose.2_orientation");
    // ---- 2. orientation (indigo.h:235-247) ----
    HIP_TRY(b_sc2.ensure(sizeof(int32_t) * 2 * (size_t)nt));
    // job->oriented: the references were anchored and oriented by the caller (indexed genome, indigo.h:213-218):
    // no orientation scores; oriented[t] = rs.forward only steers rs.pos in trimReferenceSlice
    // FASTA / indexed reference: orientation and the alignment of the trimmed trace run through the stages `tracy align`
    // uses (checkpointed 16-bit score pass, strand by certificate when the job opts in, band traceback).
    // Wildtype-trace reference: profile x profile, full-matrix traceback (the caller picked the strand).
    h_sc2.assign(2 * (size_t)nt, 0);
    if (!given && !shared_stages) {
      DpProblem pb;
      DpProblemLease lease(ctx, pb);
      pb.mode = MODE_QP; pb.a1_profile = true; pb.d_a1 = d_prof; pb.d_a2 = ctx->codes();
      pb.desc.resize(2 * (size_t)nt); pb.k.resize(2 * (size_t)nt);
      for (uint32_t t = 0; t < nt; ++t) {
        PairDesc d = qp_desc(t, true);
        pb.desc[t] = d;
        d.out = nt + t; d.flags = PAIR_A2_REVCOMP;
        pb.desc[nt + t] = d;
        pb.k[t] = pb.k[nt + t] = choose_k(d.m, MODE_QP);
      }
      if ((rc = run_dp(ctx, pb, &p, false, false, static_cast<int32_t*>(b_sc2.p), nullptr, nullptr, nullptr))) return rc;
      HIP_TRY(hipMemcpy(h_sc2.data(), b_sc2.p, sizeof(int32_t) * 2 * (size_t)nt, hipMemcpyDeviceToHost));
    }
    h_fwd.assign(nt, 0); h_rc.assign(nt, 0);
    if (!shared_stages)
      for (uint32_t t = 0; t < nt; ++t) {
        if (given) { h_fwd[t] = job->oriented[t] ? 1 : 0; h_rc[t] = 0; }
        else { h_fwd[t] = h_sc2[t] > h_sc2[nt + t] ? 1 : 0; h_rc[t] = !h_fwd[t]; }
      }

    return TRACYHIP_OK;
  }

  int gotoh_rows() {
    int rc;
    stage_clock.mark("decompose.3_gotoh");
    // ---- 3. gotoh(trimmedtrace, prefslice) + alignment rows (indigo.h:302) ----
    off1.resize(nt);
    for (uint32_t t = 0; t < nt; ++t) { off1[t] = tot1; tot1 += (uint64_t)mt[t] + rn[t]; }
    HIP_TRY(b_ops1.ensure(tot1 ? tot1 : 1));
    HIP_TRY(b_len1.ensure(sizeof(uint32_t) * (size_t)nt));
    HIP_TRY(b_r0.ensure(tot1 ? tot1 : 1));
    HIP_TRY(b_r1.ensure(tot1 ? tot1 : 1));
    if ((rc = upload(ctx, buf(), off1, &d_off1))) return rc;
    if (shared_stages) {
      std::vector<uint64_t> a1o(nt), a2o(nt);
      for (uint32_t t = 0; t < nt; ++t) { a1o[t] = sp.offset[t] + tl[t]; a2o[t] = sr.offset[ridx[t]]; }
      OrientIn oi{};
      oi.nt = nt; oi.d_prof = d_prof; oi.a1_off = a1o.data(); oi.mf = mf.data(); oi.mt = mt.data(); oi.a2_off = a2o.data(); oi.rn = rn.data();
      oi.oriented = job->oriented; oi.exact = job->strand_by_certificate == 0; oi.d_verr = nullptr;
      oi.d_ops = static_cast<uint8_t*>(b_ops1.p); oi.d_ops_off = d_off1; oi.d_ops_len = static_cast<uint32_t*>(b_len1.p);
      oi.d_score = static_cast<int32_t*>(d_strim);
      // the traceback of the trimmed trace on the band kernels (band16.h): substitution tables of the profiles
      std::vector<B16TableDesc> tdp;
      if (p.ge < 0 && p.go <= 0 && sub_limit(&p) <= kWideScore && !ctx->knobs.no_band16) {
        tdp.resize(nt);
        for (uint32_t t = 0; t < nt; ++t) tdp[t] = B16TableDesc{sp.offset[t], 0, mf[t], mf[t], 0, 0};
        HIP_TRY(hipMemsetAsync(ctx->d_err.p, 0, sizeof(int32_t) * kErrWords, st));
        if ((rc = build_b16_tables(ctx, ctx->d_b16tab[2], d_prof, false, tdp, &p))) return rc;
        oi.d_qp = static_cast<const int16_t*>(ctx->d_b16tab[2].p); oi.td = tdp.data(); oi.row0 = tl.data();
      }
      OrientOut oo;
      if ((rc = orient_and_align(ctx, p, oi, oo))) return rc;
      h_sc2 = oo.sc2; h_fwd = oo.fwd; h_rc = oo.rc;
    }
    desc_trim.resize(nt);
    {
      DpProblem pb;
      DpProblemLease lease(ctx, pb);
      pb.mode = wildtype ? MODE_PROF : MODE_QP; pb.a1_profile = true; pb.a2_profile = wildtype; pb.d_a1 = d_prof;
      pb.d_a2 = wildtype ? d_refprof : ctx->codes();
      pb.desc.resize(nt); pb.k.resize(nt);
      for (uint32_t t = 0; t < nt; ++t) {
        PairDesc d = qp_desc(t, true);
        d.flags = h_rc[t] ? PAIR_A2_REVCOMP : 0;
        if (wildtype) d.a2_off = srp.offset[ridx[t]];
        pb.desc[t] = d;
        desc_trim[t] = d;
        pb.k[t] = choose_k(d.m, pb.mode);
      }
      if (!shared_stages &&
          (rc = run_dp(ctx, pb, &p, false, true, static_cast<int32_t*>(d_strim), static_cast<uint8_t*>(b_ops1.p), d_off1,
                       static_cast<uint32_t*>(b_len1.p))))
        return rc;
    }
    {
      const PairDesc* dd;
      if ((rc = upload(ctx, buf(), desc_trim, &dd))) return rc;
      RowsArgs ra{};
      ra.pairs = dd;
      ra.a1 = d_prof; ra.a2 = wildtype ? d_refprof : d_ref;  // row 1: consensus characters of the (oriented) reference profile
      ra.a1_profile = 1; ra.a2_profile = wildtype ? 1 : 0; ra.a2_revcomp_flag = 1; ra.a2_onehot = wildtype ? 0 : 1;
      ra.ops = static_cast<const uint8_t*>(b_ops1.p);
      ra.ops_off = d_off1;
      ra.ops_len = static_cast<const uint32_t*>(b_len1.p);
      ra.rows0 = static_cast<uint8_t*>(b_r0.p);
      ra.rows1 = static_cast<uint8_t*>(b_r1.p);
      ra.npairs = nt;
      HIP_TRY(launch_alignment_rows(ra, st));
    }
    return TRACYHIP_OK;
  }

  int deconvolution() {
    int rc;
    stage_clock.mark("decompose.1_findBreakpoint");
    // ---- 1. findBreakpoint(trimmedtrace) (indigo.h:196) ----
    // (launched here, behind the sweeps: nothing before stage 4 reads it, and a kernel of single-wavefront workgroups that
    // is the first thing an idle GPU gets to run has been measured at 20 ms instead of 2)
    {
      std::vector<BpDesc> hd(nt);
      for (uint32_t t = 0; t < nt; ++t) hd[t] = BpDesc{sp.offset[t] + tl[t], mf[t], mt[t]};
      const BpDesc* dd;
      if ((rc = upload(ctx, buf(), hd, &dd))) return rc;
      if ((rc = launch_breakpoint(ctx, dd, nt, maxcol, static_cast<const float*>(d_prof), static_cast<BreakpointOut*>(d_bp)))) return rc;
    }

    // the alignment lengths stay on the device for the next kernels (they come to the host with the results of stage 5: a
    // synchronisation here leaves the GPU idle for 2 ms, and the short-wavefront kernels that follow then start at idle clocks)
    d_len1 = static_cast<const uint32_t*>(b_len1.p);

    stage_clock.mark("decompose.4_findHomozygousBreakpoint");
    // ---- 4. findHomozygousBreakpoint where the trace shows no shift (indigo.h:314-317) ----
    HIP_TRY(b_hst.ensure(sizeof(int32_t) * (size_t)nt));
    {
      std::vector<RowsDesc> hd(nt);
      for (uint32_t t = 0; t < nt; ++t) hd[t] = RowsDesc{off1[t], 0, 0};
      const RowsDesc* dd;
      if ((rc = upload(ctx, buf(), hd, &dd))) return rc;
      if ((rc = launch_homozygous(ctx, dd, static_cast<const uint8_t*>(b_r0.p), static_cast<const uint8_t*>(b_r1.p), nt,
                                  static_cast<BreakpointOut*>(d_bp), static_cast<int32_t*>(b_hst.p), d_len1)))
        return rc;
    }

    stage_clock.mark("decompose.5_decomposeAlleles");
    // ---- 5. decomposeAlleles, generateSecondaryDecomposed, allelicFraction (indigo.h:340-350) ----
    {
      std::vector<DecompDesc> hd(nt);
      for (uint32_t t = 0; t < nt; ++t) hd[t] = DecompDesc{off1[t], bc.bc_offset[t], out->dcp_offset[t], 0, mf[t], rn[t], 0};
      const DecompDesc* dd;
      if ((rc = upload(ctx, buf(), hd, &dd))) return rc;
      DecompArgs a{};
      a.desc = dd;
      a.rows0 = static_cast<const uint8_t*>(b_r0.p);
      a.rows1 = static_cast<const uint8_t*>(b_r1.p);
      a.primary = static_cast<uint8_t*>(d_pri);
      a.secondary = static_cast<uint8_t*>(d_sec);
      a.dcp_indel = static_cast<int32_t*>(d_di);
      a.dcp_err = static_cast<int32_t*>(d_de);
      a.out = static_cast<DecompOut*>(d_dst);
      a.prm = DecompParams{dp.trim_left, dp.trim_right, dp.maxindel, dp.madc};
      a.ntraces = nt;
      a.lens = d_len1;
      if ((rc = launch_decompose(ctx, a, static_cast<const BreakpointOut*>(d_bp), maxbc, 0, 0))) return rc;  // accounted below, once the lengths are here
      std::vector<BcDesc> hb(nt);
      for (uint32_t t = 0; t < nt; ++t) hb[t] = BcDesc{bc.peaks ? 0ull : bc.signal_offset[t], bc.bc_offset[t], bc.peaks ? 0u : bc.nsamples[t], mf[t]};
      const BcDesc* db;
      if ((rc = upload(ctx, buf(), hb, &db))) return rc;
      if (!d_peaks) {  // no table from the caller: built once, read by both kernels below
        DevBuf& pb = buf();
        HIP_TRY(pb.ensure(bext * 16 + 16));
        if ((rc = launch_peaks(ctx, db, nt, maxbc, static_cast<const int32_t*>(d_sig), static_cast<const int32_t*>(d_pos), static_cast<int32_t*>(pb.p)))) return rc;
        d_peaks = pb.p;
      }
      if ((rc = launch_secdecomp(ctx, db, nt, maxbc, static_cast<const int32_t*>(d_peaks),
                                 static_cast<const uint8_t*>(d_pri), static_cast<const uint8_t*>(d_sec), static_cast<uint8_t*>(d_sd))))
        return rc;
      if ((rc = launch_allelic_fraction(ctx, db, nt, maxbc, static_cast<const int32_t*>(d_peaks),
                                        static_cast<const uint8_t*>(d_pri), static_cast<const uint8_t*>(d_sd), TL, TR,
                                        static_cast<double*>(d_fr), 18ull * std::accumulate(mf.begin(), mf.end(), 0ull),
                                        [&] { uint64_t e = 0; for (uint32_t t = 0; t < nt; ++t) e = std::max<uint64_t>(e, bc.bc_offset[t] + mf[t]); return e; }())))
        return rc;
    }
    return TRACYHIP_OK;
  }

  // strings through the query-profile table where the basecall strings allow it; the verdicts of stages 3-5
  int read_back_verdicts() {
    // The allele-specific alignments below are string x string.  Basecall strings hold A, C, G, T, N only, so "row char == column
    // char ? match : mismatch" can come out of the query-profile table (MODE_CQ: one table look-up per step instead of compare +
    // select per cell); checked here on the strings as they are now, with the byte-compare kernels as the fallback.
    HIP_TRY(b_cq1.ensure((er ? er : 1) + 2 * kCodePad));
    HIP_TRY(b_cq2.ensure((bext ? bext : 1) + 2 * kCodePad));
    HIP_TRY(b_cqf.ensure(sizeof(int32_t)));
    d_cq_ref = static_cast<uint8_t*>(b_cq1.p) + kCodePad;
    d_cq_sd = static_cast<uint8_t*>(b_cq2.p) + kCodePad;
    try_cq = !ctx->knobs.no_cq && sub_limit(&p) <= kWideScore;
    h_cq_flag = 1;
    if (try_cq) {
      HIP_TRY(hipMemsetAsync(b_cq1.p, 5, (er ? er : 1) + 2 * kCodePad, st));
      HIP_TRY(hipMemsetAsync(b_cq2.p, 5, (bext ? bext : 1) + 2 * kCodePad, st));
      HIP_TRY(hipMemsetAsync(b_cqf.p, 0, sizeof(int32_t), st));
      if (er) hipLaunchKernelGGL(encode_cq_kernel, dim3((unsigned)((er + 4095) / 4096)), dim3(256), 0, st, static_cast<const uint8_t*>(d_ref), d_cq_ref, er, static_cast<int32_t*>(b_cqf.p), (uint8_t*)nullptr);
      if (bext) {
        hipLaunchKernelGGL(encode_cq_kernel, dim3((unsigned)((bext + 4095) / 4096)), dim3(256), 0, st, static_cast<const uint8_t*>(d_sd), d_cq_sd, bext, static_cast<int32_t*>(b_cqf.p), (uint8_t*)nullptr);
        hipLaunchKernelGGL(cq_rows_kernel, dim3((unsigned)((bext + 4095) / 4096)), dim3(256), 0, st, static_cast<const uint8_t*>(d_pri), bext, static_cast<int32_t*>(b_cqf.p));
        hipLaunchKernelGGL(cq_rows_kernel, dim3((unsigned)((bext + 4095) / 4096)), dim3(256), 0, st, static_cast<const uint8_t*>(d_sd), bext, static_cast<int32_t*>(b_cqf.p));
      }
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpyAsync(&h_cq_flag, b_cqf.p, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    }
    h_hst.resize(nt);
    h_len1.resize(nt);
    h_strim.resize(nt);
    HIP_TRY(hipMemcpyAsync(h_hst.data(), b_hst.p, sizeof(int32_t) * (size_t)nt, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h_len1.data(), b_len1.p, sizeof(uint32_t) * (size_t)nt, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h_strim.data(), d_strim, sizeof(int32_t) * (size_t)nt, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx_sync(ctx));
    h_status.assign(nt, 0);
    for (uint32_t t = 0; t < nt; ++t) {  // indigo.h:303-309
      const double seqsize = (double)mt[t];
      const double thr = seqsize * 0.35 * prm->match + seqsize * (1 - 0.35) * prm->mismatch;
      if ((double)h_strim[t] <= thr) h_status[t] = -1;
    }
    if (ctx->timing) {  // decomposeAlleles launch above: alignment columns walked; rows + basecalls read, basecalls rewritten
      uint64_t wc = 0, wb = 0;
      for (uint32_t t = 0; t < nt; ++t) { wc += h_len1[t]; wb += 2ull * h_len1[t] + 4ull * mf[t]; }
      ctx->acc[TRACYHIP_TIMER_DECOMP].cells += wc;
      ctx->acc[TRACYHIP_TIMER_DECOMP].bytes += wb;
    }
    use_cq = try_cq && (h_cq_flag & 1) == 0;
    cq_codes = (!use_cq || (h_cq_flag & 2)) ? 6 : (h_cq_flag & 4) ? 5 : 4;
    for (uint32_t t = 0; t < nt; ++t)
      if (h_status[t] == 0 && h_hst[t] != 1) h_status[t] = h_hst[t] == 0 ? -2 : -3;

    return TRACYHIP_OK;
  }

  int allele_setup() {
    int rc;
    stage_clock.mark("decompose.6_allele");
    // ---- 6. allele-specific alignments (indigo.h:355-387): string x string Gotoh ----
    // allele k in {0: primary, 1: secDecompose}: gotoh(seq, rs.refslice) -> trimReferenceSlice -> gotoh(seq, slice)
    HIP_TRY(b_ends.ensure(sizeof(uint32_t) * 2 * (size_t)nt));
    HIP_TRY(b_opsA.ensure(tot1 + 2ull * TL * nt + 16));
    HIP_TRY(b_lenA.ensure(sizeof(uint32_t) * (size_t)nt));
    HIP_TRY(b_trimA.ensure(sizeof(TrimOut) * (size_t)nt));
    HIP_TRY(b_rnfw.ensure(sizeof(uint32_t) * (size_t)nt + nt));
    {
      std::vector<uint8_t> tmp(sizeof(uint32_t) * (size_t)nt + nt);
      std::memcpy(tmp.data(), rn.data(), sizeof(uint32_t) * (size_t)nt);
      std::memcpy(tmp.data() + sizeof(uint32_t) * (size_t)nt, h_fwd.data(), nt);
      HIP_TRY(hipMemcpy(b_rnfw.p, tmp.data(), tmp.size(), hipMemcpyHostToDevice));
    }
    offA.resize(nt);
    {
      uint64_t tot = 0;
      for (uint32_t t = 0; t < nt; ++t) { offA[t] = tot; tot += (uint64_t)sl[t] + rn[t]; }
    }
    if ((rc = upload(ctx, buf(), offA, &d_offA))) return rc;
    for (int k = 0; k < 3; ++k) {
      uint64_t cap = 0;
      for (uint32_t t = 0; t < nt; ++t) cap = std::max<uint64_t>(cap, out->ops_offset[k][t] + (uint64_t)sl[t] + (k < 2 ? rn[t] : sl[t]));
      if ((rc = io(out->score[k], sizeof(int32_t) * (size_t)nt, false, &d_scoreK[k]))) return rc;
      if ((rc = io(out->ops[k], cap ? cap : 1, false, &d_opsK[k]))) return rc;
      if ((rc = io(out->ops_len[k], sizeof(uint32_t) * (size_t)nt, false, &d_lenK[k]))) return rc;
    }
    StageClock sc6;
    return TRACYHIP_OK;
  }

  // 6.a: the pairs gotoh(allele k, window), their substitution tables; whether the origin-tracking sweep applies
  int allele_begin(Allele& A, int k) {
    int rc;
    A.seq = (k == 0) ? d_pri : d_sd;
    const void* seq = A.seq;
    sc6.mark("6.a desc+tables");
    DpProblem& pb = A.pb;
    pb.mode = use_cq ? MODE_CQ : MODE_CHAR; pb.d_a1 = seq; pb.d_a2 = use_cq ? static_cast<const void*>(d_cq_ref) : d_ref;
    pb.cq_codes = ctx->knobs.no_compact ? 6 : cq_codes;
    pb.desc.resize(nt); pb.k.resize(nt);
    parallel_for(nt, [&](uint32_t lo, uint32_t hi, uint32_t) {
      for (uint32_t t = lo; t < hi; ++t) {
        PairDesc d{};
        d.a1_off = bc.bc_offset[t] + soff[t];
        d.m = sl[t]; d.a1_stride = sl[t];
        d.a2_off = sr.offset[ridx[t]];
        d.n = rn[t]; d.a2_stride = rn[t];
        d.flags = h_rc[t] ? PAIR_A2_REVCOMP : 0;
        d.out = t;
        pb.desc[t] = d;
        pb.k[t] = choose_k(d.m, MODE_CHAR);
      }
    });
    // Band kernels (band16.h): the score S* of the certifying sweep below bounds the gap steps of every optimal alignment,
    // g = (best m - S*) / |ge|, and with them the diagonals it can visit: the origin-tracking sweep and the traceback against the
    // trimmed slice run on that band only (four pairs per wave, sixteen lanes per pair).  TRACYHIP_NO_BAND16=1: whole matrices.
    const bool b16 = A.b16 = use_cq && p.ge < 0 && p.go <= 0 && !ctx->knobs.no_band16;
    std::vector<B16TableDesc>& td = A.td;
    if (b16) {
      td.resize(nt);
      for (uint32_t t = 0; t < nt; ++t) td[t] = B16TableDesc{bc.bc_offset[t] + soff[t], 0, 0, sl[t], 0, 0};
      HIP_TRY(hipMemsetAsync(ctx->d_err.p, 0, sizeof(int32_t) * kErrWords, st));
      if ((rc = build_b16_tables(ctx, ctx->d_b16tab[k], seq, true, td, &p))) return rc;
      if (k == 0) td_pri = td;
    }
    A.h_s1.assign(nt, 0);
    A.gap_of.assign(nt, -1);
    // gotoh(seq, rs.refslice) is only read by trimReferenceSlice: when the pairs fit its packed fields the origin-tracking
    // sweep delivers the two ends of that alignment without traceback words, walker or ops (TRACYHIP_NO_ORIGIN=1: off)
    bool& use_origin = A.use_origin;
    use_origin = !ctx->knobs.no_origin;
    for (uint32_t t = 0; t < nt && use_origin; ++t) use_origin = origin_ok(&p, pb.desc[t].m, pb.desc[t].n, pb.k[t]);
    if (use_origin) {
      // The origin-tracking sweep is tagged int32 arithmetic (~40 cycles per cell); the plain 16-bit score sweep costs half of
      // that and yields S* = H(m, n) and c_e, which bound where the alignment can lie: a path from (0, lead) to (m, c_e) with
      // score S* has at most g = (best * m - S*) / |ge| horizontal gap columns (best = the largest substitution score, every
      // gap column costs at least |ge|), so lead >= c_e - m - g.  Every optimal path -- and with it every tie the traceback
      // tests, which would be an optimal path too -- lies in the columns (a, c_e], a = c_e - m - g - 2: the origin sweep runs
      // on that sub-window only (a third of a 3 kb window for a 1 kb allele) and its two ends are shifted back by a.
      // TRACYHIP_NO_SUBWINDOW=1: the whole window, as before.
      A.shift.assign(nt, 0);
      A.d_shift = nullptr;
      A.subwin = use_cq && !ctx->knobs.no_subwindow && !ctx->knobs.no_narrow;
      for (uint32_t t = 0; t < nt && A.subwin; ++t) A.subwin = narrow_ok(&p, pb.desc[t].m, pb.k[t]);
    }
    return TRACYHIP_OK;
  }

  // 6.b, 6.c: S* and c_e of gotoh(allele, window) by the pruned sweep (what fails: swept in full), the sub-window they allow
  int allele_locate(Allele& A, int k) {
    int rc;
    const void* seq = A.seq;
    DpProblem& pb = A.pb;
    const bool b16 = A.b16;
    std::vector<B16TableDesc>& td = A.td;
    std::vector<int32_t>& h_s1 = A.h_s1;
    std::vector<int64_t>& gap_of = A.gap_of;
    std::vector<uint32_t>& shift = A.shift;
    uint32_t*& d_shift = A.d_shift;
    bool& subwin = A.subwin;
    (void)seq; (void)b16; (void)td; (void)h_s1; (void)gap_of; (void)shift; (void)d_shift; (void)subwin;
    std::vector<RowEndDesc> hre(nt);
    uint64_t lr_tot = 0;
    for (uint32_t t = 0; t < nt; ++t) { pb.desc[t].lastrow_off = lr_tot; hre[t] = RowEndDesc{lr_tot, rn[t], 0}; lr_tot += (uint64_t)rn[t] + 2; }
    HIP_TRY(ctx->d_lastrow.ensure(lr_tot * 4 + 64));
    DevBuf &b_sw = buf(), &b_zero = buf();
    HIP_TRY(b_sw.ensure(sizeof(int32_t) * (size_t)nt + sizeof(uint32_t) * 2 * (size_t)nt));
    int32_t* d_swscore = static_cast<int32_t*>(b_sw.p);
    uint32_t* d_ce = reinterpret_cast<uint32_t*>(d_swscore + nt);
    d_shift = d_ce + nt;
    if (pb.cq_codes == 4) {  // every column is one of A C G T: an all-clear block map sends every pair to the compact form
      HIP_TRY(b_zero.ensure((er >> 8) + 2));
      HIP_TRY(hipMemsetAsync(b_zero.p, 0, (er >> 8) + 2, st));
      pb.d_special = static_cast<const uint8_t*>(b_zero.p);
    }
    DpCkpt sc;
    sc.B = 0x7fffffffu;  // row m only: no wavefront checkpoints
    sc.narrow = true;
    sc.d_ckpt = static_cast<int32_t*>(ctx->d_lastrow.p);  // (never written)
    sc.d_lastrow = static_cast<int32_t*>(ctx->d_lastrow.p);
    sc6.mark("6.b sweep run_dp");
    // The pruned sweep (front.h), as for the orientation of the trace: rows 1 .. R of the allele over the whole window (row R kept),
    // the rows below them on the diagonals around the best column of row R, and a certificate per pair that no path outside those
    // diagonals reaches the band's score (rest = best x the rows below R: a row of a string scores `match` at most).  Certified
    // pairs have S* and c_e without the window under their first R rows having been swept for the other m - R; the others -- a
    // second copy of the allele's locus in the window, an allele that lost more than the band pays for -- are swept in full.
    std::vector<int8_t> pruned(nt, 0);
    std::vector<int32_t> fscore;
    std::vector<uint32_t> fce;
    if (b16 && !ctx->knobs.no_front) {
      const uint32_t R = kFrontRows;
      const int64_t bestq = std::max<int64_t>(std::max<int64_t>(p.match, p.mismatch), 0);
      // (laid out by a few threads in trace order: eligibility per trace, a scan, the fill)
      std::vector<uint8_t> elig6(nt, 0);
      uint32_t cnt6[kHostThreads] = {};
      parallel_for(nt, [&](uint32_t lo, uint32_t hi, uint32_t tid) {
        uint32_t c = 0;
        for (uint32_t t = lo; t < hi; ++t) {
          const PairDesc& d = pb.desc[t];
          elig6[t] = d.m > R + 2u * (uint32_t)kFrontK && d.n >= 1 && origin16_ok(&p, d.m, d.m - R + 2u * (uint32_t)kFrontHalfW + 16u);
          c += elig6[t];
        }
        cnt6[tid] = c;
      });
      uint32_t at6[kHostThreads + 1] = {};
      for (uint32_t i = 0; i < kHostThreads; ++i) at6[i + 1] = at6[i] + cnt6[i];
      std::vector<PairDesc> pre(at6[kHostThreads]);
      std::vector<FrontDesc> fd(at6[kHostThreads]);
      std::vector<uint32_t> ft(at6[kHostThreads]);
      parallel_for(nt, [&](uint32_t lo, uint32_t hi, uint32_t tid) {
        uint32_t w = at6[tid];
        for (uint32_t t = lo; t < hi; ++t) {
          if (!elig6[t]) continue;
          const PairDesc& d = pb.desc[t];
          PairDesc q = d;
          q.flags |= PAIR_KEEP_ROW;
          pre[w] = q;
          FrontDesc f{};
          f.row_off = d.lastrow_off;
          f.a2_off = d.a2_off;
          f.tab_off = td[t].out_off + R;
          f.tab_stride = td[t].stride;
          f.m_rest = d.m - R;
          f.n = d.n;
          f.flags = d.flags & PAIR_A2_REVCOMP;
          f.out = w;
          f.R = R;
          f.rest = (int32_t)(bestq * (int64_t)(d.m - R));
          fd[w] = f;
          ft[w++] = t;
        }
      });
      if (!fd.empty()) {
        if ((rc = run_prefix_keep_cq(ctx, pb.d_a1, pb.d_a2, pb.d_special, pre, &p, static_cast<int32_t*>(ctx->d_lastrow.p)))) return rc;
        FrontResult fres;
        if ((rc = run_front(ctx, fd, static_cast<const int16_t*>(ctx->d_b16tab[k].p), static_cast<const uint32_t*>(ctx->d_lastrow.p), &p, fres, d_cq_ref, true)))
          return rc;
        fscore.assign(nt, 0);
        fce.assign(nt, 0);
        uint32_t nok = 0;
        for (size_t i = 0; i < ft.size(); ++i)
          if (fres.fo[i].ok && fres.ce[i]) { pruned[ft[i]] = 1; fscore[ft[i]] = fres.score[i]; fce[ft[i]] = fres.ce[i]; ++nok; }
        ctx->stats.allele_pruned[k] += (uint32_t)ft.size(); ctx->stats.allele_uncertified[k] += (uint32_t)ft.size() - nok;
        if (ctx->knobs.verbose) fprintf(stderr, "decompose allele %d: pruned sweep of %zu of %u alleles, %u certified\n", k, ft.size(), nt, nok);
      }
    }
    {
      DpProblem full;  // what is swept in full
      full.mode = pb.mode; full.a1_profile = pb.a1_profile; full.a2_profile = pb.a2_profile; full.d_a1 = pb.d_a1; full.d_a2 = pb.d_a2;
      full.d_a2_chars = pb.d_a2_chars; full.d_special = pb.d_special; full.cq_codes = pb.cq_codes;
      for (uint32_t t = 0; t < nt; ++t)
        if (!pruned[t]) { full.desc.push_back(pb.desc[t]); full.k.push_back(pb.k[t]); }
      rc = full.desc.empty() ? TRACYHIP_OK : run_dp(ctx, full, &p, false, false, d_swscore, nullptr, nullptr, nullptr, DP_CKPT, &sc);
    }
    pb.d_special = nullptr;
    if (rc == kWiden) subwin = false;
    else if (rc) return rc;
    sc6.mark("6.c rowend+subwindow");
    if (subwin) {
      uint32_t npruned = 0;
      for (uint32_t t = 0; t < nt; ++t)
        if (pruned[t]) { hre[t].n = 0; ++npruned; }  // (row m of a pruned pair was never written: its c_e is the band's)
      std::vector<int32_t> h_s(nt);
      std::vector<uint32_t> h_ce(nt);
      if (npruned < nt) {  // (every allele pruned -- the usual case: nothing to read off row m, no round trip)
        const RowEndDesc* d_re;
        if ((rc = upload(ctx, buf(), hre, &d_re))) return rc;
        hipLaunchKernelGGL(row_m_end_kernel, dim3(nt), dim3(64), 0, st, d_re, static_cast<const int32_t*>(ctx->d_lastrow.p), p.go + p.ge, d_ce);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h_s.data(), d_swscore, sizeof(int32_t) * (size_t)nt, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(h_ce.data(), d_ce, sizeof(uint32_t) * (size_t)nt, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx_sync(ctx));
      }
      if (!fscore.empty())
        for (uint32_t t = 0; t < nt; ++t)
          if (pruned[t]) { h_s[t] = fscore[t]; h_ce[t] = fce[t]; }
      const int64_t best = std::max<int64_t>(std::max<int64_t>(p.match, p.mismatch), 0), age = -(int64_t)p.ge;
      parallel_for(nt, [&](uint32_t lo, uint32_t hi, uint32_t) {
        for (uint32_t t = lo; t < hi; ++t) {
          PairDesc& d = pb.desc[t];
          const int64_t ce = h_ce[t];
          if (d.m == 0 || d.n == 0) continue;
          if (ce <= 0) {  // H(m, c) == E(m, c) everywhere: n 'h' then m 'v', both ends 0 -- column 1 alone reproduces it (see orient_and_align_impl)
            d.a2_off += h_rc[t] ? (uint64_t)(d.n - 1u) : 0ull;
            d.n = 1;
            d.a2_stride = 1;
            continue;
          }
          const int64_t loss = best * (int64_t)d.m - (int64_t)h_s[t];
          const int64_t g = loss > 0 ? loss / age : 0;
          int64_t a = ce - (int64_t)d.m - g - 2;
          if (a < 0) a = 0;
          shift[t] = (uint32_t)a;
          d.a2_off += h_rc[t] ? (uint64_t)(d.n - (uint32_t)ce) : (uint64_t)a;  // reverse view: column c is byte n - c
          d.n = (uint32_t)(ce - a);
          d.a2_stride = d.n;
          h_s1[t] = h_s[t];
          gap_of[t] = g;
        }
      });
      HIP_TRY(hipMemcpyAsync(d_shift, shift.data(), sizeof(uint32_t) * (size_t)nt, hipMemcpyHostToDevice, st));
    }
    return TRACYHIP_OK;
  }

  // 6.d: the two ends of gotoh(allele, window) by the origin-tracking sweep (on its band where that fits), trimReferenceSlice
  int allele_origin(Allele& A, int k) {
    int rc;
    const void* seq = A.seq;
    DpProblem& pb = A.pb;
    const bool b16 = A.b16;
    std::vector<B16TableDesc>& td = A.td;
    std::vector<int32_t>& h_s1 = A.h_s1;
    std::vector<int64_t>& gap_of = A.gap_of;
    std::vector<uint32_t>& shift = A.shift;
    uint32_t*& d_shift = A.d_shift;
    bool& subwin = A.subwin;
    (void)seq; (void)b16; (void)td; (void)h_s1; (void)gap_of; (void)shift; (void)d_shift; (void)subwin;
    sc6.mark("6.d origin band plan+launch");
    DpCkpt oc;
    oc.d_ends = static_cast<uint32_t*>(b_ends.p);
    // the alignment ends in the last column of its sub-window with at most g gap steps behind it: diagonals n' - m - g .. n' - m + g
    Band16Job jo;
    Band16Lease<Band16Job> jo_lease(ctx, jo);
    DpProblem rest;
    if (b16) {
      jo.kind = 1; jo.d_qp = static_cast<const int16_t*>(ctx->d_b16tab[k].p); jo.d_codes = d_cq_ref;
      rest.mode = pb.mode; rest.d_a1 = pb.d_a1; rest.d_a2 = pb.d_a2; rest.cq_codes = pb.cq_codes;
      jo.desc.resize(nt);
      jo.k.assign(nt, 0);
      parallel_for(nt, [&](uint32_t lo, uint32_t hi, uint32_t) {
        for (uint32_t t = lo; t < hi; ++t) {
          PairDesc d = pb.desc[t];
          const int64_t g = gap_of[t];
          const int32_t d1 = (int32_t)d.n - (int32_t)d.m;
          const int32_t dlo = d1 - (int32_t)std::min<int64_t>(g, 1 << 20) - 1, dhi = d1 + (int32_t)std::min<int64_t>(g, 1 << 20) + 1;
          const int K = g >= 0 ? band16_pick_k(dlo, dhi) : 0;
          if (K && origin16_ok(&p, d.m, d.n)) {
            d.a1_off = td[t].out_off; d.a1_stride = td[t].stride; d.ckpt_off = band_pack(dlo, dhi); d.lastrow_off = 0;
            jo.desc[t] = d;
            jo.k[t] = K;
          }
        }
      });
      for (uint32_t t = 0; t < nt; ++t)
        if (jo.k[t] == 0) { rest.desc.push_back(pb.desc[t]); rest.k.push_back(pb.k[t]); }
      if ((rc = run_band16(ctx, jo, &p, nullptr, oc.d_ends, nullptr, nullptr, nullptr))) return rc;
    }
    if ((rc = run_dp(ctx, b16 ? rest : pb, &p, false, false, nullptr, nullptr, nullptr, nullptr, DP_ORIGIN, &oc))) return rc;
    if (subwin) {
      hipLaunchKernelGGL(ends_shift_kernel, dim3((nt + 255) / 256), dim3(256), 0, st, static_cast<uint32_t*>(b_ends.p),
                         static_cast<const uint32_t*>(d_shift), nt);
    }
    hipLaunchKernelGGL(trim_from_ends_kernel, dim3((nt + 255) / 256), dim3(256), 0, st, static_cast<const uint32_t*>(b_ends.p),
                       static_cast<const uint32_t*>(b_rnfw.p), reinterpret_cast<const uint8_t*>(static_cast<const uint32_t*>(b_rnfw.p) + nt),
                       TL, TR, nt, static_cast<TrimOut*>(b_trimA.p));
    return TRACYHIP_OK;
  }

  // (no origin-tracking sweep for these pairs: the whole-matrix traceback, trimReferenceSlice on its string)
  int allele_plain(Allele& A, int k) {
    int rc;
    const void* seq = A.seq;
    DpProblem& pb = A.pb;
    const bool b16 = A.b16;
    std::vector<B16TableDesc>& td = A.td;
    std::vector<int32_t>& h_s1 = A.h_s1;
    std::vector<int64_t>& gap_of = A.gap_of;
    std::vector<uint32_t>& shift = A.shift;
    uint32_t*& d_shift = A.d_shift;
    bool& subwin = A.subwin;
    (void)seq; (void)b16; (void)td; (void)h_s1; (void)gap_of; (void)shift; (void)d_shift; (void)subwin;
    if ((rc = run_dp(ctx, pb, &p, false, true, nullptr, static_cast<uint8_t*>(b_opsA.p), d_offA, static_cast<uint32_t*>(b_lenA.p)))) return rc;
    hipLaunchKernelGGL(trim_kernel, dim3(nt), dim3(64), 0, st, static_cast<const uint8_t*>(b_opsA.p), d_offA,
                       static_cast<const uint32_t*>(b_lenA.p), static_cast<const uint32_t*>(b_rnfw.p),
                       reinterpret_cast<const uint8_t*>(static_cast<const uint32_t*>(b_rnfw.p) + nt), TL, TR, nt,
                       static_cast<TrimOut*>(b_trimA.p));
    return TRACYHIP_OK;
  }

  // 6.e - 6.h: gotoh(allele, trimmed slice) (indigo.h:365) on the band around its known end
  int allele_slice(Allele& A, int k) {
    int rc;
    const void* seq = A.seq;
    DpProblem& pb = A.pb;
    const bool b16 = A.b16;
    std::vector<B16TableDesc>& td = A.td;
    std::vector<int32_t>& h_s1 = A.h_s1;
    std::vector<int64_t>& gap_of = A.gap_of;
    std::vector<uint32_t>& shift = A.shift;
    uint32_t*& d_shift = A.d_shift;
    bool& subwin = A.subwin;
    (void)seq; (void)b16; (void)td; (void)h_s1; (void)gap_of; (void)shift; (void)d_shift; (void)subwin;
    sc6.mark("6.e trim readback (waits for origin)");
    HIP_TRY(hipGetLastError());
    h_trimA[k].resize(nt);
    std::vector<uint32_t> h_ends;
    HIP_TRY(hipMemcpyAsync(h_trimA[k].data(), b_trimA.p, sizeof(TrimOut) * (size_t)nt, hipMemcpyDeviceToHost, st));
    if (b16 && A.use_origin) {
      h_ends.resize(2 * (size_t)nt);
      HIP_TRY(hipMemcpyAsync(h_ends.data(), b_ends.p, sizeof(uint32_t) * 2 * (size_t)nt, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(ctx_sync(ctx));
    sc6.mark("6.f slice descs+plan");
    for (uint32_t t = 0; t < nt; ++t) {
      PairDesc& d = pb.desc[t];
      d.n = h_trimA[k][t].len; d.a2_stride = d.n;
      d.a2_off = sr.offset[ridx[t]] + (h_rc[t] ? rn[t] - h_trimA[k][t].ri - h_trimA[k][t].len : h_trimA[k][t].ri);
    }
    const uint64_t* d_offK;
    std::vector<uint64_t> offK(out->ops_offset[k], out->ops_offset[k] + nt);
    if ((rc = upload(ctx, buf(), offK, &d_offK))) return rc;
    // gotoh(seq, trimmed slice) (indigo.h:365): the slice holds the alignment just located, so its score is S* again, it ends in
    // column c_e - slice_begin of row m and every optimal path stays within g gap steps of that diagonal.  Banded pairs are
    // checked against S* afterwards (a walk that left its band reports no ops); what fails goes to the whole matrix with the rest.
    Band16Job jt;
    Band16Lease<Band16Job> jt_lease(ctx, jt);
    DpProblem rest;
    rest.mode = pb.mode; rest.d_a1 = pb.d_a1; rest.d_a2 = pb.d_a2; rest.cq_codes = pb.cq_codes;
    if (!h_ends.empty()) {
      jt.kind = 0; jt.d_qp = static_cast<const int16_t*>(ctx->d_b16tab[k].p); jt.d_codes = d_cq_ref;
      jt.desc.resize(nt);
      jt.k.assign(nt, 0);
      parallel_for(nt, [&](uint32_t lo, uint32_t hi, uint32_t) {
        for (uint32_t t = lo; t < hi; ++t) {
          PairDesc d = pb.desc[t];
          const int64_t g = gap_of[t];
          const int64_t ce = (int64_t)h_ends[2 * t + 1] - (int64_t)h_trimA[k][t].ri;  // last column of the alignment, in the slice
          int K = 0;
          int32_t dlo = 0, dhi = 0;
          if (g >= 0 && d.m && d.n && ce >= 1 && ce <= (int64_t)d.n) {
            const int32_t d1 = (int32_t)ce - (int32_t)d.m;
            dlo = d1 - (int32_t)std::min<int64_t>(g, 1 << 20) - 1;
            dhi = d1 + (int32_t)std::min<int64_t>(g, 1 << 20) + 1;
            K = band16_pick_k(dlo, dhi);
          }
          if (K) {
            d.a1_off = td[t].out_off; d.a1_stride = td[t].stride; d.ckpt_off = band_pack(dlo, dhi); d.lastrow_off = 0;
            jt.desc[t] = d;
            jt.k[t] = K;
          }
        }
      });
      for (uint32_t t = 0; t < nt; ++t)
        if (jt.k[t] == 0) { rest.desc.push_back(pb.desc[t]); rest.k.push_back(pb.k[t]); }
      const size_t nb16 = nt - rest.desc.size();
      sc6.mark("6.g run_band16 traceback");
      if ((rc = run_band16(ctx, jt, &p, static_cast<int32_t*>(d_scoreK[k]), nullptr, static_cast<uint8_t*>(d_opsK[k]), d_offK, static_cast<uint32_t*>(d_lenK[k])))) return rc;
      sc6.mark("6.h check readback (waits for traceback)");
      if (nb16) {
        std::vector<int32_t> h_sc(nt);
        std::vector<uint32_t> h_ol(nt);
        HIP_TRY(hipMemcpyAsync(h_sc.data(), d_scoreK[k], sizeof(int32_t) * (size_t)nt, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(h_ol.data(), d_lenK[k], sizeof(uint32_t) * (size_t)nt, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx_sync(ctx));
        uint32_t nfail = 0;
        for (uint32_t t = 0; t < nt; ++t)
          if (jt.k[t] && (h_sc[t] != h_s1[t] || h_ol[t] == 0)) {
            if (ctx->knobs.verbose && nfail < 6)
              fprintf(stderr, "  fail t=%u m=%u n=%u S1=%d got=%d len=%u g=%lld ends=(%u,%u) ri=%u rc=%d\n", t, pb.desc[t].m, pb.desc[t].n, h_s1[t], h_sc[t], h_ol[t],
                      (long long)gap_of[t], h_ends[2 * t], h_ends[2 * t + 1], h_trimA[k][t].ri, (int)h_rc[t]);
            rest.desc.push_back(pb.desc[t]); rest.k.push_back(pb.k[t]); ++nfail;
          }
        ctx->stats.allele_banded[k] += (uint32_t)nb16; ctx->stats.allele_repeated[k] += nfail;
        if (ctx->knobs.verbose) fprintf(stderr, "decompose allele %d: %zu of %u slices banded, %u repeated\n", k, nb16, nt, nfail);
      }
    }
    if ((rc = run_dp(ctx, h_ends.empty() ? pb : rest, &p, false, true, static_cast<int32_t*>(d_scoreK[k]), static_cast<uint8_t*>(d_opsK[k]), d_offK,
                     static_cast<uint32_t*>(d_lenK[k]))))
      return rc;
    return TRACYHIP_OK;
  }

  // allele 1 vs allele 2, global (indigo.h:379-387)
  int allele12() {
    int rc;
    sc6.mark("6.i allele1v2 setup");
    {  // allele 1 vs allele 2, global (indigo.h:379-387)
      DpProblem pb;
      DpProblemLease lease(ctx, pb);
      pb.mode = use_cq ? MODE_CQ : MODE_CHAR; pb.d_a1 = d_pri; pb.d_a2 = use_cq ? static_cast<const void*>(d_cq_sd) : d_sd;
      pb.desc.resize(nt); pb.k.resize(nt);
      parallel_for(nt, [&](uint32_t lo, uint32_t hi, uint32_t) {
        for (uint32_t t = lo; t < hi; ++t) {
          PairDesc d{};
          d.a1_off = bc.bc_offset[t] + soff[t];
          d.a2_off = bc.bc_offset[t] + soff[t];
          d.m = sl[t]; d.n = sl[t]; d.a1_stride = sl[t]; d.a2_stride = sl[t];
          d.out = t;
          pb.desc[t] = d;
          pb.k[t] = choose_k(d.m, MODE_CHAR);
        }
      });
      const uint64_t* d_offK;
      std::vector<uint64_t> offK(out->ops_offset[2], out->ops_offset[2] + nt);
      if ((rc = upload(ctx, buf(), offK, &d_offK))) return rc;
      // On a band (band16.h) where it can be certified afterwards.  Both ends are fixed here: a path that leaves the diagonals
      // [-W - (m-n)+, W + (n-m)+] makes at least v = W + 1 + (m-n)+ vertical and h = W + 1 + (n-m)+ horizontal gap steps in two runs, so
      // it scores at most best (m - v) - |ge| (v + h) - 2 |go|; a banded score above that is the optimum and bits and path are the
      // whole matrix's.  W is guessed from what the two alleles lost against the reference (they differ from each other by about
      // as much as both differ from it); pairs that do not certify are repeated on the whole matrix.
      Band16Job jg;
      Band16Lease<Band16Job> jg_lease(ctx, jg);
      DpProblem rest;
      rest.mode = pb.mode; rest.d_a1 = pb.d_a1; rest.d_a2 = pb.d_a2; rest.cq_codes = pb.cq_codes;
      const bool b16g = use_cq && !td_pri.empty() && pglobal.ge < 0 && pglobal.go <= 0 && !ctx->knobs.no_band16;
      std::vector<int64_t> bound_of(nt, 0);
      if (b16g) {
        std::vector<int32_t> h_a[2] = {std::vector<int32_t>(nt), std::vector<int32_t>(nt)};
        for (int k = 0; k < 2; ++k) HIP_TRY(hipMemcpyAsync(h_a[k].data(), d_scoreK[k], sizeof(int32_t) * (size_t)nt, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx_sync(ctx));
        jg.kind = 0; jg.d_qp = static_cast<const int16_t*>(ctx->d_b16tab[0].p); jg.d_codes = d_cq_sd;
        const int64_t best = std::max<int64_t>(std::max<int64_t>(pglobal.match, pglobal.mismatch), 0), age = -(int64_t)pglobal.ge, ago = -(int64_t)pglobal.go;
        jg.desc.resize(nt);
        jg.k.assign(nt, 0);
        parallel_for(nt, [&](uint32_t lo_, uint32_t hi_, uint32_t) {
         for (uint32_t t = lo_; t < hi_; ++t) {
          PairDesc d = pb.desc[t];
          int K = 0;
          int32_t dlo = 0, dhi = 0;
          if (d.m && d.n) {
            const int64_t lost = std::max<int64_t>(0, best * d.m - h_a[0][t]) + std::max<int64_t>(0, best * d.m - h_a[1][t]);
            const int64_t per = best + 2 * age;
            int64_t W = (5 * lost / 2 + 40) / (per > 0 ? per : 1) + 2;
            const int64_t over = (int64_t)d.n - (int64_t)d.m;
            if (W > 90) W = 90;
            dlo = (int32_t)(-W - (over < 0 ? -over : 0));
            dhi = (int32_t)(W + (over > 0 ? over : 0));
            K = band16_pick_k(dlo, dhi);
            const int64_t v = W + 1 + (over < 0 ? -over : 0), h = W + 1 + (over > 0 ? over : 0);
            bound_of[t] = best * ((int64_t)d.m - v) - age * (v + h) - 2 * ago;
          }
          if (K) {
            d.a1_off = td_pri[t].out_off; d.a1_stride = td_pri[t].stride; d.ckpt_off = band_pack(dlo, dhi); d.lastrow_off = 0;
            jg.desc[t] = d;
            jg.k[t] = K;
          }
         }
        });
        for (uint32_t t = 0; t < nt; ++t)
          if (jg.k[t] == 0) { rest.desc.push_back(pb.desc[t]); rest.k.push_back(pb.k[t]); }
        const size_t nb16 = nt - rest.desc.size();
        HIP_TRY(hipMemsetAsync(ctx->d_err.p, 0, sizeof(int32_t) * kErrWords, st));
        sc6.mark("6.j allele1v2 run_band16");
        if ((rc = run_band16(ctx, jg, &pglobal, static_cast<int32_t*>(d_scoreK[2]), nullptr, static_cast<uint8_t*>(d_opsK[2]), d_offK, static_cast<uint32_t*>(d_lenK[2])))) return rc;
        if (nb16) {
          std::vector<int32_t> h_sc(nt);
          std::vector<uint32_t> h_ol(nt);
          HIP_TRY(hipMemcpyAsync(h_sc.data(), d_scoreK[2], sizeof(int32_t) * (size_t)nt, hipMemcpyDeviceToHost, st));
          HIP_TRY(hipMemcpyAsync(h_ol.data(), d_lenK[2], sizeof(uint32_t) * (size_t)nt, hipMemcpyDeviceToHost, st));
          HIP_TRY(ctx_sync(ctx));
          uint32_t nfail = 0;
          for (uint32_t t = 0; t < nt; ++t)
            if (jg.k[t] && ((int64_t)h_sc[t] <= bound_of[t] || h_ol[t] == 0)) { rest.desc.push_back(pb.desc[t]); rest.k.push_back(pb.k[t]); ++nfail; }
          ctx->stats.allele_banded[2] += (uint32_t)nb16; ctx->stats.allele_repeated[2] += nfail;
          if (ctx->knobs.verbose) fprintf(stderr, "decompose allele 1 vs 2: %zu of %u pairs banded, %u repeated\n", nb16, nt, nfail);
        }
      }
      if ((rc = run_dp(ctx, b16g ? rest : pb, &pglobal, false, true, static_cast<int32_t*>(d_scoreK[2]), static_cast<uint8_t*>(d_opsK[2]), d_offK,
                       static_cast<uint32_t*>(d_lenK[2]))))
        return rc;
    }

    return TRACYHIP_OK;
  }

  int results() {
    sc6.mark("6.k results");
    // ---- results ----
    const hipMemcpyKind up = (mem == TRACYHIP_MEM_HOST) ? hipMemcpyHostToHost : hipMemcpyHostToDevice;
    std::vector<uint32_t> hb[2], hl[2], hp[2];
    for (int k = 0; k < 2; ++k) {
      hb[k].resize(nt); hl[k].resize(nt); hp[k].resize(nt);
      for (uint32_t t = 0; t < nt; ++t) { hb[k][t] = h_trimA[k][t].ri; hl[k][t] = h_trimA[k][t].len; hp[k][t] = h_trimA[k][t].pos; }
      HIP_TRY(hipMemcpyAsync(out->slice_begin[k], hb[k].data(), sizeof(uint32_t) * (size_t)nt, up, st));
      HIP_TRY(hipMemcpyAsync(out->slice_len[k], hl[k].data(), sizeof(uint32_t) * (size_t)nt, up, st));
      HIP_TRY(hipMemcpyAsync(out->ref_pos[k], hp[k].data(), sizeof(uint32_t) * (size_t)nt, up, st));
    }
    HIP_TRY(hipMemcpyAsync(out->score_fwd, h_sc2.data(), sizeof(int32_t) * (size_t)nt, up, st));
    HIP_TRY(hipMemcpyAsync(out->score_rev, h_sc2.data() + nt, sizeof(int32_t) * (size_t)nt, up, st));
    HIP_TRY(hipMemcpyAsync(out->forward, h_fwd.data(), nt, up, st));
    HIP_TRY(hipMemcpyAsync(out->status, h_status.data(), sizeof(int32_t) * (size_t)nt, up, st));
    for (const DevOut& o : outs)
      if (o.bytes) HIP_TRY(hipMemcpyAsync(o.user, o.dev, o.bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx_sync(ctx));
    timing_collect(ctx);
    return TRACYHIP_OK;
  }
};

int tracyhip::decompose_traces_legacy(tracyhip_ctx* ctx, const tracyhip_decompose_job* job, const tracyhip_params* prm, int mem,
                                      const tracyhip_decompose_result* out) {
  TRACYHIP_HOST_SCOPE(hs_call, "decompose_traces");
  bool empty = false;
  int rc = decompose_check_args(ctx, job, prm, mem, out, &empty);
  if (rc || empty) return rc;
  DecomposeRun r(ctx, job, prm, mem, out);
  if ((rc = r.setup())) return rc;
  if ((rc = r.orientation())) return rc;         // 2. orientation (indigo.h:235-247)
  if ((rc = r.gotoh_rows())) return rc;          // 3. gotoh(trimmed trace, window) + alignment rows (indigo.h:302)
  if ((rc = r.deconvolution())) return rc;       // 1., 4., 5. findBreakpoint, findHomozygousBreakpoint, decomposeAlleles ... allelicFraction
  if ((rc = r.read_back_verdicts())) return rc;
  if ((rc = r.allele_setup())) return rc;        // 6. allele-specific alignments (indigo.h:355-387)
  for (int k = 0; k < 2; ++k) {
    DecomposeRun::Allele A(ctx);
    if ((rc = r.allele_begin(A, k))) return rc;
    if (A.use_origin) {
      if (A.subwin && (rc = r.allele_locate(A, k))) return rc;
      if ((rc = r.allele_origin(A, k))) return rc;
    } else if ((rc = r.allele_plain(A, k))) return rc;
    if ((rc = r.allele_slice(A, k))) return rc;
  }
  if ((rc = r.allele12())) return rc;
  return r.results();
}

namespace {
bool decompose_splittable(const tracyhip_decompose_job* job, const tracyhip_decompose_result* out, const tracyhip_params* prm, uint32_t parts) {
  bool split = parts >= 2 && job && out && prm && job->ntraces >= parts * kMinLaneChunk && job->profiles.offset && job->profiles.length &&
               job->refs.offset && job->refs.length && job->profiles.count >= job->ntraces && job->bc.ntraces >= job->ntraces &&
               (job->bc.peaks || (job->bc.signal_offset && job->bc.nsamples)) && job->bc.bc_offset && job->bc.bc_len && out->dcp_offset &&
               (job->ref_index || job->refs.count >= job->ntraces);
  if (split) {
    const uint32_t nt = job->ntraces;
    split = (job->bc.peaks || nondecreasing(job->bc.signal_offset, nt)) && nondecreasing(job->bc.bc_offset, nt) && nondecreasing(out->dcp_offset, nt);
    for (int k = 0; k < 3 && split; ++k) split = out->ops_offset[k] && nondecreasing(out->ops_offset[k], nt);
    if (job->ref_profiles.data && !job->ref_index)
      split = split && job->ref_profiles.offset && job->ref_profiles.length && job->ref_profiles.count >= nt;
  }
  return split;
}
struct DecomposeChunk {
  tracyhip_decompose_job j;
  tracyhip_decompose_result o;
  SubSet sp, sr, srp;
  SubOffsets sig, bco, dcp, ops[3];
  DecomposeChunk(const tracyhip_decompose_job* job, const tracyhip_decompose_result* out, uint32_t lo, uint32_t k) : j(*job), o(*out) {
    j.ntraces = k;
    sub_seqset(job->profiles, lo, k, sizeof(float), sp);
    j.profiles = sp.s;
    if (job->ref_index) j.ref_index = job->ref_index + lo;
    else {
      sub_seqset(job->refs, lo, k, 1, sr);
      j.refs = sr.s;
      if (job->ref_profiles.data) { sub_seqset(job->ref_profiles, lo, k, sizeof(float), srp); j.ref_profiles = srp.s; }
    }
    j.oriented = shifted(job->oriented, lo);
    // Trace + BaseCalls: signal by signal_offset, the per-base arrays by bc_offset
    sub_offsets(job->bc.bc_offset, lo, k, bco);
    j.bc.ntraces = k;
    if (job->bc.peaks) { j.bc.peaks = shifted(job->bc.peaks, 4 * bco.base); j.bc.signal = nullptr; j.bc.signal_offset = nullptr; j.bc.nsamples = nullptr; j.bc.bcpos = nullptr; }
    else {
      sub_offsets(job->bc.signal_offset, lo, k, sig);
      j.bc.signal = shifted(job->bc.signal, sig.base); j.bc.signal_offset = sig.off.data(); j.bc.nsamples = job->bc.nsamples + lo;
      j.bc.bcpos = shifted(job->bc.bcpos, bco.base);
    }
    j.bc.primary = shifted(job->bc.primary, bco.base);
    j.bc.secondary = shifted(job->bc.secondary, bco.base); j.bc.bc_offset = bco.off.data(); j.bc.bc_len = job->bc.bc_len + lo;
    // results
    o.bp = shifted(out->bp, lo); o.status = shifted(out->status, lo); o.score_fwd = shifted(out->score_fwd, lo);
    o.score_rev = shifted(out->score_rev, lo); o.forward = shifted(out->forward, lo); o.score_trim = shifted(out->score_trim, lo);
    sub_offsets(out->dcp_offset, lo, k, dcp);
    o.dcp_indel = shifted(out->dcp_indel, dcp.base); o.dcp_err = shifted(out->dcp_err, dcp.base); o.dcp_offset = dcp.off.data();
    o.dstatus = shifted(out->dstatus, lo); o.secdecomp = shifted(out->secdecomp, bco.base); o.fractions = shifted(out->fractions, 2ull * lo);
    for (int a = 0; a < 2; ++a) {
      o.slice_begin[a] = shifted(out->slice_begin[a], lo); o.slice_len[a] = shifted(out->slice_len[a], lo); o.ref_pos[a] = shifted(out->ref_pos[a], lo);
    }
    for (int a = 0; a < 3; ++a) {
      sub_offsets(out->ops_offset[a], lo, k, ops[a]);
      o.score[a] = shifted(out->score[a], lo); o.ops[a] = shifted(out->ops[a], ops[a].base);
      o.ops_offset[a] = ops[a].off.data(); o.ops_len[a] = shifted(out->ops_len[a], lo);
    }
  }
};
}  // namespace

// lanes (tracyhip_set_lanes): contiguous chunks of the batch in flight, as tracyhip_align_traces
extern "C" int tracyhip_decompose_traces(tracyhip_ctx* ctx, const tracyhip_decompose_job* job, const tracyhip_params* prm, int mem,
                                         const tracyhip_decompose_result* out) {
  if (!ctx) return set_error(TRACYHIP_ERR_ARG, "null context");
  reset_call_stats(ctx, job ? job->ntraces : 0);
  const uint32_t L = (uint32_t)ctx->lanes.size() + 1;
  if (!decompose_splittable(job, out, prm, L)) return decompose_traces_one(ctx, job, prm, mem, out);
  int rc = ctx_begin(ctx);
  if (rc) return rc;
  return run_lanes(ctx, job->ntraces, [&](tracyhip_ctx* lane, uint32_t, uint32_t lo, uint32_t k) -> int {
    if (k == 0) return TRACYHIP_OK;
    if (lane != ctx) lane->stats.traces = k;
    DecomposeChunk c(job, out, lo, k);
    return decompose_traces_one(lane, &c.j, prm, mem, &c.o);
  });
}

// =====================================================================================================
// Device groups: one context per GPU of the node, one host thread per context.  A batch call on a group cuts the batch
// into contiguous blocks of traces (the pair list of a DP call into slices of equal cell count), runs every block
// through the single-device entry point of its context -- which may split it further over its lanes -- and returns when
// all are complete.  Host buffers only: every block stages its own part through its device, nothing crosses devices
// (the path shards embarrassingly; a multi-process job uses one context per rank and RCCL for the gather
// instead, tracy_amd/shard.py).
// =====================================================================================================
struct tracyhip_group {
  std::vector<tracyhip_ctx*> ctx;
};

extern "C" {

int tracyhip_group_create(const int* devices, int ndevices, tracyhip_group** out) {
  if (!out) return set_error(TRACYHIP_ERR_ARG, "null out pointer");
  *out = nullptr;
  std::vector<int> devs;
  if (devices) {
    if (ndevices < 1) return set_error(TRACYHIP_ERR_ARG, "empty device list");
    devs.assign(devices, devices + ndevices);
  } else {  // every visible device
    int n = 0;
    int rc = tracyhip_device_count(&n);
    if (rc) return rc;
    if (n <= 0) return set_error(TRACYHIP_ERR_NODEVICE, "no HIP device visible (this library has no CPU fallback)");
    if (ndevices > 0 && ndevices < n) n = ndevices;
    for (int i = 0; i < n; ++i) devs.push_back(i);
  }
  tracyhip_group* g = new tracyhip_group();
  for (int d : devs) {
    tracyhip_ctx* c = nullptr;
    const int rc = tracyhip_create(d, &c);
    if (rc) {
      const std::string msg = tracyhip_last_error();
      for (auto* x : g->ctx) tracyhip_destroy(x);
      delete g;
      return set_error(rc, "%s", msg.c_str());
    }
    g->ctx.push_back(c);
  }
  *out = g;
  return TRACYHIP_OK;
}

int tracyhip_group_destroy(tracyhip_group* g) {
  if (!g) return TRACYHIP_OK;
  for (auto* c : g->ctx) tracyhip_destroy(c);
  delete g;
  return TRACYHIP_OK;
}

int tracyhip_group_size(const tracyhip_group* g) { return g ? (int)g->ctx.size() : 0; }

tracyhip_ctx* tracyhip_group_context(tracyhip_group* g, int i) {
  return (g && i >= 0 && i < (int)g->ctx.size()) ? g->ctx[i] : nullptr;
}

int tracyhip_group_set_lanes(tracyhip_group* g, uint32_t lanes) {
  if (!g) return set_error(TRACYHIP_ERR_ARG, "null group");
  for (auto* c : g->ctx) { const int rc = tracyhip_set_lanes(c, lanes); if (rc) return rc; }
  return TRACYHIP_OK;
}

int tracyhip_group_align_traces(tracyhip_group* g, const tracyhip_align_job* job, const tracyhip_params* prm, const tracyhip_align_result* out) {
  if (!g || g->ctx.empty()) return set_error(TRACYHIP_ERR_ARG, "null / empty group");
  const uint32_t D = (uint32_t)g->ctx.size();
  if (D == 1 || !align_splittable(job, out, prm, D)) return tracyhip_align_traces(g->ctx[0], job, prm, TRACYHIP_MEM_HOST, out);
  return run_chunks(g->ctx, job->ntraces, [&](tracyhip_ctx* c, uint32_t, uint32_t lo, uint32_t k) -> int {
    if (k == 0) return TRACYHIP_OK;
    AlignChunk ch(job, out, lo, k);
    return tracyhip_align_traces(c, &ch.j, prm, TRACYHIP_MEM_HOST, &ch.o);
  });
}

int tracyhip_group_decompose_traces(tracyhip_group* g, const tracyhip_decompose_job* job, const tracyhip_params* prm,
                                    const tracyhip_decompose_result* out) {
  if (!g || g->ctx.empty()) return set_error(TRACYHIP_ERR_ARG, "null / empty group");
  const uint32_t D = (uint32_t)g->ctx.size();
  if (D == 1 || !decompose_splittable(job, out, prm, D)) return tracyhip_decompose_traces(g->ctx[0], job, prm, TRACYHIP_MEM_HOST, out);
  return run_chunks(g->ctx, job->ntraces, [&](tracyhip_ctx* c, uint32_t, uint32_t lo, uint32_t k) -> int {
    if (k == 0) return TRACYHIP_OK;
    DecomposeChunk ch(job, out, lo, k);
    return tracyhip_decompose_traces(c, &ch.j, prm, TRACYHIP_MEM_HOST, &ch.o);
  });
}

// Cut a pair list into `parts` contiguous slices of (nearly) equal DP cell count m * n: bounds[0 .. parts], bounds[parts] = npairs.
// Host arithmetic only (no device needed): the same rule shards the all-pairs matrix of msa.h:33-42 over the devices of a
// group and over the ranks of a multi-process job (tracy_amd/shard.py).
int tracyhip_pair_bounds(const tracyhip_pairs* pairs, uint32_t parts, uint64_t* bounds) {
  if (!pairs || !bounds || parts < 1) return set_error(TRACYHIP_ERR_ARG, "null pairs / bounds, or zero parts");
  const uint32_t np = pairs->npairs;
  if (np && (!pairs->a1.length || !pairs->a2.length)) return set_error(TRACYHIP_ERR_ARG, "null length arrays");
  std::vector<uint64_t> cum(np);
  uint64_t tot = 0;
  for (uint32_t i = 0; i < np; ++i) {
    const uint32_t i1 = pairs->a1_index ? pairs->a1_index[i] : i, i2 = pairs->a2_index ? pairs->a2_index[i] : i;
    if (i1 >= pairs->a1.count || i2 >= pairs->a2.count) return set_error(TRACYHIP_ERR_ARG, "pair %u indexes past the sequence sets", i);
    tot += (uint64_t)pairs->a1.length[i1] * pairs->a2.length[i2];
    cum[i] = tot;
  }
  bounds[0] = 0;
  for (uint32_t r = 1; r < parts; ++r) {
    const uint64_t want = tot / parts * r + (tot % parts) * r / parts;  // floor(tot * r / parts) without overflow
    bounds[r] = (uint64_t)(std::lower_bound(cum.begin(), cum.end(), want) - cum.begin());
    if (bounds[r] < bounds[r - 1]) bounds[r] = bounds[r - 1];
  }
  bounds[parts] = np;
  return TRACYHIP_OK;
}

int tracyhip_group_gotoh_score(tracyhip_group* g, const tracyhip_pairs* pairs, const tracyhip_params* prm, int32_t* scores) {
  if (!g || g->ctx.empty()) return set_error(TRACYHIP_ERR_ARG, "null / empty group");
  if (!pairs) return set_error(TRACYHIP_ERR_ARG, "null pairs");
  const uint32_t D = (uint32_t)g->ctx.size();
  if (D == 1 || pairs->npairs < D * kMinLaneChunk) return tracyhip_gotoh_score(g->ctx[0], pairs, prm, TRACYHIP_MEM_HOST, scores);
  std::vector<uint64_t> b(D + 1);
  int rc = tracyhip_pair_bounds(pairs, D, b.data());
  if (rc) return rc;
  // identity index arrays are made explicit so that a slice can start anywhere; the sequence sets travel whole (replicated)
  std::vector<uint32_t> id1, id2;
  if (!pairs->a1_index) { id1.resize(pairs->npairs); for (uint32_t i = 0; i < pairs->npairs; ++i) id1[i] = i; }
  if (!pairs->a2_index) { id2.resize(pairs->npairs); for (uint32_t i = 0; i < pairs->npairs; ++i) id2[i] = i; }
  const uint32_t* x1 = pairs->a1_index ? pairs->a1_index : id1.data();
  const uint32_t* x2 = pairs->a2_index ? pairs->a2_index : id2.data();
  return run_chunks(g->ctx, D, [&](tracyhip_ctx* c, uint32_t part, uint32_t, uint32_t) -> int {
    const uint64_t lo = b[part], hi = b[part + 1];
    if (hi == lo) return TRACYHIP_OK;
    tracyhip_pairs p = *pairs;
    p.npairs = (uint32_t)(hi - lo);
    p.a1_index = x1 + lo;
    p.a2_index = x2 + lo;
    return tracyhip_gotoh_score(c, &p, prm, TRACYHIP_MEM_HOST, scores + lo);
  });
}

}  // extern "C"

extern "C" int tracyhip_trim_reference_slice(tracyhip_ctx* ctx, uint32_t ntraces, const uint8_t* rows0, const uint8_t* rows1,
                                             const uint64_t* rows_offset, const uint32_t* rows_len, const uint32_t* refslice_len,
                                             const uint8_t* forward, uint32_t trim_left, uint32_t trim_right, int mem,
                                             uint32_t* slice_begin, uint32_t* slice_len, uint32_t* ref_pos) {
  int rc = ctx_begin(ctx);
  if (rc) return rc;
  if (mem != TRACYHIP_MEM_HOST && mem != TRACYHIP_MEM_DEVICE) return set_error(TRACYHIP_ERR_ARG, "bad mem kind");
  if (ntraces == 0) return TRACYHIP_OK;
  if (!rows0 || !rows1 || !rows_offset || !rows_len || !refslice_len || !forward || !slice_begin || !slice_len || !ref_pos)
    return set_error(TRACYHIP_ERR_ARG, "null argument");
  hipStream_t st = ctx->stream;
  uint64_t ext = 0;
  std::vector<TrimRowsDesc> hd(ntraces);
  for (uint32_t t = 0; t < ntraces; ++t) {
    ext = std::max<uint64_t>(ext, rows_offset[t] + rows_len[t]);
    hd[t] = TrimRowsDesc{rows_offset[t], rows_len[t], refslice_len[t], (uint8_t)(forward[t] ? 1 : 0), {0, 0, 0, 0, 0, 0, 0}};
  }
  const void *d_r0, *d_r1;
  if ((rc = stage_in(ctx, ctx->d_rows0, rows0, ext, mem, &d_r0))) return rc;
  if ((rc = stage_in(ctx, ctx->d_rows1, rows1, ext, mem, &d_r1))) return rc;
  HIP_TRY(ctx->d_desc.ensure(sizeof(TrimRowsDesc) * (size_t)ntraces));
  HIP_TRY(hipMemcpyAsync(ctx->d_desc.p, hd.data(), sizeof(TrimRowsDesc) * (size_t)ntraces, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx->d_tmp[5].ensure(sizeof(TrimOut) * (size_t)ntraces));
  hipLaunchKernelGGL(trim_rows_kernel, dim3(ntraces), dim3(64), 0, st, static_cast<const TrimRowsDesc*>(ctx->d_desc.p),
                     static_cast<const uint8_t*>(d_r0), static_cast<const uint8_t*>(d_r1), trim_left, trim_right, ntraces,
                     static_cast<TrimOut*>(ctx->d_tmp[5].p));
  HIP_TRY(hipGetLastError());
  std::vector<TrimOut> h(ntraces);
  HIP_TRY(hipMemcpyAsync(h.data(), ctx->d_tmp[5].p, sizeof(TrimOut) * (size_t)ntraces, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx_sync(ctx));  // (hd is pageable: the upload above has completed by now as well)
  std::vector<uint32_t> b(ntraces), l(ntraces), p(ntraces);
  for (uint32_t t = 0; t < ntraces; ++t) { b[t] = h[t].ri; l[t] = h[t].len; p[t] = h[t].pos; }
  const hipMemcpyKind up = (mem == TRACYHIP_MEM_HOST) ? hipMemcpyHostToHost : hipMemcpyHostToDevice;
  HIP_TRY(hipMemcpy(slice_begin, b.data(), sizeof(uint32_t) * (size_t)ntraces, up));
  HIP_TRY(hipMemcpy(slice_len, l.data(), sizeof(uint32_t) * (size_t)ntraces, up));
  HIP_TRY(hipMemcpy(ref_pos, p.data(), sizeof(uint32_t) * (size_t)ntraces, up));
  return TRACYHIP_OK;
}
