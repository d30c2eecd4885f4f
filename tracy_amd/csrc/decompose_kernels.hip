// decompose_kernels.hip -- gfx950 kernels for the allele-deconvolution stage (decompose.h) and their
// C-ABI entry points.  Byte/integer and fp64 work, no MFMA; see decompose_kernels.h for the phases.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "../../include/tracy_hip.h"
#include "capi_internal.h"
#include "decompose_kernels.h"
#include "decompose_launch.h"
#include "decompose_wave.h"

using namespace tracyhip;

#define HIP_TRY(expr)                                                                               \
  do {                                                                                              \
    hipError_t _e = (expr);                                                                         \
    if (_e != hipSuccess)                                                                           \
      return set_error(_e == hipErrorOutOfMemory ? TRACYHIP_ERR_OOM : TRACYHIP_ERR_HIP, "%s failed: %s (%s:%d)", \
                       #expr, hipGetErrorString(_e), __FILE__, __LINE__);                           \
  } while (0)

namespace {

// dynamic LDS a staging kernel may ask for (the static arrays of allelic_fraction_kernel come on top); beyond it the staging
// arrays move to global scratch.  TRACYHIP_LDS_STAGE_LIMIT (bytes) lowers it, for tests of the global variants.
static size_t lds_stage_limit() {
  static size_t v = [] { const char* e = getenv("TRACYHIP_LDS_STAGE_LIMIT"); return e ? (size_t)atol(e) : (size_t)(140 * 1024); }();
  return v;
}
#define kLdsStageLimit lds_stage_limit()

__device__ __forceinline__ void wg_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __syncthreads();
}

// ---- decomposeAlleles: one 64-lane workgroup per trace ---------------------------------------------
template <int MAXI>
__global__ __launch_bounds__(64) void decompose_kernel(DecompArgs a, const BreakpointOut* bps) {
  extern __shared__ __attribute__((aligned(16))) char decomp_smem[];  // dynamic: the MAXI = 4096 state (80 KB) exceeds the static limit
  DecompSharedT<MAXI>& sh = *reinterpret_cast<DecompSharedT<MAXI>*>(decomp_smem);
  const uint32_t t = blockIdx.x;
  if (a.skip && a.skip[t]) return;
  if (a.only && !a.only[t]) return;
  DecompDesc d = a.desc[t];
  d.breakpoint = bps[t].breakpoint;
  if (a.lens) d.L = a.lens[t];
  const uint32_t lane = threadIdx.x;
  DecompOut out{};
  for (int st = 0; st < kDecompSteps; ++st) {
    if (lane == 0 || decomp_step_all_lanes(st)) decomp_step(st, a, d, sh, out, lane);
    wg_sync();
  }
  if (lane == 0) a.out[t] = out;
}

// ---- decomposeAlleles, one wave per trace with its working set in LDS (decompose_wave.h) ----
struct DecompDevWave {
  __device__ __forceinline__ uint32_t lane() const { return threadIdx.x; }
  __device__ __forceinline__ uint64_t ballot(bool p) const { return __ballot(p); }
  __device__ __forceinline__ uint32_t bcast(uint32_t x, uint32_t src_lane) const { return (uint32_t)__builtin_amdgcn_readlane((int)x, (int)src_lane); }
  __device__ __forceinline__ void sync() const { wg_sync(); }
  __device__ __forceinline__ char* lds() const {
    extern __shared__ __attribute__((aligned(16))) char decomp_smem[];
    return decomp_smem;
  }
  // reductions over the 64 lanes: four DPP steps leave every lane of a row of sixteen with the row's result (quad_perm [1,0,3,2] and
  // [2,3,0,1], row_half_mirror, row_mirror), four v_readlane join the rows -- the result is wave-uniform (an SGPR)
  template <class F>
  __device__ __forceinline__ uint32_t reduce(uint32_t x, F f) const {
    x = f(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, false));
    x = f(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, false));
    x = f(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xf, 0xf, false));
    x = f(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xf, 0xf, false));
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)x, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)x, 16);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)x, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)x, 48);
    return f(f(a, b), f(c, d));
  }
  __device__ __forceinline__ uint32_t sum(uint32_t x) const { return reduce(x, [](uint32_t a, uint32_t b) { return a + b; }); }
  __device__ __forceinline__ uint32_t umin(uint32_t x) const { return reduce(x, [](uint32_t a, uint32_t b) { return a < b ? a : b; }); }
  __device__ __forceinline__ uint32_t umax(uint32_t x) const { return reduce(x, [](uint32_t a, uint32_t b) { return a > b ? a : b; }); }
  __device__ __forceinline__ uint32_t excl_sum(uint32_t x) const {  // sum of the lanes below (Hillis-Steele, six shuffles)
    uint32_t v = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = (uint32_t)__shfl_up((int)v, o, 64);
      if ((int)threadIdx.x >= o) v += y;
    }
    return v - x;
  }
};
__global__ __launch_bounds__(64) void decompose_wave_kernel(DecompWaveArgs wa) {
  DecompDevWave w;
  decomp_wave_body(w, wa, blockIdx.x);
}

// the same phases with the scan state in global memory: a workgroup keeps its slot and takes traces blockIdx.x, + gridDim.x, ...
__global__ __launch_bounds__(64) void decompose_kernel_global(DecompArgs a, const BreakpointOut* bps, char* state) {
  DecompSharedT<kMaxIndelGlobal>& sh = *reinterpret_cast<DecompSharedT<kMaxIndelGlobal>*>(state + (size_t)blockIdx.x * sizeof(DecompSharedT<kMaxIndelGlobal>));
  const uint32_t lane = threadIdx.x;
  for (uint32_t t = blockIdx.x; t < a.ntraces; t += gridDim.x) {
    if (a.skip && a.skip[t]) continue;
    DecompDesc d = a.desc[t];
    d.breakpoint = bps[t].breakpoint;
    if (a.lens) d.L = a.lens[t];
    DecompOut out{};
    for (int st = 0; st < kDecompSteps; ++st) {
      if (lane == 0 || decomp_step_all_lanes(st)) decomp_step(st, a, d, sh, out, lane);
      wg_sync();
    }
    if (lane == 0) a.out[t] = out;
    wg_sync();
  }
}

// ---- findBreakpoint: one workgroup per profile; sig/diff in dynamic LDS (ncol doubles each) --------
// GLOBAL: the staging arrays of a profile too long for LDS live in a per-workgroup slice of a global scratch buffer
template <bool GLOBAL>
__global__ __launch_bounds__(64) void breakpoint_kernel(const BpDesc* desc, const float* prof, BreakpointOut* out, char* scratch, size_t stride) {
  extern __shared__ __attribute__((aligned(16))) char lds_stage[];
  char* smem = GLOBAL ? scratch + (size_t)blockIdx.x * stride : lds_stage;
  const BpDesc d = desc[blockIdx.x];
  // sig[] holds the signal ratios, then (they are dead once the window sums exist) the differences; lsum[i] = sig[i-25] + ... + sig[i-1]
  // summed in the reference's order (decompose.h:32-38): its right-hand window at i is the left-hand window at i + 25, term for term, so
  // one sum per position serves both
  double* sig = reinterpret_cast<double*>(smem);
  double* lsum = sig + d.ncol;
  double* diff = sig;
  uint8_t* ltr = reinterpret_cast<uint8_t*>(lsum + d.ncol);
  const float* p = prof + d.off;
  const uint32_t lane = threadIdx.x;
  // the six rows of eight rounds of columns requested together (clamped columns, no branch): a wave that waited for every round of 64
  // columns on its own spent 15 memory round trips -- three quarters of its life -- here, with nine waves per CU (LDS) to hide them
  constexpr uint32_t kCols = 8;
  for (uint32_t j0 = 0; j0 < d.ncol; j0 += 64 * kCols) {
    float v[kCols][6];
#pragma unroll
    for (uint32_t u = 0; u < kCols; ++u) {
      const uint32_t j = j0 + 64 * u + lane, jc = j < d.ncol ? j : d.ncol - 1;
#pragma unroll
      for (uint32_t i = 0; i < 6; ++i) v[u][i] = p[(uint64_t)i * d.stride + jc];
    }
#pragma unroll
    for (uint32_t u = 0; u < kCols; ++u) {
      const uint32_t j = j0 + 64 * u + lane;
      double best = 0.001, snd = 0.001;  // signal_ratio (decompose_kernels.h) on registers, selects for its if / else-if
#pragma unroll
      for (uint32_t i = 0; i < 6; ++i) {
        const double x = v[u][i];
        const bool gb = x > best, gs = x > snd;
        snd = gb ? best : gs ? x : snd;
        best = gb ? x : best;
      }
      if (j < d.ncol) sig[j] = __dsub_rn(best, snd);
    }
  }
  wg_sync();
  // two positions per lane and pass: two independent chains of 25 additions
  for (uint32_t i = 25 + lane; i < d.ncol; i += 128) {
    const uint32_t i2 = i + 64 < d.ncol ? i + 64 : i;
    double s1 = 0, s2 = 0;
#pragma unroll
    for (uint32_t k = 25; k > 0; --k) { s1 += sig[i - k]; s2 += sig[i2 - k]; }
    lsum[i] = s1;
    lsum[i2] = s2;
  }
  wg_sync();
  // breakpoint_select (decompose.h:27-55) by the 64 lanes.  The reference walks the positions with a float-typed running
  // maximum: position i is taken when diff[i] > (double)best, and best becomes (float)diff[i].  Rounding is monotone, so the
  // walk ends with best = F = max(0, max_i (float)diff[i]); the first position whose float equals F is always taken, and after
  // it exactly the positions with diff[i] > (double)F (they round down to F) -- the answer is the last of those, or that first
  // position.  Three reductions instead of one lane reading ncol doubles one after the other.
  const uint32_t lo = 25, hi = (25 < d.ncol) ? d.ncol - 25 : 25;
  float fmax_l = 0.0f;
  for (uint32_t i = lo + lane; i < hi; i += 64) {
    const double left = lsum[i] / 25.0, right = lsum[i + 25] / 25.0;
    const double dd = right - left;
    const double v = dd < 0 ? -dd : dd;
    diff[i] = v;  // (position i of the dead sig[] is this lane's alone)
    ltr[i] = (left < right) ? 1 : 0;
    const float g = (float)v;
    if (v > 0.0 && g > fmax_l) fmax_l = g;
  }
  for (int o = 32; o > 0; o >>= 1) { const float x = __shfl_xor(fmax_l, o, 64); if (x > fmax_l) fmax_l = x; }
  const float F = fmax_l;
  uint32_t first_l = 0xffffffffu, last_l = 0;  // last_l: position + 1, 0 = none
  for (uint32_t i = lo + lane; i < hi; i += 64) {
    const double v = diff[i];
    if (F > 0.0f && (float)v == F && v > 0.0 && i < first_l) first_l = i;
    if (v > (double)F) last_l = i + 1;
  }
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t a = (uint32_t)__shfl_xor((int)first_l, o, 64), b = (uint32_t)__shfl_xor((int)last_l, o, 64);
    if (a < first_l) first_l = a;
    if (b > last_l) last_l = b;
  }
  wg_sync();  // (ltr[] of the other lanes)
  if (lane == 0) {
    BreakpointOut bp;
    bp.bestDiff = 0; bp.traceleft = 1; bp.breakpoint = 0;
    const bool any = last_l != 0 || first_l != 0xffffffffu;
    if (any) {
      const uint32_t idx = last_l ? last_l - 1 : first_l;
      bp.breakpoint = idx;
      bp.bestDiff = F;
      bp.traceleft = ltr[idx] ? 0 : 1;
    }
    bp.indelshift = 1;
    if ((double)bp.bestDiff < 0.25) {
      bp.indelshift = 0;
      bp.breakpoint = d.ncol;
      bp.traceleft = 1;
      bp.bestDiff = 0;
    }
    out[blockIdx.x] = bp;
  }
}

// ---- findHomozygousBreakpoint (decompose.h:59-128): one wavefront per trace; the formulation is in decompose_kernels.h ----
// (One lane per trace read the rows a byte at a time from 10^5 different pages: 1.4 ms at best and 27 ms on its bad days.)
__device__ __forceinline__ HomChunk hom_chunk(const uint8_t* r0, const uint8_t* r1, uint32_t L, uint64_t b, uint32_t lane) {
  const uint64_t j = b + lane;
  uint8_t x = '-', y = '-';
  if (j < L) { x = r0[j]; y = r1[j]; }
  HomChunk c;
  c.mm = __ballot(x != y);
  c.ng = __ballot(x != '-');
  return c;
}

// SELECT = false: F = max(0, max_i (float)diff_i) and the varIndex the walk ends with; SELECT = true: the column the
// reference's walk ends on, given F
template <bool SELECT>
__device__ __forceinline__ void hom_sweep(const uint8_t* r0, const uint8_t* r1, uint32_t L, uint32_t lo, uint32_t hi, uint32_t lane,
                                          float& F, uint32_t& var_end, uint32_t& var_at, int32_t& left_lt_right) {
  float fmax_l = 0.0f;
  uint32_t vbase = 0, vend = 0;
  HomPick pk;
  hom_pick_init(pk);
  HomChunk cur = hom_chunk(r0, r1, L, 0, lane), nxt = hom_chunk(r0, r1, L, 64, lane);
  uint64_t prev = 0;
  for (uint64_t b = 0; b < hi; b += 64) {
    const HomChunk nn = hom_chunk(r0, r1, L, b + 128, lane);
    const uint64_t i = b + lane;
    if (i >= lo && i < hi) {
      const HomLane h = hom_lane(prev, cur, nxt.mm, vbase, lane);
      if (!SELECT) { if (h.diff > 0.0 && h.g > fmax_l) fmax_l = h.g; }
      else hom_pick(pk, h, F, (uint32_t)i);
    }
    if (!SELECT) vend += hom_bases_below(cur, b, hi);
    vbase += (uint32_t)__popcll(cur.ng);
    prev = cur.mm; cur = nxt; nxt = nn;
  }
  if (!SELECT) {
    for (int o = 32; o > 0; o >>= 1) { const float x = __shfl_xor(fmax_l, o, 64); if (x > fmax_l) fmax_l = x; }
    F = fmax_l;
    var_end = vend;
    return;
  }
  uint32_t first = pk.first, last = pk.last;
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t a = (uint32_t)__shfl_xor((int)first, o, 64), c = (uint32_t)__shfl_xor((int)last, o, 64);
    if (a < first) first = a;
    if (c > last) last = c;
  }
  // the columns are distinct across lanes: exactly one lane holds the chosen one
  const bool mine = last ? pk.last == last : pk.first == first;
  const uint64_t owner = __ballot(mine);
  const int src = owner ? __builtin_ctzll(owner) : 0;
  var_at = (uint32_t)__shfl((int)(last ? pk.last_var : pk.first_var), src, 64);
  left_lt_right = __shfl(last ? pk.last_tl : pk.first_tl, src, 64);
}

__global__ __launch_bounds__(64) void homozygous_kernel(const RowsDesc* desc, const uint8_t* rows0, const uint8_t* rows1,
                                                        uint32_t n, BreakpointOut* bps, int32_t* status, const uint32_t* lens) {
  const uint32_t t = blockIdx.x, lane = threadIdx.x;
  BreakpointOut bp = bps[t];
  if (bp.indelshift) {  // only when the trace shows no shift (indigo.h:314-317)
    if (lane == 0) status[t] = 1;
    return;
  }
  const RowsDesc d = desc[t];
  const uint8_t* r0 = rows0 + d.off;
  const uint8_t* r1 = rows1 + d.off;
  const uint32_t L = lens ? lens[t] : d.L;
  // first and last column with a base in both rows
  int64_t align_start = 0, align_end = 0;
  for (uint64_t b = 0; b < L; b += 64) {
    const uint64_t j = b + lane;
    const uint64_t m = __ballot(j < L && r0[j] != '-' && r1[j] != '-');
    if (m) { align_start = (int64_t)b + __builtin_ctzll(m); break; }
  }
  for (int64_t b = L ? (int64_t)((L - 1) / 64) * 64 : -1; b >= 0; b -= 64) {
    const uint64_t j = (uint64_t)b + lane;
    const uint64_t m = __ballot(j < L && r0[j] != '-' && r1[j] != '-');
    if (m) { align_end = b + 63 - __builtin_clzll(m); break; }
  }
  uint32_t lo = 0, hi = 0;
  const int rc = hom_range(align_start, align_end, bp, lo, hi);
  if (rc == 1) {
    float F = 0.0f;
    uint32_t var_end = 0, var_at = 0;
    int32_t ltr = 0;
    hom_sweep<false>(r0, r1, L, lo, hi, lane, F, var_end, var_at, ltr);
    hom_finish(bp, F, var_end);
    if (bp.indelshift) {
      hom_sweep<true>(r0, r1, L, lo, hi, lane, F, var_end, var_at, ltr);
      bp.breakpoint = var_at;
      bp.bestDiff = F;
      bp.traceleft = ltr;
    }
  }
  if (lane == 0) { status[t] = rc; bps[t] = bp; }
}

// ---- the peak table ------------------------------------------------------------------------------------------------------------
// Every read of the chromatogram on this path is at a basecall's peak position: createProfile (profile.h:21-52, host),
// generateSecondaryDecomposed (decompose.h:378-410) and allelicFraction (decompose.h:445-470) all index traceACGT[k][bcPos[i]].  The
// kernels therefore read a compact table -- peaks[bc_off + i] = {A, C, G, T at bcPos[i]}, 16 bytes per basecall instead of the 192 KB
// chromatogram of a 1 kb trace, consecutive basecalls in consecutive lanes -- which the caller passes (tracyhip_basecalls::peaks) or
// peaks_kernel builds once per call from signal + bcpos (one pass over the chromatogram, four gathered words per basecall).
__global__ __launch_bounds__(256) void peaks_kernel(const BcDesc* __restrict__ desc, const int32_t* __restrict__ signal, const int32_t* __restrict__ bcpos,
                                                    int4* __restrict__ peaks) {
  const BcDesc d = desc[blockIdx.y];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.nbc) return;
  const int32_t* sg = signal + d.sig_off;
  const uint64_t tpos = (uint32_t)bcpos[d.bc_off + i];
  peaks[d.bc_off + i] = make_int4(sg[tpos], sg[(uint64_t)d.nsamples + tpos], sg[2ull * d.nsamples + tpos], sg[3ull * d.nsamples + tpos]);
}

__global__ __launch_bounds__(256) void secdecomp_kernel(const BcDesc* __restrict__ desc, const int4* __restrict__ peaks, const uint8_t* __restrict__ pri,
                                                        const uint8_t* __restrict__ sec, uint8_t* __restrict__ outp) {
  const BcDesc d = desc[blockIdx.y];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.nbc) return;
  const uint8_t p = pri[d.bc_off + i], s = sec[d.bc_off + i];
  uint8_t r = p;
  if (p != s) {
    r = s;
    if (!(s == 'A' || s == 'C' || s == 'G' || s == 'T')) {  // an IUPAC code: the larger of its two channels (the only case that reads the table)
      const int4 k = peaks[d.bc_off + i];
      r = secondary_decomposed(p, s, k.x, k.y, k.z, k.w);
    }
  }
  outp[d.bc_off + i] = r;
}

// ---- allelicFraction (decompose.h:412-621) -----------------------------------------------------------
// Block of 256 threads per trace.  The reference keeps the first candidate (i,j,k ascending) whose full SSE
// is strictly below everything before it, starting from SSE(0.5,0.5,0,0); its `break` only skips terms of
// candidates that already lost.  Its answer is therefore argmin over the 171 700 grid points of the SSE as
// summed in the reference's order (fp64, sub/mul/add, no FMA), first index on ties, provided it beats the
// start point.  Summing 4*d terms for every grid point is what costs the reference its time; here the
// candidates are first screened with the closed form  SSE = sum_c (n_c v_c^2 - 2 v_c S_c + Q_c)  over the
// five classes c (position belongs to the primary / secondary / tertiary / quaternary allele or none),
// which differs from the sequentially rounded sum by far less than kScreenMargin; only candidates whose
// screened value lies within that margin of the screened minimum can attain (or tie) the exact minimum and
// are evaluated exactly, in the reference's order.  The selected pair is bit-identical.
constexpr int AF_THREADS = 256;
constexpr double kScreenMargin = 1e-6;  // >> 4d * 2^-53 * SSE rounding differences (SSE <= 4d <= 8192)

constexpr int AF_SURVIVORS = 32;  // near-minimum candidates evaluated cooperatively; any further ones by their finder

// the reference's summation for one candidate (decompose.h:596-606): sequential over the 4*dn terms
__device__ __forceinline__ double af_exact_sse(const double* tp, const uint8_t* cls, uint32_t terms, const double pv[5]) {
  // the class value is picked with selects: a dynamically indexed private array would live in scratch memory
  const double p0 = pv[0], p1 = pv[1], p2 = pv[2], p3 = pv[3], p4 = pv[4];
  auto term = [&](uint32_t q) {
    const uint32_t c = cls[q];
    const double v = c == 1 ? p1 : c == 2 ? p2 : c == 3 ? p3 : c == 4 ? p4 : p0;
    const double df = __dsub_rn(v, tp[q]);
    return __dmul_rn(df, df);
  };
  double sse = 0;
  uint32_t q = 0;
  for (; q + 4 <= terms; q += 4) {  // the squares are independent; only the additions form the sequential chain
    const double x0 = term(q), x1 = term(q + 1), x2 = term(q + 2), x3 = term(q + 3);
    sse = __dadd_rn(sse, x0); sse = __dadd_rn(sse, x1); sse = __dadd_rn(sse, x2); sse = __dadd_rn(sse, x3);
  }
  for (; q < terms; ++q) sse = __dadd_rn(sse, term(q));
  return sse;
}

// The (i, j, k) grid of decompose.h:593-595 is the same for every trace: i, j, k run over the 100 doubles
// 0, 0.01, 0.01+0.01, ... and a triple is visited iff i+j <= 1 and (i+j)+k <= 1 in double arithmetic.  The host
// enumerates the visited (i, j) pairs once with the number of k values each admits, sorted by that number:
// dealing the sorted list round-robin gives the 64 lanes of a wave k-loops of (almost) equal length.
struct AfGrid {
  uint32_t npairs;
  const uint32_t* pair;  // [npairs] (i_index * 100 + j_index) | k_count << 16, k_count descending
  const double* vals;    // the 100 doubles 0, 0.01, 0.01 + 0.01, ... (af_search_kernel)
  const uint32_t* pair8; // [npairs] i_index | j_index << 8 | k_count << 16, in the order of `pair` (af_search_kernel: no divisions)
};
constexpr int AF_STEPS = 21;  // >= ceil(npairs / AF_THREADS) (5151 pairs)

__device__ __forceinline__ double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

template <bool GLOBAL>
__global__ __launch_bounds__(AF_THREADS) void allelic_fraction_kernel(const BcDesc* desc, const int4* peaks, const uint8_t* pri_all,
                                                                      const uint8_t* sec_all, uint32_t trimLeft,
                                                                      uint32_t trimRight, AfGrid grid, double* fractions,
                                                                      char* scratch, size_t stride) {
  extern __shared__ __attribute__((aligned(16))) char lds_stage[];
  char* smem = GLOBAL ? scratch + (size_t)blockIdx.x * stride : lds_stage;  // tp / cls of a trace too long for LDS: global scratch
  __shared__ double vals[100], f1[100], f2[100], f3[100];
  __shared__ double red_sse[AF_THREADS];
  __shared__ uint32_t red_idx[AF_THREADS];
  __shared__ uint32_t s_cnt[AF_THREADS];
  __shared__ uint32_t s_nsurv;
  __shared__ uint32_t s_surv[AF_SURVIVORS];
  __shared__ double s_pv[AF_SURVIVORS + 1][5];
  __shared__ double s_amin;
  __shared__ double s_mom[AF_THREADS / 64][9];
  const BcDesc d = desc[blockIdx.x];
  const uint8_t* pri = pri_all + d.bc_off;
  const uint8_t* sec = sec_all + d.bc_off;
  // trimmedSeq (abif.h:68-75)
  uint32_t off = trimLeft, len;
  if ((uint64_t)(uint32_t)(trimLeft + trimRight + 1) >= (uint64_t)d.nbc) { off = 0; len = d.nbc; }
  else len = d.nbc - trimLeft - trimRight;
  double* tp = reinterpret_cast<double*>(smem);                 // [4][dn], the reference's m-outer / n-inner order
  uint8_t* cls = reinterpret_cast<uint8_t*>(tp + 4 * (size_t)len);
  const int tid = threadIdx.x;
  // positions where primary != secondary (decompose.h:431-436), compacted in order: every thread owns a chunk
  const uint32_t chunk = (len + AF_THREADS - 1) / AF_THREADS;
  const uint32_t c_lo = min((uint32_t)tid * chunk, len), c_hi = min(c_lo + chunk, len);
  {
    uint32_t n = 0;
    for (uint32_t i = c_lo; i < c_hi; ++i) n += pri[off + i] != sec[off + i];
    s_cnt[tid] = n;
    if (tid == 0) {
      double x = 0;
      for (int a = 0; a < 100; ++a) { vals[a] = x; x = __dadd_rn(x, 0.01); }  // for (double i = 0; i <= 1; i += 0.01)
      s_nsurv = 0;
    }
  }
  __syncthreads();
  uint32_t np = 0, dn = 0;
  for (int t = 0; t < AF_THREADS; ++t) {
    const uint32_t c = s_cnt[t];
    if (t < tid) np += c;
    dn += c;
  }
  if (dn == 0) {
    if (tid == 0) { fractions[2 * blockIdx.x] = 0.5; fractions[2 * blockIdx.x + 1] = 0.5; }
    return;
  }
  {
    for (uint32_t i = c_lo; i < c_hi; ++i) {
      const uint8_t pc = pri[off + i], sc = sec[off + i];
      if (pc == sc) continue;
      const uint32_t bi_ = (i + trimLeft < d.nbc) ? i + trimLeft : d.nbc - 1;  // the reference indexes bcPos[i + trimLeft] (decompose.h:445)
      const int4 pk = peaks[d.bc_off + bi_];
      const int32_t s4[4] = {pk.x, pk.y, pk.z, pk.w};
      const double sigsum = (double)(s4[0] + s4[1] + s4[2] + s4[3]);
      for (int k = 0; k < 4; ++k) { tp[(size_t)k * dn + np] = __ddiv_rn((double)s4[k], sigsum); cls[(size_t)k * dn + np] = 0; }
      const int pi = pc == 'A' ? 0 : pc == 'C' ? 1 : pc == 'G' ? 2 : pc == 'T' ? 3 : -1;
      const int si = sc == 'A' ? 0 : sc == 'C' ? 1 : sc == 'G' ? 2 : sc == 'T' ? 3 : -1;
      if (pi >= 0 && si >= 0) {
        int rest[2], nr = 0;
        for (int k = 0; k < 4; ++k) if (k != pi && k != si) rest[nr++] = k;
        cls[(size_t)pi * dn + np] = 1;
        cls[(size_t)si * dn + np] = 2;
        const bool first = s4[rest[0]] > s4[rest[1]];  // tertiary = the larger of the two remaining channels
        cls[(size_t)rest[first ? 0 : 1] * dn + np] = 3;
        cls[(size_t)rest[first ? 1 : 0] * dn + np] = 4;
      }
      ++np;
    }
  }
  __syncthreads();
  const uint32_t terms = 4 * dn;
  // per-class moments for the screen: sum_c (n_c v^2 - 2 v S_c) + Q, Q = sum of all squares
  {
    double m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // n1..n4, S1..S4, Q
    for (uint32_t q = tid; q < terms; q += AF_THREADS) {
      const int c = cls[q];
      const double x = tp[q];
      m[8] += x * x;
      for (int k = 1; k <= 4; ++k) { m[k - 1] += (c == k) ? 1.0 : 0.0; m[3 + k] += (c == k) ? x : 0.0; }
    }
    for (int k = 0; k < 9; ++k) {
      const double w = wave_sum(m[k]);
      if ((tid & 63) == 0) s_mom[tid >> 6][k] = w;
    }
  }
  __syncthreads();
  double mom[9];
  for (int k = 0; k < 9; ++k) {
    double v = 0;
    for (int w = 0; w < AF_THREADS / 64; ++w) v += s_mom[w][k];
    mom[k] = v;
  }
  const double n4 = mom[3], s4c2 = 2.0 * mom[7], qtot = mom[8];
  if (tid < 100) {
    const double v = vals[tid];
    f1[tid] = v * (mom[0] * v - 2.0 * mom[4]);
    f2[tid] = v * (mom[1] * v - 2.0 * mom[5]);
    f3[tid] = v * (mom[2] * v - 2.0 * mom[6]);
  }
  __syncthreads();
  // closed-form SSE of candidate (pair, ic): the SAME expression in both passes
  auto screen = [&](double fij, double sij, uint32_t ic) {
    const double vl = __dsub_rn(1.0, __dadd_rn(sij, vals[ic]));
    return (fij + f3[ic]) + vl * (n4 * vl - s4c2);
  };
  // pass 1: minimum of every pair this thread owns (kept in registers: the step loop is fully unrolled)
  double pair_min[AF_STEPS];
  double my_min = 1e300;
#pragma unroll
  for (int st = 0; st < AF_STEPS; ++st) {
    const uint32_t p = (uint32_t)st * AF_THREADS + (uint32_t)tid;
    double pm = 1e300;
    if (p < grid.npairs) {
      const uint32_t e = grid.pair[p];
      const uint32_t ia = (e & 0xffffu) / 100u, ib = (e & 0xffffu) % 100u, cnt = e >> 16;
      const double sij = __dadd_rn(vals[ia], vals[ib]);
      const double fij = qtot + f1[ia] + f2[ib];
      for (uint32_t ic = 0; ic < cnt; ++ic) {
        const double a = screen(fij, sij, ic);
        pm = a < pm ? a : pm;
      }
    }
    pair_min[st] = pm;
    my_min = pm < my_min ? pm : my_min;
  }
  {
    double m = my_min;
    for (int o = 32; o > 0; o >>= 1) { const double x = __shfl_down(m, o, 64); m = x < m ? x : m; }
    if ((tid & 63) == 0) red_sse[tid >> 6] = m;
  }
  __syncthreads();
  if (tid == 0) {
    double m = red_sse[0];
    for (int q = 1; q < AF_THREADS / 64; ++q) m = red_sse[q] < m ? red_sse[q] : m;
    s_amin = m;
  }
  __syncthreads();
  // pass 2: re-scan only the pairs that reach the margin of the screened minimum, list their near-minimum
  // candidates; those get the exact (sequentially rounded) sum of the reference
  const double cut = s_amin + kScreenMargin;
  double my_sse = 1e300;
  uint32_t my_idx = 0xffffffffu;
  auto exact_of = [&](uint32_t idx) {
    const double vi = vals[idx / 10000], vj = vals[(idx / 100) % 100], vk = vals[idx % 100];
    const double pv[5] = {0.0, vi, vj, vk, __dsub_rn(1.0, __dadd_rn(__dadd_rn(vi, vj), vk))};
    return af_exact_sse(tp, cls, terms, pv);
  };
#pragma unroll
  for (int st = 0; st < AF_STEPS; ++st) {
    if (!(pair_min[st] <= cut)) continue;
    const uint32_t e = grid.pair[(uint32_t)st * AF_THREADS + (uint32_t)tid];
    const uint32_t code = e & 0xffffu, ia = code / 100u, ib = code % 100u, cnt = e >> 16;
    const double sij = __dadd_rn(vals[ia], vals[ib]);
    const double fij = qtot + f1[ia] + f2[ib];
    for (uint32_t ic = 0; ic < cnt; ++ic) {
      if (screen(fij, sij, ic) > cut) continue;
      const uint32_t idx = code * 100u + ic;
      const uint32_t slot = atomicAdd(&s_nsurv, 1u);
      if (slot < AF_SURVIVORS) { s_surv[slot] = idx; continue; }
      const double sse = exact_of(idx);  // list full: evaluate right here
      if (sse < my_sse || (sse == my_sse && idx < my_idx)) { my_sse = sse; my_idx = idx; }
    }
  }
  __syncthreads();
  // exact stage: lane 0 sums the start point (0.5, 0.5, 0, 0) (decompose.h:586-591), lanes 1.. one listed
  // survivor each -- the same sequential loop side by side, class values looked up in LDS
  const uint32_t nsurv = min(s_nsurv, (uint32_t)AF_SURVIVORS);
  if ((uint32_t)tid <= nsurv) {
    double pv[5] = {0.0, 0.5, 0.5, 0.0, 0.0};
    if (tid > 0) {
      const uint32_t idx = s_surv[tid - 1];
      const double vi = vals[idx / 10000], vj = vals[(idx / 100) % 100], vk = vals[idx % 100];
      pv[1] = vi; pv[2] = vj; pv[3] = vk;
      pv[4] = __dsub_rn(1.0, __dadd_rn(__dadd_rn(vi, vj), vk));
    }
    for (int c = 0; c < 5; ++c) s_pv[tid][c] = pv[c];
    const double* mine = s_pv[tid];
    double sse = 0;
    for (uint32_t q = 0; q < terms; ++q) {
      const double df = __dsub_rn(mine[cls[q]], tp[q]);
      sse = __dadd_rn(sse, __dmul_rn(df, df));
    }
    if (tid == 0) s_amin = sse;  // sse0 (s_amin is free again)
    else if (sse < my_sse || (sse == my_sse && s_surv[tid - 1] < my_idx)) { my_sse = sse; my_idx = s_surv[tid - 1]; }
  }
  red_sse[tid] = my_sse;
  red_idx[tid] = my_idx;
  __syncthreads();
  if (tid == 0) {
    // lowest-index strict minimum below the start SSE == the reference's chain of strict improvements
    double best = s_amin;
    uint32_t bidx = 0xffffffffu;
    for (int q = 0; q < AF_THREADS; ++q) {
      if (red_idx[q] == 0xffffffffu) continue;
      if (red_sse[q] < best || (red_sse[q] == best && bidx != 0xffffffffu && red_idx[q] < bidx)) { best = red_sse[q]; bidx = red_idx[q]; }
    }
    double bi = 0.5, bj = 0.5;
    if (bidx != 0xffffffffu) { bi = vals[bidx / 10000]; bj = vals[(bidx / 100) % 100]; }
    fractions[2 * blockIdx.x] = bi;
    fractions[2 * blockIdx.x + 1] = bj;
  }
}

// ---- allelicFraction in two launches (round 5) ---------------------------------------------------------
// allelic_fraction_kernel keeps tp / cls of a trace in LDS (36 bytes per basecall: three workgroups per CU), screens all 171 700 grid
// points and then has ONE of its four waves walk the exact sums while the other three wait at a barrier (rocprofv3: 61 % of the wave
// cycles in s_waitcnt / s_barrier, 10 k VALU instructions per wave).  Here:
//   af_prepare_kernel  one wave per trace: the het positions, tp = signal / sum in the reference's order and the class of every term go
//                      to global scratch, the nine moments of the screen to a header;
//   af_search_kernel   one wave per trace, a few hundred bytes of LDS.  The screen is a parabola in k for a fixed (i, j) -- SSE(k) =
//                      const + v (n3 v - 2 S3) + (c - v)(n4 (c - v) - 2 S4), v = vals[k], c = 1 - (i + j) -- so the k that minimises it is
//                      known in closed form: five candidates around it give the pair's minimum instead of up to a hundred; the pairs within
//                      the margin of the overall minimum are then scanned in full for the survivor list, exactly as before.  The exact sums
//                      (decompose.h:596-606, sequentially rounded) read tp / cls through uniform, read-only pointers: scalar loads, the
//                      term in SGPRs, three fp64 operations and one LDS look-up per term and lane.
// The screen only decides WHO is evaluated exactly; the selected pair is the first candidate that attains the minimum of the exact sums,
// as in allelic_fraction_kernel (whose comment has the argument); the two kernels are compared with each other and with the oracle.
struct AfHeader {
  double mom[9];  // n1..n4, S1..S4, Q
  uint32_t dn;    // positions with primary != secondary
  uint32_t pad;
};
struct AfScratch {
  double* tp;     // trace t: [4][dn] at 4 * bc_off[t]
  uint8_t* cls;   // idem
  AfHeader* hdr;  // [ntraces]
};
__device__ __forceinline__ double wave_sum_all(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_min_all(double v) {
  for (int o = 32; o > 0; o >>= 1) { const double x = __shfl_xor(v, o, 64); v = x < v ? x : v; }
  return v;
}
// Lane = position (round 6): the bytes of 64 consecutive basecalls, their table entries and the compacted tp / cls entries of the het
// positions among them are consecutive addresses (a ballot gives every het lane its place).  Two passes over the basecall bytes: tp is
// laid out [channel][het position] -- the reference's summation order, decompose.h:596-606 -- so the count comes first.
__global__ __launch_bounds__(64) void af_prepare_kernel(const BcDesc* __restrict__ desc, const int4* __restrict__ peaks, const uint8_t* __restrict__ pri_all,
                                                        const uint8_t* __restrict__ sec_all, uint32_t trimLeft, uint32_t trimRight, AfScratch sc,
                                                        double* __restrict__ fractions) {
  const uint32_t t = blockIdx.x, lane = threadIdx.x;
  const BcDesc d = desc[t];
  const uint8_t* pri = pri_all + d.bc_off;
  const uint8_t* sec = sec_all + d.bc_off;
  uint32_t off = trimLeft, len;  // trimmedSeq (abif.h:68-75)
  if ((uint64_t)(uint32_t)(trimLeft + trimRight + 1) >= (uint64_t)d.nbc) { off = 0; len = d.nbc; }
  else len = d.nbc - trimLeft - trimRight;
  // positions where primary != secondary (decompose.h:431-436)
  uint32_t dn = 0;
  for (uint32_t i0 = 0; i0 < len; i0 += 64u) {
    const uint32_t i = i0 + lane;
    dn += (uint32_t)__popcll(__ballot(i < len && pri[off + i] != sec[off + i]));
  }
  double m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // n1..n4, S1..S4, Q over this lane's positions
  if (dn) {
    double* tp = sc.tp + 4ull * d.bc_off;
    uint8_t* cls = sc.cls + 4ull * d.bc_off;
    const int4* pk_t = peaks + d.bc_off;
    uint32_t base = 0;
    for (uint32_t i0 = 0; i0 < len; i0 += 64u) {
      const uint32_t i = i0 + lane;
      uint8_t pc = 0, scd = 0;
      if (i < len) { pc = pri[off + i]; scd = sec[off + i]; }
      const bool het = pc != scd;  // (both 0 beyond the end)
      const unsigned long long bal = __ballot(het);
      const uint32_t np = base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
      base += (uint32_t)__popcll(bal);
      if (!het) continue;
      const uint32_t bi_ = (i + trimLeft < d.nbc) ? i + trimLeft : d.nbc - 1;  // the reference indexes bcPos[i + trimLeft] (decompose.h:445)
      const int4 pk = pk_t[bi_];
      const int32_t s4[4] = {pk.x, pk.y, pk.z, pk.w};
      const double sigsum = (double)(s4[0] + s4[1] + s4[2] + s4[3]);
      uint32_t c4[4] = {0, 0, 0, 0};
      const int pi = pc == 'A' ? 0 : pc == 'C' ? 1 : pc == 'G' ? 2 : pc == 'T' ? 3 : -1;
      const int si = scd == 'A' ? 0 : scd == 'C' ? 1 : scd == 'G' ? 2 : scd == 'T' ? 3 : -1;
      if (pi >= 0 && si >= 0) {
        int rest[2], nr = 0;
        for (int k = 0; k < 4; ++k) if (k != pi && k != si) rest[nr++] = k;
        const bool first = s4[rest[0]] > s4[rest[1]];  // tertiary = the larger of the two remaining channels
        for (int k = 0; k < 4; ++k) c4[k] = k == pi ? 1u : k == si ? 2u : k == rest[first ? 0 : 1] ? 3u : 4u;
      }
      for (int k = 0; k < 4; ++k) {
        const double x = __ddiv_rn((double)s4[k], sigsum);
        tp[(size_t)k * dn + np] = x;
        cls[(size_t)k * dn + np] = (uint8_t)c4[k];
        m[8] += x * x;
        for (int q = 1; q <= 4; ++q) { m[q - 1] += (c4[k] == (uint32_t)q) ? 1.0 : 0.0; m[3 + q] += (c4[k] == (uint32_t)q) ? x : 0.0; }
      }
    }
  }
  for (int k = 0; k < 9; ++k) m[k] = wave_sum_all(m[k]);
  if (lane == 0) {
    AfHeader h;
    for (int k = 0; k < 9; ++k) h.mom[k] = m[k];
    h.dn = dn; h.pad = 0;
    sc.hdr[t] = h;
    if (dn == 0) { fractions[2 * t] = 0.5; fractions[2 * t + 1] = 0.5; }
  }
}

using AfGrid2 = AfGrid;
constexpr int AF2_SURVIVORS = 32;
__global__ __launch_bounds__(64) void af_search_kernel(const BcDesc* __restrict__ desc, const double* __restrict__ tp_all, const uint8_t* __restrict__ cls_all,
                                                       const AfHeader* __restrict__ hdr, AfGrid2 grid, double* __restrict__ fractions) {
  __shared__ double vals[100], f1[100], f2[100], f3[100];
  __shared__ double s_pv[AF2_SURVIVORS + 1][5];
  __shared__ uint32_t s_surv[AF2_SURVIVORS];
  __shared__ uint32_t s_nsurv;
  const uint32_t t = blockIdx.x, lane = threadIdx.x;
  const AfHeader h = hdr[t];
  const uint32_t dn = h.dn;
  if (dn == 0) return;  // (0.5, 0.5) written by af_prepare_kernel
  const uint64_t base = 4ull * desc[t].bc_off;
  const double* __restrict__ tp = tp_all + base;
  const uint8_t* __restrict__ cls = cls_all + base;
  const uint32_t terms = 4u * dn;
  const double n3 = h.mom[2], n4 = h.mom[3], s3 = h.mom[6], s4c = h.mom[7], s4c2 = 2.0 * h.mom[7], qtot = h.mom[8];
  for (uint32_t a = lane; a < 100; a += 64) {
    const double v = grid.vals[a];
    vals[a] = v;
    f1[a] = v * (h.mom[0] * v - 2.0 * h.mom[4]);
    f2[a] = v * (h.mom[1] * v - 2.0 * h.mom[5]);
    f3[a] = v * (n3 * v - 2.0 * s3);
  }
  if (lane == 0) s_nsurv = 0;
  __syncthreads();
  // closed-form SSE of candidate (pair, ic): the SAME expression in both passes (and in allelic_fraction_kernel)
  auto screen = [&](double fij, double sij, uint32_t ic) {
    const double vl = __dsub_rn(1.0, __dadd_rn(sij, vals[ic]));
    return (fij + f3[ic]) + vl * (n4 * vl - s4c2);
  };
  // minimum of the screen over the k of one pair: in k it is the parabola (n3 + n4) v^2 - 2 v (S3 + n4 c - S4) + const, c = 1 - sij, sampled
  // at the ascending vals[k]; its discrete minimum lies next to the vertex (or at the end of the range the vertex lies beyond)
  const double curv = n3 + n4;
  const bool flat = !(curv > 0.5);  // no position with two plain bases: the screen does not depend on k
  const double inv_curv = flat ? 0.0 : 1.0 / curv;
  auto pair_min = [&](double fij, double sij, uint32_t cnt) {
    double pm = 1e300;
    uint32_t lo = 0, hi = cnt;
    if (!flat) {
      const double vstar = (s3 + n4 * (1.0 - sij) - s4c) * inv_curv;
      const int kc = vstar > 0.0 ? (vstar < 1.5 ? (int)(vstar * 100.0 + 0.5) : 150) : 0;
      lo = (uint32_t)max(0, min(kc - 2, (int)cnt - 5));
      hi = min(cnt, lo + 5u);
    }
    for (uint32_t ic = lo; ic < hi; ++ic) {
      const double a = screen(fij, sij, ic);
      pm = a < pm ? a : pm;
    }
    return pm;
  };
  // ONE pass over the lane's pairs: the minimum of each (five candidates around the vertex), the lane's minimum, and the four pairs with
  // the smallest minima -- the only ones that can lie within the margin of the overall minimum unless the screen is (nearly) flat over
  // more than four of a lane's pairs, which the fourth-smallest shows: then the lane looks at all its pairs again, as the two-pass form did
  constexpr int kKeep = 4;
  double kv[kKeep];
  uint32_t kp[kKeep];
#pragma unroll
  for (int i = 0; i < kKeep; ++i) { kv[i] = 1e300; kp[i] = 0xffffffffu; }
  auto pair_of = [&](uint32_t p, double& fij, double& sij, uint32_t& code, uint32_t& cnt) {
    const uint32_t e = grid.pair8[p];  // i | j << 8 | k_count << 16
    const uint32_t ia = e & 0xffu, ib = (e >> 8) & 0xffu;
    cnt = e >> 16;
    code = ia * 100u + ib;
    sij = __dadd_rn(vals[ia], vals[ib]);
    fij = qtot + f1[ia] + f2[ib];
  };
  for (uint32_t p = lane; p < grid.npairs; p += 64) {
    double fij, sij;
    uint32_t code, cnt;
    pair_of(p, fij, sij, code, cnt);
    const double pm = pair_min(fij, sij, cnt);
    if (pm < kv[kKeep - 1]) {  // (rare after the first few pairs)
      kv[kKeep - 1] = pm; kp[kKeep - 1] = p;
#pragma unroll
      for (int i = kKeep - 1; i > 0; --i)
        if (kv[i] < kv[i - 1]) { const double tv = kv[i]; kv[i] = kv[i - 1]; kv[i - 1] = tv; const uint32_t tq = kp[i]; kp[i] = kp[i - 1]; kp[i - 1] = tq; }
    }
  }
  const double cut = wave_min_all(kv[0]) + kScreenMargin;
  // the reference's summation for one candidate (decompose.h:596-606); tp / cls are wave-uniform and read-only here: scalar loads, four
  // class bytes to a word
  const uint32_t* __restrict__ clsw = reinterpret_cast<const uint32_t*>(cls);  // (4 bc_off: a multiple of four)
  auto exact_sse = [&](const double* mine) {
    double sse = 0;
    for (uint32_t q = 0; q < terms; q += 4) {  // terms = 4 dn
      const uint32_t cw = clsw[q >> 2];
#pragma unroll
      for (uint32_t j = 0; j < 4; ++j) {
        const double df = __dsub_rn(mine[(cw >> (8u * j)) & 0xffu], tp[q + j]);
        sse = __dadd_rn(sse, __dmul_rn(df, df));
      }
    }
    return sse;
  };
  double my_sse = 1e300;
  uint32_t my_idx = 0xffffffffu;
  auto scan_pair = [&](uint32_t p) {  // every k of a pair that reaches the margin: listed, or evaluated right here once the list is full
    double fij, sij;
    uint32_t code, cnt;
    pair_of(p, fij, sij, code, cnt);
    for (uint32_t ic = 0; ic < cnt; ++ic) {
      if (screen(fij, sij, ic) > cut) continue;
      const uint32_t idx = code * 100u + ic;
      const uint32_t slot = atomicAdd(&s_nsurv, 1u);
      if (slot < (uint32_t)AF2_SURVIVORS) { s_surv[slot] = idx; continue; }
      const double vi = vals[idx / 10000], vj = vals[(idx / 100) % 100], vk = vals[idx % 100];
      const double pv[5] = {0.0, vi, vj, vk, __dsub_rn(1.0, __dadd_rn(__dadd_rn(vi, vj), vk))};
      const double sse = af_exact_sse(tp, cls, terms, pv);
      if (sse < my_sse || (sse == my_sse && idx < my_idx)) { my_sse = sse; my_idx = idx; }
    }
  };
  if (kv[kKeep - 1] <= cut) {  // more than kKeep of this lane's pairs may reach the margin
    for (uint32_t p = lane; p < grid.npairs; p += 64) {
      double fij, sij;
      uint32_t code, cnt;
      pair_of(p, fij, sij, code, cnt);
      if (pair_min(fij, sij, cnt) <= cut) scan_pair(p);
    }
  } else {
#pragma unroll
    for (int i = 0; i < kKeep; ++i)
      if (kv[i] <= cut) scan_pair(kp[i]);
  }
  __syncthreads();
  // exact stage: lane 0 sums the start point (0.5, 0.5, 0, 0) (decompose.h:586-591), lanes 1.. one listed survivor each
  const uint32_t nsurv = min(s_nsurv, (uint32_t)AF2_SURVIVORS);
  double sse0 = 0;
  if (lane <= nsurv) {
    double pv[5] = {0.0, 0.5, 0.5, 0.0, 0.0};
    if (lane > 0) {
      const uint32_t idx = s_surv[lane - 1];
      const double vi = vals[idx / 10000], vj = vals[(idx / 100) % 100], vk = vals[idx % 100];
      pv[1] = vi; pv[2] = vj; pv[3] = vk;
      pv[4] = __dsub_rn(1.0, __dadd_rn(__dadd_rn(vi, vj), vk));
    }
    for (int c = 0; c < 5; ++c) s_pv[lane][c] = pv[c];
    const double sse = exact_sse(s_pv[lane]);
    if (lane == 0) sse0 = sse;
    else if (sse < my_sse || (sse == my_sse && s_surv[lane - 1] < my_idx)) { my_sse = sse; my_idx = s_surv[lane - 1]; }
  }
  sse0 = __shfl(sse0, 0, 64);
  // lowest-index strict minimum below the start SSE == the reference's chain of strict improvements
  const double best = wave_min_all(my_idx != 0xffffffffu ? my_sse : 1e300);
  uint32_t bidx = (my_idx != 0xffffffffu && my_sse == best) ? my_idx : 0xffffffffu;
  for (int o = 32; o > 0; o >>= 1) { const uint32_t x = (uint32_t)__shfl_xor((int)bidx, o, 64); bidx = x < bidx ? x : bidx; }
  if (lane == 0) {
    double bi = 0.5, bj = 0.5;
    if (bidx != 0xffffffffu && best < sse0) { bi = vals[bidx / 10000]; bj = vals[(bidx / 100) % 100]; }
    fractions[2 * t] = bi;
    fractions[2 * t + 1] = bj;
  }
}

template <class T>
int to_device(tracyhip_ctx* ctx, DevBuf& b, const std::vector<T>& v, const T** out) {
  HIP_TRY(b.ensure(sizeof(T) * std::max<size_t>(v.size(), 1)));
  if (!v.empty()) HIP_TRY(hipMemcpyAsync(b.p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice, ctx->stream));
  *out = static_cast<const T*>(b.p);
  return TRACYHIP_OK;
}

// stage an in/out or output payload: DEVICE -> use as is; HOST -> device buffer (+ optional upload)
int stage_io(tracyhip_ctx* ctx, DevBuf& b, void* user, uint64_t bytes, int mem, bool upload, void** dev) {
  if (mem == TRACYHIP_MEM_DEVICE) { *dev = user; return TRACYHIP_OK; }
  HIP_TRY(b.ensure(bytes ? bytes : 1));
  if (upload && bytes) HIP_TRY(hipMemcpyAsync(b.p, user, bytes, hipMemcpyHostToDevice, ctx->stream));
  *dev = b.p;
  return TRACYHIP_OK;
}
int unstage(tracyhip_ctx* ctx, void* user, const void* dev, uint64_t bytes, int mem) {
  if (mem == TRACYHIP_MEM_DEVICE || bytes == 0) return TRACYHIP_OK;
  HIP_TRY(hipMemcpyAsync(user, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
  return TRACYHIP_OK;
}

uint64_t extent64(const uint64_t* off, const uint32_t* len, uint32_t n, uint64_t mult = 1) {
  uint64_t e = 0;
  for (uint32_t i = 0; i < n; ++i) e = std::max<uint64_t>(e, off[i] + mult * len[i]);
  return e;
}

}  // namespace

namespace tracyhip {

int launch_breakpoint(tracyhip_ctx* ctx, const BpDesc* d_desc, uint32_t n, uint32_t maxcol, const float* d_prof, BreakpointOut* d_out) {
  if (n == 0) return TRACYHIP_OK;
  const size_t lds = (((size_t)maxcol * 17 + 32) + 15) & ~(size_t)15;
  const bool global = lds > kLdsStageLimit;  // staging of a profile too long for LDS: a slice of a global scratch buffer per profile
  { int trc_ = timing_begin(ctx, TRACYHIP_TIMER_MISC, 0, 0); if (trc_) return trc_; }
  if (global) {
    HIP_TRY(ctx->d_bits.ensure(lds * (size_t)n));
    hipLaunchKernelGGL(breakpoint_kernel<true>, dim3(n), dim3(64), 0, ctx->stream, d_desc, d_prof, d_out, static_cast<char*>(ctx->d_bits.p), lds);
  } else {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(breakpoint_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(breakpoint_kernel<false>, dim3(n), dim3(64), lds, ctx->stream, d_desc, d_prof, d_out, nullptr, (size_t)0);
  }
  HIP_TRY(hipGetLastError());
  { int trc_ = timing_end(ctx); if (trc_) return trc_; }
  return TRACYHIP_OK;
}
int launch_homozygous(tracyhip_ctx* ctx, const RowsDesc* d_desc, const uint8_t* d_rows0, const uint8_t* d_rows1, uint32_t n,
                      BreakpointOut* d_bps, int32_t* d_status, const uint32_t* d_lens) {
  if (n == 0) return TRACYHIP_OK;
  { int trc_ = timing_begin(ctx, TRACYHIP_TIMER_MISC, 0, 0); if (trc_) return trc_; }
  hipLaunchKernelGGL(homozygous_kernel, dim3(n), dim3(64), 0, ctx->stream, d_desc, d_rows0, d_rows1, n, d_bps, d_status, d_lens);
  HIP_TRY(hipGetLastError());
  { int trc_ = timing_end(ctx); if (trc_) return trc_; }
  return TRACYHIP_OK;
}
int decompose_limits(int32_t maxindel, uint32_t maxbc) {
  if (maxindel < 1 || maxindel > kMaxIndelGlobal) return set_error(TRACYHIP_ERR_RANGE, "maxindel must be in [1, %d]", kMaxIndelGlobal);
  if (maxbc >= 2u * kMaxIndelGlobal) return set_error(TRACYHIP_ERR_RANGE, "a trace has %u basecalls; the scan tables hold < %d", maxbc, 2 * kMaxIndelGlobal);
  return TRACYHIP_OK;
}
int launch_decompose(tracyhip_ctx* ctx, const DecompArgs& a, const BreakpointOut* d_bps, uint32_t maxbc, uint64_t work_cells, uint64_t work_bytes) {
  if (a.ntraces == 0) return TRACYHIP_OK;
  int rc = decompose_limits(a.prm.maxindel, maxbc);
  if (rc) return rc;
  // scan state in LDS: 21 KB for maxindel <= 1024 and traces < 2048 basecalls (every Sanger run), 80 KB up to 4096 / 8191
  const bool large = a.prm.maxindel > kMaxIndelDev || maxbc >= 2u * kMaxIndelDev;
  const bool global = a.prm.maxindel > kMaxIndelLarge || maxbc >= 2u * kMaxIndelLarge;
  // the one-wave body with its working set in LDS (decompose_wave.h) where the size class allows it; what it is not provisioned for
  // (an alignment longer than the staged rows, trims outside the trace) it marks in a to-do word and decompose_kernel does
  DecompArgs rest = a;
  bool wave = !large && !global && !ctx->knobs.no_decomp_wave && a.prm.maxindel <= kMaxIndelDev;
  DecompWaveArgs wa{};
  DecompWaveLayout lay{};
  if (wave) {
    const uint32_t capB = std::max<uint32_t>(64u, (maxbc + 63u) & ~63u);
    const uint32_t capI = (uint32_t)a.prm.maxindel;
    // reference-row span: the columns of the trace's bases and gaps + what the widest deletion scan reaches behind them
    const uint32_t capL = (capB + capI + 256u + 63u) & ~63u;
    wa.caps = DecompWaveCaps{capL, capB, capI, std::min<uint32_t>(capI, capB / 2u + 1u)};
    lay = decomp_wave_layout(wa.caps);
    wave = decomp_wave_caps_ok(wa.caps) && lay.total <= 48u * 1024u;
  }
  if (wave) {
    if (!ctx->declut_ready) {
      static const std::vector<uint8_t> lut = [] { std::vector<uint8_t> t(kLutBytes); decomp_lut_build(t.data()); return t; }();
      HIP_TRY(ctx->d_declut.ensure(kLutBytes));
      HIP_TRY(hipMemcpyAsync(ctx->d_declut.p, lut.data(), kLutBytes, hipMemcpyHostToDevice, ctx->stream));
      ctx->declut_ready = true;
    }
    HIP_TRY(ctx->d_dectodo.ensure(sizeof(uint32_t) * (size_t)a.ntraces + 8 + sizeof(unsigned long long) * kDecompWaveStages * kDecompWaveClockRows));
    wa.a = a;
    wa.bps = d_bps;
    wa.lut = static_cast<const uint8_t*>(ctx->d_declut.p);
    wa.todo = static_cast<uint32_t*>(ctx->d_dectodo.p);
    rest.only = wa.todo;
#ifdef TRACY_PHASE_CLOCKS
    wa.clocks = reinterpret_cast<unsigned long long*>(static_cast<char*>(ctx->d_dectodo.p) + ((sizeof(uint32_t) * (size_t)a.ntraces + 7) & ~(size_t)7));
    HIP_TRY(hipMemsetAsync(wa.clocks, 0, sizeof(unsigned long long) * kDecompWaveStages * kDecompWaveClockRows, ctx->stream));
#endif
  }
  { int trc_ = timing_begin(ctx, TRACYHIP_TIMER_DECOMP, work_cells, work_bytes); if (trc_) return trc_; }
  if (wave) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(decompose_wave_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lay.total));
    hipLaunchKernelGGL(decompose_wave_kernel, dim3(a.ntraces), dim3(64), lay.total, ctx->stream, wa);
    HIP_TRY(hipGetLastError());
#ifdef TRACY_PHASE_CLOCKS
    {
      static std::vector<unsigned long long> rows(kDecompWaveStages * kDecompWaveClockRows);
      HIP_TRY(hipMemcpyAsync(rows.data(), wa.clocks, sizeof(unsigned long long) * rows.size(), hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(hipStreamSynchronize(ctx->stream));
      unsigned long long hc[kDecompWaveStages] = {};
      for (size_t i = 0; i < rows.size(); ++i) hc[i % kDecompWaveStages] += rows[i];
      fprintf(stderr, "tracyhip: decompose_wave_kernel cycles per trace by stage (loads + ballots, words + prefix sums, walk + bounds, staging, phase to the breakpoint, sets, scans, cut-off, picks, complex, apply):");
      for (int i = 0; i < 11; ++i) fprintf(stderr, " %.0f", (double)hc[i] / a.ntraces);
      fprintf(stderr, "  (LDS %u bytes)\n", lay.total);
    }
#endif
  }
  const DecompArgs& a_ = rest;
  if (global) {
    // scan state in global memory: up to 512 workgroups in flight, each with a slot of its own (0.7 GB)
    const uint32_t slots = std::min<uint32_t>(a.ntraces, 512u);
    HIP_TRY(ctx->d_band.ensure((size_t)slots * sizeof(DecompSharedT<kMaxIndelGlobal>)));
    hipLaunchKernelGGL(decompose_kernel_global, dim3(slots), dim3(64), 0, ctx->stream, a_, d_bps, static_cast<char*>(ctx->d_band.p));
  } else if (large) {
    const size_t lds = sizeof(DecompSharedT<kMaxIndelLarge>);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(decompose_kernel<kMaxIndelLarge>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(decompose_kernel<kMaxIndelLarge>, dim3(a.ntraces), dim3(64), lds, ctx->stream, a_, d_bps);
  } else {
    hipLaunchKernelGGL(decompose_kernel<kMaxIndelDev>, dim3(a.ntraces), dim3(64), sizeof(DecompSharedT<kMaxIndelDev>), ctx->stream, a_, d_bps);
  }
  HIP_TRY(hipGetLastError());
  { int trc_ = timing_end(ctx); if (trc_) return trc_; }
  return TRACYHIP_OK;
}
int launch_peaks(tracyhip_ctx* ctx, const BcDesc* d_desc, uint32_t n, uint32_t maxbc, const int32_t* d_sig, const int32_t* d_pos, int32_t* d_peaks) {
  if (n == 0 || maxbc == 0) return TRACYHIP_OK;
  { int trc_ = timing_begin(ctx, TRACYHIP_TIMER_MISC, 0, 0); if (trc_) return trc_; }
  hipLaunchKernelGGL(peaks_kernel, dim3((maxbc + 255) / 256, n), dim3(256), 0, ctx->stream, d_desc, d_sig, d_pos, reinterpret_cast<int4*>(d_peaks));
  HIP_TRY(hipGetLastError());
  { int trc_ = timing_end(ctx); if (trc_) return trc_; }
  return TRACYHIP_OK;
}
int launch_secdecomp(tracyhip_ctx* ctx, const BcDesc* d_desc, uint32_t n, uint32_t maxbc, const int32_t* d_peaks, const uint8_t* d_pri, const uint8_t* d_sec,
                     uint8_t* d_out) {
  if (n == 0 || maxbc == 0) return TRACYHIP_OK;
  { int trc_ = timing_begin(ctx, TRACYHIP_TIMER_MISC, 0, 0); if (trc_) return trc_; }
  hipLaunchKernelGGL(secdecomp_kernel, dim3((maxbc + 255) / 256, n), dim3(256), 0, ctx->stream, d_desc, reinterpret_cast<const int4*>(d_peaks), d_pri, d_sec, d_out);
  HIP_TRY(hipGetLastError());
  { int trc_ = timing_end(ctx); if (trc_) return trc_; }
  return TRACYHIP_OK;
}
// the trace-independent enumeration of the (i, j, k) grid, uploaded once per context
static int ensure_af_grid(tracyhip_ctx* ctx, AfGrid& g) {
  static const std::vector<uint32_t> tab = [] {  // thread-safe one-time initialisation
    std::vector<uint32_t> t;
    double vals[100];
    double x = 0;
    for (int a = 0; a < 100; ++a) { vals[a] = x; x = x + 0.01; }
    for (int ia = 0; ia < 100; ++ia)
      for (int ib = 0; ib < 100; ++ib) {
        const double sij = vals[ia] + vals[ib];
        if (!(sij <= 1.0)) continue;
        uint32_t cnt = 0;
        for (int ic = 0; ic < 100; ++ic) {
          if (!(sij + vals[ic] <= 1.0)) break;  // vals ascend: later k fail as well
          ++cnt;
        }
        if (cnt) t.push_back((uint32_t)(ia * 100 + ib) | (cnt << 16));
      }
    std::stable_sort(t.begin(), t.end(), [](uint32_t a, uint32_t b) { return (a >> 16) > (b >> 16); });
    return t;
  }();
  if (tab.size() > (size_t)AF_STEPS * AF_THREADS) return set_error(TRACYHIP_ERR_RANGE, "allelicFraction grid larger than the unrolled schedule");
  static const std::vector<double> vals = [] {  // for (double i = 0; i <= 1; i += 0.01): the same doubles the kernels build
    std::vector<double> v(100);
    double x = 0;
    for (int a = 0; a < 100; ++a) { v[a] = x; x = x + 0.01; }
    return v;
  }();
  static const std::vector<uint32_t> tab8 = [] {
    std::vector<uint32_t> t(tab.size());
    for (size_t i = 0; i < tab.size(); ++i) t[i] = ((tab[i] & 0xffffu) / 100u) | (((tab[i] & 0xffffu) % 100u) << 8) | ((tab[i] >> 16) << 16);
    return t;
  }();
  const size_t pair_bytes = (tab.size() * sizeof(uint32_t) + 7) & ~(size_t)7, vals_bytes = vals.size() * sizeof(double);
  if (!ctx->aftab_ready) {
    HIP_TRY(ctx->d_aftab.ensure(2 * pair_bytes + vals_bytes));
    char* base = static_cast<char*>(ctx->d_aftab.p);
    HIP_TRY(hipMemcpyAsync(base, tab.data(), tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(base + pair_bytes, vals.data(), vals_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(base + pair_bytes + vals_bytes, tab8.data(), tab8.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    ctx->aftab_ready = true;
  }
  g.npairs = (uint32_t)tab.size();
  g.pair = static_cast<const uint32_t*>(ctx->d_aftab.p);
  g.vals = reinterpret_cast<const double*>(static_cast<const char*>(ctx->d_aftab.p) + pair_bytes);
  g.pair8 = reinterpret_cast<const uint32_t*>(static_cast<const char*>(ctx->d_aftab.p) + pair_bytes + vals_bytes);
  return TRACYHIP_OK;
}

int launch_allelic_fraction(tracyhip_ctx* ctx, const BcDesc* d_desc, uint32_t n, uint32_t maxbc, const int32_t* d_peaks_, const uint8_t* d_pri, const uint8_t* d_sec,
                            uint32_t trim_left, uint32_t trim_right, double* d_out, uint64_t work_bytes, uint64_t bext) {
  const int4* d_peaks = reinterpret_cast<const int4*>(d_peaks_);
  if (n == 0) return TRACYHIP_OK;
  const size_t lds = (((size_t)maxbc * 36 + 64) + 15) & ~(size_t)15;  // tp (4 doubles per basecall) + class bytes
  const bool global = lds > kLdsStageLimit;
  AfGrid grid{};
  int rc = ensure_af_grid(ctx, grid);
  if (rc) return rc;
  // two launches (af_prepare_kernel / af_search_kernel) when the extent of the basecall arrays is known and the scratch can be had
  bool split = bext != 0 && !ctx->knobs.no_af_split;
  AfScratch sc{};
  if (split) {
    const size_t tp_bytes = (size_t)bext * 32, cls_bytes = ((size_t)bext * 4 + 63) & ~(size_t)63, hdr_bytes = sizeof(AfHeader) * (size_t)n;
    if (ctx->d_afscratch.ensure(tp_bytes + cls_bytes + hdr_bytes) != hipSuccess) { (void)hipGetLastError(); split = false; }
    else {
      char* p = static_cast<char*>(ctx->d_afscratch.p);
      sc.tp = reinterpret_cast<double*>(p);
      sc.cls = reinterpret_cast<uint8_t*>(p + tp_bytes);
      sc.hdr = reinterpret_cast<AfHeader*>(p + tp_bytes + cls_bytes);
    }
  }
  { int trc_ = timing_begin(ctx, TRACYHIP_TIMER_AFRAC, 0, work_bytes); if (trc_) return trc_; }
  if (split) {
    hipLaunchKernelGGL(af_prepare_kernel, dim3(n), dim3(64), 0, ctx->stream, d_desc, d_peaks, d_pri, d_sec, trim_left, trim_right, sc, d_out);
    hipLaunchKernelGGL(af_search_kernel, dim3(n), dim3(64), 0, ctx->stream, d_desc, sc.tp, sc.cls, sc.hdr, grid, d_out);
  } else if (global) {
    HIP_TRY(ctx->d_bits.ensure(lds * (size_t)n));
    hipLaunchKernelGGL(allelic_fraction_kernel<true>, dim3(n), dim3(AF_THREADS), 0, ctx->stream, d_desc, d_peaks, d_pri, d_sec, trim_left,
                       trim_right, grid, d_out, static_cast<char*>(ctx->d_bits.p), lds);
  } else {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(allelic_fraction_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(allelic_fraction_kernel<false>, dim3(n), dim3(AF_THREADS), lds, ctx->stream, d_desc, d_peaks, d_pri, d_sec, trim_left,
                       trim_right, grid, d_out, nullptr, (size_t)0);
  }
  HIP_TRY(hipGetLastError());
  { int trc_ = timing_end(ctx); if (trc_) return trc_; }
  return TRACYHIP_OK;
}

}  // namespace tracyhip

extern "C" {

int tracyhip_find_breakpoint(tracyhip_ctx* ctx, const tracyhip_seqset* profiles, int mem, tracyhip_breakpoint* out) {
  int rc = ctx_begin(ctx);
  if (rc) return rc;
  if (!profiles || !out) return set_error(TRACYHIP_ERR_ARG, "null argument");
  if (profiles->kind != TRACYHIP_SEQ_PROFILE) return set_error(TRACYHIP_ERR_ARG, "findBreakpoint takes profiles");
  const uint32_t n = profiles->count;
  if (n == 0) return TRACYHIP_OK;
  static_assert(sizeof(tracyhip_breakpoint) == sizeof(BreakpointOut), "layout");
  hipStream_t st = ctx->stream;
  const void* d_prof;
  if ((rc = stage_in(ctx, ctx->d_in1, profiles->data, seqset_extent(*profiles) * 4, mem, &d_prof))) return rc;
  std::vector<BpDesc> hd(n);
  uint32_t maxcol = 0;
  for (uint32_t i = 0; i < n; ++i) { hd[i] = BpDesc{profiles->offset[i], profiles->length[i], profiles->length[i]}; maxcol = std::max(maxcol, profiles->length[i]); }
  const BpDesc* dd;
  if ((rc = to_device(ctx, ctx->d_desc, hd, &dd))) return rc;
  void* d_out;
  if ((rc = stage_io(ctx, ctx->d_tmp[0], out, sizeof(BreakpointOut) * (size_t)n, mem, false, &d_out))) return rc;
  if ((rc = launch_breakpoint(ctx, dd, n, maxcol, static_cast<const float*>(d_prof), static_cast<BreakpointOut*>(d_out)))) return rc;
  if ((rc = unstage(ctx, out, d_out, sizeof(BreakpointOut) * (size_t)n, mem))) return rc;
  HIP_TRY(hipStreamSynchronize(st));
  return TRACYHIP_OK;
}

int tracyhip_find_homozygous_breakpoint(tracyhip_ctx* ctx, uint32_t ntraces, const uint8_t* rows0, const uint8_t* rows1,
                                        const uint64_t* rows_offset, const uint32_t* rows_len, int mem,
                                        tracyhip_breakpoint* bps, int32_t* status) {
  int rc = ctx_begin(ctx);
  if (rc) return rc;
  if (ntraces == 0) return TRACYHIP_OK;
  if (!rows0 || !rows1 || !rows_offset || !rows_len || !bps || !status) return set_error(TRACYHIP_ERR_ARG, "null argument");
  hipStream_t st = ctx->stream;
  const uint64_t ext = extent64(rows_offset, rows_len, ntraces);
  const void *d_r0, *d_r1;
  if ((rc = stage_in(ctx, ctx->d_rows0, rows0, ext, mem, &d_r0))) return rc;
  if ((rc = stage_in(ctx, ctx->d_rows1, rows1, ext, mem, &d_r1))) return rc;
  std::vector<RowsDesc> hd(ntraces);
  for (uint32_t i = 0; i < ntraces; ++i) hd[i] = RowsDesc{rows_offset[i], rows_len[i], 0};
  const RowsDesc* dd;
  if ((rc = to_device(ctx, ctx->d_desc, hd, &dd))) return rc;
  void *d_bp, *d_stat;
  if ((rc = stage_io(ctx, ctx->d_tmp[0], bps, sizeof(BreakpointOut) * (size_t)ntraces, mem, true, &d_bp))) return rc;
  if ((rc = stage_io(ctx, ctx->d_tmp[1], status, sizeof(int32_t) * (size_t)ntraces, mem, false, &d_stat))) return rc;
  if ((rc = launch_homozygous(ctx, dd, static_cast<const uint8_t*>(d_r0), static_cast<const uint8_t*>(d_r1), ntraces,
                              static_cast<BreakpointOut*>(d_bp), static_cast<int32_t*>(d_stat))))
    return rc;
  if ((rc = unstage(ctx, bps, d_bp, sizeof(BreakpointOut) * (size_t)ntraces, mem))) return rc;
  if ((rc = unstage(ctx, status, d_stat, sizeof(int32_t) * (size_t)ntraces, mem))) return rc;
  HIP_TRY(hipStreamSynchronize(st));
  return TRACYHIP_OK;
}

int tracyhip_decompose_alleles(tracyhip_ctx* ctx, const tracyhip_basecalls* bc, const uint8_t* rows0, const uint8_t* rows1,
                               const uint64_t* rows_offset, const uint32_t* rows_len, const tracyhip_breakpoint* bps,
                               const uint32_t* refslice_len, const tracyhip_decomp_params* prm, int mem,
                               int32_t* dcp_indel, int32_t* dcp_err, const uint64_t* dcp_offset, tracyhip_decomp_status* status) {
  int rc = ctx_begin(ctx);
  if (rc) return rc;
  if (!bc || !prm) return set_error(TRACYHIP_ERR_ARG, "null argument");
  const uint32_t n = bc->ntraces;
  if (n == 0) return TRACYHIP_OK;
  if (!rows0 || !rows1 || !rows_offset || !rows_len || !bps || !refslice_len || !dcp_indel || !dcp_err || !dcp_offset || !status ||
      !bc->primary || !bc->secondary || !bc->bc_offset || !bc->bc_len)
    return set_error(TRACYHIP_ERR_ARG, "null argument");
  uint32_t maxbc_all = 0;
  for (uint32_t i = 0; i < n; ++i) maxbc_all = std::max(maxbc_all, bc->bc_len[i]);
  if ((rc = decompose_limits(prm->maxindel, maxbc_all))) return rc;
  static_assert(sizeof(tracyhip_decomp_status) == sizeof(DecompOut), "layout");
  hipStream_t st = ctx->stream;
  const uint64_t rext = extent64(rows_offset, rows_len, n), bext = extent64(bc->bc_offset, bc->bc_len, n);
  uint64_t dext = 0;
  std::vector<DecompDesc> hd(n);
  for (uint32_t i = 0; i < n; ++i) {

    hd[i] = DecompDesc{rows_offset[i], bc->bc_offset[i], dcp_offset[i], rows_len[i], bc->bc_len[i], refslice_len[i], 0};
    dext = std::max<uint64_t>(dext, dcp_offset[i] + 2ull * prm->maxindel + 2);
  }
  const void *d_r0, *d_r1, *d_bp;
  if ((rc = stage_in(ctx, ctx->d_rows0, rows0, rext, mem, &d_r0))) return rc;
  if ((rc = stage_in(ctx, ctx->d_rows1, rows1, rext, mem, &d_r1))) return rc;
  if ((rc = stage_in(ctx, ctx->d_tmp[0], bps, sizeof(BreakpointOut) * (size_t)n, mem, &d_bp))) return rc;
  void *d_pri, *d_sec, *d_di, *d_de, *d_stat;
  if ((rc = stage_io(ctx, ctx->d_tmp[1], bc->primary, bext, mem, true, &d_pri))) return rc;
  if ((rc = stage_io(ctx, ctx->d_tmp[2], bc->secondary, bext, mem, true, &d_sec))) return rc;
  if ((rc = stage_io(ctx, ctx->d_tmp[3], dcp_indel, dext * 4, mem, false, &d_di))) return rc;
  if ((rc = stage_io(ctx, ctx->d_tmp[4], dcp_err, dext * 4, mem, false, &d_de))) return rc;
  if ((rc = stage_io(ctx, ctx->d_tmp[5], status, sizeof(DecompOut) * (size_t)n, mem, false, &d_stat))) return rc;
  const DecompDesc* dd;
  if ((rc = to_device(ctx, ctx->d_desc, hd, &dd))) return rc;
  DecompArgs a{};
  a.desc = dd;
  a.rows0 = static_cast<const uint8_t*>(d_r0);
  a.rows1 = static_cast<const uint8_t*>(d_r1);
  a.primary = static_cast<uint8_t*>(d_pri);
  a.secondary = static_cast<uint8_t*>(d_sec);
  a.dcp_indel = static_cast<int32_t*>(d_di);
  a.dcp_err = static_cast<int32_t*>(d_de);
  a.out = static_cast<DecompOut*>(d_stat);
  a.prm = DecompParams{prm->trim_left, prm->trim_right, prm->maxindel, prm->madc};
  a.ntraces = n;
  if ((rc = launch_decompose(ctx, a, static_cast<const BreakpointOut*>(d_bp), maxbc_all))) return rc;
  if ((rc = unstage(ctx, bc->primary, d_pri, bext, mem))) return rc;
  if ((rc = unstage(ctx, bc->secondary, d_sec, bext, mem))) return rc;
  if ((rc = unstage(ctx, dcp_indel, d_di, dext * 4, mem))) return rc;
  if ((rc = unstage(ctx, dcp_err, d_de, dext * 4, mem))) return rc;
  if ((rc = unstage(ctx, status, d_stat, sizeof(DecompOut) * (size_t)n, mem))) return rc;
  HIP_TRY(hipStreamSynchronize(st));
  return TRACYHIP_OK;
}

static int bc_descs(tracyhip_ctx* ctx, const tracyhip_basecalls* bc, int mem, const BcDesc** dd, const int32_t** d_peaks, uint64_t* bext) {
  // descriptors + the peak table of the batch: the caller's (tracyhip_basecalls::peaks), or built here from the chromatograms
  const uint32_t n = bc->ntraces;
  if (!bc->bc_offset || !bc->bc_len) return set_error(TRACYHIP_ERR_ARG, "null basecall arrays");
  if (!bc->peaks && (!bc->signal || !bc->signal_offset || !bc->nsamples || !bc->bcpos))
    return set_error(TRACYHIP_ERR_ARG, "null basecall arrays: neither a peak table nor signal + bcpos");
  std::vector<BcDesc> hd(n);
  uint64_t sext = 0;
  uint32_t maxbc = 0;
  for (uint32_t i = 0; i < n; ++i) {
    hd[i] = BcDesc{bc->peaks ? 0ull : bc->signal_offset[i], bc->bc_offset[i], bc->peaks ? 0u : bc->nsamples[i], bc->bc_len[i]};
    if (!bc->peaks) sext = std::max<uint64_t>(sext, bc->signal_offset[i] + 4ull * bc->nsamples[i]);
    maxbc = std::max(maxbc, bc->bc_len[i]);
  }
  *bext = extent64(bc->bc_offset, bc->bc_len, n);
  int rc;
  if ((rc = to_device(ctx, ctx->d_desc, hd, dd))) return rc;
  const void* p = nullptr;
  if (bc->peaks) {
    if ((rc = stage_in(ctx, ctx->d_in1, bc->peaks, *bext * 16, mem, &p))) return rc;
    *d_peaks = static_cast<const int32_t*>(p);
    return TRACYHIP_OK;
  }
  const void *d_sig, *d_pos;
  if ((rc = stage_in(ctx, ctx->d_in1, bc->signal, sext * 4, mem, &d_sig))) return rc;
  if ((rc = stage_in(ctx, ctx->d_in2, bc->bcpos, *bext * 4, mem, &d_pos))) return rc;
  HIP_TRY(ctx->d_tmp[5].ensure(*bext * 16 + 16));
  if ((rc = launch_peaks(ctx, *dd, n, maxbc, static_cast<const int32_t*>(d_sig), static_cast<const int32_t*>(d_pos), static_cast<int32_t*>(ctx->d_tmp[5].p)))) return rc;
  *d_peaks = static_cast<const int32_t*>(ctx->d_tmp[5].p);
  return TRACYHIP_OK;
}

int tracyhip_secondary_decomposed(tracyhip_ctx* ctx, const tracyhip_basecalls* bc, int mem, uint8_t* secdecomp) {
  int rc = ctx_begin(ctx);
  if (rc) return rc;
  if (!bc || !secdecomp) return set_error(TRACYHIP_ERR_ARG, "null argument");
  const uint32_t n = bc->ntraces;
  if (n == 0) return TRACYHIP_OK;
  if (!bc->primary || !bc->secondary) return set_error(TRACYHIP_ERR_ARG, "null basecalls");
  const BcDesc* dd;
  const void *d_pri, *d_sec;
  const int32_t* d_peaks;
  uint64_t bext;
  if ((rc = bc_descs(ctx, bc, mem, &dd, &d_peaks, &bext))) return rc;
  if ((rc = stage_in(ctx, ctx->d_tmp[1], bc->primary, bext, mem, &d_pri))) return rc;
  if ((rc = stage_in(ctx, ctx->d_tmp[2], bc->secondary, bext, mem, &d_sec))) return rc;
  void* d_out;
  if ((rc = stage_io(ctx, ctx->d_tmp[3], secdecomp, bext, mem, false, &d_out))) return rc;
  uint32_t maxbc = 0;
  for (uint32_t i = 0; i < n; ++i) maxbc = std::max(maxbc, bc->bc_len[i]);
  if ((rc = launch_secdecomp(ctx, dd, n, maxbc, d_peaks,
                             static_cast<const uint8_t*>(d_pri), static_cast<const uint8_t*>(d_sec), static_cast<uint8_t*>(d_out))))
    return rc;
  if ((rc = unstage(ctx, secdecomp, d_out, bext, mem))) return rc;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return TRACYHIP_OK;
}

int tracyhip_allelic_fraction(tracyhip_ctx* ctx, const tracyhip_basecalls* bc, const uint8_t* secdecomp, uint32_t trim_left,
                              uint32_t trim_right, int mem, double* fractions) {
  int rc = ctx_begin(ctx);
  if (rc) return rc;
  if (!bc || !secdecomp || !fractions) return set_error(TRACYHIP_ERR_ARG, "null argument");
  const uint32_t n = bc->ntraces;
  if (n == 0) return TRACYHIP_OK;
  if (!bc->primary) return set_error(TRACYHIP_ERR_ARG, "null basecalls");
  const BcDesc* dd;
  const void *d_pri, *d_sec;
  const int32_t* d_peaks;
  uint64_t bext;
  if ((rc = bc_descs(ctx, bc, mem, &dd, &d_peaks, &bext))) return rc;
  if ((rc = stage_in(ctx, ctx->d_tmp[1], bc->primary, bext, mem, &d_pri))) return rc;
  if ((rc = stage_in(ctx, ctx->d_tmp[2], secdecomp, bext, mem, &d_sec))) return rc;
  void* d_out;
  if ((rc = stage_io(ctx, ctx->d_tmp[3], fractions, sizeof(double) * 2 * (size_t)n, mem, false, &d_out))) return rc;
  uint32_t maxbc = 0;
  for (uint32_t i = 0; i < n; ++i) maxbc = std::max(maxbc, bc->bc_len[i]);
  if ((rc = launch_allelic_fraction(ctx, dd, n, maxbc, d_peaks,
                                    static_cast<const uint8_t*>(d_pri), static_cast<const uint8_t*>(d_sec), trim_left, trim_right,
                                    static_cast<double*>(d_out), 0, bext)))
    return rc;
  if ((rc = unstage(ctx, fractions, d_out, sizeof(double) * 2 * (size_t)n, mem))) return rc;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return TRACYHIP_OK;
}

}  // extern "C"
