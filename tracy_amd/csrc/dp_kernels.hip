// dp_kernels.hip -- gfx950 __global__ wrappers and launchers for the wave bodies in dp_kernels.h.
// One 64-thread workgroup (= one wave) per pair; the launch grid is the number of pairs in the bucket
// (thousands), so every CU holds several independent waves and nothing is shared between workgroups
// (no inter-workgroup traffic => no XCD placement concerns; block b lands on XCD b % 8 and only ever
// touches its own pair's inputs and traceback stripe).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "dp_kernels.h"
#include "launch.h"

namespace tracyhip {

// development knob: extra dynamic LDS per workgroup (bytes) to study occupancy sensitivity
static uint32_t lds_pad() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("TRACYHIP_LDS_PAD"); v = e ? atoi(e) : 0; }
  return (uint32_t)v;
}

struct DeviceWave {
  __device__ __forceinline__ uint32_t lane() const { return threadIdx.x; }
  // lane L <- lane L-1 (wave_shr:1); lane 0 has no source lane and reads 0 (bound_ctrl), the caller overwrites it
  __device__ __forceinline__ int32_t shift_up(int32_t x) const {
    return __builtin_amdgcn_update_dpp(0, x, 0x138, 0xf, 0xf, true);
  }
  // lane L <- lane L-1 inside a row of sixteen lanes (row_shr:1); the first lane of a row reads 0
  __device__ __forceinline__ int32_t shift_up_row(int32_t x) const {
    return __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);
  }
  // same shift, lane 0 keeps `first`: the DPP `old` operand is what a lane without a source keeps (bound_ctrl off)
  __device__ __forceinline__ int32_t shift_up_or(int32_t x, int32_t first) const {
    return __builtin_amdgcn_update_dpp(first, x, 0x138, 0xf, 0xf, false);
  }
  __device__ __forceinline__ uint64_t ballot(bool p) const { return __ballot(p); }
  __device__ __forceinline__ uint32_t bcast(uint32_t x, uint32_t src_lane) const { return (uint32_t)__shfl((int)x, (int)src_lane, 64); }
  __device__ __forceinline__ void sync() const { __syncthreads(); }
  // global memory written by some lanes of this wave is read by others afterwards (band buffer, pass hand-over): the
  // workgroup IS the wave, so workgroup scope is all that is needed.  An agent-scope fence (__threadfence) writes back
  // and invalidates the XCD's L2 on this multi-XCD part -- hundreds of microseconds per call.
  __device__ __forceinline__ void sync_global() const {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  __device__ __forceinline__ char* lds() const {
    extern __shared__ __attribute__((aligned(16))) char tracy_smem[];
    return tracy_smem;
  }
};

// COMPACT (16-bit query-profile sweeps only): the form with the four-code table, for references of A C G T (gotoh_narrow_qp_body)
// The workgroup that wrote a pair's traceback words walks them at once (DpArgs::walk_ops set): the words are still in the
// cache, the walk of an early pair fills the tail of the sweep of a late one, and one launch less per stage.  The words were
// written by the lanes of this very wave: a workgroup-scope fence orders them before the walker's loads.
__device__ __forceinline__ void walk_own_pair(DeviceWave& w, const DpArgs& a, int K) {
  if (!a.walk_ops) return;
  w.sync_global();
  WalkArgs wa{};
  wa.pairs = a.pairs; wa.bits = a.bits; wa.ops = a.walk_ops; wa.ops_off = a.walk_ops_off; wa.ops_len = a.walk_ops_len; wa.err = a.err; wa.K = K;
  gotoh_walk_wave<DeviceWave>(w, wa, blockIdx.x);
}

template <int K, int MODE, bool TRACE, bool NARROW = false, bool COMPACT = false>
__global__ __launch_bounds__(64) void gotoh_kernel(DpArgs a) {
  DeviceWave w;
  gotoh_body<DeviceWave, K, MODE, TRACE, NARROW, false, 0, COMPACT>(w, a, blockIdx.x);
  if constexpr (TRACE) walk_own_pair(w, a, K);
}
// profile x profile with the number of substitution terms fixed per launch (NT = 4: PAIR_ROW4_ZERO pairs, 5: the rest): one
// body per kernel keeps the register count where four waves per SIMD fit
// A16 (score kernels): 16-bit cell arithmetic (gotoh_body's second form for MODE_PROF)
template <int K, bool TRACE, int NT, bool A16 = false>
__global__ __launch_bounds__(64) void gotoh_prof_kernel(DpArgs a) {
  DeviceWave w;
  gotoh_body<DeviceWave, K, MODE_PROF, TRACE, false, false, NT, A16>(w, a, blockIdx.x);
  if constexpr (TRACE) walk_own_pair(w, a, K);
}

// checkpointed score pass (wavefront checkpoints + last row) and the band traceback that consumes them
template <int K, int MODE, bool NARROW, bool COMPACT = false>
__global__ __launch_bounds__(64) void gotoh_ckpt_kernel(DpArgs a) {
  DeviceWave w;
  gotoh_body<DeviceWave, K, MODE, false, NARROW, true, 0, COMPACT>(w, a, blockIdx.x);
}
// origin-tracking sweep (string x string): score + the two ends of the alignment, no traceback words
template <int K, int TABLE = 0, int NC = 6>
__global__ __launch_bounds__(64) void gotoh_origin_kernel(DpArgs a) {
  DeviceWave w;
  gotoh_origin_body<DeviceWave, K, TABLE, NC>(w, a, blockIdx.x);
}
// one launch, two kinds of workgroups: blocks [0, nfull) run the checkpointed 16-bit score sweep of `full`, the rest the
// prefix bound of `pre` (GL lanes per pair) -- the short prefix workgroups fill the tail of the long sweeps
// (the prefix workgroups come first: they walk as many columns as a sweep does, with fewer rows -- started last they would be the tail)
#ifdef TRACY_SWEEP_WAVES
#define TRACY_SWEEP_ATTR __attribute__((amdgpu_waves_per_eu(TRACY_SWEEP_WAVES)))
#else
#define TRACY_SWEEP_ATTR
#endif
template <int K, int GL, bool COMPACT = false, int KP = K>
__global__ __launch_bounds__(64) TRACY_SWEEP_ATTR void gotoh_ckpt_prefix_kernel(DpArgs full, uint32_t nfull, DpArgs pre, uint32_t npre) {
  DeviceWave w;
  const uint32_t ngroups = (npre + 64u / GL - 1u) / (64u / GL);
  if (blockIdx.x < ngroups) gotoh_prefix_body<DeviceWave, KP, GL, COMPACT>(w, pre, blockIdx.x * (64u / GL), npre);
  else {
#ifdef TRACY_SWEEP_STAGGER
    // (experiment: the waves of a launch's first round start within microseconds of each other and run the same instruction stream)
    for (uint32_t h = (blockIdx.x * 2654435761u) >> 27; h; --h) __builtin_amdgcn_s_sleep(TRACY_SWEEP_STAGGER);
#endif
    gotoh_body<DeviceWave, K, MODE_QP, false, true, true, 0, COMPACT>(w, full, blockIdx.x - ngroups);
  }
}
// prefix bound of the semiglobal score: GL lanes per pair, 64/GL pairs per workgroup
template <int K, int GL, bool COMPACT = false, bool STRINGS = false>
__global__ __launch_bounds__(64) void gotoh_prefix_kernel(DpArgs a, uint32_t npairs) {
  DeviceWave w;
  gotoh_prefix_body<DeviceWave, K, GL, COMPACT, STRINGS>(w, a, blockIdx.x * (64u / GL), npairs);
}
// at least three waves per SIMD: the K = 15 / 16 instantiations would otherwise settle at 190-200 VGPRs and two waves, and a
// wave issues a VALU instruction only every ~4.5 cycles (6-24 spilled registers outside the sweep: band traceback 5.5 -> 5.0 ms)
template <int K, int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3))) void gotoh_band_kernel(DpArgs a, WalkArgs wa) {
  DeviceWave w;
  gotoh_band_trace_body<DeviceWave, K, MODE>(w, a, wa, blockIdx.x);
}

template <int K, int MODE, bool TRACE>
__global__ __launch_bounds__(64) void needle_kernel(DpArgs a) {
  DeviceWave w;
  needle_body<DeviceWave, K, MODE, TRACE>(w, a, blockIdx.x);
}

__global__ __launch_bounds__(64) void gotoh_walk_kernel(WalkArgs a) {
  DeviceWave w;
  gotoh_walk_wave<DeviceWave>(w, a, blockIdx.x);
}

__global__ __launch_bounds__(64) void needle_walk_kernel(WalkArgs a, const uint32_t* bits32) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < a.npairs) needle_walk_one(a, bits32, i);
}

// _createAlignment (align.h:196-223 / 254-293): one lane per output column, forward order.
// ops are in push order, so column ai corresponds to ops[L-1-ai].  One wave per pair: 64 columns per round,
// the consumed-row / consumed-column counts of a column are prefix popcounts of the wave's ballots.
// The common case of `tracy decompose` (a trace with a shift: row 0 only says where the trace has bases; row 1 the characters of a string
// reference) without a divergent branch: round 5's counters had the kernel at 9 500 scalar and 3 200 vector instructions per trace, 38 % of
// its wave cycles waiting for an instruction -- every conditional load of the general form below is a branch of its own.  Here the op
// bytes and the reference bytes are loaded unconditionally from clamped indices and everything else is a select.
__device__ __forceinline__ void alignment_rows_gaps_only(const RowsArgs& a, const PairDesc& d, const uint8_t* __restrict__ ops, uint8_t* __restrict__ r0,
                                                         uint8_t* __restrict__ r1, uint32_t L, uint32_t lane) {
  const bool rc = a.a2_revcomp_flag && (d.flags & PAIR_A2_REVCOMP);
  const bool onehot = a.a2_onehot != 0;
  const uint8_t* __restrict__ ref = static_cast<const uint8_t*>(a.a2) + d.a2_off;
  const uint32_t nlast = d.n ? d.n - 1u : 0u;
  const bool have_ref = d.n != 0u;  // (wave-uniform; without columns no op takes one)
  const uint64_t below = (1ull << lane) - 1ull;
  uint32_t col_base = 0;
  constexpr uint32_t kBatch = 8;
  for (uint32_t base0 = 0; base0 < L; base0 += 64 * kBatch) {
    uint8_t opb[kBatch];
#pragma unroll
    for (uint32_t k = 0; k < kBatch; ++k) {
      const uint32_t ai = base0 + 64 * k + lane;
      opb[k] = ops[ai < L ? L - 1 - ai : 0u];
    }
    uint32_t colk[kBatch];
    uint32_t take = 0;  // bit k: takes a row, bit 8 + k: takes a column
#pragma unroll
    for (uint32_t k = 0; k < kBatch; ++k) {
      const uint32_t ai = base0 + 64 * k + lane;
      const bool active = ai < L;
      const bool takes_row = active & (opb[k] != 'h'), takes_col = active & (opb[k] != 'v');
      const uint64_t mcol = __ballot(takes_col);
      colk[k] = col_base + (uint32_t)__popcll(mcol & below);
      col_base += (uint32_t)__popcll(mcol);
      take |= (takes_row ? 1u : 0u) << k | (takes_col ? 1u : 0u) << (8 + k);
    }
    uint8_t c1k[kBatch];
#pragma unroll
    for (uint32_t k = 0; k < kBatch; ++k) {
      const uint32_t c = colk[k] < nlast ? colk[k] : nlast;
      c1k[k] = have_ref ? ref[rc ? nlast - c : c] : (uint8_t)'-';
    }
#pragma unroll
    for (uint32_t k = 0; k < kBatch; ++k) {
      const uint32_t base = base0 + 64 * k;
      if (base >= L) break;  // (wave-uniform)
      const uint32_t ai = base + lane;
      uint8_t c1 = c1k[k];
      if (rc) c1 = complement_char(c1);
      if (onehot) {  // consensus character of _createProfile(string) (align.h:121-136, 254-270): base_code 0..3 -> A C G T, N and '-' -> N, others -> A
        const uint8_t up = c1 & 0xDFu;
        uint8_t r = 'A';
        r = up == 'C' ? 'C' : r;
        r = up == 'G' ? 'G' : r;
        r = up == 'T' ? 'T' : r;
        r = (up == 'N' || c1 == '-') ? 'N' : r;
        c1 = r;
      }
      const uint8_t o1 = ((take >> (8 + k)) & 1u) ? c1 : (uint8_t)'-';
      const uint8_t o0 = ((take >> k) & 1u) ? (uint8_t)'N' : (uint8_t)'-';
      if (ai < L) { r0[ai] = o0; r1[ai] = o1; }
    }
  }
}

__global__ __launch_bounds__(64) void alignment_rows_kernel(RowsArgs a) {
  const uint32_t i = blockIdx.x;
  const uint32_t lane = threadIdx.x;
  const PairDesc d = a.pairs[i];
  const uint64_t off = a.ops_off[d.out];
  const uint32_t L = a.ops_len[d.out];
  const uint8_t* ops = a.ops + off;
  uint8_t* r0 = a.rows0 + off;
  uint8_t* r1 = a.rows1 + off;
  const bool gaps_only = a.row0_gaps_only && a.row0_gaps_only[(size_t)d.out * a.row0_gaps_only_stride] != 0;
#ifndef TRACY_ROWS_NO_FAST
  if (gaps_only && !a.a2_profile) { alignment_rows_gaps_only(a, d, ops, r0, r1, L, lane); return; }
#endif
  const uint64_t below = (1ull << lane) - 1ull;
  uint32_t row_base = 0, col_base = 0;
  constexpr uint32_t kBatch = 8;  // rounds whose op bytes are requested together: one wait per 512 columns instead of one per 64
  for (uint32_t base0 = 0; base0 < L; base0 += 64 * kBatch) {
    uint8_t opb[kBatch];
#pragma unroll
    for (uint32_t k = 0; k < kBatch; ++k) {
      const uint32_t ai = base0 + 64 * k + lane;
      opb[k] = ops[ai < L ? L - 1 - ai : 0u];
    }
    // rows / columns the batch's alignment columns consume (prefix popcounts of the ballots), then ALL its reference bytes requested
    // together, then the rows: the kernel waited for a reference byte per round of 64 columns (48 dependent round trips per trace)
    uint32_t rowk[kBatch], colk[kBatch];
    uint8_t c1k[kBatch];
    uint32_t take = 0;  // bit k: takes a row, bit 8 + k: takes a column
    const bool rc = a.a2_revcomp_flag && (d.flags & PAIR_A2_REVCOMP);
    const uint32_t mlast = d.m ? d.m - 1u : 0u, nlast = d.n ? d.n - 1u : 0u;
    const bool have_a1 = d.m != 0u, have_a2 = d.n != 0u;  // (without rows / columns no op takes one)
#pragma unroll
    for (uint32_t k = 0; k < kBatch; ++k) {
      const uint32_t ai = base0 + 64 * k + lane;
      const bool active = ai < L;
      const uint8_t op = opb[k];
      const bool takes_row = active && op != 'h';  // consumes a1
      const bool takes_col = active && op != 'v';  // consumes a2
      const uint64_t mrow = __ballot(takes_row), mcol = __ballot(takes_col);
      rowk[k] = row_base + (uint32_t)__popcll(mrow & below);
      colk[k] = col_base + (uint32_t)__popcll(mcol & below);
      row_base += (uint32_t)__popcll(mrow);
      col_base += (uint32_t)__popcll(mcol);
      take |= (takes_row ? 1u : 0u) << k | (takes_col ? 1u : 0u) << (8 + k);
    }
    // (every load below is unconditional from a clamped index and every choice a select: a conditional load per lane is a branch of its own)
    if (!a.a2_profile) {
#pragma unroll
      for (uint32_t k = 0; k < kBatch; ++k) {
        const uint32_t c = colk[k] < nlast ? colk[k] : nlast;
        c1k[k] = have_a2 ? static_cast<const uint8_t*>(a.a2)[d.a2_off + (rc ? nlast - c : c)] : (uint8_t)'-';
      }
    }
#pragma unroll
    for (uint32_t k = 0; k < kBatch; ++k) {
      const uint32_t base = base0 + 64 * k;
      if (base >= L) break;  // (wave-uniform)
      const uint32_t ai = base + lane;
      const bool active = ai < L;
      const bool takes_row = (take >> k) & 1u, takes_col = (take >> (8 + k)) & 1u;
      const uint32_t row = rowk[k] < mlast ? rowk[k] : mlast, col = colk[k] < nlast ? colk[k] : nlast;
      uint8_t c0 = 'N', c1 = '-';
      if (!gaps_only && have_a1) {  // (wave-uniform)
        if (a.a1_profile) {
          float p[6];
          for (int q = 0; q < 6; ++q) p[q] = static_cast<const float*>(a.a1)[d.a1_off + (uint64_t)q * d.a1_stride + row];
          c0 = cons_char(p);
        } else {
          c0 = static_cast<const uint8_t*>(a.a1)[d.a1_off + row];
        }
      }
      c0 = takes_row ? c0 : (uint8_t)'-';
      if (a.a2_profile) {
        if (have_a2) {  // (wave-uniform)
          float p[6];
          for (int q = 0; q < 6; ++q) p[q] = static_cast<const float*>(a.a2)[d.a2_off + (uint64_t)q * d.a2_stride + col];
          c1 = cons_char(p);
        }
      } else {
        c1 = c1k[k];
        if (rc) c1 = complement_char(c1);
        if (a.a2_onehot) {  // consensus character of _createProfile(string) (align.h:121-136, 254-270)
          const uint32_t code = base_code(c1);
          c1 = code == 0 ? 'A' : code == 1 ? 'C' : code == 2 ? 'G' : code == 3 ? 'T' : code == 6 ? 'A' : 'N';
        }
      }
      c1 = takes_col ? c1 : (uint8_t)'-';
      if (active) {
        r0[ai] = c0;
        r1[ai] = c1;
      }
    }
  }
}

// ---- launchers ---------------------------------------------------------------------------------
template <int K, int MODE, bool TRACE, bool NARROW = false>
static hipError_t launch_gotoh_t(const DpArgs& a, uint32_t npairs, hipStream_t s) {
  if constexpr (NARROW && MODE == MODE_QP) {
    // the 16-bit query-profile sweep: both forms over the same pairs, every pair is swept by the one its reference calls for
    // (a workgroup of the other form leaves at once); without the block map only the six-code form knows what to do
    if (a.special_blocks)
      hipLaunchKernelGGL((gotoh_kernel<K, MODE, TRACE, true, true>), dim3(npairs), dim3(64), lds_bytes_sweep16(K, true) + lds_pad(), s, a);
    hipLaunchKernelGGL((gotoh_kernel<K, MODE, TRACE, true, false>), dim3(npairs), dim3(64), lds_bytes_sweep16(K, false), s, a);
  } else {
    hipLaunchKernelGGL((gotoh_kernel<K, MODE, TRACE, NARROW>), dim3(npairs), dim3(64), lds_bytes(MODE, K), s, a);
  }
  return hipGetLastError();
}
template <int K, int MODE, bool TRACE>
static hipError_t launch_needle_t(const DpArgs& a, uint32_t npairs, hipStream_t s) {
  hipLaunchKernelGGL((needle_kernel<K, MODE, TRACE>), dim3(npairs), dim3(64), needle_lds_bytes(MODE, K), s, a);
  return hipGetLastError();
}

template <int MODE, bool TRACE>
static hipError_t launch_gotoh_k(int K, const DpArgs& a, uint32_t npairs, hipStream_t s) {
  switch (K) {
    case 4: return launch_gotoh_t<4, MODE, TRACE>(a, npairs, s);
    case 8: return launch_gotoh_t<8, MODE, TRACE>(a, npairs, s);
    case 12:
      if constexpr (MODE != MODE_PROF) return launch_gotoh_t<12, MODE, TRACE>(a, npairs, s);
      return hipErrorInvalidValue;
    case 15:
      if constexpr (MODE != MODE_PROF) return launch_gotoh_t<15, MODE, TRACE>(a, npairs, s);
      return hipErrorInvalidValue;
    case 16:
      if constexpr (MODE != MODE_PROF) return launch_gotoh_t<16, MODE, TRACE>(a, npairs, s);
      return hipErrorInvalidValue;
    default: return hipErrorInvalidValue;
  }
}
template <int MODE>
static hipError_t launch_gotoh_narrow(int K, const DpArgs& a, uint32_t npairs, hipStream_t s) {
  switch (K) {
    case 4: return launch_gotoh_t<4, MODE, false, true>(a, npairs, s);
    case 8: return launch_gotoh_t<8, MODE, false, true>(a, npairs, s);
    case 12: return launch_gotoh_t<12, MODE, false, true>(a, npairs, s);
    case 15: return launch_gotoh_t<15, MODE, false, true>(a, npairs, s);
    case 16: return launch_gotoh_t<16, MODE, false, true>(a, npairs, s);
    default: return hipErrorInvalidValue;
  }
}

template <bool TRACE, int NT, bool A16 = false>
static hipError_t launch_prof_k(int K, const DpArgs& a, uint32_t npairs, hipStream_t s) {
  switch (K) {
    case 4: hipLaunchKernelGGL((gotoh_prof_kernel<4, TRACE, NT, A16>), dim3(npairs), dim3(64), lds_bytes(MODE_PROF, 4) + lds_pad(), s, a); break;
    case 8: hipLaunchKernelGGL((gotoh_prof_kernel<8, TRACE, NT, A16>), dim3(npairs), dim3(64), lds_bytes(MODE_PROF, 8) + lds_pad(), s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t launch_gotoh_prof(int K, bool trace, bool row4_zero, bool arith16, const DpArgs& a, uint32_t npairs, hipStream_t s) {
  if (npairs == 0) return hipSuccess;
  if (trace) return row4_zero ? launch_prof_k<true, 4>(K, a, npairs, s) : launch_prof_k<true, 5>(K, a, npairs, s);
  if (arith16) return row4_zero ? launch_prof_k<false, 4, true>(K, a, npairs, s) : launch_prof_k<false, 5, true>(K, a, npairs, s);
  return row4_zero ? launch_prof_k<false, 4>(K, a, npairs, s) : launch_prof_k<false, 5>(K, a, npairs, s);
}

// traceback of profile rows with the table holding unshifted scores (gotoh_body RAWTAB): scorings beyond |1000|
static hipError_t launch_gotoh_rawtab(int K, const DpArgs& a, uint32_t npairs, hipStream_t s) {
#define TRACY_RAW(KK)                                                                                                         \
  case KK: hipLaunchKernelGGL((gotoh_kernel<KK, MODE_QP, true, false, true>), dim3(npairs), dim3(64), lds_bytes(MODE_QP, KK), s, a); break;
  switch (K) {
    TRACY_RAW(4) TRACY_RAW(8) TRACY_RAW(12) TRACY_RAW(15) TRACY_RAW(16)
    default: return hipErrorInvalidValue;
  }
#undef TRACY_RAW
  return hipGetLastError();
}

hipError_t launch_gotoh(int mode, int K, bool trace, bool narrow, const DpArgs& a, uint32_t npairs, hipStream_t s, bool rawtab) {
  if (npairs == 0) return hipSuccess;
  if (rawtab) return (trace && mode == MODE_QP) ? launch_gotoh_rawtab(K, a, npairs, s) : hipErrorInvalidValue;
  if (narrow && !trace) {
    if (mode == MODE_CHAR) return launch_gotoh_narrow<MODE_CHAR>(K, a, npairs, s);
    if (mode == MODE_QP) return launch_gotoh_narrow<MODE_QP>(K, a, npairs, s);
  }
  switch (mode) {
    case MODE_CHAR: return trace ? launch_gotoh_k<MODE_CHAR, true>(K, a, npairs, s) : launch_gotoh_k<MODE_CHAR, false>(K, a, npairs, s);
    case MODE_QP: return trace ? launch_gotoh_k<MODE_QP, true>(K, a, npairs, s) : launch_gotoh_k<MODE_QP, false>(K, a, npairs, s);
    case MODE_CQ: return trace ? launch_gotoh_k<MODE_CQ, true>(K, a, npairs, s) : hipErrorInvalidValue;  // tracebacks only
    case MODE_PROF: return trace ? launch_gotoh_k<MODE_PROF, true>(K, a, npairs, s) : launch_gotoh_k<MODE_PROF, false>(K, a, npairs, s);
    default: return hipErrorInvalidValue;
  }
}

template <int MODE, bool TRACE>
static hipError_t launch_needle_k(int K, const DpArgs& a, uint32_t npairs, hipStream_t s) {
  switch (K) {
    case 4: return launch_needle_t<4, MODE, TRACE>(a, npairs, s);
    case 8: return launch_needle_t<8, MODE, TRACE>(a, npairs, s);
    case 16:
      if constexpr (MODE != MODE_PROF) return launch_needle_t<16, MODE, TRACE>(a, npairs, s);
      return hipErrorInvalidValue;
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_needle(int mode, int K, bool trace, const DpArgs& a, uint32_t npairs, hipStream_t s) {
  if (npairs == 0) return hipSuccess;
  switch (mode) {
    case MODE_CHAR: return trace ? launch_needle_k<MODE_CHAR, true>(K, a, npairs, s) : launch_needle_k<MODE_CHAR, false>(K, a, npairs, s);
    case MODE_PROF: return trace ? launch_needle_k<MODE_PROF, true>(K, a, npairs, s) : launch_needle_k<MODE_PROF, false>(K, a, npairs, s);
    default: return hipErrorInvalidValue;
  }
}

template <int MODE>
static hipError_t launch_ckpt_m(int K, bool narrow, const DpArgs& a, uint32_t npairs, hipStream_t s) {
#define TRACY_CK(KK)                                                                                                   \
  case KK:                                                                                                              \
    if (narrow && qp_like(MODE)) {                                                                                      \
      if (a.special_blocks)                                                                                             \
        hipLaunchKernelGGL((gotoh_ckpt_kernel<KK, MODE, true, true>), dim3(npairs), dim3(64), lds_bytes_sweep16(KK, true) + lds_pad(), s, a); \
      hipLaunchKernelGGL((gotoh_ckpt_kernel<KK, MODE, true, false>), dim3(npairs), dim3(64), lds_bytes_sweep16(KK, false), s, a);             \
    } else if (narrow) hipLaunchKernelGGL((gotoh_ckpt_kernel<KK, MODE, true>), dim3(npairs), dim3(64), lds_bytes(MODE, KK), s, a);           \
    else hipLaunchKernelGGL((gotoh_ckpt_kernel<KK, MODE, false>), dim3(npairs), dim3(64), lds_bytes(MODE, KK), s, a);        \
    return hipGetLastError();
  switch (K) {
    TRACY_CK(4) TRACY_CK(8) TRACY_CK(12) TRACY_CK(15) TRACY_CK(16)
    default: return hipErrorInvalidValue;
  }
#undef TRACY_CK
}
// strings through the query-profile table (MODE_CQ): the 16-bit sweep only, both forms like MODE_QP
static hipError_t launch_ckpt_cq(int K, const DpArgs& a, uint32_t npairs, hipStream_t s) {
#define TRACY_CKQ(KK)                                                                                                   \
  case KK:                                                                                                              \
    if (a.special_blocks)                                                                                               \
      hipLaunchKernelGGL((gotoh_ckpt_kernel<KK, MODE_CQ, true, true>), dim3(npairs), dim3(64), lds_bytes_sweep16(KK, true) + lds_pad(), s, a); \
    hipLaunchKernelGGL((gotoh_ckpt_kernel<KK, MODE_CQ, true, false>), dim3(npairs), dim3(64), lds_bytes_sweep16(KK, false), s, a);             \
    return hipGetLastError();
  switch (K) {
    TRACY_CKQ(4) TRACY_CKQ(8) TRACY_CKQ(12) TRACY_CKQ(15) TRACY_CKQ(16)
    default: return hipErrorInvalidValue;
  }
#undef TRACY_CKQ
}
hipError_t launch_gotoh_ckpt(int mode, int K, bool narrow, const DpArgs& a, uint32_t npairs, hipStream_t s) {
  if (npairs == 0) return hipSuccess;
  if (mode == MODE_CHAR) return launch_ckpt_m<MODE_CHAR>(K, narrow, a, npairs, s);
  if (mode == MODE_QP) return launch_ckpt_m<MODE_QP>(K, narrow, a, npairs, s);
  if (mode == MODE_CQ && narrow) return launch_ckpt_cq(K, a, npairs, s);
  return hipErrorInvalidValue;
}
template <int MODE>
static hipError_t launch_band_m(int K, const DpArgs& a, const WalkArgs& wa, uint32_t npairs, hipStream_t s) {
#define TRACY_BD(KK)                                                                                                   \
  case KK:                                                                                                              \
    hipLaunchKernelGGL((gotoh_band_kernel<KK, MODE>), dim3(npairs), dim3(64), lds_bytes(MODE, KK), s, a, wa);           \
    return hipGetLastError();
  switch (K) {
    TRACY_BD(4) TRACY_BD(8) TRACY_BD(12) TRACY_BD(15) TRACY_BD(16)
    default: return hipErrorInvalidValue;
  }
#undef TRACY_BD
}
hipError_t launch_band_trace(int mode, int K, const DpArgs& a, const WalkArgs& wa, uint32_t npairs, hipStream_t s) {
  if (npairs == 0) return hipSuccess;
  if (mode == MODE_CHAR) return launch_band_m<MODE_CHAR>(K, a, wa, npairs, s);
  if (mode == MODE_QP) return launch_band_m<MODE_QP>(K, a, wa, npairs, s);
  return hipErrorInvalidValue;
}

hipError_t launch_gotoh_origin(int K, int table, int codes, const DpArgs& a, uint32_t npairs, hipStream_t s) {
  if (npairs == 0) return hipSuccess;
  if (table == 2) {  // profile rows (MODE_QP): six code rows
#define TRACY_ORIGIN_QP(KK)                                                                                                            \
  case KK:                                                                                                                              \
    if (a.special_blocks) hipLaunchKernelGGL((gotoh_origin_kernel<KK, 2, 4>), dim3(npairs), dim3(64), 4u * 64u * KK * 2u + lds_pad(), s, a); \
    hipLaunchKernelGGL((gotoh_origin_kernel<KK, 2, 6>), dim3(npairs), dim3(64), lds_bytes(MODE_QP, KK), s, a);                              \
    break;
    switch (K) {
      TRACY_ORIGIN_QP(4) TRACY_ORIGIN_QP(8) TRACY_ORIGIN_QP(12) TRACY_ORIGIN_QP(15) TRACY_ORIGIN_QP(16)
      default: return hipErrorInvalidValue;
    }
#undef TRACY_ORIGIN_QP
    return hipGetLastError();
  }
  if (table) {
#define TRACY_ORIGIN_CASE(KK)                                                                                                          \
  case KK:                                                                                                                              \
    if (codes <= 4) hipLaunchKernelGGL((gotoh_origin_kernel<KK, 1, 4>), dim3(npairs), dim3(64), 4u * 64u * KK * 2u + lds_pad(), s, a);      \
    else if (codes == 5) hipLaunchKernelGGL((gotoh_origin_kernel<KK, 1, 5>), dim3(npairs), dim3(64), 5u * 64u * KK * 2u + lds_pad(), s, a); \
    else hipLaunchKernelGGL((gotoh_origin_kernel<KK, 1, 6>), dim3(npairs), dim3(64), lds_bytes(MODE_CQ, KK), s, a);                          \
    break;
    switch (K) {
      TRACY_ORIGIN_CASE(4) TRACY_ORIGIN_CASE(8) TRACY_ORIGIN_CASE(12) TRACY_ORIGIN_CASE(15) TRACY_ORIGIN_CASE(16)
      default: return hipErrorInvalidValue;
    }
#undef TRACY_ORIGIN_CASE
    return hipGetLastError();
  }
  switch (K) {
    case 4: hipLaunchKernelGGL((gotoh_origin_kernel<4>), dim3(npairs), dim3(64), 0, s, a); break;
    case 8: hipLaunchKernelGGL((gotoh_origin_kernel<8>), dim3(npairs), dim3(64), 0, s, a); break;
    case 12: hipLaunchKernelGGL((gotoh_origin_kernel<12>), dim3(npairs), dim3(64), 0, s, a); break;
    case 15: hipLaunchKernelGGL((gotoh_origin_kernel<15>), dim3(npairs), dim3(64), 0, s, a); break;
    case 16: hipLaunchKernelGGL((gotoh_origin_kernel<16>), dim3(npairs), dim3(64), 0, s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// LDS of a launch that holds sweeps and prefix-bound workgroups: the larger of the two tables
static constexpr uint32_t lds_combo(int K, bool compact) {
  const uint32_t pre = lds_bytes_prefix(K, compact), sw = lds_bytes_sweep16(K, compact);
  return pre > sw ? pre : sw;
}
hipError_t launch_gotoh_ckpt_prefix(int K, const DpArgs& full, uint32_t nfull, const DpArgs& pre, uint32_t npre, hipStream_t s) {
  if (nfull + npre == 0) return hipSuccess;
  constexpr int GL = kPrefixLanes;
  const dim3 grid(nfull + (npre + 64 / GL - 1) / (64 / GL));
  // both forms over the same sweeps and prefix groups: each is worked on in the form its reference calls for
#define TRACY_COMBO_CASE(KK)                                                                                            \
  case KK:                                                                                                              \
    if (full.special_blocks) {                                                                                          \
      hipLaunchKernelGGL((gotoh_ckpt_prefix_kernel<KK, GL, true>), grid, dim3(64), lds_combo(KK, true), s, full, nfull, pre, npre); \
      hipLaunchKernelGGL((gotoh_ckpt_prefix_kernel<KK, GL, false>), grid, dim3(64), lds_combo(KK, false), s, full, nfull, pre, npre); \
    } else {                                                                                                            \
      hipLaunchKernelGGL((gotoh_ckpt_prefix_kernel<KK, GL, false>), grid, dim3(64), lds_combo(KK, false), s, full, nfull, pre, npre); \
    }                                                                                                                   \
    break;
  switch (K) {
    TRACY_COMBO_CASE(4) TRACY_COMBO_CASE(8) TRACY_COMBO_CASE(12) TRACY_COMBO_CASE(15) TRACY_COMBO_CASE(16)
    default: return hipErrorInvalidValue;
  }
#undef TRACY_COMBO_CASE
  return hipGetLastError();
}

// the same with the prefix shape of the pruned orientation sweep (front.h): kFrontRows rows per pair as sixteen lanes of eight rows (four pairs
// per wave: twice the waves, what a batch of 10 000 traces needs to fill the device), or -- from front_prefix_tall_min pairs on -- as eight
// lanes of sixteen rows: a step's fixed work (hand-over, code look-up, the kept row's store) is paid per sixteen cells instead of eight
// (measured: 100 000 traces -1.6 ms per decompose step; 10 000 / 12 500 traces +0.3 / +0.4 ms -- the tall shape halves the waves of a launch
// that is a few waves deep)
static uint32_t front_prefix_tall_min() {
  static const uint32_t v = [] { const char* e = getenv("TRACYHIP_PREFIX_TALL_MIN"); return e ? (uint32_t)strtoul(e, nullptr, 10) : 40000u; }();
  return v;
}
template <int GL, int KP>
static hipError_t launch_gotoh_ckpt_front_t(int K, const DpArgs& full, uint32_t nfull, const DpArgs& pre, uint32_t npre, hipStream_t s) {
  static_assert(GL * KP == (int)kFrontRows, "the rows above the kept one");
  const dim3 grid(nfull + (npre + 64 / GL - 1) / (64 / GL));
  auto lds = [](int KK, bool compact) {
    const uint32_t pre_b = lds_bytes_prefix(KP, compact), sw = lds_bytes_sweep16(KK, compact);
    return (pre_b > sw ? pre_b : sw) + lds_pad();
  };
#define TRACY_FRONT_CASE(KK)                                                                                            \
  case KK:                                                                                                              \
    if (full.special_blocks) {                                                                                          \
      /* (the six-code form usually has nothing to do -- references of A C G T --, but its 12 KB workgroups find no room while the   */ \
      /* four-code form's 7.5 KB workgroups fill the CUs: it goes last, where the device drains; first was measured and waits as long) */ \
      hipLaunchKernelGGL((gotoh_ckpt_prefix_kernel<KK, GL, true, KP>), grid, dim3(64), lds(KK, true), s, full, nfull, pre, npre); \
      hipLaunchKernelGGL((gotoh_ckpt_prefix_kernel<KK, GL, false, KP>), grid, dim3(64), lds(KK, false), s, full, nfull, pre, npre); \
    } else {                                                                                                            \
      hipLaunchKernelGGL((gotoh_ckpt_prefix_kernel<KK, GL, false, KP>), grid, dim3(64), lds(KK, false), s, full, nfull, pre, npre); \
    }                                                                                                                   \
    break;
  switch (K) {
    TRACY_FRONT_CASE(4) TRACY_FRONT_CASE(8) TRACY_FRONT_CASE(12) TRACY_FRONT_CASE(15) TRACY_FRONT_CASE(16)
    default: return hipErrorInvalidValue;
  }
#undef TRACY_FRONT_CASE
  return hipGetLastError();
}
hipError_t launch_gotoh_ckpt_front(int K, const DpArgs& full, uint32_t nfull, const DpArgs& pre, uint32_t npre, hipStream_t s) {
  if (nfull + npre == 0) return hipSuccess;
  // (a launch that carries full sweeps keeps the small prefix table: its LDS request is the larger of the two bodies')
  if (nfull == 0 && npre >= front_prefix_tall_min()) return launch_gotoh_ckpt_front_t<(int)kFrontRows / 16, 16>(K, full, nfull, pre, npre, s);
  return launch_gotoh_ckpt_front_t<kFrontPrefixLanes, kFrontPrefixK>(K, full, nfull, pre, npre, s);
}

hipError_t launch_gotoh_prefix(int K, const DpArgs& a, uint32_t npairs, hipStream_t s) {
  if (npairs == 0) return hipSuccess;
  constexpr int GL = kPrefixLanes;
  const dim3 grid((npairs + 64 / GL - 1) / (64 / GL));
#define TRACY_PREFIX_CASE(KK) \
  case KK:                                                                                                                      \
    hipLaunchKernelGGL((gotoh_prefix_kernel<KK, GL, false>), grid, dim3(64), lds_bytes_prefix(KK, false), s, a, npairs);                        \
    if (a.special_blocks) hipLaunchKernelGGL((gotoh_prefix_kernel<KK, GL, true>), grid, dim3(64), lds_bytes_prefix(KK, true), s, a, npairs); \
    break;
  switch (K) {
    TRACY_PREFIX_CASE(4) TRACY_PREFIX_CASE(8) TRACY_PREFIX_CASE(12) TRACY_PREFIX_CASE(15) TRACY_PREFIX_CASE(16)
    default: return hipErrorInvalidValue;
  }
#undef TRACY_PREFIX_CASE
  return hipGetLastError();
}
// the prefix rows of the pruned sweep (front.h) for string x code pairs: the shapes of launch_gotoh_ckpt_front, row R kept
template <int GL, int KP>
static hipError_t launch_gotoh_front_prefix_cq_t(const DpArgs& a, uint32_t npairs, hipStream_t s) {
  const dim3 grid((npairs + 64 / GL - 1) / (64 / GL));
  hipLaunchKernelGGL((gotoh_prefix_kernel<KP, GL, false, true>), grid, dim3(64), lds_bytes_prefix(KP, false), s, a, npairs);  // (first: launch_gotoh_ckpt_front)
  if (a.special_blocks) hipLaunchKernelGGL((gotoh_prefix_kernel<KP, GL, true, true>), grid, dim3(64), lds_bytes_prefix(KP, true), s, a, npairs);
  return hipGetLastError();
}
hipError_t launch_gotoh_front_prefix_cq(const DpArgs& a, uint32_t npairs, hipStream_t s) {
  if (npairs == 0) return hipSuccess;
  if (npairs >= front_prefix_tall_min()) return launch_gotoh_front_prefix_cq_t<(int)kFrontRows / 16, 16>(a, npairs, s);
  return launch_gotoh_front_prefix_cq_t<kFrontPrefixLanes, kFrontPrefixK>(a, npairs, s);
}
hipError_t launch_gotoh_walk(const WalkArgs& a, hipStream_t s) {
  if (a.npairs == 0) return hipSuccess;
  hipLaunchKernelGGL(gotoh_walk_kernel, dim3(a.npairs), dim3(64), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_needle_walk(const WalkArgs& a, const uint32_t* bits32, hipStream_t s) {
  if (a.npairs == 0) return hipSuccess;
  hipLaunchKernelGGL(needle_walk_kernel, dim3((a.npairs + 63) / 64), dim3(64), 0, s, a, bits32);
  return hipGetLastError();
}
hipError_t launch_alignment_rows(const RowsArgs& a, hipStream_t s) {
  if (a.npairs == 0) return hipSuccess;
  hipLaunchKernelGGL(alignment_rows_kernel, dim3(a.npairs), dim3(64), 0, s, a);
  return hipGetLastError();
}

}  // namespace tracyhip
