// dp_kernels.h -- wave-level bodies of the batched Gotoh / NW kernels and the traceback walker.
//
// The bodies are templates over a "wave policy" W that supplies lane id, the one-lane shift
// (DPP wave_shr:1 on the device) and the LDS base.  dp_kernels.hip instantiates them with the device
// policy inside __global__ wrappers; tests/emu instantiates the SAME bodies with a 64-thread host
// policy to check the index math against the oracle without a GPU (test infrastructure only -- the
// product library contains no host execution path).
//
// Work decomposition: one wave (one 64-thread workgroup) per pair, thousands of pairs per launch.
// Lane L owns rows base + L*K + 1 .. base + L*K + K of the current pass; step t puts it on column
// t - L (see dp_lane.h).  HBM traffic per pair: inputs once (1 B per reference base, 24 B per profile
// column), 8 B per lane and step of traceback nibbles (= 0.5 B per cell, written as one contiguous
// 512-byte wave store per step), 4 B score.
#ifndef TRACY_AMD_DP_KERNELS_H
#define TRACY_AMD_DP_KERNELS_H

#include "dp_lane.h"

// an empty volatile asm keeps a wave-uniform branch a branch (the compiler would if-convert it into selects)
#if defined(__HIP_DEVICE_COMPILE__)
#define TRACY_KEEP_BRANCH() asm volatile("" ::: "memory")
#else
#define TRACY_KEEP_BRANCH() ((void)0)
#endif

namespace tracyhip {

// MODE_CQ: string x string scored through the query-profile table (rows are chars, a2 = case-sensitive codes): for row strings
// over {A,C,G,T,N} -- every basecall string -- "row char == column char ? match : mismatch" is a table look-up per step instead
// of compare + select per cell.  Same kernels as MODE_QP otherwise.
enum : int { MODE_CHAR = 0, MODE_QP = 1, MODE_PROF = 2, MODE_CQ = 3 };
TR_HD constexpr bool qp_like(int mode) { return mode == MODE_QP || mode == MODE_CQ; }
// case-SENSITIVE column code of MODE_CQ: only the five upper-case letters a row may hold match anything
TR_HD uint32_t cq_code(uint8_t c) { return c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : c == 'N' ? 4u : 5u; }
TR_HD bool cq_row_char(uint8_t c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'N'; }
enum : uint32_t {
  PAIR_A2_REVCOMP = 1u,  // read a2 reversed and complemented (profile.h:74-90)
  PAIR_ROW4_ZERO = 2u,   // profile x profile: row 4 ('N') is zero in BOTH profiles (the host classified the sequences): launches of
                         // such pairs run the 16-term body
  PAIR_BANDED = 4u,      // full-matrix traceback on the diagonals band_dmin .. band_dmax only (multi-pass form: every pass sweeps
                         // the columns its rows can reach inside the band; cells outside read as -inf).  The caller certifies the
                         // result (pipeline.hip, final alignments) and repeats the pair without the flag otherwise.
  PAIR_KEEP_ROW = 8u,    // prefix-bound kernel: leave row R of the prefix behind -- one dword per column c at lastrow[lastrow_off + c],
                         // low half H(R, c) + (go + ge), high half F(R, c) -- for the band kernels to continue from (band16.h)
  PAIR_SKIP = 16u,       // an empty slot of a list laid out on the device (stream.hip): the 16-bit sweeps, the prefix kernels and the
                         // band kernels leave at once and write nothing
};

// one DP problem; lives in device memory, built on the host
struct PairDesc {
  uint64_t a1_off;      // element offset of a1 (bytes for chars, floats for profiles: &p[0][0])
  uint64_t a2_off;
  uint64_t bits_off;    // traceback words: offset in 8-byte units (Gotoh) / 4-byte units (NW)
  uint64_t scratch_off; // boundary-row scratch for multi-pass problems, in int2 units
  uint32_t a1_stride;   // profiles: distance between profile rows k (>= m; a trimmed view keeps the full stride)
  uint32_t a2_stride;
  uint32_t m;           // rows    = _size(a1, 1)
  uint32_t n;           // columns = _size(a2, 1)
  uint32_t out;         // index into the per-pair outputs
  uint32_t flags;
  uint64_t ckpt_off;    // wavefront checkpoints of this pair (int32 units) -- checkpointed score / band traceback
  uint64_t lastrow_off; // {H, E} of row m per column (int32 units, 2 per column)
};
// PAIR_BANDED pairs (full-matrix traceback, no checkpoints) keep the diagonals c - r of their band in ckpt_off: dmin (<= 0) in
// the low half, dmax (>= n - m) in the high half
TR_HD uint64_t band_pack(int32_t dmin, int32_t dmax) { return (uint64_t)(uint32_t)dmin | ((uint64_t)(uint32_t)dmax << 32); }
TR_HD int32_t band_dmin(const PairDesc& d) { return (int32_t)(uint32_t)d.ckpt_off; }
TR_HD int32_t band_dmax(const PairDesc& d) { return (int32_t)(uint32_t)(d.ckpt_off >> 32); }


struct DpArgs {
  const PairDesc* pairs;
  const void* a1;       // uint8_t* or float*
  const void* a2;       // MODE_QP: base codes 0..6 (dp_lane.h base_code), one byte per column
  uint64_t* bits;       // Gotoh traceback words
  uint32_t* bits32;     // NW traceback words
  int32_t* scratch;     // int2 per column: {H, F} of the last row of the previous pass
  int32_t* scores;      // may be null in traceback kernels
  int32_t* err;         // device error block, kErrWords int32: [0] flags (bit 0: a query-profile value does not fit int16,
                        // bit 1: a traceback walk left the matrix), [1] largest |query-profile score| seen above qlimit,
                        // [2] / [3] largest column mass sum_k |p[k][j]| of an a1 / a2 profile above 1 (float bits; profile x profile)
  int32_t match, mismatch, go, ge;
  int32_t qlimit;       // max(|match|, |mismatch|): what a substitution score of NORMALISED profiles cannot exceed.  The host-side
                        // range guards (narrow_ok, origin_ok, check_params) assume it; kernels report anything larger in err[1..3]
  int32_t hfree, vfree;
  int32_t screen;       // profile x profile: substitution scores by the screened short form where it is proven (SubProf::screen)
  // traceback kernels: where the walker's output goes when the workgroup walks its own pair right after the sweep (null: a
  // separate walk launch).  Same meaning as WalkArgs::ops / ops_off / ops_len.
  uint8_t* walk_ops;
  const uint64_t* walk_ops_off;
  uint32_t* walk_ops_len;
  const uint8_t* special_blocks;  // MODE_QP: one byte per 256 code bytes of the a2 buffer, non-zero where the block holds an N or a
                                  // '-' / other code (written by the encoders); null = unknown.  The 16-bit sweep exists in two
                                  // forms (gotoh_narrow_qp_body): references without such codes take the one with the small table
  const uint8_t* colcode;  // profile x profile: class of every a2 column, indexed like row 0 of the a2 buffer (column_class);
                           // null = no screening
  int32_t* ckpt;        // wavefront checkpoints (score kernel writes, band traceback reads)
  int32_t* lastrow;     // last-row {H, E} per column
  uint64_t* band;       // band traceback: per-workgroup nibble words of the current band (ckpt_B * 64 words each)
  uint32_t ckpt_B;      // steps between wavefront checkpoints
  int32_t ckpt_narrow;  // checkpoints hold raw registers of the 16-bit kernel (values in the low halves)
  uint32_t* ends;       // origin-tracking sweep: {leading 'h' columns, last column that is not a trailing 'h'} per pair
  unsigned long long* swept;  // band traceback: DP cells actually re-swept, summed over the launch (or null)
  const uint32_t* index;  // prefix sweeps (gotoh_prefix_body): pair i of the launch is pairs[index[i]] and only *count of them exist -- a list
  const uint32_t* count;  // laid out on the device, the grid sized for its worst case (null: pairs[i], npairs)
  const uint32_t* votes;  // checkpointed 16-bit sweep of both orientations (PairDesc::out = orientation * vote_nt + trace): {vf, vr} per
  uint32_t vote_nt;       // trace, or null.  Sweeps of the likely losing strand (vote_skips_checkpoints) write no checkpoints / row m
};

// device error flags are OR-ed (several kernels share the word)
constexpr int kErrWords = 4;
TR_HD void flag_error(int32_t* err, int32_t bits) {
  if (!err) return;
#if defined(__HIP_DEVICE_COMPILE__)
  atomicOr(err, bits);
#else
  *err |= bits;
#endif
}
// err[word] = max(err[word], v) for non-negative v (also used on the bits of non-negative floats, which order like ints)
TR_HD void flag_max(int32_t* err, int word, int32_t v) {
  if (!err) return;
#if defined(__HIP_DEVICE_COMPILE__)
  atomicMax(err + word, v);
#else
  if (v > err[word]) err[word] = v;
#endif
}
TR_HD int32_t float_bits(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __float_as_int(f);
#else
  int32_t i;
  __builtin_memcpy(&i, &f, 4);
  return i;
#endif
}
// Profile x profile scoring (align.h:103-118) is bounded by |score| <= (sum_k |p1[k][row]|) (sum_k |p2[k][col]|) max(|match|, |mismatch|).
// Column masses of createProfile output are <= 1 (+ rounding), which is what the host-side range checks assume; a lane that sees
// a larger mass (or a NaN) reports it, and the host re-checks the value range with the real bound after the launch.
TR_HD void report_mass(int32_t* err, int word, float mass) {
  if (!(mass <= 1.001f)) flag_max(err, word, float_bits(mass != mass ? __builtin_inff() : mass));
}
// running maximum that keeps a NaN once it has seen one
TR_HD float mass_max(float acc, float s) { return (s > acc || s != s) ? s : acc; }
TR_HD float column_mass(const float* p, uint64_t stride, uint32_t j) {
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < 5; ++k) { const float v = p[(uint64_t)k * stride + j]; s += v < 0.0f ? -v : v; }
  return s;
}

// Class of a profile column for the screened profile x profile score: 0..4 = one-hot column (1.0 in that row, +0 elsewhere:
// _createProfile of a string, or a trace position whose other channels are silent), 5 = the uniform column createProfile
// writes when a position has no signal (0.25 x 4, profile.h:39-40), 6 = anything else.  Against a column of class < 6 the
// float chain of align.h:112-116 depends on the row only, so its int comes from a per-row table (gotoh_body).
constexpr uint32_t kColClassOther = 6;
TR_HD uint32_t column_class(const float* p, uint64_t stride, uint32_t j) {
  uint32_t bits[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) bits[k] = (uint32_t)float_bits(p[(uint64_t)k * stride + j]);
  uint32_t ones = 0, zeros = 0, quarters = 0, hot = 0;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    if (bits[k] == 0x3f800000u) { ++ones; hot = (uint32_t)k; }
    if (bits[k] == 0u) ++zeros;
    if (k < 4 && bits[k] == 0x3e800000u) ++quarters;
  }
  if (ones == 1u && zeros == 4u) return hot;
  if (quarters == 4u && bits[4] == 0u) return 5u;
  return kColClassOther;
}

// ---- substitution-score providers -------------------------------------------------------------
template <int K>
struct SubChar {
  int32_t rc[K];  // row characters of this lane (-1 beyond m: never equals a byte)
  int32_t cc;
  int32_t vmatch, vmis;
  TR_HD int32_t operator()(int i) const { return rc[i] == cc ? vmatch : vmis; }
  TR_HD int32_t lo16(int i) const { return rc[i] == cc ? vmatch : vmis; }
};

// query-profile table of the sweeps and tracebacks: int16 [codes][K rows][64 lanes] -- a row's 64 lanes are contiguous, so a
// wave's 16-bit reads of one row hit every bank once when the lanes agree on the code (a per-lane strip layout costs an 8-way
// bank conflict on each of them).  Codes 0..4 = A C G T N, 5 = '-' / any other letter; kernels that know their columns hold
// fewer codes lay out fewer rows (gotoh_narrow_qp_body, gotoh_origin_body): the table size decides the waves per CU.
template <int K>
TR_HD uint32_t qp6_index(uint32_t code, uint32_t row, uint32_t lane) { return (code * (uint32_t)K + row) * 64u + lane; }

// A strip read row by row from the [code][row][lane] table (qp6_index): one sign-extending 16-bit LDS read per row gives an
// operand that needs no unpacking.  SHIFT: applied when a value is used (the origin-tracking sweep keeps scores << 18).
template <int K, int SHIFT = 0>
struct SubRows {
  int32_t sv[K];
  TR_HD int32_t operator()(int i) const { return (int32_t)((uint32_t)sv[i] << SHIFT); }
  TR_HD int32_t lo16(int i) const { return sv[i]; }
};
template <int K, int SHIFT>
TR_HD void qp_fetch_rows(const int16_t* lane_col, uint32_t code, SubRows<K, SHIFT>& q) {
  const int16_t* p = lane_col + code * ((uint32_t)K * 64u);
#pragma unroll
  for (int i = 0; i < K; ++i) q.sv[i] = p[i * 64];
}

// profile x profile: the profile columns of this lane's K rows live in registers for the whole pass, the column
// of a2 changes per step.  NT = 4 leaves out row 4 ('N') when it is zero in both profiles (profile_score<4>).
template <int K, int NT = 5>
struct SubProf {
  float a[K][5];
  float b[5];
  float fmatch, fmis;
  int shift;
  int32_t sv[K];  // scores of the current column, filled by prepare()
  // Two rows at a time: the same k1-outer / k2-inner chain of rounded multiplies and adds per row (align.h:112-116),
  // carried in the two halves of packed fp32 operations (v_pk_mul_f32 / v_pk_add_f32: one issue slot for two rows).
  TR_HD void prepare() {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma clang fp contract(off)
    typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i + 1 < K; i += 2) {
      f2 acc = {0.0f, 0.0f};
#pragma unroll
      for (int k1 = 0; k1 < NT; ++k1) {
        const f2 x = {a[i][k1], a[i + 1][k1]};
#pragma unroll
        for (int k2 = 0; k2 < NT; ++k2) {
          const float w = (k1 == k2) ? fmatch : fmis;
          const f2 bb = {b[k2], b[k2]};
          const f2 ww = {w, w};
          const f2 t = x * bb;
          const f2 u = t * ww;
          acc = acc + u;
        }
      }
      sv[i] = (int32_t)((uint32_t)(int32_t)acc.x << shift);
      sv[i + 1] = (int32_t)((uint32_t)(int32_t)acc.y << shift);
    }
    if (K & 1) sv[K - 1] = (int32_t)((uint32_t)profile_score<NT>(a[K - 1], b, fmatch, fmis) << shift);
#else
#pragma unroll
    for (int i = 0; i < K; ++i) sv[i] = (int32_t)((uint32_t)profile_score<NT>(a[i], b, fmatch, fmis) << shift);
#endif
  }
  // ---- screened evaluation -------------------------------------------------------------------------------------------
  // The reference truncates its float sum R to an int (align.h:117), so all that is needed of R is trunc(R).  In exact
  // arithmetic the 25 terms collapse to  T = sum_k a_k e_k  with  e_k = (match - mismatch) b_k + mismatch (sum b): NT fused
  // multiply-adds per cell once the column's e_k are known.  Rounding keeps both R and that short form within a proven
  // distance of T (u = 2^-24, Q = max(|match|, |mismatch|), M = (column mass of a) (column mass of b) <= 1.001^2 -- gotoh_body
  // only screens pairs whose masses it has verified):
  //     |R - T| <= gamma_27 Q M    (25 products of three factors rounded twice, 24 rounded additions)
  //     |x - T| <= 23 u Q M        (sum b, mismatch * sum, NT fmas for e_k: 8 u Q M;  NT fmas of the chain over partial sums
  //                                 <= 3 Q M: 15 u Q M)
  // With delta = 96 u Q > 50.2 u Q the interval [x - delta, x + delta] holds R; when it holds no integer, trunc(x) ==
  // trunc(R) and the chain is not needed.  The fma chain starts from +delta, so the test is  fract(x + delta) >= 2 delta
  // (one v_fract, one subtract, the sign bits of the strip OR-ed together).  A strip that fails the test in some lane is
  // evaluated by the float chain (prepare()): the result is the reference's int either way.
  float fdelta, fD;
  TR_HD void screen_setup() {
    const float q = __builtin_fmaxf(__builtin_fabsf(fmatch), __builtin_fabsf(fmis));
    fdelta = 96.0f * 5.9604644775390625e-08f * __builtin_fmaxf(q, 1.0f);
    fD = fmatch - fmis;
  }
  // fills sv[]; the sign bit of the result is set when some truncation of this strip is not proven
  TR_HD int32_t screen() {
    float s2 = b[0];
#pragma unroll
    for (int k = 1; k < NT; ++k) s2 += b[k];
    const float ms = fmis * s2;
    float e[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) e[k] = __builtin_fmaf(fD, b[k], ms);
    const float two_delta = 2.0f * fdelta;
    int32_t bad = 0;
#pragma unroll
    for (int i = 0; i < K; ++i) {
      float x = __builtin_fmaf(a[i][0], e[0], fdelta);
#pragma unroll
      for (int k = 1; k < NT; ++k) x = __builtin_fmaf(a[i][k], e[k], x);
#if defined(TRACY_ABLATE_SCREEN)
      // measurement only: the four fused multiply-adds alone -- no conversion, no test; results are wrong
      sv[i] = float_bits(x) >> 20;
      (void)two_delta;
#else
      sv[i] = (int32_t)((uint32_t)(int32_t)x << shift);
#if defined(__HIP_DEVICE_COMPILE__)
      const float g = __builtin_amdgcn_fractf(x) - two_delta;
#else
      const float g = (x - __builtin_floorf(x)) - two_delta;
#endif
      bad |= float_bits(g);
#endif
    }
    return bad;
  }
  TR_HD int32_t operator()(int i) const { return sv[i]; }
  TR_HD int32_t lo16(int i) const { return sv[i]; }
};

// LDS bytes a (mode, K) kernel needs
// LDS bytes of the Needleman-Wunsch kernels (profile rows are staged in LDS there)
TR_HD constexpr uint32_t needle_lds_bytes(int mode, int K) { return mode == MODE_PROF ? 5u * 64u * K * 4u : 0u; }
TR_HD constexpr uint32_t lds_bytes(int mode, int K) {
  // 6 code rows x K rows x 64 lanes of int16 (qp6_index).  MODE_PROF keeps its rows in registers; its table holds the ints of the
  // float chain against one-hot / uniform columns (column_class)
  return (qp_like(mode) || mode == MODE_PROF) ? 6u * 64u * (uint32_t)K * 2u : 0u;
}

// MODE_QP sweeps read the code buffer up to kCodeBias bytes before / behind a sequence (idle lanes, look-ahead)
constexpr uint32_t kCodeBias = 96;

TR_HD uint32_t a2_index(const PairDesc& d, uint32_t c /*1-based column*/) {
  return (d.flags & PAIR_A2_REVCOMP) ? d.n - c : c - 1u;
}

// ------------------------------------------------------------------------------------------------
// Gotoh, one pair per wave.  TRACE=true: tagged x16 arithmetic + traceback words; false: plain int32.
// ------------------------------------------------------------------------------------------------
// wavefront checkpoint record per lane: Hl[K], El[K], bot_h, bot_f, prev_up_h (plain int32 scores)
TR_HD constexpr uint32_t ckpt_fields(int K) { return 2u * K + 3u; }
// the 16-bit query-profile sweep packs its frontier: field i < K = {H (low half), E - goe (high half)} of slot i,
// field K = {H received from the strip above, F - goe}: K + 1 dwords per lane instead of 2K + 3
TR_HD constexpr uint32_t ckpt_fields_qp16(int K) { return (uint32_t)K + 1u; }
TR_HD uint64_t ckpt_index(uint32_t j /*1-based*/, uint32_t field, uint32_t lane, int K) {
  return ((uint64_t)(j - 1) * ckpt_fields(K) + field) * 64u + lane;
}

template <class W, int K, bool CKPT, bool COMPACT, bool STRINGS>
TR_HD void gotoh_narrow_qp_body(W& w, const DpArgs& a, uint32_t pair_idx);  // the 16-bit query-profile sweep, below

// NT (profile x profile only): 5 = the 25-term substitution score, 4 = the 16-term one (row 4 zero in both profiles of every
// pair of the launch: PAIR_ROW4_ZERO), 0 = decide per pair inside the kernel (both bodies in one kernel: more registers)
// COMPACT selects the second form of two kernels: the 16-bit query-profile sweep with the four-code table (NARROW, MODE_QP), and
// the profile x profile score kernel with 16-bit cells (MODE_PROF, !TRACE: any AlignConfig, any number of passes; v_add_u16 /
// v_max_i16 instead of int32 maxima, which issue at half the rate -- arith16_ok in capi.hip admits the launch)
template <class W, int K, int MODE, bool TRACE, bool NARROW = false, bool CKPT = false, int NT = 0, bool COMPACT = false>
TR_HD void gotoh_body(W& w, const DpArgs& a, uint32_t pair_idx) {
  static_assert(!(NARROW && TRACE), "the 16-bit formulation exists for the score-only kernel");
  constexpr bool A16 = MODE == MODE_PROF && !TRACE && !NARROW && !CKPT && COMPACT;
  // traceback with table scores (TRACE, MODE_QP / MODE_CQ) and COMPACT: the table holds the scores as they are and a cell shifts its
  // entry into the tagged field when it uses it (one more operation per cell) -- for scorings whose entries x 32 leave int16
  // (|match| or |mismatch| > 1000: the reference takes any int, align.h:11-32)
  constexpr bool RAWTAB = TRACE && qp_like(MODE) && COMPACT;
  constexpr int TSH = RAWTAB ? 0 : (TRACE ? kTagShift : 0);  // shift applied when the table is filled
  static_assert(!(CKPT && TRACE), "checkpoints are written by the score-only kernel");
  // NARROW / CKPT kernels: single pass, free end gaps on the first/last row only, rows anchored at the bottom
  constexpr bool BOTTOM = NARROW || CKPT;
  if constexpr (NARROW && qp_like(MODE)) {  // the hot kernel of `tracy align` has its own body (MODE_CQ: its rows are characters)
    gotoh_narrow_qp_body<W, K, CKPT, COMPACT, MODE == MODE_CQ>(w, a, pair_idx);
    return;
  }
  const PairDesc d = a.pairs[pair_idx];
  const uint32_t L = w.lane();
  const uint32_t m = d.m, n = d.n;
  const int32_t go = a.go, ge = a.ge;
  const bool hfree = a.hfree != 0, vfree = a.vfree != 0;
  constexpr int SH = TRACE ? kTagShift : 0;

  if (m == 0 || n == 0) {  // only the init row / column exists (gotoh.h:106-123)
    if (L == 0 && a.scores)
      a.scores[d.out] = (m == 0) ? (n == 0 ? 0 : edge_value(hfree, go, ge, (int32_t)n)) : edge_value(vfree, go, ge, (int32_t)m);
    return;
  }

  constexpr bool A1CHARS = MODE == MODE_CHAR || MODE == MODE_CQ;
  const uint8_t* a1c = static_cast<const uint8_t*>(a.a1) + (A1CHARS ? d.a1_off : 0);
  const float* a1p = static_cast<const float*>(a.a1) + (!A1CHARS ? d.a1_off : 0);
  const uint8_t* a2c = static_cast<const uint8_t*>(a.a2) + (MODE != MODE_PROF ? d.a2_off : 0);
  const float* a2p = static_cast<const float*>(a.a2) + (MODE == MODE_PROF ? d.a2_off : 0);
  int16_t* qp_tab = reinterpret_cast<int16_t*>(w.lds());
  const float fmatch = (float)a.match, fmis = (float)a.mismatch;
  // profile x profile: is row 4 ('N') zero in both profiles?  (NaN counts as non-zero.)
  bool skip4 = false, screen_pair = false;
  if (MODE == MODE_PROF) {
    bool nz = false;
    float ma = 0.0f, mb = 0.0f;
    for (uint32_t r = L; r < m; r += 64) { nz |= !(a1p[4ull * d.a1_stride + r] == 0.0f); const float s = column_mass(a1p, d.a1_stride, r); ma = mass_max(ma, s); }
    for (uint32_t c = L; c < n; c += 64) { nz |= !(a2p[4ull * d.a2_stride + c] == 0.0f); const float s = column_mass(a2p, d.a2_stride, c); mb = mass_max(mb, s); }
    skip4 = (NT == 4) || (NT == 0 && w.ballot(nz) == 0);
    report_mass(a.err, 2, ma);
    report_mass(a.err, 3, mb);
    // the screened substitution score (SubProf::screen) is proven for column masses <= 1.001 only
    screen_pair = a.screen != 0 && a.colcode != nullptr && w.ballot(!(ma <= 1.001f) || !(mb <= 1.001f)) == 0;
  }

  const uint32_t P = num_passes(m, K);
  const uint32_t T = steps_per_pass(n);
  uint64_t* bits = TRACE ? a.bits + d.bits_off : nullptr;
  int32_t* scratch = a.scratch ? a.scratch + 2 * d.scratch_off : nullptr;
  const int32_t neg = (NARROW || A16) ? kNegInf16 : (int32_t)((uint32_t)kNegInf << SH);

  for (uint32_t p = 0; p < P; ++p) {
    const uint32_t base = p * 64u * K;  // rows base+1 .. base+64K
    const uint32_t rows_here = (m - base < 64u * K) ? m - base : 64u * K;
    const uint32_t lanes_used = (rows_here + K - 1) / K;
    const bool last_pass = (p + 1 == P);
    // columns of this pass: all of them, or (PAIR_BANDED) those its rows can reach inside the band; c_plo / c_phi: the same
    // of the previous pass (what its hand-over row holds)
    const bool banded = TRACE && (d.flags & PAIR_BANDED) != 0;
    int32_t c_lo = 1, c_hi = (int32_t)n, c_plo = 1, c_phi = (int32_t)n;
    if (banded) {
      c_lo = imax(1, (int32_t)base + 1 + band_dmin(d));
      c_hi = imin32((int32_t)n, (int32_t)(base + rows_here) + band_dmax(d));
      if (p) {
        c_plo = imax(1, (int32_t)(base - 64u * K) + 1 + band_dmin(d));
        c_phi = imin32((int32_t)n, (int32_t)base + band_dmax(d));
      }
    }
    const uint32_t t_begin = (uint32_t)c_lo;
    const uint32_t t_end = (uint32_t)c_hi + lanes_used - 1;

    // Checkpointed score pass (single pass, free end gaps on row 0): rows are anchored at the BOTTOM of the
    // strips, so row m always sits in the last slot of the last used lane (its H is bot_h, its E is El[K-1]);
    // the `pad` slots above row 1 hold H = E = 0 with a zero extension cost and a zero substitution
    // score: they reproduce row 0 (H = 0) for every column and hand F = go+ge down, which makes row 1 open
    // its vertical gap from H exactly as it does against the -inf of the reference (needs ge < 0).
    const uint32_t pad = BOTTOM ? lanes_used * K - m : 0u;
    const int32_t goe = go + ge;
    const int32_t goe_n = NARROW ? goe : 0;  // the 16-bit kernel keeps Hg = H + (go+ge) (score_step16g)
    // ---- per-lane state at column 0 (gotoh.h:117-123) ----
    TraceLane<K> ts;
    ScoreLane<K> ss;
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const uint32_t r = base + L * K + i + 1 - pad;  // wraps for padding slots (r - 1 >= m)
      const bool hz = hfree && (r == m);
      const int32_t h0 = edge_value(vfree, go, ge, (int32_t)r);
      if (BOTTOM && (L * K + i < pad)) {
        ss.Hl[i] = goe_n; ss.El[i] = 0; ss.hopen[i] = go + ge; ss.hext[i] = 0;
      } else if (TRACE) {
        ts.Hc[i] = (c_lo > 1) ? neg : (int32_t)((uint32_t)h0 << SH);  // (banded: the column left of the window is outside the band)
        ts.Ec[i] = neg;
        ts.cx1[i] = trace_cx1(hz ? 0 : go + ge);
        ts.cx2[i] = trace_cx2(hz ? 0 : ge);
      } else {
        ss.Hl[i] = h0 + goe_n;
        ss.El[i] = neg;
        ss.hopen[i] = hz ? 0 : go + ge;
        ss.hext[i] = hz ? 0 : ge;
      }
    }
    const uint32_t row_above = (base + L * K > pad) ? base + L * K - pad : 0u;
    int32_t prev_up_h = ((row_above == 0) ? 0 : (int32_t)((uint32_t)edge_value(vfree, go, ge, (int32_t)row_above) << SH)) + goe_n;
    if (c_lo > 1) {  // banded, not the first pass: H(row above, c_lo - 1) is the hand-over row's value for lane 0 where the previous
                     // pass computed it (a step along the band's lowest diagonal), -inf everywhere else
      const int32_t cl = c_lo - 1;
      prev_up_h = (L == 0 && cl >= c_plo && cl <= c_phi) ? scratch[2 * cl] : neg;
    }
    const int32_t delta_last = (NARROW && hfree && L == lanes_used - 1) ? -goe : 0;  // row m: horizontal open costs 0
    int32_t bot_h = 0, bot_f = 0;

    // ---- substitution set-up for this pass ----
    const bool rc_view = (d.flags & PAIR_A2_REVCOMP) != 0;
    SubChar<K> sub_c;
    SubProf<K> sub_p;
    if (MODE == MODE_CHAR) {
#pragma unroll
      for (int i = 0; i < K; ++i) {
        const uint32_t r = base + L * K + i + 1 - pad;
        sub_c.rc[i] = (r - 1 < m) ? (int32_t)a1c[r - 1] : -1;
      }
      sub_c.vmatch = (int32_t)((uint32_t)a.match << SH) - goe_n;
      sub_c.vmis = (int32_t)((uint32_t)a.mismatch << SH) - goe_n;
      sub_c.cc = 0;
    } else if (qp_like(MODE)) {
      // query-profile table [6 codes][K rows][64 lanes] (qp6_index): entry = substitution score << SH (- goe_n)
      w.sync();  // previous pass may still be reading the table
      bool overflow = false;
      int32_t qabs = 0;
#pragma unroll 1
      for (int i = 0; i < K; ++i) {  // not unrolled: the set-up must not dictate the kernel's register budget
        const uint32_t r = base + L * K + i + 1 - pad;
        const bool real = r - 1 < m;
        float pr[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        uint8_t rch = 0;
        if (MODE == MODE_QP) {
#pragma unroll
          for (int k = 0; k < 5; ++k) pr[k] = real ? a1p[(uint64_t)k * d.a1_stride + (r - 1)] : 0.0f;
        } else {
          rch = real ? a1c[r - 1] : 0;
        }
#pragma unroll
        for (uint32_t b = 0; b < 5; ++b) {
          int32_t q;
          if (MODE == MODE_QP) q = real ? onehot_score(pr, b, fmatch, fmis) : 0;
          else q = real ? (rch == (uint8_t)"ACGTN"[b] ? a.match : a.mismatch) : 0;  // byte equality (align.h:96-101)
          const int32_t qs = (int32_t)((uint32_t)q << TSH) - goe_n;
          overflow |= (qs > 32767) || (qs < -32768) || (q > 32767) || (q < -32768);
          qabs = imax(qabs, q < 0 ? -q : q);
          // reverse-complement view of a2: the complement is folded into the table (the entries of code b serve code 3-b),
          // the sweep selects with the raw codes
          const uint32_t row = (rc_view && b < 4u) ? 3u - b : b;
          qp_tab[qp6_index<K>(row, (uint32_t)i, L)] = (int16_t)qs;
        }
        // code 5: '-' / any other letter.  A profile column of zeros scores 0; a string column that no row can equal mismatches
        const int32_t q5 = (MODE == MODE_CQ && real) ? a.mismatch : 0;
        qp_tab[qp6_index<K>(5u, (uint32_t)i, L)] = (int16_t)((int32_t)((uint32_t)q5 << TSH) - goe_n);
      }
      if (overflow) flag_error(a.err, 1);
      if (MODE == MODE_QP && qabs > a.qlimit) flag_max(a.err, 1, qabs);  // un-normalised profile: the host re-checks the value range (capi.hip)
      w.sync();
    } else {
#pragma unroll
      for (int i = 0; i < K; ++i) {
        // slots beyond row m (unused lanes, the tail of the last strip) repeat row m: their cells are never read, and a row of
        // zeros would score exactly 0 against every column -- an int the screen can never prove
        const uint32_t r = base + L * K + i + 1;
        const uint32_t rr = (r <= m) ? r : m;
#pragma unroll
        for (int k = 0; k < 5; ++k) sub_p.a[i][k] = a1p[(uint64_t)k * d.a1_stride + (rr - 1)];
      }
      sub_p.fmatch = fmatch;
      sub_p.fmis = fmis;
      sub_p.shift = SH;
      if (screen_pair) {
        // the ints of the float chain against the column classes 0..5 (column_class), [class][row][lane] like the query profile:
        // a one-hot column leaves the terms of its own row of weights (onehot_score), the uniform column is evaluated in full
        w.sync();
#pragma unroll 1
        for (int i = 0; i < K; ++i) {  // from memory, not unrolled: the set-up must not dictate the kernel's register budget
          const uint32_t r = base + L * K + i + 1;
          float pr[5];
#pragma unroll
          for (int k = 0; k < 5; ++k) pr[k] = (r <= m) ? a1p[(uint64_t)k * d.a1_stride + (r - 1)] : 0.0f;
#pragma unroll
          for (uint32_t b = 0; b < 5; ++b) qp_tab[qp6_index<K>(b, (uint32_t)i, L)] = (int16_t)onehot_score(pr, b, fmatch, fmis);
          const float uni[5] = {0.25f, 0.25f, 0.25f, 0.25f, 0.0f};
          qp_tab[qp6_index<K>(5u, (uint32_t)i, L)] = (int16_t)profile_score<5>(pr, uni, fmatch, fmis);
        }
        w.sync();
      }
    }

    // ---- anti-diagonal sweep ----
    // One step = one column of this lane's strip.  The per-column inputs are fetched ahead of use so that
    // their latency overlaps the K-cell compute of earlier steps instead of stalling the wave: the
    // reference code two steps ahead (it selects the LDS row), the LDS strip one step ahead into the other
    // half of a ping-pong pair (the loop is unrolled by two so no register copies are needed).
    auto col_at = [&](int32_t cc) -> uint32_t {  // clamp: prefetches of idle lanes stay in bounds
      const int32_t x = cc < 1 ? 1 : (cc > (int32_t)n ? (int32_t)n : cc);
      return a2_index(d, (uint32_t)x);
    };
    const bool rcflag = (d.flags & PAIR_A2_REVCOMP) != 0;
    auto do_step = [&](uint32_t t, const auto& sub) {
      const int32_t c = (int32_t)t - (int32_t)L;
      int32_t up_h, up_f;
      if (p == 0) {
        // row 0 (gotoh.h:112-116) enters through the shift itself: lane 0 has no source lane and keeps the DPP `old`
        // operand, which is H(0, t) / F(0, t) -- lane 0 sits in column c = t, so the value is scalar work
        up_h = w.shift_up_or(bot_h, (int32_t)((uint32_t)edge_value(hfree, go, ge, (int32_t)t) << SH) + goe_n);
        up_f = w.shift_up_or(bot_f, neg);
      } else {
        up_h = w.shift_up(bot_h);
        up_f = w.shift_up(bot_f);
      }
      const bool active = (uint32_t)(c - c_lo) <= (uint32_t)(c_hi - c_lo);  // c_lo <= c <= c_hi (1 .. n unless banded)
      if (active) {
        if (p != 0 && L == 0) {  // last row of the previous pass
          const bool held = !banded || (c >= c_plo && c <= c_phi);
          up_h = held ? scratch[2 * c] : neg;
          up_f = held ? scratch[2 * c + 1] : neg;
        }
        int32_t vopen = go + ge, vext = ge;
        if (vfree) {  // free end gap in the last column.  vfree is wave-uniform: keep it a scalar branch
#if defined(__HIP_DEVICE_COMPILE__)
          asm volatile("" ::: "memory");
#endif
          if (c == (int32_t)n) vopen = vext = 0;
        }
        int32_t nb_h, nb_f;
        uint32_t w0 = 0, w1 = 0;
        if (TRACE) trace_step<K>(ts, up_h, up_f, prev_up_h, trace_cy1(vopen), trace_cy2(vext), sub, w0, w1, nb_h, nb_f);
        else if (NARROW) score_step16g<K>(ss, up_h, up_f, prev_up_h, vext, goe, delta_last, sub, nb_h, nb_f);
        else if (A16) score_step16<K>(ss, up_h, up_f, prev_up_h, vopen, vext, sub, nb_h, nb_f);
        else score_step<K>(ss, up_h, up_f, prev_up_h, vopen, vext, sub, nb_h, nb_f);
        prev_up_h = up_h;
        bot_h = nb_h;
        bot_f = nb_f;
        if (TRACE && L < lanes_used) bits[word_index(p, t, L, n)] = ((uint64_t)w1 << 32) | w0;
        if (!last_pass && L == 63) {
          scratch[2 * c] = bot_h;
          scratch[2 * c + 1] = bot_f;
        }
        if (CKPT && L == lanes_used - 1) {  // {H, E} of row m (last slot of the last used lane) for the band traceback
          int32_t* lr = a.lastrow + d.lastrow_off;
          lr[2 * c] = NARROW ? sext16(bot_h - goe) : bot_h;
          lr[2 * c + 1] = NARROW ? sext16(ss.El[K - 1]) : ss.El[K - 1];
        }
      }
      if (CKPT && (t % a.ckpt_B) == 0) {  // wavefront checkpoint: the whole frontier, one coalesced store per field
        // raw registers are stored (the 16-bit kernel keeps its values in the low halves; the band traceback
        // sign-extends them on restore, DpArgs::ckpt_narrow)
        int32_t* ck = a.ckpt + d.ckpt_off + (uint64_t)(t / a.ckpt_B - 1) * (ckpt_fields(K) * 64u) + L;
#pragma unroll
        for (int i = 0; i < K; ++i) {
          ck[(uint32_t)i * 64u] = ss.Hl[i];
          ck[(uint32_t)(K + i) * 64u] = ss.El[i];
        }
        ck[(2u * K) * 64u] = bot_h;
        ck[(2u * K + 1) * 64u] = bot_f;
        ck[(2u * K + 2) * 64u] = prev_up_h;
      }
    };
    if (qp_like(MODE)) {
      // The code of column c = t - L (+ look-ahead) is read WITHOUT clamping: the code buffer carries kCodePad spare
      // bytes on both sides (capi_internal.h), idle lanes read into them (or into a neighbouring sequence) and discard
      // the result.  Forward view: byte c-1; reverse-complement view: byte n-c (its complement sits in the table).
      // byte(t) = lane_base + dir * t with dir = +-1 uniform over the wave: one VALU add per step.
      SubRows<K, RAWTAB ? kTagShift : 0> qa, qb;
      const int16_t* lane_col = qp_tab + L;
      const uint8_t* a2v = a2c - kCodeBias;
      const int32_t lane_base = (int32_t)kCodeBias + (rcflag ? (int32_t)n + (int32_t)L : -(int32_t)L - 1);
      const int32_t dir = rcflag ? -1 : 1;  // wave-uniform: dir * t is scalar work
      auto raw_at = [&](int32_t tt) -> uint32_t { return a2v[(uint32_t)(lane_base + dir * tt)]; };
      uint32_t raw_next;
      {
        const uint32_t raw1 = raw_at((int32_t)t_begin);
        raw_next = raw_at((int32_t)t_begin + 1);
        qp_fetch_rows<K>(lane_col, raw1, qa);
      }
      for (uint32_t t = t_begin; t <= t_end; t += 2) {
        {
          const uint32_t raw_nn = raw_at((int32_t)t + 2);
          qp_fetch_rows<K>(lane_col, raw_next, qb);
          do_step(t, qa);
          raw_next = raw_nn;
        }
        if (t + 1 > t_end) break;
        {
          const uint32_t raw_nn = raw_at((int32_t)t + 3);
          qp_fetch_rows<K>(lane_col, raw_next, qa);
          do_step(t + 1, qb);
          raw_next = raw_nn;
        }
      }
    } else if (MODE == MODE_CHAR) {
      int32_t cc_next = (int32_t)a2c[col_at(1 - (int32_t)L)];
      for (uint32_t t = 1; t <= t_end; ++t) {
        sub_c.cc = rcflag ? (int32_t)complement_char((uint8_t)cc_next) : cc_next;
        cc_next = (int32_t)a2c[col_at((int32_t)t - (int32_t)L + 1)];
        do_step(t, sub_c);
      }
    } else {
      // the column of a2 is fetched one step ahead of its use
      auto sweep = [&](auto& sub) {
        float nb[5];
        {
          const uint32_t ci = col_at(1 - (int32_t)L);
#pragma unroll
          for (int k = 0; k < 5; ++k) nb[k] = a2p[(uint64_t)k * d.a2_stride + ci];
        }
        // screen_on is wave-uniform: strips go through the screened score, and through the exact chain when some lane
        // cannot prove a truncation; a pair that keeps failing (columns whose scores ARE integers: one-hot or uniform
        // columns) stops screening
        bool screen_on = screen_pair;
        uint32_t nfail = 0;
        const uint8_t* cls = screen_on ? a.colcode + d.a2_off : nullptr;
        uint32_t ncls = kColClassOther;
        if (screen_on) {
          sub.screen_setup();
          ncls = cls[col_at(1 - (int32_t)L)];
        }
        for (uint32_t t = 1; t <= t_end; ++t) {
#pragma unroll
          for (int k = 0; k < 5; ++k) sub.b[k] = nb[k];
          const uint32_t ccls = ncls;
          const uint32_t ci = col_at((int32_t)t - (int32_t)L + 1);
#pragma unroll
          for (int k = 0; k < 5; ++k) nb[k] = a2p[(uint64_t)k * d.a2_stride + ci];
          if (screen_on) {
            ncls = cls[ci];
            int32_t bad = sub.screen();
            if (ccls < kColClassOther) {  // one-hot / uniform column: the ints were tabulated per row at set-up
              const int16_t* tp = qp_tab + qp6_index<K>(ccls, 0u, L);
#pragma unroll
              for (int i = 0; i < K; ++i) sub.sv[i] = (int32_t)((uint32_t)(int32_t)tp[i * 64] << SH);
              bad = 0;
            }
            if (w.ballot(bad < 0) != 0) {  // rare: some truncation of this step is not proven -- the strip takes the float chain
              sub.prepare();
              if (++nfail > 64u && 2u * nfail > t) screen_on = false;
            }
          } else {
            sub.prepare();
          }
          do_step(t, sub);
        }
      };
      auto sweep4 = [&]() {
        SubProf<K, 4> s4;
#pragma unroll
        for (int i = 0; i < K; ++i)
#pragma unroll
          for (int k = 0; k < 5; ++k) s4.a[i][k] = sub_p.a[i][k];
        s4.fmatch = fmatch; s4.fmis = fmis; s4.shift = SH;
        sweep(s4);
      };
      if constexpr (NT == 4) sweep4();
      else if constexpr (NT == 5) sweep(sub_p);
      else { if (skip4) sweep4(); else sweep(sub_p); }
    }
    (void)T;

    // ---- score = H[m][n] (gotoh.h:173) sits in the lane / slot that owns row m ----
    if (BOTTOM) {
      if (a.scores && L == lanes_used - 1) a.scores[d.out] = NARROW ? sext16(bot_h - goe) : bot_h;
    } else if (last_pass && a.scores) {
      const uint32_t g = m - 1 - base;
      if (L == g / K) {
        int32_t v = 0;
#pragma unroll
        for (int i = 0; i < K; ++i)
          if ((uint32_t)i == g % K) v = TRACE ? (ts.Hc[i] >> SH) : ((NARROW || A16) ? sext16(ss.Hl[i]) : ss.Hl[i]);
        a.scores[d.out] = v;
      }
    }
    if (!last_pass) w.sync_global();  // scratch written by lane 63 is read by lane 0 in the next pass
  }
}

// ------------------------------------------------------------------------------------------------
// The 16-bit query-profile sweep: gotohScore(trace profile, _createProfile(reference)) with AlignConfig<true,false>
// (sage.h:239-240, indigo.h:235-247), optionally leaving wavefront checkpoints + row m for the band traceback.
// This is the kernel the `tracy align` step spends three quarters of its time in; it is written for VALU issue:
//
//   * All values are kept minus (go+ge):  Hl = H,  El = E - goe,  f = F - goe.  The recurrences keep their form
//       E' = max(H_left, E'_left + ge)   F' = max(H_up, F'_up + ge)   H = max(H_diag + (sub - goe), E', F') + goe
//     (8 VALU ops per cell, v_add_u16 / v_max_i16), and row 0 becomes H = 0, F' <= 0: exactly what a DPP wave_shr:1 with
//     bound_ctrl writes into lane 0, so the hand-off of {H, F'} is two DPP moves and nothing else.
//   * The strip above hands its last H straight out of the state register; the value received in the previous step is the
//     diagonal of this one (two registers used alternately by the two halves of the unrolled loop): no copies.
//   * The reference code selects one of SIX tables (5 = all-zero column for '-' / other letters): address = lane column +
//     code * stride, one v_lshl_add / v_mad.  The table is laid out [code][row][lane]; every row of the strip arrives by its
//     own conflict-free 16-bit LDS read (LDS issue is not what this kernel is short of), so no cell needs its operand
//     shifted into place.
//   * Strips are swept by four asm statements per step (dp_lane.h strip_left16 / strip_down16).
//   * Between ramp-up and ramp-down every used lane is on a real column: those steps run without an activity test.
//   * Row m goes out as one {H, E'} pair of int16 per column (4 B instead of 8), checkpoints every ckpt_B steps.
// Domain (narrow_ok in capi.hip): one pass, hfree, !vfree, go <= 0, ge < 0, every value inside int16.
// ------------------------------------------------------------------------------------------------
template <bool G>
struct SweepGuard { static constexpr bool value = G; };  // tag: does a step test whether its lane is on a column?

template <int K>
struct QpStrip {
  uint32_t v[K];  // row i in the low half (16-bit LDS reads zero the high half; the 16-bit ops ignore it)
  TR_HD int32_t lo16(int i) const { return (int32_t)v[i]; }
};
// The sweep's own table: int16 entries at byte (row / 2) * (NC * 256) + code * 256 + (row % 2) * 128 + 2 * lane column.  A code is
// byte 1 of the address, so the address of a column's strip is ONE v_perm_b32 of the dword of codes and the lane's own byte
// (strip_of below) instead of a v_bfe + v_mad per step; the rows of the strip are immediate offsets of the reads.  NC = 6 (A C G T N,
// '-' / other) is 12 KB at K = 15; references that hold A C G T only -- almost all of them -- take the COMPACT form of the kernel
// with NC = 4: 8 KB.  The table is what decides how many waves a CU holds, and this kernel needs them: one wave issues a VALU
// instruction every ~4.5 cycles, a SIMD takes one every 2 (tools/ubench/clock_probe.hip); 13 -> 20 workgroups per CU was 32.4 ->
// 28.2 ms per launch.  Measured (round 5, one box, tools/ab.sh): the [code][row][lane] layout of 7.5 KB holds twenty workgroups on a
// CU, this one eighteen (4 608 pairs are one round of waves, 4 864 are not) -- 5 120 pairs 8.4 -> 9.4 ms, but 20 480 pairs 30.1 ->
// 29.0 ms and the `tracy align` step 19.8 -> 19.4 ms: the instruction saved counts for more than the twentieth workgroup.
TR_HD constexpr uint32_t lds_bytes_sweep16(int K, bool compact) { return (compact ? 4u : 6u) * 256u * (uint32_t)((K + 1) / 2); }
// (the prefix rows of the pruned sweeps, gotoh_prefix_body, lay their table out the same way)
TR_HD constexpr uint32_t lds_bytes_prefix(int K, bool compact) { return lds_bytes_sweep16(K, compact); }
template <int NC>
TR_HD constexpr uint32_t qp16_row_byte(uint32_t row) { return (row >> 1) * ((uint32_t)NC * 256u) + (row & 1u) * 128u; }
template <int NC>
TR_HD uint32_t qp16_byte(uint32_t code, uint32_t row, uint32_t lanecol) { return qp16_row_byte<NC>(row) + code * 256u + 2u * lanecol; }
// does the reference of this pair hold N / '-' / other codes?  (special_blocks: one byte per 256 code bytes.)  Wave-uniform.
template <class W>
TR_HD bool reference_is_plain(W& w, const DpArgs& a, const PairDesc& d) {
  if (!a.special_blocks) return false;
  if (d.n == 0) return true;
  const uint64_t first = d.a2_off >> 8, last = (d.a2_off + d.n - 1) >> 8;
  uint32_t seen = 0;
  for (uint64_t b = first + w.lane(); b <= last; b += 64) seen |= a.special_blocks[b];
  return w.ballot(seen != 0) == 0;
}
// `addr`: byte address of the strip's first row (code * 256 + the lane's byte; on the device an LDS address, on the host an offset
// into `tab`)
template <int K, int NC>
TR_HD void qp_fetch6(const char* tab, uint32_t addr, QpStrip<K>& q) {
#if defined(__HIP_DEVICE_COMPILE__)
  // Every LDS read of the sweep is issued by hand and waited for by hand (qp_wait6): one 16-bit read per row, so that no
  // cell needs its operand shifted into place (VALU work, which is what this kernel is short of), and no compiler
  // wait-count bookkeeping that would stall each step on the strip that was only just requested.
#pragma unroll
  for (int i = 0; i < K; ++i) asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(q.v[i]) : "v"(addr), "n"(qp16_row_byte<NC>((uint32_t)i)));
#else
  for (int i = 0; i < K; ++i) {
    uint16_t x;
    __builtin_memcpy(&x, tab + addr + qp16_row_byte<NC>((uint32_t)i), 2);
    q.v[i] = x;
  }
#endif
}
// Before a strip is used: LDS operations complete in order, so once at most the K reads of the NEXT strip (issued after
// this one) are outstanding, this strip has arrived.  The wait is tied to the registers it guards, so no use of them can
// be scheduled above it.
template <int K>
TR_HD void qp_wait6(QpStrip<K>& q) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int newer = K < 15 ? K : 15;  // lgkmcnt is a 4-bit counter
#define TR_TIE4(o) "+v"(q.v[o]), "+v"(q.v[o + 1]), "+v"(q.v[o + 2]), "+v"(q.v[o + 3])
  if constexpr (K == 15) asm volatile("s_waitcnt lgkmcnt(%15)" : TR_TIE4(0), TR_TIE4(4), TR_TIE4(8), "+v"(q.v[12]), "+v"(q.v[13]), "+v"(q.v[14]) : "n"(newer));
  else if constexpr (K == 16) asm volatile("s_waitcnt lgkmcnt(%16)" : TR_TIE4(0), TR_TIE4(4), TR_TIE4(8), TR_TIE4(12) : "n"(newer));
  else if constexpr (K == 12) asm volatile("s_waitcnt lgkmcnt(%12)" : TR_TIE4(0), TR_TIE4(4), TR_TIE4(8) : "n"(newer));
  else if constexpr (K == 8) asm volatile("s_waitcnt lgkmcnt(%8)" : TR_TIE4(0), TR_TIE4(4) : "n"(newer));
  else if constexpr (K == 4) asm volatile("s_waitcnt lgkmcnt(%4)" : TR_TIE4(0) : "n"(newer));
  else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef TR_TIE4
#endif
}

// accessors of the band traceback for what the sweep left behind (DpArgs::ckpt_narrow): row m per column, frontier fields
TR_HD int32_t lastrow_h(const int32_t* lr, uint32_t c, bool narrow) { return narrow ? sext16(lr[c]) : lr[2 * c]; }
TR_HD int32_t lastrow_e(const int32_t* lr, uint32_t c, bool narrow, int32_t goe) { return narrow ? (lr[c] >> 16) + goe : lr[2 * c + 1]; }

// STRINGS: a1 is a string over A C G T N and a2 holds case-sensitive codes (MODE_CQ): table entry = match / mismatch by byte
// equality (align.h:96-101), a column of any other letter mismatches every row
template <class W, int K, bool CKPT, bool COMPACT, bool STRINGS>
TR_HD void gotoh_narrow_qp_body(W& w, const DpArgs& a, uint32_t pair_idx) {
  const PairDesc d = a.pairs[pair_idx];
  if (d.flags & PAIR_SKIP) return;
  // both forms of the kernel are launched over the same pairs; each pair is swept by the one its reference calls for
  if (reference_is_plain(w, a, d) != COMPACT) return;
  const uint32_t L = w.lane();
  const uint32_t m = d.m, n = d.n;
  const int32_t go = a.go, ge = a.ge, goe = go + ge;
  if (m == 0 || n == 0) {  // only the init row / column exists (gotoh.h:106-123); hfree, !vfree
    if (L == 0 && a.scores) a.scores[d.out] = (m == 0) ? 0 : edge_value(false, go, ge, (int32_t)m);
    return;
  }
  const float* a1p = static_cast<const float*>(a.a1) + (STRINGS ? 0 : d.a1_off);
  const uint8_t* a1c = static_cast<const uint8_t*>(a.a1) + (STRINGS ? d.a1_off : 0);
  const uint8_t* a2c = static_cast<const uint8_t*>(a.a2) + d.a2_off;
  constexpr int NC = COMPACT ? 4 : 6;
  int16_t* qp_tab = reinterpret_cast<int16_t*>(w.lds());
  const float fmatch = (float)a.match, fmis = (float)a.mismatch;
  const uint32_t lanes_used = (m + K - 1) / K;
  const uint32_t pad = lanes_used * K - m;  // rows anchored at the bottom: row m is the last slot of the last used lane
  const uint32_t t_end = n + lanes_used - 1;
  const bool rcflag = (d.flags & PAIR_A2_REVCOMP) != 0;
  const bool lastlane = L == lanes_used - 1;

  // ---- per-lane state at column 0 (gotoh.h:117-123), minus (go+ge) ----
  int32_t Hl[K], El[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const uint32_t r = L * K + i + 1 - pad;
    if (L * K + i < pad) { Hl[i] = 0; El[i] = -goe; }  // padding slots above row 1 reproduce row 0: H = 0 in every column
    else { Hl[i] = edge_value(false, go, ge, (int32_t)r); El[i] = kNegInf16; }
  }
  int32_t gev = ge, goev = goe;
  int32_t hext_last = lastlane ? 0 : ge;      // row m: horizontal gaps are free (AlignConfig<true,.>)
  int32_t delta_last = lastlane ? -goe : 0;   // ... so E' = max(H - goe, E') there
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(gev), "+v"(goev), "+v"(hext_last), "+v"(delta_last));  // four live VGPRs for the whole sweep, not re-materialised per step
#endif

  // ---- query profile (qp16_byte), entry = (int)(sum_k p[k][row] w[k][b]) - goe for b = A C G T (N);
  // code 5 ('-' / other) and rows off the trace score 0 ----
  // Lane L's column of a row is 2 (L mod 32) + L / 32: a 16-bit read is served in two groups of 32 lanes, and with the lanes in order
  // lanes 2 j and 2 j + 1 share a dword -- a bank -- while they usually ask for different CODES (different dwords of that bank): every
  // read of the sweep was a two-way conflict (rocprofv3: SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE, the LDS busy 63 % of the
  // launch).  Interleaved like this the 32 lanes of a group hit 32 different banks whatever their codes.
  const uint32_t Lc = ((L & 31u) << 1) | (L >> 5);
  {
    bool overflow = false;
    int32_t qabs = 0;
#pragma unroll 1
    for (int i = 0; i < K; ++i) {  // (not unrolled: the set-up must not dictate the kernel's register budget)
      const uint32_t r = L * K + i + 1 - pad;
      const bool real = r - 1 < m;
      float pr[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
      uint8_t rch = 0;
      if (STRINGS) rch = real ? a1c[r - 1] : 0;
      else {
#pragma unroll
        for (int k = 0; k < 5; ++k) pr[k] = real ? a1p[(uint64_t)k * d.a1_stride + (r - 1)] : 0.0f;
      }
#pragma unroll
      for (uint32_t b = 0; b < (COMPACT ? 4u : 5u); ++b) {
        const int32_t q = !real ? 0 : STRINGS ? (rch == (uint8_t)"ACGTN"[b] ? a.match : a.mismatch) : onehot_score(pr, b, fmatch, fmis);
        const int32_t qs = q - goe;
        overflow |= (qs > 32767) || (qs < -32768) || (q > 32767) || (q < -32768);
        qabs = imax(qabs, q < 0 ? -q : q);
        const uint32_t row = (rcflag && b < 4u) ? 3u - b : b;  // reverse-complement view: the complement is folded into the table
        qp_tab[qp16_byte<NC>(row, (uint32_t)i, Lc) >> 1] = (int16_t)qs;
      }
      if (!COMPACT) qp_tab[qp16_byte<NC>(5u, (uint32_t)i, Lc) >> 1] = (int16_t)(((STRINGS && real) ? a.mismatch : 0) - goe);
    }
    if (overflow) flag_error(a.err, 1);
    if (!STRINGS && qabs > a.qlimit) flag_max(a.err, 1, qabs);
    w.sync();
  }

  // ---- sweep ----
  // the strip of a column: byte 1 of its address is the column's code, byte 0 this lane's column of the table
  const char* tabc = reinterpret_cast<const char*>(qp_tab);
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t lane_addr = (uint32_t)reinterpret_cast<uintptr_t>(tabc) + Lc * 2u;  // low half of the flat address = LDS byte address
  if ((lane_addr >> 8) & 0xffu) flag_error(a.err, 1);  // (the table starts the workgroup's LDS: byte 1 is free for the code)
#else
  const uint32_t lane_addr = Lc * 2u;
#endif
  const uint8_t* a2v = a2c - kCodeBias;
  const int32_t lane_base = (int32_t)kCodeBias + (rcflag ? (int32_t)n + (int32_t)L : -(int32_t)L - 1);
  const int32_t dir = rcflag ? -1 : 1;
  char* lrb = CKPT ? reinterpret_cast<char*>(a.lastrow + d.lastrow_off) : nullptr;
  const int32_t lr_lane = -4 * (int32_t)L;  // byte offset of column c = t - L in the row-m array: 4 t + lr_lane
  const uint32_t B = CKPT ? a.ckpt_B : 1u;
  bool keep = CKPT;  // does this sweep leave checkpoints and row m behind?  (wave-uniform)
  if (CKPT && a.votes) {
    const uint32_t tr = d.out % a.vote_nt;
    keep = !vote_skips_checkpoints(a.votes[2 * tr], a.votes[2 * tr + 1], d.out / a.vote_nt);
  }
#if defined(__HIP_DEVICE_COMPILE__)
  keep = __builtin_amdgcn_readfirstlane((int)keep) != 0;  // (a scalar: the steps branch on it)
#endif
  uint32_t ck_left = B;                                         // steps until the next wavefront checkpoint (wave-uniform)
  int32_t* ck_next = CKPT ? a.ckpt + d.ckpt_off + L : nullptr;   // its record
  int32_t f = 0;
  // the H received from the strip above: the two registers alternate between "this column's upper neighbour" and "the diagonal"
  const uint32_t row_above = (L * K > pad) ? L * K - pad : 0u;
  int32_t upA = 0, upB = (row_above == 0) ? 0 : edge_value(false, go, ge, (int32_t)row_above);

  // `row_m` receives this step's {H, E'} pair of row m (last lane only); the caller stores it
  auto step = [&](auto guard, uint32_t t, QpStrip<K>& q, int32_t& up_cur, const int32_t& diag, uint32_t& row_m) {
    constexpr bool GUARD = decltype(guard)::value;
    up_cur = w.shift_up(Hl[K - 1]);  // lane 0 receives 0 = H(0, t)
    f = w.shift_up(f);               // ... and F' = 0, which loses against H(0, t) + 0 as -inf would
    qp_wait6<K>(q);
    const bool active = !GUARD || (uint32_t)(t - 1u - L) < n;
    if (active) {
      int32_t sub[K];
#pragma unroll
      for (int i = 0; i < K; ++i) sub[i] = q.lo16(i);
      if constexpr (K == 15) {
        strip_left16<7, true>(Hl + 8, El + 8, sub + 8, Hl[7], gev, hext_last, delta_last);
        strip_left16<8, false>(Hl, El, sub, diag, gev, hext_last, delta_last);
        strip_down16<8>(Hl, El, up_cur, f, gev, goev);
        strip_down16<7>(Hl + 8, El + 8, Hl[7], f, gev, goev);
      } else if constexpr (K == 16) {
        strip_left16<8, true>(Hl + 8, El + 8, sub + 8, Hl[7], gev, hext_last, delta_last);
        strip_left16<8, false>(Hl, El, sub, diag, gev, hext_last, delta_last);
        strip_down16<8>(Hl, El, up_cur, f, gev, goev);
        strip_down16<8>(Hl + 8, El + 8, Hl[7], f, gev, goev);
      } else {
#pragma unroll
        for (int i = K - 1; i >= 0; --i) {
          const int32_t dg = i == 0 ? diag : Hl[i - 1];
          if (i == K - 1) cell_left16_last(Hl[i], El[i], hext_last, dg, sub[i], delta_last);
          else cell_left16(Hl[i], El[i], gev, dg, sub[i]);
        }
        int32_t uh = up_cur;
#pragma unroll
        for (int i = 0; i < K; ++i) { cell_down16(Hl[i], El[i], uh, f, gev, goev); uh = Hl[i]; }
      }
    }
    // a sweep that keeps nothing (the likely losing strand of a checkpointed launch) skips the row-m pack and the checkpoint test
    // behind ONE wave-uniform branch: VALU issue is what the kernel is short of, a scalar branch is not
    if (CKPT && keep) {
      TRACY_KEEP_BRANCH();
      if (active) {
        row_m = ((uint32_t)El[K - 1] << 16) | ((uint32_t)Hl[K - 1] & 0xffffu);
        if (GUARD && lastlane)  // ramp phases: row m column by column (the steady state stores four columns at once)
          *reinterpret_cast<uint32_t*>(lrb + (uint32_t)(4 * (int32_t)t + lr_lane)) = row_m;
      }
      if (t <= t_end && --ck_left == 0) {  // wavefront checkpoint every B steps: the whole frontier (raw registers), one coalesced store per field
        ck_left = B;
        int32_t* ck = ck_next;
        ck_next += ckpt_fields_qp16(K) * 64u;
#pragma unroll
        for (int i = 0; i < K; ++i) ck[(uint32_t)i * 64u] = (int32_t)(((uint32_t)El[i] << 16) | ((uint32_t)Hl[i] & 0xffffu));
        ck[(uint32_t)K * 64u] = (int32_t)(((uint32_t)f << 16) | ((uint32_t)up_cur & 0xffffu));
      }
    }
  };
  using Guarded = SweepGuard<true>;
  using Free = SweepGuard<false>;

  // Software pipeline, four steps per round (nothing is copied between the two halves of the ping-pong pairs):
  //   * reference codes: one unaligned dword per lane and round = the codes of its next four columns, requested two rounds
  //     (~8 steps) ahead, so the wait for it never sees the latency -- nor the acknowledgements of the row-m / checkpoint
  //     stores that share the memory counter on this architecture;
  //   * the strip of column t+1 is read from LDS while step t runs.
  // Forward view: column c is byte c-1; reverse-complement view: byte n-c (its complement sits in the table), so a round's
  // four columns are the dword's bytes 0..3 or 3..0 -- a wave-uniform choice of four shift counts.
  const int32_t byte0 = lane_base + (rcflag ? -3 : 0);  // dword of the round that starts at step t: a2v + byte0 + dir * t
  auto codes_at = [&](uint32_t tt) -> uint32_t {
    uint32_t v;
    __builtin_memcpy(&v, a2v + (uint32_t)(byte0 + dir * (int32_t)tt), 4);  // unaligned dword load
    return v;
  };
  // The in-flight request is issued and waited for by hand.  gfx950 counts loads and stores in ONE in-order counter; the
  // compiler's wait for this load, placed where the loop carries it round, would sit behind the acknowledgement of the
  // row-m store issued a moment earlier.  Here the wait comes BEFORE that store: whatever is older than the load (the stores
  // of the previous round) was issued four steps ago and is long acknowledged, so vmcnt(0) costs nothing.
#if defined(__HIP_DEVICE_COMPILE__)
  // the base address is the same for the whole wave; say so, for the scalar-base form of the load
  const uint64_t a2v_bits = reinterpret_cast<uint64_t>(a2v);
  // (the builtin returns a signed int: through uint32_t, or the low half sign-extends into the high one)
  const uint32_t a2v_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a2v_bits >> 32));
  const uint32_t a2v_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a2v_bits);
  const uint8_t* a2v_s = reinterpret_cast<const uint8_t*>(((uint64_t)a2v_hi << 32) | (uint64_t)a2v_lo);
#endif
  auto codes_request = [&](uint32_t tt, uint32_t& v) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t voff = (uint32_t)(byte0 + dir * (int32_t)tt);
    asm volatile("global_load_dword %0, %1, %2" : "=v"(v) : "v"(voff), "s"(a2v_s));
#else
    v = codes_at(tt);
#endif
  };
  auto codes_arrived = [&](uint32_t& v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v) : : "memory");  // "memory": the stores below stay below
#endif
  };
  // byte j of a round's dword (forward) or byte 3 - j (reverse-complement view) is the code of its step j: a wave-uniform selector
  // v_perm_b32 {lane_addr.b3, lane_addr.b2, codes.b[k], lane_addr.b0}
  const uint32_t sel0 = 0x03020400u + ((rcflag ? 3u : 0u) << 8), sel1 = 0x03020400u + ((rcflag ? 2u : 1u) << 8),
                 sel2 = 0x03020400u + ((rcflag ? 1u : 2u) << 8), sel3 = 0x03020400u + ((rcflag ? 0u : 3u) << 8);
  auto strip_of = [&](uint32_t codes, uint32_t sel) -> uint32_t {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(codes, lane_addr, sel);
#else
    return (((codes >> (8u * (((sel >> 8) & 0xffu) - 4u))) & 0xffu) << 8) | lane_addr;
#endif
  };
  uint32_t t = 1;
  uint32_t cw_cur = codes_at(1), cw_next = codes_at(5), cw_pend = 0;
  QpStrip<K> qa, qb;
  qp_fetch6<K, NC>(tabc, strip_of(cw_cur, sel0), qa);
  auto four_steps = [&](auto guard) {
    constexpr bool GUARD = decltype(guard)::value;
    codes_request(t + 8, cw_pend);
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r0, r1, r2, r3;  // (written and read by sweeps that keep row m only: no initialising moves in the loop of the others)
    asm volatile("" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3));
#else
    uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
#endif
    qp_fetch6<K, NC>(tabc, strip_of(cw_cur, sel1), qb);
    step(guard, t, qa, upA, upB, r0);
    qp_fetch6<K, NC>(tabc, strip_of(cw_cur, sel2), qa);
    step(guard, t + 1, qb, upB, upA, r1);
    qp_fetch6<K, NC>(tabc, strip_of(cw_cur, sel3), qb);
    step(guard, t + 2, qa, upA, upB, r2);
    qp_fetch6<K, NC>(tabc, strip_of(cw_next, sel0), qa);
    step(guard, t + 3, qb, upB, upA, r3);
    codes_arrived(cw_pend);
    if (CKPT && !GUARD && lastlane && keep) {  // {H, E'} of row m for the band traceback: one int16 pair per column, four columns per store
      const uint32_t v[4] = {r0, r1, r2, r3};
      __builtin_memcpy(lrb + (uint32_t)(4 * (int32_t)t + lr_lane), v, 16);
    }
    cw_cur = cw_next;
    cw_next = cw_pend;
    t += 4;
  };
  while (t < lanes_used && t <= t_end) four_steps(Guarded{});  // ramp-up
  while (t + 3 <= n) four_steps(Free{});                       // every used lane is on a column of the reference
  while (t <= t_end) four_steps(Guarded{});                    // ramp-down (steps past t_end find no lane on a column)

  if (a.scores && lastlane) a.scores[d.out] = sext16(Hl[K - 1]);
}

// ------------------------------------------------------------------------------------------------
// Origin-tracking sweep (string x string, AlignConfig<true,false>, one pass): gotoh() for callers that only need the
// two ends of the alignment -- trimReferenceSlice (fmindex.h:429-463) reads nothing else from it.  The reference's
// traceback from (m, n) first walks the trailing run of row m, whose horizontal moves are free: it stops at the last
// column c_e whose H(m, c) is strictly greater than E(m, c) (bit3 clear); from there it follows the trace bits to row
// 0, which it reaches at some column `lead` and leaves with `lead` leading 'h'.  origin_step carries `lead` in the
// low bits of every DP value, selected by the very maxima whose tie order defines the trace bits, and c_e is watched
// in the slot that owns row m.  Writes a.scores[out] = H(m, n), a.ends[2 out] = lead, a.ends[2 out + 1] = c_e.
// Domain (origin_ok() in the C ABI): m <= 64 K, n + 64 < 2^13, scores within the 14-bit field.
// ------------------------------------------------------------------------------------------------
constexpr int32_t kNegInfOrigin = -6000;

// TABLE: the rows are over {A,C,G,T,N} and a2 holds case-sensitive codes (MODE_CQ): substitution scores come from the
// [code][row][lane] table in LDS, one shift-add per cell instead of compare + select + add
// NC (with TABLE): code rows of the table.  The caller knows which characters the columns of the launch hold (encode_cq_kernel):
// A C G T only -> 4 rows, with N -> 5, anything else -> 6.  7.5 / 9.4 / 11.3 KB at K = 15: 20 / 17 / 14 workgroups per CU.
// TABLE: 0 = byte compare per cell (strings), 1 = strings through the table (MODE_CQ: rows over A C G T N, a2 holds
// case-sensitive codes), 2 = profile rows through the table (MODE_QP: a1 is a float profile, a2 holds reference codes) -- the
// preliminary alignment of `tracy align`, which only trimReferenceSlice reads (pipeline.hip)
template <class W, int K, int TABLE = 0, int NC = 6>
TR_HD void gotoh_origin_body(W& w, const DpArgs& a, uint32_t pair_idx) {
  const PairDesc d = a.pairs[pair_idx];
  // profile rows: two forms over the same pairs, as the 16-bit sweep -- four code rows for references of A C G T, six otherwise
  if (TABLE == 2 && reference_is_plain(w, a, d) != (NC == 4)) return;
  const uint32_t L = w.lane();
  const uint32_t m = d.m, n = d.n;
  const int32_t go = a.go, ge = a.ge;
  constexpr int SH = kOriginShift;
  constexpr int TS = kOriginBits;
  if (m == 0 || n == 0) {  // only the init row / column exists: all columns are leading 'h' (m == 0) or there are none
    if (L == 0) {
      if (a.scores) a.scores[d.out] = (m == 0) ? 0 : edge_value(false, go, ge, (int32_t)m);
      a.ends[2 * d.out] = (m == 0) ? n : 0u;
      a.ends[2 * d.out + 1] = (m == 0) ? n : 0u;
    }
    return;
  }
  const uint8_t* a1c = static_cast<const uint8_t*>(a.a1) + (TABLE == 2 ? 0 : d.a1_off);
  const float* a1p = static_cast<const float*>(a.a1) + (TABLE == 2 ? d.a1_off : 0);
  const uint8_t* a2c = static_cast<const uint8_t*>(a.a2) + d.a2_off;
  const uint32_t lanes_used = (m + K - 1) / K;
  const uint32_t t_end = n + lanes_used - 1;
  const uint32_t lane_m = (m - 1) / K, slot_m = (m - 1) % K;  // wave-uniform
  const int32_t neg = (int32_t)((uint32_t)kNegInfOrigin << SH);

  TraceLane<K> ts;
  SubChar<K> sub_c;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const uint32_t r = L * K + i + 1;
    const bool hz = (r == m);  // free horizontal gaps on the last row
    ts.Hc[i] = (int32_t)((uint32_t)edge_value(false, go, ge, (int32_t)r) << SH);  // column 0: origin 0
    ts.Ec[i] = neg;
    ts.cx1[i] = trace_cx1<TS>(hz ? 0 : go + ge);
    ts.cx2[i] = trace_cx2<TS>(hz ? 0 : ge);
    sub_c.rc[i] = (TABLE != 2 && r - 1 < m) ? (int32_t)a1c[r - 1] : -1;
  }
  sub_c.vmatch = (int32_t)((uint32_t)a.match << SH);
  sub_c.vmis = (int32_t)((uint32_t)a.mismatch << SH);
  sub_c.cc = 0;
  int16_t* qp_tab = reinterpret_cast<int16_t*>(w.lds());
  if (TABLE) {  // raw scores; the shift into the score field happens where a value is used (SubRows<K, SH>)
    const float fmatch = (float)a.match, fmis = (float)a.mismatch;
    int32_t qabs = 0;
#pragma unroll 1
    for (int i = 0; i < K; ++i) {
      const uint32_t r = L * K + i + 1;
      const bool real = r - 1 < m;
      const bool rcv = (d.flags & PAIR_A2_REVCOMP) != 0;
      uint8_t rch = 0;
      float pr[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
      if (TABLE == 2) {
#pragma unroll
        for (int k = 0; k < 5; ++k) pr[k] = real ? a1p[(uint64_t)k * d.a1_stride + (r - 1)] : 0.0f;
      } else {
        rch = real ? a1c[r - 1] : 0;
      }
#pragma unroll
      for (uint32_t b = 0; b < (NC < 5 ? (uint32_t)NC : 5u); ++b) {
        const uint32_t row = (rcv && b < 4u) ? 3u - b : b;
        int32_t q;
        if (TABLE == 2) q = real ? onehot_score(pr, b, fmatch, fmis) : 0;
        else q = real ? (rch == (uint8_t)"ACGTN"[b] ? a.match : a.mismatch) : 0;
        qabs = imax(qabs, q < 0 ? -q : q);
        qp_tab[qp6_index<K>(row, (uint32_t)i, L)] = (int16_t)q;
      }
      // code 5: a column no row letter can equal mismatches (strings); an all-zero profile column scores 0 (profiles)
      if (NC > 5) qp_tab[qp6_index<K>(5u, (uint32_t)i, L)] = (int16_t)((TABLE == 1 && real) ? a.mismatch : 0);
    }
    if (TABLE == 2 && qabs > a.qlimit) flag_max(a.err, 1, qabs);  // un-normalised profile: the packed score field is sized for |q| <= qlimit
    w.sync();
  }
  const uint32_t row_above = L * K;
  int32_t prev_up_h = (row_above == 0) ? 0 : (int32_t)((uint32_t)edge_value(false, go, ge, (int32_t)row_above) << SH);
  int32_t bot_h = 0, bot_f = 0;
  const int32_t cy1 = trace_cy1<TS>(go + ge), cy2 = trace_cy2<TS>(ge);
  uint32_t c_end = 0;

  const bool rcflag = (d.flags & PAIR_A2_REVCOMP) != 0;
  auto col_at = [&](int32_t cc) -> uint32_t {  // clamp: the look-ahead of idle lanes stays in bounds
    const int32_t x = cc < 1 ? 1 : (cc > (int32_t)n ? (int32_t)n : cc);
    return a2_index(d, (uint32_t)x);
  };
  // One step = one column of this lane's strip.  The strip above hands H of its last slot straight out of the state
  // register; the value received one step earlier is this step's diagonal (two registers used alternately by the two halves
  // of the unrolled loop: no copies).  Between ramp-up and ramp-down every used lane is on a real column: those steps run
  // without an activity test, so nothing has to be merged back into fixed registers at a join.
  // TABLE: code of column t - L read without clamping from the padded code buffer (as the query-profile sweeps), the strip
  // of the next column read row by row one step ahead
  SubRows<K, SH> sub_r, sub_n;
  const int16_t* lane_col = qp_tab + L;
  const uint8_t* a2v = a2c - kCodeBias;
  const int32_t lane_base = (int32_t)kCodeBias + (rcflag ? (int32_t)n + (int32_t)L : -(int32_t)L - 1);
  const int32_t dir = rcflag ? -1 : 1;
  auto raw_at = [&](int32_t tt) -> uint32_t { return a2v[(uint32_t)(lane_base + dir * tt)]; };
  uint32_t raw_next = 0;
  int32_t cc_next = 0;
  if (TABLE) {
    qp_fetch_rows<K>(lane_col, raw_at(1), sub_n);
    raw_next = raw_at(2);
  } else {
    cc_next = (int32_t)a2c[col_at(1 - (int32_t)L)];
  }
  int32_t f_x = 0;                 // F of the last slot, exchanged like H
  int32_t upA = 0, upB = prev_up_h;
  auto step = [&](auto guard, uint32_t t, int32_t& up_cur, const int32_t& diag) {
    constexpr bool GUARD = decltype(guard)::value;
    if (TABLE) {
      sub_r = sub_n;
      qp_fetch_rows<K>(lane_col, raw_next, sub_n);
      raw_next = raw_at((int32_t)t + 2);
    } else {
      sub_c.cc = rcflag ? (int32_t)complement_char((uint8_t)cc_next) : cc_next;
      cc_next = (int32_t)a2c[col_at((int32_t)t - (int32_t)L + 1)];
    }
    // row 0 (free leading gaps: H(0, c) = 0, F = -inf) enters through the shift; its origin is its own column
    up_cur = w.shift_up_or(ts.Hc[K - 1], (int32_t)t);
    const int32_t up_f = w.shift_up_or(f_x, neg);
    if (!GUARD || (uint32_t)(t - 1u - L) < n) {
      int32_t nb_h;
      if (TABLE) origin_step<K>(ts, up_cur, up_f, diag, cy1, cy2, sub_r, nb_h, f_x);
      else origin_step<K>(ts, up_cur, up_f, diag, cy1, cy2, sub_c, nb_h, f_x);
      // watch row m: slot_m is wave-uniform, the switch stays a scalar branch (no select chain over the strip)
      int32_t hv = 0, ev = 0;
      switch (slot_m) {
#define TRACY_ROW_M(I)                                  \
  case I:                                               \
    if (I < K) { hv = ts.Hc[I < K ? I : 0]; ev = ts.Ec[I < K ? I : 0]; } \
    TRACY_KEEP_BRANCH();                                \
    break;
        TRACY_ROW_M(0) TRACY_ROW_M(1) TRACY_ROW_M(2) TRACY_ROW_M(3) TRACY_ROW_M(4) TRACY_ROW_M(5) TRACY_ROW_M(6) TRACY_ROW_M(7)
        TRACY_ROW_M(8) TRACY_ROW_M(9) TRACY_ROW_M(10) TRACY_ROW_M(11) TRACY_ROW_M(12) TRACY_ROW_M(13) TRACY_ROW_M(14) TRACY_ROW_M(15)
#undef TRACY_ROW_M
        default: break;
      }
      if (L == lane_m && (hv >> SH) > (ev >> SH)) c_end = t - L;  // bit3 clear at (m, c): the trailing run ends here
    }
  };
  {
    using Guarded = SweepGuard<true>;
    using Free = SweepGuard<false>;
    uint32_t t = 1;
    while (t < lanes_used && t + 1 <= t_end) { step(Guarded{}, t, upA, upB); step(Guarded{}, t + 1, upB, upA); t += 2; }  // ramp-up
    while (t + 1 <= n) { step(Free{}, t, upA, upB); step(Free{}, t + 1, upB, upA); t += 2; }                              // steady state
    while (t + 1 <= t_end) { step(Guarded{}, t, upA, upB); step(Guarded{}, t + 1, upB, upA); t += 2; }                    // ramp-down
    if (t <= t_end) step(Guarded{}, t, upA, upB);
  }
  if (L == lane_m) {
    int32_t hv = 0;
#pragma unroll
    for (int i = 0; i < K; ++i)
      if ((uint32_t)i == slot_m) hv = ts.Hc[i];
    if (a.scores) a.scores[d.out] = hv >> SH;
    a.ends[2 * d.out] = (uint32_t)(hv & kOriginMask);
    a.ends[2 * d.out + 1] = c_end;
  }
}

// ------------------------------------------------------------------------------------------------
// Prefix bound of the semiglobal score (AlignConfig<true,false>, 16-bit formulation, profile x reference codes).
// GL lanes per pair, 64/GL pairs per wave: lane group g sweeps rows 1..R (R = GL*K) of its pair over all columns
// and reports  max_j max(H(R, j), F(R, j)).  Every alignment of the whole trace passes row R in state H or F, and
// each of the remaining rows adds at most max(0, its best substitution score) (gaps cost <= 0 in this domain), so
// that maximum + the row-maxima of rows R+1..m bounds gotohScore from above.  The align pipeline uses it to decide
// the strand without running the second orientation over all m rows (pipeline.hip).
// Domain (checked by the caller): hfree, !vfree, ge < 0, go <= 0, m > R, narrow_ok (int16 range).
// ------------------------------------------------------------------------------------------------
constexpr int kPrefixLanes = 8;  // lanes per pair of the prefix-bound kernel: rows 1 .. 8*K, eight pairs per wave
// the prefix of the pruned orientation sweep (front.h): sixteen lanes of eight rows -- four pairs per wave, twice the waves of the
// 8 x K shape for the same rows (a batch of 10 000 traces is 2 500 waves of the latter: two or three per SIMD, issue bound)
#ifndef TRACY_FRONT_LANES
#define TRACY_FRONT_LANES 16
#endif
constexpr int kFrontPrefixLanes = TRACY_FRONT_LANES, kFrontPrefixK = 8;
constexpr uint32_t kFrontRows = (uint32_t)kFrontPrefixLanes * kFrontPrefixK;

// COMPACT: the form for references of A C G T (four code rows in LDS); as with the sweeps, both forms are launched over the
// same pairs and every group of lanes works on its pair in the form the reference calls for (DpArgs::special_blocks)
// STRINGS: the rows are characters (gotoh(allele, window) of `tracy decompose`, indigo.h:359): match / mismatch by byte equality
// (align.h:96-101) against the codes A C G T N, mismatch against '-' / any other letter -- the table of the MODE_CQ sweeps
template <class W, int K, int GL, bool COMPACT = false, bool STRINGS = false>
TR_HD void gotoh_prefix_body(W& w, const DpArgs& a, uint32_t group_base, uint32_t npairs) {
  static_assert(64 % GL == 0, "whole groups per wave");
  constexpr uint32_t NCODES = COMPACT ? 4u : 5u;
  const uint32_t L = w.lane();
  const uint32_t Lg = L % GL;
  const uint32_t pair_idx = group_base + L / GL;
  if (a.count) {
    const uint32_t listed = *a.count;
    npairs = listed < npairs ? listed : npairs;
    if (group_base >= npairs) return;
  }
  bool valid = pair_idx < npairs;
  PairDesc d{};
  if (valid) d = a.pairs[a.index ? a.index[pair_idx] : pair_idx];
  valid = valid && !(d.flags & PAIR_SKIP);
  {
    bool plain = a.special_blocks != nullptr;  // every lane of a group looks at its pair's blocks: the same answer in all of them
    if (plain && valid && d.n)
      for (uint64_t b = d.a2_off >> 8; b <= (d.a2_off + d.n - 1) >> 8; ++b) plain = plain && a.special_blocks[b] == 0;
    valid = valid && plain == COMPACT;
    if (w.ballot(valid) == 0) return;
  }
  const uint32_t m = d.m, n = valid ? d.n : 0u;
  const int32_t go = a.go, ge = a.ge, goe = go + ge;
  const float fmatch = (float)a.match, fmis = (float)a.mismatch;
  const float* a1p = static_cast<const float*>(a.a1) + (STRINGS ? 0 : d.a1_off);
  const uint8_t* a1c = static_cast<const uint8_t*>(a.a1) + (STRINGS ? d.a1_off : 0);
  const uint8_t* a2c = static_cast<const uint8_t*>(a.a2) + d.a2_off;
  int16_t* qp_tab = reinterpret_cast<int16_t*>(w.lds());
  constexpr uint32_t R = (uint32_t)GL * K;

  // wave-uniform sweep length: the longest reference of the groups in this wave; nmin: the shortest of the groups that have a pair
  uint32_t nmax = 0, nmin = 0xffffffffu;
  for (uint32_t g = 0; g < 64u / GL; ++g) {
    const uint32_t ng = w.bcast(n, g * GL), vg = w.bcast(valid ? 1u : 0u, g * GL);
    nmax = ng > nmax ? ng : nmax;
    if (vg) nmin = ng < nmin ? ng : nmin;
  }
#if defined(__HIP_DEVICE_COMPILE__)
  nmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)nmax);  // (wave-uniform: the loops below are scalar loops)
  nmin = (uint32_t)__builtin_amdgcn_readfirstlane((int)nmin);
#endif
  const uint32_t t_end = nmax + GL - 1;

  // The sweep is the 16-bit sweep's (gotoh_narrow_qp_body), GL lanes to a pair: values minus (go+ge) -- Hl = H, El = E - goe,
  // f = F - goe --, so row 0 is H = 0, F' = 0: what a DPP row_shr:1 writes into the first lane of a row of sixteen (the first lane
  // of a group of eight gets it through the lane mask `gmask`); the table [code page][row][lane] read by one 16-bit LDS read per
  // row, issued and waited for by hand; the codes of four columns one dword, requested two rounds ahead; strips swept by asm
  // statements; between ramp-up and ramp-down the steps test nothing.  (Rounds 1-4 ran the generic lane code here -- packed dword
  // strips with an unpacking shift per odd row, selects for row 0, a guard per step: 10.5-12.2 instructions per cell, half of the
  // LDS cycles bank conflicts, a fifth of the wave cycles in s_waitcnt.)
  // per-lane state at column 0 (gotoh.h:117-123): H(r, 0) = go + r*ge
  int32_t Hl[K], El[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    Hl[i] = edge_value(false, go, ge, (int32_t)(Lg * K + i + 1));
    El[i] = kNegInf16;
  }
  int32_t gev = ge, goev = goe;
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(gev), "+v"(goev));  // live VGPRs for the whole sweep, not re-materialised per step
#endif
  const uint32_t row_above = Lg * K;
  int32_t upA = 0, upB = (row_above == 0) ? 0 : edge_value(false, go, ge, (int32_t)row_above);
  int32_t f = 0;
  const bool rcflag = (d.flags & PAIR_A2_REVCOMP) != 0;

  // query profile (qp16_byte): entry = score - (go+ge); the complement of a reverse-complement view is folded into the table
  constexpr int NC = COMPACT ? 4 : 6;
  const uint32_t Lc = ((L & 31u) << 1) | (L >> 5);
  {
    bool overflow = false;
    int32_t qabs = 0;
#pragma unroll 1
    for (int i = 0; i < K; ++i) {
      const uint32_t r = Lg * K + i + 1;
      const bool real = valid && r - 1 < m;
      float pr[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
      uint8_t rch = 0;
      if (STRINGS) rch = real ? a1c[r - 1] : 0;
      else {
#pragma unroll
        for (int k = 0; k < 5; ++k) pr[k] = real ? a1p[(uint64_t)k * d.a1_stride + (r - 1)] : 0.0f;
      }
#pragma unroll
      for (uint32_t b = 0; b < NCODES; ++b) {
        int32_t q = 0;
        if (real) q = STRINGS ? (rch == (uint8_t)"ACGTN"[b] ? a.match : a.mismatch) : onehot_score(pr, b, fmatch, fmis);
        const int32_t qs = q - goe;
        overflow |= (qs > 32767) || (qs < -32768) || (q > 32767) || (q < -32768);
        qabs = imax(qabs, q < 0 ? -q : q);
        const uint32_t row = (rcflag && b < 4u) ? 3u - b : b;
        qp_tab[qp16_byte<NC>(row, (uint32_t)i, Lc) >> 1] = (int16_t)qs;
      }
      // '-' / any other letter: an all-zero profile column scores 0, a string column that no row can equal mismatches
      if (!COMPACT) qp_tab[qp16_byte<NC>(5u, (uint32_t)i, Lc) >> 1] = (int16_t)((STRINGS ? a.mismatch : 0) - goe);
    }
    if (overflow) flag_error(a.err, 1);
    if (!STRINGS && qabs > a.qlimit) flag_max(a.err, 1, qabs);
    w.sync();
  }
  const char* tabc = reinterpret_cast<const char*>(qp_tab);
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t lane_addr = (uint32_t)reinterpret_cast<uintptr_t>(tabc) + Lc * 2u;
  if ((lane_addr >> 8) & 0xffu) flag_error(a.err, 1);  // (the table starts the workgroup's LDS: byte 1 is free for the code)
#else
  const uint32_t lane_addr = Lc * 2u;
#endif
  // byte j of a round's dword (forward) or byte 3 - j (reverse-complement view) is the code of its step j -- per lane here
  const uint32_t sel0 = 0x03020400u + ((rcflag ? 3u : 0u) << 8), sel1 = 0x03020400u + ((rcflag ? 2u : 1u) << 8),
                 sel2 = 0x03020400u + ((rcflag ? 1u : 2u) << 8), sel3 = 0x03020400u + ((rcflag ? 0u : 3u) << 8);
  auto strip_of = [&](uint32_t codes, uint32_t sel) -> uint32_t {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(codes, lane_addr, sel);
#else
    return (((codes >> (8u * (((sel >> 8) & 0xffu) - 4u))) & 0xffu) << 8) | lane_addr;
#endif
  };

  // The codes of the four columns c .. c + 3 of a lane (c = t - Lg at the round that starts with step t) are one dword: bytes
  // c - 1 .. c + 2 of the window, or n - c - 3 .. n - c of its reverse-complement view.  Lanes off their window read the pad of the
  // code buffer (kCodePad) or a neighbouring window -- codes either way -- and drop the result; the clamp keeps a short window's
  // lanes from running on to the end of the longest one's.
  // (a group without a pair in this form -- beyond the list, skipped, the other form's -- stays on the first bytes of its window:
  // its lanes run through the steady state of the others and must not walk on behind the buffer)
  const uint8_t* code_base = !valid ? a2c : a2c + (rcflag ? (int64_t)n - 3 : (int64_t)-1);
  const int32_t code_dir = !valid ? 0 : (rcflag ? -1 : 1);
  auto codes_ptr = [&](uint32_t tt) -> const uint8_t* {
    const int32_t c0 = (int32_t)tt - (int32_t)Lg;
    const int32_t x = c0 < -64 ? -64 : (c0 > (int32_t)n + 64 ? (int32_t)n + 64 : c0);
    return code_base + (int64_t)(code_dir * x);
  };
  auto codes_at = [&](uint32_t tt) -> uint32_t {
    uint32_t v;
    __builtin_memcpy(&v, codes_ptr(tt), 4);
    return v;
  };
  auto codes_request = [&](const uint8_t* ptr, uint32_t& v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(ptr));
#else
    __builtin_memcpy(&v, ptr, 4);
#endif
  };
  auto codes_arrived = [&](uint32_t& v) {  // before the round's stores: gfx950 counts loads and stores in one in-order counter
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v) : : "memory");
#endif
  };

  // running maxima of H and F' of row R, and the row itself (PAIR_KEEP_ROW): the last slot of the group's last lane
  const bool lastlane = Lg == GL - 1;
  int32_t mx_h = edge_value(false, go, ge, (int32_t)R), mx_f = kNegInf16;
  uint32_t* keep_row = (valid && lastlane && (d.flags & PAIR_KEEP_ROW) && a.lastrow) ? reinterpret_cast<uint32_t*>(a.lastrow + d.lastrow_off) : nullptr;
  const uint32_t goe2 = ((uint32_t)goe << 16) | ((uint32_t)goe & 0xffffu);
  // the lane above: inside a row of sixteen lanes (a group is a row, or half of one whose first lane is masked)
  const int32_t gmask = (GL < 16 && Lg == 0) ? 0 : -1;
  // (one DPP instruction each: the mask rides in the shift -- v_and_b32_dpp; a VALU write needs two wait states before a DPP read)
  auto shift_group = [&](int32_t h, int32_t fv, int32_t& h_up, int32_t& f_up) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("s_nop 1\n\tv_and_b32_dpp %0, %2, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_and_b32_dpp %1, %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "=&v"(h_up), "=&v"(f_up) : "v"(h), "v"(fv), "v"(gmask));
#else
    h_up = w.shift_up_row(h) & gmask;
    f_up = w.shift_up_row(fv) & gmask;
#endif
  };

  // `kept`: {H + goe, F} of row R at this step's column, the format the stages below the kept row read (front.h, band16.h CONT)
  auto step = [&](auto guard, uint32_t t, QpStrip<K>& q, int32_t& up_cur, const int32_t& diag, uint32_t& kept) {
    constexpr bool GUARD = decltype(guard)::value;
    shift_group(Hl[K - 1], f, up_cur, f);
    qp_wait6<K>(q);
    const bool active = !GUARD || (uint32_t)(t - 1u - Lg) < n;
    if (active) {
      int32_t sub[K];
#pragma unroll
      for (int i = 0; i < K; ++i) sub[i] = q.lo16(i);
      if constexpr (K == 8) {
        strip_left16<8, false>(Hl, El, sub, diag, gev, gev, 0);
        strip_down16<8>(Hl, El, up_cur, f, gev, goev);
      } else if constexpr (K == 16) {
        strip_left16<8, false>(Hl + 8, El + 8, sub + 8, Hl[7], gev, gev, 0);
        strip_left16<8, false>(Hl, El, sub, diag, gev, gev, 0);
        strip_down16<8>(Hl, El, up_cur, f, gev, goev);
        strip_down16<8>(Hl + 8, El + 8, Hl[7], f, gev, goev);
      } else if constexpr (K == 15) {
        strip_left16<7, false>(Hl + 8, El + 8, sub + 8, Hl[7], gev, gev, 0);
        strip_left16<8, false>(Hl, El, sub, diag, gev, gev, 0);
        strip_down16<8>(Hl, El, up_cur, f, gev, goev);
        strip_down16<7>(Hl + 8, El + 8, Hl[7], f, gev, goev);
      } else {
#pragma unroll
        for (int i = K - 1; i >= 0; --i) cell_left16(Hl[i], El[i], gev, i == 0 ? diag : Hl[i - 1], sub[i]);
        int32_t uh = up_cur;
#pragma unroll
        for (int i = 0; i < K; ++i) { cell_down16(Hl[i], El[i], uh, f, gev, goev); uh = Hl[i]; }
      }
      mx_h = max16(mx_h, Hl[K - 1]);
      mx_f = max16(mx_f, f);
#if defined(__HIP_DEVICE_COMPILE__)
      asm("v_perm_b32 %0, %1, %2, %3\n\tv_pk_add_u16 %0, %0, %4" : "=&v"(kept) : "v"(f), "v"(Hl[K - 1]), "s"(0x05040100u), "v"(goe2));
#else
      kept = ((uint32_t)(Hl[K - 1] + goe) & 0xffffu) | ((uint32_t)(f + goe) << 16);
#endif
      if (GUARD && keep_row) keep_row[t - Lg] = kept;  // ramp phases: column by column (the steady state stores four at once)
    }
  };
  using Guarded = SweepGuard<true>;
  using Free = SweepGuard<false>;
  uint32_t t = 1;
  uint32_t cw_cur = codes_at(1), cw_next = codes_at(5), cw_pend = 0;
  QpStrip<K> qa, qb;
  qp_fetch6<K, NC>(tabc, strip_of(cw_cur, sel0), qa);
  const uint8_t* code_run = nullptr;  // the unclamped request pointer of the steady state
  auto four_steps = [&](auto guard) {
    constexpr bool GUARD = decltype(guard)::value;
    if (GUARD) codes_request(codes_ptr(t + 8), cw_pend);
    else {
      codes_request(code_run, cw_pend);
      code_run += 4 * code_dir;
    }
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r0, r1, r2, r3;  // (no initialising moves in the loop)
    asm volatile("" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3));
#else
    uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
#endif
    qp_fetch6<K, NC>(tabc, strip_of(cw_cur, sel1), qb);
    step(guard, t, qa, upA, upB, r0);
    qp_fetch6<K, NC>(tabc, strip_of(cw_cur, sel2), qa);
    step(guard, t + 1, qb, upB, upA, r1);
    qp_fetch6<K, NC>(tabc, strip_of(cw_cur, sel3), qb);
    step(guard, t + 2, qa, upA, upB, r2);
    qp_fetch6<K, NC>(tabc, strip_of(cw_next, sel0), qa);
    step(guard, t + 3, qb, upB, upA, r3);
    codes_arrived(cw_pend);
    if (!GUARD && keep_row) {  // four columns of row R in one store
      const uint32_t v[4] = {r0, r1, r2, r3};
      __builtin_memcpy(keep_row + (t - Lg), v, 16);
    }
    cw_cur = cw_next;
    cw_next = cw_pend;
    t += 4;
  };
  while (t < (uint32_t)GL && t <= t_end) four_steps(Guarded{});  // ramp-up
  if (nmin != 0xffffffffu && t + 3 <= nmin) {
    code_run = code_base + (int64_t)code_dir * ((int64_t)t + 8 - (int64_t)Lg);  // (inside every window of the wave, pads included)
    while (t + 3 <= nmin) four_steps(Free{});  // every lane of every group that has a pair is on a column of its window
  }
  while (t <= t_end) four_steps(Guarded{});  // ramp-down: the windows end one after the other
  if (valid && lastlane && a.scores) {
    const int32_t h = sext16(mx_h), fm = sext16(mx_f) + goe;
    a.scores[d.out] = h > fm ? h : fm;
  }
}

// ------------------------------------------------------------------------------------------------
// Traceback walker (gotoh.h:143-167): one lane per pair.  Emits the reference's `btr` in push order.
// ------------------------------------------------------------------------------------------------
struct WalkArgs {
  const PairDesc* pairs;
  const uint64_t* bits;
  uint8_t* ops;               // output, pair i at ops + ops_off[i]
  const uint64_t* ops_off;    // device copy of the host offsets, indexed by PairDesc::out
  uint32_t* ops_len;          // indexed by PairDesc::out
  int32_t* err;               // bit 1: walk left the matrix (corrupt planes / absurd parameters)
  uint32_t npairs;
  int32_t K;
};

TR_HD void gotoh_walk_one(const WalkArgs& a, uint32_t pair_idx) {
  const PairDesc d = a.pairs[pair_idx];
  const uint64_t* bits = a.bits + d.bits_off;
  uint8_t* out = a.ops + a.ops_off[d.out];
  const uint32_t n = d.n;
  uint32_t row = d.m, col = d.n, k = 0;
  char last = 's';
  const uint32_t limit = d.m + d.n;
  while ((row > 0 || col > 0) && k <= limit) {
    TraceBits b;
    if (row == 0) {  // first row: bit3 set for col > 0, nothing else (gotoh.h:112-116)
      b.bit1 = b.bit2 = b.bit4 = false;
      b.bit3 = true;
    } else if (col == 0) {  // first column: bit4 only (gotoh.h:117-123)
      b.bit1 = b.bit2 = b.bit3 = false;
      b.bit4 = true;
    } else {
      const CellAddr ca = cell_addr(row, a.K);
      const uint64_t wd = bits[word_index(ca.pass, col + ca.lane, ca.lane, n)];
      b = decode_nibble((uint32_t)(wd >> (4u * ca.slot)) & 15u);
    }
    if (last == 's') {
      if (b.bit3) last = 'h';
      else if (b.bit4) last = 'v';
      else { --row; --col; out[k++] = 's'; }
    } else if (last == 'h') {
      if (col == 0) break;  // unreachable with sane parameters; the reference would run off the matrix
      if (b.bit1) last = 's';
      --col;
      out[k++] = 'h';
    } else {
      if (row == 0) break;
      if (b.bit2) last = 's';
      --row;
      out[k++] = 'v';
    }
  }
  if (row > 0 || col > 0) flag_error(a.err, 2);
  a.ops_len[d.out] = k;
}


// ------------------------------------------------------------------------------------------------
// Wave-cooperative traceback walker: the state machine of gotoh.h:143-167, one wave per pair, up to 64
// cells per memory round trip.  In state 's' the walk continues diagonally while a cell has neither bit3
// nor bit4; in 'h' ('v') it continues left (up) until the first cell with bit1 (bit2).  Each lane fetches
// the nibble of one candidate cell of the current run, ballots find where the run ends, and the run is
// emitted with one coalesced store.  W provides lane(), ballot(pred), bcast(x, l).
//
// walk_core walks while the current cell is an interior cell whose sweep step (column + owning lane)
// is > tmin; fetch(r, c) returns the stored nibble of an interior cell inside that range.
// ------------------------------------------------------------------------------------------------
TR_HD uint32_t first_set(uint64_t m) {
  uint32_t i = 0;
  if (!m) return 64;
  while (!((m >> i) & 1ull)) ++i;
  return i;
}

template <class W, class Fetch>
TR_HD void walk_core(W& w, const Fetch& fetch, uint32_t& row, uint32_t& col, int& state, uint32_t& k, uint8_t* out,
                     uint32_t tmin, int K, uint32_t limit, uint32_t pad = 0) {
  const uint32_t lane = w.lane();
  while (row > 0 && col > 0 && k <= limit) {
    if (col + cell_addr(row + pad, K).lane <= tmin) break;  // the current cell belongs to an earlier band
    const uint32_t r = (state == 1) ? row : row - lane;
    const uint32_t c = (state == 2) ? col : col - lane;
    const bool inside = (state == 1) ? (lane < col) : (state == 2) ? (lane < row) : (lane < row && lane < col);
    const bool inband = inside && (c + cell_addr(r + pad, K).lane > tmin);
    TraceBits b = {false, false, false, false};
    if (inband) b = decode_nibble(fetch(r, c));
    const bool hit = inband && (state == 0 ? (b.bit3 || b.bit4) : state == 1 ? b.bit1 : b.bit2);
    const uint32_t first_hit = first_set(w.ballot(hit));
    const uint32_t first_out = first_set(w.ballot(!inband));
    if (state == 0) {  // cells before the first hit / boundary are diagonal steps
      const uint32_t x = first_hit < first_out ? first_hit : first_out;
      if (lane < x) out[k + lane] = 's';
      k += x; row -= x; col -= x;
      if (first_hit < first_out) state = (int)w.bcast((uint32_t)(b.bit3 ? 1 : 2), first_hit);  // switch matrix, no move
    } else if (state == 1) {  // every visited cell emits 'h'; the one with bit1 is the last of the run
      const bool found = first_hit < first_out;
      const uint32_t x = found ? first_hit + 1 : first_out;
      if (lane < x) out[k + lane] = 'h';
      k += x; col -= x;
      if (found) state = 0;
    } else {
      const bool found = first_hit < first_out;
      const uint32_t x = found ? first_hit + 1 : first_out;
      if (lane < x) out[k + lane] = 'v';
      k += x; row -= x;
      if (found) state = 0;
    }
  }
}

// first row / first column tails (gotoh.h:112-123): 'h' down to column 0, or 'v' down to row 0
template <class W>
TR_HD bool walk_tails(W& w, uint32_t& row, uint32_t& col, int state, uint32_t& k, uint8_t* out) {
  const uint32_t lane = w.lane();
  if (row == 0) {
    if (state == 2 && col > 0) return false;  // 'v' ran into row 0: unreachable with sane parameters
    for (uint32_t i = lane; i < col; i += 64) out[k + i] = 'h';
    k += col; col = 0;
  } else if (col == 0) {
    if (state == 1) return false;
    for (uint32_t i = lane; i < row; i += 64) out[k + i] = 'v';
    k += row; row = 0;
  }
  return true;
}

struct FullMatrixFetch {
  const uint64_t* bits;
  uint32_t n;
  int K;
  TR_HD uint32_t operator()(uint32_t r, uint32_t c) const {
    const CellAddr ca = cell_addr(r, K);
    const uint64_t wd = bits[word_index(ca.pass, c + ca.lane, ca.lane, n)];
    return (uint32_t)(wd >> (4u * ca.slot)) & 15u;
  }
};

template <class W>
TR_HD void gotoh_walk_wave(W& w, const WalkArgs& a, uint32_t pair_idx) {
  const PairDesc d = a.pairs[pair_idx];
  uint8_t* out = a.ops + a.ops_off[d.out];
  uint32_t row = d.m, col = d.n, k = 0;
  int state = 0;
  FullMatrixFetch fetch{a.bits + d.bits_off, d.n, a.K};
  walk_core(w, fetch, row, col, state, k, out, 0u, a.K, d.m + d.n);
  const bool ok = walk_tails(w, row, col, state, k, out);
  if (!ok || row > 0 || col > 0) {
    if (w.lane() == 0) flag_error(a.err, 2);
  }
  if (w.lane() == 0) a.ops_len[d.out] = k;
}

// ------------------------------------------------------------------------------------------------
// Band traceback: the traceback of a pair whose score pass left wavefront checkpoints every ckpt_B steps
// and the {H, E} values of row m.  Instead of storing 0.5 B for every cell of the matrix, only the bands
// of the sweep that the path actually crosses are recomputed (with the tagged traceback arithmetic) into
// a small per-workgroup buffer and walked at once.  The recomputed nibbles are functions of exact DP
// values restored from the checkpoints, so the emitted btr is identical to the full-matrix traceback.
//   1. row m: the trailing run is decided from the saved {H, E}: bit3 = (H == E), bit1 = (E != Eleft + hext);
//      as soon as the walk wants to leave row m (bit3 clear) the band machinery takes over at that cell.
//   2. bands: restore the frontier of step j*B, sweep steps j*B+1 .. step(current cell), walk inside the band.
// Single-pass problems (m <= 64*K) with string or query-profile scoring.
// ------------------------------------------------------------------------------------------------
struct BandFetch {
  const uint64_t* band;
  uint32_t t0;
  int K;
  uint32_t pad;
  TR_HD uint32_t operator()(uint32_t r, uint32_t c) const {
    const CellAddr ca = cell_addr(r + pad, K);
    const uint64_t wd = band[(uint64_t)(c + ca.lane - t0 - 1u) * 64u + ca.lane];
    return (uint32_t)(wd >> (4u * ca.slot)) & 15u;
  }
};

template <class W, int K, int MODE>
TR_HD void gotoh_band_trace_body(W& w, const DpArgs& a, const WalkArgs& wa, uint32_t pair_idx) {
  static_assert(MODE == MODE_CHAR || MODE == MODE_QP, "band traceback: string or query-profile scoring");
  const PairDesc d = a.pairs[pair_idx];
  const uint32_t L = w.lane();
  const uint32_t m = d.m, n = d.n;
  const int32_t go = a.go, ge = a.ge;
  const bool hfree = a.hfree != 0, vfree = a.vfree != 0;
  constexpr int SH = kTagShift;
  uint8_t* out = wa.ops + wa.ops_off[d.out];
  uint32_t row = m, col = n, k = 0;
  int state = 0;
  bool ok = true;
  if (m > 0 && n > 0) {
    const uint8_t* a1c = static_cast<const uint8_t*>(a.a1) + (MODE == MODE_CHAR ? d.a1_off : 0);
    const float* a1p = static_cast<const float*>(a.a1) + (MODE != MODE_CHAR ? d.a1_off : 0);
    const uint8_t* a2c = static_cast<const uint8_t*>(a.a2) + d.a2_off;
    int16_t* qp_tab = reinterpret_cast<int16_t*>(w.lds());
    const float fmatch = (float)a.match, fmis = (float)a.mismatch;
    const int32_t neg = (int32_t)((uint32_t)kNegInf << SH);
    const uint32_t lanes_used = (m + K - 1) / K;
    const uint32_t pad = lanes_used * K - m;  // bottom-anchored rows, as in the checkpointed score pass
    const uint32_t B = a.ckpt_B;
    const int32_t* ck = a.ckpt + d.ckpt_off;
    const int32_t* lr = a.lastrow + d.lastrow_off;
    uint64_t* band = a.band + (uint64_t)pair_idx * B * 64u;
    const bool rcflag = (d.flags & PAIR_A2_REVCOMP) != 0;

    // ---- substitution set-up (as gotoh_body) ----
    SubChar<K> sub_c;
    if (MODE == MODE_CHAR) {
#pragma unroll
      for (int i = 0; i < K; ++i) {
        const uint32_t r = L * K + i + 1 - pad;
        sub_c.rc[i] = (r - 1 < m) ? (int32_t)a1c[r - 1] : -1;
      }
      sub_c.vmatch = (int32_t)((uint32_t)a.match << SH);
      sub_c.vmis = (int32_t)((uint32_t)a.mismatch << SH);
      sub_c.cc = 0;
    } else {
      bool band_overflow = false;
#pragma unroll 1
      for (int i = 0; i < K; ++i) {
        const uint32_t r = L * K + i + 1 - pad;
        float pr[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) pr[q] = (r - 1 < m) ? a1p[(uint64_t)q * d.a1_stride + (r - 1)] : 0.0f;
#pragma unroll
        for (uint32_t bb = 0; bb < 5; ++bb) {
          const int32_t qv = (r - 1 < m) ? onehot_score(pr, bb, fmatch, fmis) : 0;
          const uint32_t rowsel = (rcflag && bb < 4u) ? 3u - bb : bb;  // complement folded into the table (as gotoh_body)
          band_overflow |= (qv > (32767 >> SH)) || (qv < -(32768 >> SH));
          qp_tab[qp6_index<K>(rowsel, (uint32_t)i, L)] = (int16_t)((uint32_t)qv << SH);
        }
        qp_tab[qp6_index<K>(5u, (uint32_t)i, L)] = 0;
      }
      if (band_overflow) flag_error(a.err, 1);
      w.sync();
    }

    // ---- 1. row m from the saved {H, E} ----
    const int32_t hextm = hfree ? 0 : ge;
    const uint32_t limit = m + n;
    // row m as the score pass stored it: {H, E} int32 per column, or one {H, E - (go+ge)} int16 pair (the 16-bit query-profile sweep)
    const bool lr16 = MODE == MODE_QP && a.ckpt_narrow != 0;
    const int32_t goe_b = go + ge;
    while (col > 0 && k <= limit) {
      if (state == 0) {
        if (lastrow_h(lr, col, lr16) == lastrow_e(lr, col, lr16, goe_b)) state = 1;  // bit3: H == E (gotoh.h:135)
        else break;
      }
      const bool valid = L < col;
      const uint32_t cl = col - (valid ? L : 0);
      const bool bit1 = valid && (cl == 1 ? true : (lastrow_e(lr, cl, lr16, goe_b) != lastrow_e(lr, cl - 1, lr16, goe_b) + hextm));  // gotoh.h:137
      const uint32_t first_hit = first_set(w.ballot(bit1));
      const uint32_t first_out = first_set(w.ballot(!valid));
      const bool found = first_hit < first_out;
      const uint32_t x = found ? first_hit + 1 : first_out;
      if (L < x) out[k + L] = 'h';
      k += x; col -= x;
      if (found) state = 0;
    }

    // ---- 2. bands ----
    uint32_t swept_steps = 0;
    while (row > 0 && col > 0 && k <= limit) {
      const uint32_t t_cur = col + cell_addr(row + pad, K).lane;
      const uint32_t j = (t_cur - 1) / B;
      const uint32_t t0 = j * B;
      TraceLane<K> ts;
      int32_t bot_h, bot_f, prev_up_h;
      const int32_t hbias = (MODE == MODE_QP) ? 0 : go + ge, ebias = (MODE == MODE_QP) ? go + ge : 0;
      const bool qp16 = MODE == MODE_QP && a.ckpt_narrow != 0;
      auto qp16_index = [&](uint32_t jj, uint32_t field) -> uint64_t { return ((uint64_t)(jj - 1) * ckpt_fields_qp16(K) + field) * 64u + L; };
#pragma unroll
      for (int i = 0; i < K; ++i) {
        const uint32_t r = L * K + i + 1 - pad;
        const bool padding = (L * K + i < pad);
        const bool hz = hfree && (r == m);
        ts.cx1[i] = trace_cx1(hz ? 0 : go + ge);
        ts.cx2[i] = trace_cx2((hz || padding) ? 0 : ge);
        if (j == 0) {
          ts.Hc[i] = padding ? 0 : (int32_t)((uint32_t)edge_value(vfree, go, ge, (int32_t)r) << SH);
          ts.Ec[i] = padding ? 0 : neg;
        } else {
          // frontier fields of the 16-bit sweeps: string kernel {H + goe, E} as int32 fields (gotoh_body); query-profile kernel
          // one packed {H, E - goe} dword per slot (gotoh_narrow_qp_body)
          int32_t hv, ev;
          if (qp16) { const int32_t pk = ck[qp16_index(j, (uint32_t)i)]; hv = pk; ev = pk >> 16; }
          else { hv = ck[ckpt_index(j, (uint32_t)i, L, K)]; ev = ck[ckpt_index(j, (uint32_t)(K + i), L, K)]; }
          ts.Hc[i] = (int32_t)((uint32_t)(a.ckpt_narrow ? sext16(hv - hbias) : hv) << SH);
          ts.Ec[i] = (int32_t)((uint32_t)(a.ckpt_narrow ? sext16(ev) + ebias : ev) << SH);
        }
      }
      if (j == 0) {
        const uint32_t row_above = (L * K > pad) ? L * K - pad : 0u;
        prev_up_h = (row_above == 0) ? 0 : (int32_t)((uint32_t)edge_value(vfree, go, ge, (int32_t)row_above) << SH);
        bot_h = 0; bot_f = 0;
      } else if (qp16) {
        const int32_t pk = ck[qp16_index(j, (uint32_t)K)];
        bot_h = ts.Hc[K - 1];  // the strip's last H is what it hands down
        bot_f = (int32_t)((uint32_t)((pk >> 16) + ebias) << SH);
        prev_up_h = (int32_t)((uint32_t)sext16(pk) << SH);
      } else {
        const int32_t bh = ck[ckpt_index(j, 2u * K, L, K)], bf = ck[ckpt_index(j, 2u * K + 1, L, K)];
        const int32_t pu = ck[ckpt_index(j, 2u * K + 2, L, K)];
        bot_h = (int32_t)((uint32_t)(a.ckpt_narrow ? sext16(bh - hbias) : bh) << SH);
        bot_f = (int32_t)((uint32_t)(a.ckpt_narrow ? sext16(bf) + ebias : bf) << SH);
        prev_up_h = (int32_t)((uint32_t)(a.ckpt_narrow ? sext16(pu - hbias) : pu) << SH);
      }
      auto band_step = [&](uint32_t t, const auto& sub) {
        const int32_t c = (int32_t)t - (int32_t)L;
        int32_t up_h = w.shift_up(bot_h);
        int32_t up_f = w.shift_up(bot_f);
        uint32_t w0 = 0, w1 = 0;
        if ((c >= 1) && (c <= (int32_t)n)) {
          if (L == 0) {
            up_h = (int32_t)((uint32_t)edge_value(hfree, go, ge, c) << SH);
            up_f = neg;
          }
          const bool vz = vfree && (c == (int32_t)n);
          const int32_t vopen = vz ? 0 : go + ge, vext = vz ? 0 : ge;
          int32_t nb_h, nb_f;
          trace_step<K>(ts, up_h, up_f, prev_up_h, trace_cy1(vopen), trace_cy2(vext), sub, w0, w1, nb_h, nb_f);
          prev_up_h = up_h;
          bot_h = nb_h;
          bot_f = nb_f;
        }
        if (L < lanes_used) band[(uint64_t)(t - t0 - 1u) * 64u + L] = ((uint64_t)w1 << 32) | w0;
      };
      if (MODE == MODE_CHAR) {
        for (uint32_t t = t0 + 1; t <= t_cur; ++t) {
          const int32_t c = (int32_t)t - (int32_t)L;
          if ((c >= 1) && (c <= (int32_t)n)) {
            sub_c.cc = (int32_t)a2c[a2_index(d, (uint32_t)c)];
            if (rcflag) sub_c.cc = (int32_t)complement_char((uint8_t)sub_c.cc);
          }
          band_step(t, sub_c);
        }
      } else {
        // same software pipeline as the sweep of gotoh_body: the code two steps ahead (unclamped read of the padded
        // code buffer), the strip one step ahead, row by row, into the other half of a ping-pong pair
        SubRows<K> qa, qb;
        const int16_t* lane_col = qp_tab + L;
        const uint8_t* a2v = a2c - kCodeBias;
        const int32_t lane_base = (int32_t)kCodeBias + (rcflag ? (int32_t)n + (int32_t)L : -(int32_t)L - 1);
        const int32_t dir = rcflag ? -1 : 1;
        auto raw_at = [&](uint32_t tt) -> uint32_t { return a2v[(uint32_t)(lane_base + dir * (int32_t)tt)]; };
        uint32_t raw_next = raw_at(t0 + 2);
        qp_fetch_rows<K>(lane_col, raw_at(t0 + 1), qa);
        for (uint32_t t = t0 + 1; t <= t_cur; t += 2) {
          {
            const uint32_t raw_nn = raw_at(t + 2);
            qp_fetch_rows<K>(lane_col, raw_next, qb);
            band_step(t, qa);
            raw_next = raw_nn;
          }
          if (t + 1 > t_cur) break;
          {
            const uint32_t raw_nn = raw_at(t + 3);
            qp_fetch_rows<K>(lane_col, raw_next, qa);
            band_step(t + 1, qb);
            raw_next = raw_nn;
          }
        }
      }
      swept_steps += t_cur - t0;
      w.sync_global();
      BandFetch fetch{band, t0, K, pad};
      walk_core(w, fetch, row, col, state, k, out, t0, K, limit, pad);
      w.sync_global();
    }
    if (L == 0 && a.swept && swept_steps) {  // cells of the strip x steps re-swept (what the band timer reports as evaluated cells)
#if defined(__HIP_DEVICE_COMPILE__)
      atomicAdd(a.swept, (unsigned long long)swept_steps * (unsigned long long)(lanes_used * K));
#else
      *a.swept += (unsigned long long)swept_steps * (unsigned long long)(lanes_used * K);
#endif
    }
  }
  ok = walk_tails(w, row, col, state, k, out);
  if (!ok || row > 0 || col > 0) {
    if (L == 0) flag_error(wa.err, 2);
  }
  if (L == 0) wa.ops_len[d.out] = k;
}

// ------------------------------------------------------------------------------------------------
// Needleman-Wunsch with linear gaps (needle.h:12-138), same wave decomposition.  Profiles are scored
// in double with a float accumulator (needle.h:26 makes TProfile double; align.h:112-116).
// ------------------------------------------------------------------------------------------------
template <int K>
struct SubProfD {
  const float* p1;  // LDS [5][64*K]
  uint32_t row0;
  double b[5];
  double dmatch, dmis;
  int shift;
  TR_HD int32_t operator()(int i) const {
    float acc = 0.0f;
#pragma unroll
    for (int k1 = 0; k1 < 5; ++k1) {
      const double x = (double)p1[k1 * (64 * K) + row0 + i];
#pragma unroll
      for (int k2 = 0; k2 < 5; ++k2) {
        const double wgt = (k1 == k2) ? dmatch : dmis;
#if defined(__HIP_DEVICE_COMPILE__)
        acc = (float)__dadd_rn((double)acc, __dmul_rn(__dmul_rn(x, b[k2]), wgt));
#else
        acc = (float)((double)acc + (x * b[k2]) * wgt);
#endif
      }
    }
    return (int32_t)((uint32_t)(int32_t)acc << shift);
  }
};

template <class W, int K, int MODE, bool TRACE>
TR_HD void needle_body(W& w, const DpArgs& a, uint32_t pair_idx) {
  static_assert(MODE == MODE_CHAR || MODE == MODE_PROF, "NW supports string and profile inputs");
  const PairDesc d = a.pairs[pair_idx];
  const uint32_t L = w.lane();
  const uint32_t m = d.m, n = d.n;
  const int32_t ge = a.ge;
  const bool hfree = a.hfree != 0, vfree = a.vfree != 0;
  constexpr int SH = TRACE ? 2 : 0;

  if (m == 0 || n == 0) {  // needle.h:36-45
    if (L == 0 && a.scores)
      a.scores[d.out] = (m == 0) ? (n == 0 ? 0 : edge_value(hfree, 0, ge, (int32_t)n)) : edge_value(vfree, 0, ge, (int32_t)m);
    return;
  }
  const uint8_t* a1c = static_cast<const uint8_t*>(a.a1) + (MODE == MODE_CHAR ? d.a1_off : 0);
  const float* a1p = static_cast<const float*>(a.a1) + (MODE != MODE_CHAR ? d.a1_off : 0);
  const uint8_t* a2c = static_cast<const uint8_t*>(a.a2) + (MODE != MODE_PROF ? d.a2_off : 0);
  const float* a2p = static_cast<const float*>(a.a2) + (MODE == MODE_PROF ? d.a2_off : 0);
  float* p1_tab = reinterpret_cast<float*>(w.lds());
  if (MODE == MODE_PROF) {
    float ma = 0.0f, mb = 0.0f;
    for (uint32_t r = L; r < m; r += 64) { const float s = column_mass(a1p, d.a1_stride, r); ma = mass_max(ma, s); }
    for (uint32_t c = L; c < n; c += 64) { const float s = column_mass(a2p, d.a2_stride, c); mb = mass_max(mb, s); }
    report_mass(a.err, 2, ma);
    report_mass(a.err, 3, mb);
  }

  const uint32_t P = num_passes(m, K);
  uint32_t* bits = TRACE ? a.bits32 + d.bits_off : nullptr;
  int32_t* scratch = a.scratch ? a.scratch + 2 * d.scratch_off : nullptr;

  for (uint32_t p = 0; p < P; ++p) {
    const uint32_t base = p * 64u * K;
    const uint32_t rows_here = (m - base < 64u * K) ? m - base : 64u * K;
    const uint32_t lanes_used = (rows_here + K - 1) / K;
    const uint32_t t_end = n + lanes_used - 1;
    const bool last_pass = (p + 1 == P);

    NeedleLane<K> ns;
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const uint32_t r = base + L * K + i + 1;
      const bool hz = hfree && (r == m);
      ns.Sc[i] = (int32_t)((uint32_t)edge_value(vfree, 0, ge, (int32_t)r) << SH);
      ns.hx[i] = TRACE ? (hz ? 0 : ge) * 4 + 2 : (hz ? 0 : ge);
    }
    const uint32_t row_above = base + L * K;
    int32_t prev_up = (row_above == 0) ? 0 : (int32_t)((uint32_t)edge_value(vfree, 0, ge, (int32_t)row_above) << SH);
    int32_t bot = 0;

    SubChar<K> sub_c;
    SubProfD<K> sub_p;
    if (MODE == MODE_CHAR) {
#pragma unroll
      for (int i = 0; i < K; ++i) {
        const uint32_t r = base + L * K + i + 1;
        sub_c.rc[i] = (r <= m) ? (int32_t)a1c[r - 1] : -1;
      }
      sub_c.vmatch = (int32_t)((uint32_t)a.match << SH);
      sub_c.vmis = (int32_t)((uint32_t)a.mismatch << SH);
      sub_c.cc = 0;
    } else {
      w.sync();
#pragma unroll
      for (int i = 0; i < K; ++i) {
        const uint32_t r = base + L * K + i + 1;
#pragma unroll
        for (int k = 0; k < 5; ++k)
          p1_tab[k * (64 * K) + L * K + i] = (r <= m) ? a1p[(uint64_t)k * d.a1_stride + (r - 1)] : 0.0f;
      }
      w.sync();
      sub_p.p1 = p1_tab;
      sub_p.row0 = L * K;
      sub_p.dmatch = (double)a.match;
      sub_p.dmis = (double)a.mismatch;
      sub_p.shift = SH;
    }

    for (uint32_t t = 1; t <= t_end; ++t) {
      const int32_t c = (int32_t)t - (int32_t)L;
      int32_t up = w.shift_up(bot);
      const bool active = (c >= 1) && (c <= (int32_t)n);
      if (active) {
        if (L == 0) up = (p == 0) ? (int32_t)((uint32_t)edge_value(hfree, 0, ge, c) << SH) : scratch[2 * c];
        const bool vz = vfree && (c == (int32_t)n);
        const int32_t vy = TRACE ? (vz ? 0 : ge) * 4 + 1 : (vz ? 0 : ge);
        const uint32_t ci = a2_index(d, (uint32_t)c);
        uint32_t w0 = 0;
        int32_t nb;
        if (MODE == MODE_CHAR) {
          sub_c.cc = (int32_t)a2c[ci];
          needle_step<K, TRACE>(ns, up, prev_up, vy, sub_c, w0, nb);
        } else {
#pragma unroll
          for (int k = 0; k < 5; ++k) sub_p.b[k] = (double)a2p[(uint64_t)k * d.a2_stride + ci];
          needle_step<K, TRACE>(ns, up, prev_up, vy, sub_p, w0, nb);
        }
        prev_up = up;
        bot = nb;
        if (TRACE && L < lanes_used) bits[word_index(p, t, L, n)] = w0;
        if (!last_pass && L == 63) scratch[2 * c] = bot;
      }
    }
    if (last_pass && a.scores) {
      const uint32_t g = m - 1 - base;
      if (L == g / K) {
        int32_t v = 0;
#pragma unroll
        for (int i = 0; i < K; ++i)
          if ((uint32_t)i == g % K) v = ns.Sc[i] >> SH;
        a.scores[d.out] = v;
      }
    }
    if (!last_pass) w.sync_global();
  }
}

// needle traceback, needle.h:113-131
TR_HD void needle_walk_one(const WalkArgs& a, const uint32_t* bits32, uint32_t pair_idx) {
  const PairDesc d = a.pairs[pair_idx];
  const uint32_t* bits = bits32 + d.bits_off;
  uint8_t* out = a.ops + a.ops_off[d.out];
  const uint32_t n = d.n;
  uint32_t row = d.m, col = d.n, k = 0;
  while (row > 0 || col > 0) {
    bool b3, b4;
    if (row == 0) { b3 = true; b4 = false; }
    else if (col == 0) { b3 = false; b4 = true; }
    else {
      const CellAddr ca = cell_addr(row, a.K);
      const uint32_t wd = bits[word_index(ca.pass, col + ca.lane, ca.lane, n)];
      const uint32_t two = (wd >> (2u * ca.slot)) & 3u;
      b3 = (two & 2u) != 0;
      b4 = (two & 1u) != 0;
    }
    if (b3) { --col; out[k++] = 'h'; }
    else if (b4) { --row; out[k++] = 'v'; }
    else { --row; --col; out[k++] = 's'; }
  }
  a.ops_len[d.out] = k;
}

}  // namespace tracyhip
#endif
