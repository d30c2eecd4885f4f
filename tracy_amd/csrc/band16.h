// band16.h -- Gotoh on a diagonal band, four pairs per wave, sixteen lanes per pair.
//
// The row-strip wavefront of dp_kernels.h (lane L owns K rows, the wave walks the anti-diagonals) keeps 64 lanes busy only
// when a strip has columns to sweep for as long as the 64 strips below it need to start: on a band of `width` diagonals a
// strip is active for K + width steps out of the 64 (K + 1) it takes to get all lanes going.  Here a pair is swept by the 16
// lanes of one DPP row instead: strip s (rows sK+1 .. sK+K) belongs to lane s mod 16 and starts at step s (K + 1), one
// step per column behind the strip above it -- whose last row arrives through a row rotate (row_ror:1), lane 15 handing to
// lane 0 -- and sweeps its own window of the band, the columns sK+1+dmin .. sK+K+dmax, in S = K + dmax - dmin steps.  With
// S <= 15 (K + 1) a lane is done with strip s before strip s + 16 is due, the lane above has gone quiet (-inf) before a
// strip runs past the end of its window, and four pairs fill the wave.  A 1 kb x 1 kb pair on a band of 100 diagonals costs
// ~1100 steps of a quarter wave with K = 4 instead of ~1060 steps of a whole wave with K = 15.
//
// Cells outside the band read as -inf; the caller proves that the band holds every optimal path (a-priori from a known
// score, or a-posteriori from the banded score: pipeline.hip), which makes scores, trace bits and path those of the whole
// matrix.  Two kinds share the sweep: KIND 0 stores the trace nibbles of the band -- K/2 bytes per strip
// and step, (NS-1) S + S_last words per pair, nothing outside the band -- and walks them (gotoh.h:143-167) with the 16 lanes
// of the pair; KIND 1 is the origin-tracking sweep (dp_lane.h origin_step): score and the two ends trimReferenceSlice reads.
//
// Substitution scores come from a table in global memory, int16 [6 codes][rows] per sequence, written once per stage by
// b16_table_kernel (band16.hip) -- the query profile of align.h:103-118 against one-hot columns for profile rows, byte
// equality (align.h:96-101) for strings; a lane copies the K rows of its strip into its own column of the LDS table
// [code][row][lane] when it starts the strip (the rows of the strip after that are already on their way).  The reference
// codes of the four pairs are staged in LDS once.
//
// Domain: AlignConfig<hfree, false> (no free vertical end gaps), go <= 0, ge < 0, one of K = 4 / 8 / 12, S <= 15 (K + 1).
#ifndef TRACY_AMD_BAND16_H
#define TRACY_AMD_BAND16_H

#include <type_traits>

#include "dp_kernels.h"

namespace tracyhip {

struct Band16Args {
  const PairDesc* pairs;  // a1_off / a1_stride: first row and code-row stride of the pair's rows in `qp` (int16 units); a2_off: codes;
                          // bits_off: BYTE offset of the pair's words (KIND 0); ckpt_off: band_pack(dmin, dmax); flags: PAIR_A2_REVCOMP;
                          // lastrow_off (KIND 0): 'h' columns to put before (low half) / behind (high half) the pair's string
  uint32_t npairs;
  const int16_t* qp;      // substitution tables (b16_table_kernel): entries are scores << kTagShift (the origin-tracking sweep shifts them on)
  const uint8_t* codes;   // reference codes 0..5 (ctx->codes())
  uint8_t* bits;          // KIND 0: trace words
  int32_t* scores;        // H(m, n) per pair (PairDesc::out), or null
  uint32_t* ends;         // KIND 1: {lead, c_e} per pair
  int32_t* err;           // DpArgs::err
  int32_t go, ge, hfree;
  uint32_t code_cap;      // LDS bytes reserved per pair for its reference codes (>= the longest n of the launch, multiple of 4)
  uint8_t* ops;           // KIND 0: walker output (push order), pair i at ops + ops_off[out]
  const uint64_t* ops_off;
  uint32_t* ops_len;
  // CONT (KIND 1): the sweep continues below row R of a prefix sweep over all columns (gotoh_prefix_body, PAIR_KEEP_ROW) instead of
  // starting at row 0: `row` holds one dword per column -- low half H(R, c) + (go + ge), high half F(R, c) -- the pair's column c at
  // row[lastrow_off + c]; PairDesc::bits_off is R (the rows above the pair's first one), m the rows below it.  Origins are not
  // tracked (ends[2 out] = 0); score and c_e are those of the band under row R.
  const uint32_t* row;
  // lists laid out on the device (stream.hip): pair i of the launch is pairs[index[i]] (null: pairs[i]), and the number of pairs is
  // *count (null: npairs) -- the grid is sized for the worst case and the waves past the count leave at once
  const uint32_t* index;
  const uint32_t* count;
};
constexpr uint32_t kB16RowCap = 200;  // CONT: row-R entries staged per pair (the window of strip 0 and the column before it)
constexpr uint32_t kB16QuadRowCap = 24;  // ... of the quad form: a window of at most fifteen steps

constexpr uint32_t kB16Codes = 6;
TR_HD constexpr uint32_t b16_period(int K) { return 16u * ((uint32_t)K + 1u); }
TR_HD constexpr uint32_t b16_max_window(int K) { return 15u * ((uint32_t)K + 1u); }
TR_HD constexpr uint32_t b16_word_bytes(int K) { return K <= 4 ? 2u : K <= 8 ? 4u : 8u; }
TR_HD constexpr uint32_t b16_table_bytes(int K) { return kB16Codes * (uint32_t)K * 64u * 2u; }
TR_HD uint32_t b16_strips(uint32_t m, int K) { return (m + (uint32_t)K - 1u) / (uint32_t)K; }
// steps of a strip's window: its K + dmax - dmin columns rounded up to whole blocks of K + 1 steps (band16_body: a lane's activity
// changes at block boundaries only; the extra columns lie outside the band and hold lower bounds like every cell there)
TR_HD uint32_t b16_window(int K, int32_t dmin, int32_t dmax) {
  const uint32_t s = (uint32_t)K + (uint32_t)(dmax - dmin), kp = (uint32_t)K + 1u;
  return (s + kp - 1u) / kp * kp;
}
// first column of strip s's window (may be <= 0: the lane waits for column 1)
TR_HD int32_t b16_first_col(uint32_t s, int K, int32_t dmin) { return (int32_t)(s * (uint32_t)K) + 1 + dmin; }
// the last strip sweeps on to column n: the trailing run of row m (free end gap) lies outside the band's diagonals
TR_HD uint32_t b16_last_window(uint32_t m, uint32_t n, int K, int32_t dmin, int32_t dmax) {
  const uint32_t S = b16_window(K, dmin, dmax);
  const int32_t ext = (int32_t)n - b16_first_col(b16_strips(m, K) - 1u, K, dmin) + 1;
  return (ext > 0 && (uint32_t)ext > S) ? (uint32_t)ext : S;
}
// words a traceback sweep evaluates: the windows of the strips (cells = words x K; the algorithmic bytes of the launch)
TR_HD uint64_t b16_words(uint32_t m, uint32_t n, int K, int32_t dmin, int32_t dmax) {
  return (uint64_t)(b16_strips(m, K) - 1u) * b16_window(K, dmin, dmax) + b16_last_window(m, n, K, dmin, dmax);
}
// ... and where they are kept (KIND 0).  In block b of the sweep the strips b, b - 1, .., b - NB + 1 are live (NB = blocks of a window) and
// each writes the K + 1 words of one block of its window: the word of strip s, window block bw, step k lies at ((s + bw) NB + bw) (K + 1) + k
// -- the lanes of a pair write ONE run of NB (K + 1) words per block and the pair's words grow like a stream, a line is complete a block
// or two after it was begun.  (Strip by strip -- word (s, u) at s S + u -- a pair had as many lines open as it has live strips, each
// filled ten bytes a block: 100 000 wide bands held more half-written lines than the L2s hold, and rocprofv3's WRITE_SIZE was twice the
// bytes written.)  Slots of strips that do not exist (s < 0, s >= NS) stay unwritten.  The last strip runs on to column n: its blocks
// behind the NB-th follow the rest, one after the other.
struct B16Layout {
  uint32_t NS, NB, NB_last, KP;
  uint64_t tail;  // first word of the last strip's blocks NB, NB + 1, ..
  TR_HD uint64_t words() const { return tail + (uint64_t)(NB_last > NB ? NB_last - NB : 0u) * KP; }
  TR_HD uint64_t block_word(uint32_t s, uint32_t bw) const {  // first word of block bw of strip s's window
    return bw < NB ? (uint64_t)(((s + bw) * NB + bw) * KP) : tail + (uint64_t)((bw - NB) * KP);  // (32-bit products: m + n < 2^24 rows / columns)
  }
};
TR_HD B16Layout b16_layout(uint32_t m, uint32_t n, int K, int32_t dmin, int32_t dmax) {
  B16Layout l;
  l.KP = (uint32_t)K + 1u;
  l.NS = b16_strips(m, K);
  l.NB = b16_window(K, dmin, dmax) / l.KP;
  l.NB_last = (b16_last_window(m, n, K, dmin, dmax) + (uint32_t)K) / l.KP;
  if (l.NB_last < l.NB) l.NB_last = l.NB;
  l.tail = (uint64_t)(l.NS ? l.NS - 1u + l.NB : 0u) * l.NB * l.KP;
  return l;
}
TR_HD uint64_t b16_store_words(uint32_t m, uint32_t n, int K, int32_t dmin, int32_t dmax) { return b16_layout(m, n, K, dmin, dmax).words(); }
// the quad form (band16_body P = 4, strip height 4): the window fits the three blocks a lane has before its next strip is due
TR_HD bool b16_narrow_ok(int32_t dmin, int32_t dmax) { return dmax >= dmin && b16_window(4, dmin, dmax) <= 3u * 5u; }
TR_HD constexpr uint32_t b16_packed_row(uint32_t code_cap) { return ((code_cap + 1u) / 2u + 3u) & ~3u; }  // the quad form keeps two codes to a byte
TR_HD constexpr uint32_t b16_quad_lds(uint32_t code_cap) { return 16u * b16_packed_row(code_cap) + b16_table_bytes(4); }  // LDS of a quad-form workgroup
TR_HD constexpr uint32_t b16_cont_quad_lds(uint32_t code_cap) { return b16_quad_lds(code_cap) + 16u * kB16QuadRowCap * 4u; }
// LDS of a sixteen-lane workgroup of band16_cont16_body: the codes of its four sub-windows two to a byte, the table, the kept row's
// entries of strip 0 as the dwords they are kept in (16 KB -> 11 KB at 900 columns: ten -> fourteen workgroups on a CU)
TR_HD constexpr uint32_t b16_cont16_lds(int K, uint32_t code_cap) { return 4u * b16_packed_row(code_cap) + b16_table_bytes(K) + 4u * kB16RowCap * 4u; }
// smallest strip height whose lanes are done with a strip before the next one is due (K + width <= 15 (K + 1)); 0: the band is too wide
TR_HD int b16_pick_k(int32_t dmin, int32_t dmax) {
  if (dmax < dmin) return 0;
  if (b16_window(4, dmin, dmax) <= b16_max_window(4)) return 4;
  if (b16_window(8, dmin, dmax) <= b16_max_window(8)) return 8;
  if (b16_window(12, dmin, dmax) <= b16_max_window(12)) return 12;
  return 0;
}
// value ranges of the origin-tracking sweep on the band kernels (packed 14-bit score field, 13-bit origin) for m rows / n columns,
// AlignConfig<true,false> (capi_internal.h origin16_ok: the same test with a tracyhip_params)
TR_HD bool b16_origin_ok(int32_t match, int32_t mismatch, int32_t go, int32_t ge, uint32_t maxm, uint32_t maxn) {
  if (go > 0 || ge >= 0) return false;
  if ((uint64_t)maxn + 64 >= (1u << kOriginBits)) return false;
  auto ab = [](int32_t x) { return x < 0 ? -(int64_t)x : (int64_t)x; };
  const int64_t rows = maxm, q = ab(match) > ab(mismatch) ? ab(match) : ab(mismatch);
  const int64_t low = ab(go) + rows * ab(ge) + 2 * (ab(go) + ab(ge)) + ab(mismatch) + ab(match);
  const int64_t high = rows * q;
  return (low < -(int64_t)kNegInfOrigin - 16 * ab(ge) - 64) && (-(int64_t)kNegInfOrigin + ab(go) + 16 * ab(ge) < 8000) && (high < 8000);
}
// rows of a sequence's table: whole strips for every K, so that a strip's load never runs off the code row
TR_HD uint32_t b16_table_stride(uint32_t m) { return ((m + 15u) & ~15u) + 16u; }

// One row of a sequence's substitution table (b16_table_kernel; the host emulator builds its tables with the same function).
// Profile rows: the int of the fp32 chain of align.h:112-117 against the one-hot column of base b (onehot_score) for b = A C G T N,
// 0 for '-' / any other letter (an all-zero column); string rows: match / mismatch by byte equality (align.h:96-101), mismatch
// for a column no row letter can equal.
TR_HD void b16_table_row(const void* a1, bool strings, uint64_t a1_off, uint32_t a1_stride, uint32_t r, int32_t match, int32_t mismatch,
                         int32_t q[kB16Codes]) {
  if (strings) {
    const uint8_t ch = static_cast<const uint8_t*>(a1)[a1_off + r];
    for (uint32_t b = 0; b < 5; ++b) q[b] = ch == (uint8_t)"ACGTN"[b] ? match : mismatch;
    q[5] = mismatch;
  } else {
    float pr[5];
    for (int k = 0; k < 5; ++k) pr[k] = static_cast<const float*>(a1)[a1_off + (uint64_t)k * a1_stride + r];
    for (uint32_t b = 0; b < 5; ++b) q[b] = onehot_score(pr, b, (float)match, (float)mismatch);
    q[5] = 0;
  }
}

template <int P = 16>
TR_HD uint32_t ctz16(uint32_t x) {  // P when none of the low P bits is set
  uint32_t i = 0;
  x |= 1u << P;
  while (!((x >> i) & 1u)) ++i;
  return i;
}

template <int K>
struct Band16Fetch {
  const uint8_t* bits;
  uint32_t S, S_last, NS, n;
  int32_t dmin;
  B16Layout lay;
  TR_HD bool inside(uint32_t r, uint32_t c) const {
    const uint32_t s = (r - 1u) / (uint32_t)K;
    const int32_t u = (int32_t)c - b16_first_col(s, K, dmin);
    return c >= 1u && c <= n && u >= 0 && (uint32_t)u < (s + 1u == NS ? S_last : S);
  }
  TR_HD uint32_t operator()(uint32_t r, uint32_t c) const {
    constexpr uint32_t KP = (uint32_t)K + 1u;
    const uint32_t s = (r - 1u) / (uint32_t)K, slot = (r - 1u) % (uint32_t)K;
    const uint32_t u = (uint32_t)((int32_t)c - b16_first_col(s, K, dmin));
    const uint64_t idx = lay.block_word(s, u / KP) + u % KP;
    uint64_t wd;
    if (K <= 4) wd = reinterpret_cast<const uint16_t*>(bits)[idx];
    else if (K <= 8) wd = reinterpret_cast<const uint32_t*>(bits)[idx];
    else wd = reinterpret_cast<const uint64_t*>(bits)[idx];
    return (uint32_t)(wd >> (4u * slot)) & 15u;
  }
};

// The traceback state machine of gotoh.h:143-167 walked by the P lanes of each pair (walk_core of dp_kernels.h, a row of P
// candidates per round; the 64 / P groups of the wave run side by side).  A cell outside the stored band ends the walk with
// an error flag: the caller's certificate has failed for that pair and the pair is repeated on the whole matrix.
// C: cells per lane and round.  A round costs a memory round trip whatever it looks at; the quad form's four lanes take four cells of
// the run each (four loads in flight), so that a round advances by up to sixteen cells there as well.
template <class W, class Fetch, int P = 16, int C = 1>
TR_HD void walk16(W& w, const Fetch& fetch, bool have, uint32_t m, uint32_t n, uint8_t* out, uint32_t* ops_len, int32_t* err,
                   uint32_t pre_h = 0, uint32_t post_h = 0) {
  // pre_h / post_h: the pair is a sub-window of a wider reference whose columns right / left of it are free end-gap columns of the
  // alignment ('h' before the first / after the last op of the sub-window's string: PairDesc::lastrow_off)
  constexpr uint32_t PM = (1u << P) - 1u;
  const uint32_t lane = w.lane() % (uint32_t)P, gsh = (w.lane() / (uint32_t)P) * (uint32_t)P;
  uint32_t row = m, col = n, k = pre_h;
  int state = 0;
  const uint32_t limit = m + n + pre_h;
  bool lost = false;
  if (have)
    for (uint32_t i = lane; i < pre_h; i += P) out[i] = 'h';
  for (;;) {
    const bool running = have && !lost && row > 0 && col > 0 && k <= limit;
    if (w.ballot(running) == 0) break;
    // this lane's C cells of the run: offsets lane C .. lane C + C - 1 from the cell the walk stands on
    uint32_t my_hit = C, my_out = C, my_state = 2;
#pragma unroll
    for (int q = C - 1; q >= 0; --q) {
      const uint32_t off = lane * (uint32_t)C + (uint32_t)q;
      const uint32_t r = (state == 1) ? row : row - off;
      const uint32_t c = (state == 2) ? col : col - off;
      const bool inside = running && ((state == 1) ? (off < col) : (state == 2) ? (off < row) : (off < row && off < col));
      const bool inband = inside && fetch.inside(r, c);
      TraceBits b = {false, false, false, false};
      if (inband) b = decode_nibble(fetch(r, c));
      const bool hit = inband && (state == 0 ? (b.bit3 || b.bit4) : state == 1 ? b.bit1 : b.bit2);
      if (hit) { my_hit = (uint32_t)q; my_state = b.bit3 ? 1u : 2u; }
      if (!inband) my_out = (uint32_t)q;
    }
    const uint32_t lane_hit = ctz16<P>((uint32_t)(w.ballot(my_hit < (uint32_t)C) >> gsh) & PM);   // first lane with a hit / a cell outside
    const uint32_t lane_out = ctz16<P>((uint32_t)(w.ballot(my_out < (uint32_t)C) >> gsh) & PM);
    const uint32_t src_hit = gsh + (lane_hit & ((uint32_t)P - 1u)), src_out = gsh + (lane_out & ((uint32_t)P - 1u));
    uint32_t first_hit, first_out, to_state;
    if (C == 1) {
      first_hit = lane_hit;
      first_out = lane_out;
      to_state = w.bcast(my_state, src_hit);
    } else {
      const uint32_t packed = w.bcast(my_hit | (my_state << 8), src_hit);
      first_hit = lane_hit < (uint32_t)P ? lane_hit * (uint32_t)C + (packed & 0xffu) : (uint32_t)(P * C);
      to_state = packed >> 8;
      const uint32_t o = w.bcast(my_out, src_out);
      first_out = lane_out < (uint32_t)P ? lane_out * (uint32_t)C + o : (uint32_t)(P * C);
    }
    if (!running) continue;
    if (first_out == 0) { lost = true; continue; }  // the cell the walk stands on is not in the band
    uint32_t x;
    char op;
    if (state == 0) {
      x = first_hit < first_out ? first_hit : first_out;
      op = 's';
    } else {
      x = first_hit < first_out ? first_hit + 1 : first_out;
      op = state == 1 ? 'h' : 'v';
    }
#pragma unroll
    for (uint32_t q = 0; q < (uint32_t)C; ++q)
      if (lane * (uint32_t)C + q < x) out[k + lane * (uint32_t)C + q] = (uint8_t)op;
    k += x;
    if (state == 0) {
      row -= x; col -= x;
      if (first_hit < first_out) state = (int)to_state;
    } else {
      if (state == 1) col -= x; else row -= x;
      if (first_hit < first_out) state = 0;
    }
  }
  bool ok = have && !lost;
  if (ok) {  // first row / first column (gotoh.h:112-123)
    if (row == 0) {
      if (state == 2 && col > 0) ok = false;
      else { for (uint32_t i = lane; i < col; i += P) out[k + i] = 'h'; k += col; col = 0; }
    } else if (col == 0) {
      if (state == 1) ok = false;
      else { for (uint32_t i = lane; i < row; i += P) out[k + i] = 'v'; k += row; row = 0; }
    }
    if (row > 0 || col > 0) ok = false;
    if (ok) { for (uint32_t i = lane; i < post_h; i += P) out[k + i] = 'h'; k += post_h; }
  }
  if (have && lane == 0) {
    if (!ok) flag_error(err, 2);
    *ops_len = ok ? k : 0u;
  }
}

// P: lanes per pair.  Sixteen (a DPP row) is the general form; four (a quad: 16 pairs per wave, the hand-over a quad rotate) sweeps
// the narrow bands -- windows of at most 3 (K + 1) steps, b16_narrow_ok -- for which twelve of sixteen lanes would idle: a pair costs
// its strips x (K + 1) steps of a quarter of the lanes instead.
template <class W, int K, int KIND, bool CONT = false, int P = 16>
TR_HD void band16_body(W& w, const Band16Args& a, uint32_t wave_idx) {
  static_assert(K == 4 || K == 8 || K == 12, "strip heights of the band kernels");
  static_assert(P == 16 || P == 4, "a DPP row or a quad per pair");
  static_assert(!CONT || KIND == 1, "only the score / ends sweep continues from a stored row");
  static_assert(!CONT || P == 16, "the sweep below a stored row stages it for four pairs");
  static_assert(b16_max_window(K) + 2u <= kB16RowCap, "row staging");
  constexpr uint32_t NPW = 64u / (uint32_t)P;  // pairs per wave
  using Lanes = std::integral_constant<int, P>;
  constexpr int SH = KIND == 0 ? kTagShift : kOriginShift;
  constexpr int TS = KIND == 0 ? 0 : kOriginBits;
  constexpr uint32_t KP = (uint32_t)K + 1u;  // steps of a block: the stagger between two strips
  constexpr uint32_t WB = b16_word_bytes(K);
  const uint32_t L = w.lane(), g = L / (uint32_t)P, j = L % (uint32_t)P;
  const uint32_t pair_idx = wave_idx * NPW + g;
  const uint32_t npairs = a.count ? *a.count : a.npairs;
  if (wave_idx * NPW >= npairs) return;  // (wave-uniform)
  bool have = pair_idx < npairs;
  PairDesc d{};
  if (have) d = a.pairs[a.index ? a.index[pair_idx] : pair_idx];
  if (d.flags & PAIR_SKIP) { have = false; d = PairDesc{}; }
  const uint32_t m = d.m, n = d.n;
  const int32_t dmin = band_dmin(d), dmax = band_dmax(d);
  const int32_t go = a.go, ge = a.ge, goe = go + ge;
  const bool hfree = a.hfree != 0;
  const bool rcflag = (d.flags & PAIR_A2_REVCOMP) != 0;
  const uint32_t NS = have ? b16_strips(m, K) : 0u;
  const uint32_t S = b16_window(K, dmin, dmax);
  const uint32_t S_last = have ? b16_last_window(m, n, K, dmin, dmax) : 0u;
  // The sweep runs in blocks of K + 1 steps: block b begins strip b (lane b mod 16 of every pair), and because a window is a whole
  // number of blocks (b16_window) a lane's activity changes at block boundaries only -- inside a block a step tests nothing but
  // "is my column one of 1 .. n".
  const uint32_t NB = S / KP, NB_last = (S_last + (uint32_t)K) / KP;
  const B16Layout lay = b16_layout(have ? m : 0u, n, K, dmin, dmax);  // KIND 0: where the trace words go
  const int32_t neg = KIND == 0 ? (int32_t)((uint32_t)kNegInf << SH) : (int32_t)((uint32_t)kNegInfOrigin << SH);
  const uint32_t rbase = CONT ? (uint32_t)d.bits_off : 0u;  // rows above the pair's first one
  auto edge = [&](uint32_t r) -> int32_t { return (int32_t)((uint32_t)edge_value(false, go, ge, (int32_t)(r + rbase)) << SH); };  // H(r, 0), r >= 1
  // H(0, c): the free (or paid) leading gap; KIND 1 carries the column itself as the origin
  auto row0 = [&](int32_t c) -> int32_t {
    if (c <= 0) return 0;
    return (int32_t)((uint32_t)edge_value(hfree, go, ge, c) << SH) + ((KIND == 1 && !CONT) ? c : 0);
  };

  // ---- LDS: the reference codes of the wave's pairs in view order (column c at byte c - 1), then the lanes' tables ----
  // (the quad form stages sixteen references: two codes to a byte, or the code rows alone would leave room for seven workgroups on a CU)
  constexpr bool PACKED = P == 4;
  const uint32_t code_row = PACKED ? b16_packed_row(a.code_cap) : a.code_cap;  // LDS bytes per pair
  uint8_t* lcodes = reinterpret_cast<uint8_t*>(w.lds()) + g * code_row;
  int16_t* tab = reinterpret_cast<int16_t*>(w.lds() + NPW * code_row) + (((L & 31u) << 1) | (L >> 5));  // (lane columns interleaved: gotoh_narrow_qp_body)
  if (have) {
    const uint8_t* src = a.codes + d.a2_off;
    if (!PACKED) {
      for (uint32_t i = j; i < n; i += P) lcodes[i] = src[rcflag ? n - 1u - i : i];
    } else {
      for (uint32_t i = j; 2u * i < n; i += P) {
        const uint32_t c0 = 2u * i, c1 = c0 + 1u < n ? c0 + 1u : c0;
        const uint32_t lo = src[rcflag ? n - 1u - c0 : c0], hi = src[rcflag ? n - 1u - c1 : c1];
        lcodes[i] = (uint8_t)(lo | (hi << 4));
      }
    }
  }
  // CONT: {H, F} of row R for the columns strip 0 sweeps and the one before them, in the sweep's own units
  int32_t* lrow = reinterpret_cast<int32_t*>(w.lds() + NPW * code_row + b16_table_bytes(K)) + g * (2u * kB16RowCap);
  const int32_t crow0 = dmin;  // column of lrow[0]: b16_first_col(0) - 1
  if (CONT && have) {
    const uint32_t* src = a.row + d.lastrow_off;
    for (uint32_t i = j; i < kB16RowCap; i += P) {
      const int32_t cc = crow0 + (int32_t)i;
      int32_t hh = neg, ff = neg;
      if (cc == 0) hh = edge(0);
      else if (cc >= 1 && cc <= (int32_t)n) {
        const uint32_t v = src[cc];
        hh = (int32_t)((uint32_t)(sext16((int32_t)v) - goe) << SH);
        ff = (int32_t)((uint32_t)((int32_t)v >> 16) << SH);
      }
      lrow[2u * i] = hh;
      lrow[2u * i + 1u] = ff;
    }
  }
  w.sync();
  // code of column cm1 + 1; lanes off the reference (cm1 wraps below column 1) read its last column and discard the result
  const uint32_t nclamp = n ? n - 1u : 0u;
  auto code_at = [&](uint32_t cm1) -> uint32_t {
    const uint32_t i = cm1 < nclamp ? cm1 : nclamp;
    if (!PACKED) return lcodes[i];
    return ((uint32_t)lcodes[i >> 1] >> ((i & 1u) << 2)) & 15u;
  };

  // ---- wave-uniform block counts ----
  uint32_t B_end = 0, b_last = ~0u;
  {
    const uint32_t mine = have ? (NS - 1u) + NB_last : 0u;
    const uint32_t lastbeg = have ? NS - 1u : ~0u;
    for (uint32_t q = 0; q < NPW; ++q) {
      const uint32_t x = w.bcast(mine, q * (uint32_t)P), y = w.bcast(lastbeg, q * (uint32_t)P);
      B_end = x > B_end ? x : B_end;
      b_last = y < b_last ? y : b_last;
    }
  }

  // ---- per-lane state ----
  TraceLane<K> ts;
#pragma unroll
  for (int i = 0; i < K; ++i) { ts.Hc[i] = neg; ts.Ec[i] = neg; ts.cx1[i] = trace_cx1<TS>(goe); ts.cx2[i] = trace_cx2<TS>(ge); }
  int32_t bot_h = neg, bot_f = neg, prev_up_h = neg;
  uint32_t s_cur = j;   // the strip the lane sweeps (or swept last)
  uint32_t cm1 = 0;     // its column - 1 (wraps while the window is still left of column 1)
  uint32_t left = 0;    // blocks its strip still takes
  bool live = false;
  uint32_t raw_next = 0;
  uint32_t c_end = 0;
  const int32_t cy1 = trace_cy1<TS>(goe), cy2 = trace_cy2<TS>(ge);
  uint8_t* const bits = a.bits + d.bits_off;
  uint8_t* wp = bits;   // KIND 0: where the word of the next step goes
  const uint32_t slot_m = have && m ? (m - 1u) % (uint32_t)K : 0u;

  // rows of strip s: K int16 per code, straight from the sequence's table
  constexpr int ND = K / 2;
  uint32_t pf[kB16Codes][ND];
  // The rows of strip b + P are requested when strip b begins and used P blocks later.  gfx950 counts loads and stores in ONE in-order
  // counter, and the compiler -- which cannot count memory operations across the block loop -- waits for vmcnt(0) where the rows are
  // used: behind the acknowledgement of the trace words stored a few instructions earlier, once per block (round 5: 29-34 % of the
  // traceback kernels' wave cycles in s_waitcnt).  The traceback sweeps on strips of 4 / 8 rows therefore issue the loads by hand and
  // wait for vmcnt(kPfBehind): every block between the request and the use stores its words (a pair with a strip b + P has a live
  // strip in each of them: >= 2 store instructions per block at K = 4 -- ten bytes --, >= 3 at K = 8 -- thirty-six), so the rows are
  // older than the (P - 1) x that many operations the wait leaves outstanding.  The first P blocks use the rows requested before the
  // loop, with no such guarantee: vmcnt(0).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(TRACY_B16_NO_HANDLOAD)
  constexpr bool HANDLOAD = KIND == 0 && K <= 8;
#else
  constexpr bool HANDLOAD = false;
#endif
  constexpr int kPfBehind = (P - 1) * (K <= 4 ? 2 : 3);
  static_assert(!HANDLOAD || kPfBehind <= 63, "vmcnt is a six-bit counter");
#if defined(__clang__)
  typedef uint32_t pf_vec __attribute__((ext_vector_type(K <= 4 ? 2 : 4)));
#else
  struct pf_vec { uint32_t v[4]; uint32_t operator[](int i) const { return v[i]; } };  // (host emulator build: never loaded by hand)
#endif
  pf_vec pv[kB16Codes];
  auto prefetch = [&](uint32_t s) {
    const int16_t* src = a.qp + d.a1_off + (uint64_t)s * K;
#if defined(__HIP_DEVICE_COMPILE__)
    if (HANDLOAD) {
#pragma unroll
      for (uint32_t b = 0; b < kB16Codes; ++b) {
        const int16_t* q = src + (uint64_t)b * d.a1_stride;  // (any alignment: a trimmed view may start on an odd row)
        if (K <= 4) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(pv[b]) : "v"(q));
        else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pv[b]) : "v"(q));
      }
      return;
    }
#endif
#pragma unroll
    for (uint32_t b = 0; b < kB16Codes; ++b) __builtin_memcpy(pf[b], src + (uint64_t)b * d.a1_stride, 2 * K);  // (any alignment: a trimmed view may start on an odd row)
  };
  // the requested rows have arrived: into pf (hand-issued loads only)
  auto arrived = [&](auto first) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (HANDLOAD) {
      if (decltype(first)::value)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(pv[0]), "+v"(pv[1]), "+v"(pv[2]), "+v"(pv[3]), "+v"(pv[4]), "+v"(pv[5]) : : "memory");
      else
        asm volatile("s_waitcnt vmcnt(%6)" : "+v"(pv[0]), "+v"(pv[1]), "+v"(pv[2]), "+v"(pv[3]), "+v"(pv[4]), "+v"(pv[5]) : "n"(kPfBehind) : "memory");
#pragma unroll
      for (uint32_t b = 0; b < kB16Codes; ++b)
#pragma unroll
        for (int h = 0; h < ND; ++h) pf[b][h] = pv[b][h];
    }
#endif
  };
  if (have && j < NS) prefetch(j);

  SubRows<K, KIND == 0 ? 0 : SH - kTagShift> sub;  // table entries are scores << kTagShift for both kinds

  // first = std::true_type for the blocks in which strip 0 can be live: its rows take row 0 (or the stored row) from above
  auto block = [&](uint32_t b, auto first) {
    constexpr bool FIRST = decltype(first)::value;
    if (j == b % (uint32_t)P && have && b < NS) {  // ---- begin strip b ----
      live = true;
      s_cur = b;
      const uint32_t r0 = b * (uint32_t)K;
      const int32_t cw = b16_first_col(b, K, dmin);
      cm1 = (uint32_t)(cw - 1);
      left = (b + 1u == NS) ? NB_last : NB;
      if (cw <= 1) {  // left of the window: column 0 (gotoh.h:117-123) ...
#pragma unroll
        for (int i = 0; i < K; ++i) ts.Hc[i] = edge(r0 + (uint32_t)i + 1u);
      } else {        // ... or outside the band
#pragma unroll
        for (int i = 0; i < K; ++i) ts.Hc[i] = neg;
      }
#pragma unroll
      for (int i = 0; i < K; ++i) ts.Ec[i] = neg;
      if (hfree && b + 1u == NS) {  // row m: free end gap (the last strip of the pair: the constants are never needed again)
#pragma unroll
        for (int i = 0; i < K; ++i)
          if ((uint32_t)i == slot_m) { ts.cx1[i] = trace_cx1<TS>(0); ts.cx2[i] = trace_cx2<TS>(0); }
      }
      bot_h = cw <= 0 ? edge(r0 + (uint32_t)K) : neg;  // what the strip below sees while this one waits for column 1
      bot_f = neg;
      if (b == 0) prev_up_h = CONT ? lrow[2 * (cw - 1 - crow0)] : row0(cw - 1);
      arrived(first);
#pragma unroll
      for (uint32_t q = 0; q < kB16Codes; ++q) {
        const uint32_t rowsel = (rcflag && q < 4u) ? 3u - q : q;  // reverse-complement view: the complement is folded into the table
#pragma unroll
        for (int h = 0; h < ND; ++h) {
          tab[(rowsel * K + 2 * h) * 64] = (int16_t)(pf[q][h] & 0xffffu);
          tab[(rowsel * K + 2 * h + 1) * 64] = (int16_t)(pf[q][h] >> 16);
        }
      }
      raw_next = code_at(cm1);
      if (b + (uint32_t)P < NS) prefetch(b + (uint32_t)P);
    }
    if (KIND == 0 && live) wp = bits + lay.block_word(s_cur, b - s_cur) * WB;  // this block's K + 1 words (B16Layout)
    uint32_t blk[K <= 4 ? 3 : K <= 8 ? 9 : 1];
#pragma unroll
    for (uint32_t q = 0; q < sizeof(blk) / sizeof(blk[0]); ++q) blk[q] = 0u;
#pragma unroll
    for (uint32_t k = 0; k < KP; ++k) {
      int32_t up_h = w.rot(bot_h, Lanes{});
      int32_t up_f = w.rot(bot_f, Lanes{});
      if (FIRST) {  // strip 0: row 0 instead of a strip above (gotoh.h:112-116)
        if (live && s_cur == 0) {
          const int32_t c = (int32_t)cm1 + 1;
          if (CONT) {
            const uint32_t i = (uint32_t)(c - crow0) < kB16RowCap ? (uint32_t)(c - crow0) : kB16RowCap - 1u;  // (past the staged columns only
            up_h = lrow[2u * i];                                                                              // when strip 0 is the last one and runs
            up_f = lrow[2u * i + 1u];                                                                         // on along row m)
            if ((uint32_t)(c - crow0) >= kB16RowCap) { up_h = neg; up_f = neg; }
          } else {
            up_h = row0(c);
            up_f = neg;
          }
        }
      }
      const uint32_t raw = raw_next;
      raw_next = code_at(cm1 + 1u);
      if (live && cm1 < n) {  // on a column 1 .. n
        qp_fetch_rows<K>(tab, raw, sub);
        int32_t nb_h, nb_f;
        if (KIND == 0) {
          uint32_t w0 = 0, w1 = 0;
          trace_step<K>(ts, up_h, up_f, prev_up_h, cy1, cy2, sub, w0, w1, nb_h, nb_f);
          if (K <= 4) blk[k >> 1] |= (w0 & 0xffffu) << (16u * (k & 1u));  // (a block's words go out together, below)
          else if (K <= 8) blk[k] = w0;
          else *reinterpret_cast<uint64_t*>(wp) = ((uint64_t)w1 << 32) | w0;
        } else {
          origin_step<K>(ts, up_h, up_f, prev_up_h, cy1, cy2, sub, nb_h, nb_f);
          if (b >= b_last) {  // watch row m (the last strip): the trailing run ends at the last column with H > E
            if (s_cur + 1u == NS) {
              int32_t hv = 0, ev = 0;
#pragma unroll
              for (int i = 0; i < K; ++i)
                if ((uint32_t)i == slot_m) { hv = ts.Hc[i]; ev = ts.Ec[i]; }
              if ((hv >> SH) > (ev >> SH)) c_end = cm1 + 1u;
            }
          }
        }
        bot_h = nb_h;
        bot_f = nb_f;
      }
      prev_up_h = up_h;
      ++cm1;
      if (KIND == 0) wp += WB;
    }
    // K = 4: the words of a block are ten consecutive bytes -- one 8-byte and one 2-byte store per block instead of five 2-byte ones
    // (a word of a column off 1 .. n is written as 0; nobody reads it).  K = 8: nine dwords, as two 16-byte stores and one of four bytes.
    if (KIND == 0 && K <= 8 && live) {
      uint8_t* const w0p = wp - KP * WB;
      if (K <= 4) {
        __builtin_memcpy(w0p, blk, 8);
        *reinterpret_cast<uint16_t*>(w0p + 8) = (uint16_t)blk[2];
      } else {
        __builtin_memcpy(w0p, blk, 4u * KP);
      }
    }
    if (live && --left == 0u) {  // past the window: the strip below finds -inf above its last K columns
      live = false;
      bot_h = neg;
      bot_f = neg;
    }
  };
  uint32_t b = 0;
  for (; b < (uint32_t)P && b < B_end; ++b) block(b, std::true_type{});
  for (; b < B_end; ++b) block(b, std::false_type{});

  // ---- score, ends, walk ----
  if (have && j == (NS - 1u) % (uint32_t)P) {
    int32_t hv = 0;
#pragma unroll
    for (int i = 0; i < K; ++i)
      if ((uint32_t)i == slot_m) hv = ts.Hc[i];
    if (a.scores) a.scores[d.out] = hv >> SH;
    if (KIND == 1) {
      a.ends[2 * d.out] = (uint32_t)(hv & kOriginMask);
      a.ends[2 * d.out + 1] = c_end;
    }
  }
  if (KIND == 0) {
    w.sync_global();
    Band16Fetch<K> fetch{bits, S, S_last, NS, n, dmin, lay};
#ifndef TRACY_B16_WALK_C
#define TRACY_B16_WALK_C 2  // (two cells per lane and round on sixteen lanes: a round of the walk costs a memory round trip whatever it looks at; A/B: band tracebacks of a decompose step 13.7 -> 13.4 ms)
#endif
    walk16<W, Band16Fetch<K>, P, (P == 4 ? 4 : TRACY_B16_WALK_C)>(w, fetch, have, m, n, have ? a.ops + a.ops_off[d.out] : nullptr, have ? a.ops_len + d.out : nullptr, a.err,
           (uint32_t)d.lastrow_off, (uint32_t)(d.lastrow_off >> 32));
  }
}


// ---- CONT on 16-bit cells ---------------------------------------------------------------------------------------------------
// The sweep below a stored row tracks no origins (Band16Args::row), so where every DP value fits int16 -- narrow_ok(): the condition
// under which the prefix rows above it were swept in 16 bits already -- it runs on the cell of the 16-bit score sweeps (dp_lane.h
// cell_left16_last / cell_down16: v_add_u16 / v_max_i16, the 2-cycle class) instead of the tagged int32 recurrence of origin_step
// (v_max_i32 / v_max3_i32, the 4.4-cycle class, + tag stripping): 9 operations per cell instead of 11 slower ones.  Values are kept
// as Hg = H + (go + ge), E, F as in score_step16g; the free trailing run of row m is a per-slot constant (rows are anchored at the
// top here, so row m can be any slot of the last strip); table entries (scores << kTagShift) are turned into score - (go + ge) when a
// lane copies the rows of its strip into LDS.  Same blocks, windows and hand-over as band16_body; score and c_e are the int32 form's.
// P = 4: the quad form (band16_body) for the narrow first tier of the pruned sweeps -- sixteen pairs per workgroup, codes two to a
// byte, kB16QuadRowCap entries of the kept row staged per pair.
template <class W, int K, int P = 16>
TR_HD void band16_cont16_body(W& w, const Band16Args& a, uint32_t wave_idx) {
  static_assert(K == 4 || K == 8 || K == 12, "strip heights of the band kernels");
  static_assert(P == 16 || (P == 4 && K == 4), "a DPP row per pair, or a quad on strips of four rows");
  constexpr uint32_t KP = (uint32_t)K + 1u;
  constexpr uint32_t NPW = 64u / (uint32_t)P;
  constexpr uint32_t RC = P == 16 ? kB16RowCap : kB16QuadRowCap;  // kept-row entries staged per pair
  constexpr bool PACKED = true;  // codes two to a byte in both forms (the LDS block decides how many workgroups a CU holds)
  using Lanes = std::integral_constant<int, P>;
  const uint32_t L = w.lane(), g = L / (uint32_t)P, j = L % (uint32_t)P;
  const uint32_t pair_idx = wave_idx * NPW + g;
  const uint32_t npairs = a.count ? *a.count : a.npairs;
  if (wave_idx * NPW >= npairs) return;
  bool have = pair_idx < npairs;
  PairDesc d{};
  if (have) d = a.pairs[a.index ? a.index[pair_idx] : pair_idx];
  if (d.flags & PAIR_SKIP) { have = false; d = PairDesc{}; }
  const uint32_t m = d.m, n = d.n;
  const int32_t dmin = band_dmin(d), dmax = band_dmax(d);
  const int32_t go = a.go, ge = a.ge, goe = go + ge;
  const bool hfree = a.hfree != 0;
  const bool rcflag = (d.flags & PAIR_A2_REVCOMP) != 0;
  const uint32_t NS = have ? b16_strips(m, K) : 0u;
  const uint32_t S = b16_window(K, dmin, dmax);
  const uint32_t S_last = have ? b16_last_window(m, n, K, dmin, dmax) : 0u;
  const uint32_t NB = S / KP, NB_last = (S_last + (uint32_t)K) / KP;
  const int32_t neg = kNegInf16;
  const uint32_t rbase = (uint32_t)d.bits_off;  // rows above the pair's first one
  auto edge_g = [&](uint32_t r) -> int32_t { return edge_value(false, go, ge, (int32_t)(r + rbase)) + goe; };  // H(r, 0) + goe

  const uint32_t code_row = PACKED ? b16_packed_row(a.code_cap) : a.code_cap;
  uint8_t* lcodes = reinterpret_cast<uint8_t*>(w.lds()) + g * code_row;
  int16_t* tab = reinterpret_cast<int16_t*>(w.lds() + NPW * code_row) + (((L & 31u) << 1) | (L >> 5));  // (lane columns interleaved: gotoh_narrow_qp_body)
  if (have) {
    const uint8_t* src = a.codes + d.a2_off;
    if (!PACKED) {
      for (uint32_t i = j; i < n; i += P) lcodes[i] = src[rcflag ? n - 1u - i : i];
    } else {
      for (uint32_t i = j; 2u * i < n; i += P) {
        const uint32_t c0 = 2u * i, c1 = c0 + 1u < n ? c0 + 1u : c0;
        const uint32_t lo = src[rcflag ? n - 1u - c0 : c0], hi = src[rcflag ? n - 1u - c1 : c1];
        lcodes[i] = (uint8_t)(lo | (hi << 4));
      }
    }
  }
  // {Hg, F} of row R for the columns strip 0 sweeps and the one before them: the dwords of the kept row (Hg low, F high)
  uint32_t* lrow = reinterpret_cast<uint32_t*>(w.lds() + NPW * code_row + b16_table_bytes(K)) + g * RC;
  const int32_t crow0 = dmin;  // column of lrow[0]: b16_first_col(0) - 1
  const uint32_t neg2 = ((uint32_t)neg & 0xffffu) | ((uint32_t)neg << 16);
  if (have) {
    const uint32_t* src = a.row + d.lastrow_off;
    for (uint32_t i = j; i < RC; i += P) {
      const int32_t cc = crow0 + (int32_t)i;
      uint32_t v = neg2;
      if (cc == 0) v = ((uint32_t)edge_g(0) & 0xffffu) | ((uint32_t)neg << 16);
      else if (cc >= 1 && cc <= (int32_t)n) v = src[cc];
      lrow[i] = v;
    }
  }
  w.sync();
  const uint32_t nclamp = n ? n - 1u : 0u;
  auto code_at = [&](uint32_t cm1) -> uint32_t {
    const uint32_t i = cm1 < nclamp ? cm1 : nclamp;
    if (!PACKED) return lcodes[i];
    return ((uint32_t)lcodes[i >> 1] >> ((i & 1u) << 2)) & 15u;
  };

  uint32_t B_end = 0, b_last = ~0u;
  {
    const uint32_t mine = have ? (NS - 1u) + NB_last : 0u;
    const uint32_t lastbeg = have ? NS - 1u : ~0u;
    for (uint32_t q = 0; q < NPW; ++q) {
      const uint32_t x = w.bcast(mine, q * (uint32_t)P), y = w.bcast(lastbeg, q * (uint32_t)P);
      B_end = x > B_end ? x : B_end;
      b_last = y < b_last ? y : b_last;
    }
  }

  int32_t Hl[K], El[K], hext[K], dlt[K];
#pragma unroll
  for (int i = 0; i < K; ++i) { Hl[i] = neg; El[i] = neg; hext[i] = ge; dlt[i] = 0; }
  int32_t bot_h = neg, bot_f = neg, prev_up_h = neg;
  int32_t gev = ge, goev = goe;
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(gev), "+v"(goev));  // two live VGPRs for the whole sweep, not re-materialised per step
#endif
  uint32_t s_cur = j, cm1 = 0, left = 0;
  bool live = false;
  uint32_t raw_next = 0, c_end = 0;
  const uint32_t slot_m = have && m ? (m - 1u) % (uint32_t)K : 0u;

  constexpr int ND = K / 2;
  uint32_t pf[kB16Codes][ND];
  auto prefetch = [&](uint32_t s) {
    const int16_t* src = a.qp + d.a1_off + (uint64_t)s * K;
#pragma unroll
    for (uint32_t b = 0; b < kB16Codes; ++b) __builtin_memcpy(pf[b], src + (uint64_t)b * d.a1_stride, 2 * K);
  };
  if (have && j < NS) prefetch(j);
  SubRows<K, 0> sub;

  auto block = [&](uint32_t b, auto first) {
    constexpr bool FIRST = decltype(first)::value;
    if (j == b % (uint32_t)P && have && b < NS) {  // ---- begin strip b ----
      live = true;
      s_cur = b;
      const uint32_t r0 = b * (uint32_t)K;
      const int32_t cw = b16_first_col(b, K, dmin);
      cm1 = (uint32_t)(cw - 1);
      left = (b + 1u == NS) ? NB_last : NB;
#pragma unroll
      for (int i = 0; i < K; ++i) {
        Hl[i] = cw <= 1 ? edge_g(r0 + (uint32_t)i + 1u) : neg;  // left of the window: column 0 (gotoh.h:117-123), or outside the band
        El[i] = neg;
      }
      if (hfree && b + 1u == NS) {  // row m: free end gap -- E = max(H, E) there
#pragma unroll
        for (int i = 0; i < K; ++i)
          if ((uint32_t)i == slot_m) { hext[i] = 0; dlt[i] = -goe; }
      }
      bot_h = cw <= 0 ? edge_g(r0 + (uint32_t)K) : neg;
      bot_f = neg;
      if (b == 0) prev_up_h = (int32_t)(lrow[cw - 1 - crow0] & 0xffffu);
#pragma unroll
      for (uint32_t q = 0; q < kB16Codes; ++q) {
        const uint32_t rowsel = (rcflag && q < 4u) ? 3u - q : q;
#pragma unroll
        for (int h = 0; h < ND; ++h) {  // entries are scores << kTagShift: back to the score, minus (go + ge) (the cell adds it to Hg)
          tab[(rowsel * K + 2 * h) * 64] = (int16_t)(((int32_t)(int16_t)(pf[q][h] & 0xffffu) >> kTagShift) - goe);
          tab[(rowsel * K + 2 * h + 1) * 64] = (int16_t)(((int32_t)(int16_t)(pf[q][h] >> 16) >> kTagShift) - goe);
        }
      }
      raw_next = code_at(cm1);
      if (b + (uint32_t)P < NS) prefetch(b + (uint32_t)P);
    }
#pragma unroll
    for (uint32_t k = 0; k < KP; ++k) {
      int32_t up_h = w.rot(bot_h, Lanes{});
      int32_t up_f = w.rot(bot_f, Lanes{});
      if (FIRST) {
        if (live && s_cur == 0) {
          const int32_t c = (int32_t)cm1 + 1;
          const uint32_t i = (uint32_t)(c - crow0) < RC ? (uint32_t)(c - crow0) : RC - 1u;
          const uint32_t v = (uint32_t)(c - crow0) < RC ? lrow[i] : neg2;
          up_h = (int32_t)(v & 0xffffu);
          up_f = (int32_t)(v >> 16);
        }
      }
      const uint32_t raw = raw_next;
      raw_next = code_at(cm1 + 1u);
      if (live && cm1 < n) {
        qp_fetch_rows<K>(tab, raw, sub);
#pragma unroll
        for (int i = K - 1; i >= 0; --i) cell_left16_last(Hl[i], El[i], hext[i], i == 0 ? prev_up_h : Hl[i - 1], sub.lo16(i), dlt[i]);
        int32_t uh = up_h, uf = up_f;
#pragma unroll
        for (int i = 0; i < K; ++i) { cell_down16(Hl[i], El[i], uh, uf, gev, goev); uh = Hl[i]; }
        if (b >= b_last) {  // watch row m (the last strip): the trailing run ends at the last column with H > E
          if (s_cur + 1u == NS) {
            int32_t hv = 0, ev = 0;
#pragma unroll
            for (int i = 0; i < K; ++i)
              if ((uint32_t)i == slot_m) { hv = Hl[i]; ev = El[i]; }
            if (sext16(hv) - goe > sext16(ev)) c_end = cm1 + 1u;
          }
        }
        bot_h = uh;
        bot_f = uf;
      }
      prev_up_h = up_h;
      ++cm1;
    }
    if (live && --left == 0u) { live = false; bot_h = neg; bot_f = neg; }
  };
  uint32_t b = 0;
  for (; b < (uint32_t)P && b < B_end; ++b) block(b, std::true_type{});
  for (; b < B_end; ++b) block(b, std::false_type{});

  if (have && j == (NS - 1u) % (uint32_t)P) {
    int32_t hv = 0;
#pragma unroll
    for (int i = 0; i < K; ++i)
      if ((uint32_t)i == slot_m) hv = Hl[i];
    if (a.scores) a.scores[d.out] = sext16(hv) - goe;
    a.ends[2 * d.out] = 0u;
    a.ends[2 * d.out + 1] = c_end;
  }
}

}  // namespace tracyhip
#endif
