// front.h -- the orientation score of a trace against its window without sweeping the window under the trace's first rows.
//
// gotohScore(trimmed trace, window) (sage.h:239-240, AlignConfig<true,false>) is the maximum of row m.  The prefix sweep
// (gotoh_prefix_body: rows 1..R over ALL columns, R = 8K) leaves row R behind (PAIR_KEEP_ROW): v(c) = max(H, F)(R, c) is the most
// any path can have collected when it leaves row R at column c, and below row R a path gains at most
// rest = sum over the rows > R of max(0, best substitution score of the row) and pays at least |ge| per gap step (go <= 0, ge < 0;
// only the trailing run of row m is free).  So:
//   1. front_place: c* = the first column with the largest v; the band kernels (band16.h, CONT) continue rows R+1..m from the
//      stored row on the diagonals c* - w .. c* + w (diagonal = column at row R), which yields S' = the best path that stays on them
//      (the last strip runs on to the end of the window: the free trailing run).
//   2. front_certify: a path that leaves row R at column c and is NOT on those diagonals all the way makes more than
//      margin(c) = min(c - (c* - w), (c* + w) - c) gap steps (-1 outside the band: none needed), so it scores at most
//      v(c) + rest - |ge| (margin(c) + 1).  If that is < S' for every column 0..n, every path scoring >= S' was inside the band:
//      S' = gotohScore, and c_e -- the first column of row m that reaches it (row_m_end_kernel) -- is the band's.  Otherwise the pair
//      takes the full sweep.  Nothing is ever decided on an uncertified score.
// A trace that matches its window somewhere (loss below |ge| w) certifies; a repeat of the target far away, or a trace of noise,
// does not.
// (A second certificate with a per-row allowance of max(row maximum, -1) instead of max(row maximum, 0) serves profiles whose rows
// cannot all reach 0: front_certify_body.)
#ifndef TRACY_AMD_FRONT_H
#define TRACY_AMD_FRONT_H

#include "band16.h"

namespace tracyhip {

struct FrontDesc {
  uint64_t row_off;     // row R of the prefix sweep: column c at row[row_off + c] (PAIR_KEEP_ROW layout)
  uint64_t a2_off;      // the window's codes
  uint64_t tab_off;     // row R + 1 of the trace in the substitution tables (int16 units)
  uint32_t tab_stride;  // rows per code of that table
  uint32_t m_rest;      // rows below R
  uint32_t n;           // columns of the window
  uint32_t flags;       // PAIR_A2_REVCOMP; PAIR_SKIP: an empty slot (front_place leaves a PAIR_SKIP pair behind, front_certify ok = 0)
  uint32_t out;         // index of the pair's outputs
  uint32_t R;           // rows of the prefix
  int32_t rest;         // sum over the rows > R of max(0, best substitution score of the row)
  uint32_t tight;       // 0: none.  Else rest - rest1 + 1, rest1 = sum over the rows > R of max(-1, best substitution score of the row):
                        // the allowance of the second certificate (front_certify_body), for profiles whose rows cannot all reach 0
};
struct FrontOut {
  int32_t vmax;     // max_c v(c)
  uint32_t cstar;   // its first column
  uint32_t shift;   // columns of the window left of the sub-window the band kernels sweep
  uint32_t ok;      // front_certify: 1 = the band's score is gotohScore
};

TR_HD int32_t front_v(uint32_t x, int32_t goe) {
  const int32_t h = sext16((int32_t)x) - goe, f = (int32_t)x >> 16;
  return h > f ? h : f;
}

// One wave per pair: c*, the band around it, the sub-window that holds the band, the pair the band kernels sweep.
// known: the pair's {vmax, c*} from an earlier tier's placement over the same kept row (or null: the row is scanned)
template <class W>
TR_HD void front_place_body(W& w, const FrontDesc& f, const uint32_t* row, int32_t goe, int32_t halfw, PairDesc* pair, FrontOut* fo,
                            const FrontOut* known = nullptr) {
  const uint32_t L = w.lane();
  if (f.flags & PAIR_SKIP) {
    if (L == 0) {
      PairDesc d{};
      d.flags = PAIR_SKIP;
      d.out = f.out;
      *pair = d;
      *fo = FrontOut{0, 0u, 0u, 0u};
    }
    return;
  }
  int32_t vmax = INT32_MIN;
  uint32_t cstar = 0;
  if (known) {  // (the same row, the same maximum: a later tier only changes the band around it)
    vmax = known->vmax;
    cstar = known->cstar;
  } else {
    const uint32_t* r = row + f.row_off;
    int32_t best = INT32_MIN;
    uint32_t bc = 0;
    // (eight rounds of 64 columns requested together, from clamped columns: a round per wait was 47 memory round trips for a 3 kb window)
    constexpr uint32_t kRounds = 8;
    for (uint32_t c0 = 1u; c0 <= f.n; c0 += 64u * kRounds) {
      uint32_t x[kRounds];
#pragma unroll
      for (uint32_t u = 0; u < kRounds; ++u) {
        const uint32_t c = c0 + 64u * u + L;
        x[u] = r[c <= f.n ? c : f.n];
      }
#pragma unroll
      for (uint32_t u = 0; u < kRounds; ++u) {
        const uint32_t c = c0 + 64u * u + L;
        const int32_t v = front_v(x[u], goe);
        const bool take = c <= f.n && v > best;
        best = take ? v : best;
        bc = take ? c : bc;
      }
    }
    for (uint32_t l = 0; l < 64u; ++l) {
      const int32_t v = (int32_t)w.bcast((uint32_t)best, l);
      const uint32_t c = w.bcast(bc, l);
      if (c != 0 && (v > vmax || (v == vmax && c < cstar))) { vmax = v; cstar = c; }
    }
  }
  if (L != 0) return;
  const int32_t dlo = (int32_t)cstar - halfw, dhi = (int32_t)cstar + halfw;
  const uint32_t a = dlo > 1 ? (uint32_t)(dlo - 1) : 0u;
  uint32_t nsub = f.n - a;
  const uint64_t reach = (uint64_t)f.m_rest + (uint64_t)(dhi - (int32_t)a) + 1ull;  // past it no cell of the band
  if ((uint64_t)nsub > reach) nsub = (uint32_t)reach;
  PairDesc d{};
  d.a1_off = f.tab_off;
  d.a1_stride = f.tab_stride;
  d.m = f.m_rest;
  d.n = nsub;
  d.a2_stride = nsub;
  d.a2_off = (f.flags & PAIR_A2_REVCOMP) ? f.a2_off + (uint64_t)(f.n - a - nsub) : f.a2_off + a;  // reverse view: column c is byte n - c
  d.out = f.out;
  d.flags = f.flags & PAIR_A2_REVCOMP;
  d.ckpt_off = band_pack(dlo - (int32_t)a, dhi - (int32_t)a);
  d.lastrow_off = f.row_off + a;
  d.bits_off = f.R;
  *pair = d;
  fo->vmax = vmax;
  fo->cstar = cstar;
  fo->shift = a;
  fo->ok = 0;
}

// One wave per pair, after the band sweep: does any path outside the band reach its score?
template <class W>
TR_HD void front_certify_body(W& w, const FrontDesc& f, const uint32_t* row, int32_t go, int32_t ge, int32_t halfw, int32_t score, uint32_t c_end,
                              FrontOut* fo) {
  const uint32_t L = w.lane();
  if (f.flags & PAIR_SKIP) return;  // (front_place left ok = 0)
  const uint32_t* r = row + f.row_off;
  const int32_t goe = go + ge;
  const int64_t age = -(int64_t)ge;
  const int64_t dlo = (int64_t)fo->cstar - halfw, dhi = (int64_t)fo->cstar + halfw;
  // Second certificate (FrontDesc::tight): a row consumed by a diagonal step gives at most its best substitution score, a row
  // consumed by a vertical gap step at most ge -- so every row gives at most x_r = max(row maximum, -1) as long as ge <= -2, a
  // vertical step then loses at least q = |ge| - 1 against that allowance and a horizontal one |ge|.  A path that leaves the band
  // below its column makes more than (c - dlo) vertical steps, one that leaves above it more than (dhi - c) horizontal ones:
  // it scores at most v(c) + rest1 - min(|ge| (dhi - c + 1), q (c - dlo + 1)).  Tighter than the first bound wherever rows cannot
  // reach 0 (a heterozygous position with two equal peaks scores -1 against either base); either certificate suffices.
  const bool two = f.tight != 0u && age >= 2;
  const int64_t rest1 = (int64_t)f.rest - ((int64_t)f.tight - 1);
  const int64_t q = age - 1;
  bool bad = false, bad1 = false;
  // (eight rounds of 64 columns requested together, from clamped columns; column 0 is the edge value, not a row entry)
  constexpr uint32_t kRounds = 8;
  for (uint32_t c0 = 0; c0 <= f.n; c0 += 64u * kRounds) {
    uint32_t x[kRounds];
#pragma unroll
    for (uint32_t u = 0; u < kRounds; ++u) {
      const uint32_t c = c0 + 64u * u + L;
      x[u] = f.n ? r[c == 0 ? 1u : c <= f.n ? c : f.n] : 0u;
    }
#pragma unroll
    for (uint32_t u = 0; u < kRounds; ++u) {
      const uint32_t c = c0 + 64u * u + L;
      const int64_t v = c == 0 ? (int64_t)edge_value(false, go, ge, (int32_t)f.R) : (int64_t)front_v(x[u], goe);
      const int64_t cc = (int64_t)c;
      const bool live = c <= f.n;
      const bool inside = cc >= dlo && cc <= dhi;
      const int64_t margin = inside ? (cc - dlo < dhi - cc ? cc - dlo : dhi - cc) : -1;
      bad = bad || (live && v + (int64_t)f.rest - age * (margin + 1) >= (int64_t)score);
      if (two) {
        const int64_t ph = age * (dhi - cc + 1), pv = q * (cc - dlo + 1);
        const int64_t pen = inside ? (ph < pv ? ph : pv) : 0;
        bad1 = bad1 || (live && v + rest1 - pen >= (int64_t)score);
      }
    }
  }
  const bool any = w.ballot(bad) != 0;
  const bool any1 = !two || w.ballot(bad1) != 0;
  if (L == 0) fo->ok = ((!any || !any1) && c_end != 0) ? 1u : 0u;
}

}  // namespace tracyhip
#endif
