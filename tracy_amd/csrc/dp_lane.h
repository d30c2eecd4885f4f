// dp_lane.h -- per-lane arithmetic of the anti-diagonal Gotoh / NW wavefront kernels.
//
// One 64-lane wave aligns one pair.  Lane L owns K consecutive DP rows (a "row strip"); at step t it
// computes the K cells of column c = t - L, so the wave as a whole walks the anti-diagonals of the
// (row-strip x column) grid.  Everything a lane needs from its own previous column lives in
// registers (the structs below); the bottom row of the strip above arrives through one DPP
// wave_shr:1 per value per step.  This header holds only the per-lane math (no memory, no cross-lane
// traffic) so that tests/emu can run exactly the same code on the host against the oracle.
//
// Exactness contract (reference: /root/reference/src/gotoh.h:103-141):
//   E = max(Hleft + hgap(go+ge), Eleft + hgap(ge))        newhoz   gotoh.h:130
//   F = max(Hup   + vgap(go+ge), Fup   + vgap(ge))        v[col]   gotoh.h:131
//   H = max(Hdiag + sub, E, F)                            s[col]   gotoh.h:132
//   bit3 = (H == E); bit4 = !bit3 && (H == F); bit1 = (E != Eleft + hgap(ge)); bit2 = (F != Fup + vgap(ge))
//
// Tagged formulation used by the traceback kernel: all scores are kept multiplied by 32 and the four
// predicates ride in the low 5 bits of the maxima, so no compare instructions are needed:
//   X1 = Hleft32 + (hopen*32 + 7)      X2 = Eleft32 + (hext*32 + 10)     Et = max(X1, X2)   tag 7 / 10
//   Y1 = Hup32   + (vopen*32 + 1)      Y2 = Fup32   + (vext*32 + 6)      Ft = max(Y1, Y2)   tag 1 / 6
//   Dt = Hdiag32 + sub*32                                                 Ht = max3(Dt, Et, Ft)
// Equal scores are separated by the tags exactly as the reference's predicates order them: extend beats
// open on ties (bit1/bit2 are set only when open wins strictly: 10 > 7, 6 > 1), E beats F beats diagonal
// (min E tag 7 > max F tag 6 > 0).  The stored nibble is (Et + Ft + Ht) mod 16: with these tag values
// the twelve (bit1, bit2, source) combinations give twelve different nibbles (kNibbleDecode), so two
// plain adds replace the compare / select / bit-field sequence.  On gfx950 v_add_u32 and v_and_b32 issue
// in ~2 cycles per wave64 while v_max_i32, v_max3_i32, v_bfi_b32, v_alignbit_b32 take ~4
// (tools/ubench/valu_rate.hip), which is what this encoding is shaped for.
// Scores stay exact because tags never exceed 31 and are stripped (& ~31) before a value is reused.
#ifndef TRACY_AMD_DP_LANE_H
#define TRACY_AMD_DP_LANE_H

#include <stdint.h>

#if defined(__HIPCC__)
#define TR_HD __host__ __device__ __forceinline__
#else
#define TR_HD inline
#endif

namespace tracyhip {

constexpr int32_t kNegInf = -1000000;  // -sc.inf, align.h:26,30
constexpr int kTagShift = 5;
constexpr int32_t kTagMask = 31;
constexpr int32_t kTagEOpen = 7, kTagEExt = 10, kTagFOpen = 1, kTagFExt = 6;

TR_HD int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }
TR_HD int32_t imax3(int32_t a, int32_t b, int32_t c) { return imax(imax(a, b), c); }
TR_HD int32_t imin32(int32_t a, int32_t b) { return a < b ? a : b; }

// shift a nibble into the top of a 32-bit accumulator: v_alignbit_b32
TR_HD uint32_t push_nibble(uint32_t acc, uint32_t nib_src) { return (acc >> 4) | (nib_src << 28); }

// ---- traceback lane -------------------------------------------------------------------------
template <int K>
struct TraceLane {
  int32_t Hc[K];   // H[r][c-1] * 32
  int32_t Ec[K];   // E[r][c-1] * 32
  int32_t cx1[K];  // horizontal open  constant of row r: hgap(go+ge)*32 + 7
  int32_t cx2[K];  // horizontal extend constant of row r: hgap(ge)*32 + 10
};

// TS = position of the five tag bits.  0: the traceback kernels (score above bit 5).  kOriginBits: the origin-tracking
// sweep, where the low kOriginBits bits of every value carry a column index along (see origin_step).
constexpr int kOriginBits = 13;                        // reference columns 0 .. 8191
constexpr int kOriginShift = kOriginBits + kTagShift;  // score field of the origin-tracking sweep: bits 18 .. 31
constexpr int32_t kOriginMask = (1 << kOriginBits) - 1;
template <int TS = 0>
TR_HD int32_t trace_cx1(int32_t open_cost) { return (int32_t)((uint32_t)open_cost << (TS + kTagShift)) + (kTagEOpen << TS); }
template <int TS = 0>
TR_HD int32_t trace_cx2(int32_t ext_cost) { return (int32_t)((uint32_t)ext_cost << (TS + kTagShift)) + (kTagEExt << TS); }
template <int TS = 0>
TR_HD int32_t trace_cy1(int32_t open_cost) { return (int32_t)((uint32_t)open_cost << (TS + kTagShift)) + (kTagFOpen << TS); }
template <int TS = 0>
TR_HD int32_t trace_cy2(int32_t ext_cost) { return (int32_t)((uint32_t)ext_cost << (TS + kTagShift)) + (kTagFExt << TS); }

// One column of the strip.  up_h/up_f: H,F (x32, clean) of the row above at this column; diag: H of
// the row above at the previous column; sub(i) returns the substitution score of slot i x32.
// Outputs: w0/w1 = 16 nibbles (slot i at bits 4i of w1:w0), bot_h/bot_f = clean H,F of the last slot.
//
// Two sweeps over the strip so that every state register is updated in place (no end-of-step copies):
//   A (bottom-up, independent of the vertical chain): Hc[i] <- Hdiag + sub = old Hc[i-1] + sub(i),
//     Ec[i] <- Et.  Going upwards, old Hc[i] has been consumed by slot i+1 before slot i overwrites it.
//   B (top-down, the F/H chain): Ft, Ht, nibble, strip the tags.
template <int K, class Sub>
TR_HD void trace_step(TraceLane<K>& s, int32_t up_h, int32_t up_f, int32_t diag, int32_t cy1, int32_t cy2,
                      const Sub& sub, uint32_t& w0, uint32_t& w1, int32_t& bot_h, int32_t& bot_f) {
#pragma unroll
  for (int i = K - 1; i >= 0; --i) {
    const int32_t et = imax(s.Hc[i] + s.cx1[i], s.Ec[i] + s.cx2[i]);
    const int32_t dt = (i == 0 ? diag : s.Hc[i - 1]) + sub(i);
    s.Hc[i] = dt;
    s.Ec[i] = et;
  }
  uint32_t a0 = 0, a1 = 0;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int32_t ft = imax(up_h + cy1, up_f + cy2);
    const int32_t ht = imax3(s.Hc[i], s.Ec[i], ft);
    const uint32_t nib = (uint32_t)s.Ec[i] + (uint32_t)ft + (uint32_t)ht;  // low 4 bits = tag sum mod 16
    if (i < 8) a0 = push_nibble(a0, nib);
    else a1 = push_nibble(a1, nib);
    s.Hc[i] = ht & ~kTagMask;
    s.Ec[i] = s.Ec[i] & ~kTagMask;
    up_h = s.Hc[i];
    up_f = ft & ~kTagMask;
  }
  if (K <= 8) a0 >>= (4 * (8 - K)) & 31;
  else a1 >>= (4 * (16 - K)) & 31;
  w0 = a0;
  w1 = a1;
  bot_h = up_h;
  bot_f = up_f;
}

// Origin-tracking sweep: the column at which the reference's traceback (gotoh.h:143-167) from a cell reaches row 0,
// carried along with the scores instead of being recovered from stored trace bits.  Every value is
//     score << 18  |  tag << 13  |  origin (13 bits)
// The tags order equal scores exactly as in trace_step (extend beats open; E beats F beats the diagonal), so each
// maximum selects the very predecessor the traceback would follow, and the winner's low bits -- its origin -- ride
// along for free:   O_s(r,c) = bit3 ? O_h(r,c) : bit4 ? O_v(r,c) : O_s(r-1,c-1),   O_h(r,c) = bit1 ? O_s(r,c-1) : O_h(r,c-1),
// O_v(r,c) = bit2 ? O_s(r-1,c) : O_v(r-1,c),   O_s(0,c) = c.   Costs are added as multiples of 2^18 and never touch
// the low 18 bits; tags are stripped after each maximum.  No traceback words are written: trimReferenceSlice
// (fmindex.h:429-463) only needs the two ends of the alignment.
template <int K, class Sub>
TR_HD void origin_step(TraceLane<K>& s, int32_t up_h, int32_t up_f, int32_t diag, int32_t cy1, int32_t cy2, const Sub& sub,
                       int32_t& bot_h, int32_t& bot_f) {
  constexpr int32_t strip = ~(kTagMask << kOriginBits);
#pragma unroll
  for (int i = K - 1; i >= 0; --i) {
    const int32_t et = imax(s.Hc[i] + s.cx1[i], s.Ec[i] + s.cx2[i]);
    const int32_t dt = (i == 0 ? diag : s.Hc[i - 1]) + sub(i);
    s.Hc[i] = dt;
    s.Ec[i] = et;
  }
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int32_t ft = imax(up_h + cy1, up_f + cy2);
    const int32_t ht = imax3(s.Hc[i], s.Ec[i], ft);  // tags: E (7 / 10) beats F (1 / 6) beats the diagonal (0)
    s.Hc[i] = ht & strip;
    s.Ec[i] = s.Ec[i] & strip;
    up_h = s.Hc[i];
    up_f = ft & strip;
  }
  bot_h = up_h;
  bot_f = up_f;
}

// decode one stored nibble into the reference's four trace bits: nibble = (Te + Tf + W) mod 16 with
// Te in {7 open, 10 extend}, Tf in {1 open, 6 extend}, W in {0 diagonal, Te (E won), Tf (F won)}
struct TraceBits {
  bool bit1, bit2, bit3, bit4;
};
TR_HD TraceBits decode_nibble(uint32_t nib) {
  TraceBits b = {false, false, false, false};
#pragma unroll
  for (int eo = 0; eo < 2; ++eo) {
#pragma unroll
    for (int fo = 0; fo < 2; ++fo) {
      const int te = eo ? kTagEOpen : kTagEExt, tf = fo ? kTagFOpen : kTagFExt;
#pragma unroll
      for (int src = 0; src < 3; ++src) {
        const int wv = src == 0 ? 0 : src == 1 ? te : tf;
        if ((uint32_t)((te + tf + wv) & 15) == nib) {
          b.bit1 = eo != 0;   // E opened from H strictly (gotoh.h:137)
          b.bit2 = fo != 0;   // F opened from H strictly (gotoh.h:138)
          b.bit3 = src == 1;  // H == E (gotoh.h:135)
          b.bit4 = src == 2;  // else H == F (gotoh.h:136)
        }
      }
    }
  }
  return b;
}

// ---- score-only lane (plain int32, no tags) ---------------------------------------------------
template <int K>
struct ScoreLane {
  int32_t Hl[K];    // H[r][c-1]
  int32_t El[K];    // E[r][c-1]
  int32_t hopen[K]; // hgap(go+ge) of row r
  int32_t hext[K];  // hgap(ge) of row r
};

template <int K, class Sub>
TR_HD void score_step(ScoreLane<K>& s, int32_t up_h, int32_t up_f, int32_t diag, int32_t vopen, int32_t vext,
                      const Sub& sub, int32_t& bot_h, int32_t& bot_f) {
  // same two in-place sweeps as trace_step
#pragma unroll
  for (int i = K - 1; i >= 0; --i) {
    const int32_t e = imax(s.Hl[i] + s.hopen[i], s.El[i] + s.hext[i]);
    const int32_t d = (i == 0 ? diag : s.Hl[i - 1]) + sub(i);
    s.Hl[i] = d;
    s.El[i] = e;
  }
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int32_t f = imax(up_h + vopen, up_f + vext);
    const int32_t h = imax3(s.Hl[i], s.El[i], f);
    s.Hl[i] = h;
    up_h = h;
    up_f = f;
  }
  bot_h = up_h;
  bot_f = up_f;
}

// ---- narrow (16-bit) score-only lane ------------------------------------------------------------
// gfx950 issues the 16-bit VOP2 forms v_add_u16 / v_max_i16 in ~2 cycles per wave64, v_max_i32 and
// v_max3_i32 in ~4 (tools/ubench/valu_rate.hip).  With free end gaps on the first/last row every real
// DP value lies in [go + rows*ge + 2(go+ge), rows*match], which fits int16 for Sanger-sized traces, and
// the -inf sentinel only has to lose its one comparison (kNegInf16 + ge < any real value): the 16-bit
// kernel then computes exactly the same maxima.  Registers are int32; only their low halves are live.
constexpr int32_t kNegInf16 = -20000;

TR_HD int32_t add16(int32_t a, int32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  int32_t d;
  asm("v_add_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
#else
  return (int32_t)(uint16_t)((uint32_t)a + (uint32_t)b);
#endif
}
TR_HD int32_t max16(int32_t a, int32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  int32_t d;
  asm("v_max_i16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
#else
  const int16_t x = (int16_t)(uint16_t)a, y = (int16_t)(uint16_t)b;
  return (int32_t)(uint16_t)(x > y ? x : y);
#endif
}
TR_HD int32_t sext16(int32_t a) { return (int32_t)(int16_t)(uint16_t)a; }

// max(a, b + c) in one asm statement (see chain16)
TR_HD int32_t maxadd16(int32_t a, int32_t b, int32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  int32_t d;
  asm("v_add_u16 %0, %2, %3\n\tv_max_i16 %0, %1, %0" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
#else
  return max16(a, add16(b, c));
#endif
}

// the vertical chain of one cell in ONE asm statement:  f = max(up_hg, up_f + vext);  hg = max(t, f) + goe.
// Separate asm statements make the compiler put an s_nop between every dependent pair (it cannot see what an
// asm writes); plain VALU -> VALU dependencies are interlocked by the hardware and need none.
TR_HD void chain16(int32_t up_hg, int32_t up_f, int32_t vext, int32_t t, int32_t goe, int32_t& f, int32_t& hg) {
#if defined(__HIP_DEVICE_COMPILE__)
  int32_t f_, hg_;
  asm("v_add_u16 %0, %3, %4\n\tv_max_i16 %0, %2, %0\n\tv_max_i16 %1, %5, %0\n\tv_add_u16 %1, %1, %6"
      : "=&v"(f_), "=&v"(hg_)
      : "v"(up_hg), "v"(up_f), "v"(vext), "v"(t), "v"(goe));
  f = f_;
  hg = hg_;
#else
  f = max16(up_hg, add16(up_f, vext));
  hg = add16(max16(t, f), goe);
#endif
}

// sub.lo16(i): a register whose low 16 bits hold the substitution score of slot i
template <int K, class Sub>
TR_HD void score_step16(ScoreLane<K>& s, int32_t up_h, int32_t up_f, int32_t diag, int32_t vopen, int32_t vext,
                        const Sub& sub, int32_t& bot_h, int32_t& bot_f) {
#pragma unroll
  for (int i = K - 1; i >= 0; --i) {
    const int32_t e = max16(add16(s.Hl[i], s.hopen[i]), add16(s.El[i], s.hext[i]));
    const int32_t d = add16(i == 0 ? diag : s.Hl[i - 1], sub.lo16(i));
    s.Hl[i] = d;
    s.El[i] = e;
  }
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int32_t f = max16(add16(up_h, vopen), add16(up_f, vext));
    const int32_t h = max16(max16(s.Hl[i], s.El[i]), f);
    s.Hl[i] = h;
    up_h = h;
    up_f = f;
  }
  bot_h = up_h;
  bot_f = up_f;
}

// 16-bit score step with the gap-open term shared between the horizontal and the vertical move.
// Each cell stores Hg = H + (go+ge) instead of H: the cell to the right needs H + hopen, the cell below
// H + vopen, and both equal Hg except on the free last row (delta_last = hopen_m - (go+ge), applied to the
// last slot only: rows are anchored at the bottom, so row m is always slot K-1).  The diagonal term
// H + sub becomes Hg + (sub - (go+ge)): the constant is folded into the query-profile table.
// 8 VALU ops per cell: E: add, max; F: add, max; diagonal: add; H: max, max; Hg: add.
// The state registers are updated IN PLACE by two asm statements per cell (tied "+v" operands).  The steps of the
// sweep sit under `if (active)`: with out-of-place results the join needs the new value back in the old register,
// and whenever the scheduler lets two generations of a slot overlap that costs a v_mov per cell and step.
//   cell_left16:  E = max(Hg + delta, E + hext);  Hg <- Hg(row above, previous column) + sub     (3 ops, 4 on row m)
//   cell_down16:  f = max(up_hg, up_f + vext);  Hg <- max(max(Hg, E), f) + goe                    (5 ops)
TR_HD void cell_left16(int32_t& hl, int32_t& el, int32_t hext, int32_t diag_hg, int32_t sub) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("v_add_u16 %1, %1, %2\n\tv_max_i16 %1, %0, %1\n\tv_add_u16 %0, %3, %4" : "+v"(hl), "+v"(el) : "v"(hext), "v"(diag_hg), "v"(sub));
#else
  el = max16(hl, add16(el, hext));
  hl = add16(diag_hg, sub);
#endif
}
TR_HD void cell_left16_last(int32_t& hl, int32_t& el, int32_t hext, int32_t diag_hg, int32_t sub, int32_t delta) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("v_add_u16 %0, %0, %5\n\tv_add_u16 %1, %1, %2\n\tv_max_i16 %1, %0, %1\n\tv_add_u16 %0, %3, %4"
      : "+v"(hl), "+v"(el) : "v"(hext), "v"(diag_hg), "v"(sub), "v"(delta));
#else
  el = max16(add16(hl, delta), add16(el, hext));
  hl = add16(diag_hg, sub);
#endif
}
TR_HD void cell_down16(int32_t& hl, int32_t el, int32_t up_hg, int32_t& up_f, int32_t vext, int32_t goe) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("v_max_i16 %0, %0, %2\n\tv_add_u16 %1, %1, %4\n\tv_max_i16 %1, %3, %1\n\tv_max_i16 %0, %0, %1\n\tv_add_u16 %0, %0, %5"
      : "+v"(hl), "+v"(up_f) : "v"(el), "v"(up_hg), "v"(vext), "v"(goe));
#else
  up_f = max16(up_hg, add16(up_f, vext));
  hl = add16(max16(max16(hl, el), up_f), goe);
#endif
}

// ---- whole strips in a few asm statements (the 16-bit query-profile sweep, dp_kernels.h gotoh_narrow_qp_body) ----------------
// The compiler cannot see what an asm statement writes and puts an s_nop between dependent statements; with one statement
// per cell that is ~16 wasted issue cycles per step.  These helpers run N cells per statement (operand limit: 30).
// Semantics per cell are those of cell_left16 / cell_down16 above.
//   strip_left16<N, LAST>:  bottom-up over slots N-1 .. 0:  E <- max(H, E + hext);  H <- H(slot above, previous column) + sub.
//     dg_in = that H for slot 0 of the chunk.  LAST: slot N-1 is the strip's last slot -- it uses its own extension cost
//     hext_last and first adds delta_last to H (row m: free horizontal gaps, dp_kernels.h).
//   strip_down16<N>:  top-down:  f <- max(up, f + vext);  H <- max(max(H, E), f) + goe;  up <- H.
template <int N, bool LAST>
TR_HD void strip_left16(int32_t* hl, int32_t* el, const int32_t* sub, int32_t dg_in, int32_t hext, int32_t hext_last, int32_t delta_last) {
#if defined(__HIP_DEVICE_COMPILE__)
#define TR_LCELL(i, dg) "v_add_u16 %[e" #i "], %[e" #i "], %[hx]\n\tv_max_i16 %[e" #i "], %[h" #i "], %[e" #i "]\n\tv_add_u16 %[h" #i "], %[" dg "], %[s" #i "]\n\t"
#define TR_LCELL_LAST(i, dg) "v_add_u16 %[h" #i "], %[h" #i "], %[dl]\n\tv_add_u16 %[e" #i "], %[e" #i "], %[hl]\n\tv_max_i16 %[e" #i "], %[h" #i "], %[e" #i "]\n\tv_add_u16 %[h" #i "], %[" dg "], %[s" #i "]\n\t"
#define TR_LOPS8 [h0] "+v"(hl[0]), [h1] "+v"(hl[1]), [h2] "+v"(hl[2]), [h3] "+v"(hl[3]), [h4] "+v"(hl[4]), [h5] "+v"(hl[5]), [h6] "+v"(hl[6]), [h7] "+v"(hl[7]), \
                 [e0] "+v"(el[0]), [e1] "+v"(el[1]), [e2] "+v"(el[2]), [e3] "+v"(el[3]), [e4] "+v"(el[4]), [e5] "+v"(el[5]), [e6] "+v"(el[6]), [e7] "+v"(el[7])
#define TR_LOPS7 [h0] "+v"(hl[0]), [h1] "+v"(hl[1]), [h2] "+v"(hl[2]), [h3] "+v"(hl[3]), [h4] "+v"(hl[4]), [h5] "+v"(hl[5]), [h6] "+v"(hl[6]), \
                 [e0] "+v"(el[0]), [e1] "+v"(el[1]), [e2] "+v"(el[2]), [e3] "+v"(el[3]), [e4] "+v"(el[4]), [e5] "+v"(el[5]), [e6] "+v"(el[6])
#define TR_LIN8 [s0] "v"(sub[0]), [s1] "v"(sub[1]), [s2] "v"(sub[2]), [s3] "v"(sub[3]), [s4] "v"(sub[4]), [s5] "v"(sub[5]), [s6] "v"(sub[6]), [s7] "v"(sub[7]), \
                [dg] "v"(dg_in), [hx] "v"(hext)
#define TR_LIN7 [s0] "v"(sub[0]), [s1] "v"(sub[1]), [s2] "v"(sub[2]), [s3] "v"(sub[3]), [s4] "v"(sub[4]), [s5] "v"(sub[5]), [s6] "v"(sub[6]), \
                [dg] "v"(dg_in), [hx] "v"(hext)
  static_assert(N == 7 || N == 8, "chunks of 7 or 8 cells");
  if constexpr (N == 8 && !LAST) {
    asm(TR_LCELL(7, "h6") TR_LCELL(6, "h5") TR_LCELL(5, "h4") TR_LCELL(4, "h3") TR_LCELL(3, "h2") TR_LCELL(2, "h1") TR_LCELL(1, "h0") TR_LCELL(0, "dg")
        : TR_LOPS8 : TR_LIN8);
  } else if constexpr (N == 8 && LAST) {
    asm(TR_LCELL_LAST(7, "h6") TR_LCELL(6, "h5") TR_LCELL(5, "h4") TR_LCELL(4, "h3") TR_LCELL(3, "h2") TR_LCELL(2, "h1") TR_LCELL(1, "h0") TR_LCELL(0, "dg")
        : TR_LOPS8 : TR_LIN8, [hl] "v"(hext_last), [dl] "v"(delta_last));
  } else if constexpr (N == 7 && !LAST) {
    asm(TR_LCELL(6, "h5") TR_LCELL(5, "h4") TR_LCELL(4, "h3") TR_LCELL(3, "h2") TR_LCELL(2, "h1") TR_LCELL(1, "h0") TR_LCELL(0, "dg")
        : TR_LOPS7 : TR_LIN7);
  } else {
    asm(TR_LCELL_LAST(6, "h5") TR_LCELL(5, "h4") TR_LCELL(4, "h3") TR_LCELL(3, "h2") TR_LCELL(2, "h1") TR_LCELL(1, "h0") TR_LCELL(0, "dg")
        : TR_LOPS7 : TR_LIN7, [hl] "v"(hext_last), [dl] "v"(delta_last));
  }
#undef TR_LCELL
#undef TR_LCELL_LAST
#undef TR_LOPS8
#undef TR_LOPS7
#undef TR_LIN8
#undef TR_LIN7
#else
  for (int i = N - 1; i >= 0; --i) {
    const int32_t dg = i == 0 ? dg_in : hl[i - 1];
    if (LAST && i == N - 1) cell_left16_last(hl[i], el[i], hext_last, dg, sub[i], delta_last);
    else cell_left16(hl[i], el[i], hext, dg, sub[i]);
  }
#endif
}

template <int N>
TR_HD void strip_down16(int32_t* hl, const int32_t* el, int32_t up_in, int32_t& f, int32_t vext, int32_t goe) {
#if defined(__HIP_DEVICE_COMPILE__)
#define TR_DCELL(i, up) "v_max_i16 %[h" #i "], %[h" #i "], %[e" #i "]\n\tv_add_u16 %[f], %[f], %[vx]\n\tv_max_i16 %[f], %[" up "], %[f]\n\tv_max_i16 %[h" #i "], %[h" #i "], %[f]\n\tv_add_u16 %[h" #i "], %[h" #i "], %[go]\n\t"
  static_assert(N == 7 || N == 8, "chunks of 7 or 8 cells");
  if constexpr (N == 8) {
    asm(TR_DCELL(0, "up") TR_DCELL(1, "h0") TR_DCELL(2, "h1") TR_DCELL(3, "h2") TR_DCELL(4, "h3") TR_DCELL(5, "h4") TR_DCELL(6, "h5") TR_DCELL(7, "h6")
        : [h0] "+v"(hl[0]), [h1] "+v"(hl[1]), [h2] "+v"(hl[2]), [h3] "+v"(hl[3]), [h4] "+v"(hl[4]), [h5] "+v"(hl[5]), [h6] "+v"(hl[6]), [h7] "+v"(hl[7]), [f] "+v"(f)
        : [e0] "v"(el[0]), [e1] "v"(el[1]), [e2] "v"(el[2]), [e3] "v"(el[3]), [e4] "v"(el[4]), [e5] "v"(el[5]), [e6] "v"(el[6]), [e7] "v"(el[7]),
          [up] "v"(up_in), [vx] "v"(vext), [go] "v"(goe));
  } else {
    asm(TR_DCELL(0, "up") TR_DCELL(1, "h0") TR_DCELL(2, "h1") TR_DCELL(3, "h2") TR_DCELL(4, "h3") TR_DCELL(5, "h4") TR_DCELL(6, "h5")
        : [h0] "+v"(hl[0]), [h1] "+v"(hl[1]), [h2] "+v"(hl[2]), [h3] "+v"(hl[3]), [h4] "+v"(hl[4]), [h5] "+v"(hl[5]), [h6] "+v"(hl[6]), [f] "+v"(f)
        : [e0] "v"(el[0]), [e1] "v"(el[1]), [e2] "v"(el[2]), [e3] "v"(el[3]), [e4] "v"(el[4]), [e5] "v"(el[5]), [e6] "v"(el[6]),
          [up] "v"(up_in), [vx] "v"(vext), [go] "v"(goe));
  }
#undef TR_DCELL
#else
  for (int i = 0; i < N; ++i) {
    cell_down16(hl[i], el[i], up_in, f, vext, goe);
    up_in = hl[i];
  }
#endif
}

template <int K, class Sub>
TR_HD void score_step16g(ScoreLane<K>& s, int32_t up_hg, int32_t up_f, int32_t diag_hg, int32_t vext, int32_t goe,
                         int32_t delta_last, const Sub& subg, int32_t& bot_hg, int32_t& bot_f) {
#pragma unroll
  for (int i = K - 1; i >= 0; --i) {
    const int32_t dg = i == 0 ? diag_hg : s.Hl[i - 1];
    if (i == K - 1) cell_left16_last(s.Hl[i], s.El[i], s.hext[i], dg, subg.lo16(i), delta_last);
    else cell_left16(s.Hl[i], s.El[i], s.hext[i], dg, subg.lo16(i));
  }
#pragma unroll
  for (int i = 0; i < K; ++i) {
    cell_down16(s.Hl[i], s.El[i], up_hg, up_f, vext, goe);
    up_hg = s.Hl[i];
  }
  bot_hg = up_hg;
  bot_f = up_f;
}

// ---- Needleman-Wunsch lane (needle.h:101-110): one value per cell, two trace bits ----------------
//   S = max(max(Sdiag + sub, Sup + vgap(ge)), Sleft + hgap(ge)); bit3 = (S == hor) else bit4 = (S == ver)
// tags: hor*4 + 2, ver*4 + 1, diag*4 + 0  -> low two bits of the max are (bit3, bit4).
template <int K>
struct NeedleLane {
  int32_t Sc[K];   // S[r][c-1] * 4 (traceback) or plain (score only)
  int32_t hx[K];   // hgap(ge) of row r (x4 + 2 for traceback)
};

template <int K, bool TRACE, class Sub>
TR_HD void needle_step(NeedleLane<K>& s, int32_t up_s, int32_t diag, int32_t vy, const Sub& sub, uint32_t& w0,
                       int32_t& bot_s) {
  uint32_t a0 = 0;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int32_t hor = s.Sc[i] + s.hx[i];
    const int32_t ver = up_s + vy;
    const int32_t st = imax3(diag + sub(i), ver, hor);
    diag = s.Sc[i];
    if (TRACE) {
      a0 = (a0 >> 2) | ((uint32_t)st << 30);
      s.Sc[i] = st & ~3;
    } else {
      s.Sc[i] = st;
    }
    up_s = s.Sc[i];
  }
  if (TRACE) a0 >>= (2 * (16 - K)) & 31;
  w0 = a0;
  bot_s = up_s;
}

// ---- orientation vote (pipeline.hip kmer_vote_kernel): shared 11-mers of a trace with its window read forward (vf) and as
// the reverse complement (vr).  A clear majority marks the other strand as the likely loser: its sweep leaves no
// checkpoints / row-m values behind (nobody will trace back from them).  Only a guess about which work is worth keeping:
// when the likely loser wins after all, the pipeline sweeps it once more, with checkpoints.
TR_HD bool vote_skips_checkpoints(uint32_t vf, uint32_t vr, uint32_t orient /*0 forward, 1 reverse*/) {
  const uint32_t hi = vf >= vr ? vf : vr, lo = vf >= vr ? vr : vf;
  const bool clear = hi >= 32u && hi >= 2u * lo;
  const uint32_t likely = vf >= vr ? 0u : 1u;
  return clear && orient != likely;
}

// ---- geometry shared by the DP kernels and the traceback walker ---------------------------------
// A pass covers 64*K rows; pass p holds rows p*64K+1 .. (p+1)*64K.  Step t of a pass (1-based) puts
// lane L on column t - L.  The traceback words of pair are laid out [pass][step][lane] (8 bytes per
// lane and step for Gotoh, 4 for NW), i.e. each step of a wave is one contiguous 512-byte store.
struct CellAddr {
  uint32_t pass, lane, slot;
};
TR_HD CellAddr cell_addr(uint32_t row /*>=1*/, int K) {
  const uint32_t g = row - 1;
  const uint32_t rows_per_pass = 64u * (uint32_t)K;
  CellAddr a;
  a.pass = g / rows_per_pass;
  const uint32_t gl = g % rows_per_pass;
  a.lane = gl / (uint32_t)K;
  a.slot = gl % (uint32_t)K;
  return a;
}
TR_HD uint32_t steps_per_pass(uint32_t n) { return n + 63u; }
TR_HD uint64_t word_index(uint32_t pass, uint32_t step /*1-based*/, uint32_t lane, uint32_t n) {
  return ((uint64_t)pass * steps_per_pass(n) + (step - 1u)) * 64u + lane;
}
TR_HD uint32_t num_passes(uint32_t m, int K) { return (m + 64u * (uint32_t)K - 1u) / (64u * (uint32_t)K); }

// first-row / first-column values (gotoh.h:112-123): go + i*ge unless that edge is free
TR_HD int32_t edge_value(bool is_free, int32_t go, int32_t ge, int32_t i) { return is_free ? 0 : go + i * ge; }

// reference character -> profile row (align.h:121-136); 6 = "other" (all-zero column)
TR_HD uint32_t base_code(uint8_t c) {
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    case 'N': case 'n': return 4;
    case '-': return 5;
    default: return 6;
  }
}
// what the DP kernels read as a reference column: '-' and every other letter score 0 against all rows, one code for both
TR_HD uint32_t dp_code(uint8_t c) { const uint32_t k = base_code(c); return k > 5u ? 5u : k; }
// reverseComplementProfile (profile.h:74-90) seen on codes: rows 0<->3, 1<->2; 4, 5 and "other" stay
TR_HD uint32_t complement_code(uint32_t code) { return code < 4 ? 3u - code : code; }

// reverseComplement(std::string) (fmindex.h:8-24) for the [ACGTN] alphabet it rewrites
TR_HD uint8_t complement_char(uint8_t c) {
  return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c;
}

// _score for a float profile row against a one-hot column b (align.h:103-118): the 20 terms whose
// p2 entry is 0 contribute +-0 and leave the float accumulator unchanged, so only k2 == b remains.
TR_HD int32_t onehot_score(const float p1[5], uint32_t b, float fmatch, float fmismatch) {
  float acc = 0.0f;
#pragma unroll
  for (uint32_t k1 = 0; k1 < 5; ++k1) {
    const float w = (k1 == b) ? fmatch : fmismatch;
#if defined(__HIP_DEVICE_COMPILE__)
    acc = __fadd_rn(acc, __fmul_rn(p1[k1], w));
#else
    acc = acc + p1[k1] * w;
#endif
  }
  return (int32_t)acc;
}

// full 25-term float _score (align.h:112-116), k1 outer / k2 inner, every operation rounded to float
template <int NT = 5>
TR_HD int32_t profile_score(const float a[5], const float b[5], float fmatch, float fmismatch) {
  // NT = 4: row 4 ('N') of BOTH profiles is zero everywhere (trace profiles, profile.h:37-38).  Its nine terms are
  // +-0, and acc + (+-0) == acc for every acc this sum can hold (acc is never -0), so they are left out exactly.
  float acc = 0.0f;
#pragma unroll
  for (int k1 = 0; k1 < NT; ++k1) {
#pragma unroll
    for (int k2 = 0; k2 < NT; ++k2) {
      const float w = (k1 == k2) ? fmatch : fmismatch;
#if defined(__HIP_DEVICE_COMPILE__)
      acc = __fadd_rn(acc, __fmul_rn(__fmul_rn(a[k1], b[k2]), w));
#else
      acc = acc + (a[k1] * b[k2]) * w;
#endif
    }
  }
  return (int32_t)acc;
}

// _profileConsChar (align.h:254-270) on one column
TR_HD uint8_t cons_char(const float p[6]) {
  uint32_t maxidx = 0;
  double maxval = p[0];
  for (uint32_t k = 1; k < 6; ++k) {
    if (p[k] > maxval) { maxval = p[k]; maxidx = k; }
  }
  return maxidx == 0 ? 'A' : maxidx == 1 ? 'C' : maxidx == 2 ? 'G' : maxidx == 3 ? 'T' : 'N';
}

}  // namespace tracyhip
#endif
