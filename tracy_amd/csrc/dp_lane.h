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
// Tagged formulation used by the traceback kernel: all scores are kept multiplied by 16 and the four
// predicates ride in the low nibble of the maxima, so no compare instructions are needed:
//   X1 = Hleft16 + (hopen*16 + 8)      X2 = Eleft16 + (hext*16 + 8 + 1)      Et = max(X1, X2)
//   Y1 = Hup16   + (vopen*16 + 4)      Y2 = Fup16   + (vext*16 + 4 + 2)      Ft = max(Y1, Y2)
//   Dt = Hdiag16 + sub*16                                                     Ht = max3(Dt, Et, Ft)
// Equal scores are separated by the tags exactly as the reference's predicates order them:
//   extend beats open on ties (bit1/bit2 are set only when open wins strictly), E beats F beats diag.
//   nibble = (Ht & 12) | (Ft & 2) | (Et & 1):
//     bits[3:2] = 10 -> bit3, 01 -> bit4, 00 -> diagonal;  bit 1 = !bit2;  bit 0 = !bit1.
// Scores stay exact because tags never exceed 15 and are stripped (& ~15) before a value is reused.
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
constexpr int kTagShift = 4;

TR_HD int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }
TR_HD int32_t imax3(int32_t a, int32_t b, int32_t c) { return imax(imax(a, b), c); }

// (a & mask) | (b & ~mask); v_bfi_b32 on the device (the generic form gets split into 3 ands + or3
// once the compiler sees only the low nibble is live)
TR_HD uint32_t bit_select(uint32_t mask, uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t d;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "v"(mask), "v"(a), "v"(b));
  return d;
#else
  return (a & mask) | (b & ~mask);
#endif
}

// shift a nibble into the top of a 32-bit accumulator: v_alignbit_b32
TR_HD uint32_t push_nibble(uint32_t acc, uint32_t nib_src) { return (acc >> 4) | (nib_src << 28); }

// ---- traceback lane -------------------------------------------------------------------------
template <int K>
struct TraceLane {
  int32_t Hc[K];   // H[r][c-1] * 16
  int32_t Ec[K];   // E[r][c-1] * 16
  int32_t cx1[K];  // horizontal open  constant of row r: hgap(go+ge)*16 + 8
  int32_t cx2[K];  // horizontal extend constant of row r: hgap(ge)*16 + 9
};

TR_HD int32_t trace_cx1(int32_t open_cost) { return open_cost * 16 + 8; }
TR_HD int32_t trace_cx2(int32_t ext_cost) { return ext_cost * 16 + 9; }
TR_HD int32_t trace_cy1(int32_t open_cost) { return open_cost * 16 + 4; }
TR_HD int32_t trace_cy2(int32_t ext_cost) { return ext_cost * 16 + 6; }

// One column of the strip.  up_h/up_f: H,F (x16, clean) of the row above at this column; diag: H of
// the row above at the previous column; sub(i) returns the substitution score of slot i x16.
// Outputs: w0/w1 = 16 nibbles (slot i at bits 4i of w1:w0), bot_h/bot_f = clean H,F of the last slot.
template <int K, class Sub>
TR_HD void trace_step(TraceLane<K>& s, int32_t up_h, int32_t up_f, int32_t diag, int32_t cy1, int32_t cy2,
                      const Sub& sub, uint32_t& w0, uint32_t& w1, int32_t& bot_h, int32_t& bot_f) {
  uint32_t a0 = 0, a1 = 0;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int32_t et = imax(s.Hc[i] + s.cx1[i], s.Ec[i] + s.cx2[i]);
    const int32_t ft = imax(up_h + cy1, up_f + cy2);
    const int32_t dt = diag + sub(i);
    const int32_t ht = imax3(dt, et, ft);
    const uint32_t nib = bit_select(3u, bit_select(1u, (uint32_t)et, (uint32_t)ft), (uint32_t)ht);
    if (i < 8) a0 = push_nibble(a0, nib);
    else a1 = push_nibble(a1, nib);
    diag = s.Hc[i];
    s.Hc[i] = ht & ~15;
    s.Ec[i] = et & ~15;
    up_h = s.Hc[i];
    up_f = ft & ~15;
  }
  if (K <= 8) a0 >>= (4 * (8 - K)) & 31;
  else a1 >>= (4 * (16 - K)) & 31;
  w0 = a0;
  w1 = a1;
  bot_h = up_h;
  bot_f = up_f;
}

// decode one stored nibble into the reference's four trace bits
struct TraceBits {
  bool bit1, bit2, bit3, bit4;
};
TR_HD TraceBits decode_nibble(uint32_t nib) {
  TraceBits b;
  b.bit3 = (nib >> 3) & 1u;
  b.bit4 = (nib >> 2) & 1u;
  b.bit2 = !((nib >> 1) & 1u);
  b.bit1 = !(nib & 1u);
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
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int32_t e = imax(s.Hl[i] + s.hopen[i], s.El[i] + s.hext[i]);
    const int32_t f = imax(up_h + vopen, up_f + vext);
    const int32_t h = imax3(diag + sub(i), e, f);
    diag = s.Hl[i];
    s.Hl[i] = h;
    s.El[i] = e;
    up_h = h;
    up_f = f;
  }
  bot_h = up_h;
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
TR_HD int32_t profile_score(const float a[5], const float b[5], float fmatch, float fmismatch) {
  float acc = 0.0f;
#pragma unroll
  for (int k1 = 0; k1 < 5; ++k1) {
#pragma unroll
    for (int k2 = 0; k2 < 5; ++k2) {
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
