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

namespace tracyhip {

enum : int { MODE_CHAR = 0, MODE_QP = 1, MODE_PROF = 2 };
enum : uint32_t { PAIR_A2_REVCOMP = 1u };  // read a2 reversed and complemented (profile.h:74-90)

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
};

struct DpArgs {
  const PairDesc* pairs;
  const void* a1;       // uint8_t* or float*
  const void* a2;       // MODE_QP: base codes 0..6 (dp_lane.h base_code), one byte per column
  uint64_t* bits;       // Gotoh traceback words
  uint32_t* bits32;     // NW traceback words
  int32_t* scratch;     // int2 per column: {H, F} of the last row of the previous pass
  int32_t* scores;      // may be null in traceback kernels
  int32_t* err;         // device error flags (bit 0: query-profile value does not fit int16)
  int32_t match, mismatch, go, ge;
  int32_t hfree, vfree;
};

// device error flags are OR-ed (several kernels share the word)
TR_HD void flag_error(int32_t* err, int32_t bits) {
  if (!err) return;
#if defined(__HIP_DEVICE_COMPILE__)
  atomicOr(err, bits);
#else
  *err |= bits;
#endif
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

template <int K>
struct SubTable {
  int32_t sv[K];
  TR_HD int32_t operator()(int i) const { return sv[i]; }
  TR_HD int32_t lo16(int i) const { return sv[i]; }
};

// LDS query-profile table: int16 [6 codes][64*K rows] (codes 5 = '-' and 6 = other share the zero row)
// table row stride per lane: K rounded up to an even count (int16 pairs), so odd strip heights work too
TR_HD constexpr int qp_stride(int K) { return (K + 1) & ~1; }

template <int K, bool NARROW = false>
TR_HD void qp_load(const int16_t* tab, uint32_t code, uint32_t lane, SubTable<K>& s) {
  constexpr int KP = qp_stride(K);
  const uint32_t row = (code < 5u ? code : 5u) * (64u * KP) + lane * KP;
  const uint32_t* p = reinterpret_cast<const uint32_t*>(tab + row);
#pragma unroll
  for (int j = 0; j < KP / 2; ++j) {
    const uint32_t w = p[j];
    s.sv[2 * j] = NARROW ? (int32_t)w : (int32_t)(int16_t)(w & 0xffffu);  // 16-bit consumers read the low half only
    if (2 * j + 1 < K) s.sv[2 * j + 1] = ((int32_t)w) >> 16;
  }
}

template <int K>
struct SubProf {
  const float* p1;  // LDS [5][64*K]
  uint32_t row0;    // lane * K
  float b[5];
  float fmatch, fmis;
  int shift;
  TR_HD int32_t operator()(int i) const {
    float a[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) a[k] = p1[k * (64 * K) + row0 + i];
    return (int32_t)((uint32_t)profile_score(a, b, fmatch, fmis) << shift);
  }
  TR_HD int32_t lo16(int i) const { return (*this)(i); }
};

// LDS bytes a (mode, K) kernel needs
TR_HD constexpr uint32_t lds_bytes(int mode, int K) {
  return mode == MODE_QP ? 6u * 64u * (uint32_t)qp_stride(K) * 2u : mode == MODE_PROF ? 5u * 64u * K * 4u : 0u;
}

TR_HD uint32_t a2_index(const PairDesc& d, uint32_t c /*1-based column*/) {
  return (d.flags & PAIR_A2_REVCOMP) ? d.n - c : c - 1u;
}

// ------------------------------------------------------------------------------------------------
// Gotoh, one pair per wave.  TRACE=true: tagged x16 arithmetic + traceback words; false: plain int32.
// ------------------------------------------------------------------------------------------------
template <class W, int K, int MODE, bool TRACE, bool NARROW = false>
TR_HD void gotoh_body(W& w, const DpArgs& a, uint32_t pair_idx) {
  static_assert(!(NARROW && TRACE), "the 16-bit formulation exists for the score-only kernel");
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

  const uint8_t* a1c = static_cast<const uint8_t*>(a.a1) + (MODE == MODE_CHAR ? d.a1_off : 0);
  const float* a1p = static_cast<const float*>(a.a1) + (MODE != MODE_CHAR ? d.a1_off : 0);
  const uint8_t* a2c = static_cast<const uint8_t*>(a.a2) + (MODE != MODE_PROF ? d.a2_off : 0);
  const float* a2p = static_cast<const float*>(a.a2) + (MODE == MODE_PROF ? d.a2_off : 0);
  int16_t* qp_tab = reinterpret_cast<int16_t*>(w.lds());
  float* p1_tab = reinterpret_cast<float*>(w.lds());
  const float fmatch = (float)a.match, fmis = (float)a.mismatch;

  const uint32_t P = num_passes(m, K);
  const uint32_t T = steps_per_pass(n);
  uint64_t* bits = TRACE ? a.bits + d.bits_off : nullptr;
  int32_t* scratch = a.scratch ? a.scratch + 2 * d.scratch_off : nullptr;
  const int32_t neg = NARROW ? kNegInf16 : (int32_t)((uint32_t)kNegInf << SH);

  for (uint32_t p = 0; p < P; ++p) {
    const uint32_t base = p * 64u * K;  // rows base+1 .. base+64K
    const uint32_t rows_here = (m - base < 64u * K) ? m - base : 64u * K;
    const uint32_t lanes_used = (rows_here + K - 1) / K;
    const uint32_t t_end = n + lanes_used - 1;
    const bool last_pass = (p + 1 == P);

    // ---- per-lane state at column 0 (gotoh.h:117-123) ----
    TraceLane<K> ts;
    ScoreLane<K> ss;
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const uint32_t r = base + L * K + i + 1;
      const bool hz = hfree && (r == m);
      const int32_t h0 = edge_value(vfree, go, ge, (int32_t)r);
      if (TRACE) {
        ts.Hc[i] = (int32_t)((uint32_t)h0 << SH);
        ts.Ec[i] = neg;
        ts.cx1[i] = trace_cx1(hz ? 0 : go + ge);
        ts.cx2[i] = trace_cx2(hz ? 0 : ge);
      } else {
        ss.Hl[i] = h0;
        ss.El[i] = neg;
        ss.hopen[i] = hz ? 0 : go + ge;
        ss.hext[i] = hz ? 0 : ge;
      }
    }
    const uint32_t row_above = base + L * K;
    int32_t prev_up_h = (row_above == 0) ? 0 : (int32_t)((uint32_t)edge_value(vfree, go, ge, (int32_t)row_above) << SH);
    int32_t bot_h = 0, bot_f = 0;

    // ---- substitution set-up for this pass ----
    SubChar<K> sub_c;
    SubTable<K> sub_t;
    SubProf<K> sub_p;
    if (MODE == MODE_CHAR) {
#pragma unroll
      for (int i = 0; i < K; ++i) {
        const uint32_t r = base + L * K + i + 1;
        sub_c.rc[i] = (r <= m) ? (int32_t)a1c[r - 1] : -1;
      }
      sub_c.vmatch = (int32_t)((uint32_t)a.match << SH);
      sub_c.vmis = (int32_t)((uint32_t)a.mismatch << SH);
      sub_c.cc = 0;
    } else if (MODE == MODE_QP) {
      w.sync();  // previous pass may still be reading the table
      bool overflow = false;
#pragma unroll
      for (int i = 0; i < K; ++i) {
        const uint32_t r = base + L * K + i + 1;
        float pr[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) pr[k] = (r <= m) ? a1p[(uint64_t)k * d.a1_stride + (r - 1)] : 0.0f;
#pragma unroll
        for (uint32_t b = 0; b < 5; ++b) {
          const int32_t q = (r <= m) ? onehot_score(pr, b, fmatch, fmis) : 0;
          const int32_t qs = (int32_t)((uint32_t)q << SH);
          overflow |= (qs > 32767) || (qs < -32768);
          qp_tab[b * (64 * qp_stride(K)) + L * qp_stride(K) + i] = (int16_t)qs;
        }
        qp_tab[5 * (64 * qp_stride(K)) + L * qp_stride(K) + i] = 0;
      }
      if (overflow) flag_error(a.err, 1);
      w.sync();
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
      sub_p.fmatch = fmatch;
      sub_p.fmis = fmis;
      sub_p.shift = SH;
    }

    // ---- anti-diagonal sweep ----
    for (uint32_t t = 1; t <= t_end; ++t) {
      const int32_t c = (int32_t)t - (int32_t)L;
      int32_t up_h = w.shift_up(bot_h);
      int32_t up_f = w.shift_up(bot_f);
      const bool active = (c >= 1) && (c <= (int32_t)n);
      if (active) {
        if (L == 0) {
          if (p == 0) {  // row 0 (gotoh.h:112-116)
            up_h = (int32_t)((uint32_t)edge_value(hfree, go, ge, c) << SH);
            up_f = neg;
          } else {  // last row of the previous pass
            up_h = scratch[2 * c];
            up_f = scratch[2 * c + 1];
          }
        }
        const bool vz = vfree && (c == (int32_t)n);
        const int32_t vopen = vz ? 0 : go + ge, vext = vz ? 0 : ge;
        const uint32_t ci = a2_index(d, (uint32_t)c);
        int32_t nb_h, nb_f;
        uint32_t w0 = 0, w1 = 0;
        if (MODE == MODE_CHAR) {
          sub_c.cc = (int32_t)a2c[ci];
          if (d.flags & PAIR_A2_REVCOMP) sub_c.cc = (int32_t)complement_char((uint8_t)sub_c.cc);
          if (TRACE) trace_step<K>(ts, up_h, up_f, prev_up_h, trace_cy1(vopen), trace_cy2(vext), sub_c, w0, w1, nb_h, nb_f);
          else if (NARROW) score_step16<K>(ss, up_h, up_f, prev_up_h, vopen, vext, sub_c, nb_h, nb_f);
          else score_step<K>(ss, up_h, up_f, prev_up_h, vopen, vext, sub_c, nb_h, nb_f);
        } else if (MODE == MODE_QP) {
          uint32_t code = a2c[ci];  // MODE_QP: a2 holds profile-row codes (encode_kernel), not characters
          if (d.flags & PAIR_A2_REVCOMP) code = complement_code(code);
          qp_load<K, NARROW>(qp_tab, code, L, sub_t);
          if (TRACE) trace_step<K>(ts, up_h, up_f, prev_up_h, trace_cy1(vopen), trace_cy2(vext), sub_t, w0, w1, nb_h, nb_f);
          else if (NARROW) score_step16<K>(ss, up_h, up_f, prev_up_h, vopen, vext, sub_t, nb_h, nb_f);
          else score_step<K>(ss, up_h, up_f, prev_up_h, vopen, vext, sub_t, nb_h, nb_f);
        } else {
#pragma unroll
          for (int k = 0; k < 5; ++k) sub_p.b[k] = a2p[(uint64_t)k * d.a2_stride + ci];
          if (TRACE) trace_step<K>(ts, up_h, up_f, prev_up_h, trace_cy1(vopen), trace_cy2(vext), sub_p, w0, w1, nb_h, nb_f);
          else score_step<K>(ss, up_h, up_f, prev_up_h, vopen, vext, sub_p, nb_h, nb_f);
        }
        prev_up_h = up_h;
        bot_h = nb_h;
        bot_f = nb_f;
        if (TRACE && L < lanes_used) bits[word_index(p, t, L, n)] = ((uint64_t)w1 << 32) | w0;
        if (!last_pass && L == 63) {
          scratch[2 * c] = bot_h;
          scratch[2 * c + 1] = bot_f;
        }
      }
    }
    (void)T;

    // ---- score = H[m][n] (gotoh.h:173) sits in the lane / slot that owns row m ----
    if (last_pass && a.scores) {
      const uint32_t g = m - 1 - base;
      if (L == g / K) {
        int32_t v = 0;
#pragma unroll
        for (int i = 0; i < K; ++i)
          if ((uint32_t)i == g % K) v = TRACE ? (ts.Hc[i] >> SH) : (NARROW ? sext16(ss.Hl[i]) : ss.Hl[i]);
        a.scores[d.out] = v;
      }
    }
    if (!last_pass) w.sync_global();  // scratch written by lane 63 is read by lane 0 in the next pass
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
// Wave-cooperative traceback walker: the same state machine (gotoh.h:143-167), but one wave per pair
// and up to 64 cells per memory round trip.  In state 's' the walk continues diagonally while a cell
// has neither bit3 nor bit4; in 'h' ('v') it continues left (up) until the first cell with bit1 (bit2).
// Each lane fetches the nibble of one candidate cell of the current run, a ballot finds where the run
// ends, and the run is emitted with one coalesced store.  W provides lane(), ballot(pred), bcast(x, l).
// ------------------------------------------------------------------------------------------------
template <class W>
TR_HD void gotoh_walk_wave(W& w, const WalkArgs& a, uint32_t pair_idx) {
  const PairDesc d = a.pairs[pair_idx];
  const uint64_t* bits = a.bits + d.bits_off;
  uint8_t* out = a.ops + a.ops_off[d.out];
  const uint32_t n = d.n, lane = w.lane();
  const int K = a.K;
  uint32_t row = d.m, col = d.n, k = 0;
  int state = 0;  // 0 = s, 1 = h, 2 = v
  const uint32_t limit = d.m + d.n;
  bool bad = false;
  while ((row > 0 || col > 0) && k <= limit) {
    if (row == 0) {  // first row: bit3 everywhere, bit1 nowhere -> 'h' down to column 0 (gotoh.h:112-116)
      for (uint32_t i = lane; i < col; i += 64) out[k + i] = 'h';
      k += col; col = 0;
      break;
    }
    if (col == 0) {  // first column: bit4 only -> 'v' down to row 0 (gotoh.h:117-123)
      if (state == 1) { bad = true; break; }
      for (uint32_t i = lane; i < row; i += 64) out[k + i] = 'v';
      k += row; row = 0;
      break;
    }
    // candidate cell of this lane along the current run
    const uint32_t r = (state == 1) ? row : row - lane;
    const uint32_t c = (state == 2) ? col : col - lane;
    const bool inside = (state == 1) ? (lane < col) : (state == 2) ? (lane < row) : (lane < row && lane < col);
    TraceBits b = {false, false, false, false};
    if (inside) {
      const CellAddr ca = cell_addr(r, K);
      const uint64_t wd = bits[word_index(ca.pass, c + ca.lane, ca.lane, n)];
      b = decode_nibble((uint32_t)(wd >> (4u * ca.slot)) & 15u);
    }
    const bool stop = !inside || (state == 0 ? (b.bit3 || b.bit4) : state == 1 ? b.bit1 : b.bit2);
    const uint64_t m = w.ballot(stop);
    uint32_t first = 64;
    if (m) {
      first = 0;
      while (!((m >> first) & 1ull)) ++first;
    }
    if (state == 0) {
      // cells 0..first-1 are diagonal steps
      if (lane < first) out[k + lane] = 's';
      k += first; row -= first; col -= first;
      if (first < 64 && row > 0 && col > 0) {  // the stopping cell is inside: switch matrix, no move
        const uint32_t code = w.bcast((uint32_t)(b.bit3 ? 1 : 2), first);
        state = (int)code;
      }
      // first < 64 with row == 0 or col == 0: handled by the boundary rules on the next iteration
    } else if (state == 1) {
      // every visited cell emits 'h'; the one with bit1 is the last of the run
      const uint32_t inside_n = col < 64 ? col : 64;
      const bool hit = first < inside_n;
      const uint32_t cnt = hit ? first + 1 : inside_n;
      if (lane < cnt) out[k + lane] = 'h';
      k += cnt; col -= cnt;
      if (hit) state = 0;
    } else {
      const uint32_t inside_n = row < 64 ? row : 64;
      const bool hit = first < inside_n;
      const uint32_t cnt = hit ? first + 1 : inside_n;
      if (lane < cnt) out[k + lane] = 'v';
      k += cnt; row -= cnt;
      if (hit) state = 0;
      else if (row == 0 && col > 0) { bad = true; break; }  // 'v' ran into row 0: unreachable with sane parameters
    }
  }
  if (row > 0 || col > 0 || bad) {
    if (lane == 0) flag_error(a.err, 2);
  }
  if (lane == 0) a.ops_len[d.out] = k;
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
