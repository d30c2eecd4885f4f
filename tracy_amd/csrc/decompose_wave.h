// decompose_wave.h -- decomposeAlleles (decompose.h:179-376) as ONE wave per trace with its working set in LDS.
//
// The phase functions of decompose_kernels.h walk the alignment rows and the basecalls in global memory, a byte at a time and one
// dependent access after the other (rocprofv3, 100 000 traces: 435 vector loads per wave, 46 % of the wave cycles in s_waitcnt, 47 k
// instructions per trace of which 16 k build the basecall bit sets through six calls of phaseRefAllele per basecall; 21 KB of LDS =
// seven waves per CU).  This body does the same arithmetic on a staged copy:
//   * the two basecall strings and the span of the alignment's reference row the algorithm can look at (under the trace bases it phases,
//     and as far right of the breakpoint as the widest shift scan reaches) are copied to LDS once (coalesced loads); of the trace row only
//     the gap / no-gap bit per column is kept (ballots) -- nothing else of it is ever read (decompose.h:184-208, 327-343);
//   * the walk to the breakpoint is a prefix sum of popcounts + one select-nth-bit; the basecall a column phases is its rank;
//   * the class bit sets of the shift scans come from ballots: the six "incompatible with class c" bits of a basecall are ONE look-up
//     in a 7 x 12 table over (primary class, secondary class) built from ref_compatible itself (decomp_lut_build);
//   * the scans run on 32-bit words with v_alignbit_b32 and carry the upper word of one step into the next;
//   * median and MAD (decompose.h:129-145) by a counting search over the value range (wave sums) instead of a histogram in LDS;
//   * every rewritten basecall goes to LDS and, write-through, to the caller's arrays.
// LDS per trace of 1000 basecalls: 11 KB (fourteen waves per CU).  Everything is integer / byte work and bit-identical with the phase
// functions, which stay as they are: the host emulator compares the two, and traces this body is not provisioned for (alignment or
// basecalls longer than the launch's LDS holds, trims outside the trace) are handed to decompose_kernel through a to-do word.
//
// W: the wave abstraction of dp_kernels.h (lane, ballot, bcast, sync, lds) + sum / umin / umax / excl_sum over the 64 lanes.
#ifndef TRACY_AMD_DECOMPOSE_WAVE_H
#define TRACY_AMD_DECOMPOSE_WAVE_H

#include <type_traits>

#include "decompose_kernels.h"

namespace tracyhip {

// ---- (primary class, secondary class) -> the classes A C G T N '-' the basecall is NOT compatible with ----
// ref_compatible(p, s, r) looks at p only through p == r and C / G / T / other (iupac2), at s through s == 'N', s == r and the six
// two-base IUPAC letters: seven classes of p and twelve of s decide it for every r of the six reference classes.
constexpr int kLutP = 7, kLutS = 12, kLutBytes = (kLutP * kLutS + 3) & ~3;
TR_HD uint32_t lut_pclass(uint8_t p) { return p == 'A' ? 0u : p == 'C' ? 1u : p == 'G' ? 2u : p == 'T' ? 3u : p == 'N' ? 4u : p == '-' ? 5u : 6u; }
TR_HD uint32_t lut_sclass(uint8_t s) {
  return s == 'A' ? 0u : s == 'C' ? 1u : s == 'G' ? 2u : s == 'T' ? 3u : s == 'N' ? 4u : s == 'R' ? 5u : s == 'Y' ? 6u : s == 'S' ? 7u :
         s == 'W' ? 8u : s == 'K' ? 9u : s == 'M' ? 10u : 11u;
}
inline void decomp_lut_build(uint8_t* lut) {  // host: kLutBytes bytes
  const char prep[kLutP + 1] = "ACGTN-X", srep[kLutS + 1] = "ACGTNRYSWKMX";
  for (int i = 0; i < kLutBytes; ++i) lut[i] = 0;
  for (int p = 0; p < kLutP; ++p)
    for (int s = 0; s < kLutS; ++s) {
      uint8_t m = 0;
      for (int c = 0; c < kRefClasses; ++c)
        if (!ref_compatible(prep[p], srep[s], class_char(c))) m |= (uint8_t)(1u << c);
      lut[p * kLutS + s] = m;
    }
}

// The alignment may span a whole reference window (`tracy decompose` aligns the trimmed trace to 3 kb of reference: 2 kb of leading and
// trailing gap columns), but the body only ever looks at the reference row (i) under the trace bases it phases and (ii) in the columns
// the shift scans can reach, alignIndex + 1 .. + NV + the largest deletion: that span is what is staged.
constexpr uint32_t kDecompWaveMaxL = 4096;  // alignment columns (one 64-bit gap mask per lane)
struct DecompWaveCaps {
  uint32_t capL;  // reference-row bytes staged (multiple of 64)
  uint32_t capB;  // basecalls staged (multiple of 64, <= 2048)
  uint32_t capI;  // deletion shifts (>= maxindel)
  uint32_t capF;  // insertion shifts (<= capI; maxins / 2 <= basecalls / 2 bounds them)
};
// byte offsets into the workgroup's dynamic LDS
struct DecompWaveLayout {
  uint32_t row1, pri, sec, ng0, pre0, fref, fins, is, bad, lut, tmp, total;
  uint32_t is_stride, bad_stride;  // 64-bit words per class: is = 1 zero word + (capB + capI)/64 + 2 zero words; bad = capB/64 + 2
};
TR_HD DecompWaveLayout decomp_wave_layout(const DecompWaveCaps& c) {
  DecompWaveLayout l{};
  uint32_t o = 0;
  auto take = [&](uint32_t bytes) { const uint32_t at = o; o += (bytes + 7u) & ~7u; return at; };
  l.row1 = take(c.capL + 8);
  l.pri = take(c.capB + 8);
  l.sec = take(c.capB + 8);
  l.ng0 = take(kDecompWaveMaxL / 8);
  l.pre0 = take(64 * 4);
  l.fref = take(c.capI * 2 + 8);
  l.fins = take(c.capF * 2 + 8);
  l.is_stride = 1u + (c.capB + c.capI + 63u) / 64u + 2u;  // the scans reach NV + the largest deletion reference columns
  l.bad_stride = c.capB / 64u + 2u;
  l.is = take(kRefClasses * l.is_stride * 8u);
  l.bad = take(kRefClasses * l.bad_stride * 8u);
  l.lut = take(kLutBytes);
  l.tmp = take(2u * 4u * (kDecompWaveMaxL / 256u) * 8u);  // gap-bit ballots of the two rows in lane order
  l.total = o;
  return l;
}
TR_HD bool decomp_wave_caps_ok(const DecompWaveCaps& c) {
  return c.capL && c.capB && c.capI && c.capF && c.capL % 64 == 0 && c.capB % 64 == 0 && c.capL <= 8192 && c.capB <= 2048 && c.capI <= 1024 && c.capF <= c.capI;  // (=> a class set has at most 51 words: one per lane)
}

TR_HD uint32_t popc32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__popc(x);
#else
  return (uint32_t)__builtin_popcount(x);
#endif
}
TR_HD uint32_t ctz64(uint64_t x) {  // x != 0
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)(__ffsll((long long)x) - 1);
#else
  return (uint32_t)__builtin_ctzll(x);
#endif
}
TR_HD uint32_t funnel32(uint32_t hi, uint32_t lo, uint32_t sh) {  // bits sh .. sh + 31 of hi:lo, sh < 32
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
  return sh ? (lo >> sh) | (hi << (32u - sh)) : lo;
#endif
}
TR_HD uint32_t low32(int32_t nbits) { return nbits >= 32 ? ~0u : nbits <= 0 ? 0u : ((1u << nbits) - 1u); }

// the bit sets as 32-bit words: position q of class c's reference set is bit q & 31 of is32[c * is_stride32 + 2 + (q >> 5)] (two zero
// words in front: a word of a diagonal may begin up to 31 columns left of the window), basecall s bit s & 31 of bad32[c * bad_stride32 + (s >> 5)]
struct DecompSets {
  const uint32_t* is32;
  const uint32_t* bad32;
  uint32_t is_stride32, bad_stride32;
  uint32_t classes;  // bit c: class c occurs in the window
  int32_t Lw, NV;
};
// 32 positions s = 32 wq .. 32 wq + 31 of diagonal u: OR_c bad_c[s] & is_c[s + u]
TR_HD uint32_t diag_word32(const DecompSets& z, int32_t u, int32_t wq) {
  const int32_t a = 32 * wq + u, ai = a >> 5;  // (floor)
  const uint32_t sh = (uint32_t)a & 31u;
  uint32_t r = 0;
#pragma unroll
  for (int c = 0; c < kRefClasses; ++c) {
    const uint32_t* p = z.is32 + (uint32_t)c * z.is_stride32 + 2 + ai;  // (a class that does not occur: zero words)
    r |= funnel32(p[1], p[0], sh) & z.bad32[(uint32_t)c * z.bad_stride32 + (uint32_t)wq];
  }
  return r;
}
// The scans' inner loops carry no branch on the classes that occur: a class that does not occur has zero words, and the one class that is
// rare in a reference row -- N -- is a template parameter (WITH_N: class 4 takes part).  (A wave-uniform `continue` per class was a scalar
// compare and branch per class, word and diagonal, and a basic-block boundary the LDS reads could not be moved across.)
template <bool WITH_N>
struct ScanClasses {
  static constexpr int n = WITH_N ? kRefClasses : kRefClasses - 1;
  TR_HD static constexpr int cls(int i) { return WITH_N ? i : (i < 4 ? i : 5); }
};

// failed(del, ins) for del - ins == u and from == ins (decompose.h:215-222 and its two copies): positions s in [from, min(NV, Lw - u))
template <bool WITH_N>
TR_HD int32_t diag_count32_t(const DecompSets& z, int32_t u, int32_t from) {
  using SC = ScanClasses<WITH_N>;
  const int32_t lim_ref = z.Lw - u;
  const int32_t limit = z.NV < lim_ref ? z.NV : lim_ref;
  if (limit <= from) return 0;
  const uint32_t sh = (uint32_t)u & 31u;
  const int32_t w0 = from >> 5;
  uint32_t lo[SC::n];
#pragma unroll
  for (int i = 0; i < SC::n; ++i) lo[i] = z.is32[(uint32_t)SC::cls(i) * z.is_stride32 + 2 + ((32 * w0 + u) >> 5)];
  int32_t f = 0;
  // four words per pass, all of the pass's LDS words requested before the first is used (a word per pass waited for an LDS round trip 16
  // times per diagonal).  Words behind the limit are read and masked away: they lie inside the sets' own LDS (two zero words behind every
  // class, the next class or the table behind the last).  Passes that lie inside [from, limit) altogether need no mask.
  constexpr int kPass = 4;
  auto pass = [&](int32_t wq, auto masked) {
    const int32_t ai = (32 * wq + u) >> 5;
    uint32_t hi[SC::n][kPass], bd[SC::n][kPass];
#pragma unroll
    for (int i = 0; i < SC::n; ++i) {
#pragma unroll
      for (int k = 0; k < kPass; ++k) {
        hi[i][k] = z.is32[(uint32_t)SC::cls(i) * z.is_stride32 + 2 + ai + 1 + k];
        bd[i][k] = z.bad32[(uint32_t)SC::cls(i) * z.bad_stride32 + (uint32_t)(wq + k)];
      }
    }
#pragma unroll
    for (int k = 0; k < kPass; ++k) {
      uint32_t r = 0;
#pragma unroll
      for (int i = 0; i < SC::n; ++i) {
        r |= funnel32(hi[i][k], lo[i], sh) & bd[i][k];
        lo[i] = hi[i][k];
      }
      if (decltype(masked)::value) {
        r &= low32(limit - 32 * (wq + k));
        if (wq + k == w0) r &= ~low32(from - 32 * w0);
      }
      f += (int32_t)popc32(r);
    }
  };
  int32_t wq = w0;
  if (32 * wq < limit) { pass(wq, std::true_type{}); wq += kPass; }  // (the pass that holds `from`)
  for (; 32 * (wq + kPass) <= limit; wq += kPass) pass(wq, std::false_type{});
  if (32 * wq < limit) pass(wq, std::true_type{});
  return f;
}
TR_HD int32_t diag_count32(const DecompSets& z, int32_t u, int32_t from) {
  return ((z.classes >> 4) & 1u) ? diag_count32_t<true>(z, u, from) : diag_count32_t<false>(z, u, from);  // (wave-uniform)
}

// failed(del, 0) for ND diagonals del = u0, u0 + 64, ... of one lane at once (the deletion scans, decompose.h:215-222): the words of the
// basecall sets are the same for every diagonal and the reference words of neighbouring diagonals overlap (diagonal + 64 = two words on),
// so a pass of four words reads 2 ND + 2 reference words and 4 basecall words per class instead of 8 ND.  on[j] = 0: diagonal j is not
// wanted (its count is not defined); diagonal 0 is.  (Words read for an unwanted diagonal or behind a limit lie at most six words behind the
// class's own: in the next class's words or, behind the last class, in the basecall sets.)
template <int ND, bool WITH_N>
TR_HD void diag_count32_multi_t(const DecompSets& z, int32_t u0, const bool on[ND], int32_t f[ND]) {
  using SC = ScanClasses<WITH_N>;
  constexpr int kPass = 4, kWords = 2 * (ND - 1) + kPass;
  int32_t limit[ND], minlim = 0x7fffffff;
#pragma unroll
  for (int j = 0; j < ND; ++j) {
    const int32_t lim_ref = z.Lw - (u0 + 64 * j);
    limit[j] = on[j] ? (z.NV < lim_ref ? z.NV : lim_ref) : 0;
    if (on[j]) minlim = limit[j] < minlim ? limit[j] : minlim;
    f[j] = 0;
  }
  if (limit[0] <= 0) return;  // (the later diagonals end earlier)
  const uint32_t sh = (uint32_t)u0 & 31u;
  const int32_t a0 = u0 >> 5;
  uint32_t lo[SC::n][ND];
#pragma unroll
  for (int i = 0; i < SC::n; ++i)
#pragma unroll
    for (int j = 0; j < ND; ++j) lo[i][j] = z.is32[(uint32_t)SC::cls(i) * z.is_stride32 + 2 + a0 + 2 * j];
  auto pass = [&](int32_t wq, auto masked) {
    uint32_t x[SC::n][kWords], bd[SC::n][kPass];
#pragma unroll
    for (int i = 0; i < SC::n; ++i) {
#pragma unroll
      for (int q = 0; q < kWords; ++q) x[i][q] = z.is32[(uint32_t)SC::cls(i) * z.is_stride32 + 2 + a0 + wq + 1 + q];
#pragma unroll
      for (int k = 0; k < kPass; ++k) bd[i][k] = z.bad32[(uint32_t)SC::cls(i) * z.bad_stride32 + (uint32_t)(wq + k)];
    }
#pragma unroll
    for (int j = 0; j < ND; ++j) {
#pragma unroll
      for (int k = 0; k < kPass; ++k) {
        uint32_t r = 0;
#pragma unroll
        for (int i = 0; i < SC::n; ++i) r |= funnel32(x[i][2 * j + k], k ? x[i][2 * j + k - 1] : lo[i][j], sh) & bd[i][k];
        if (decltype(masked)::value) r &= low32(limit[j] - 32 * (wq + k));
        f[j] += (int32_t)popc32(r);
      }
#pragma unroll
      for (int i = 0; i < SC::n; ++i) lo[i][j] = x[i][2 * j + kPass - 1];
    }
  };
  int32_t wq = 0;
  for (; 32 * (wq + kPass) <= minlim; wq += kPass) pass(wq, std::false_type{});  // (inside every wanted diagonal's range: no mask; an unwanted one's count is not read)
  for (; 32 * wq < limit[0]; wq += kPass) pass(wq, std::true_type{});
}
template <int ND>
TR_HD void diag_count32_multi(const DecompSets& z, int32_t u0, const bool on[ND], int32_t f[ND]) {
  if ((z.classes >> 4) & 1u) diag_count32_multi_t<ND, true>(z, u0, on, f);  // (wave-uniform)
  else diag_count32_multi_t<ND, false>(z, u0, on, f);
}
// Z_u word of 64 positions (decompose_kernels.h diag_word), cut at s < limit
TR_HD uint64_t diag_word64(const DecompSets& z, int32_t u, int32_t w, int32_t limit) {
  const uint64_t v = (uint64_t)diag_word32(z, u, 2 * w) | ((uint64_t)diag_word32(z, u, 2 * w + 1) << 32);
  return v & low_mask(limit - 64 * w);
}

// value at sorted position n / 2 of n values v(i) (getMedian, decompose.h:129-135): the smallest x with #{v <= x} > n / 2.
// get(i): value i (>= 0); the lanes take i = lane, lane + 64, ...
template <class W, class Get>
TR_HD int32_t wave_median(W& w, uint32_t n, Get get) {
  const uint32_t lane = w.lane();
  uint32_t mx = 0;
  for (uint32_t i = lane; i < n; i += 64) { const uint32_t v = (uint32_t)get(i); mx = v > mx ? v : mx; }
  uint32_t lo = 0, hi = w.umax(mx);
  const uint32_t half = n / 2;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    uint32_t c = 0;
    for (uint32_t i = lane; i < n; i += 64) c += (uint32_t)get(i) <= mid ? 1u : 0u;
    if (w.sum(c) > half) hi = mid; else lo = mid + 1;
  }
  return (int32_t)lo;
}

struct DecompWaveArgs {
  DecompArgs a;
  const BreakpointOut* bps;
  const uint8_t* lut;    // decomp_lut_build
  uint32_t* todo;        // per trace: 1 = left to decompose_kernel (not provisioned here), 0 = done or skipped
  DecompWaveCaps caps;
  unsigned long long* clocks;  // development: cycles per stage, summed over the traces (or null)
};
constexpr int kDecompWaveStages = 16;
constexpr uint32_t kDecompWaveClockRows = 1024;  // (development: a row of stage counters per trace & 1023 -- one shared row serialises the atomics)
#ifndef TRACY_DW_STOP_AFTER
#define TRACY_DW_STOP_AFTER 99
#endif
#ifdef TRACY_DW_NO_ATOMICS
constexpr bool kDwAtomics = false;
#else
constexpr bool kDwAtomics = true;
#endif

template <class W>
TR_HD void decomp_wave_body(W& w, const DecompWaveArgs& wa, uint32_t t) {
  const DecompArgs& a = wa.a;
  const uint32_t lane = w.lane();
  // (the trace's words and the lane's word of the table are requested together: a look at the skip word first, then at the descriptor, then
  // at the rows was three round trips to memory before the first useful one)
  const uint32_t skip_word = a.skip ? a.skip[t] : 0u;
  DecompDesc d = a.desc[t];
  d.breakpoint = wa.bps[t].breakpoint;
  if (a.lens) d.L = a.lens[t];
  const uint32_t lut_word = lane < (uint32_t)kLutBytes / 4u ? reinterpret_cast<const uint32_t*>(wa.lut)[lane] : 0u;
  if (skip_word) { if (lane == 0) wa.todo[t] = 0; return; }
  const uint32_t L = d.L, nbc = d.nbc;
  const uint32_t ltrim = (uint32_t)a.prm.trimLeft;
  const int32_t rtrim = a.prm.trimRight;
  const uint32_t mi = (uint32_t)a.prm.maxindel;
  const DecompWaveCaps caps = wa.caps;
  const uint32_t bp = d.breakpoint + ltrim;
  const uint32_t maxins = (uint32_t)((int32_t)nbc - (int32_t)((uint32_t)rtrim + bp));
  const uint32_t nfins = [&] { const uint32_t x = mi < maxins / 2 ? mi : maxins / 2; return x > 1u ? x : 1u; }();  // fins[0], then ins = 1 .. while ins < maxindel && ins < maxins / 2
  if (!(L <= kDecompWaveMaxL && nbc <= caps.capB && mi <= caps.capI && nfins <= caps.capF && rtrim >= 0 && (uint32_t)rtrim <= nbc &&
        a.prm.trimLeft >= 0 && ltrim <= nbc)) {
    if (lane == 0) wa.todo[t] = 1;
    return;
  }
#if defined(__HIP_DEVICE_COMPILE__) && defined(TRACY_PHASE_CLOCKS)
  unsigned long long clk_prev = __builtin_readcyclecounter();
  int clk_stage = 0;
#define DW_CLOCK()                                                                                          \
  do {                                                                                                      \
    const unsigned long long now_ = __builtin_readcyclecounter();                                           \
    if (kDwAtomics && wa.clocks && lane == 0) atomicAdd(wa.clocks + (size_t)(t & (kDecompWaveClockRows - 1u)) * kDecompWaveStages + clk_stage, now_ - clk_prev); \
    if (clk_stage == TRACY_DW_STOP_AFTER) { if (lane == 0) wa.todo[t] = 0; return; }                          \
    clk_prev = now_;                                                                                        \
    ++clk_stage;                                                                                            \
  } while (0)
#else
#define DW_CLOCK() do { } while (0)
#endif
  const DecompWaveLayout lay = decomp_wave_layout(caps);
  char* lds = w.lds();
  uint8_t* lrow1 = reinterpret_cast<uint8_t*>(lds + lay.row1);
  uint8_t* lpri = reinterpret_cast<uint8_t*>(lds + lay.pri);
  uint8_t* lsec = reinterpret_cast<uint8_t*>(lds + lay.sec);
  uint64_t* ng0 = reinterpret_cast<uint64_t*>(lds + lay.ng0);
  uint32_t* lpre0 = reinterpret_cast<uint32_t*>(lds + lay.pre0);
  uint16_t* fref = reinterpret_cast<uint16_t*>(lds + lay.fref);
  uint16_t* fins = reinterpret_cast<uint16_t*>(lds + lay.fins);
  uint64_t* is64 = reinterpret_cast<uint64_t*>(lds + lay.is);
  uint64_t* bad64 = reinterpret_cast<uint64_t*>(lds + lay.bad);
  uint8_t* llut = reinterpret_cast<uint8_t*>(lds + lay.lut);
  const uint8_t* g0 = a.rows0 + d.rows_off;
  const uint8_t* g1 = a.rows1 + d.rows_off;
  uint8_t* gpri = a.primary + d.bc_off;
  uint8_t* gsec = a.secondary + d.bc_off;

  // ---- 1. gap bits of both rows over the whole alignment, basecalls, table: ONE round trip to memory ----
  // A lane loads four consecutive columns (an unaligned dword) of each chunk of 256: the rows are in registers after a single wait.
  // Four ballots per chunk give the gap bits in lane order -- bit l of ballot k is column 256 i + 4 l + k --, which go through LDS and are
  // put into column order by the lane that owns the word (bits of four 16-bit fields interleaved: spread16).
  constexpr uint32_t kChunks = kDecompWaveMaxL / 256u;
  const uint32_t nwL = (L + 63u) >> 6, nch = (L + 255u) >> 8;
  uint32_t v0[kChunks], v1[kChunks];
  // bytes c .. c + 3 of g[0, n), `fill` beyond n: the dword that ends at n, shifted, for the lanes over the end -- no branch, no second wait
  // (n < 4, wave-uniform: byte by byte)
  auto load4 = [&](const uint8_t* g, uint32_t c, uint32_t n, uint32_t fill) -> uint32_t {
    uint32_t v = fill;
    if (n >= 4u) {
      const uint32_t cc = c + 4u <= n ? c : n - 4u, sh = c - cc;
      uint32_t x;
      __builtin_memcpy(&x, g + cc, 4);
      v = sh >= 4u ? fill : (uint32_t)((((uint64_t)fill << 32) | x) >> (8u * (sh & 3u)));
    } else {
      for (uint32_t k = 0; k < 4; ++k)
        if (c + k < n) v = (v & ~(0xffu << (8u * k))) | ((uint32_t)g[c + k] << (8u * k));
    }
    return v;
  };
#pragma unroll
  for (uint32_t i = 0; i < kChunks; ++i) {
    v0[i] = 0x2d2d2d2du; v1[i] = 0x2d2d2d2du;
    if (i < nch) { v0[i] = load4(g0, 256u * i + 4u * lane, L, 0x2d2d2d2du); v1[i] = load4(g1, 256u * i + 4u * lane, L, 0x2d2d2d2du); }
  }
  {
    constexpr uint32_t kBcChunks = 2048u / 256u;
    uint32_t vp[kBcChunks], vs[kBcChunks];
#pragma unroll
    for (uint32_t i = 0; i < kBcChunks; ++i) {
      vp[i] = 0; vs[i] = 0;
      if (256u * i < nbc) { vp[i] = load4(gpri, 256u * i + 4u * lane, nbc, 0u); vs[i] = load4(gsec, 256u * i + 4u * lane, nbc, 0u); }
    }
#pragma unroll
    for (uint32_t i = 0; i < kBcChunks; ++i)
      if (256u * i < nbc && 256u * i + 4u * lane < caps.capB) {
        reinterpret_cast<uint32_t*>(lpri)[64u * i + lane] = vp[i];
        reinterpret_cast<uint32_t*>(lsec)[64u * i + lane] = vs[i];
      }
  }
  uint64_t* tmp0 = reinterpret_cast<uint64_t*>(lds + lay.tmp);  // [chunk][k]: ballot k of the chunk, trace row
  uint64_t* tmp1 = tmp0 + 4u * kChunks;                          // ... reference row
#pragma unroll
  for (uint32_t i = 0; i < kChunks; ++i) {
    if (i < nch) {  // (wave-uniform)
#pragma unroll
      for (uint32_t k = 0; k < 4; ++k) {
        const uint64_t q0 = w.ballot(((v0[i] >> (8u * k)) & 0xffu) != (uint32_t)'-'), q1 = w.ballot(((v1[i] >> (8u * k)) & 0xffu) != (uint32_t)'-');
        if (lane == k) { tmp0[4u * i + k] = q0; tmp1[4u * i + k] = q1; }
      }
    }
  }
  DW_CLOCK();  // loads + ballots
  w.sync();
  uint32_t c0 = 0, c1 = 0;    // lane w: bases of word w (columns 64 w .. 64 w + 63) of the trace row / the reference row
  uint64_t m0w = 0, m1w = 0;  // ... and the words themselves
  if (lane < nwL) {
    auto spread16 = [](uint64_t x) -> uint64_t {  // bit b of the low 16 -> bit 4 b
      x = (x | (x << 24)) & 0x000000FF000000FFull;
      x = (x | (x << 12)) & 0x000F000F000F000Full;
      x = (x | (x << 6)) & 0x0303030303030303ull;
      x = (x | (x << 3)) & 0x1111111111111111ull;
      return x;
    };
    const uint32_t ci = lane >> 2, sh = 16u * (lane & 3u);
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
      m0w |= spread16((tmp0[4u * ci + k] >> sh) & 0xffffull) << k;
      m1w |= spread16((tmp1[4u * ci + k] >> sh) & 0xffffull) << k;
    }
    ng0[lane] = m0w;
    c0 = (uint32_t)popc64(m0w);
    c1 = (uint32_t)popc64(m1w);
  }
  if (lane < (uint32_t)kLutBytes / 4u) reinterpret_cast<uint32_t*>(llut)[lane] = lut_word;
  const uint32_t pre0 = w.excl_sum(c0), pre1 = w.excl_sum(c1);
  const uint32_t total0 = w.sum(c0), total1 = w.sum(c1);
  DW_CLOCK();  // words, prefix sums

  // ---- 2. walk to the breakpoint (decompose.h:184-208): the column of trace base number `stop` ----
  const uint32_t stop = d.breakpoint;  // (bp - ltrim with bp = breakpoint + ltrim)
  const bool found = stop != 0 && stop <= total0;
  uint32_t alignIndex = 0, varIndex = 0, refPointer = total1;
  if (found) {
    const bool mine = pre0 < stop && stop <= pre0 + c0;  // exactly one lane
    uint32_t col = 0, refp = 0;
    if (mine) {
      uint64_t m = m0w;
      for (uint32_t i = 1; i < stop - pre0; ++i) m &= m - 1;
      const uint32_t bit = ctz64(m);
      col = 64u * lane + bit;
      refp = pre1 + (uint32_t)popc64(m1w & low_mask((int32_t)bit));
    }
    const uint32_t src = ctz64(w.ballot(mine));
    alignIndex = w.bcast(col, src);
    refPointer = w.bcast(refp, src);
    varIndex = ltrim + stop;
  }
  // scan bounds (decompose.h:210-213, 248-250)
  uint32_t maxdel = 2;
  if ((uint64_t)d.refslice_len > (uint64_t)(uint32_t)(refPointer + (uint32_t)rtrim + 2u))
    maxdel = (uint32_t)((uint64_t)d.refslice_len - (uint64_t)(uint32_t)(refPointer + (uint32_t)rtrim));
  const uint32_t nfref = mi < maxdel / 2 ? mi : maxdel / 2;  // del = 0 .. while del < maxindel && del < maxdel / 2
  const uint32_t winstart = alignIndex + 1u;
  const uint64_t vend = (uint64_t)nbc - (uint64_t)(int64_t)rtrim;  // (<= nbc: 0 <= rtrim <= nbc)
  DecompSets z{};
  {
    const int64_t nv = (int64_t)vend - (int64_t)varIndex;
    z.NV = (int32_t)(nv < 0 ? 0 : nv);
    // columns from alignIndex + 1 on; a scan pairs basecall s < NV with column s + del, del < nfref: later columns are never looked at
    const int64_t lw = (int64_t)L - (int64_t)winstart, reach = (int64_t)z.NV + (int64_t)nfref;
    z.Lw = (int32_t)(lw < 0 ? 0 : lw < reach ? lw : reach);
  }
  // The reference row is read under the trace bases that are phased -- the first base .. alignIndex, or all of them when the walk never
  // stops or nothing is picked ("traverse the whole alignment", :327-343) -- and in the window: ONE span [lo1, hi1) of it is staged.
  uint32_t lo1, hi1;
  {
    const uint32_t first0 = w.umin(c0 ? 64u * lane + ctz64(m0w) : 0xffffffffu);                        // first column with a trace base
    const uint32_t last0 = w.umax(c0 ? 64u * lane + 64u - (uint32_t)__builtin_clzll(m0w) : 0u);       // last such column + 1
    const uint32_t w_lo = winstart < L ? winstart : L, w_hi = w_lo + (uint32_t)z.Lw;
    lo1 = w_lo; hi1 = w_hi;
    if (last0) { lo1 = first0 < lo1 ? first0 : lo1; hi1 = last0 > hi1 ? last0 : hi1; }
    lo1 &= ~3u;  // (whole dwords)
  }
  if (hi1 - lo1 + 4u > caps.capL || ltrim + total0 > nbc) {  // (or a trace row with more bases than basecalls behind the trim: not this body's business)
    if (lane == 0) wa.todo[t] = 1;
    return;
  }
  DW_CLOCK();  // walk, bounds, span
  if (lane == 0) wa.todo[t] = 0;
  lpre0[lane] = pre0;
#pragma unroll
  for (uint32_t i = 0; i < kChunks; ++i) {  // the staged span of the reference row, from the registers that hold the row
    const uint32_t c = 256u * i + 4u * lane;
    if (i < nch && c >= lo1 && c < hi1) reinterpret_cast<uint32_t*>(lrow1)[(c - lo1) >> 2] = v1[i];
  }
  const uint8_t* row1v = lrow1 - lo1;  // row1v[j] for j in [lo1, hi1)
  w.sync();
  DW_CLOCK();

  auto phase_pos = [&](uint32_t vi, uint8_t r) {  // decompose.h:196-203
    const uint8_t p = lpri[vi];
    if (r != p) {
      const char s = phase_ref_allele((char)p, (char)lsec[vi], (char)r);
      if (s != 'N') { lpri[vi] = r; lsec[vi] = (uint8_t)s; gpri[vi] = r; gsec[vi] = (uint8_t)s; }
    }
  };
  // phase the trace bases of the columns [0, jend): basecall ltrim + (bases before the column)
  auto phase_columns = [&](uint32_t jend) {
    for (uint32_t wd = lo1 >> 6; 64u * wd < jend; ++wd) {
      const uint32_t j = 64u * wd + lane;
      const uint64_t m = ng0[wd];
      if (j < jend && ((m >> lane) & 1ull)) phase_pos(ltrim + lpre0[wd] + (uint32_t)popc64(m & low_mask((int32_t)lane)), row1v[j]);
    }
  };
  phase_columns(found ? alignIndex + 1u : L);
  w.sync();
  DW_CLOCK();

  // ---- 3. class bit sets by ballot ----
  // Four words per pass: the bytes of all four are read from LDS before the first ballot (a word per pass waited for its LDS round trip,
  // two dependent ones for the basecalls' table look-up, 50 times per trace).  Lane k keeps word k of every class in registers (a select
  // per ballot) and stores its six words once at the end: a store per ballot and class was an exec-mask change and an LDS instruction each,
  // 300 times per trace.  (A set has at most 64 words: decomp_wave_caps_ok.)
  uint32_t seen = 0;
  bool exotic = false;
  constexpr uint32_t kSetPass = 4;
  {
    uint64_t mine[kRefClasses];
#pragma unroll
    for (int kk = 0; kk < kRefClasses; ++kk) mine[kk] = 0;  // (word 0 -- the columns left of the window -- and every word behind the window: zero)
    for (uint32_t k0 = 1; k0 < lay.is_stride && 64 * ((int32_t)k0 - 1) < z.Lw; k0 += kSetPass) {
      int cls[kSetPass];
#pragma unroll
      for (uint32_t j = 0; j < kSetPass; ++j) {
        const int32_t q = 64 * ((int32_t)(k0 + j) - 1) + (int32_t)lane;
        const int cc = ref_class(row1v[winstart + (uint32_t)(q < z.Lw ? q : z.Lw - 1)]);  // (Lw > 0 inside the loop)
        cls[j] = q < z.Lw ? cc : 7;
      }
#pragma unroll
      for (uint32_t j = 0; j < kSetPass; ++j) {
        const uint32_t k = k0 + j;
        const int c = cls[j];  // (7 in every lane for a word behind the window: its ballots are zero)
#pragma unroll
        for (int kk = 0; kk < kRefClasses; ++kk) {
          const uint64_t mk = w.ballot(c == kk);
          mine[kk] = lane == k ? mk : mine[kk];
          if (mk) seen |= 1u << kk;
        }
        if (w.ballot(c == 6)) exotic = true;
      }
    }
    if (lane < lay.is_stride) {
#pragma unroll
      for (int kk = 0; kk < kRefClasses; ++kk) is64[(uint32_t)kk * lay.is_stride + lane] = mine[kk];
    }
#pragma unroll
    for (int kk = 0; kk < kRefClasses; ++kk) mine[kk] = 0;
    for (uint32_t k0 = 0; k0 < lay.bad_stride && 64 * (int32_t)k0 < z.NV; k0 += kSetPass) {
      uint32_t pc[kSetPass], mask[kSetPass];
#pragma unroll
      for (uint32_t j = 0; j < kSetPass; ++j) {
        const int32_t sj = 64 * (int32_t)(k0 + j) + (int32_t)lane;
        const uint32_t at = varIndex + (uint32_t)(sj < z.NV ? sj : z.NV - 1);  // (NV > 0 inside the loop)
        pc[j] = lut_pclass(lpri[at]) * (uint32_t)kLutS + lut_sclass(lsec[at]);
      }
#pragma unroll
      for (uint32_t j = 0; j < kSetPass; ++j) {
        const int32_t sj = 64 * (int32_t)(k0 + j) + (int32_t)lane;
        const uint32_t m = llut[pc[j]];
        mask[j] = sj < z.NV ? m : 0u;
      }
#pragma unroll
      for (uint32_t j = 0; j < kSetPass; ++j) {
        const uint32_t k = k0 + j;
#pragma unroll
        for (int kk = 0; kk < kRefClasses; ++kk) {
          const uint64_t mk = w.ballot((mask[j] >> kk) & 1u);
          mine[kk] = lane == k ? mk : mine[kk];
        }
      }
    }
    if (lane < lay.bad_stride) {
#pragma unroll
      for (int kk = 0; kk < kRefClasses; ++kk) bad64[(uint32_t)kk * lay.bad_stride + lane] = mine[kk];
    }
  }
  z.is32 = reinterpret_cast<const uint32_t*>(is64);
  z.bad32 = reinterpret_cast<const uint32_t*>(bad64);
  z.is_stride32 = 2u * lay.is_stride;
  z.bad_stride32 = 2u * lay.bad_stride;
  z.classes = seen;
  w.sync();
  DW_CLOCK();

  // ---- 4. deletion and insertion scans (decompose.h:214-224, 251-261) ----
  // (count_failed's column bound: the scans stop at basecall vend after at most NV steps, i.e. inside the staged window)
  const uint32_t Lscan = winstart + (uint32_t)z.Lw;  // == min(L, winstart + NV + nfref)
  if (exotic) {  // a window with characters outside ACGTN-: byte-wise
    for (uint32_t del = lane; del < nfref; del += 64) fref[del] = (uint16_t)count_failed(row1v, Lscan, lpri, lsec, vend, alignIndex + del + 1, varIndex);
    for (uint32_t ins = 1 + lane; ins < nfins; ins += 64) fins[ins] = (uint16_t)count_failed(row1v, Lscan, lpri, lsec, vend, alignIndex + 1, varIndex + ins);
  } else {
    constexpr int kDiags = 4;  // deletion shifts per lane and pass: lane, + 64, + 128, + 192
    for (uint32_t d0 = 0; d0 < nfref; d0 += 64 * kDiags) {
      bool on[kDiags];
      int32_t fd[kDiags];
#pragma unroll
      for (int j = 0; j < kDiags; ++j) on[j] = d0 + 64u * (uint32_t)j + lane < nfref;
      if (on[0]) {
        diag_count32_multi<kDiags>(z, (int32_t)(d0 + lane), on, fd);
#pragma unroll
        for (int j = 0; j < kDiags; ++j)
          if (on[j]) fref[d0 + 64u * (uint32_t)j + lane] = (uint16_t)fd[j];
      }
    }
    for (uint32_t ins = 1 + lane; ins < nfins; ins += 64) fins[ins] = (uint16_t)diag_count32(z, -(int32_t)ins, (int32_t)ins);
  }
  w.sync();
  if (lane == 0) fins[0] = fref[0];  // decompose.h:249
  DW_CLOCK();

  // ---- 5. cut-off (decompose.h:227-247) ----
  const int32_t med = wave_median(w, nfref, [&](uint32_t i) { return (int32_t)fref[i]; });
  const int32_t mad = wave_median(w, nfref, [&](uint32_t i) { const int32_t dv = (int32_t)fref[i] - med; return dv < 0 ? -dv : dv; });
  int32_t thres = 0;
  if (med > a.prm.madc * mad) thres = med - a.prm.madc * mad;
  if (thres < 10) thres = 10;
  w.sync();  // fins[0]
  DW_CLOCK();

  // ---- 6. picks (decompose.h:236-270) and the decomposition table (:273-285) ----
  auto picks = [&](const uint16_t* f, uint32_t n, uint32_t& cnt, uint32_t& first, uint32_t& last1) {  // first: smallest pick (or ~0), last1: largest + 1 (or 0)
    uint32_t c = 0, fi = 0xffffffffu, la = 0;
    for (uint32_t i = lane; i < n; i += 64) {
      const int32_t fi0 = f[i];
      if (fi0 < thres) {
        bool take = false;
        if ((i + 1 < n) && (2 * fi0 < (int32_t)f[i + 1])) take = true;
        else if ((i > 0) && (2 * fi0 < (int32_t)f[i - 1])) take = true;
        else if ((i == 0) && (i + 2 < n) && (2 * fi0 < (int32_t)f[i + 2])) take = true;
        if (take) { if (fi == 0xffffffffu) fi = i; la = i + 1; ++c; }
      }
    }
    cnt = w.sum(c); first = w.umin(fi); last1 = w.umax(la);
  };
  uint32_t ndel, first_del, last_del1, nins, first_ins, last_ins1;
  picks(fref, nfref, ndel, first_del, last_del1);
  picks(fins, nfins, nins, first_ins, last_ins1);
  DecompOut out{};
  {
    int32_t defins = 15;
    if (ndel == 0 && nins == 0) defins = 50;
    if (nins && (int32_t)last_ins1 - 1 + 15 > defins) defins = (int32_t)last_ins1 - 1 + 15;
    if (defins > (int32_t)nfins) defins = (int32_t)nfins;
    int32_t defdel = 15;
    if (ndel == 0 && nins == 0) defdel = 50;
    if (ndel && (int32_t)last_del1 - 1 + 15 > defdel) defdel = (int32_t)last_del1 - 1 + 15;
    if (defdel > (int32_t)nfref) defdel = (int32_t)nfref;
    int32_t* di = a.dcp_indel + d.dcp_off;
    int32_t* de = a.dcp_err + d.dcp_off;
    const uint32_t ndl = (uint32_t)defdel, nd = ndl + (defins > 1 ? (uint32_t)defins - 1u : 0u);
    for (uint32_t e = lane; e < nd; e += 64) {
      if (e < ndl) { const int32_t i = defdel - 1 - (int32_t)e; di[e] = -i; de[e] = (int32_t)fref[i]; }
      else { const int32_t i = (int32_t)(e - ndl) + 1; di[e] = i; de[e] = (int32_t)fins[i]; }
    }
    out.dcp_n = nd;
    out.kind = 0;
    out.bestIns = 0; out.bestDel = 0; out.bestFR = 1000;
    out.pad = 0;
  }
  DW_CLOCK();

  // ---- 7. nothing picked: complex ins x del search (decompose.h:290-313; decomp_phase_complex) ----
  if (ndel == 0 && nins == 0) {
    const int32_t NI = (int32_t)(mi < maxins / 2 ? mi : maxins / 2), ND = (int32_t)nfref;
    int32_t bfr = 1000, bi = 0, bd = 0;
    if (exotic) {
      for (int32_t ins = (int32_t)lane; ins < NI; ins += 64) {
        int32_t prev = 0;
        for (int32_t del = 0; del < ND; ++del) {
          const int32_t f = count_failed(row1v, Lscan, lpri, lsec, vend, alignIndex + (uint32_t)del + 1, varIndex + (uint32_t)ins);
          complex_consider(f, prev, ins, del, bfr, bi, bd);
          prev = f;
        }
      }
    } else if (NI > 0 && ND > 1) {
      for (int32_t u = 2 - NI + (int32_t)lane; u <= ND - 1; u += 64) {
        const int32_t ins_lo = u >= 1 ? 0 : 1 - u;
        const int32_t ins_hi = (NI - 1) < (ND - 1 - u) ? (NI - 1) : (ND - 1 - u);
        if (ins_lo > ins_hi) continue;
        const int32_t lim_u = z.NV < z.Lw - u ? z.NV : z.Lw - u;
        const int32_t lim_p = z.NV < z.Lw - (u - 1) ? z.NV : z.Lw - (u - 1);
        const int32_t top = (lim_u > lim_p ? lim_u : lim_p);
        int32_t suf_u = 0, suf_p = 0;
        int32_t ins = ins_hi;
        for (int32_t wq = top > 0 ? (top - 1) >> 6 : -1; wq >= (ins_lo >> 6); --wq) {
          const uint64_t zu = diag_word64(z, u, wq, lim_u), zp = diag_word64(z, u - 1, wq, lim_p);
          for (; ins >= ins_lo && ins >= 64 * wq; --ins) {
            if (ins >= 64 * wq + 64) continue;
            const int sft = ins & 63;
            complex_consider(suf_u + popc64(zu >> sft), suf_p + popc64(zp >> sft), ins, ins + u, bfr, bi, bd);
          }
          suf_u += popc64(zu);
          suf_p += popc64(zp);
        }
      }
    }
    // smallest f, then smallest (ins, del): one unsigned minimum (f <= 1000, ins and del < 1024)
    const uint32_t key = w.umin(((uint32_t)bfr << 20) | ((uint32_t)bi << 10) | (uint32_t)bd);
    bfr = (int32_t)(key >> 20);
    out.bestFR = bfr;
    out.bestIns = bfr != 1000 ? (int32_t)((key >> 10) & 1023u) : 0;
    out.bestDel = bfr != 1000 ? (int32_t)(key & 1023u) : 0;
    out.kind = (bfr != 1000) ? 1 : 2;
  }
  DW_CLOCK();

  // ---- 8. rewrite the basecalls along the chosen shift (decompose.h:317-326, 351-371) ----
  if (ndel == 0 && nins == 0 && out.kind != 1) {  // "No InDel detected, traverse the whole alignment" (:327-343)
    phase_columns(L);
  } else {
    uint32_t jstart, vi0;
    if (ndel == 0 && nins == 0) { jstart = alignIndex + (uint32_t)out.bestDel + 1; vi0 = varIndex + (uint32_t)out.bestIns; }
    else if (ndel != 0) { jstart = alignIndex + first_del + 1; vi0 = varIndex; }
    else { jstart = alignIndex + 1; vi0 = varIndex + first_ins; }
    for (uint64_t k = lane;; k += 64) {
      const uint64_t j = (uint64_t)jstart + k, vi = (uint64_t)vi0 + k;
      if (!(j < L && vi < vend)) break;
      phase_pos((uint32_t)vi, row1v[j]);
    }
  }
  if (lane == 0) a.out[t] = out;
  DW_CLOCK();
#undef DW_CLOCK
}

}  // namespace tracyhip
#endif
