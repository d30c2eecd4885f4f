// decompose_kernels.h -- allele deconvolution (decompose.h) as per-trace workgroup phases.
//
// One 64-lane workgroup per trace.  Serial bookkeeping (the walk to the breakpoint, pick rules, table
// emission) runs on lane 0; the expensive parts -- the indel-shift scans (decompose.h:214-224, 251-261,
// 293-313) and the rewrite of the basecalls (:317-326, 351-371) -- spread over the lanes.  Phases
// communicate only through the DecompShared block (LDS on the device) and are separated by barriers,
// so tests/emu can run the same phase functions on the host by looping over lanes.
//
// Everything here is byte/integer work and must be bit-exact with the reference, including its
// unsigned wrap-arounds.
#ifndef TRACY_AMD_DECOMPOSE_KERNELS_H
#define TRACY_AMD_DECOMPOSE_KERNELS_H

#include "dp_lane.h"

namespace tracyhip {

// iupac(char,char), abif.h:142-161
TR_HD char iupac2(char one, char two) {
  int a = one == 'C' ? 1 : one == 'G' ? 2 : one == 'T' ? 3 : 0;
  int b = two == 'C' ? 1 : two == 'G' ? 2 : two == 'T' ? 3 : 0;
  if (b < a) { const int t = a; a = b; b = t; }
  if (a == 0 && b == 2) return 'R';
  if (a == 1 && b == 3) return 'Y';
  if (a == 1 && b == 2) return 'S';
  if (a == 0 && b == 3) return 'W';
  if (a == 2 && b == 3) return 'K';
  if (a == 0 && b == 1) return 'M';
  return 'N';
}

// phaseRefAllele, decompose.h:147-175 (p, s = primary/secondary at the position, r = reference char)
TR_HD char phase_ref_allele(char p, char s, char r) {
  if (r == '-' || s == 'N') return 'N';
  if (s == r) return p;
  char x = 0, y = 0;  // the two bases the IUPAC secondary stands for
  switch (s) {
    case 'R': x = 'A'; y = 'G'; break;
    case 'Y': x = 'C'; y = 'T'; break;
    case 'S': x = 'C'; y = 'G'; break;
    case 'W': x = 'A'; y = 'T'; break;
    case 'K': x = 'G'; y = 'T'; break;
    case 'M': x = 'A'; y = 'C'; break;
    default: return 'N';
  }
  if (r == x) return iupac2(p, y);
  if (r == y) return iupac2(p, x);
  return 'N';
}

struct DecompParams {  // the IndigoConfig fields decomposeAlleles reads (indigo.h:16-40)
  int32_t trimLeft, trimRight, maxindel, madc;
};

// per-trace descriptor (device memory)
struct DecompDesc {
  uint64_t rows_off;     // alignment rows: row0 at rows0 + rows_off, row1 at rows1 + rows_off
  uint64_t bc_off;       // primary / secondary (in-out copies) at bc_off, nbc bytes each
  uint64_t dcp_off;      // decomposition table: dcp_indel/dcp_err + dcp_off, capacity 2*maxindel+2
  uint32_t L;            // alignment columns
  uint32_t nbc;          // bc.consensus.size()
  uint32_t refslice_len; // rs.refslice.size()
  uint32_t breakpoint;   // bp.breakpoint (trimmed-trace coordinates)
};

struct DecompOut {       // per trace
  int32_t kind;          // 0 = an indel shift was applied, 1 = complex (:315), 2 = none (:327)
  int32_t bestIns, bestDel, bestFR;
  uint32_t dcp_n;
  uint32_t pad;
};

struct DecompArgs {
  const DecompDesc* desc;
  const uint8_t* rows0;
  const uint8_t* rows1;
  uint8_t* primary;
  uint8_t* secondary;
  int32_t* dcp_indel;
  int32_t* dcp_err;
  DecompOut* out;
  DecompParams prm;
  uint32_t ntraces;
  const uint32_t* lens;  // alignment columns per trace where they are still on the device (overrides DecompDesc::L), or null
  const uint32_t* skip;  // or null: traces with a non-zero word are left alone (stream.hip: their alignment was not certified, the
                         // host-planned tiers redo them from the untouched basecalls)
  const uint32_t* only;  // or null: only traces with a non-zero word are decomposed (what decompose_wave.h left to this kernel)
};

constexpr int kMaxIndelDev = 1024;  // maxindel handled in LDS (CLI default 1000)

// ---- bit-set formulation of the shift scans ---------------------------------------------------------
// Every scan of decomposeAlleles (decompose.h:214-224, 251-261, 293-313) counts, for a pair of offsets
// (del, ins), the positions t with  !compatible(row1[alignIndex+1+del+t], basecall[varIndex+ins+t]).
// The reference character falls in one of six classes (A C G T N '-'); per class c keep
//   is_c[q]  = 1 iff row1[alignIndex+1+q] is of class c                  (q < kWinBits)
//   bad_c[s] = 1 iff basecall varIndex+s is NOT compatible with class c  (s < kVarBits)
// then  failed(del, ins) = sum_c popc( bad_c[ins+t] & is_c[del+t] ), t < min(Lw-del, NV-ins).
// Writing s = ins+t and u = del-ins this is  sum over s in [ins, min(NV, Lw-u))  of  Z_u[s],
// Z_u[s] = OR_c bad_c[s] & is_c[s+u]: the upper limit depends on the DIAGONAL u only, so one pass
// over a diagonal yields failed() for all its (ins, del) as suffix popcounts -- the complex ins x del
// search costs O((maxins+maxdel) * NV/64) word operations instead of O(maxins * maxdel * NV) byte compares.
constexpr int kRefClasses = 6;
// Sizes of the LDS-resident scan state for a given largest maxindel.  Two instantiations exist: MAXI = 1024 (21 KB, the CLI default
// maxindel 1000 and every Sanger-sized trace) and MAXI = 4096 (80 KB: maxindel up to 4096, traces of up to 8191 basecalls).
template <int MAXI>
struct DecompDims {
  static constexpr int kMaxIndel = MAXI;
  static constexpr int kVarBits = 2 * MAXI;                // basecalls per trace handled (nbc < 2*MAXI)
  static constexpr int kWinBits = kVarBits + MAXI;         // reference columns a scan can reach
  static constexpr int kVarWords = kVarBits / 64 + 2;      // + zero padding for unaligned 64-bit fetches
  static constexpr int kWinWords = kWinBits / 64 + 2;
};
constexpr int kMaxIndelLarge = 4096;
// beyond that the scan state of a trace (1.3 MB) lives in global memory, one slot per resident workgroup: any maxindel / trace length a
// Sanger run can produce, slowly (the reference's vectors have no limit, decompose.h:179-376)
constexpr int kMaxIndelGlobal = 65536;

TR_HD int ref_class(uint8_t r) {
  return r == 'A' ? 0 : r == 'C' ? 1 : r == 'G' ? 2 : r == 'T' ? 3 : r == 'N' ? 4 : r == '-' ? 5 : 6;
}
TR_HD char class_char(int c) { return c == 0 ? 'A' : c == 1 ? 'C' : c == 2 ? 'G' : c == 3 ? 'T' : c == 4 ? 'N' : '-'; }
// the predicate the scans count the negation of (decompose.h:218-220)
TR_HD bool ref_compatible(char p, char s, char r) { return r == p || phase_ref_allele(p, s, r) != 'N'; }

TR_HD int popc64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __popcll(x);
#else
  return __builtin_popcountll(x);
#endif
}
// 64 bits starting at (possibly negative) bit position pos of a zero-padded bit set of nw words
TR_HD uint64_t bits_at(const uint64_t* a, int32_t pos, int32_t nw) {
  const int32_t w = pos >> 6;  // arithmetic shift: floor
  const uint32_t sft = (uint32_t)pos & 63u;
  const uint64_t lo = (w >= 0 && w < nw) ? a[w] : 0;
  if (sft == 0) return lo;
  const uint64_t hi = (w + 1 >= 0 && w + 1 < nw) ? a[w + 1] : 0;
  return (lo >> sft) | (hi << (64 - sft));
}
TR_HD uint64_t low_mask(int32_t nbits) { return nbits >= 64 ? ~0ull : nbits <= 0 ? 0ull : ((1ull << nbits) - 1ull); }

template <int MAXI>
struct DecompSharedT : DecompDims<MAXI> {
  using Dims = DecompDims<MAXI>;
  int32_t fref[MAXI];
  int32_t fins[MAXI];
  int32_t hist[MAXI * 2 + 2];
  uint64_t is_c[kRefClasses][Dims::kWinWords];
  uint64_t bad_c[kRefClasses][Dims::kVarWords];
  uint32_t seg0[64], seg1[64];  // per-lane segment counts of trace / reference bases (alignment walk); later the scratch of the pick phases
  uint32_t segt[64];            // the trace-base counts again, kept to the end: the traverse of the whole alignment (decompose.h:327-343) starts
                                // every lane's segment at its own basecall (seg0 is long overwritten by then)
  uint32_t nfref, nfins;
  uint32_t varIndex, refPointer, alignIndex;
  uint32_t maxdel, maxins, bp;
  uint32_t found;               // the walk reached the breakpoint
  uint32_t exotic;              // the scan window holds a character outside ACGTN- : byte-wise fallback
  uint32_t classes;             // bit c set iff class c occurs in the window
  int32_t Lw, NV;               // columns from alignIndex+1 to L, basecalls from varIndex to vend (clamped to the bit sets)
  int32_t med, thres;          // median of the failed counts; the cut-off derived from it and the MAD
  int32_t pick_del, pick_ins;  // smallest picked deletion / insertion, -1 = none
  int32_t ndel, nins;          // number of picks
  int32_t maxpick_del, maxpick_ins;
  int32_t best_fr[64], best_ins[64], best_del[64];
};
using DecompShared = DecompSharedT<kMaxIndelDev>;

// failedref of one shift (decompose.h:215-222 and the two other copies of that loop), byte-wise: the
// fallback for windows with exotic characters and the cross-check of the bit-set path in tests/emu
TR_HD int32_t count_failed(const uint8_t* row1, uint32_t L, const uint8_t* pri, const uint8_t* sec, uint64_t vend,
                           uint32_t jstart, uint32_t vi) {
  int32_t failed = 0;
  for (uint32_t j = jstart; (j < L) && ((uint64_t)vi < vend); ++j, ++vi) {
    const char r = (char)row1[j], p = (char)pri[vi];
    if (r != p) {
      if (phase_ref_allele(p, (char)sec[vi], r) == 'N') ++failed;
    }
  }
  return failed;
}

TR_HD uint64_t decomp_vend(const DecompArgs& a, const DecompDesc& d) {
  return (uint64_t)d.nbc - (uint64_t)(int64_t)a.prm.trimRight;  // size_t arithmetic of decompose.h:216
}

// segment of alignment columns lane l walks
TR_HD void lane_segment(uint32_t L, uint32_t lane, uint32_t& lo, uint32_t& hi) {
  const uint32_t seg = (L + 63) / 64;
  lo = lane * seg < L ? lane * seg : L;
  hi = lo + seg < L ? lo + seg : L;
}

// ---- phase 0 (all lanes): trace / reference bases per segment of the alignment ----
template <class SH>
TR_HD void decomp_phase_count(const DecompArgs& a, const DecompDesc& d, SH& sh, uint32_t lane) {
  const uint8_t* row0 = a.rows0 + d.rows_off;
  const uint8_t* row1 = a.rows1 + d.rows_off;
  uint32_t lo, hi, c0 = 0, c1 = 0;
  lane_segment(d.L, lane, lo, hi);
  for (uint32_t j = lo; j < hi; ++j) {
    c0 += row0[j] != '-';
    c1 += row1[j] != '-';
  }
  sh.seg0[lane] = c0;
  sh.seg1[lane] = c1;
  sh.segt[lane] = c0;
  if (lane == 0) { sh.found = 0; sh.alignIndex = 0; sh.varIndex = 0; sh.exotic = 0; sh.classes = 0; }
}

// phase the basecall at vi against reference character r (decompose.h:196-203)
TR_HD void phase_position(uint8_t* pri, uint8_t* sec, uint32_t vi, uint8_t r) {
  if (r != pri[vi]) {
    const char s = phase_ref_allele((char)pri[vi], (char)sec[vi], (char)r);
    if (s != 'N') { pri[vi] = r; sec[vi] = (uint8_t)s; }
  }
}

// ---- phase 1 (all lanes): walk to the breakpoint, phasing as we go (decompose.h:184-208) ----
// The k-th trace base of the alignment is basecall trimLeft+k-1; the walk stops after the base that
// makes vi == bp, i.e. after trace base number bp - trimLeft.  Every lane walks its own segment.
template <class SH>
TR_HD void decomp_phase_walk(const DecompArgs& a, const DecompDesc& d, SH& sh, uint32_t lane) {
  const uint8_t* row0 = a.rows0 + d.rows_off;
  const uint8_t* row1 = a.rows1 + d.rows_off;
  uint8_t* pri = a.primary + d.bc_off;
  uint8_t* sec = a.secondary + d.bc_off;
  const uint32_t ltrim = (uint32_t)a.prm.trimLeft;
  const uint32_t bp = d.breakpoint + ltrim;
  uint32_t base0 = 0, base1 = 0;
  for (uint32_t l = 0; l < lane; ++l) { base0 += sh.seg0[l]; base1 += sh.seg1[l]; }
  const uint32_t stop = bp - ltrim;  // 1-based number of the last trace base walked (0 or > total: never reached)
  if (stop != 0 && base0 >= stop) return;
  uint32_t lo, hi;
  lane_segment(d.L, lane, lo, hi);
  uint32_t k = base0, ref = base1;
  for (uint32_t j = lo; j < hi; ++j) {
    if (row0[j] != '-') {
      phase_position(pri, sec, ltrim + k, row1[j]);
      ++k;
      if (k == stop) {
        sh.found = 1; sh.alignIndex = j; sh.varIndex = ltrim + k; sh.refPointer = ref;
        return;
      }
    }
    if (row1[j] != '-') ++ref;
  }
}

// ---- phase 2 (lane 0): scan bounds (decompose.h:210-213, 248-250) ----
template <class SH>
TR_HD void decomp_phase_bounds(const DecompArgs& a, const DecompDesc& d, SH& sh) {
  const int32_t rtrim = a.prm.trimRight;
  const uint32_t bp = d.breakpoint + (uint32_t)a.prm.trimLeft;
  if (!sh.found) {
    uint32_t ref = 0;
    for (int l = 0; l < 64; ++l) ref += sh.seg1[l];
    sh.refPointer = ref;
  }
  const uint32_t refPointer = sh.refPointer;
  uint32_t maxdel = 2;
  if ((uint64_t)d.refslice_len > (uint64_t)(uint32_t)(refPointer + (uint32_t)rtrim + 2u))
    maxdel = (uint32_t)((uint64_t)d.refslice_len - (uint64_t)(uint32_t)(refPointer + (uint32_t)rtrim));
  sh.maxdel = maxdel;
  sh.bp = bp;
  sh.maxins = (uint32_t)((int32_t)d.nbc - (int32_t)((uint32_t)rtrim + bp));
  uint32_t nf = 0;
  for (uint32_t del = 0; (del < (uint32_t)a.prm.maxindel) && (del < maxdel / 2); ++del) ++nf;
  sh.nfref = nf;
  uint32_t ni = 1;
  for (uint32_t ins = 1; (ins < (uint32_t)a.prm.maxindel) && (ins < sh.maxins / 2); ++ins) ++ni;
  sh.nfins = ni;
  // extent of the bit sets
  const uint32_t winstart = sh.alignIndex + 1;
  const int64_t lw = (int64_t)d.L - (int64_t)winstart;
  sh.Lw = (int32_t)(lw < 0 ? 0 : lw > SH::kWinBits ? SH::kWinBits : lw);
  uint64_t vend = decomp_vend(a, d);
  if (vend > d.nbc) vend = d.nbc;  // rtrim < 0 or > nbc: the reference reads out of bounds there
  const int64_t nv = (int64_t)vend - (int64_t)sh.varIndex;
  sh.NV = (int32_t)(nv < 0 ? 0 : nv > SH::kVarBits ? SH::kVarBits : nv);
}

// ---- phase 3 (all lanes): build the class bit sets; lane l builds word l, l+64, ... of each ----
template <class SH>
TR_HD void decomp_phase_bitsets(const DecompArgs& a, const DecompDesc& d, SH& sh, uint32_t lane) {
  const uint8_t* row1 = a.rows1 + d.rows_off;
  const uint8_t* pri = a.primary + d.bc_off;
  const uint8_t* sec = a.secondary + d.bc_off;
  const uint32_t winstart = sh.alignIndex + 1;
  uint32_t seen = 0, exotic = 0;
  for (int32_t w = (int32_t)lane; w < SH::kWinWords; w += 64) {
    uint64_t bits[kRefClasses] = {0, 0, 0, 0, 0, 0};
    for (int32_t q = 64 * w; q < 64 * w + 64 && q < sh.Lw; ++q) {
      const int c = ref_class(row1[winstart + (uint32_t)q]);
      if (c >= kRefClasses) { exotic = 1; continue; }
      bits[c] |= 1ull << (q & 63);
      seen |= 1u << c;
    }
    for (int c = 0; c < kRefClasses; ++c) sh.is_c[c][w] = bits[c];
  }
  for (int32_t w = (int32_t)lane; w < SH::kVarWords; w += 64) {
    uint64_t bits[kRefClasses] = {0, 0, 0, 0, 0, 0};
    for (int32_t s = 64 * w; s < 64 * w + 64 && s < sh.NV; ++s) {
      const char p = (char)pri[sh.varIndex + (uint32_t)s], sc = (char)sec[sh.varIndex + (uint32_t)s];
      for (int c = 0; c < kRefClasses; ++c)
        if (!ref_compatible(p, sc, class_char(c))) bits[c] |= 1ull << (s & 63);
    }
    for (int c = 0; c < kRefClasses; ++c) sh.bad_c[c][w] = bits[c];
  }
  // lane-private results are merged through the per-lane scratch arrays (no atomics needed)
  sh.best_fr[lane] = (int32_t)seen;
  sh.best_ins[lane] = (int32_t)exotic;
}

// ---- phase 4 (lane 0): merge the per-lane class / exotic flags ----
template <class SH>
TR_HD void decomp_phase_flags(SH& sh) {
  uint32_t seen = 0, exotic = 0;
  for (int l = 0; l < 64; ++l) { seen |= (uint32_t)sh.best_fr[l]; exotic |= (uint32_t)sh.best_ins[l]; }
  sh.classes = seen;
  sh.exotic = exotic;
}

// Z_u word: bits s in [64*w, 64*w+64) of  OR_c bad_c[s] & is_c[s+u], cut at s < limit
template <class SH>
TR_HD uint64_t diag_word(const SH& sh, int32_t u, int32_t w, int32_t limit) {
  uint64_t z = 0;
  for (int c = 0; c < kRefClasses; ++c) {
    if (!((sh.classes >> c) & 1u)) continue;
    z |= sh.bad_c[c][w] & bits_at(sh.is_c[c], 64 * w + u, SH::kWinWords);
  }
  return z & low_mask(limit - 64 * w);
}
// number of positions s >= from of diagonal u  ( = failed(del, ins) for del - ins == u, from == ins)
template <class SH>
TR_HD int32_t diag_count_from(const SH& sh, int32_t u, int32_t from) {
  const int32_t lim_ref = sh.Lw - u;
  const int32_t limit = sh.NV < lim_ref ? sh.NV : lim_ref;
  int32_t f = 0;
  for (int32_t w = from >> 6; 64 * w < limit; ++w) {
    uint64_t z = diag_word(sh, u, w, limit);
    if (w == (from >> 6)) z &= ~low_mask(from & 63);
    f += popc64(z);
  }
  return f;
}

// ---- phase 5 (all lanes): deletion and insertion scans (decompose.h:214-224, 251-261) ----
template <class SH>
TR_HD void decomp_phase_scan(const DecompArgs& a, const DecompDesc& d, SH& sh, uint32_t lane) {
  if (sh.exotic) {
    const uint8_t* row1 = a.rows1 + d.rows_off;
    const uint8_t* pri = a.primary + d.bc_off;
    const uint8_t* sec = a.secondary + d.bc_off;
    const uint64_t vend = decomp_vend(a, d);
    for (uint32_t del = lane; del < sh.nfref; del += 64)
      sh.fref[del] = count_failed(row1, d.L, pri, sec, vend, sh.alignIndex + del + 1, sh.varIndex);
    for (uint32_t ins = 1 + lane; ins < sh.nfins; ins += 64)
      sh.fins[ins] = count_failed(row1, d.L, pri, sec, vend, sh.alignIndex + 1, sh.varIndex + ins);
  } else {
    for (uint32_t del = lane; del < sh.nfref; del += 64) sh.fref[del] = diag_count_from(sh, (int32_t)del, 0);
    for (uint32_t ins = 1 + lane; ins < sh.nfins; ins += 64) sh.fins[ins] = diag_count_from(sh, -(int32_t)ins, (int32_t)ins);
  }
}

// ---- cut-offs, picks, decomposition table (decompose.h:227-285) ----
// The reference takes the median and the median absolute deviation of the failed counts, derives a threshold, picks the
// shifts that undercut it and their neighbours, and prints a table.  All of it is a handful of passes over <= 2 MAXI small
// ints, done by the 64 lanes together in the steps below (a barrier between them); lane 0 only joins 64 partial results.
// (One lane walking histograms of 2 MAXI + 2 bins in LDS, one dependent access after the other, was most of this kernel's time.)
TR_HD void lds_inc(int32_t* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  atomicAdd(p, 1);
#else
  ++*p;
#endif
}
template <class SH>
TR_HD uint32_t pick_bins() { return (uint32_t)SH::kMaxIndel * 2 + 2; }
template <class SH>
TR_HD void pick_zero(SH& sh, uint32_t lane) {
  if (lane == 0) sh.fins[0] = sh.fref[0];  // decompose.h:249 (fins[0] is not read before the picks)
  for (uint32_t i = lane; i < pick_bins<SH>(); i += 64) sh.hist[i] = 0;
}
// histogram of |fref[i] - centre| (centre 0: of the values themselves), values clamped to the last bin
template <class SH>
TR_HD void pick_count(SH& sh, uint32_t lane, int32_t centre) {
  const uint32_t nh = pick_bins<SH>();
  for (uint32_t i = lane; i < sh.nfref; i += 64) {
    int32_t dv = sh.fref[i] - centre;
    if (dv < 0) dv = -dv;
    lds_inc(&sh.hist[(uint32_t)dv < nh ? (uint32_t)dv : nh - 1]);
  }
}
// per-lane totals of consecutive blocks of bins
template <class SH>
TR_HD void pick_blocks(SH& sh, uint32_t lane) {
  const uint32_t nh = pick_bins<SH>(), B = (nh + 63) / 64;
  uint32_t sum = 0;
  for (uint32_t x = lane * B; x < (lane + 1) * B && x < nh; ++x) sum += (uint32_t)sh.hist[x];
  sh.seg0[lane] = sum;
}
// value at sorted position n/2 (getMedian, decompose.h:129-135): the first bin at which the running count exceeds n/2
template <class SH>
TR_HD int32_t pick_median(const SH& sh) {
  const uint32_t nh = pick_bins<SH>(), B = (nh + 63) / 64, half = sh.nfref / 2;
  uint32_t seen = 0;
  for (uint32_t b = 0; b < 64; ++b) {
    if (seen + sh.seg0[b] > half) {
      for (uint32_t x = b * B; x < (b + 1) * B && x < nh; ++x) {
        seen += (uint32_t)sh.hist[x];
        if (seen > half) return (int32_t)x;
      }
    }
    seen += sh.seg0[b];
  }
  return 0;
}
// the shifts a lane finds among its share of fref / fins: count, first and last index (decompose.h:251-270)
template <class SH>
TR_HD void pick_candidates(SH& sh, uint32_t lane) {
  const uint32_t nfref = sh.nfref, nfins = sh.nfins;
  const int32_t thres = sh.thres;
  auto scan = [&](const int32_t* f, uint32_t n, int32_t& cnt, int32_t& first, int32_t& last) {
    cnt = 0; first = 0x7fffffff; last = -1;
    for (uint32_t i = lane; i < n; i += 64) {
      if (f[i] < thres) {
        bool take = false;
        if ((i + 1 < n) && (2 * f[i] < f[i + 1])) take = true;
        else if ((i > 0) && (2 * f[i] < f[i - 1])) take = true;
        else if ((i == 0) && (i + 2 < n) && (2 * f[i] < f[i + 2])) take = true;
        if (take) { if (first == 0x7fffffff) first = (int32_t)i; last = (int32_t)i; ++cnt; }
      }
    }
  };
  int32_t c, f, l;
  scan(sh.fref, nfref, c, f, l);
  sh.seg0[lane] = (uint32_t)c; sh.best_fr[lane] = f; sh.best_ins[lane] = l;
  scan(sh.fins, nfins, c, f, l);
  sh.seg1[lane] = (uint32_t)c; sh.best_del[lane] = f; sh.hist[lane] = l;
}
// lane 0: join the candidates, write the decomposition table (decompose.h:273-285)
template <class SH>
TR_HD void pick_finish(const DecompArgs& a, const DecompDesc& d, SH& sh, DecompOut& out) {
  const uint32_t nfref = sh.nfref, nfins = sh.nfins;
  int32_t ndel = 0, first_del = 0x7fffffff, max_del = -1, nins = 0, first_ins = 0x7fffffff, max_ins = -1;
  for (uint32_t b = 0; b < 64; ++b) {
    ndel += (int32_t)sh.seg0[b]; nins += (int32_t)sh.seg1[b];
    if (sh.best_fr[b] < first_del) first_del = sh.best_fr[b];
    if (sh.best_ins[b] > max_del) max_del = sh.best_ins[b];
    if (sh.best_del[b] < first_ins) first_ins = sh.best_del[b];
    if (sh.hist[b] > max_ins) max_ins = sh.hist[b];
  }
  if (ndel == 0) first_del = -1;
  if (nins == 0) first_ins = -1;
  sh.ndel = ndel; sh.nins = nins;
  sh.pick_del = first_del; sh.pick_ins = first_ins;

  // decomposition table (decompose.h:273-285); picks are ascending so the largest is the last one
  int32_t defins = 15;
  if (ndel == 0 && nins == 0) defins = 50;
  if (nins && max_ins + 15 > defins) defins = max_ins + 15;
  if (defins > (int32_t)nfins) defins = (int32_t)nfins;
  int32_t defdel = 15;
  if (ndel == 0 && nins == 0) defdel = 50;
  if (ndel && max_del + 15 > defdel) defdel = max_del + 15;
  if (defdel > (int32_t)nfref) defdel = (int32_t)nfref;
  int32_t* di = a.dcp_indel + d.dcp_off;
  int32_t* de = a.dcp_err + d.dcp_off;
  uint32_t nd = 0;
  for (int32_t i = defdel - 1; i >= 0; --i) { di[nd] = -i; de[nd] = sh.fref[i]; ++nd; }
  for (int32_t i = 1; i < defins; ++i) { di[nd] = i; de[nd] = sh.fins[i]; ++nd; }
  out.dcp_n = nd;
  out.kind = 0;
  out.bestIns = 0; out.bestDel = 0; out.bestFR = 1000;
  out.pad = 0;
}

// ---- phase 7 (all lanes, only when nothing was picked): complex ins x del search (:290-313) ----
// The reference runs ins outer / del inner with prevFailedRef reset to 0 per ins, and accepts a candidate
// iff 2*f < prev and f < bestFR (strict): the overall winner is the smallest f among the 2*f < prev
// candidates, earliest (ins, del) on ties.  Lane l takes the diagonals u = del - ins = umin + l, + 64, ...;
// the predecessor f(ins, del-1) lies on diagonal u-1, which the lane sweeps in lock-step (suffix counts).
TR_HD void complex_consider(int32_t f, int32_t prev, int32_t ins, int32_t del, int32_t& bfr, int32_t& bi, int32_t& bd) {
  if (!(2 * f < prev && f < 1000)) return;
  if (f < bfr || (f == bfr && (ins < bi || (ins == bi && del < bd)))) { bfr = f; bi = ins; bd = del; }
}
template <class SH>
TR_HD void decomp_phase_complex(const DecompArgs& a, const DecompDesc& d, SH& sh, uint32_t lane) {
  int32_t bfr = 1000, bi = 0, bd = 0;
  // loop limits of decompose.h:293-294
  int32_t NI = 0, ND = 0;
  for (uint32_t ins = 0; (ins < (uint32_t)a.prm.maxindel) && (ins < sh.maxins / 2); ++ins) ++NI;
  for (uint32_t del = 0; (del < (uint32_t)a.prm.maxindel) && (del < sh.maxdel / 2); ++del) ++ND;
  if (sh.exotic) {
    const uint8_t* row1 = a.rows1 + d.rows_off;
    const uint8_t* pri = a.primary + d.bc_off;
    const uint8_t* sec = a.secondary + d.bc_off;
    const uint64_t vend = decomp_vend(a, d);
    for (int32_t ins = (int32_t)lane; ins < NI; ins += 64) {
      int32_t prev = 0;
      for (int32_t del = 0; del < ND; ++del) {
        const int32_t f = count_failed(row1, d.L, pri, sec, vend, sh.alignIndex + (uint32_t)del + 1, sh.varIndex + (uint32_t)ins);
        complex_consider(f, prev, ins, del, bfr, bi, bd);
        prev = f;
      }
    }
  } else if (NI > 0 && ND > 1) {
    // del = 0 has prev = 0 and can never be accepted; diagonals u in [1 - (NI-1), ND-1] hold the del >= 1 entries
    for (int32_t u = 2 - NI + (int32_t)lane; u <= ND - 1; u += 64) {
      const int32_t ins_lo = u >= 1 ? 0 : 1 - u;
      const int32_t ins_hi = (NI - 1) < (ND - 1 - u) ? (NI - 1) : (ND - 1 - u);
      if (ins_lo > ins_hi) continue;
      const int32_t lim_u = sh.NV < sh.Lw - u ? sh.NV : sh.Lw - u;            // diagonal u:   s < lim_u
      const int32_t lim_p = sh.NV < sh.Lw - (u - 1) ? sh.NV : sh.Lw - (u - 1);  // diagonal u-1
      const int32_t top = (lim_u > lim_p ? lim_u : lim_p);
      int32_t suf_u = 0, suf_p = 0;
      int32_t ins = ins_hi;
      for (int32_t w = top > 0 ? (top - 1) >> 6 : -1; w >= (ins_lo >> 6); --w) {
        const uint64_t zu = diag_word(sh, u, w, lim_u), zp = diag_word(sh, u - 1, w, lim_p);
        for (; ins >= ins_lo && ins >= 64 * w; --ins) {
          if (ins >= 64 * w + 64) continue;  // above the populated words: both counts are zero there
          const int sft = ins & 63;
          complex_consider(suf_u + popc64(zu >> sft), suf_p + popc64(zp >> sft), ins, ins + u, bfr, bi, bd);
        }
        suf_u += popc64(zu);
        suf_p += popc64(zp);
      }
      // entries whose ins lies above every populated word have f = prev = 0: never accepted (2*0 < 0 fails)
    }
  }
  sh.best_fr[lane] = bfr; sh.best_ins[lane] = bi; sh.best_del[lane] = bd;
}
template <class SH>
TR_HD void decomp_phase_complex_reduce(SH& sh, DecompOut& out) {
  int32_t bfr = 1000, bi = 0, bd = 0;
  for (int l = 0; l < 64; ++l) {
    const int32_t f = sh.best_fr[l];
    if (f == 1000) continue;
    if (f < bfr || (f == bfr && (sh.best_ins[l] < bi || (sh.best_ins[l] == bi && sh.best_del[l] < bd)))) {
      bfr = f; bi = sh.best_ins[l]; bd = sh.best_del[l];
    }
  }
  out.bestFR = bfr; out.bestIns = bi; out.bestDel = bd;
  out.kind = (bfr != 1000) ? 1 : 2;
}

// ---- phase 9 (all lanes): rewrite the basecalls along the chosen shift (:317-326, 351-371) ----
// Each position vi is touched once and only reads its own primary/secondary: lanes split the range.
template <class SH>
TR_HD void decomp_phase_apply(const DecompArgs& a, const DecompDesc& d, const SH& sh, const DecompOut& out,
                              uint32_t lane) {
  const uint8_t* row0 = a.rows0 + d.rows_off;
  const uint8_t* row1 = a.rows1 + d.rows_off;
  uint8_t* pri = a.primary + d.bc_off;
  uint8_t* sec = a.secondary + d.bc_off;
  const uint64_t vend = decomp_vend(a, d);
  uint32_t jstart, vi0;
  if (sh.ndel == 0 && sh.nins == 0) {
    if (out.kind == 1) { jstart = sh.alignIndex + (uint32_t)out.bestDel + 1; vi0 = sh.varIndex + (uint32_t)out.bestIns; }
    else {  // "No InDel detected, traverse the whole alignment" (:327-343): vi advances only on trace bases
      uint32_t base0 = 0, lo, hi;
      for (uint32_t l = 0; l < lane; ++l) base0 += sh.segt[l];
      lane_segment(d.L, lane, lo, hi);
      uint32_t vi = (uint32_t)a.prm.trimLeft + base0;
      for (uint32_t j = lo; j < hi; ++j) {
        if (row0[j] != '-') {
          phase_position(pri, sec, vi, row1[j]);
          ++vi;
        }
      }
      return;
    }
  } else if (sh.ndel != 0) { jstart = sh.alignIndex + (uint32_t)sh.pick_del + 1; vi0 = sh.varIndex; }
  else { jstart = sh.alignIndex + 1; vi0 = sh.varIndex + (uint32_t)sh.pick_ins; }
  for (uint64_t k = lane;; k += 64) {
    const uint64_t j = (uint64_t)jstart + k, vi = (uint64_t)vi0 + k;
    if (!(j < d.L && vi < vend)) break;
    phase_position(pri, sec, (uint32_t)vi, row1[j]);
  }
}

// ---- the phase schedule, shared by the HIP kernel (a barrier after every step) and tests/emu (lanes looped) ----
constexpr int kDecompSteps = 19;
// whether step `st` runs on all lanes (true) or on lane 0 only (false)
TR_HD bool decomp_step_all_lanes(int st) {
  switch (st) {
    case 0: case 1: case 3: case 5: case 6: case 7: case 8: case 10: case 11: case 12: case 14: case 16: case 18: return true;
    default: return false;
  }
}
template <class SH>
TR_HD void decomp_step(int st, const DecompArgs& a, const DecompDesc& d, SH& sh, DecompOut& out, uint32_t lane) {
  switch (st) {
    case 0: decomp_phase_count(a, d, sh, lane); break;
    case 1: decomp_phase_walk(a, d, sh, lane); break;
    case 2: decomp_phase_bounds(a, d, sh); break;
    case 3: decomp_phase_bitsets(a, d, sh, lane); break;
    case 4: decomp_phase_flags(sh); break;
    case 5: decomp_phase_scan(a, d, sh, lane); break;
    // median of the failed counts (decompose.h:227-231) ...
    case 6: pick_zero(sh, lane); break;
    case 7: pick_count(sh, lane, 0); break;
    case 8: pick_blocks(sh, lane); break;
    case 9: sh.med = pick_median(sh); break;
    // ... their median absolute deviation, and the cut-off (decompose.h:232-247)
    case 10: pick_zero(sh, lane); break;
    case 11: pick_count(sh, lane, sh.med); break;
    case 12: pick_blocks(sh, lane); break;
    case 13: {
      const int32_t mad = pick_median(sh);
      int32_t thres = 0;
      if (sh.med > a.prm.madc * mad) thres = sh.med - a.prm.madc * mad;
      if (thres < 10) thres = 10;
      sh.thres = thres;
      break;
    }
    case 14: pick_candidates(sh, lane); break;
    case 15: pick_finish(a, d, sh, out); break;
    case 16: if (sh.ndel == 0 && sh.nins == 0) decomp_phase_complex(a, d, sh, lane); break;
    case 17:
      if (sh.ndel == 0 && sh.nins == 0) decomp_phase_complex_reduce(sh, out);
      sh.best_fr[0] = out.bestFR; sh.best_ins[0] = out.bestIns; sh.best_del[0] = out.bestDel; sh.hist[0] = out.kind;
      break;
    case 18:
      if (lane != 0) { out.bestFR = sh.best_fr[0]; out.bestIns = sh.best_ins[0]; out.bestDel = sh.best_del[0]; out.kind = sh.hist[0]; }
      decomp_phase_apply(a, d, sh, out, lane);
      break;
    default: break;
  }
}

// =====================================================================================================
// findBreakpoint (decompose.h:7-56).  sig[] (double, one per column) lives in LDS / scratch.
// =====================================================================================================
struct BreakpointOut {  // TraceBreakpoint, fmindex.h:51-56
  int32_t indelshift;
  int32_t traceleft;
  uint32_t breakpoint;
  float bestDiff;
};

// column j: best - second best of the 6 profile rows, both floored at 0.001 (decompose.h:12-24)
TR_HD double signal_ratio(const float* p, uint32_t stride, uint32_t j) {
  double best = 0.001, snd = 0.001;
  for (uint32_t i = 0; i < 6; ++i) {
    const float v = p[(uint64_t)i * stride + j];
    if (v > best) { snd = best; best = v; }
    else if (v > snd) { snd = v; }
  }
#if defined(__HIP_DEVICE_COMPILE__)
  return __dsub_rn(best, snd);
#else
  return best - snd;
#endif
}

// |mean(right 25) - mean(left 25)| at i, summed in the reference's order (decompose.h:32-38)
TR_HD double window_diff(const double* sig, uint32_t i, double* left_out, double* right_out) {
  double ls = 0, rs = 0;
  for (uint32_t k = i - 25; k < i; ++k) ls += sig[k];
  for (uint32_t k = i; k < i + 25; ++k) rs += sig[k];
  const double left = ls / 25.0, right = rs / 25.0;
  *left_out = left;
  *right_out = right;
  const double dd = right - left;
  return dd < 0 ? -dd : dd;
}

// the sequential maximum search with the float-typed bestDiff field (decompose.h:27-55)
TR_HD void breakpoint_select(const double* diff, const uint8_t* left_lt_right, uint32_t ncol, BreakpointOut& bp) {
  bp.bestDiff = 0;
  bp.traceleft = 1;
  bp.breakpoint = 0;
  if (25 < ncol) {
    for (uint32_t i = 25; i < ncol - 25; ++i) {
      if (diff[i] > (double)bp.bestDiff) {
        bp.breakpoint = i;
        bp.bestDiff = (float)diff[i];
        bp.traceleft = left_lt_right[i] ? 0 : 1;
      }
    }
  }
  bp.indelshift = 1;
  if ((double)bp.bestDiff < 0.25) {
    bp.indelshift = 0;
    bp.breakpoint = ncol;
    bp.traceleft = 1;
    bp.bestDiff = 0;
  }
}

// ---- findHomozygousBreakpoint (decompose.h:59-128): one wavefront per trace --------------------------
// The two alignment rows are read 64 columns at a time; a chunk becomes two 64-bit masks -- row0 != row1 and row0 != '-' --
// and the mismatch counts of the two 25-column windows of column b + lane are popcounts of 25 bits cut from three consecutive
// mismatch masks; varIndex (bases of row0 up to and including the column) is a running popcount of the other mask.  The
// reference's walk with its float-typed running maximum ends on the last column with diff > (double)F, or else on the first
// whose float equals F = max (float)diff (see breakpoint_kernel); it is resolved by reductions over the lanes.
// The kernel (decompose_kernels.hip) gets the masks by ballot and reduces by shuffles; the host twin below builds them with
// loops and walks the lanes one after the other -- everything a lane computes is shared.
struct HomChunk { uint64_t mm, ng; };  // columns b .. b+63: row0 != row1, row0 != '-' (columns >= L: 0)
TR_HD uint64_t mask_bits(uint64_t lo, uint64_t hi, uint32_t s) { return s ? (lo >> s) | (hi << (64 - s)) : lo; }

struct HomLane {      // what lane `lane` knows about column b + lane
  double diff;        // |right - left| of the window means, the reference's doubles
  float g;            // (float)diff
  int32_t left_lt_right;
  uint32_t var;       // varIndex after the column
};
TR_HD HomLane hom_lane(uint64_t prev_mm, const HomChunk& cur, uint64_t next_mm, uint32_t vbase, uint32_t lane) {
  const uint64_t lw = lane < 25 ? mask_bits(prev_mm, cur.mm, lane + 39) : mask_bits(cur.mm, next_mm, lane - 25);
  const uint64_t rw = mask_bits(cur.mm, next_mm, lane);
  const int32_t lc = popc64(lw & 0x1ffffffull), rc = popc64(rw & 0x1ffffffull);
  const double left = (double)lc / 25.0, right = (double)rc / 25.0;
  HomLane h;
  h.diff = right - left;
  if (h.diff < 0) h.diff = -h.diff;
  h.g = (float)h.diff;
  h.left_lt_right = (left < right) ? 1 : 0;
  h.var = vbase + (uint32_t)popc64(cur.ng & ((2ull << lane) - 1ull));
  return h;
}
// bases of row0 in the columns of the chunk at b that lie below hi
TR_HD uint32_t hom_bases_below(const HomChunk& cur, uint64_t b, uint32_t hi) {
  return (uint32_t)popc64(hi - b >= 64 ? cur.ng : cur.ng & ((1ull << (hi - b)) - 1ull));
}
// the per-lane state of the selecting sweep
struct HomPick { uint32_t first, last, first_var, last_var; int32_t first_tl, last_tl; };  // last: column + 1, 0 = none
TR_HD void hom_pick_init(HomPick& p) { p.first = 0xffffffffu; p.last = 0; p.first_var = p.last_var = 0; p.first_tl = p.last_tl = 0; }
TR_HD void hom_pick(HomPick& p, const HomLane& h, float F, uint32_t col) {
  if (h.g == F && h.diff > 0.0 && col < p.first) { p.first = col; p.first_var = h.var; p.first_tl = h.left_lt_right; }
  if (h.diff > (double)F) { p.last = col + 1; p.last_var = h.var; p.last_tl = h.left_lt_right; }
}
// alignStart / alignEnd checks and the two early ways out (decompose.h:62-85); 1 = go on with columns [lo, hi)
TR_HD int hom_range(int64_t align_start, int64_t align_end, BreakpointOut& bp, uint32_t& lo, uint32_t& hi) {
  if (align_start >= align_end) return 0;
  bp.bestDiff = 0;
  bp.traceleft = 1;
  bp.breakpoint = 0;
  if (align_end < align_start + 50) return -1;
  lo = (uint32_t)(align_start + 25);
  hi = (uint32_t)(align_end - 25);
  return 1;
}
TR_HD void hom_finish(BreakpointOut& bp, float F, uint32_t var_end) {  // decompose.h:119-126 when no window pair reaches 0.25
  bp.indelshift = 1;
  if ((double)F < 0.25) {
    bp.indelshift = 0;
    bp.breakpoint = var_end;
    bp.traceleft = 1;
    bp.bestDiff = 0;
  }
}

#if !defined(__HIP_DEVICE_COMPILE__)
// host twin of homozygous_kernel (tests/emu): 1 on success, 0 / -1 for the reference's two failure messages
inline int homozygous_breakpoint(const uint8_t* row0, const uint8_t* row1, uint32_t L, BreakpointOut& bp) {
  auto chunk = [&](uint64_t b) {
    HomChunk c{0, 0};
    for (uint32_t l = 0; l < 64; ++l) {
      const uint64_t j = b + l;
      if (j >= L) break;
      if (row0[j] != row1[j]) c.mm |= 1ull << l;
      if (row0[j] != '-') c.ng |= 1ull << l;
    }
    return c;
  };
  int64_t align_start = 0, align_end = 0;
  for (int64_t j = 0; j < (int64_t)L; ++j)
    if (row0[j] != '-' && row1[j] != '-') { align_start = j; break; }
  for (int64_t j = (int64_t)L - 1; j >= 0; --j)
    if (row0[j] != '-' && row1[j] != '-') { align_end = j; break; }
  uint32_t lo = 0, hi = 0;
  const int rc = hom_range(align_start, align_end, bp, lo, hi);
  if (rc != 1) return rc;
  float F = 0.0f;
  uint32_t var_end = 0;
  HomPick best;
  hom_pick_init(best);
  for (int pass = 0; pass < 2; ++pass) {
    HomChunk cur = chunk(0), nxt = chunk(64);
    uint64_t prev = 0;
    uint32_t vbase = 0;
    for (uint64_t b = 0; b < hi; b += 64) {
      const HomChunk nn = chunk(b + 128);
      for (uint32_t lane = 0; lane < 64; ++lane) {
        const uint64_t i = b + lane;
        if (i < lo || i >= hi) continue;
        const HomLane h = hom_lane(prev, cur, nxt.mm, vbase, lane);
        if (pass == 0) { if (h.diff > 0.0 && h.g > F) F = h.g; }
        else hom_pick(best, h, F, (uint32_t)i);  // one HomPick for all lanes: columns ascend, so first / last come out the same
      }
      if (pass == 0) var_end += hom_bases_below(cur, b, hi);
      vbase += (uint32_t)popc64(cur.ng);
      prev = cur.mm; cur = nxt; nxt = nn;
    }
    if (pass == 0) {
      hom_finish(bp, F, var_end);
      if (!bp.indelshift) return 1;
    }
  }
  bp.breakpoint = best.last ? best.last_var : best.first_var;
  bp.traceleft = best.last ? best.last_tl : best.first_tl;
  bp.bestDiff = F;
  return 1;
}
#endif

// generateSecondaryDecomposed (decompose.h:378-410), one position.  A, Cc, G, T: the four channels at the basecall's peak position
// (traceACGT[k][bcPos[i]]: one entry of the peak table, below)
TR_HD uint8_t secondary_decomposed(uint8_t p, uint8_t s, int32_t A, int32_t Cc, int32_t G, int32_t T) {
  if (p == s) return p;
  if (s == 'A' || s == 'C' || s == 'G' || s == 'T') return s;
  switch (s) {
    case 'R': return A > G ? 'A' : 'G';
    case 'Y': return Cc > T ? 'C' : 'T';
    case 'S': return Cc > G ? 'C' : 'G';
    case 'W': return A > T ? 'A' : 'T';
    case 'K': return G > T ? 'G' : 'T';
    case 'M': return A > Cc ? 'A' : 'C';
    default: return 'N';
  }
}
// ... reading the chromatogram itself (the host emulator's entry point; the kernels read the peak table)
TR_HD uint8_t secondary_decomposed(uint8_t p, uint8_t s, const int32_t* trace, uint64_t nsamples, int32_t pos) {
  if (p == s) return p;
  return secondary_decomposed(p, s, trace[pos], trace[nsamples + pos], trace[2 * nsamples + pos], trace[3 * nsamples + pos]);
}

}  // namespace tracyhip
#endif
