// stream_plan.h -- what the stream-ordered pipelines (stream.hip) keep on the device between two launches, and the planning steps
// that used to run on the host between two synchronisations (pipeline.hip) as functions of that state: which strand to prune,
// which band an alignment's score allows, which strip height sweeps it, whether a certificate held.  One thread per trace; the
// arithmetic is the host planner's, line by line (cited where it is), so that a trace takes the same tier on either path -- and a
// trace whose tier the device cannot give it (a failed certificate, a band wider than the band kernels hold) is marked `dead` and
// handed to the host-planned tiers afterwards.  TR_HD: the same functions compile for the host (tests).
#ifndef TRACY_AMD_STREAM_PLAN_H
#define TRACY_AMD_STREAM_PLAN_H

#include "band16.h"
#include "front.h"

namespace tracyhip {

// why a trace left the stream-ordered pass (bit set of SDead; tracyhip_call_stats::fallback_traces counts traces with any)
enum : uint32_t {
  SD_FRONT = 1u,          // pruned sweep of the voted strand not certified in either tier (pipeline.hip: swept in full)
  SD_STRAND = 2u,         // strand by certificate: the loser's bound does not decide (its full sweep is needed)
  SD_LOSER_WON = 4u,      // the strand the vote marked as the likely loser won: its row m was not kept
  SD_JUNK = 8u,           // c_e = 0: the all-gap path is optimal (or an empty pair)
  SD_PRELIM_BAND = 16u,   // preliminary alignment: its score allows a band wider than the band kernels hold
  SD_PRELIM_CHECK = 32u,  // ... or the banded result is not the sweep's (score differs, walk left the band)
  SD_FINAL_BAND = 64u,    // `tracy align`: final alignment outside the band kernels' shapes
  SD_FINAL_CHECK = 128u,  // ... or its band certificate failed
  SD_MEM = 256u,          // traceback words beyond the workspace planned for the launch
  SD_ALLELE_FRONT = 512u, // gotoh(allele, window): pruned sweep not eligible / not certified
  SD_ALLELE_ORIGIN = 1024u,  // its origin band too wide
  SD_ALLELE_BAND = 2048u,    // gotoh(allele, slice): band too wide / ends outside the slice
  SD_ALLELE_CHECK = 4096u,   // ... or the banded result is not S*
  SD_A12_BAND = 8192u,       // allele 1 vs allele 2: no band
  SD_A12_CHECK = 16384u,     // ... or its bound not beaten
  SD_SHAPE = 32768u,         // a sub-window longer than the launch's LDS staging was sized for
};

// counters the planning kernels keep (tracyhip_call_stats, kernel timers): one block of 64-bit words per call
enum : int {
  SC_PRUNED = 0, SC_PRUNED_UNCERT, SC_PRELIM_BANDED, SC_PRELIM_REPEATED, SC_FINAL_BANDED, SC_FINAL_REPEATED,
  SC_ALLELE_PRUNED0, SC_ALLELE_PRUNED1, SC_ALLELE_UNCERT0, SC_ALLELE_UNCERT1,
  SC_ALLELE_BANDED0, SC_ALLELE_BANDED1, SC_ALLELE_BANDED2, SC_ALLELE_REPEATED0, SC_ALLELE_REPEATED1, SC_ALLELE_REPEATED2,
  SC_SWEEP_CELLS, SC_SWEEP_BYTES,  // the combined sweep / prefix launches (TRACYHIP_TIMER_SCORE)
  SC_DECOMP_CELLS, SC_DECOMP_BYTES,
  SC_FRONT_CELLS, SC_FRONT_BYTES,  // the first tier of the pruned sweeps (TRACYHIP_TIMER_FRONT: strips of 8 rows on c* +- 60)
  SC_ALLELE_SHARED,                // allele 2 reading the prefix row allele 1 keeps (s_allele_plan0_kernel)
  SC_COUNT
};
// per band stage (bucket scan): cells and algorithmic bytes of the launch (kernel timers), bytes of its traceback words
// ... and how wide its bands were: pairs by diagonals (dmax - dmin + 1) <= 8, 16, 24, 32, 48, 64, 96, more (option `verbose` prints them)
enum : int { SB_CELLS = 0, SB_BYTES = 1, SB_WORDS = 2, SB_HIST = 4, SB_HIST_N = 8, SB_COUNT = 12 };
TR_HD int s_width_bucket(int32_t dmin, int32_t dmax) {
  const int32_t w = dmax - dmin + 1;
  return w <= 8 ? 0 : w <= 16 ? 1 : w <= 24 ? 2 : w <= 32 ? 3 : w <= 48 ? 4 : w <= 64 ? 5 : w <= 96 ? 6 : 7;
}

enum : uint32_t { SG_FRONT_OK = 1u };

struct SGeom {          // one trace: what the host knows before anything runs
  uint64_t prof_off;    // full profile &P[0][0] (floats)
  uint64_t ref_off;     // reference window (bytes of the payload = codes of the code buffer)
  uint64_t lr_off[2];   // row m / kept row of the forward / reverse-complement sweep (int32 units)
  uint64_t tab_off;     // substitution table of the full profile (int16 units)
  uint64_t ops_off;     // `tracy align`: the final alignment's ops (tracyhip_align_result::ops_offset); `tracy decompose`: ops and rows of the trimmed trace
  uint32_t mf, mt, tl, rn;
  uint32_t tab_stride;
  uint32_t full_a, full_b;  // its two slots in the list of full sweeps (sorted by strip height, then size)
  uint32_t flags;       // SG_FRONT_OK
};
struct SGeomD {         // `tracy decompose`: the rest
  uint64_t bc_off;      // basecalls (primary / secondary / secDecompose / bcPos)
  uint64_t sig_off;
  uint64_t dcp_off;
  uint64_t opsk_off[3];  // allele alignments (tracyhip_decompose_result::ops_offset[k])
  uint64_t atab_off[2];  // substitution tables of the allele strings (int16 units)
  uint64_t alr_off[2];   // kept rows of their prefix sweeps (int32 units)
  uint32_t nsamples, sl, soff, atab_stride;
  uint32_t flags[2];     // SG_FRONT_OK per allele
};

struct TrimRec { uint32_t ri, len, pos, pad; };  // trimReferenceSlice's three numbers (pipe_kernels.h TrimOut)

struct STrace {         // one trace: what the stages leave for each other
  int32_t sc[2];        // gsFwd, gsRev (the loser's may be its certified bound)
  int32_t sstar;        // the winner's: score of the preliminary alignment
  uint32_t ce;          // where that alignment ends on row m (window column, 1-based)
  uint32_t gap;         // gap columns its score allows
  uint32_t shift;       // columns of the window left of the sub-window
  int32_t bw;           // `tracy align`: half width of the final alignment's band
  uint8_t g, cls, rc, fwd;
  TrimRec trim;
};
struct SAllele {        // one allele of one trace (`tracy decompose`)
  int32_t sstar;
  uint32_t ce;
  int64_t gap;
  uint32_t shift;
  uint32_t pad;
  TrimRec trim;
};

struct SParams {        // scoring + switches every planning kernel sees
  int32_t match, mismatch, go, ge;
  uint32_t nt;
  uint32_t exact;       // both orientation scores exact (no strand by certificate)
  uint32_t ncap;        // longest sub-window the band launches staged LDS for
  uint32_t trim_left, trim_right;
  uint32_t use_votes;   // the full sweeps skip row m of the likely loser (DpArgs::votes)
  uint32_t split_prefix;  // the voted strands' prefixes run in a launch of their own beside the full sweeps (timed and credited with the pruned sweep)
};

TR_HD int64_t s_abs64(int32_t x) { return x < 0 ? -(int64_t)x : (int64_t)x; }
TR_HD int64_t s_best(const SParams& p) { const int64_t b = p.match > p.mismatch ? p.match : p.mismatch; return b > 0 ? b : 0; }

// the widening / clamping / rs.pos part of trimReferenceSlice (fmindex.h:443-461)
TR_HD TrimRec s_trim_finish(uint32_t ri, uint32_t risize, uint32_t n, uint32_t trim_left, uint32_t trim_right, bool forward) {
  if (ri >= trim_left) { ri -= trim_left; risize += trim_left; }
  if ((uint32_t)(ri + risize + trim_right) < n) risize += trim_right;
  TrimRec r;
  r.ri = ri;
  r.len = (ri <= n) ? ((risize < n - ri) ? risize : n - ri) : 0;  // substr(ri, risize)
  r.pos = 0;
  if (forward) r.pos = ri;
  else {
    const int32_t offset = (int32_t)n - (int32_t)ri - (int32_t)risize;
    if (offset >= 0) r.pos = (uint32_t)offset;  // negative: the reference only warns (fmindex.h:457-459)
  }
  r.pad = 0;
  return r;
}

// A path that ends at (m, c_e) with score S* and can collect at most `top` on its diagonal steps has at most g = (top - S*) / |ge|
// gap columns: it lies in the columns (a, c_e], a = c_e - m - g - 2 (pipeline.hip, stage 2 of orient_and_align and the allele
// stage), on the diagonals (n' - m) +- (g + 1) of that sub-window of n' = c_e - a columns.
struct SubWindow { int64_t g; uint32_t a, n; int32_t dlo, dhi; int K; };
TR_HD SubWindow s_sub_window(uint32_t m, uint32_t ce, int64_t top, int64_t sstar, int32_t ge) {
  SubWindow s;
  const int64_t age = -(int64_t)ge, loss = top - sstar;
  s.g = loss > 0 ? loss / age : 0;
  int64_t a = (int64_t)ce - (int64_t)m - s.g - 2;
  if (a < 0) a = 0;
  s.a = (uint32_t)a;
  s.n = (uint32_t)((int64_t)ce - a);
  const int64_t gg = s.g < (1 << 20) ? s.g : (1 << 20);
  const int32_t d1 = (int32_t)s.n - (int32_t)m;
  s.dlo = d1 - (int32_t)gg - 1;
  s.dhi = d1 + (int32_t)gg + 1;
  s.K = s.g < (1 << 20) ? b16_pick_k(s.dlo, s.dhi) : 0;
  return s;
}
// LDS of a band launch: the codes of four pairs + the tables (run_band16's staging limit, as the host planners test it)
TR_HD bool s_fits_lds(uint32_t n, int K) { return 4ull * ((n + 7u) & ~3u) + b16_table_bytes(K) <= 60u * 1024u; }

// what the first tier of a pruned sweep over m_rest rows below the kept row and n columns is credited with (run_front_once's sums)
TR_HD uint64_t s_front_cells(uint32_t m_rest) { return (uint64_t)b16_strips(m_rest, 8) * 8u * (8u + 2u * 60u); }
TR_HD uint64_t s_front_bytes(uint32_t m_rest, uint32_t n) { return 12ull * m_rest + m_rest + 2ull * 60 + 4ull * (2 * 60 + 8) + 8ull * n; }

TR_HD PairDesc s_skip_pair(uint32_t out) {
  PairDesc d{};
  d.flags = PAIR_SKIP;
  d.out = out;
  return d;
}

}  // namespace tracyhip
#endif
