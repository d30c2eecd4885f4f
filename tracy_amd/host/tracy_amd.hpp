// tracy_amd.hpp -- C++ host mirror of the reference's interface for the hot path.
//
// The reference's boundary is a set of header-only templates (no FFI).  This header keeps their names,
// argument order and meaning so that reference call sites compile against it unchanged, and forwards
// to the C ABI (include/tracy_hip.h) with a batch of one.  The batch drivers call the C ABI directly;
// this layer exists for the single-trace CLI path.
//
//   tracy_amd::DnaScore<T>            <- align.h:11-32
//   tracy_amd::AlignConfig<H,V>       <- align.h:37-50
//   tracy_amd::gotohScore / gotoh     <- gotoh.h:12-68 / 71-174   (std::string or Profile arguments)
//   tracy_amd::needleScore / needle   <- needle.h:12-57 / 59-138
//   tracy_amd::findBreakpoint         <- decompose.h:7-56
//   tracy_amd::findHomozygousBreakpoint     <- decompose.h:59-128
//   tracy_amd::decomposeAlleles             <- decompose.h:179-376
//   tracy_amd::generateSecondaryDecomposed  <- decompose.h:378-410
//   tracy_amd::allelicFraction              <- decompose.h:412-621
//   tracy_amd::trimReferenceSlice           <- fmindex.h:429-463
//   tracy_amd::TraceBreakpoint, ReferenceSlice  <- fmindex.h:51-56, 28-37
// so that the hot section of indigo.h:314-387 compiles against this header as it stands (tests/cpp/test_mirror.cpp).
//
// Errors: the reference's DP functions cannot fail; here a failing device call throws std::runtime_error
// with tracyhip_last_error() (there is no CPU fallback to return to).
#ifndef TRACY_AMD_HPP
#define TRACY_AMD_HPP

#include <iostream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/tracy_hip.h"
#include "tracy_host.hpp"

namespace tracy_amd {

template <typename TScoreValue>
struct DnaScore {
  typedef TScoreValue TValue;
  TScoreValue match, mismatch, go, ge, inf;
  DnaScore() : match(5), mismatch(-4), go(-10), ge(-1), inf(1000000) {}
  DnaScore(TScoreValue m, TScoreValue mm, TScoreValue gapopen, TScoreValue gapext) : match(m), mismatch(mm), go(gapopen), ge(gapext), inf(1000000) {}
};

template <bool THorizontal = false, bool TVertical = false>
class AlignConfig {};

// boost::multi_array<char,2> stand-in for the alignment output: rows x columns of characters
struct Alignment {
  std::vector<std::string> rows;
  std::string const& operator[](std::size_t i) const { return rows[i]; }
  std::string& operator[](std::size_t i) { return rows[i]; }
  std::size_t shape(int d) const { return d == 0 ? rows.size() : (rows.empty() ? 0 : rows[0].size()); }
};

struct TraceBreakpoint {  // fmindex.h:51-56
  bool indelshift;
  bool traceleft;
  uint32_t breakpoint;
  float bestDiff;
};

namespace detail {
inline tracyhip_ctx* context() {
  static tracyhip_ctx* ctx = nullptr;
  if (!ctx) {
    if (tracyhip_create(0, &ctx) != TRACYHIP_OK) throw std::runtime_error(std::string("tracy_amd: ") + tracyhip_last_error());
  }
  return ctx;
}
inline void check(int rc) {
  if (rc != TRACYHIP_OK) throw std::runtime_error(std::string("tracy_amd: ") + tracyhip_last_error());
}
inline tracyhip_seqset seqset(std::string const& s, uint64_t* off, uint32_t* len) {
  *off = 0;
  *len = (uint32_t)s.size();
  return tracyhip_seqset{TRACYHIP_SEQ_CHAR, s.data(), off, len, 1};
}
inline tracyhip_seqset seqset(Profile const& p, uint64_t* off, uint32_t* len) {
  *off = 0;
  *len = (uint32_t)p.cols;
  return tracyhip_seqset{TRACYHIP_SEQ_PROFILE, p.data(), off, len, 1};
}
template <bool H, bool V, typename TScore>
inline tracyhip_params params(AlignConfig<H, V> const&, TScore const& sc) {
  return tracyhip_params{(int32_t)sc.match, (int32_t)sc.mismatch, (int32_t)sc.go, (int32_t)sc.ge, H ? 1 : 0, V ? 1 : 0};
}
inline std::size_t length(std::string const& s) { return s.size(); }
inline std::size_t length(Profile const& p) { return p.cols; }

template <bool NEEDLE, typename T1, typename T2, typename TCfg, typename TScore>
inline int score_one(T1 const& a1, T2 const& a2, TCfg const& ac, TScore const& sc) {
  uint64_t o1, o2;
  uint32_t l1, l2;
  tracyhip_pairs pr{1, seqset(a1, &o1, &l1), seqset(a2, &o2, &l2), nullptr, nullptr};
  tracyhip_params p = params(ac, sc);
  int32_t s = 0;
  check(NEEDLE ? tracyhip_needle_score(context(), &pr, &p, TRACYHIP_MEM_HOST, &s)
               : tracyhip_gotoh_score(context(), &pr, &p, TRACYHIP_MEM_HOST, &s));
  return s;
}
template <bool NEEDLE, typename T1, typename T2, typename TCfg, typename TScore>
inline int align_one(T1 const& a1, T2 const& a2, Alignment& align, TCfg const& ac, TScore const& sc) {
  uint64_t o1, o2, ooff = 0;
  uint32_t l1, l2, olen = 0;
  tracyhip_pairs pr{1, seqset(a1, &o1, &l1), seqset(a2, &o2, &l2), nullptr, nullptr};
  tracyhip_params p = params(ac, sc);
  int32_t s = 0;
  std::vector<uint8_t> ops(length(a1) + length(a2) + 1);
  check(NEEDLE ? tracyhip_needle_align(context(), &pr, &p, TRACYHIP_MEM_HOST, &s, ops.data(), &ooff, &olen)
               : tracyhip_gotoh_align(context(), &pr, &p, TRACYHIP_MEM_HOST, &s, ops.data(), &ooff, &olen));
  std::vector<uint8_t> r0(olen + 1), r1(olen + 1);
  check(tracyhip_alignment_rows(context(), &pr, TRACYHIP_MEM_HOST, ops.data(), &ooff, &olen, r0.data(), r1.data()));
  align.rows.assign(2, std::string());
  align.rows[0].assign(r0.begin(), r0.begin() + olen);
  align.rows[1].assign(r1.begin(), r1.begin() + olen);
  return s;
}
}  // namespace detail

template <typename TAlign1, typename TAlign2, typename TAlignConfig, typename TScoreObject>
inline int gotohScore(TAlign1 const& a1, TAlign2 const& a2, TAlignConfig const& ac, TScoreObject const& sc) {
  return detail::score_one<false>(a1, a2, ac, sc);
}
template <typename TAlign1, typename TAlign2, typename TAlignConfig, typename TScoreObject>
inline int gotoh(TAlign1 const& a1, TAlign2 const& a2, Alignment& align, TAlignConfig const& ac, TScoreObject const& sc) {
  return detail::align_one<false>(a1, a2, align, ac, sc);
}
template <typename TAlign1, typename TAlign2, typename TAlignConfig>
inline int gotoh(TAlign1 const& a1, TAlign2 const& a2, Alignment& align, TAlignConfig const& ac) {  // gotoh.h:176-182
  DnaScore<int> dnasc;
  return gotoh(a1, a2, align, ac, dnasc);
}
template <typename TAlign1, typename TAlign2>
inline int gotoh(TAlign1 const& a1, TAlign2 const& a2, Alignment& align) {  // gotoh.h:184-190
  AlignConfig<false, false> ac;
  return gotoh(a1, a2, align, ac);
}
template <typename TAlign1, typename TAlign2, typename TAlignConfig, typename TScoreObject>
inline int needleScore(TAlign1 const& a1, TAlign2 const& a2, TAlignConfig const& ac, TScoreObject const& sc) {
  return detail::score_one<true>(a1, a2, ac, sc);
}
template <typename TAlign1, typename TAlign2, typename TAlignConfig, typename TScoreObject>
inline int needle(TAlign1 const& a1, TAlign2 const& a2, Alignment& align, TAlignConfig const& ac, TScoreObject const& sc) {
  return detail::align_one<true>(a1, a2, align, ac, sc);
}

// findBreakpoint(ptrace, bp), decompose.h:7-56
inline void findBreakpoint(Profile const& ptrace, TraceBreakpoint& bp) {
  uint64_t off;
  uint32_t len;
  tracyhip_seqset s = detail::seqset(ptrace, &off, &len);
  tracyhip_breakpoint out;
  detail::check(tracyhip_find_breakpoint(detail::context(), &s, TRACYHIP_MEM_HOST, &out));
  bp.indelshift = out.indelshift != 0;
  bp.traceleft = out.traceleft != 0;
  bp.breakpoint = out.breakpoint;
  bp.bestDiff = out.best_diff;
}

namespace detail {
// Trace + BaseCalls of one trace as the C ABI's flattened batch of one (abif.h:28-57)
struct OneTrace {
  std::vector<int32_t> signal, bcpos;
  std::vector<uint8_t> primary, secondary;
  uint64_t sig_off = 0, bc_off = 0;
  uint32_t nsamples = 0, bc_len = 0;
  tracyhip_basecalls b{};
  OneTrace(Trace const* tr, BaseCalls const& bc) {
    if (tr) {
      nsamples = tr->traceACGT.empty() ? 0 : (uint32_t)tr->traceACGT[0].size();
      signal.resize(4 * (std::size_t)nsamples + 1);
      for (int k = 0; k < 4; ++k)
        for (uint32_t i = 0; i < nsamples; ++i) signal[(std::size_t)k * nsamples + i] = (int32_t)tr->traceACGT[k][i];
    } else {
      signal.assign(1, 0);
    }
    bc_len = (uint32_t)bc.primary.size();
    bcpos.assign(bc.bcPos.begin(), bc.bcPos.end());
    bcpos.resize(bc_len + 1, 0);
    primary.assign(bc.primary.begin(), bc.primary.end());
    secondary.assign(bc.secondary.begin(), bc.secondary.end());
    secondary.resize(bc_len + 1, 'N');
    primary.resize(bc_len + 1, 'N');
    b = tracyhip_basecalls{1, signal.data(), &sig_off, &nsamples, bcpos.data(), primary.data(), secondary.data(), &bc_off, &bc_len};
  }
};
template <typename TAlign>
inline void rows_of(TAlign const& align, std::vector<uint8_t>& r0, std::vector<uint8_t>& r1, uint32_t& len) {
  len = (uint32_t)align.shape(1);
  r0.assign(len + 1, 0);
  r1.assign(len + 1, 0);
  for (uint32_t j = 0; j < len; ++j) { r0[j] = (uint8_t)align[0][j]; r1[j] = (uint8_t)align[1][j]; }
}
}  // namespace detail

// findHomozygousBreakpoint(align, bp), decompose.h:59-128: false + the reference's stderr text when the alignment is unusable
template <typename TAlign>
inline bool findHomozygousBreakpoint(TAlign& align, TraceBreakpoint& bp) {
  std::vector<uint8_t> r0, r1;
  uint32_t len;
  detail::rows_of(align, r0, r1, len);
  uint64_t off = 0;
  tracyhip_breakpoint b{0, bp.traceleft ? 1 : 0, bp.breakpoint, bp.bestDiff};  // indelshift 0: "look at this trace" (indigo.h:314-317)
  int32_t status = 0;
  detail::check(tracyhip_find_homozygous_breakpoint(detail::context(), 1, r0.data(), r1.data(), &off, &len, TRACYHIP_MEM_HOST, &b, &status));
  if (status == 0) { std::cerr << "No valid alignment found between consensus and reference!" << std::endl; return false; }
  if (status < 0) { std::cerr << "Alignment too short between consensus and reference!" << std::endl; return false; }
  bp.indelshift = b.indelshift != 0;
  bp.traceleft = b.traceleft != 0;
  bp.breakpoint = b.breakpoint;
  bp.bestDiff = b.best_diff;
  return true;
}

// decomposeAlleles(c, align, bc, bp, rs, dcp), decompose.h:179-376.  TConfig needs trimLeft, trimRight, maxindel, madc (IndigoConfig,
// indigo.h:16-40); bc.primary / bc.secondary are rewritten, dcp receives the (indel, error) table, the two stdout lines of
// decompose.h:315 / :327 are printed as the reference prints them.  Always returns true, as the reference does.
template <typename TConfig, typename TAlign, typename TDecomp>
inline bool decomposeAlleles(TConfig const& c, TAlign const& align, BaseCalls& bc, TraceBreakpoint bp, ReferenceSlice& rs, TDecomp& dcp) {
  std::vector<uint8_t> r0, r1;
  uint32_t len;
  detail::rows_of(align, r0, r1, len);
  detail::OneTrace ot(nullptr, bc);
  uint64_t off = 0, doff = 0;
  const uint32_t rlen = (uint32_t)rs.refslice.size();
  tracyhip_breakpoint b{bp.indelshift ? 1 : 0, bp.traceleft ? 1 : 0, bp.breakpoint, bp.bestDiff};
  tracyhip_decomp_params p{(int32_t)c.trimLeft, (int32_t)c.trimRight, (int32_t)c.maxindel, (int32_t)c.madc};
  std::vector<int32_t> di(2 * (std::size_t)p.maxindel + 4), de(2 * (std::size_t)p.maxindel + 4);
  tracyhip_decomp_status st{};
  detail::check(tracyhip_decompose_alleles(detail::context(), &ot.b, r0.data(), r1.data(), &off, &len, &b, &rlen, &p, TRACYHIP_MEM_HOST, di.data(),
                                           de.data(), &doff, &st));
  bc.primary.assign(ot.primary.begin(), ot.primary.begin() + ot.bc_len);
  bc.secondary.assign(ot.secondary.begin(), ot.secondary.begin() + std::min<std::size_t>(ot.bc_len, bc.secondary.size()));
  for (uint32_t k = 0; k < st.dcp_n; ++k) dcp.push_back(std::make_pair(di[k], de[k]));
  if (st.kind == 1)
    std::cout << "Complex mutation, decomposition: ins: " << st.best_ins << ", del: " << st.best_del << ", error: " << st.best_fr << std::endl;
  else if (st.kind == 2)
    std::cout << "No InDel detected, traverse the whole alignment." << std::endl;
  return true;
}

// generateSecondaryDecomposed(tr, bc), decompose.h:378-410
inline void generateSecondaryDecomposed(Trace const& tr, BaseCalls& bc) {
  detail::OneTrace ot(&tr, bc);
  std::vector<uint8_t> sd(ot.bc_len + 1, 0);
  detail::check(tracyhip_secondary_decomposed(detail::context(), &ot.b, TRACYHIP_MEM_HOST, sd.data()));
  bc.secDecompose.assign(sd.begin(), sd.begin() + bc.secondary.size());
}

// allelicFraction(c, tr, bc), decompose.h:412-621
template <typename TConfig>
inline std::pair<double, double> allelicFraction(TConfig const& c, Trace const& tr, BaseCalls const& bc) {
  detail::OneTrace ot(&tr, bc);
  std::vector<uint8_t> sd(bc.secDecompose.begin(), bc.secDecompose.end());
  sd.resize(ot.bc_len + 1, 'N');
  double fr[2] = {0, 0};
  detail::check(tracyhip_allelic_fraction(detail::context(), &ot.b, sd.data(), (uint32_t)c.trimLeft, (uint32_t)c.trimRight, TRACYHIP_MEM_HOST, fr));
  return std::make_pair(fr[0], fr[1]);
}

// trimReferenceSlice(c, align, rs), fmindex.h:429-463: rs.refslice and rs.pos are updated in place
template <typename TConfig, typename TAlign>
inline void trimReferenceSlice(TConfig const& c, TAlign const& align, ReferenceSlice& rs) {
  std::vector<uint8_t> r0, r1;
  uint32_t len;
  detail::rows_of(align, r0, r1, len);
  uint64_t off = 0;
  const uint32_t rlen = (uint32_t)rs.refslice.size();
  const uint8_t fwd = rs.forward ? 1 : 0;
  uint32_t ri = 0, rl = 0, dp = 0;
  detail::check(tracyhip_trim_reference_slice(detail::context(), 1, r0.data(), r1.data(), &off, &len, &rlen, &fwd, (uint32_t)c.trimLeft,
                                              (uint32_t)c.trimRight, TRACYHIP_MEM_HOST, &ri, &rl, &dp));
  // (the reference's "Offset smaller than zero" warning, fmindex.h:457-459, cannot fire: ri + risize never exceeds the reference
  // bases the alignment holds, which are all of rs.refslice)
  rs.refslice = rs.refslice.substr(ri, rl);
  rs.pos += dp;
}

}  // namespace tracy_amd
#endif
