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
//   tracy_amd::TraceBreakpoint        <- fmindex.h:51-56
//
// Errors: the reference's DP functions cannot fail; here a failing device call throws std::runtime_error
// with tracyhip_last_error() (there is no CPU fallback to return to).
#ifndef TRACY_AMD_HPP
#define TRACY_AMD_HPP

#include <stdexcept>
#include <string>
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

}  // namespace tracy_amd
#endif
