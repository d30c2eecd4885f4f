// consensus_out.hpp -- host pieces of `tracy consensus` (two-trace consensus) after the device alignment.
//
// Mirrors of /root/reference/src/consensus.h:
//   consensusFastaOut / consensusFastqOut      :65-92
//   gtLetter                                   :94-171  (genotype-likelihood letter + quality of one column)
//   consLetter                                 :173-187
//   pairwiseConsensus                          :189-238
//   plotClustalPairwise                        :241-329
// boost::math::round rounds half away from zero, as std::round does.  PARITY UNPINNED (consensus.h needs Boost,
// htslib, sdsl); cross-checked by the Python restatement in tests/.
#ifndef TRACY_AMD_CONSENSUS_OUT_HPP
#define TRACY_AMD_CONSENSUS_OUT_HPP

#include <cmath>
#include <iomanip>

#include "sage_out.hpp"

namespace tracy_amd {

struct ConsensusOptions {
  bool computeUnion = true, useIUPAC = false;
  std::string label = "Consensus";
};

constexpr double kSmallestGL = -1000;  // SMALLEST_GL

// one consensus letter and its quality from the six summed class weights (A C G T N -)
inline void gtLetter(ConsensusOptions const& c, double cl[6], std::string& cons, std::vector<uint32_t>& qual) {
  double gl[6];
  double total = 0;
  for (int k = 0; k < 6; ++k) total += cl[k];
  for (int k = 0; k < 6; ++k) {
    cl[k] = total > 0 ? cl[k] / total : 0;
    if (cl[k] > 0) {
      gl[k] = std::log10(cl[k]);
      if (gl[k] < kSmallestGL) gl[k] = kSmallestGL;
    } else gl[k] = kSmallestGL;
  }
  uint32_t best = 0, second = 1;
  if (gl[best] < gl[second]) { best = 1; second = 0; }
  for (uint32_t k = 2; k < 6; ++k) {
    if (gl[k] > gl[best]) { second = best; best = k; }
    else if (gl[k] > gl[second]) second = k;
  }
  const double bestVal = gl[best];
  const bool ambiguous = c.useIUPAC && gl[second] > -1 && best <= 3 && second <= 3;
  for (int k = 0; k < 6; ++k) gl[k] -= bestVal;
  const uint32_t bestPL = (uint32_t)std::round(-10 * gl[best]);
  const uint32_t secondPL = (uint32_t)std::round(-10 * gl[second]);
  double likelihood = std::log10(1 - 1 / (std::pow((double)10, -((double)bestPL / (double)10)) + std::pow((double)10, -((double)secondPL / (double)10))));
  likelihood = likelihood > kSmallestGL ? likelihood : kSmallestGL;
  int32_t gq = (int32_t)std::round(-10 * likelihood);
  if (gq < 0) gq = 0;
  static const char letters[6] = {'A', 'C', 'G', 'T', 'N', '-'};
  cons += ambiguous ? iupac(letters[best], letters[second]) : letters[best];
  qual.push_back((uint32_t)gq);
}

// pairwiseConsensus, consensus.h:189-238: aligned columns combine both profiles, unaligned ones keep the one trace
// (only with computeUnion)
inline void pairwiseConsensus(ConsensusOptions const& c, AlignRows const& al, Profile const& p1, Profile const& p2, std::string& cons,
                              std::vector<uint32_t>& qual) {
  int32_t s1 = 0, s2 = 0;
  for (std::size_t j = 0; j < al.cols(); ++j) {
    const bool g0 = al.row0[j] == '-', g1 = al.row1[j] == '-';
    double cl[6];
    if (g0 || g1) {
      if (!g0) {
        if (c.computeUnion) { for (int k = 0; k < 6; ++k) cl[k] = p1(k, s1); gtLetter(c, cl, cons, qual); }
        ++s1;
      }
      if (!g1) {
        if (c.computeUnion) { for (int k = 0; k < 6; ++k) cl[k] = p2(k, s2); gtLetter(c, cl, cons, qual); }
        ++s2;
      }
    } else {
      for (int k = 0; k < 6; ++k) cl[k] = p1(k, s1) + p2(k, s2);  // float + float, then widened (consensus.h:177)
      gtLetter(c, cl, cons, qual);
      ++s1;
      ++s2;
    }
  }
}

inline void consensusFastaOut(std::ostream& out, ConsensusOptions const& c, std::string const& cons) {
  out << ">" << c.label << std::endl << cons << std::endl;
}

inline void consensusFastqOut(std::ostream& out, ConsensusOptions const& c, std::string const& cons, std::vector<uint32_t> const& qual) {
  out << "@" << c.label << std::endl << cons << std::endl << "+" << std::endl;
  for (uint32_t q : qual) {
    int32_t v = (int32_t)q + 33;
    if (v > 122) v = 122;
    out << (char)v;
  }
  out << std::endl;
}

// plotClustalPairwise, consensus.h:241-329
inline void plotClustalPairwise(std::ostream& out, AlignRows const& al, std::string const& stem1, std::string const& stem2, bool forward,
                                int32_t score, uint32_t linelimit) {
  const uint32_t fald = linelimit + 14;
  auto ungapped = [&](std::string const& row) {
    int32_t count = 0;
    for (char ch : row) {
      if (ch == '-') continue;
      out << ch;
      if ((count + 1) % fald == 0) out << std::endl;
      ++count;
    }
    if (count % fald != 0) out << std::endl;
  };
  auto rule = [&]() {
    out << "#";
    for (uint32_t i = 1; i < fald; ++i) out << "-";
    out << std::endl;
  };
  out << ">" << stem1 << std::endl;
  ungapped(al.row0);
  out << ">" << stem2 << (forward ? " (forward)" : " (reverse)") << std::endl;
  ungapped(al.row1);
  out << std::endl;
  out << "Alignment score: " << score << std::endl;
  rule();
  out << std::endl;
  std::string f1 = stem1.substr(0, 8), f2 = stem2.substr(0, 8);
  f1.resize(8, ' ');
  f2.resize(8, ' ');
  int32_t vi = 1, ri = 1;
  uint32_t blocks = 0;
  const int64_t cols = (int64_t)al.cols();
  for (int64_t s = 0; s < cols; s += linelimit, ++blocks) {
    const int64_t e = std::min<int64_t>(cols, s + linelimit);
    out << f1 << std::setw(5) << vi << ' ';
    for (int64_t j = s; j < e; ++j) { out << al.row0[j]; if (al.row0[j] != '-') ++vi; }
    out << std::endl;
    out << "              ";
    for (int64_t j = s; j < e; ++j) out << (al.row0[j] == al.row1[j] ? "|" : " ");
    out << std::endl;
    out << f2 << std::setw(5) << ri << ' ';
    for (int64_t j = s; j < e; ++j) { out << al.row1[j]; if (al.row1[j] != '-') ++ri; }
    out << std::endl;
    out << std::endl;
  }
  for (uint32_t i = blocks; i < 6; ++i)
    for (uint32_t k = 0; k < 4; ++k) out << std::endl;
  rule();
  rule();
  out << std::endl;
  out << std::endl;
}

}  // namespace tracy_amd
#endif
