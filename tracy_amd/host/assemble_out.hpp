// assemble_out.hpp -- host pieces of `tracy assemble` around the device alignments: hard trimming of the basecalls,
// reverse complement of a whole chromatogram, the per-row JSON record of the multiple alignment.
//
// Mirrors of /root/reference/src:
//   trimTrace(tr, bc, trimLeft, trimRight, nbc)   trim.h:75-98    (hard trim, keeps every sampling position)
//   reverseComplement(char)                       trim.h:101-121
//   reverseComplementTrace                        trim.h:123-149
//   alignedTraceByRow                             json.h:220-246
// PARITY UNPINNED (trim.h / json.h need Boost + htslib); cross-checked by the Python restatement in tests/.
#ifndef TRACY_AMD_ASSEMBLE_OUT_HPP
#define TRACY_AMD_ASSEMBLE_OUT_HPP

#include "msa.hpp"
#include "sage_out.hpp"

namespace tracy_amd {

// the basecalls [trimLeft, size - trimRight) that the sample loop reaches, in sample order
inline void trimTrace(Trace const& tr, BaseCalls const& bc, uint32_t trimLeft, uint32_t trimRight, BaseCalls& nbc) {
  const uint32_t len = (uint32_t)bc.primary.size() - trimRight;
  uint32_t call = 0;
  int32_t next = bc.bcPos[0];
  const int32_t ns = (int32_t)tr.traceACGT[0].size();
  for (int32_t x = 0; x < ns; ++x) {
    if (next != x) continue;
    if (call >= trimLeft && call < len) {
      nbc.bcPos.push_back(x);
      nbc.primary.push_back(bc.primary[call]);
      nbc.secondary.push_back(bc.secondary[call]);
      nbc.consensus.push_back(bc.consensus[call]);
      nbc.estQual.push_back(bc.estQual[call]);
    }
    if (call < bc.bcPos.size() - 1) next = bc.bcPos[++call];
  }
}

inline char reverseComplementChar(char c) {  // reverseComplement(char), trim.h:101-121 (IUPAC aware)
  static const char from[] = "ACGTNHVMYDBKRUSW", to[] = "TGCANDBKRHVMYASW";
  for (int i = 0; from[i]; ++i)
    if (c == from[i]) return to[i];
  return c;
}

// mirror the samples, swap the channels A<->T and C<->G, complement the calls (trim.h:123-149)
inline void reverseComplementTrace(Trace const& tr, BaseCalls const& bc, Trace& ntr, BaseCalls& nbc) {
  uint32_t call = (uint32_t)bc.bcPos.size() - 1;
  int32_t next = bc.bcPos[call];
  ntr.traceACGT.assign(4, Trace::TMountains());
  int32_t out = 0;
  for (int32_t x = (int32_t)tr.traceACGT[0].size(); x > 0; --x, ++out) {
    if (next == x - 1) {
      nbc.bcPos.push_back(out);
      nbc.primary.push_back(reverseComplementChar(bc.primary[call]));
      nbc.secondary.push_back(reverseComplementChar(bc.secondary[call]));
      nbc.consensus.push_back(reverseComplementChar(bc.consensus[call]));
      nbc.estQual.push_back(bc.estQual[call]);
      ntr.qual.push_back(call < tr.qual.size() ? tr.qual[call] : 0);  // the reference indexes tr.qual unchecked
      if (call > 0) next = bc.bcPos[--call];
    }
    for (int k = 0; k < 4; ++k) ntr.traceACGT[3 - k].push_back(tr.traceACGT[k][x - 1]);
  }
}

// one row of the multiple alignment as a JSON object, flanking gaps stripped and counted (json.h:220-246)
inline void alignedTraceByRow(std::ostream& out, CharAlign const& align, uint32_t row, std::string const& traceFileName, bool forward, bool ref) {
  std::string const& r = align[row];
  uint32_t leading = 0, trailing = 0;
  bool in_lead = true;
  for (char ch : r) {
    if (in_lead) {
      if (ch == '-') ++leading;
      else in_lead = false;
    }
    trailing = ch != '-' ? 0 : trailing + 1;
  }
  out << "{" << std::endl;
  out << "\"reference\": " << (ref ? "true" : "false") << "," << std::endl;
  out << "\"forward\": " << (forward ? "true" : "false") << "," << std::endl;
  out << "\"traceFileName\": \"" << traceFileName << "\"," << std::endl;
  out << "\"leadingGaps\": \"" << leading << "\"," << std::endl;
  out << "\"trailingGaps\": \"" << trailing << "\"," << std::endl;
  out << "\"align\": \"";
  for (uint32_t j = leading; j < r.size() - trailing; ++j) out << r[j];
  out << "\"" << std::endl;
  out << "}" << std::endl;
}

}  // namespace tracy_amd
#endif
