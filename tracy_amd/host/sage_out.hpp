// sage_out.hpp -- host pieces of `tracy align` around the device pipeline: reference loading, the
// quality-based trim estimate, and the four output files.
//
// Mirrors of /root/reference/src (same names, argument meaning and file contents):
//   ReferenceSlice                       fmindex.h:28-37
//   reverseComplement(std::string&)      fmindex.h:11-25
//   genomeType                           fmindex.h:58-71
//   loadSingleFasta (+ name/IUPAC fixes) fasta.h:16-95
//   nearestSNP / trimTrace               trim.h:10-73
//   plotAlignment                        fmindex.h:329-427
//   assemblyTrace                        json.h:108-194
//   traceAlignJsonOut                    json.h:197-217
//   alignmentTracePadding                json.h:383-472
// The reference headers behind these need Boost/htslib/sdsl and cannot be compiled in this image:
// PARITY UNPINNED (careful restatement, cross-checked by an independent Python restatement in tests/).
// An alignment is passed as its two gapped rows (what boost::multi_array<char,2> align[0], align[1] hold).
#ifndef TRACY_AMD_SAGE_OUT_HPP
#define TRACY_AMD_SAGE_OUT_HPP

#include <fcntl.h>
#include <unistd.h>

#include "text_buf.hpp"
#include <algorithm>
#include <cctype>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "trace_io.hpp"
#include "tracy_host.hpp"

namespace tracy_amd {

constexpr int32_t kEmptyTraceSignal = -99;     // EMPTY_TRACE_SIGNAL, json.h:12-14
constexpr std::size_t kMaxSingleFasta = 50000;  // MAX_SINGLE_FASTA_SIZE, fasta.h:10-12

// gapped rows of one pairwise alignment
struct AlignRows {
  std::string row0, row1;
  std::size_t cols() const { return row0.size(); }
};

// reverseComplement, fmindex.h:11-25: complements the upper-cased reversed string into `sequence`; for
// a letter outside ACGTN the reference leaves that output position untouched, i.e. the ORIGINAL byte stays.
inline void reverseComplement(std::string& sequence) {
  // (a 256-entry table instead of toupper + switch per letter: the windows of an indexed genome are reverse-complemented per trace,
  // 4 kb each, on the host threads that seed -- a fifth of the seeding time went here)
  static const struct Table {
    unsigned char t[256];
    Table() {
      std::memset(t, 0, sizeof(t));
      const char* from = "ACGTNacgtn";
      const char* to = "TGCANTGCAN";
      for (int i = 0; from[i]; ++i) t[(unsigned char)from[i]] = (unsigned char)to[i];
    }
  } tab;
  const std::string src = sequence;
  const std::size_t n = src.size();
  for (std::size_t i = 0; i < n; ++i) {
    const unsigned char c = tab.t[(unsigned char)src[n - 1 - i]];
    if (c) sequence[i] = (char)c;  // (any other letter: the output position keeps its ORIGINAL byte)
  }
}

// genomeType, fmindex.h:58-71
inline int32_t genomeType(std::string const& path) {
  unsigned char magic[4] = {0, 0, 0, 0};
  const int fd = ::open(path.c_str(), O_RDONLY | O_CLOEXEC);
  if (fd < 0) return -1;
  const ssize_t got = ::read(fd, magic, 4);
  ::close(fd);
  if (magic[0] == 0x1f && magic[1] == 0x8b) return 0;
  if (std::memcmp(magic, "TAMD", 4) == 0) return 0;  // an index written by `tracy_amd_cli index` (seed.hpp)
  if (got == 4 && (std::memcmp(magic, "ABIF", 4) == 0 || std::memcmp(magic, ".scf", 4) == 0)) return 2;  // traceFormat(path) >= 0
  if (magic[0] == '>') return 1;
  return -1;
}

// loadSingleFasta, fasta.h:54-95: one record, upper-cased, IUPAC codes -> N, anything else is an error;
// the record name loses the characters VCF dislikes (fasta.h:16-35)
inline bool loadSingleFasta(std::string const& filename, std::string& faname, std::string& seq) {
  faname.clear();
  // what a letter of the record becomes: itself (A C G T N, either case), 'N' (the other IUPAC codes), 0 = not a nucleotide code
  static const struct Letters {
    char t[256];
    Letters() {
      std::memset(t, 0, sizeof(t));
      for (const char* c = "ACGTN"; *c; ++c) { t[(unsigned char)*c] = *c; t[(unsigned char)(*c + 32)] = *c; }
      for (const char* c = "WSMKRYBDHV"; *c; ++c) { t[(unsigned char)*c] = 'N'; t[(unsigned char)(*c + 32)] = 'N'; }
    }
  } letters;
  std::string body;
  bool foreign = false;
  detail::FileBytes f;
  if (f.load(filename)) {
    const char* p = f.chars();
    const char* const end = p + f.size();
    body.reserve(f.size());
    while (p < end) {  // the lines std::getline would hand out: split at '\n', a last line without one included, empty ones skipped
      const char* nl = static_cast<const char*>(std::memchr(p, '\n', (std::size_t)(end - p)));
      const char* le = nl ? nl : end;
      if (le > p) {
        const std::size_t len = (std::size_t)(le - p) - (le[-1] == '\r' ? 1 : 0);
        if (*p == '>') {
          if (!faname.empty()) {
            std::cerr << "Only single-chromosome FASTA files are supported." << std::endl;
            return false;
          }
          faname.assign(p + 1, len ? len - 1 : 0);
        } else {
          for (std::size_t i = 0; i < len; ++i) {
            const char c = letters.t[(unsigned char)p[i]];
            foreign = foreign || c == 0;
            body.push_back(c);
          }
        }
      }
      p = nl ? nl + 1 : end;
    }
  }
  if (foreign) {
    std::cerr << "FASTA file contains non-IUPAC characters." << std::endl;
    return false;
  }
  seq += body;
  static const std::string banned = "\\,'\"()[]{}<>:\t\r#";
  faname.erase(std::remove_if(faname.begin(), faname.end(), [](char c) { return banned.find(c) != std::string::npos; }), faname.end());
  return true;
}

// nearestSNP, trim.h:10-33 (used by the decompose driver) -- first heterozygous call around rtp
inline uint32_t nearestSNP(uint32_t trimLeft, uint32_t trimRight, BaseCalls const& bc, uint32_t rtp) {
  for (uint32_t offset = 0;; ++offset) {
    bool alive = false;
    if (rtp + offset + trimRight < bc.secondary.size() && rtp + offset + trimRight < bc.primary.size()) {
      if (trimLeft < rtp + offset && bc.primary[rtp + offset] != bc.secondary[rtp + offset]) return rtp + offset - trimLeft;
      alive = true;
    }
    if (offset + trimLeft < rtp) {
      if (bc.primary[rtp - offset] != bc.secondary[rtp - offset]) return rtp - offset - trimLeft;
      alive = true;
    }
    if (!alive) break;
  }
  return rtp > trimLeft ? rtp - trimLeft : trimLeft;
}

// trimTrace, trim.h:35-73: walk outwards from the cleanest stretch until the windowed penalty exceeds
// stringency x (mean penalty of that stretch)
inline void trimTrace(float trimStringency, BaseCalls const& bc, uint32_t& leftTrim, uint32_t& rightTrim) {
  const uint32_t win = 10;
  const uint32_t n = (uint32_t)bc.secondary.size();
  std::vector<int32_t> penalty(n, 0);
  const std::pair<uint32_t, double> best = findBestTraceSection(bc, penalty, win);
  const uint32_t centre = best.first;
  const double limit = (trimStringency * best.second) * win;
  auto window_sum = [&]() {
    double s = 0;
    for (uint32_t i = centre; i < centre + win && i < n; ++i) s += penalty[i];
    return s;
  };
  rightTrim = n;
  leftTrim = 0;
  double local = window_sum();
  for (uint32_t i = centre; i + win < n; ++i) {
    local -= penalty[i];
    local += penalty[i + win];
    if (local > limit) { rightTrim = i; break; }
  }
  local = window_sum();
  for (int32_t i = (int32_t)(centre - 1); i >= 0; --i) {
    if ((uint32_t)(i + win) < n) local -= penalty[i + win];
    local += penalty[i];
    if (local > limit) { leftTrim = i + win - 1; break; }
  }
  rightTrim = rightTrim < n ? n - rightTrim : 0;
}

// plotAlignment, fmindex.h:329-427.  key 0: ">Alt" vs ">Ref"; 1 / 2: allele 1 / 2 vs reference with the
// allelic fraction in the header; 3: allele 1 vs allele 2.
template <class Out, typename std::enable_if<!std::is_same<Out, std::string>::value, int>::type = 0>  // (a stream, not a file name)
inline void plotAlignment(Out& out, AlignRows const& al, ReferenceSlice const& rs, int32_t key, int32_t score,
                          std::pair<double, double> const& a1a2, uint32_t linelimit) {
  const int64_t cols = (int64_t)al.cols();
  int32_t ri = rs.pos + 1;
  const int32_t riend = rs.pos + rs.refslice.size();
  int32_t vi = 1;
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
  // position of reference base `r` (1-based, forward numbering) when the slice is the reverse complement
  auto mirrored = [&](int32_t r) { return rs.pos + rs.refslice.size() - (r - rs.pos) + 1; };

  if (key == 0) out << ">Alt" << std::endl;
  else if (key == 2) out << ">Alt2 (Estimated allelic Fraction: " << a1a2.second << ")" << std::endl;
  else out << ">Alt1 (Estimated allelic Fraction: " << a1a2.first << ")" << std::endl;
  ungapped(al.row0);
  if (key == 3) out << ">Alt2 (Estimated allelic Fraction: " << a1a2.second << ")" << std::endl;
  else if (rs.forward) out << ">Ref " << rs.chr << ":" << ri << "-" << riend << " forward" << std::endl;
  else out << ">Ref " << rs.chr << ":" << mirrored(riend) << "-" << mirrored(ri) << " reversecomplement" << std::endl;
  ungapped(al.row1);
  out << std::endl;
  out << "Alignment score: " << score << std::endl;
  rule();
  out << std::endl;

  uint32_t blocks = 0;
  for (int64_t s = 0; s < cols; s += linelimit, ++blocks) {
    const int64_t e = std::min<int64_t>(cols, s + linelimit);
    if (key != 3) out << "Alt" << std::setw(10) << vi << ' ';
    else out << "Alt1" << std::setw(9) << vi << ' ';
    for (int64_t j = s; j < e; ++j) {
      out << al.row0[j];
      if (al.row0[j] != '-') ++vi;
    }
    out << std::endl;
    out << "              ";
    for (int64_t j = s; j < e; ++j) out << (al.row0[j] == al.row1[j] ? "|" : " ");
    out << std::endl;
    if (key == 3) out << "Alt2" << std::setw(9) << ri << ' ';
    else if (rs.forward) out << "Ref" << std::setw(10) << ri << ' ';
    else out << "Ref" << std::setw(10) << mirrored(ri) << ' ';
    for (int64_t j = s; j < e; ++j) {
      out << al.row1[j];
      if (al.row1[j] != '-') ++ri;
    }
    out << std::endl;
    out << std::endl;
  }
  for (uint32_t i = blocks; i < 6; ++i)  // spacer for small alignments
    for (uint32_t k = 0; k < 4; ++k) out << std::endl;
  rule();
  rule();
  out << std::endl;
  out << std::endl;
}

inline void plotAlignment(std::string const& filename, AlignRows const& al, ReferenceSlice const& rs, int32_t score, uint32_t linelimit) {
  TextBuf out;
  plotAlignment(out, al, rs, 0, score, std::make_pair(0.0, 0.0), linelimit);
  out.write(filename);
}

// a trace re-sampled along an alignment carries its flanking gap counts (Trace::leadingGaps / trailingGaps)
struct PaddedTrace {
  Trace tr;
  BaseCalls bc;
  uint32_t leadingGaps = 0;
  uint32_t trailingGaps = 0;
};

// alignmentTracePadding, json.h:383-472: every internal gap run of `row` inserts gap-many pseudo calls
// ('-', quality 0) of `step` empty samples each, half way between the neighbouring peaks
inline void alignmentTracePadding(std::string const& row, Trace const& tr, BaseCalls const& bc, PaddedTrace& out) {
  uint32_t step = 6;
  if (bc.bcPos.size() > 1) {
    double avg = 0;
    for (uint32_t i = 1; i < bc.bcPos.size(); ++i) avg += (bc.bcPos[i] - bc.bcPos[i - 1]);
    avg /= (bc.bcPos.size() - 1);
    step = (uint32_t)avg;
  }
  std::vector<uint32_t> at, len;
  uint32_t called = 0, run = 0;
  out.leadingGaps = 0;
  for (char ch : row) {
    if (ch == '-') { ++run; continue; }
    if (run) {
      if (called) {
        at.push_back((uint32_t)((bc.bcPos[called - 1] + bc.bcPos[called]) / 2.0));
        len.push_back(run);
      } else out.leadingGaps = run;
      run = 0;
    }
    ++called;
  }
  out.trailingGaps = run;

  Trace& ntr = out.tr;
  BaseCalls& nbc = out.bc;
  ntr.traceACGT.resize(4);
  uint32_t call = 0, offset = 0, ins = 0;
  int32_t next_call = bc.bcPos[0];
  int32_t next_ins = at.empty() ? -1 : (int32_t)at[0];
  const int32_t ns = (int32_t)tr.traceACGT[0].size();
  {
    std::size_t gaps = 0;
    for (uint32_t l : len) gaps += l;
    for (int k = 0; k < 4; ++k) ntr.traceACGT[k].reserve(ntr.traceACGT[k].size() + (std::size_t)ns + gaps * step);
    const std::size_t ncalls = bc.bcPos.size() + gaps;
    nbc.bcPos.reserve(ncalls); nbc.estQual.reserve(ncalls); nbc.primary.reserve(ncalls); nbc.secondary.reserve(ncalls); nbc.consensus.reserve(ncalls);
  }
  for (int32_t x = 0; x < ns; ++x) {
    for (int k = 0; k < 4; ++k) ntr.traceACGT[k].push_back(tr.traceACGT[k][x]);
    if (next_ins == x) {
      for (uint32_t g = 0; g < len[ins]; ++g) {
        nbc.bcPos.push_back((int32_t)(x + offset + (uint32_t)(step / 2.0)));
        nbc.estQual.push_back(0);
        nbc.primary.push_back('-');
        nbc.secondary.push_back('-');
        nbc.consensus.push_back('-');
        for (uint32_t s = 0; s < step; ++s, ++offset)
          for (int k = 0; k < 4; ++k) ntr.traceACGT[k].push_back(kEmptyTraceSignal);
      }
      if (ins < at.size() - 1) next_ins = (int32_t)at[++ins];
    }
    if (next_call == x) {
      nbc.bcPos.push_back((int32_t)(next_call + offset));
      nbc.estQual.push_back(bc.estQual[call]);
      nbc.primary.push_back(bc.primary[call]);
      nbc.secondary.push_back(bc.secondary[call]);
      nbc.consensus.push_back(bc.consensus[call]);
      if (call < bc.bcPos.size() - 1) next_call = bc.bcPos[++call];
    }
  }
}

// assemblyTrace, json.h:108-194
template <class Out, typename std::enable_if<!std::is_same<Out, std::string>::value, int>::type = 0>  // (a stream, not a file name)
inline void assemblyTrace(Out& out, PaddedTrace const& p, std::string const& traceFileName) {
  Trace const& tr = p.tr;
  BaseCalls const& bc = p.bc;
  const int32_t ns = (int32_t)tr.traceACGT[0].size();
  out << "{" << std::endl;
  out << "\"traceFileName\": \"" << traceFileName << "\"," << std::endl;
  out << "\"leadingGaps\": " << p.leadingGaps << "," << std::endl;
  out << "\"trailingGaps\": " << p.trailingGaps << "," << std::endl;
  static const char* channel[4] = {"peakA", "peakC", "peakG", "peakT"};
  for (int k = 0; k < 4; ++k) {
    out << "\"" << channel[k] << "\": [";
    write_int_list(out, (std::size_t)ns, [&](std::size_t i) { return tr.traceACGT[k][i]; });
    out << "]," << std::endl;
  }
  // one visit per called sample, in sample order; calls whose position never comes up are skipped
  auto for_each_call = [&](auto&& emit) {
    uint32_t call = 0;
    int32_t next = bc.bcPos[0];
    for (int32_t i = 0; i < ns; ++i) {
      if (next != i) continue;
      if (i != bc.bcPos[0]) out << ", ";
      emit(i, call);
      if (call < bc.bcPos.size() - 1) next = bc.bcPos[++call];
    }
  };
  out << "\"basecallPos\": [";
  for_each_call([&](int32_t i, uint32_t) { out << (i + 1); });
  out << "]," << std::endl;
  out << "\"basecallQual\": [";
  for_each_call([&](int32_t, uint32_t call) { out << (int32_t)bc.estQual[call]; });
  out << "]," << std::endl;
  uint32_t gapless = 0;
  out << "\"basecalls\": {";
  for_each_call([&](int32_t i, uint32_t call) {
    if (bc.primary[call] == '-') {
      out << "\"" << (i + 1) << "\":\"-\"";
      return;
    }
    out << "\"" << (i + 1) << "\":\"" << (++gapless) << ":" << bc.primary[call];
    if (bc.primary[call] != bc.secondary[call]) out << "|" << bc.secondary[call];
    out << "\"";
  });
  out << "}" << std::endl;
  out << "}" << std::endl;
}

// traceAlignJsonOut, json.h:197-217
template <class Out, typename std::enable_if<!std::is_same<Out, std::string>::value, int>::type = 0>  // (a stream, not a file name)
inline void traceAlignJsonOut(Out& out, PaddedTrace const& p, ReferenceSlice const& rs, AlignRows const& al) {
  out << "{" << std::endl;
  out << "\"gappedTrace\":" << std::endl;
  assemblyTrace(out, p, "trace");
  out << "," << std::endl;
  out << "\"refchr\": \"" << rs.chr << "\"," << std::endl;
  out << "\"refpos\": " << (rs.pos + 1) << "," << std::endl;
  out << "\"altalign\": \"" << al.row0 << "\"," << std::endl;
  out << "\"refalign\": \"" << al.row1 << "\"," << std::endl;
  out << "\"forward\": " << rs.forward << std::endl;
  out << "}" << std::endl;
}

inline void traceAlignJsonOut(std::string const& outfile, PaddedTrace const& p, ReferenceSlice const& rs, AlignRows const& al) {
  TextBuf out(1 << 20);
  traceAlignJsonOut(out, p, rs, al);
  out.write(outfile);
}

// the two-record FASTA of the final alignment, sage.h:328-339
template <class Out, typename std::enable_if<!std::is_same<Out, std::string>::value, int>::type = 0>  // (a stream, not a file name)
inline void alignFastaOut(Out& out, std::string const& traceStem, ReferenceSlice const& rs, AlignRows const& al) {
  out << ">" << traceStem << std::endl;
  out << al.row0 << std::endl;
  out << ">" << rs.chr << (rs.forward ? " (forward)" : " (reverse)") << std::endl;
  out << al.row1 << std::endl;
}

}  // namespace tracy_amd
#endif
