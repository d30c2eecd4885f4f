// indigo_out.hpp -- host pieces of `tracy decompose` after the device chain: the decomposition table,
// variant calling on the allele alignments and the JSON report.
//
// Mirrors of /root/reference/src (same names, argument meaning and file contents):
//   writeDecomposition                   decompose.h:622-632
//   Variant, insertVariant, callVariants, variantType     variants.h:9-138
//   _metaOut, _traceJsonOut              json.h:17-105
//   xWindowViewport                      json.h:248-257
//   traceAlleleAlignJsonOut              json.h:260-381
//   _reverseReferenceSlize               fmindex.h:40-49
// vcfOutput (variants.h:141-261) writes BCF through htslib, which this image does not have: variants are
// written as VCF TEXT (<prefix>.vcf, same header lines / columns / INFO / FORMAT values) -- a documented
// deviation.  PARITY UNPINNED (json.h / variants.h need htslib and Boost), cross-checked by an
// independent Python restatement in tests/.
#ifndef TRACY_AMD_INDIGO_OUT_HPP
#define TRACY_AMD_INDIGO_OUT_HPP

#include "text_buf.hpp"
#include <ctime>

#include "sage_out.hpp"

namespace tracy_amd {

static const char* const kTracyVersion = "0.9.1";  // version.h:8 of the reference this build mirrors

struct TraceBreakpoint {  // fmindex.h:51-56
  bool indelshift = false;
  bool traceleft = true;
  uint32_t breakpoint = 0;
  float bestDiff = 0;
};

// the IndigoConfig fields the writers read (indigo.h:16-40)
struct ReportConfig {
  uint16_t trimLeft = 50, trimRight = 50, qualCut = 45;
  float pratio = 0.33f;
  std::string genomeName, inputName;  // c.genome.filename(), c.ab.filename()
};

typedef std::vector<std::pair<int32_t, int32_t>> Decomposition;

template <class Out, typename std::enable_if<!std::is_same<Out, std::string>::value, int>::type = 0>  // (a stream, not a file name)
inline void writeDecomposition(Out& out, Decomposition const& dcp) {
  out << "indel\tdecomp" << std::endl;
  for (auto const& row : dcp) out << row.first << "\t" << row.second << std::endl;
}

struct Variant {  // variants.h:9-23
  int32_t pos, basenum, gt;
  std::string chr, ref, alt, id;
  bool operator<(Variant const& o) const {
    if (chr != o.chr) return chr < o.chr;
    if (pos != o.pos) return pos < o.pos;
    return basenum < o.basenum;
  }
};

inline bool strInclN(std::string const& s) { return s.find_first_of("nN") != std::string::npos; }

// insertVariant, variants.h:34-53: a variant seen on both alleles becomes homozygous (gt + 1)
inline void insertVariant(std::vector<Variant>& var, int32_t pos, int32_t basenum, int32_t gt, std::string const& chr,
                          std::string const& ref, std::string const& alt) {
  for (Variant& v : var)
    if (v.pos == pos && v.chr == chr && v.ref == ref && v.alt == alt) {
      v.gt += 1;
      return;
    }
  if (pos > 0 && !strInclN(ref)) var.push_back(Variant{pos, basenum, gt, chr, ref, alt, "."});
}

// callVariants, variants.h:56-126: SNVs and left-anchored indels between the first and last aligned base
// of row 0; positions are 1-based on the reference slice's chromosome
inline void callVariants(AlignRows const& al, ReferenceSlice const& rs, std::vector<Variant>& var) {
  int32_t ri = rs.pos;
  int32_t first = -1, last = -1;
  for (uint32_t j = 0; j < al.cols(); ++j) {
    if (al.row0[j] != '-') {
      if (first == -1) first = (int32_t)j;
      last = (int32_t)j;
    }
    if (al.row1[j] != '-' && first == -1) ++ri;
  }
  if (first < 0) return;
  int32_t vi = 0, delStart = 0, insStart = 0;
  std::string del, ins;
  char anchor = 'N';  // reference base before a leading event is unknown
  for (int32_t j = first; j <= last; ++j) {
    const char a = al.row0[j], r = al.row1[j];
    if (!del.empty() && a != '-') {
      insertVariant(var, delStart, vi, 1, rs.chr, del, std::string(1, del[0]));
      del.clear();
    }
    if (!ins.empty() && r != '-') {
      insertVariant(var, insStart, vi, 1, rs.chr, std::string(1, ins[0]), ins);
      ins.clear();
    }
    if (a != '-') ++vi;
    if (r != '-') ++ri;
    if (a != r) {
      if (a != '-' && r != '-') {
        insertVariant(var, ri, vi, 1, rs.chr, std::string(1, r), std::string(1, a));
      } else if (a == '-') {
        if (del.empty()) { del.push_back(anchor); delStart = ri - 1; }
        del.push_back(r);
      } else {
        if (ins.empty()) { ins.push_back(anchor); insStart = ri; }
        ins.push_back(a);
      }
    }
    if (r != '-') anchor = r;
  }
}

inline std::string variantType(std::string const& ref, std::string const& alt) {  // variants.h:129-138
  if (ref.size() == 1 && alt.size() == 1) return "SNV";
  if (ref.size() > alt.size()) return "Deletion";
  if (ref.size() < alt.size()) return "Insertion";
  return "Complex";
}

inline void reverseReferenceSlice(ReferenceSlice const& in, ReferenceSlice& out) {  // _reverseReferenceSlize, fmindex.h:40-49
  out = in;
  out.forward = !in.forward;
  reverseComplement(out.refslice);
}

// xWindowViewport, json.h:248-257: +-150 samples around basecall `pos`, clamped to the called range
inline std::pair<int32_t, int32_t> xWindowViewport(BaseCalls const& bc, int32_t pos) {
  const int32_t centre = bc.bcPos[pos] + 1, lastpeak = bc.bcPos[bc.bcPos.size() - 1];
  return std::make_pair(centre <= 150 ? 1 : centre - 150, centre + 150 < lastpeak ? centre + 150 : lastpeak);
}

// index into the trace's basecalls of variant base number `basenum` (counted on the trimmed allele)
inline uint32_t variantCallIndex(ReportConfig const& c, BaseCalls const& bc, bool forward, int32_t basenum) {
  return forward ? (uint32_t)(c.trimLeft + basenum - 1) : (uint32_t)(bc.primary.size() - (c.trimRight + basenum));
}

template <class Out, typename std::enable_if<!std::is_same<Out, std::string>::value, int>::type = 0>  // (a stream, not a file name)
inline void metaOut(Out& out, ReportConfig const& c) {  // _metaOut, json.h:17-31
  out << "\"meta\": {\"program\": \"tracy\", \"version\": \"" << kTracyVersion << "\", \"arguments\": {\"trimLeft\": " << c.trimLeft
      << ", \"trimRight\": " << c.trimRight << ", \"pratio\": " << c.pratio << ", \"genome\": \"" << c.genomeName << "\", \"input\": \""
      << c.inputName << "\"}}," << std::endl;
}

// _traceJsonOut, json.h:33-105
template <class Out, typename std::enable_if<!std::is_same<Out, std::string>::value, int>::type = 0>  // (a stream, not a file name)
inline void traceJsonBody(Out& out, BaseCalls const& bc, Trace const& tr) {
  const int32_t ns = (int32_t)tr.traceACGT[0].size();
  out << "\"pos\": [";
  write_int_list(out, (std::size_t)ns, [](std::size_t i) { return (int32_t)i + 1; });
  out << "]," << std::endl;
  static const char* channel[4] = {"peakA", "peakC", "peakG", "peakT"};
  for (int k = 0; k < 4; ++k) {
    out << "\"" << channel[k] << "\": [";
    write_int_list(out, (std::size_t)ns, [&](std::size_t i) { return tr.traceACGT[k][i]; });
    out << "]," << std::endl;
  }
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
  out << "\"basecalls\": {";
  for_each_call([&](int32_t i, uint32_t call) {
    out << "\"" << (i + 1) << "\":\"" << (call + 1) << ":" << bc.primary[call];
    if (bc.primary[call] != bc.secondary[call]) {
      static const std::map<char, const char*> expand = {{'A', "A"}, {'C', "C"}, {'G', "G"}, {'T', "T"}, {'N', "N"}, {'R', "A|G"},
                                                         {'Y', "C|T"}, {'S', "C|G"}, {'W', "A|T"}, {'K', "G|T"}, {'M', "A|C"}};
      auto it = expand.find(bc.secondary[call]);  // expandIUPAC, abif.h:99-113
      out << "|" << (it == expand.end() ? "N" : it->second);
    }
    out << "\"";
  });
  out << "}," << std::endl;
  out << "\"primarySeq\": \"" << bc.primary << "\"," << std::endl;
  out << "\"secondarySeq\": \"" << bc.secondary << "\"" << std::endl;
}

struct AlleleReport {
  ReferenceSlice rs1, rs2;
  AlignRows align1, align2, align3;
  int32_t a1Score = 0, a2Score = 0, a3Score = 0;
  TraceBreakpoint bp;
  std::pair<double, double> a1a2;
  Decomposition dcp;
  std::vector<Variant> var;
};

// traceAlleleAlignJsonOut, json.h:260-381
template <class Out, typename std::enable_if<!std::is_same<Out, std::string>::value, int>::type = 0>  // (a stream, not a file name)
inline void traceAlleleAlignJsonOut(Out& out, ReportConfig const& c, BaseCalls const& bc, Trace const& tr, AlleleReport const& r) {
  out << "{" << std::endl;
  metaOut(out, c);
  traceJsonBody(out, bc, tr);
  out << "," << std::endl;
  const std::pair<int32_t, int32_t> xwin = xWindowViewport(bc, (int32_t)(c.trimLeft + r.bp.breakpoint));
  out << "\"chartConfig\": { \"x\": { \"axis\": { \"range\": [" << xwin.first << ", " << xwin.second << "] }}}," << std::endl;
  auto allele = [&](const char* n, ReferenceSlice const& rs, AlignRows const& al, int32_t score) {
    out << "\"ref" << n << "chr\": \"" << rs.chr << "\"," << std::endl;
    out << "\"ref" << n << "pos\": " << (rs.pos + 1) << "," << std::endl;
    out << "\"alt" << n << "align\": \"" << al.row0 << "\"," << std::endl;
    out << "\"ref" << n << "align\": \"" << al.row1 << "\"," << std::endl;
    out << "\"ref" << n << "forward\": " << rs.forward << "," << std::endl;
    out << "\"align" << n << "score\": " << score << "," << std::endl;
  };
  allele("1", r.rs1, r.align1, r.a1Score);
  allele("2", r.rs2, r.align2, r.a2Score);
  out << "\"allele1fraction\": " << r.a1a2.first << "," << std::endl;
  out << "\"allele1align\": \"" << r.align3.row0 << "\"," << std::endl;
  out << "\"allele2fraction\": " << r.a1a2.second << "," << std::endl;
  out << "\"allele2align\": \"" << r.align3.row1 << "\"," << std::endl;
  out << "\"align3score\": " << r.a3Score << "," << std::endl;
  out << "\"hetindel\": " << r.bp.indelshift << "," << std::endl;
  out << "\"decomposition\": {" << std::endl;
  out << "\"x\": [";
  for (std::size_t i = 0; i < r.dcp.size(); ++i) out << (i ? ", " : "") << r.dcp[i].first;
  out << "]," << std::endl;
  out << "\"y\": [";
  for (std::size_t i = 0; i < r.dcp.size(); ++i) out << (i ? ", " : "") << r.dcp[i].second;
  out << "]" << std::endl;
  out << "}," << std::endl;
  out << "\"variants\": {" << std::endl;
  out << "\"columns\": [\"chr\", \"pos\", \"id\", \"ref\", \"alt\", \"qual\", \"filter\", \"type\", \"genotype\", \"basepos\", \"signalpos\"]," << std::endl;
  out << "\"rows\": [" << std::endl;
  const bool fwd = r.rs1.forward;
  for (std::size_t i = 0; i < r.var.size(); ++i) {
    Variant const& v = r.var[i];
    if (i) out << "," << std::endl;
    const uint32_t q = variantCallIndex(c, bc, fwd, v.basenum);
    static const char* gt[3] = {"hom. REF", "het.", "hom. ALT"};
    out << "[\"" << v.chr << "\", " << v.pos << ", \"" << v.id << "\", \"" << v.ref << "\", \"" << v.alt << "\", " << (int32_t)bc.estQual[q] << ", "
        << ((int32_t)bc.estQual[q] < c.qualCut ? "\"LowQual\", " : "\"PASS\", ") << "\"" << variantType(v.ref, v.alt) << "\", \""
        << ((v.gt >= 0 && v.gt <= 2) ? gt[v.gt] : "missing") << "\", ";
    if (fwd) out << c.trimLeft + v.basenum << ", ";
    else out << bc.primary.size() - (c.trimRight + v.basenum) + 1 << ", ";
    out << bc.bcPos[q] + 1 << "]";
  }
  out << "]," << std::endl;
  out << "\"xranges\": [" << std::endl;
  for (std::size_t i = 0; i < r.var.size(); ++i) {
    if (i) out << "," << std::endl;
    const std::pair<int32_t, int32_t> w = xWindowViewport(bc, (int32_t)variantCallIndex(c, bc, fwd, r.var[i].basenum));
    out << "[" << w.first << ", " << w.second << "]";
  }
  out << "]" << std::endl;
  out << "}" << std::endl;
  out << "}" << std::endl;
}

// variants as VCF text: the header lines, columns, INFO and FORMAT values vcfOutput (variants.h:141-261)
// puts into its BCF
// contigs: (name, length + 1) of every sequence of an indexed genome (rs.filetype == 0, variants.h:176-186); NULL
// for a single FASTA / wildtype reference, whose one contig line carries rs.refslice.size()
template <class Out, typename std::enable_if<!std::is_same<Out, std::string>::value, int>::type = 0>  // (a stream, not a file name)
inline void vcfTextOutput(Out& out, ReportConfig const& c, BaseCalls const& bc, std::vector<Variant> const& var, ReferenceSlice const& rs,
                          std::vector<std::pair<std::string, uint64_t>> const* contigs = nullptr) {
  char date[16];
  std::time_t t = std::time(nullptr);
  std::tm tmv;
  localtime_r(&t, &tmv);
  std::strftime(date, sizeof(date), "%Y%m%d", &tmv);
  out << "##fileformat=VCFv4.2\n##FILTER=<ID=PASS,Description=\"All filters passed\">\n##fileDate=" << date << "\n"
      << "##FILTER=<ID=LowQual,Description=\"Low quality variant call.\">\n"
      << "##INFO=<ID=BASEPOS,Number=1,Type=Integer,Description=\"Basecall position in trace\">\n"
      << "##INFO=<ID=SIGNALPOS,Number=1,Type=Integer,Description=\"Trace signal position\">\n"
      << "##INFO=<ID=TYPE,Number=1,Type=String,Description=\"Variant type\">\n"
      << "##INFO=<ID=METHOD,Number=1,Type=String,Description=\"Type of approach used to detect variant\">\n"
      << "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n"
      << "##FORMAT=<ID=GQ,Number=1,Type=Integer,Description=\"Genotype Quality\">\n"
      << "##reference=" << c.genomeName << "\n";
  if (contigs) {
    for (auto const& ctg : *contigs) out << "##contig=<ID=" << ctg.first << ",length=" << ctg.second << ">\n";
  } else {
    out << "##contig=<ID=" << rs.chr << ",length=" << rs.refslice.size() << ">\n";
  }
  out << "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tsample\n";
  for (Variant const& v : var) {
    const uint32_t q = variantCallIndex(c, bc, rs.forward, v.basenum);
    const int32_t qual = strInclN(v.alt) ? 0 : (int32_t)bc.estQual[q];
    const int64_t basepos = rs.forward ? (int64_t)c.trimLeft + v.basenum : (int64_t)bc.primary.size() - (c.trimRight + v.basenum) + 1;
    static const char* gt[3] = {"0/0", "0/1", "1/1"};
    out << v.chr << "\t" << v.pos << "\t" << v.id << "\t" << v.ref << "\t" << v.alt << "\t" << qual << "\t" << (qual < c.qualCut ? "LowQual" : "PASS")
        << "\tTYPE=" << variantType(v.ref, v.alt) << ";METHOD=EMBL.TRACYv" << kTracyVersion << ";BASEPOS=" << basepos << ";SIGNALPOS=" << bc.bcPos[q] + 1
        << "\tGT:GQ\t" << ((v.gt >= 0 && v.gt <= 2) ? gt[v.gt] : "./.") << ":" << (int32_t)bc.estQual[q] << "\n";
  }
}

}  // namespace tracy_amd
#endif
