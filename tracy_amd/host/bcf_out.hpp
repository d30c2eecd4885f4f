// bcf_out.hpp -- <prefix>.bcf of `tracy decompose -v` (vcfOutput, variants.h:141-261) without htslib.
//
// The reference builds its records through htslib's bcf_update_* calls and writes them with bcf_write1 into a BGZF stream.  htslib is not
// in this image; the container formats are small and public (SAM/VCF specification, sections "BGZF" and "BCF2"), so the file is written
// directly: BGZF blocks through zlib's raw deflate, the BCF2.2 header (the VCF header text vcfTextOutput writes -- PASS first, the FILTER /
// INFO / FORMAT lines in vcfOutput's order, which fixes the dictionary indices), one typed-value record per variant with the fields in
// the order vcfOutput sets them: ID, REF/ALT, FILTER, INFO TYPE, METHOD, BASEPOS, SIGNALPOS, FORMAT GT, GQ.  Integers take the smallest
// type that holds them (htslib's rule: int8 down to -120, int16 down to -32760, else int32).  The .csi index the reference builds next
// (bcf_index_build) is not written.  Parity status: unpinned -- there is no htslib / bcftools here to read the file back; the tests decode
// it with their own reader (tests/bcf_reader.py) and compare field by field with the VCF text of the same variants.
#ifndef TRACY_AMD_HOST_BCF_OUT_HPP
#define TRACY_AMD_HOST_BCF_OUT_HPP

#include <zlib.h>

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "indigo_out.hpp"

namespace tracy_amd {

namespace bcfdetail {

// ---- BGZF: a series of gzip members of at most 64 KB each, with the member's compressed size in an extra field ----
inline bool bgzf_block(std::vector<uint8_t>& out, const uint8_t* data, std::size_t n) {
  uint8_t buf[65536 + 1024];
  z_stream zs;
  std::memset(&zs, 0, sizeof(zs));
  if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
  zs.next_in = const_cast<Bytef*>(data);
  zs.avail_in = (uInt)n;
  zs.next_out = buf;
  zs.avail_out = sizeof(buf);
  const int rc = deflate(&zs, Z_FINISH);
  const std::size_t clen = sizeof(buf) - zs.avail_out;
  deflateEnd(&zs);
  if (rc != Z_STREAM_END) return false;
  const uint32_t bsize = (uint32_t)(clen + 25);  // total block size - 1
  const uint8_t head[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, (uint8_t)(bsize & 0xff), (uint8_t)(bsize >> 8)};
  out.insert(out.end(), head, head + 18);
  out.insert(out.end(), buf, buf + clen);
  const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), data, (uInt)n), isize = (uint32_t)n;
  for (uint32_t v : {crc, isize})
    for (int k = 0; k < 4; ++k) out.push_back((uint8_t)(v >> (8 * k)));
  return true;
}
inline bool bgzf_compress(std::vector<uint8_t> const& raw, std::vector<uint8_t>& out) {
  constexpr std::size_t kBlock = 0xff00;  // (htslib's BGZF_BLOCK_SIZE: the deflated block stays below 64 KB whatever the data)
  for (std::size_t at = 0; at < raw.size(); at += kBlock)
    if (!bgzf_block(out, raw.data() + at, raw.size() - at < kBlock ? raw.size() - at : kBlock)) return false;
  static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  out.insert(out.end(), eof, eof + 28);
  return true;
}

// ---- BCF2 typed values ----
struct Enc {
  std::vector<uint8_t>& b;
  void u8(uint8_t v) { b.push_back(v); }
  void le(uint64_t v, int bytes) { for (int k = 0; k < bytes; ++k) b.push_back((uint8_t)(v >> (8 * k))); }
  static int int_type(int64_t v) { return (v >= -120 && v <= 127) ? 1 : (v >= -32760 && v <= 32767) ? 2 : 3; }
  void size_type(std::size_t n, int type) {  // descriptor byte (+ the length as a typed integer from 15 elements on)
    if (n < 15) { u8((uint8_t)((n << 4) | (unsigned)type)); return; }
    u8((uint8_t)(0xf0 | (unsigned)type));
    integer((int64_t)n);
  }
  void integer(int64_t v) {  // a typed scalar
    const int t = int_type(v);
    u8((uint8_t)(0x10 | t));
    le((uint64_t)v, t == 1 ? 1 : t == 2 ? 2 : 4);
  }
  void string(std::string const& s) {
    size_type(s.size(), 7);
    b.insert(b.end(), s.begin(), s.end());
  }
};

}  // namespace bcfdetail

// the BCF of vcfOutput (variants.h:141-261); contigs as for vcfTextOutput.  false: the file could not be written.
inline bool bcfOutput(std::string const& outfile, ReportConfig const& c, BaseCalls const& bc, std::vector<Variant> const& var, ReferenceSlice const& rs,
                      std::vector<std::pair<std::string, uint64_t>> const* contigs = nullptr) {
  using bcfdetail::Enc;
  // ---- header: the text of the VCF writer, cut behind its #CHROM line ----
  std::string text;
  {
    std::ostringstream os;
    vcfTextOutput(os, c, bc, std::vector<Variant>(), rs, contigs);
    text = os.str();
  }
  // htslib's BCF headers carry the dictionary index of every FILTER / INFO / FORMAT / contig line as a trailing IDX key
  // (bcf_hdr_format with is_bcf): added here to the VCF writer's lines, in the order that fixes the indices
  {
    std::string withidx;
    int idx_id = 0, idx_ctg = 0;
    bool pass_seen = false;
    std::size_t at = 0;
    while (at < text.size()) {
      std::size_t nl = text.find('\n', at);
      if (nl == std::string::npos) nl = text.size();
      std::string ln = text.substr(at, nl - at);
      const bool dict = ln.rfind("##FILTER=<", 0) == 0 || ln.rfind("##INFO=<", 0) == 0 || ln.rfind("##FORMAT=<", 0) == 0;
      const bool ctg = ln.rfind("##contig=<", 0) == 0;
      if ((dict || ctg) && !ln.empty() && ln.back() == '>') {
        int idx;
        if (ctg) idx = idx_ctg++;
        else if (ln.rfind("##FILTER=<ID=PASS,", 0) == 0) { idx = 0; pass_seen = true; if (idx_id == 0) idx_id = 1; }
        else { if (idx_id == 0 && !pass_seen) idx_id = 1; idx = idx_id++; }
        ln.insert(ln.size() - 1, ",IDX=" + std::to_string(idx));
      }
      withidx += ln;
      if (nl < text.size()) withidx.push_back('\n');
      at = nl + 1;
    }
    text.swap(withidx);
  }
  // dictionary indices follow from the order of the header lines (PASS = 0 by definition)
  enum : int { kPass = 0, kLowQual = 1, kBasepos = 2, kSignalpos = 3, kType = 4, kMethod = 5, kGt = 6, kGq = 7 };
  std::vector<std::string> names;  // contig dictionary, in header order
  if (contigs) for (auto const& ctg : *contigs) names.push_back(ctg.first);
  else names.push_back(rs.chr);
  std::vector<uint8_t> raw;
  Enc e{raw};
  raw.insert(raw.end(), {'B', 'C', 'F', 2, 2});
  e.le(text.size() + 1, 4);
  raw.insert(raw.end(), text.begin(), text.end());
  raw.push_back(0);
  // ---- records ----
  for (Variant const& v : var) {
    const uint32_t q = variantCallIndex(c, bc, rs.forward, v.basenum);
    const int32_t qual = strInclN(v.alt) ? 0 : (int32_t)bc.estQual[q];
    const int64_t basepos = rs.forward ? (int64_t)c.trimLeft + v.basenum : (int64_t)bc.primary.size() - (c.trimRight + v.basenum) + 1;
    int32_t rid = -1;
    for (std::size_t i = 0; i < names.size(); ++i)
      if (names[i] == v.chr) { rid = (int32_t)i; break; }
    std::vector<uint8_t> shared, indiv;
    Enc s{shared}, g{indiv};
    s.le((uint32_t)rid, 4);
    s.le((uint32_t)(v.pos - 1), 4);
    s.le((uint32_t)v.ref.size(), 4);  // rlen
    const float fq = (float)qual;
    uint32_t fbits;
    std::memcpy(&fbits, &fq, 4);
    s.le(fbits, 4);
    s.le((2u << 16) | 4u, 4);  // n_allele << 16 | n_info
    s.le((2u << 24) | 1u, 4);  // n_fmt << 24 | n_sample
    // (htslib's bcf1_sync writes an id of "." -- the default of bcf_update_id -- as a missing value, not as a one-character string)
    if (v.id.empty() || v.id == ".") s.size_type(0, 7);
    else s.string(v.id);
    s.string(v.ref);
    s.string(v.alt);
    s.size_type(1, 1);         // FILTER: one int8
    s.u8((uint8_t)(qual < c.qualCut ? kLowQual : kPass));
    s.integer(kType); s.string(variantType(v.ref, v.alt));
    s.integer(kMethod); s.string(std::string("EMBL.TRACYv") + kTracyVersion);
    s.integer(kBasepos); s.integer(basepos);
    s.integer(kSignalpos); s.integer((int64_t)bc.bcPos[q] + 1);
    g.integer(kGt);
    g.size_type(2, 1);         // two int8 per sample: (allele + 1) << 1, unphased; 0 = missing
    const uint8_t a0 = (v.gt == 0 || v.gt == 1) ? 2 : v.gt == 2 ? 4 : 0, a1 = v.gt == 0 ? 2 : (v.gt == 1 || v.gt == 2) ? 4 : 0;
    g.u8(a0); g.u8(a1);
    g.integer(kGq);
    const int64_t gq = (int64_t)(int32_t)bc.estQual[q];
    const int t = Enc::int_type(gq);
    g.size_type(1, t);
    g.le((uint64_t)gq, t == 1 ? 1 : t == 2 ? 2 : 4);
    e.le(shared.size(), 4);
    e.le(indiv.size(), 4);
    raw.insert(raw.end(), shared.begin(), shared.end());
    raw.insert(raw.end(), indiv.begin(), indiv.end());
  }
  std::vector<uint8_t> file;
  if (!bcfdetail::bgzf_compress(raw, file)) return false;
  TextBuf f(file.size() + 64);
  f.put(reinterpret_cast<const char*>(file.data()), file.size());
  return f.write(outfile);
}

}  // namespace tracy_amd
#endif
