// bcf_out.hpp -- <prefix>.bcf of `tracy decompose -v` (vcfOutput, variants.h:141-261) without htslib.
//
// The reference builds its records through htslib's bcf_update_* calls and writes them with bcf_write1 into a BGZF stream.  htslib is not
// in this image; the container formats are small and public (SAM/VCF specification, sections "BGZF" and "BCF2"), so the file is written
// directly: BGZF blocks through zlib's raw deflate, the BCF2.2 header (the VCF header text vcfTextOutput writes -- PASS first, the FILTER /
// INFO / FORMAT lines in vcfOutput's order, which fixes the dictionary indices), one typed-value record per variant with the fields in
// the order vcfOutput sets them: ID, REF/ALT, FILTER, INFO TYPE, METHOD, BASEPOS, SIGNALPOS, FORMAT GT, GQ.  Integers take the smallest
// type that holds them (htslib's rule: int8 down to -120, int16 down to -32760, else int32).  The .csi index the reference builds next
// (bcf_index_build) is written beside it (csi_build below).  Parity status: unpinned -- there is no htslib / bcftools here to read the file back; the tests decode
// it with their own reader (tests/bcf_reader.py) and compare field by field with the VCF text of the same variants.
#ifndef TRACY_AMD_HOST_BCF_OUT_HPP
#define TRACY_AMD_HOST_BCF_OUT_HPP

#include <zlib.h>

#include <algorithm>
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
constexpr std::size_t kBgzfBlock = 0xff00;  // (htslib's BGZF_BLOCK_SIZE: the deflated block stays below 64 KB whatever the data)
// block_start (or null): the file offset at which every block of kBgzfBlock uncompressed bytes begins (virtual file offsets of the index)
inline bool bgzf_compress(std::vector<uint8_t> const& raw, std::vector<uint8_t>& out, std::vector<uint64_t>* block_start = nullptr) {
  constexpr std::size_t kBlock = kBgzfBlock;
  for (std::size_t at = 0; at < raw.size(); at += kBlock) {
    if (block_start) block_start->push_back(out.size());
    if (!bgzf_block(out, raw.data() + at, raw.size() - at < kBlock ? raw.size() - at : kBlock)) return false;
  }
  if (block_start) block_start->push_back(out.size());  // (where a block behind the last one would begin: the end-of-file marker)
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

// ---- CSI index (bcf_index_build(fn, 14), variants.h:263): binning index over the records' [pos, pos + rlen) per contig, written from the
// CSIv1 layout of the SAM/VCF specification: "CSI\1", min_shift, depth, l_aux = 0, n_ref, per contig its bins {bin, loffset, chunks} and
// the pseudo-bin with the contig's file span and record count; BGZF-compressed like the file it indexes.  A chunk is a run of records
// that fall into the same bin one after the other (hts_idx_push merges them the same way); a bin's loffset is the linear index's entry
// for the bin's first 2^min_shift window (the offset of the first record that overlaps that window or a later one).  htslib also folds
// bins whose chunks lie within one 64 KB span into their parents (compress_binning): an optimisation of the index's size, not of what it
// finds -- left out, so the file is a valid index but not htslib's bytes.
struct CsiRecord { int32_t rid; int64_t beg, end; uint64_t voff_beg, voff_end; };
inline int csi_depth(int64_t max_len, int min_shift) {
  int n_lvls = 0;
  max_len += 256;
  for (int64_t s = 1ll << min_shift; max_len > s; s <<= 3) ++n_lvls;
  return n_lvls;
}
inline uint32_t csi_reg2bin(int64_t beg, int64_t end, int min_shift, int depth) {
  int l, s = min_shift;
  int64_t t = ((1ll << (depth * 3)) - 1) / 7;
  for (--end, l = depth; l > 0; --l, s += 3, t -= 1ll << (l * 3))
    if ((beg >> s) == (end >> s)) return (uint32_t)(t + (beg >> s));
  return 0;
}
inline int64_t csi_bin_first_window(uint32_t bin, int depth) {  // first 2^min_shift window a bin covers (hts_bin_bot)
  int l = 0;
  for (uint32_t b = bin; b; b = (b - 1) >> 3) ++l;  // the bin's level
  const int64_t first_of_level = ((1ll << (l * 3)) - 1) / 7;
  return ((int64_t)bin - first_of_level) << ((depth - l) * 3);
}
inline std::vector<uint8_t> csi_build(std::vector<CsiRecord> const& recs, std::size_t n_ref, int64_t max_contig_len, uint64_t voff_first) {
  constexpr int kMinShift = 14;
  const int depth = csi_depth(max_contig_len, kMinShift);
  std::vector<uint8_t> raw;
  Enc e{raw};
  raw.insert(raw.end(), {'C', 'S', 'I', 1});
  e.le((uint32_t)kMinShift, 4);
  e.le((uint32_t)depth, 4);
  e.le(0, 4);  // l_aux
  e.le((uint32_t)n_ref, 4);
  const uint32_t meta_bin = (uint32_t)(((1ll << ((depth + 1) * 3)) - 1) / 7 + 1);
  for (std::size_t r = 0; r < n_ref; ++r) {
    struct Bin { uint32_t id; std::vector<std::pair<uint64_t, uint64_t>> chunks; };
    std::vector<Bin> bins;              // in order of first appearance (a contig of a Sanger trace has a handful)
    std::vector<uint64_t> lidx;         // window -> offset of the first record that overlaps it
    uint64_t off_beg = 0, off_end = 0, n_mapped = 0;
    uint32_t last_bin = 0xffffffffu;
    for (CsiRecord const& x : recs) {
      if (x.rid != (int32_t)r) { last_bin = 0xffffffffu; continue; }
      if (n_mapped == 0) off_beg = x.voff_beg;
      off_end = x.voff_end;
      ++n_mapped;
      const int64_t end = x.end > x.beg ? x.end : x.beg + 1;
      const uint32_t b = csi_reg2bin(x.beg, end, kMinShift, depth);
      Bin* bp = nullptr;
      for (Bin& q : bins) if (q.id == b) { bp = &q; break; }
      if (!bp) { bins.push_back(Bin{b, {}}); bp = &bins.back(); }
      if (b == last_bin && !bp->chunks.empty()) bp->chunks.back().second = x.voff_end;
      else bp->chunks.emplace_back(x.voff_beg, x.voff_end);
      last_bin = b;
      const std::size_t w0 = (std::size_t)(x.beg >> kMinShift), w1 = (std::size_t)((end - 1) >> kMinShift);
      if (lidx.size() <= w1) lidx.resize(w1 + 1, ~0ull);
      for (std::size_t w = w0; w <= w1; ++w) if (lidx[w] == ~0ull) lidx[w] = x.voff_beg;
    }
    for (std::size_t w = lidx.size(); w-- > 1;) if (lidx[w - 1] == ~0ull) lidx[w - 1] = lidx[w];  // windows without a record: the next one's
    e.le((uint32_t)(bins.size() + (n_mapped ? 1 : 0)), 4);
    for (Bin const& q : bins) {
      e.le(q.id, 4);
      const int64_t w = csi_bin_first_window(q.id, depth);
      e.le((w >= 0 && (std::size_t)w < lidx.size()) ? lidx[(std::size_t)w] : 0ull, 8);
      e.le((uint32_t)q.chunks.size(), 4);
      for (auto const& c : q.chunks) { e.le(c.first, 8); e.le(c.second, 8); }
    }
    if (n_mapped) {
      e.le(meta_bin, 4);
      e.le(0, 8);  // (loffset of the pseudo-bin)
      e.le(2, 4);
      e.le(off_beg, 8); e.le(off_end, 8);
      e.le(n_mapped, 8); e.le(0, 8);  // mapped / unmapped
    }
  }
  e.le(0, 8);  // n_no_coor
  (void)voff_first;
  return raw;
}

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
  std::vector<bcfdetail::CsiRecord> index_recs;
  const std::size_t first_record = raw.size();
  for (Variant const& v : var) {
    const std::size_t rec_begin = raw.size();
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
    index_recs.push_back(bcfdetail::CsiRecord{rid, (int64_t)v.pos - 1, (int64_t)v.pos - 1 + (int64_t)v.ref.size(), (uint64_t)rec_begin, (uint64_t)raw.size()});
  }
  std::vector<uint8_t> file;
  std::vector<uint64_t> block_start;
  if (!bcfdetail::bgzf_compress(raw, file, &block_start)) return false;
  {
    TextBuf f(file.size() + 64);
    f.put(reinterpret_cast<const char*>(file.data()), file.size());
    if (!f.write(outfile)) return false;
  }
  // ---- <outfile>.csi (bcf_index_build, variants.h:263): uncompressed positions -> virtual file offsets (block start << 16 | offset in the block) ----
  auto voff = [&](uint64_t u) -> uint64_t {
    const std::size_t b = (std::size_t)(u / bcfdetail::kBgzfBlock);
    return (block_start[b < block_start.size() ? b : block_start.size() - 1] << 16) | (u % bcfdetail::kBgzfBlock);
  };
  for (auto& x : index_recs) { x.voff_beg = voff(x.voff_beg); x.voff_end = voff(x.voff_end); }
  int64_t max_len = 0;
  if (contigs) for (auto const& ctg : *contigs) max_len = std::max<int64_t>(max_len, (int64_t)ctg.second);
  else max_len = (int64_t)rs.refslice.size();
  const std::vector<uint8_t> idx_raw = bcfdetail::csi_build(index_recs, names.size(), max_len, voff(first_record));
  std::vector<uint8_t> idx_file;
  if (!bcfdetail::bgzf_compress(idx_raw, idx_file)) return false;
  TextBuf fi(idx_file.size() + 64);
  fi.put(reinterpret_cast<const char*>(idx_file.data()), idx_file.size());
  return fi.write(outfile + ".csi");
}

}  // namespace tracy_amd
#endif
