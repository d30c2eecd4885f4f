// trace_io.hpp -- chromatogram file formats on the host side of the path (north_star: "ABIF/SCF parsing,
// basecalling and FM-index seeding stay on the host").  Readers produce the Trace the basecaller and
// createProfile() consume; the ABIF writer is the build's own generator for synthetic traces
// (SURVEY.md section 8(d) config 1, Appendix B).
//
// Mirrors of /root/reference/src (names, argument meaning, return values):
//   readab(filename, Trace&)             abif.h:286-405   pinned against the reference's own abif.h (oracle/_ref)
//   traceFormat(filename)                scf.h:18-34
//   readscf(filename, Trace&)            scf.h:38-102     parity unpinned (scf.h needs boost::lexical_cast)
//   traceTxtOut                          abif.h:513-533   pinned (oracle/_ref)
// Deliberate differences, all on inputs where the reference has undefined behaviour: directory entries or
// payloads that point outside the file make the readers return false instead of reading out of bounds; a
// dye order (FWO_) longer than four letters is cut at four.
#ifndef TRACY_AMD_TRACE_IO_HPP
#define TRACY_AMD_TRACE_IO_HPP

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include "text_buf.hpp"
#include <cerrno>
#include <cstring>
#include <memory>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <ostream>
#include <string>
#include <vector>

#include "tracy_host.hpp"

namespace tracy_amd {

namespace detail {

// big-endian view of a whole file
class FileBytes {
 public:
  // one open / fstat / read (a manifest is 10^4 files of 100 KB: no stream object, no seek, no zero fill of the buffer); anything
  // that is not a regular file (a pipe) is read to its end in growing steps
  bool load(std::string const& filename) {
    n_ = 0;
    const int fd = ::open(filename.c_str(), O_RDONLY | O_CLOEXEC);
    if (fd < 0) return false;
    struct stat st;
    if (::fstat(fd, &st) != 0 || S_ISDIR(st.st_mode)) { ::close(fd); return false; }
    std::size_t cap = S_ISREG(st.st_mode) ? (std::size_t)st.st_size + 1 : (std::size_t)1 << 16;
    b_.reset(new uint8_t[cap]);
    std::size_t got = 0;
    for (;;) {
      if (got == cap) {
        std::unique_ptr<uint8_t[]> nb(new uint8_t[2 * cap]);
        std::memcpy(nb.get(), b_.get(), got);
        b_.swap(nb);
        cap *= 2;
      }
      const ssize_t r = ::read(fd, b_.get() + got, cap - got);
      if (r < 0 && errno == EINTR) continue;
      if (r < 0) { ::close(fd); return false; }
      if (r == 0) break;
      got += (std::size_t)r;
    }
    ::close(fd);
    n_ = got;
    return true;
  }
  std::size_t size() const { return n_; }
  bool has(std::size_t pos, std::size_t len) const { return pos <= n_ && len <= n_ - pos; }
  uint8_t u8(std::size_t p) const { return b_[p]; }
  int16_t i16(std::size_t p) const { return (int16_t)(uint16_t)((b_[p] << 8) | b_[p + 1]); }
  int32_t i32(std::size_t p) const {
    return (int32_t)(((uint32_t)b_[p] << 24) | ((uint32_t)b_[p + 1] << 16) | ((uint32_t)b_[p + 2] << 8) | (uint32_t)b_[p + 3]);
  }
  std::string str(std::size_t p, std::size_t len) const { return std::string(reinterpret_cast<const char*>(b_.get()) + p, len); }
  const char* chars() const { return reinterpret_cast<const char*>(b_.get()); }
  bool starts_with(const char* magic4) const { return n_ >= 4 && std::memcmp(b_.get(), magic4, 4) == 0; }

 private:
  std::unique_ptr<uint8_t[]> b_;
  std::size_t n_ = 0;
};

inline std::string only_dna(std::string s) {  // replaceNonDna, abif.h:276-284
  for (auto& c : s)
    if (!(c == 'A' || c == 'C' || c == 'G' || c == 'T')) c = 'N';
  return s;
}

}  // namespace detail

// traceFormat, scf.h:18-34: 0 = ABIF, 1 = SCF, -1 = neither / unreadable
inline int32_t traceFormat(detail::FileBytes const& f) { return f.starts_with("ABIF") ? 0 : f.starts_with(".scf") ? 1 : -1; }
inline int32_t traceFormat(std::string const& filename) {
  detail::FileBytes f;
  if (!f.load(filename)) return -1;
  return traceFormat(f);
}

// readab, abif.h:286-405.  Tags used: PBAS.2 / P2BA.1 (char), FWO_.1 (dye order), PLOC.2 (int16 peak
// positions), DATA.9-12 (int16 channels), PCON.2 (quality bytes; its element type is forced to byte).
// A payload is the directory slot itself when it fits in 4 bytes, else the file offset in the slot.
// Character payloads are read ONE BYTE PAST their declared length (abif.h:346 "+ 1"), exactly like the
// reference: the extra character becomes 'N' (or stays a DNA letter) and is then cut by the common
// length of calls / qualities / positions.  Repeated numeric tags append, repeated text tags overwrite.
inline bool readab(detail::FileBytes const& f, Trace& tr) {
  if (!f.has(0, 34)) return false;
  if (!f.starts_with("ABIF")) {
    std::cerr << "File is not in ABIF format!" << std::endl;
    return false;
  }
  const int32_t slot = f.i16(16), nslots = f.i32(18), dir = f.i32(26);
  Trace::TACGTMountains channel(4);
  std::string order;
  for (int32_t i = 0; i < nslots; ++i) {
    const int64_t at = (int64_t)i * slot + dir;
    if (at < 0 || !f.has((std::size_t)at, 28)) return false;
    const std::size_t e = (std::size_t)at;
    const std::string name = f.str(e, 4);
    const int32_t number = f.i32(e + 4);
    int32_t etype = f.i16(e + 8);
    const int32_t esize = f.i16(e + 10), count = f.i32(e + 12), dsize = f.i32(e + 16);
    if (name == "PCON") etype = 1;
    if (etype != 1 && etype != 2 && etype != 4) continue;
    const std::string key = name + "." + std::to_string(number);
    static const char* used[] = {"PBAS.2", "P2BA.1", "FWO_.1", "PLOC.2", "DATA.9", "DATA.10", "DATA.11", "DATA.12", "PCON.2"};
    bool wanted = false;
    for (const char* u : used) wanted = wanted || key == u;
    if (!wanted) continue;
    const int64_t begin = dsize > 4 ? (int64_t)f.i32(e + 20) : at + 20;
    int64_t end = begin + (int64_t)count * esize + 1;
    if (end > (int64_t)f.size()) end = (int64_t)f.size();
    if (begin < 0 || end < begin) return false;
    const std::size_t b = (std::size_t)begin, len = (std::size_t)(end - begin);
    if (etype == 2) {
      if (key == "PBAS.2") tr.basecalls1 = detail::only_dna(f.str(b, len));
      else if (key == "P2BA.1") tr.basecalls2 = detail::only_dna(f.str(b, len));
      else if (key == "FWO_.1") order = f.str(b, len);
    } else if (etype == 4) {
      Trace::TMountains* dst = nullptr;
      if (key == "PLOC.2") dst = &tr.basecallpos;
      else if (key == "DATA.9") dst = &channel[0];
      else if (key == "DATA.10") dst = &channel[1];
      else if (key == "DATA.11") dst = &channel[2];
      else if (key == "DATA.12") dst = &channel[3];
      if (dst) {
        if (count < 0 || (std::size_t)count * 2 > len) return false;
        const std::size_t had = dst->size();
        dst->resize(had + (std::size_t)count);
        Trace::TValue* o = dst->data() + had;
        for (int32_t k = 0; k < count; ++k) o[k] = f.i16(b + 2 * (std::size_t)k);
      }
    } else if (key == "PCON.2") {
      if (count < 0 || (std::size_t)count > len) return false;
      tr.qual.reserve(tr.qual.size() + (std::size_t)count);
      for (int32_t k = 0; k < count; ++k) tr.qual.push_back(f.u8(b + (std::size_t)k));
    }
  }
  // common length of calls, qualities and positions (abif.h:382-391); a missing P2BA leaves basecalls2 as NULs
  uint32_t n1 = (uint32_t)tr.basecalls1.size();
  if (!tr.basecalls2.empty()) n1 = (uint32_t)std::min(tr.basecalls1.size(), tr.basecalls2.size());
  const uint32_t n2 = (uint32_t)std::min(tr.qual.size(), tr.basecallpos.size());
  const uint32_t n = std::min(n1, n2);
  tr.basecallpos.resize(n);
  tr.basecalls1.resize(n);
  tr.basecalls2.resize(n);
  tr.qual.resize(n);
  tr.traceACGT.assign(4, Trace::TMountains());
  for (std::size_t i = 0; i < order.size() && i < 4; ++i) {
    const int k = order[i] == 'A' ? 0 : order[i] == 'C' ? 1 : order[i] == 'G' ? 2 : order[i] == 'T' ? 3 : -1;
    if (k >= 0) tr.traceACGT[k] = std::move(channel[i]);  // (each channel is handed over once)
  }
  if (n) return true;
  std::cerr << "File lacks basecalls!" << std::endl;
  return false;
}
inline bool readab(std::string const& filename, Trace& tr) {
  detail::FileBytes f;
  return f.load(filename) && readab(f, tr);
}

// readscf, scf.h:38-102.  Only SCF >= 3.0 is accepted (as in the reference); samples are 16-bit,
// stored per channel as second differences which are integrated twice with a 16-bit carry.
inline bool readscf(detail::FileBytes const& f, Trace& tr) {
  tr.traceACGT.assign(4, Trace::TMountains());
  if (!f.has(0, 40)) return false;
  if (!f.starts_with(".scf")) {
    std::cerr << "File is not in SCF format!" << std::endl;
    return false;
  }
  const int32_t samples = f.i32(4), samples_at = f.i32(8), bases = f.i32(12), bases_at = f.i32(24);
  const std::string vtxt = f.str(36, 4);
  char* stop = nullptr;
  const float version = std::strtof(vtxt.c_str(), &stop);
  if (stop != vtxt.c_str() + vtxt.size()) return false;  // the reference's lexical_cast would throw here
  if (!(version > 2.9)) {
    // the reference reads the interleaved v2 samples first and then rejects the file (scf.h:81-95)
    std::cerr << "SCF version greater 2.9 required!" << std::endl;
    return false;
  }
  if (samples < 0 || bases < 0 || samples_at < 0 || bases_at < 0) return false;
  if (!f.has((std::size_t)samples_at, (std::size_t)samples * 8) || !f.has((std::size_t)bases_at, (std::size_t)bases * 4)) return false;
  for (int32_t c = 0; c < 4; ++c) {
    Trace::TMountains& ch = tr.traceACGT[c];
    ch.reserve((std::size_t)samples);
    for (int32_t k = c * samples; k < (c + 1) * samples; ++k) ch.push_back(f.i16((std::size_t)samples_at + 2 * (std::size_t)k));
    for (int pass = 0; pass < 2; ++pass) {
      int16_t carry = 0;
      for (auto& v : ch) {
        v += carry;
        carry = (int16_t)v;
      }
    }
  }
  for (int32_t k = 0; k < bases; ++k) {
    tr.basecallpos.push_back(f.i32((std::size_t)bases_at + 4 * (std::size_t)k));
    tr.qual.push_back(0);
  }
  return true;
}
inline bool readscf(std::string const& filename, Trace& tr) {
  detail::FileBytes f;
  if (!f.load(filename)) { tr.traceACGT.assign(4, Trace::TMountains()); return false; }
  return readscf(f, tr);
}

// traceTxtOut, abif.h:513-533: one line per sample, basecall columns on called samples
template <class Out, typename std::enable_if<!std::is_same<Out, std::string>::value, int>::type = 0>  // (a stream, not a file name)
inline void traceTxtOut(Out& out, BaseCalls const& bc, Trace const& tr, uint32_t leftTrim, uint32_t rightTrim) {
  const uint32_t keep_until = rightTrim < bc.primary.size() ? (uint32_t)bc.primary.size() - rightTrim : 0;
  uint32_t call = 0;
  int32_t next = bc.bcPos[call];
  out << "pos\tpeakA\tpeakC\tpeakG\tpeakT\tbasenum\tprimary\tsecondary\tconsensus\tqual\ttrim" << std::endl;
  const int32_t ns = (int32_t)tr.traceACGT[0].size();
  for (int32_t i = 0; i < ns; ++i) {
    out << (i + 1) << "\t";
    for (int k = 0; k < 4; ++k) out << tr.traceACGT[k][i] << "\t";
    if (next != i) {
      out << "NA\tNA\tNA\tNA\tNA\tNA" << std::endl;
      continue;
    }
    out << (call + 1) << "\t" << bc.primary[call] << "\t" << bc.secondary[call] << "\t" << bc.consensus[call] << "\t"
        << (int32_t)bc.estQual[call] << "\t" << ((call < leftTrim || call >= keep_until) ? "Y" : "N") << std::endl;
    if (call < bc.bcPos.size() - 1) next = bc.bcPos[++call];
  }
}

// the same lines composed in place: a sample's line is at most 5 numbers + 6 short columns, written behind one capacity test
inline void traceTxtOut(TextBuf& out, BaseCalls const& bc, Trace const& tr, uint32_t leftTrim, uint32_t rightTrim) {
  const uint32_t keep_until = rightTrim < bc.primary.size() ? (uint32_t)bc.primary.size() - rightTrim : 0;
  uint32_t call = 0;
  int32_t next = bc.bcPos[call];
  out << "pos\tpeakA\tpeakC\tpeakG\tpeakT\tbasenum\tprimary\tsecondary\tconsensus\tqual\ttrim" << std::endl;
  const int32_t ns = (int32_t)tr.traceACGT[0].size();
  static const char na[] = "NA\tNA\tNA\tNA\tNA\tNA\n";
  for (int32_t i = 0; i < ns; ++i) {
    out.need(160);
    char* q = TextBuf::raw_int(out.p, i + 1);
    *q++ = '\t';
    for (int k = 0; k < 4; ++k) { q = TextBuf::raw_int(q, tr.traceACGT[k][i]); *q++ = '\t'; }
    if (next != i) {
      std::memcpy(q, na, sizeof(na) - 1);
      out.p = q + sizeof(na) - 1;
      continue;
    }
    q = TextBuf::raw_int(q, call + 1);
    *q++ = '\t'; *q++ = bc.primary[call];
    *q++ = '\t'; *q++ = bc.secondary[call];
    *q++ = '\t'; *q++ = bc.consensus[call];
    *q++ = '\t';
    q = TextBuf::raw_int(q, (int32_t)bc.estQual[call]);
    *q++ = '\t'; *q++ = (call < leftTrim || call >= keep_until) ? 'Y' : 'N';
    *q++ = '\n';
    out.p = q;
    if (call < bc.bcPos.size() - 1) next = bc.bcPos[++call];
  }
}

inline void traceTxtOut(std::string const& outfile, BaseCalls const& bc, Trace const& tr, uint32_t leftTrim, uint32_t rightTrim) {
  TextBuf out(48 * tr.traceACGT[0].size() + 4096);
  traceTxtOut(out, bc, tr, leftTrim, rightTrim);
  out.write(outfile);
}

// ---- the build's own ABIF writer (synthetic traces; layout per SURVEY.md Appendix B) ------------------
// Writes the tags readab() consumes: DATA.9-12 in dye order `order` (a permutation of "ACGT"), FWO_.1,
// PLOC.2, PBAS.2, PCON.2 and, when secondary is non-empty, P2BA.1.  Payloads of more than 4 bytes live
// in a data area after the 128-byte header; the directory follows the data area.
inline bool writeab(std::string const& filename, Trace::TACGTMountains const& acgt, std::vector<int32_t> const& peaks,
                    std::string const& primary, std::vector<uint8_t> const& qual, std::string const& secondary = "",
                    std::string const& order = "GATC") {
  struct Slot {
    char name[4];
    int32_t number;
    int16_t etype, esize;
    int32_t count;
    std::vector<uint8_t> payload;
  };
  auto be16 = [](std::vector<uint8_t>& v, int32_t x) { v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x); };
  auto be32 = [](std::vector<uint8_t>& v, int64_t x) { for (int s = 24; s >= 0; s -= 8) v.push_back((uint8_t)(x >> s)); };
  auto slot = [](const char* nm, int32_t num, int16_t et, int16_t es, int32_t cnt) {
    Slot s;
    for (int i = 0; i < 4; ++i) s.name[i] = nm[i];
    s.number = num; s.etype = et; s.esize = es; s.count = cnt;
    return s;
  };
  if (order.size() != 4 || acgt.size() != 4) return false;
  std::vector<Slot> slots;
  for (int i = 0; i < 4; ++i) {
    const int k = order[i] == 'A' ? 0 : order[i] == 'C' ? 1 : order[i] == 'G' ? 2 : order[i] == 'T' ? 3 : -1;
    if (k < 0) return false;
    Slot s = slot("DATA", 9 + i, 4, 2, (int32_t)acgt[k].size());
    for (int32_t v : acgt[k]) be16(s.payload, v);
    slots.push_back(s);
  }
  {
    Slot s = slot("FWO_", 1, 2, 1, 4);
    s.payload.assign(order.begin(), order.end());
    slots.push_back(s);
  }
  {
    Slot s = slot("PLOC", 2, 4, 2, (int32_t)peaks.size());
    for (int32_t v : peaks) be16(s.payload, v);
    slots.push_back(s);
  }
  {
    Slot s = slot("PBAS", 2, 2, 1, (int32_t)primary.size());
    s.payload.assign(primary.begin(), primary.end());
    slots.push_back(s);
  }
  if (!secondary.empty()) {
    Slot s = slot("P2BA", 1, 2, 1, (int32_t)secondary.size());
    s.payload.assign(secondary.begin(), secondary.end());
    slots.push_back(s);
  }
  {
    Slot s = slot("PCON", 2, 2, 1, (int32_t)qual.size());
    s.payload = qual;
    slots.push_back(s);
  }
  std::vector<uint8_t> data;  // payloads > 4 bytes, file offset 128 + position
  std::vector<uint8_t> dir;
  for (Slot const& s : slots) {
    dir.insert(dir.end(), s.name, s.name + 4);
    be32(dir, s.number);
    be16(dir, s.etype);
    be16(dir, s.esize);
    be32(dir, s.count);
    be32(dir, (int64_t)s.payload.size());
    if (s.payload.size() > 4) {
      be32(dir, 128 + (int64_t)data.size());
      data.insert(data.end(), s.payload.begin(), s.payload.end());
    } else {
      for (std::size_t i = 0; i < 4; ++i) dir.push_back(i < s.payload.size() ? s.payload[i] : 0);
    }
    be32(dir, 0);  // data handle
  }
  std::vector<uint8_t> head;
  head.insert(head.end(), {'A', 'B', 'I', 'F'});
  be16(head, 101);
  head.insert(head.end(), {'t', 'd', 'i', 'r'});
  be32(head, 1);
  be16(head, 1023);
  be16(head, 28);
  be32(head, (int64_t)slots.size());
  be32(head, (int64_t)slots.size() * 28);
  be32(head, 128 + (int64_t)data.size());
  be32(head, 0);
  head.resize(128, 0);
  std::ofstream out(filename.c_str(), std::ios::binary);
  if (!out) return false;
  out.write(reinterpret_cast<const char*>(head.data()), (std::streamsize)head.size());
  out.write(reinterpret_cast<const char*>(data.data()), (std::streamsize)data.size());
  out.write(reinterpret_cast<const char*>(dir.data()), (std::streamsize)dir.size());
  return (bool)out;
}

}  // namespace tracy_amd
#endif
