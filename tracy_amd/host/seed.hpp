// seed.hpp -- host-side anchoring of a trace in an indexed genome (BASELINE
// configs[3]): exact k-mer hits of the basecalled consensus vote for a genome offset and an orientation, and a
// reference window of trace length + 2*maxindel around the winner is handed to the device pipeline.
//
// Mirrors of /root/reference/src (names, argument meaning):
//   findMaxFreq                          fmindex.h:173-198
//   scanSequence                         fmindex.h:203-232
//   getReferenceSlice                    fmindex.h:236-326
// The reference asks an sdsl-lite FM index (csa_wt<>, count / locate; built by `tracy index`, index.h:79-124)
// over the upper-cased genome with one '\n' between contigs.  sdsl-lite is not available here and the stored
// .fm9 format is not reproduced: GenomeIndex below answers the same two queries -- number and positions of the
// exact occurrences of a pattern in that text -- from a sorted k-mer table built in memory (any exact index
// gives identical hit sets; the hits are sorted before use, fmindex.h:180).  PARITY UNPINNED (no sdsl, no
// reference tests); tests/ cross-check against a brute-force search.
// `tracy index` (index.h:79-124) has its counterpart in GenomeIndex::save / open_index: the text, the contig table, the
// bucket directory and the sorted table go to one file (magic TAMDIDX2, sections 8-byte aligned) which later runs -- and
// every rank of a multi-process job on the node -- map read-only instead of rebuilding the table (the page cache holds one copy).
// Sequence lengths follow the reference's convention seqlen = contig length + 1 (the separator).
#ifndef TRACY_AMD_SEED_HPP
#define TRACY_AMD_SEED_HPP

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <thread>
#include <cstdio>
#ifdef TRACY_SEED_PROFILE
#include <x86intrin.h>
#endif
#include <cstdlib>
#include <vector>

#include "sage_out.hpp"

namespace tracy_amd {

// Threads worth starting: the hardware threads, capped by the cgroup CPU quota of the container (a 256-thread host
// that grants 16 CPUs runs 256 compute threads slower than 16).
inline unsigned usable_threads() {
  unsigned n = std::max(1u, std::thread::hardware_concurrency());
  if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char quota[32] = {0};
    unsigned long period = 0;
    if (std::fscanf(f, "%31s %lu", quota, &period) == 2 && period > 0 && std::strcmp(quota, "max") != 0) {
      const unsigned long q = std::strtoul(quota, nullptr, 10);
      if (q > 0) n = std::min<unsigned>(n, (unsigned)std::max<unsigned long>(1, q / period));
    }
    std::fclose(f);
  }
  return n;
}

// the genome text as the index sees it: owned (built from a FASTA) or a view of a mapped index file
class TextView {
 public:
  const char* p = nullptr;
  std::size_t n = 0;
  std::size_t size() const { return n; }
  char operator[](std::size_t i) const { return p[i]; }
  std::string substr(std::size_t pos, std::size_t len = std::string::npos) const {
    if (pos > n) pos = n;
    return std::string(p + pos, std::min(len, n - pos));
  }
  std::size_t find(std::string const& pat, std::size_t from = 0) const {
    if (pat.empty() || from >= n || pat.size() > n - from) return std::string::npos;
    const void* hit = memmem(p + from, n - from, pat.data(), pat.size());
    return hit ? (std::size_t)(static_cast<const char*>(hit) - p) : std::string::npos;
  }
};

class GenomeIndex {
 public:
  // upper-cased contigs joined by '\n' (the "dump" text of index.h:101-116), one trailing '\n'
  TextView text;
  std::vector<std::string> names;   // faidx_iseq: header up to the first whitespace
  std::vector<uint32_t> lengths;    // contig lengths
  std::vector<uint64_t> starts;     // offset of contig i in `text`
  uint32_t k = 0;

  // plain or gzip-compressed multi-FASTA
  bool load(std::string const& path) {
    gzFile f = gzopen(path.c_str(), "rb");
    if (!f) return false;
    unmap();
    std::string& text = owned_text_;  // (built here, viewed through `text` afterwards)
    text.clear(); names.clear(); lengths.clear(); starts.clear();
    std::string line;
    char buf[1 << 16];
    bool first = true;
    auto flush_line = [&]() {
      if (!line.empty() && line.back() == '\r') line.pop_back();
      if (!line.empty() && line[0] == '>') {
        if (!first) { lengths.push_back((uint32_t)(text.size() - starts.back())); text.push_back('\n'); }
        first = false;
        const std::size_t ws = line.find_first_of(" \t");
        names.push_back(line.substr(1, ws == std::string::npos ? std::string::npos : ws - 1));
        starts.push_back(text.size());
      } else if (!first) {
        for (char c : line) text.push_back((char)std::toupper((unsigned char)c));
      }
      line.clear();
    };
    int n;
    while ((n = gzread(f, buf, sizeof(buf))) > 0) {
      for (int i = 0; i < n; ++i) {
        if (buf[i] == '\n') flush_line();
        else line.push_back(buf[i]);
      }
    }
    if (!line.empty()) flush_line();
    gzclose(f);
    if (first) return false;
    lengths.push_back((uint32_t)(text.size() - starts.back()));
    text.push_back('\n');
    this->text.p = owned_text_.data();
    this->text.n = owned_text_.size();
    return true;
  }

  // Table of every k-mer over ACGT (k <= 32).  A k-mer and its reverse complement share one run of the table (the run of the smaller
  // code; inside it the occurrences of that code come first, then -- bit 63 of pos set -- the occurrences of the other one, each part
  // sorted by position): getReferenceSlice looks every window of a trace up on both strands (fmindex.h:259-262), and the second
  // look-up then finds the lines the first one brought into the cache instead of missing twice more into a gigabyte of table.
  void build(uint32_t kmer, uint32_t nthreads = 0) {
    k = kmer;
    std::vector<Entry>& table_ = owned_table_;
    std::vector<uint64_t>& bucket_ = owned_bucket_;
    table_.clear();
    tab_ = nullptr; ntab_ = 0; bkt_ = nullptr;
    if (k == 0 || k > 32 || text.size() < k) return;
    const uint64_t mask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1ull);
    uint64_t code = 0;
    uint32_t valid = 0;
    table_.reserve(text.size());
    for (std::size_t p = 0; p < text.size(); ++p) {
      const int b = base2(text[p]);
      if (b < 0) { valid = 0; code = 0; continue; }
      code = ((code << 2) | (uint64_t)b) & mask;
      if (++valid >= k) {  // filed under the smaller of the k-mer and its reverse complement; bit 63 of pos: the text holds the other one
        const Key q = key_of(code);
        table_.push_back(Entry{q.code, (uint64_t)(p + 1 - k) | (q.flipped ? kFlipped : 0ull)});
      }
    }
    if (nthreads == 0) nthreads = usable_threads();
    // bucket directory over the TRAILING bits of the run's code: a lookup touches one directory slot + one short run.  (The smaller
    // of a k-mer and its reverse complement starts with A or C three times out of four: leading bits would put most of the table
    // into half of the buckets; the trailing letters are as good as uniform.)  The table is sorted by bucket, code, position.
    // directory size: 2^24 slots = 128 MB (three table entries per slot for a 50 Mb genome: one line of the table per look-up, but the
    // directory itself misses every cache); TRACY_AMD_SEED_BUCKET_BITS (development knob, read when an index is BUILT) trades longer
    // buckets -- consecutive lines of the table -- for a directory an L3 holds
    bucket_bits_ = std::min<uint32_t>(2 * k, 24);
    if (const char* e = std::getenv("TRACY_AMD_SEED_BUCKET_BITS")) { const long v = std::atol(e); if (v >= 8 && v <= 24) bucket_bits_ = std::min<uint32_t>(2 * k, (uint32_t)v); }
    sort_table(nthreads);
    bucket_.assign(((std::size_t)1 << bucket_bits_) + 1, 0);
    for (Entry const& e : table_) ++bucket_[slot_of(e.code) + 1];
    for (std::size_t b = 1; b < bucket_.size(); ++b) bucket_[b] += bucket_[b - 1];
    tab_ = table_.data(); ntab_ = table_.size(); bkt_ = bucket_.data();
#ifdef MADV_HUGEPAGE
    auto advise = [](const void* p, std::size_t bytes) {  // (the heap blocks of the two vectors: whole 2 MB pages inside them)
      const uintptr_t huge = (uintptr_t)2 << 20, lo = ((uintptr_t)p + huge - 1) & ~(huge - 1), hi = ((uintptr_t)p + bytes) & ~(huge - 1);
      if (hi > lo) madvise(reinterpret_cast<void*>(lo), hi - lo, MADV_HUGEPAGE);
    };
    advise(tab_, ntab_ * sizeof(Entry));
    advise(bkt_, bucket_.size() * sizeof(uint64_t));
#endif
  }

  // ---- persistence: `tracy index` (index.h:79-124) -------------------------------------------------------------
  // header: magic[8] "TAMDIDX2", u32 k, u32 bucket_bits, u64 text bytes, u64 table entries, u64 contigs, u64 names bytes; then
  // (each padded to 8 bytes) text, names ('\0'-separated), lengths u32[], starts u64[], bucket u64[2^bits + 1], table {code, pos}[]
  bool save(std::string const& path) const {
    if (k == 0 || !tab_ || !bkt_) return false;
    std::FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    std::string nm;
    for (auto const& x : names) { nm += x; nm.push_back('\0'); }
    const uint64_t hdr[6] = {((uint64_t)bucket_bits_ << 32) | k, text.size(), ntab_, names.size(), nm.size(), 0};
    bool ok = std::fwrite("TAMDIDX2", 1, 8, f) == 8 && std::fwrite(hdr, sizeof(hdr), 1, f) == 1;
    auto put = [&](const void* p, std::size_t bytes) {
      static const char zero[8] = {0};
      ok = ok && (bytes == 0 || std::fwrite(p, 1, bytes, f) == bytes);
      const std::size_t pad = (8 - bytes % 8) % 8;
      ok = ok && (pad == 0 || std::fwrite(zero, 1, pad, f) == pad);
    };
    put(text.p, text.size());
    put(nm.data(), nm.size());
    put(lengths.data(), lengths.size() * sizeof(uint32_t));
    put(starts.data(), starts.size() * sizeof(uint64_t));
    put(bkt_, (((std::size_t)1 << bucket_bits_) + 1) * sizeof(uint64_t));
    put(tab_, ntab_ * sizeof(Entry));
    return std::fclose(f) == 0 && ok;
  }
  // a file that begins like an index of this program but not of this format version (TAMDIDX1: one run per k-mer, no strand bit)
  static bool is_stale_index_file(std::string const& path) {
    std::FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    char m[8] = {0};
    const bool stale = std::fread(m, 1, 8, f) == 8 && std::memcmp(m, "TAMD", 4) == 0 && std::memcmp(m, "TAMDIDX2", 8) != 0;
    std::fclose(f);
    return stale;
  }
  static bool is_index_file(std::string const& path) {
    std::FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    char m[8] = {0};
    const bool ok = std::fread(m, 1, 8, f) == 8 && std::memcmp(m, "TAMDIDX2", 8) == 0;
    std::fclose(f);
    return ok;
  }
  // map an index file written by save(); every section is checked against the file size
  bool open_index(std::string const& path) {
    unmap();
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 56) { ::close(fd); return false; }
    void* m = mmap(nullptr, (std::size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0);
    ::close(fd);
    if (m == MAP_FAILED) return false;
    map_ = m; map_bytes_ = (std::size_t)st.st_size;
    const char* base = static_cast<const char*>(m);
    uint64_t hdr[6];
    std::memcpy(hdr, base + 8, sizeof(hdr));
    if (std::memcmp(base, "TAMDIDX2", 8) != 0) { unmap(); return false; }  // (TAMDIDX1: one run per k-mer, no strand bit -- rebuild)
    k = (uint32_t)hdr[0]; bucket_bits_ = (uint32_t)(hdr[0] >> 32);
    const uint64_t tbytes = hdr[1], nt = hdr[2], nc = hdr[3], nmb = hdr[4];
    if (k == 0 || k > 32 || bucket_bits_ > 24 || bucket_bits_ > 2 * k) { unmap(); return false; }
    std::size_t at = 56;
    auto take = [&](uint64_t bytes) -> const char* {
      if (bytes > map_bytes_ || at > map_bytes_ - bytes) return nullptr;
      const char* p = base + at;
      at += (std::size_t)((bytes + 7) & ~7ull);
      return p;
    };
    // the header's counts are a file's word: nothing is sized from them before they are known to fit the file (a count whose
    // byte size wraps 64 bits would pass take())
    const uint64_t fsz = (uint64_t)map_bytes_;
    if (tbytes > fsz || nmb > fsz || nc > fsz / sizeof(uint64_t) || nt > fsz / sizeof(Entry)) { unmap(); return false; }
    const char* ptext = take(tbytes);
    const char* pnm = take(nmb);
    const char* plen = take(nc * sizeof(uint32_t));
    const char* pst = take(nc * sizeof(uint64_t));
    const char* pb = take((((uint64_t)1 << bucket_bits_) + 1) * sizeof(uint64_t));
    const char* pt = take(nt * sizeof(Entry));
    if (!ptext || !pnm || !plen || !pst || !pb || !pt || at > map_bytes_ + 7) { unmap(); return false; }
    text.p = ptext; text.n = (std::size_t)tbytes;
    names.clear(); lengths.assign(reinterpret_cast<const uint32_t*>(plen), reinterpret_cast<const uint32_t*>(plen) + nc);
    starts.assign(reinterpret_cast<const uint64_t*>(pst), reinterpret_cast<const uint64_t*>(pst) + nc);
    for (std::size_t i = 0, b = 0; i < nmb && names.size() < nc; ++i)
      if (pnm[i] == '\0') { names.emplace_back(pnm + b, i - b); b = i + 1; }
    if (names.size() != nc) { unmap(); return false; }
    bkt_ = reinterpret_cast<const uint64_t*>(pb);
    tab_ = reinterpret_cast<const Entry*>(pt);
    ntab_ = (std::size_t)nt;
    // the directory is a prefix sum over the table (range() reads tab_[bkt_[b] .. bkt_[b+1])); contigs lie in the text, in order
    const std::size_t nb = (std::size_t)1 << bucket_bits_;
    bool sane = bkt_[0] == 0 && bkt_[nb] == ntab_;
    for (std::size_t b = 0; sane && b < nb; ++b) sane = bkt_[b] <= bkt_[b + 1];
    uint64_t end = 0;
    for (std::size_t i = 0; sane && i < nc; ++i) {
      sane = starts[i] >= end && starts[i] <= tbytes && lengths[i] <= tbytes - starts[i];
      end = starts[i] + lengths[i];
    }
    if (!sane) { unmap(); names.clear(); lengths.clear(); starts.clear(); return false; }
    make_resident();
    return true;
  }
  // A look-up touches two random lines of a gigabyte of directory + table: on 4 KB pages that is two TLB misses and two page walks
  // per k-mer, more than the cache misses themselves cost once those are prefetched.  The file mapping stays (text, and the table
  // as the fallback), but directory and table are copied into anonymous memory that asks for transparent huge pages (2 MB: the
  // whole table under ~600 TLB entries).  One private copy per process (TRACY_AMD_INDEX_MAPPED=1 keeps the shared file mapping
  // only -- one copy in the page cache for every rank of a node -- at about two thirds of the look-up rate).
  void make_resident() {
    if (std::getenv("TRACY_AMD_INDEX_MAPPED")) return;
    const std::size_t bbytes = (((std::size_t)1 << bucket_bits_) + 1) * sizeof(uint64_t), tbytes = ntab_ * sizeof(Entry);
    const std::size_t huge = (std::size_t)2 << 20;
    const std::size_t total = ((bbytes + huge - 1) / huge + (tbytes + huge - 1) / huge + 1) * huge;
    void* m = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) return;
    char* base = reinterpret_cast<char*>(((uintptr_t)m + huge - 1) & ~(uintptr_t)(huge - 1));
#ifdef MADV_HUGEPAGE
    madvise(base, total - (std::size_t)(base - static_cast<char*>(m)), MADV_HUGEPAGE);
#endif
    char* b = base;
    char* t = base + (bbytes + huge - 1) / huge * huge;
    const unsigned nth = std::min(8u, usable_threads());
    auto copy = [&](char* dst, const char* src, std::size_t bytes) {
      std::vector<std::thread> th;
      for (unsigned i = 0; i < nth; ++i)
        th.emplace_back([=]() { const std::size_t lo = bytes * i / nth, hi = bytes * (i + 1) / nth; std::memcpy(dst + lo, src + lo, hi - lo); });
      for (auto& x : th) x.join();
    };
    copy(b, reinterpret_cast<const char*>(bkt_), bbytes);
    copy(t, reinterpret_cast<const char*>(tab_), tbytes);
    res_ = m; res_bytes_ = total;
    bkt_ = reinterpret_cast<const uint64_t*>(b);
    tab_ = reinterpret_cast<const Entry*>(t);
  }
  ~GenomeIndex() { unmap(); }
  GenomeIndex() = default;
  GenomeIndex(GenomeIndex const&) = delete;
  GenomeIndex& operator=(GenomeIndex const&) = delete;

  // table range of one k-mer code (k-mers over ACGT only)
  void code_range(uint64_t code, std::size_t& lo, std::size_t& hi) const {
    const Key q = key_of(code);
    const std::size_t b = slot_of(q.code);
    std::size_t i = bkt_[b];
    const std::size_t e = bkt_[b + 1];
    while (i < e && tab_[i].code < q.code) ++i;  // buckets hold a handful of entries
    if (q.flipped)
      while (i < e && tab_[i].code == q.code && !(tab_[i].pos & kFlipped)) ++i;  // past the occurrences of the run's own code
    lo = i;
    while (i < e && tab_[i].code == q.code && ((tab_[i].pos & kFlipped) != 0) == q.flipped) ++i;
    hi = i;
  }
  uint64_t position(std::size_t i) const { return tab_[i].pos & ~kFlipped; }
  // cache warm-up for a batch of look-ups: the directory slot first, then (once that is in cache) the table run
  void prefetch_slot(uint64_t code) const { prefetch_slot_key(key_of(code).code); }
  void prefetch_run(uint64_t code) const { prefetch_run_key(key_of(code).code); }
  std::size_t slot_of(uint64_t key) const { return (std::size_t)(key & ((1ull << bucket_bits_) - 1ull)); }
  bool has_table() const { return tab_ != nullptr && bkt_ != nullptr; }
  // (TRACY_AMD_SEED_HINT: development knob -- 0 = into every cache level (prefetcht0, the default), 1 = L2 and below, 2 = L3)
  static int prefetch_hint() { static const int h = [] { const char* e = std::getenv("TRACY_AMD_SEED_HINT"); return e ? std::atoi(e) : 0; }(); return h; }
  static void prefetch_line(const void* p) {
    switch (prefetch_hint()) {
      case 1: __builtin_prefetch(p, 0, 2); break;
      case 2: __builtin_prefetch(p, 0, 1); break;
      default: __builtin_prefetch(p, 0, 3); break;
    }
  }
  // (a look-up reads bkt_[slot] AND bkt_[slot + 1] -- one slot in eight has its neighbour on the next line -- and a bucket of three
  // 16-byte entries crosses a line boundary every other time: both ends of both are requested)
  static bool prefetch_ends() { static const bool v = [] { const char* e = std::getenv("TRACY_AMD_SEED_ENDS"); return !e || std::atoi(e) != 0; }(); return v; }  // (development knob)
  void prefetch_slot_key(uint64_t key) const {
    const uint64_t* p = &bkt_[slot_of(key)];
    prefetch_line(p);
    if (prefetch_ends() && ((reinterpret_cast<uintptr_t>(p) >> 3) & 7u) == 7u) prefetch_line(p + 1);
  }
  void prefetch_run_key(uint64_t key) const {
    const std::size_t b = slot_of(key), i = bkt_[b], e = bkt_[b + 1];
    if (i < e) {
      prefetch_line(&tab_[i]);
      if (prefetch_ends() && ((reinterpret_cast<uintptr_t>(&tab_[i]) ^ reinterpret_cast<uintptr_t>(&tab_[e - 1])) >> 6)) prefetch_line(&tab_[e - 1]);
    }
  }
  // One run, both strands (scanBothStrands): the occurrences of the k-mer whose run `key` names vote into `fwd` (value: position - pf),
  // those of its reverse complement into `rev` (position - pr); flipped: the forward k-mer is the run's second part; a palindrome
  // is its own reverse complement (both lists read the same part).  unique / the < 1000 rule: scanSequence's, per strand.
  void both_strands(uint64_t key, bool flipped, bool palindrome, std::vector<int64_t>* fwd, int64_t pf, std::vector<int64_t>* rev, int64_t pr,
                    bool unique) const {
    const std::size_t b = slot_of(key);
    std::size_t i = bkt_[b];
    const std::size_t e = bkt_[b + 1];
    while (i < e && tab_[i].code < key) ++i;
    const std::size_t own = i;  // [own, mid): the run's own code, [mid, end): the other one
    while (i < e && tab_[i].code == key && !(tab_[i].pos & kFlipped)) ++i;
    const std::size_t mid = i;
    while (i < e && tab_[i].code == key) ++i;
    const std::size_t end = i;
    const std::size_t f_lo = flipped ? mid : own, f_hi = flipped ? end : mid;
    const std::size_t r_lo = palindrome ? f_lo : (flipped ? own : mid), r_hi = palindrome ? f_hi : (flipped ? mid : end);
    auto vote = [&](std::vector<int64_t>* out, std::size_t lo, std::size_t hi, int64_t at) {
      if (!out) return;
      const std::size_t occs = hi - lo;
      if (unique ? occs == 1 : (occs > 0 && occs < 1000))
        for (std::size_t q = lo; q < hi; ++q) out->push_back((int64_t)(tab_[q].pos & ~kFlipped) - at);
    };
    vote(fwd, f_lo, f_hi, pf);
    vote(rev, r_lo, r_hi, pr);
  }
  // the reverse complement of a k-mer code (two bits per letter, first letter in the highest pair)
  static uint64_t revcomp_code(uint64_t x, uint32_t k) {
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0f0f0f0f0f0f0f0full) | ((x & 0x0f0f0f0f0f0f0f0full) << 4);
    x = __builtin_bswap64(x);
    return (~x) >> (64 - 2 * k);
  }
  struct Key { uint64_t code; bool flipped; };  // where a k-mer is filed: the smaller of its code and its reverse complement's
  Key key_of(uint64_t code) const {
    const uint64_t rc = revcomp_code(code, k);
    return rc < code ? Key{rc, true} : Key{code, false};
  }
  static int base_code(char c) { return base2(c); }

  // occurrences of `pat` in the text (exact, over all characters the pattern may hold)
  std::size_t count(std::string const& pat) const {
    std::size_t lo, hi;
    if (range(pat, lo, hi)) return hi - lo;
    return scan(pat, nullptr);
  }
  void locate(std::string const& pat, std::vector<uint64_t>& out) const {
    out.clear();
    std::size_t lo, hi;
    if (range(pat, lo, hi)) {
      for (std::size_t i = lo; i < hi; ++i) out.push_back(tab_[i].pos & ~kFlipped);
      return;
    }
    scan(pat, &out);
  }

 private:
  struct Entry { uint64_t code, pos; };
  static constexpr uint64_t kFlipped = 1ull << 63;
  std::string owned_text_;
  std::vector<Entry> owned_table_;
  std::vector<uint64_t> owned_bucket_;
  const Entry* tab_ = nullptr;     // the sorted table: owned_table_ or a section of the mapped file
  std::size_t ntab_ = 0;
  const uint64_t* bkt_ = nullptr;  // table index of the first entry of every code prefix
  uint32_t bucket_bits_ = 0;
  void* map_ = nullptr;
  std::size_t map_bytes_ = 0;
  void* res_ = nullptr;            // directory + table in anonymous huge-page memory (make_resident)
  std::size_t res_bytes_ = 0;
  void unmap() {
    if (res_) munmap(res_, res_bytes_);
    res_ = nullptr; res_bytes_ = 0;
    if (map_) munmap(map_, map_bytes_);
    map_ = nullptr; map_bytes_ = 0;
    if (text.p != owned_text_.data()) { text.p = nullptr; text.n = 0; }
    tab_ = nullptr; ntab_ = 0; bkt_ = nullptr;
  }

  static int base2(char c) {  // (a table, not a chain of comparisons: on sequence data every comparison is a coin toss for the branch predictor)
    static const struct T { signed char t[256]; T() { std::memset(t, -1, sizeof(t)); t['A'] = 0; t['C'] = 1; t['G'] = 2; t['T'] = 3; } } tab;
    return tab.t[(unsigned char)c];
  }

  // full-length ACGT patterns are answered from the table; anything else (shorter tail patterns when
  // trimRight < kmer, patterns with other letters) by a scan of the text
  bool range(std::string const& pat, std::size_t& lo, std::size_t& hi) const {
    if (pat.size() != k || k == 0) return false;
    uint64_t code = 0;
    for (char c : pat) {
      const int b = base2(c);
      if (b < 0) return false;
      code = (code << 2) | (uint64_t)b;
    }
    if (!tab_) return false;
    code_range(code, lo, hi);
    return true;
  }
  std::size_t scan(std::string const& pat, std::vector<uint64_t>* out) const {
    if (pat.empty()) return 0;
    std::size_t n = 0;
    for (std::size_t p = text.find(pat); p != std::string::npos; p = text.find(pat, p + 1)) {
      ++n;
      if (out) out->push_back(p);
    }
    return n;
  }
  void sort_table(uint32_t nthreads) {
    std::vector<Entry>& table_ = owned_table_;
    const uint64_t low = bucket_bits_ >= 64 ? ~0ull : ((1ull << bucket_bits_) - 1ull);
    auto less = [low](Entry const& a, Entry const& b) {
      const uint64_t sa = a.code & low, sb = b.code & low;
      if (sa != sb) return sa < sb;
      return a.code < b.code || (a.code == b.code && a.pos < b.pos);
    };
    if (nthreads <= 1 || table_.size() < (1u << 16)) {
      std::sort(table_.begin(), table_.end(), less);
      return;
    }
    // sort chunks in parallel, then merge pairwise
    const std::size_t n = table_.size();
    std::vector<std::size_t> cut;
    for (uint32_t t = 0; t <= nthreads; ++t) cut.push_back(n * t / nthreads);
    {
      std::vector<std::thread> th;
      for (uint32_t t = 0; t < nthreads; ++t) th.emplace_back([&, t]() { std::sort(table_.begin() + cut[t], table_.begin() + cut[t + 1], less); });
      for (auto& x : th) x.join();
    }
    while (cut.size() > 2) {
      std::vector<std::size_t> next;
      std::vector<std::thread> th;
      for (std::size_t i = 0; i + 2 < cut.size(); i += 2)
        th.emplace_back([&, i]() { std::inplace_merge(table_.begin() + cut[i], table_.begin() + cut[i + 1], table_.begin() + cut[i + 2], less); });
      for (auto& x : th) x.join();
      for (std::size_t i = 0; i < cut.size(); i += 2) next.push_back(cut[i]);
      if (next.back() != n) next.push_back(n);
      cut.swap(next);
    }
  }
};

#ifdef TRACY_SEED_PROFILE
struct SeedProf {
  unsigned long long t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  ~SeedProf() { const char* nm[8] = {"strings", "keys", "lookups", "maxfreq", "slice", "", "", ""}; unsigned long long tot = 0; for (auto x : t) tot += x;
    for (int i = 0; i < 5; ++i) std::fprintf(stderr, "seedprof %-8s %5.1f %%\n", nm[i], tot ? 100.0 * t[i] / tot : 0.0); }
};
inline SeedProf& seed_prof() { static thread_local SeedProf p; return p; }
#define SEED_LAP(i) do { const unsigned long long now_ = __rdtsc(); seed_prof().t[i] += now_ - lap_; lap_ = now_; } while (0)
#define SEED_LAP_BEGIN() unsigned long long lap_ = __rdtsc()
#else
#define SEED_LAP(i) do { } while (0)
#define SEED_LAP_BEGIN() do { } while (0)
#endif

// findMaxFreq, fmindex.h:173-198: most frequent value of `hits` (sorted in place; smallest value on ties)
inline uint32_t findMaxFreq(std::vector<int64_t>& hits, int64_t& gpos) {
  gpos = 0;
  if (hits.empty()) return 0;
  std::sort(hits.begin(), hits.end());
  int32_t best = 1, run = 1;
  gpos = hits[0];
  for (std::size_t i = 1; i < hits.size(); ++i) {
    if (hits[i] == hits[i - 1]) {
      if (++run > best) { best = run; gpos = hits[i]; }
    } else {
      run = 1;
    }
  }
  return (uint32_t)best;
}

// scanSequence, fmindex.h:203-232: every k-mer of consensus[trimLeft, size - trimRight) without 'N' votes for
// (genome position - offset in the trace).  unique: only k-mers that occur exactly once; otherwise every
// occurrence of k-mers that occur fewer than 1000 times.  The window counter is 16 bits wide as in the reference.
//
// A look-up is two dependent cache misses into a gigabyte of table (directory slot, then the run it points to) and the look-ups of a
// trace are independent of each other: the window codes are computed first, then the loop runs software-pipelined -- the directory
// slot of window i + 2D is requested while the table run of window i + D is requested (its slot has arrived by then) and window i is
// looked up with both lines in cache.  D = 10 keeps ~20 misses in flight, what a core's miss buffers hold (measured on the EPYC 9575F
// of the GPU box: D = 6 .. 12 within 2 %, D = 4 and D >= 16 slower); two full sweeps over all 900 windows before the first look-up
// (rounds 2-3) ran past those buffers and were no faster than no prefetch at all.
inline std::size_t seed_prefetch_distance(std::size_t otherwise) {  // (development knob: the prefetch distance; >= the windows of a trace = full sweeps)
  const char* e = std::getenv("TRACY_AMD_SEED_DISTANCE");
  const long v = e ? std::atol(e) : 0;
  return v >= 1 ? (std::size_t)v : otherwise;
}
inline void scanSequence(GenomeIndex const& idx, std::string const& consensus, uint16_t trimLeft, uint16_t trimRight, uint16_t kmer,
                         std::vector<int64_t>& hits, bool unique) {
  const std::size_t size = consensus.size();
  // the windows of the loop below: p = trimLeft .. (uint16_t wrap-around as in the reference's counter)
  std::size_t nwin = 0;
  for (uint16_t p = trimLeft; (p < (size - trimRight)) && (p < size); ++p) ++nwin;
  if (nwin == 0) return;
  const bool table_ok = kmer == idx.k && kmer >= 1 && kmer <= 32;
  const uint64_t mask = kmer >= 32 ? ~0ull : ((1ull << (2 * kmer)) - 1ull);
  // per window: its 2-bit code when it is a full-length window over ACGT (answered from the table), and whether it holds an 'N'
  thread_local std::vector<uint64_t> codes;
  thread_local std::vector<uint8_t> kind;  // 0: skip (holds an N), 1: table look-up, 2: any other window without N (text scan)
  codes.assign(nwin, 0);
  kind.assign(nwin, 0);
  {
    int32_t ncount = 0;
    for (uint16_t i = trimLeft; (i < trimLeft + kmer) && (i < size); ++i)
      if (consensus[i] == 'N') ++ncount;
    uint64_t code = 0;
    uint32_t run = 0;  // ACGT letters ending at the window's last position (capped at kmer)
    auto push_letter = [&](std::size_t q) {
      const int b = q < size ? GenomeIndex::base_code(consensus[q]) : -1;
      if (b < 0) { run = 0; code = 0; return; }
      code = ((code << 2) | (uint64_t)b) & mask;
      if (run < kmer) ++run;
    };
    for (uint32_t q = trimLeft; q + 1 < (uint32_t)trimLeft + kmer; ++q) push_letter(q);
    std::size_t w = 0;
    for (uint16_t p = trimLeft; (p < (size - trimRight)) && (p < size); ++p, ++w) {
      push_letter((std::size_t)p + kmer - 1);
      if (ncount == 0) {
        if (table_ok && run == kmer) { kind[w] = 1; codes[w] = code; }
        else kind[w] = 2;
      }
      if (consensus[p] == 'N') --ncount;
      if (((uint32_t)(p + kmer) < size) && (consensus[p + kmer] == 'N')) ++ncount;
    }
  }
  std::vector<uint64_t> where;
  static const std::size_t D = seed_prefetch_distance(10);
  for (std::size_t i = 0; i < nwin + 2 * D; ++i) {
    if (i < nwin && kind[i] == 1) idx.prefetch_slot(codes[i]);
    if (i >= D && i - D < nwin && kind[i - D] == 1) idx.prefetch_run(codes[i - D]);
    if (i < 2 * D) continue;
    const std::size_t w = i - 2 * D;
    const uint64_t p = (uint64_t)(uint16_t)(trimLeft + w);
    if (kind[w] == 1) {
      std::size_t lo, hi;
      idx.code_range(codes[w], lo, hi);
      const std::size_t occs = hi - lo;
      if (unique ? occs == 1 : (occs > 0 && occs < 1000))
        for (std::size_t q = lo; q < hi; ++q) hits.push_back((int64_t)(idx.position(q) - p));
    } else if (kind[w] == 2) {
      const std::string seq = consensus.substr((std::size_t)p, kmer);
      const std::size_t occs = idx.count(seq);
      if (unique ? occs == 1 : (occs > 0 && occs < 1000)) {
        idx.locate(seq, where);
        for (uint64_t x : where) hits.push_back((int64_t)(x - p));
      }
    }
  }
}

// The two scans of getReferenceSlice (fmindex.h:259-262: the consensus with trims (l, r), its reverse complement with (r, l)) in one
// pass.  Window p of the consensus and window |consensus| - p - k of the reverse complement are the same letters read on the two
// strands: one k-mer and its reverse complement, which the table files in ONE run (GenomeIndex::build) -- so one directory slot and
// one run answer both, and each look-up is paid for once (two scans pay twice: the second finds the lines in cache, but walks the
// windows, the keys and the buckets again -- half of a trace's seeding time once the misses overlap).  Hits are the two scans' hits
// in another order (findMaxFreq sorts them).  Returns false when the shortcut does not apply -- a consensus with letters outside
// A C G T N, trims shorter than k - 1 (the reference then looks up shorter tail patterns), 16-bit window counters that would wrap
// -- and the caller runs the two scans as they are.
inline bool scanBothStrands(GenomeIndex const& idx, std::string const& consensus, uint16_t trimLeft, uint16_t trimRight, uint16_t kmer,
                            std::vector<int64_t>& hitFwd, std::vector<int64_t>& hitRev, bool unique) {
  const std::size_t S = consensus.size();
  const uint32_t k = kmer;
  if (k != idx.k || k < 1 || k > 32 || !idx.has_table()) return false;
  if (S + k >= 65536u || (uint32_t)trimLeft + 1u < k || (uint32_t)trimRight + 1u < k) return false;
  // A trim longer than the consensus wraps the reference's loop bound (fmindex.h:211 evaluates size - trimRight in size_t) and windows
  // still vote there: the two scans reproduce that, this shortcut does not
  if (S < (std::size_t)trimLeft || S < (std::size_t)trimRight) return false;
  if (S <= (std::size_t)trimLeft + trimRight) return true;  // no window on either strand
  const std::size_t p_lo = (std::size_t)trimLeft + 1u - k, p_hi = S - trimRight;  // windows p_lo .. p_hi - 1; forward: p >= trimLeft, reverse: p + k <= S - trimRight
  const std::size_t nwin = p_hi - p_lo;
  SEED_LAP_BEGIN();
  thread_local std::vector<uint64_t> keys;
  thread_local std::vector<uint8_t> info;  // bit 0: a look-up (no N); bit 1: the forward k-mer is the flipped one of its run; bit 2: palindrome
  keys.assign(nwin, 0);
  info.assign(nwin, 0);
  {
    const uint64_t mask = k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1ull);
    const uint32_t top = 2 * (k - 1);
    uint64_t fw = 0, rc = 0;
    uint32_t run = 0;
    for (std::size_t q = p_lo; q < p_hi - 1 + k; ++q) {
      const char ch = consensus[q];
      const int b = GenomeIndex::base_code(ch);
      if (b < 0) {
        if (ch != 'N') return false;
        run = 0; fw = 0; rc = 0;
      } else {
        fw = ((fw << 2) | (uint64_t)b) & mask;
        rc = (rc >> 2) | ((uint64_t)(3 - b) << top);
        if (run < k) ++run;
      }
      if (q + 1 >= p_lo + k && run == k) {
        const std::size_t w = q + 1 - k - p_lo;
        keys[w] = rc < fw ? rc : fw;
        info[w] = (uint8_t)(1u | (rc < fw ? 2u : 0u) | (rc == fw ? 4u : 0u));
      }
    }
  }
  // (one look-up per window here, not two: 12 misses in flight instead of 20 -- measured on the EPYC 9575F of the GPU box, 16 threads:
  // D = 2 / 4 / 6 / 8 / 10 / 16 -> 147 / 174 / 206-221 / 198 / 175-179 / 147 k traces/s)
  SEED_LAP(1);
  // D: windows between a run's request and its look-up; DS (TRACY_AMD_SEED_SLOT_AHEAD, default D): windows between a directory slot's
  // request and the run's request, which READS the slot -- a demand load that stalls when the slot has not arrived
  // (round 6, 16 threads on the GPU box's EPYC 9575F, three runs each, run-to-run spread +- 8 %: D / DS = 6 / 6 -- the round-5 form -- 14.6-16.5 k
  // traces/s per thread, 6 / 12 16.6-16.9, 8 / 12 15.8-18.1, 10 / 12 18.0, 8 / 16 16.4, 10 / 16 16.8, 12 / 12 15.8; prefetches into L2 / L3 only
  // -- TRACY_AMD_SEED_HINT -- no better)
  // (with both ends of the slot pair and of the bucket requested -- prefetch_slot_key / prefetch_run_key -- the same box, three runs each within
  // 1 %: ends not requested 15.2-15.8 k; D / DS = 8 / 12 20.5-20.6 k, 10 / 12 21.4-21.6, 12 / 12 21.7-21.9, 12 / 16 21.6-21.8, 14 / 14 20.9-21.0,
  // 16 / 16 16.1: the default stays two steps away from that edge)
  static const std::size_t D = seed_prefetch_distance(10);
  static const std::size_t DS = [] { const char* e = std::getenv("TRACY_AMD_SEED_SLOT_AHEAD"); const long v = e ? std::atol(e) : 0; return v >= 1 ? (std::size_t)v : (std::size_t)12; }();
  const std::size_t fwd_from = (std::size_t)trimLeft - p_lo;  // first window the forward scan holds
  const std::size_t rev_until = nwin >= k ? nwin - k + 1 : 0;  // windows [0, rev_until) are the reverse scan's
  for (std::size_t i = 0; i < nwin + D + DS; ++i) {
    if (i < nwin && (info[i] & 1u)) idx.prefetch_slot_key(keys[i]);
    if (i >= DS && i - DS < nwin && (info[i - DS] & 1u)) idx.prefetch_run_key(keys[i - DS]);
    if (i < D + DS) continue;
    const std::size_t w = i - D - DS;
    if (!(info[w] & 1u)) continue;
    const std::size_t p = p_lo + w;
    idx.both_strands(keys[w], (info[w] & 2u) != 0, (info[w] & 4u) != 0, w >= fwd_from ? &hitFwd : nullptr, (int64_t)p,
                     w < rev_until ? &hitRev : nullptr, (int64_t)(S - p - k), unique);
  }
  SEED_LAP(2);
  return true;
}

struct SeedConfig {  // the SageConfig / IndigoConfig fields getReferenceSlice reads
  uint16_t trimLeft = 50, trimRight = 50, kmer = 15, minKmerSupport = 3, maxindel = 1000;
};

// getReferenceSlice, fmindex.h:236-326 for an indexed genome (rs.filetype == 0): orientation and offset by
// k-mer votes (unique hits first, then all hits below 1000 occurrences), then the window
// [chrpos - maxindel, chrpos + |consensus| + maxindel] of that contig (inclusive end, clipped like faidx_fetch_seq),
// reverse-complemented for reverse traces.
inline bool getReferenceSlice(SeedConfig const& c, GenomeIndex const& idx, std::string const& consensus, ReferenceSlice& rs) {
  SEED_LAP_BEGIN();
  // (the vote lists keep their capacity from trace to trace; the reverse complement is only needed where the one-pass scan does not apply)
  static thread_local std::vector<int64_t> hitFwd_tls, hitRev_tls;
  std::vector<int64_t>&hitFwd = hitFwd_tls, &hitRev = hitRev_tls;
  SEED_LAP(0);
  int64_t bestFwd = 0, bestRev = 0, bestPos = 0;
  bool anchored = false;
  for (int pass = 0; pass < 2 && !anchored; ++pass) {
    hitFwd.clear();
    hitRev.clear();
    if (!scanBothStrands(idx, consensus, c.trimLeft, c.trimRight, c.kmer, hitFwd, hitRev, pass == 0)) {
      hitFwd.clear();
      hitRev.clear();
      std::string rv = consensus;
      reverseComplement(rv);
      scanSequence(idx, consensus, c.trimLeft, c.trimRight, c.kmer, hitFwd, pass == 0);
      scanSequence(idx, rv, c.trimRight, c.trimLeft, c.kmer, hitRev, pass == 0);
    }
#ifdef TRACY_SEED_PROFILE
    lap_ = __rdtsc();  // (the scans kept their own laps)
#endif
    const uint32_t freqFwd = findMaxFreq(hitFwd, bestFwd), freqRev = findMaxFreq(hitRev, bestRev);
    SEED_LAP(3);
    if (freqFwd >= c.minKmerSupport && freqFwd > 2 * freqRev) {
      rs.forward = true; rs.kmersupport = freqFwd; bestPos = bestFwd; anchored = true;
    } else if (freqRev >= c.minKmerSupport && freqRev > 2 * freqFwd) {
      rs.forward = false; rs.kmersupport = freqRev; bestPos = bestRev; anchored = true;
    }
  }
  if (!anchored) {
    std::cerr << "Couldn't anchor the Sanger trace in the selected reference genome." << std::endl;
    return false;
  }
  int64_t cumsum = 0;
  uint32_t ref = 0;
  for (; ref + 1 < idx.lengths.size() && bestPos >= cumsum + (int64_t)idx.lengths[ref] + 1; ++ref) cumsum += (int64_t)idx.lengths[ref] + 1;
  const uint32_t seqlen = idx.lengths[ref] + 1;
  rs.chr = idx.names[ref];
  const int64_t chrposSigned = bestPos - cumsum;
  const uint32_t chrpos = chrposSigned > 0 ? (uint32_t)chrposSigned : 0;
  uint32_t slicestart = 0, sliceend = seqlen;
  if (chrpos > c.maxindel) slicestart = chrpos - c.maxindel;
  const uint32_t tmpend = chrpos + (uint32_t)consensus.size() + c.maxindel;
  if (tmpend < seqlen) sliceend = tmpend;
  rs.pos = slicestart;
  // faidx_fetch_seq: inclusive end, clipped to the contig
  const int64_t last = std::min<int64_t>((int64_t)sliceend, (int64_t)idx.lengths[ref] - 1);
  rs.refslice = (int64_t)slicestart <= last ? idx.text.substr(idx.starts[ref] + slicestart, (std::size_t)(last - slicestart + 1)) : std::string();
  if (!rs.forward) reverseComplement(rs.refslice);
  SEED_LAP(4);
  return true;
}

}  // namespace tracy_amd
#endif
