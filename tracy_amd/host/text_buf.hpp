// text_buf.hpp -- a text buffer with the `out << x` interface of the writers (traceTxtOut, the .json / .txt / .fa writers of json.h
// and fmindex.h restated in sage_out.hpp / indigo_out.hpp).  A 1 kb trace is 400 KB of text per output file; through std::ofstream
// with std::endl that is a flush per line and an iostream number conversion per value (12 ms per trace and file).  The writers
// are templates over the stream type: files are composed in a TextBuf (one write per file), tests and other callers may still
// hand in any std::ostream.  Formatting is the default-locale iostream formatting the reference relies on: integers in decimal,
// bool as 0 / 1, char as the character, floating point as "%g" (precision 6).
//
// A file is ~10^5 values of one to five digits: what an insertion costs is what a file costs.  The buffer is a raw cursor into
// one allocation (one capacity test per insertion, no temporary, no std::string bookkeeping) and integers are written two digits
// at a time from a table.
#ifndef TRACY_AMD_TEXT_BUF_HPP
#define TRACY_AMD_TEXT_BUF_HPP

#include <fcntl.h>
#include <unistd.h>

#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <new>
#include <ostream>
#include <sstream>
#include <string>
#include <type_traits>

namespace tracy_amd {

struct TextBuf {
  char* b = nullptr;  // the allocation
  char* p = nullptr;  // cursor
  char* e = nullptr;  // its end
  int width = 0;      // std::setw: right-aligns the next number or string, then resets (as the iostreams do)

  explicit TextBuf(std::size_t reserve = 1 << 16) {
    if (reserve < 256) reserve = 256;
    b = static_cast<char*>(std::malloc(reserve));
    if (!b) throw std::bad_alloc();
    p = b;
    e = b + reserve;
  }
  ~TextBuf() { std::free(b); }
  TextBuf(TextBuf const&) = delete;
  TextBuf& operator=(TextBuf const&) = delete;

  std::size_t size() const { return (std::size_t)(p - b); }
  const char* data() const { return b; }
  std::string str() const { return std::string(b, size()); }

  // room for n more bytes
  void need(std::size_t n) { if ((std::size_t)(e - p) < n) grow(n); }
  __attribute__((noinline)) void grow(std::size_t n) {
    const std::size_t used = size();
    std::size_t cap = 2 * (std::size_t)(e - b);
    if (cap < used + n) cap = used + n;
    char* nb = static_cast<char*>(std::realloc(b, cap));
    if (!nb) throw std::bad_alloc();
    b = nb;
    p = nb + used;
    e = nb + cap;
  }
  void put(const char* src, std::size_t len) {
    if (width > 0) pad(len);
    need(len);
    std::memcpy(p, src, len);
    p += len;
  }
  __attribute__((noinline)) void pad(std::size_t len) {
    if ((std::size_t)width > len) {
      const std::size_t k = (std::size_t)width - len;
      need(k);
      std::memset(p, ' ', k);
      p += k;
    }
    width = 0;
  }

  TextBuf& operator<<(char c) { need(1); *p++ = c; return *this; }
  TextBuf& operator<<(signed char c) { return *this << (char)c; }
  TextBuf& operator<<(unsigned char c) { return *this << (char)c; }
  TextBuf& operator<<(const char* s) { put(s, std::strlen(s)); return *this; }
  TextBuf& operator<<(std::string const& x) { put(x.data(), x.size()); return *this; }
  TextBuf& operator<<(decltype(std::setw(0)) w) { std::ostringstream probe; probe << w; width = (int)probe.width(); return *this; }
  TextBuf& operator<<(bool v) { return *this << (v ? '1' : '0'); }

  static const char* digit_pairs() {
    static const char t[201] =
        "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263646566"
        "676869707172737475767778798081828384858687888990919293949596979899";
    return t;
  }
  static unsigned digits10(uint64_t v) {
    unsigned n = 1;
    for (;;) {
      if (v < 10) return n;
      if (v < 100) return n + 1;
      if (v < 1000) return n + 2;
      if (v < 10000) return n + 3;
      v /= 10000u;
      n += 4;
    }
  }
  // the digits of v at `at` (room for 24 bytes is the caller's business); returns the byte behind them
  static char* raw_unsigned(char* at, uint64_t v) {
    if (v < 100000u) {
      // the values of a trace (signal heights, sample numbers): all five digits are produced, the ones that count are copied -- no
      // branch on the number of digits (which is what a loop costs on values of every length)
      const unsigned x = (unsigned)v, hi = x / 100u, lo = x - 100u * hi, top = hi / 100u, mid = hi - 100u * top;
      const unsigned nd = 1u + (x >= 10u) + (x >= 100u) + (x >= 1000u) + (x >= 10000u);
      const char* t = digit_pairs();
      uint16_t m, l;
      std::memcpy(&m, t + 2 * mid, 2);
      std::memcpy(&l, t + 2 * lo, 2);
      uint64_t five = (uint64_t)('0' + top) | ((uint64_t)m << 8) | ((uint64_t)l << 24);  // the first character in the lowest byte
      five >>= 8u * (5u - nd);
      std::memcpy(at, &five, 8);  // (bytes behind the digits are overwritten by whatever comes next)
      return at + nd;
    }
    const unsigned nd = digits10(v);
    char* q = at + nd;
    const char* t = digit_pairs();
    while (v >= 100) {
      const unsigned r = (unsigned)(v % 100u);
      v /= 100u;
      q -= 2;
      std::memcpy(q, t + 2 * r, 2);
    }
    if (v >= 10) std::memcpy(q - 2, t + 2 * v, 2);
    else q[-1] = (char)('0' + v);
    return at + nd;
  }
  static char* raw_int(char* at, int64_t v) {
    if (v < 0) { *at++ = '-'; return raw_unsigned(at, (uint64_t)0 - (uint64_t)v); }
    return raw_unsigned(at, (uint64_t)v);
  }
  void put_unsigned(uint64_t v, bool neg) {
    if (width > 0) pad(digits10(v) + (neg ? 1u : 0u));
    need(25);
    if (neg) *p++ = '-';
    p = raw_unsigned(p, v);
  }
  template <class T, typename std::enable_if<std::is_integral<T>::value && !std::is_same<T, bool>::value && !std::is_same<T, char>::value &&
                                                 !std::is_same<T, signed char>::value && !std::is_same<T, unsigned char>::value, int>::type = 0>
  TextBuf& operator<<(T v) {
    if (std::is_signed<T>::value && v < 0) put_unsigned((uint64_t)0 - (uint64_t)(int64_t)v, true);
    else put_unsigned((uint64_t)v, false);
    return *this;
  }
  TextBuf& operator<<(double v) {
    char tmp[48];
    const int n = std::snprintf(tmp, sizeof(tmp), "%g", v);
    put(tmp, n > 0 ? (std::size_t)n : 0);
    return *this;
  }
  TextBuf& operator<<(float v) { return *this << (double)v; }
  TextBuf& operator<<(std::ostream& (*)(std::ostream&)) { return *this << '\n'; }  // std::endl (no flush needed)

  // get(0), get(1), ... get(n - 1) (integers) with `sep` between them: the peak arrays of the .json files, 10^4 values each
  template <class Get>
  void int_list(std::size_t n, Get get, const char* sep = ", ") {
    const std::size_t ls = std::strlen(sep);
    for (std::size_t i = 0; i < n; ++i) {
      need(24 + ls);
      if (i) { std::memcpy(p, sep, ls); p += ls; }
      p = raw_int(p, (int64_t)get(i));
    }
  }

  // files that could not be written in full (a missing directory, a full disk): counted for the command's exit code
  static std::atomic<int>& write_errors() { static std::atomic<int> n{0}; return n; }
  // to_file + the error message and the count: what the writers of the commands call
  bool write(std::string const& path) const {
    if (to_file(path)) return true;
    ++write_errors();
    const std::string msg = "Could not write output file " + path + ": " + std::strerror(errno) + "\n";
    (void)!::write(2, msg.data(), msg.size());
    return false;
  }
  // one open / write / close (no stdio buffer in between)
  bool to_file(std::string const& path) const {
    const int fd = ::open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
    if (fd < 0) return false;
    const char* q = b;
    std::size_t left = size();
    bool ok = true;
    while (left) {
      const ssize_t w = ::write(fd, q, left);
      if (w < 0) { if (errno == EINTR) continue; ok = false; break; }
      q += w;
      left -= (std::size_t)w;
    }
    return ::close(fd) == 0 && ok;
  }
};

// the same list on any stream
template <class Out, class Get>
inline void write_int_list(Out& out, std::size_t n, Get get, const char* sep = ", ") {
  for (std::size_t i = 0; i < n; ++i) {
    if (i) out << sep;
    out << get(i);
  }
}
template <class Get>
inline void write_int_list(TextBuf& out, std::size_t n, Get get, const char* sep = ", ") { out.int_list(n, get, sep); }

}  // namespace tracy_amd
#endif
