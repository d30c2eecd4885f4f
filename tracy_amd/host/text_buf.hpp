// text_buf.hpp -- a string with the `out << x` interface of the writers (traceTxtOut, the .json / .txt / .fa writers of json.h and
// fmindex.h restated in sage_out.hpp / indigo_out.hpp).  A 1 kb trace is 400 KB of text per output file; through std::ofstream
// with std::endl that is a flush per line and an iostream number conversion per value (12 ms per trace and file).  The writers
// are templates over the stream type: files are composed in a TextBuf (std::to_chars, one write per file), tests and other
// callers may still hand in any std::ostream.  Formatting is the default-locale iostream formatting the reference relies on:
// integers in decimal, bool as 0 / 1, char as the character, floating point as "%g" (precision 6).
#ifndef TRACY_AMD_TEXT_BUF_HPP
#define TRACY_AMD_TEXT_BUF_HPP

#include <charconv>
#include <cstdio>
#include <fstream>
#include <iomanip>
#include <ostream>
#include <sstream>
#include <string>
#include <type_traits>

namespace tracy_amd {

struct TextBuf {
  std::string s;
  int width = 0;  // std::setw: right-aligns the next number or string, then resets (as the iostreams do)
  void pad(std::size_t len) { if (width > 0 && (std::size_t)width > len) s.append((std::size_t)width - len, ' '); width = 0; }
  explicit TextBuf(std::size_t reserve = 1 << 16) { s.reserve(reserve); }
  TextBuf& operator<<(char c) { s.push_back(c); return *this; }
  TextBuf& operator<<(signed char c) { s.push_back((char)c); return *this; }
  TextBuf& operator<<(unsigned char c) { s.push_back((char)c); return *this; }
  TextBuf& operator<<(const char* p) { pad(std::char_traits<char>::length(p)); s.append(p); return *this; }
  TextBuf& operator<<(std::string const& x) { pad(x.size()); s.append(x); return *this; }
  TextBuf& operator<<(decltype(std::setw(0)) w) { std::ostringstream probe; probe << w; width = (int)probe.width(); return *this; }
  TextBuf& operator<<(bool b) { s.push_back(b ? '1' : '0'); return *this; }
  template <class T, typename std::enable_if<std::is_integral<T>::value && !std::is_same<T, bool>::value && !std::is_same<T, char>::value &&
                                                 !std::is_same<T, signed char>::value && !std::is_same<T, unsigned char>::value, int>::type = 0>
  TextBuf& operator<<(T v) {
    char tmp[24];
    const auto r = std::to_chars(tmp, tmp + sizeof(tmp), v);
    pad((std::size_t)(r.ptr - tmp));
    s.append(tmp, r.ptr);
    return *this;
  }
  TextBuf& operator<<(double v) {
    char tmp[48];
    const int n = std::snprintf(tmp, sizeof(tmp), "%g", v);
    pad(n > 0 ? (std::size_t)n : 0);
    s.append(tmp, n > 0 ? (std::size_t)n : 0);
    return *this;
  }
  TextBuf& operator<<(float v) { return *this << (double)v; }
  TextBuf& operator<<(std::ostream& (*)(std::ostream&)) { s.push_back('\n'); return *this; }  // std::endl (no flush needed)
  bool to_file(std::string const& path) const {
    std::FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(s.data(), 1, s.size(), f) == s.size();
    return std::fclose(f) == 0 && ok;
  }
};

}  // namespace tracy_amd
#endif
