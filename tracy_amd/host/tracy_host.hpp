// tracy_host.hpp -- host-side (CPU by design) pieces of the path that the reference keeps on the host
// too: chromatogram types, peak-based basecalling, trace -> probability profile.  north_star: "ABIF/SCF
// parsing, basecalling and FM-index seeding stay on the host".  These feed the device kernels; they
// are not a fallback for them.
//
// Mirrors (names and argument meaning) of /root/reference/src:
//   Trace, BaseCalls                      abif.h:28-57
//   trimmedSeq                            abif.h:68-75
//   iupac(char,char), isAmbiguous         abif.h:135-161
//   basecall(Trace, BaseCalls&, float)    abif.h:408-511
//   findBestTraceSection, estimateQualities   abif.h:164-253
//   createProfile(tr, bc, p, tl, tr)      profile.h:21-52
//   reverseComplementProfile              profile.h:74-90
//   _createProfile(std::string)           align.h:121-136
#ifndef TRACY_AMD_HOST_HPP
#define TRACY_AMD_HOST_HPP

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>

namespace tracy_amd {

struct Trace {  // abif.h:28-43
  typedef int32_t TValue;
  typedef std::vector<TValue> TMountains;
  typedef std::vector<TMountains> TACGTMountains;
  typedef std::vector<uint8_t> TQual;
  std::string basecalls1;  // the file's own primary / secondary calls (ABIF PBAS.2 / P2BA.1); unused by the DP path
  std::string basecalls2;
  TQual qual;
  TMountains basecallpos;
  TACGTMountains traceACGT;  // [4][samples], A C G T
};

struct BaseCalls {  // abif.h:46-57
  typedef std::vector<int32_t> TPosition;
  std::string consensus;
  std::string primary;
  std::string secondary;
  std::string secDecompose;
  TPosition bcPos;
  std::vector<uint8_t> estQual;
};

struct ReferenceSlice {  // fmindex.h:28-37
  bool forward = true;
  int32_t filetype = -1;  // -1 failure, 0 *.fa.gz (indexed genome), 1 *.fa, 2 trace
  uint32_t kmersupport = 0;
  uint32_t pos = 0;
  std::string chr;
  std::string refslice;
};

// float[6][cols], element (k, j) at k*cols + j -- the layout the C ABI takes (TRACYHIP_SEQ_PROFILE)
struct Profile {
  std::vector<float> v;
  std::size_t cols = 0;
  void resize(std::size_t c) { cols = c; v.assign(6 * c, 0.0f); }
  float& operator()(std::size_t k, std::size_t j) { return v[k * cols + j]; }
  float operator()(std::size_t k, std::size_t j) const { return v[k * cols + j]; }
  const float* data() const { return v.data(); }
};

inline std::string trimmedSeq(std::string const& str, uint32_t ltrim, uint32_t rtrim) {
  if ((std::size_t)(uint32_t)(ltrim + rtrim + 1) >= str.size()) return str;
  return str.substr(ltrim, (uint32_t)(str.size() - ltrim - rtrim));
}

inline bool isAmbiguous(char n) { return !(n == 'A' || n == 'C' || n == 'G' || n == 'T'); }

inline char iupac(char one, char two) {
  auto idx = [](char c) { return c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 0; };  // unknown letters keep 0
  int a = idx(one), b = idx(two);
  if (b < a) { int t = a; a = b; b = t; }
  static const char tab[4][4] = {{'N', 'M', 'R', 'W'}, {'N', 'N', 'S', 'Y'}, {'N', 'N', 'N', 'K'}, {'N', 'N', 'N', 'N'}};
  return tab[a][b];
}

namespace detail {
// local maxima inside [floor(s), floor(e)) of every channel, abif.h:77-97
inline bool window_peaks(Trace::TACGTMountains const& tr, float s, float e, int32_t val[4], int32_t pos[4]) {
  const int32_t fs = (int32_t)std::floor(s), fe = (int32_t)std::floor(e);
  if (fs == fe) return false;
  for (int k = 0; k < 4; ++k) {
    std::vector<int32_t> const& t = tr[k];
    int32_t bi = fs, bv = 0;
    const int32_t lo = fs > 1 ? fs : 1;
    const int32_t hi = ((int32_t)t.size() - 1 < fe) ? (int32_t)t.size() - 1 : fe;
    for (int32_t i = lo; i < hi; ++i) {
      const bool rising_edge_top = (t[i - 1] <= t[i]) && (t[i] > t[i + 1]);
      const bool plateau_end = (t[i - 1] < t[i]) && (t[i] >= t[i + 1]);
      if ((rising_edge_top || plateau_end) && t[i] > bv) { bi = i; bv = t[i]; }
    }
    val[k] = bv;
    pos[k] = bi;
  }
  return true;
}
}  // namespace detail

// findBestTraceSection, abif.h:164-220: per-base penalty = ambiguous secondary calls in a window of
// `win` + deviation of the peak spacing from its mean; returns (centre of the best 10 % stretch, its
// mean penalty).  All index arithmetic is 32-bit unsigned as in the reference.
inline std::pair<uint32_t, double> findBestTraceSection(BaseCalls const& bc, std::vector<int32_t>& penalty, uint32_t win) {
  const uint32_t n = (uint32_t)bc.secondary.size();
  const uint32_t half = win / 2;
  int32_t amb = 0;
  for (uint32_t i = 0; i < win && i < n; ++i) amb += isAmbiguous(bc.secondary[i]) ? 1 : 0;
  for (uint32_t i = 0; i < half && i < n; ++i) penalty[i] = amb;
  for (uint32_t i = win; i < n; ++i) {
    amb += (isAmbiguous(bc.secondary[i]) ? 1 : 0) - (isAmbiguous(bc.secondary[i - win]) ? 1 : 0);
    penalty[i - half] = amb;
  }
  for (uint32_t i = n - half; i < n; ++i) penalty[i] = amb;

  double mean = 0;
  for (uint32_t i = 1; i < n; ++i) mean += (bc.bcPos[i] - bc.bcPos[i - 1]);
  mean /= (bc.secondary.size() - 1);

  uint32_t spread = 0;
  for (uint32_t i = 0; i + win < n; ++i) {
    uint32_t last = i > 0 ? (uint32_t)bc.bcPos[i - 1] : 0;
    uint32_t lo = (uint32_t)bc.bcPos[n - 1], hi = 0;
    for (uint32_t k = 0; k < win; ++k) {
      const uint32_t d = (uint32_t)bc.bcPos[i + k] - last;
      last = (uint32_t)bc.bcPos[i + k];
      if (d < lo) lo = d;
      if (d > hi) hi = d;
    }
    spread = (uint32_t)(int32_t)((std::abs((double)hi - mean) + std::abs((double)lo - mean)) / 2);
    penalty[i + half] = (int32_t)((uint32_t)penalty[i + half] + spread);
    if (i == 0)
      for (uint32_t k = 0; k < half; ++k) penalty[k] = (int32_t)((uint32_t)penalty[k] + spread);
  }
  for (uint32_t i = n - half; i < n; ++i) penalty[i] = (int32_t)((uint32_t)penalty[i] + spread);

  const uint32_t stretch = (uint32_t)(int32_t)(0.1 * bc.secondary.size());
  uint32_t best_at = 0;
  int32_t best = 99999999;
  int32_t sum = 0;  // penalty[i .. i + stretch): the reference adds the stretch up at every i; the same integers, kept running
  for (uint32_t k = 0; k < stretch && k < n; ++k) sum += penalty[k];
  for (uint32_t i = 0; i + stretch < n; ++i) {
    if (i) sum += penalty[i + stretch - 1] - penalty[i - 1];
    if (sum < best) {
      best = sum;
      best_at = i + (uint32_t)(int32_t)(stretch / 2);
    }
  }
  return std::make_pair(best_at, (double)best / (double)stretch);
}

inline uint32_t findBestTraceSection(BaseCalls const& bc) {  // abif.h:222-229
  std::vector<int32_t> penalty(bc.secondary.size(), 0);
  return findBestTraceSection(bc, penalty, 10).first;
}

// estimateQualities, abif.h:232-253: penalties rescaled to 60..0
inline void estimateQualities(BaseCalls& bc) {
  bc.estQual.resize(bc.primary.size(), 0);
  std::vector<int32_t> penalty(bc.secondary.size(), 0);
  findBestTraceSection(bc, penalty, 10);
  int32_t top = 0;
  for (int32_t v : penalty)
    if (v >= top) top = v;
  const double scaling = 60.0 / (double)top;
  for (std::size_t i = 0; i < penalty.size(); ++i) {
    const double q = 60.0 - scaling * (double)penalty[i];
    int32_t v = (q != q) ? 0 : (q < 0 ? 0 : q > 60 ? 60 : (int32_t)q);  // NaN (all penalties 0) truncates below 0 on x86
    bc.estQual[i] = (uint8_t)v;
  }
}

// basecall(), abif.h:408-511
inline void basecall(Trace const& tr, BaseCalls& bc, float sigratio) {
  static const char letters[4] = {'A', 'C', 'G', 'T'};
  const std::size_t np = tr.basecallpos.size();
  bc = BaseCalls();
  if (np == 0) return;
  // window borders: midpoints between consecutive called positions (float storage, double arithmetic)
  std::vector<float> st(np), ed(np);
  int32_t prev = 0, diff = 0;
  for (std::size_t i = 0; i < np; ++i) {
    diff = tr.basecallpos[i] - prev;
    st[i] = (float)((float)tr.basecallpos[i] - 0.5 * (float)diff);
    if (i > 0) ed[i - 1] = (float)((float)tr.basecallpos[i - 1] + 0.5 * (float)diff);
    prev = tr.basecallpos[i];
  }
  ed[np - 1] = (float)(tr.basecallpos[np - 1] + 0.5 * diff);

  for (std::size_t i = 0; i < np; ++i) {
    int32_t pv[4], pi[4];
    if (!detail::window_peaks(tr.traceACGT, st[i], ed[i], pv, pi)) continue;
    int32_t mid = (int32_t)((st[i] + ed[i]) / 2.0);
    if (mid >= std::floor(ed[i])) mid = (int32_t)std::floor(st[i]);
    int32_t est = 1;
    for (int k = 0; k < 4; ++k) if (tr.traceACGT[k][mid] > est) est = tr.traceACGT[k][mid];
    const int32_t threshold = (int32_t)(sigratio * est);
    if (pv[0] <= threshold && pv[1] <= threshold && pv[2] <= threshold && pv[3] <= threshold) {
      for (int k = 0; k < 4; ++k) { pi[k] = mid; pv[k] = tr.traceACGT[k][mid]; }  // no peak: take the midpoint
    }
    int32_t top = 1;
    for (int k = 0; k < 4; ++k) if (pv[k] > top) top = pv[k];
    float ratio[4];
    for (int k = 0; k < 4; ++k) ratio[k] = (float)pv[k] / (float)top;
    float best = sigratio;
    int32_t sel = -1, selpos = pi[0], valid = 0;
    for (int k = 0; k < 4; ++k) {
      if (ratio[k] >= sigratio) {
        ++valid;
        if (ratio[k] >= best) { best = ratio[k]; selpos = pi[k]; sel = k; }  // last wins on exact ties
      }
    }
    bc.bcPos.push_back(selpos);
    if (valid == 4 || sel == -1) {
      bc.primary.push_back('N'); bc.secondary.push_back('N'); bc.consensus.push_back('N');
    } else if (valid > 1) {
      bc.primary.push_back(letters[sel]);
      int rest[3], nr = 0;
      for (int k = 0; k < 4; ++k) if (k != sel && ratio[k] >= sigratio) rest[nr++] = k;
      bc.secondary.push_back(nr == 1 ? letters[rest[0]] : iupac(letters[rest[0]], letters[rest[1]]));
      bc.consensus.push_back('N');
    } else {
      bc.primary.push_back(letters[sel]); bc.secondary.push_back(letters[sel]); bc.consensus.push_back(letters[sel]);
    }
  }
  estimateQualities(bc);
}

namespace detail {
inline bool called(uint32_t k, char p, char s) {  // _inBaseCalled, profile.h:7-19
  static const char* sets[4] = {"ARWM", "CYSM", "GRSK", "TYWK"};
  for (const char* c = sets[k]; *c; ++c) if (p == *c || s == *c) return true;
  return false;
}
}  // namespace detail

// createProfile(Trace, BaseCalls, p, trimleft, trimright), profile.h:21-52
inline void createProfile(Trace const& tr, BaseCalls const& bc, Profile& p, int32_t trimleft = 0, int32_t trimright = 0) {
  if (trimleft + trimright >= (int32_t)bc.bcPos.size()) { trimleft = 0; trimright = 0; }
  const int32_t sz = (int32_t)bc.bcPos.size() - (trimleft + trimright);
  p.resize((std::size_t)sz);
  for (int32_t j = trimleft; j < trimleft + sz; ++j) {
    const int32_t pos = bc.bcPos[j];
    float totalsig = 0, allsig = 0;
    bool in[4];
    for (uint32_t k = 0; k < 4; ++k) {
      in[k] = detail::called(k, bc.primary[j], bc.secondary[j]);
      allsig += tr.traceACGT[k][pos];
      if (in[k]) totalsig += tr.traceACGT[k][pos];
    }
    const std::size_t o = (std::size_t)(j - trimleft);
    if (totalsig == 0) {
      for (uint32_t k = 0; k < 4; ++k) p(k, o) = 0.25;
    } else {
      const float normfac = totalsig / allsig;
      for (uint32_t k = 0; k < 4; ++k) {
        const float frac = in[k] ? ((float)tr.traceACGT[k][pos] / totalsig) : 0.0f;
        p(k, o) = normfac * frac + (1 - normfac) * 0.25;  // float*float + float*double -> double -> float
      }
    }
  }
}

// reverseComplementProfile, profile.h:74-90
inline void reverseComplementProfile(Profile const& p, Profile& out) {
  out.resize(p.cols);
  static const int src[6] = {3, 2, 1, 0, 4, 5};
  for (std::size_t j = 0; j < p.cols; ++j)
    for (int k = 0; k < 6; ++k) out(k, j) = p(src[k], p.cols - 1 - j);
}

// _createProfile(std::string), align.h:121-136
inline void createProfile(std::string const& s, Profile& p) {
  p.resize(s.size());
  for (std::size_t j = 0; j < s.size(); ++j) {
    switch (s[j]) {
      case 'A': case 'a': p(0, j) = 1; break;
      case 'C': case 'c': p(1, j) = 1; break;
      case 'G': case 'g': p(2, j) = 1; break;
      case 'T': case 't': p(3, j) = 1; break;
      case 'N': case 'n': p(4, j) = 1; break;
      case '-': p(5, j) = 1; break;
      default: break;
    }
  }
}

}  // namespace tracy_amd
#endif
