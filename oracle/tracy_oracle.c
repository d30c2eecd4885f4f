/*
 * tracy_oracle.c -- CPU restatement of the tracy alignment hot path (TEST INFRASTRUCTURE ONLY).
 * See tracy_oracle.h for the parity status.  Every function cites the reference lines it follows
 * (paths relative to /root/reference/src).  Structure mirrors the reference on purpose: row-major
 * scalar loops, two rolling int rows, four packed trace bit-planes, state-machine traceback, so that
 * timing it (bench.py cpu_baseline, kind "port") is representative of the reference CPU path
 * (Makefile:45 release flags: -O3 -fno-tree-vectorize, no -march).
 */
#include "tracy_oracle.h"

#include <stdlib.h>
#include <string.h>

#define ALWAYS_INLINE static inline __attribute__((always_inline))

/* ---------------------------------------------------------------------------------------------
 * End-gap policy: _horizontalGap / _verticalGap (align.h:52-80).  With the "free" flag the cost is
 * 0 when the index sits on the first or last row/column.
 * ------------------------------------------------------------------------------------------- */
ALWAYS_INLINE int32_t gap_cost(int freeflag, size_t i, size_t iend, int32_t cost) {
  if (freeflag && ((i == 0) || (i == iend))) return 0;
  return cost;
}

/* packed bit-plane, same storage idea as boost::dynamic_bitset<unsigned long> (gotoh.h:87-91) */
typedef struct {
  uint64_t* w;
} bitplane;

static int bp_alloc(bitplane* b, size_t nbits) {
  b->w = (uint64_t*)calloc((nbits + 63) / 64 + 1, sizeof(uint64_t));
  return b->w != NULL;
}
ALWAYS_INLINE void bp_set(bitplane* b, size_t i) { b->w[i >> 6] |= (uint64_t)1 << (i & 63); }
ALWAYS_INLINE int bp_get(const bitplane* b, size_t i) { return (int)((b->w[i >> 6] >> (i & 63)) & 1); }

/* ---------------------------------------------------------------------------------------------
 * Substitution scores.
 *   string  : align.h:96-101  -- raw byte compare.
 *   profile : align.h:103-118 -- float accumulate, k1 outer / k2 inner, k<5, then (int) truncation.
 *             In gotoh the profiles are float (gotoh.h:27), in needle double (needle.h:26).
 * ------------------------------------------------------------------------------------------- */
enum { SUB_STR = 0, SUB_PROF_F = 1, SUB_PROF_D = 2 };

typedef struct {
  const char* s1;
  const char* s2;
  const float* p1;  /* [6][m] */
  const float* p2;  /* [6][n] */
  const double* d1; /* [6][m] (needle) */
  const double* d2; /* [6][n] */
  size_t m, n;
} sub_ctx;

ALWAYS_INLINE int32_t sub_score(int kind, const sub_ctx* x, size_t row, size_t col, const orc_score* sc) {
  if (kind == SUB_STR) {
    return (x->s1[row] == x->s2[col]) ? sc->match : sc->mismatch;
  } else if (kind == SUB_PROF_F) {
    float score = 0;
    for (int k1 = 0; k1 < 5; ++k1)
      for (int k2 = 0; k2 < 5; ++k2)
        score += x->p1[(size_t)k1 * x->m + row] * x->p2[(size_t)k2 * x->n + col] *
                 ((k1 == k2) ? sc->match : sc->mismatch);
    return (int32_t)score;
  } else {
    /* TProfile = multi_array<double,2>: the product is double, the accumulator stays float */
    float score = 0;
    for (int k1 = 0; k1 < 5; ++k1)
      for (int k2 = 0; k2 < 5; ++k2)
        score += x->d1[(size_t)k1 * x->m + row] * x->d2[(size_t)k2 * x->n + col] *
                 ((k1 == k2) ? sc->match : sc->mismatch);
    return (int32_t)score;
  }
}

/* ---------------------------------------------------------------------------------------------
 * gotohScore (gotoh.h:12-68) and gotoh (gotoh.h:71-174) in one body; with_trace selects the
 * bit-plane bookkeeping (gotoh.h:135-138) and the traceback (gotoh.h:143-167).
 * ------------------------------------------------------------------------------------------- */
ALWAYS_INLINE int32_t gotoh_core(int kind, const sub_ctx* x, int hfree, int vfree, const orc_score* sc,
                                 int with_trace, char* btr, size_t* btr_len) {
  const size_t m = x->m, n = x->n;
  const int32_t inf = ORC_INF;
  int32_t* s = (int32_t*)calloc(n + 1, sizeof(int32_t));
  int32_t* v = (int32_t*)calloc(n + 1, sizeof(int32_t));
  int32_t newhoz = 0, prevsub = 0;
  const size_t mf = n + 1;
  bitplane bit1 = {0}, bit2 = {0}, bit3 = {0}, bit4 = {0};
  if (with_trace) {
    bp_alloc(&bit1, (m + 1) * (n + 1));
    bp_alloc(&bit2, (m + 1) * (n + 1));
    bp_alloc(&bit3, (m + 1) * (n + 1));
    bp_alloc(&bit4, (m + 1) * (n + 1));
  }

  for (size_t row = 0; row <= m; ++row) {
    for (size_t col = 0; col <= n; ++col) {
      if ((row == 0) && (col == 0)) { /* gotoh.h:106-111 */
        s[0] = 0;
        v[0] = -inf;
        newhoz = -inf;
        if (with_trace) { bp_set(&bit1, 0); bp_set(&bit2, 0); }
      } else if (row == 0) { /* gotoh.h:112-116; the size_t arithmetic wraps to the int value */
        v[col] = -inf;
        s[col] = gap_cost(hfree, 0, m, (int32_t)(sc->go + (int64_t)col * sc->ge));
        newhoz = s[col];
        if (with_trace) bp_set(&bit3, col);
      } else if (col == 0) { /* gotoh.h:117-123 */
        newhoz = -inf;
        s[0] = gap_cost(vfree, 0, n, (int32_t)(sc->go + (int64_t)row * sc->ge));
        if (row - 1 == 0) prevsub = 0;
        else prevsub = gap_cost(vfree, 0, n, (int32_t)(sc->go + (int64_t)(row - 1) * sc->ge));
        v[0] = s[0];
        if (with_trace) bp_set(&bit4, row * mf);
      } else { /* gotoh.h:124-139 */
        int32_t prevhoz = newhoz;
        int32_t prevver = v[col];
        int32_t prevprevsub = prevsub;
        prevsub = s[col];
        int32_t hopen = s[col - 1] + gap_cost(hfree, row, m, sc->go + sc->ge);
        int32_t hext = prevhoz + gap_cost(hfree, row, m, sc->ge);
        newhoz = hopen > hext ? hopen : hext;
        int32_t vopen = prevsub + gap_cost(vfree, col, n, sc->go + sc->ge);
        int32_t vext = prevver + gap_cost(vfree, col, n, sc->ge);
        v[col] = vopen > vext ? vopen : vext;
        int32_t d = prevprevsub + sub_score(kind, x, row - 1, col - 1, sc);
        int32_t best = d > newhoz ? d : newhoz;
        s[col] = best > v[col] ? best : v[col];
        if (with_trace) {
          if (s[col] == newhoz) bp_set(&bit3, row * mf + col);
          else if (s[col] == v[col]) bp_set(&bit4, row * mf + col);
          if (newhoz != hext) bp_set(&bit1, row * mf + col);
          if (v[col] != vext) bp_set(&bit2, row * mf + col);
        }
      }
    }
  }
  int32_t result = s[n];

  if (with_trace) { /* gotoh.h:143-167 */
    size_t row = m, col = n, k = 0;
    char last = 's';
    while ((row > 0) || (col > 0)) {
      if (last == 's') {
        if (bp_get(&bit3, row * mf + col)) last = 'h';
        else if (bp_get(&bit4, row * mf + col)) last = 'v';
        else { --row; --col; btr[k++] = 's'; }
      } else if (last == 'h') {
        if (bp_get(&bit1, row * mf + col)) last = 's';
        --col;
        btr[k++] = 'h';
      } else {
        if (bp_get(&bit2, row * mf + col)) last = 's';
        --row;
        btr[k++] = 'v';
      }
    }
    *btr_len = k;
    free(bit1.w); free(bit2.w); free(bit3.w); free(bit4.w);
  }
  free(s);
  free(v);
  return result;
}

/* ---------------------------------------------------------------------------------------------
 * needleScore (needle.h:12-57) / needle (needle.h:59-138): linear gap cost ge, two bit-planes,
 * trace preference h > v > diag (needle.h:105-109), traceback needle.h:113-131.
 * ------------------------------------------------------------------------------------------- */
ALWAYS_INLINE int32_t needle_core(int kind, const sub_ctx* x, int hfree, int vfree, const orc_score* sc,
                                  int with_trace, char* btr, size_t* btr_len) {
  const size_t m = x->m, n = x->n;
  int32_t* s = (int32_t*)calloc(n + 1, sizeof(int32_t));
  int32_t prevsub = 0;
  const size_t mf = n + 1;
  bitplane bit3 = {0}, bit4 = {0};
  if (with_trace) {
    bp_alloc(&bit3, (m + 1) * (n + 1));
    bp_alloc(&bit4, (m + 1) * (n + 1));
  }
  for (size_t row = 0; row <= m; ++row) {
    for (size_t col = 0; col <= n; ++col) {
      if ((row == 0) && (col == 0)) {
        s[0] = 0;
        prevsub = 0;
      } else if (row == 0) {
        s[col] = gap_cost(hfree, 0, m, (int32_t)((int64_t)col * sc->ge));
        if (with_trace) bp_set(&bit3, col);
      } else if (col == 0) {
        s[0] = gap_cost(vfree, 0, n, (int32_t)((int64_t)row * sc->ge));
        if (row - 1 == 0) prevsub = 0;
        else prevsub = gap_cost(vfree, 0, n, (int32_t)((int64_t)(row - 1) * sc->ge));
        if (with_trace) bp_set(&bit4, row * mf);
      } else {
        int32_t prevprevsub = prevsub;
        prevsub = s[col];
        int32_t d = prevprevsub + sub_score(kind, x, row - 1, col - 1, sc);
        int32_t ver = prevsub + gap_cost(vfree, col, n, sc->ge);
        int32_t hor = s[col - 1] + gap_cost(hfree, row, m, sc->ge);
        int32_t best = d > ver ? d : ver;
        s[col] = best > hor ? best : hor;
        if (with_trace) {
          if (s[col] == hor) bp_set(&bit3, row * mf + col);
          else if (s[col] == ver) bp_set(&bit4, row * mf + col);
        }
      }
    }
  }
  int32_t result = s[n];
  if (with_trace) {
    size_t row = m, col = n, k = 0;
    while ((row > 0) || (col > 0)) {
      if (bp_get(&bit3, row * mf + col)) { --col; btr[k++] = 'h'; }
      else if (bp_get(&bit4, row * mf + col)) { --row; btr[k++] = 'v'; }
      else { --row; --col; btr[k++] = 's'; }
    }
    *btr_len = k;
    free(bit3.w); free(bit4.w);
  }
  free(s);
  return result;
}

/* ---- public wrappers ------------------------------------------------------------------------- */
int32_t orc_gotoh_score_str(const char* s1, size_t m, const char* s2, size_t n, int hfree, int vfree,
                            const orc_score* sc) {
  sub_ctx x = {s1, s2, 0, 0, 0, 0, m, n};
  return gotoh_core(SUB_STR, &x, hfree, vfree, sc, 0, 0, 0);
}
int32_t orc_gotoh_str(const char* s1, size_t m, const char* s2, size_t n, int hfree, int vfree,
                      const orc_score* sc, char* btr, size_t* btr_len) {
  sub_ctx x = {s1, s2, 0, 0, 0, 0, m, n};
  return gotoh_core(SUB_STR, &x, hfree, vfree, sc, 1, btr, btr_len);
}
int32_t orc_gotoh_score_prof(const float* p1, size_t m, const float* p2, size_t n, int hfree, int vfree,
                             const orc_score* sc) {
  sub_ctx x = {0, 0, p1, p2, 0, 0, m, n};
  return gotoh_core(SUB_PROF_F, &x, hfree, vfree, sc, 0, 0, 0);
}
int32_t orc_gotoh_prof(const float* p1, size_t m, const float* p2, size_t n, int hfree, int vfree,
                       const orc_score* sc, char* btr, size_t* btr_len) {
  sub_ctx x = {0, 0, p1, p2, 0, 0, m, n};
  return gotoh_core(SUB_PROF_F, &x, hfree, vfree, sc, 1, btr, btr_len);
}
int32_t orc_needle_score_str(const char* s1, size_t m, const char* s2, size_t n, int hfree, int vfree,
                             const orc_score* sc) {
  sub_ctx x = {s1, s2, 0, 0, 0, 0, m, n};
  return needle_core(SUB_STR, &x, hfree, vfree, sc, 0, 0, 0);
}
int32_t orc_needle_str(const char* s1, size_t m, const char* s2, size_t n, int hfree, int vfree,
                       const orc_score* sc, char* btr, size_t* btr_len) {
  sub_ctx x = {s1, s2, 0, 0, 0, 0, m, n};
  return needle_core(SUB_STR, &x, hfree, vfree, sc, 1, btr, btr_len);
}

/* _createProfile(multi_array<float,2>, TProfile<double>) align.h:183-194: widening copy */
static double* widen_profile(const float* p, size_t len) {
  double* d = (double*)malloc(sizeof(double) * 6 * (len ? len : 1));
  for (size_t i = 0; i < 6 * len; ++i) d[i] = (double)p[i];
  return d;
}
int32_t orc_needle_score_prof(const float* p1, size_t m, const float* p2, size_t n, int hfree, int vfree,
                              const orc_score* sc) {
  double* d1 = widen_profile(p1, m);
  double* d2 = widen_profile(p2, n);
  sub_ctx x = {0, 0, 0, 0, d1, d2, m, n};
  int32_t r = needle_core(SUB_PROF_D, &x, hfree, vfree, sc, 0, 0, 0);
  free(d1); free(d2);
  return r;
}
int32_t orc_needle_prof(const float* p1, size_t m, const float* p2, size_t n, int hfree, int vfree,
                        const orc_score* sc, char* btr, size_t* btr_len) {
  double* d1 = widen_profile(p1, m);
  double* d2 = widen_profile(p2, n);
  sub_ctx x = {0, 0, 0, 0, d1, d2, m, n};
  int32_t r = needle_core(SUB_PROF_D, &x, hfree, vfree, sc, 1, btr, btr_len);
  free(d1); free(d2);
  return r;
}

/* ---- alignment materialisation ---------------------------------------------------------------- */
/* _createLocalAlignment / _createAlignment(string), align.h:196-223: walk btr in reverse. */
void orc_create_alignment_str(const char* btr, size_t L, const char* s1, const char* s2, char* row0,
                              char* row1) {
  size_t row = 0, col = 0;
  for (size_t ai = 0; ai < L; ++ai) {
    char op = btr[L - 1 - ai];
    if (op == 's') { row0[ai] = s1[row++]; row1[ai] = s2[col++]; }
    else if (op == 'h') { row0[ai] = '-'; row1[ai] = s2[col++]; }
    else { row0[ai] = s1[row++]; row1[ai] = '-'; }
  }
}

/* _profileConsChar, align.h:254-270: argmax over the 6 rows, first maximum wins, compare in double */
char orc_profile_cons_char(const float* p, size_t len, size_t pos) {
  uint32_t maxidx = 0;
  double maxval = p[pos];
  for (uint32_t k = 1; k < 6; ++k) {
    if (p[(size_t)k * len + pos] > maxval) {
      maxval = p[(size_t)k * len + pos];
      maxidx = k;
    }
  }
  if (maxidx == 0) return 'A';
  else if (maxidx == 1) return 'C';
  else if (maxidx == 2) return 'G';
  else if (maxidx == 3) return 'T';
  return 'N';
}

/* _createAlignment(float profiles), align.h:272-293 */
void orc_create_alignment_prof(const char* btr, size_t L, const float* p1, size_t m, const float* p2,
                               size_t n, char* row0, char* row1) {
  size_t row = 0, col = 0;
  for (size_t ai = 0; ai < L; ++ai) {
    char op = btr[L - 1 - ai];
    if (op == 's') {
      row0[ai] = orc_profile_cons_char(p1, m, row++);
      row1[ai] = orc_profile_cons_char(p2, n, col++);
    } else if (op == 'h') {
      row0[ai] = '-';
      row1[ai] = orc_profile_cons_char(p2, n, col++);
    } else {
      row0[ai] = orc_profile_cons_char(p1, m, row++);
      row1[ai] = '-';
    }
  }
}

/* ---- profiles ---------------------------------------------------------------------------------- */
/* _createProfile(std::string), align.h:121-136 */
void orc_create_profile_str(const char* s, size_t n, float* p) {
  for (size_t j = 0; j < n; ++j) {
    for (int k = 0; k < 6; ++k) p[(size_t)k * n + j] = 0;
    char c = s[j];
    if ((c == 'A') || (c == 'a')) p[0 * n + j] += 1;
    else if ((c == 'C') || (c == 'c')) p[1 * n + j] += 1;
    else if ((c == 'G') || (c == 'g')) p[2 * n + j] += 1;
    else if ((c == 'T') || (c == 't')) p[3 * n + j] += 1;
    else if ((c == 'N') || (c == 'n')) p[4 * n + j] += 1;
    else if (c == '-') p[5 * n + j] += 1;
  }
}

/* _inBaseCalled, profile.h:7-19 */
static int in_base_called(uint32_t k, char p, char s) {
  if (k == 0) return (p == 'A') || (p == 'R') || (p == 'W') || (p == 'M') || (s == 'A') || (s == 'R') || (s == 'W') || (s == 'M');
  if (k == 1) return (p == 'C') || (p == 'Y') || (p == 'S') || (p == 'M') || (s == 'C') || (s == 'Y') || (s == 'S') || (s == 'M');
  if (k == 2) return (p == 'G') || (p == 'R') || (p == 'S') || (p == 'K') || (s == 'G') || (s == 'R') || (s == 'S') || (s == 'K');
  if (k == 3) return (p == 'T') || (p == 'Y') || (p == 'W') || (p == 'K') || (s == 'T') || (s == 'Y') || (s == 'W') || (s == 'K');
  return 0;
}

/* createProfile(Trace, BaseCalls, p, trimleft, trimright), profile.h:21-52.
 * multi_array::resize zero-fills, so non-called bases start at 0 (profile.h:29, 42-44). */
int32_t orc_create_profile_trace(const int32_t* trace, size_t nsamples, const int32_t* bcpos,
                                 const char* primary, const char* secondary, size_t nbc,
                                 int32_t trimleft, int32_t trimright, float* p) {
  if (trimleft + trimright >= (int32_t)nbc) { trimleft = 0; trimright = 0; }
  int32_t sz = (int32_t)nbc - (trimleft + trimright);
  for (size_t i = 0; i < (size_t)6 * (size_t)sz; ++i) p[i] = 0;
  for (int32_t j = trimleft; j < trimleft + sz; ++j) {
    float totalsig = 0;
    float allBaseSig = 0;
    for (uint32_t k = 0; k < 4; ++k) {
      int32_t sig = trace[(size_t)k * nsamples + bcpos[j]];
      allBaseSig += sig;
      if (in_base_called(k, primary[j], secondary[j])) totalsig += sig;
    }
    size_t o = (size_t)(j - trimleft);
    p[(size_t)4 * sz + o] = 0;
    p[(size_t)5 * sz + o] = 0;
    if (totalsig == 0) {
      for (uint32_t k = 0; k < 4; ++k) p[(size_t)k * sz + o] = 0.25;
    } else {
      for (uint32_t k = 0; k < 4; ++k) {
        if (in_base_called(k, primary[j], secondary[j]))
          p[(size_t)k * sz + o] = ((float)(trace[(size_t)k * nsamples + bcpos[j]]) / totalsig);
      }
      float normfac = totalsig / allBaseSig;
      for (uint32_t k = 0; k < 4; ++k) {
        /* profile.h:48: float*float + (float)(1 - normfac) * 0.25(double) -> double -> float */
        p[(size_t)k * sz + o] = normfac * p[(size_t)k * sz + o] + (1 - normfac) * 0.25;
      }
    }
  }
  return sz;
}

/* reverseComplementProfile, profile.h:74-90 */
void orc_revcomp_profile(const float* p, size_t n, float* out) {
  int64_t pIdx = (int64_t)n - 1;
  size_t outIdx = 0;
  while (pIdx >= 0) {
    out[0 * n + outIdx] = p[3 * n + pIdx];
    out[1 * n + outIdx] = p[2 * n + pIdx];
    out[2 * n + outIdx] = p[1 * n + pIdx];
    out[3 * n + outIdx] = p[0 * n + pIdx];
    out[4 * n + outIdx] = p[4 * n + pIdx];
    out[5 * n + outIdx] = p[5 * n + pIdx];
    --pIdx;
    ++outIdx;
  }
}

/* trimReferenceSlice, fmindex.h:429-463 */
void orc_trim_reference_slice(const char* row0, const char* row1, size_t L, uint32_t trimLeft,
                              uint32_t trimRight, size_t refslice_size, int forward,
                              orc_trim_result* out) {
  uint32_t ri = 0;
  int32_t s = -1, e = -1;
  for (int64_t j = 0; j < (int64_t)L; ++j) {
    if (row0[j] != '-') {
      if (s == -1) s = (int32_t)j;
      e = (int32_t)j + 1;
    }
    if ((s == -1) && (row1[j] != '-')) ++ri;
  }
  uint32_t risize = 0;
  for (int64_t j = s; j < (int64_t)e; ++j) {
    if (row1[j] != '-') ++risize;
  }
  if (ri >= trimLeft) { ri -= trimLeft; risize += trimLeft; }
  if ((size_t)(uint32_t)(ri + risize + trimRight) < refslice_size) risize += trimRight; /* uint32 arithmetic as in the reference */
  int32_t oldlen = (int32_t)refslice_size;
  out->ri = ri;
  out->risize = risize;
  out->pos_add = 0;
  out->warn_negative_offset = 0;
  if (forward) out->pos_add = ri;
  else {
    int32_t offset = oldlen - (int32_t)ri - (int32_t)risize;
    if (offset < 0) out->warn_negative_offset = 1;
    else out->pos_add = (uint32_t)offset;
  }
}
