/* tracy_oracle_chain.c -- the `tracy align` hot section (sage.h:233-260, 311) composed from the oracle functions,
 * one trace per call, plus a pthread driver that runs one trace per thread.  TEST INFRASTRUCTURE ONLY: this is the
 * CPU baseline leg of bench.py (kind "port") and the expected values of the parity check; the product never calls it.
 * Same structure as the reference: scalar row-major loops, one full alignment after the other, no SIMD. */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "tracy_oracle.h"

typedef struct {
  int32_t score_fwd, score_rev, forward, score_prelim;
  uint32_t slice_begin, slice_len, ref_pos;
  int32_t score_final;
  uint32_t btr_len;
  uint64_t cells;  /* 3 * mt * n + mf * slice_len */
} orc_chain_result;

static void revcomp_str(const char* s, size_t n, char* out) { /* reverseComplement(std::string&) for [ACGTN], fmindex.h:11-25 */
  for (size_t i = 0; i < n; ++i) {
    const char c = s[n - 1 - i];
    out[i] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c == 'N' ? 'N' : s[i];
  }
}

/* profile_full: float[6][mf]; ref: n chars [ACGTN]; btr (capacity mf + n) receives the final alignment's ops */
int orc_sage_chain(const float* profile_full, size_t mf, const char* ref, size_t n, const orc_score* sc, uint32_t trim_left,
                   uint32_t trim_right, orc_chain_result* out, char* btr) {
  uint32_t tl = trim_left, tr = trim_right;
  if ((size_t)tl + tr >= mf) { tl = 0; tr = 0; }  /* createProfile, profile.h:24-27 */
  const size_t mt = mf - tl - tr;
  float* trimmed = (float*)malloc(sizeof(float) * 6 * (mt ? mt : 1));
  float* fwdp = (float*)malloc(sizeof(float) * 6 * (n ? n : 1));
  float* revp = (float*)malloc(sizeof(float) * 6 * (n ? n : 1));
  char* refslice = (char*)malloc(n + 1);
  char* btr1 = (char*)malloc(mt + n + 1);
  char* row0 = (char*)malloc(mt + n + 1);
  char* row1 = (char*)malloc(mt + n + 1);
  if (!trimmed || !fwdp || !revp || !refslice || !btr1 || !row0 || !row1) return -1;
  for (int k = 0; k < 6; ++k) memcpy(trimmed + (size_t)k * mt, profile_full + (size_t)k * mf + tl, sizeof(float) * mt);
  orc_create_profile_str(ref, n, fwdp);
  orc_revcomp_profile(fwdp, n, revp);
  out->score_fwd = orc_gotoh_score_prof(trimmed, mt, fwdp, n, 1, 0, sc);
  out->score_rev = orc_gotoh_score_prof(trimmed, mt, revp, n, 1, 0, sc);
  out->forward = out->score_fwd > out->score_rev;  /* sage.h:247 */
  const float* pref = out->forward ? fwdp : revp;
  if (out->forward) memcpy(refslice, ref, n);
  else revcomp_str(ref, n, refslice);
  size_t l1 = 0;
  out->score_prelim = orc_gotoh_prof(trimmed, mt, pref, n, 1, 0, sc, btr1, &l1);
  orc_create_alignment_prof(btr1, l1, trimmed, mt, pref, n, row0, row1);
  orc_trim_result tres;
  orc_trim_reference_slice(row0, row1, l1, trim_left, trim_right, n, out->forward, &tres);
  size_t len = tres.risize;
  if (tres.ri > n) len = 0;
  else if (tres.ri + len > n) len = n - tres.ri;  /* std::string::substr clamps */
  out->slice_begin = tres.ri;
  out->slice_len = (uint32_t)len;
  out->ref_pos = tres.pos_add;
  float* slicep = (float*)malloc(sizeof(float) * 6 * (len ? len : 1));
  if (!slicep) return -1;
  orc_create_profile_str(refslice + tres.ri, len, slicep);
  size_t l2 = 0;
  out->score_final = orc_gotoh_prof(profile_full, mf, slicep, len, 1, 0, sc, btr, &l2);
  out->btr_len = (uint32_t)l2;
  out->cells = 3ull * mt * n + (uint64_t)mf * len;
  free(slicep); free(trimmed); free(fwdp); free(revp); free(refslice); free(btr1); free(row0); free(row1);
  return 0;
}

typedef struct {
  const float* profiles; size_t mf; const char* refs; size_t n; const orc_score* sc; uint32_t tl, tr;
  orc_chain_result* out; char* btr; size_t btr_cap; uint32_t ntraces, nthreads, tid;
} chain_job;

static void* chain_worker(void* arg) {
  chain_job* j = (chain_job*)arg;
  for (uint32_t t = j->tid; t < j->ntraces; t += j->nthreads)
    orc_sage_chain(j->profiles + (size_t)t * 6 * j->mf, j->mf, j->refs + (size_t)t * j->n, j->n, j->sc, j->tl, j->tr, &j->out[t],
                   j->btr + (size_t)t * j->btr_cap);
  return NULL;
}

/* ntraces traces of equal shape (profiles [nt][6][mf], refs [nt][n]), trace t on thread t % nthreads; btr [nt][btr_cap] */
int orc_sage_chain_batch(const float* profiles, size_t mf, const char* refs, size_t n, uint32_t ntraces, const orc_score* sc,
                         uint32_t trim_left, uint32_t trim_right, uint32_t nthreads, orc_chain_result* out, char* btr, size_t btr_cap) {
  if (nthreads == 0) nthreads = 1;
  if (nthreads > ntraces) nthreads = ntraces ? ntraces : 1;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * nthreads);
  chain_job* jobs = (chain_job*)malloc(sizeof(chain_job) * nthreads);
  if (!th || !jobs) return -1;
  for (uint32_t i = 0; i < nthreads; ++i) {
    jobs[i] = (chain_job){profiles, mf, refs, n, sc, trim_left, trim_right, out, btr, btr_cap, ntraces, nthreads, i};
    pthread_create(&th[i], NULL, chain_worker, &jobs[i]);
  }
  for (uint32_t i = 0; i < nthreads; ++i) pthread_join(th[i], NULL);
  free(th);
  free(jobs);
  return 0;
}

/* ---- row state of the semiglobal Gotoh DP (test oracle of the prefix-bound kernel) --------------------------------
 * H(R, j) and F(R, j), j = 0..n, of gotoh.h:39-66 run on the first R rows of p1 with AlignConfig<true,false> where row R
 * is NOT the last row (its horizontal moves cost go/ge).  Any alignment of all m rows passes row R in state H or F,
 * so max_j max(H, F)(R, j) + (an upper bound for the remaining rows) bounds the final score from above. */
static int32_t prof_score_f(const float* p1, size_t m, size_t row, const float* p2, size_t n, size_t col, const orc_score* sc) {
  float acc = 0.0f; /* align.h:103-118 */
  for (int k1 = 0; k1 < 5; ++k1)
    for (int k2 = 0; k2 < 5; ++k2)
      acc = acc + (p1[(size_t)k1 * m + row] * p2[(size_t)k2 * n + col]) * (float)(k1 == k2 ? sc->match : sc->mismatch);
  return (int32_t)acc;
}

void orc_gotoh_row_state(const float* p1, size_t m, const float* p2, size_t n, size_t R, const orc_score* sc, int32_t* H_out,
                         int32_t* F_out) {
  int32_t* s = (int32_t*)malloc(sizeof(int32_t) * (n + 1));
  int32_t* v = (int32_t*)malloc(sizeof(int32_t) * (n + 1));
  const int32_t inf = ORC_INF;
  for (size_t c = 0; c <= n; ++c) { s[c] = 0; v[c] = -inf; } /* row 0: free horizontal end gap */
  for (size_t r = 1; r <= R; ++r) {
    int32_t diag = s[0];
    s[0] = sc->go + (int32_t)r * sc->ge;
    v[0] = -inf; /* column 0 carries no vertical state of its own besides H */
    int32_t e = -inf;
    for (size_t c = 1; c <= n; ++c) {
      const int32_t eo = s[c - 1] + sc->go + sc->ge, ee = e + sc->ge;
      e = eo > ee ? eo : ee;
      const int32_t fo = s[c] + sc->go + sc->ge, fe = v[c] + sc->ge;
      const int32_t f = fo > fe ? fo : fe;
      int32_t h = diag + prof_score_f(p1, m, r - 1, p2, n, c - 1, sc);
      if (e > h) h = e;
      if (f > h) h = f;
      diag = s[c];
      s[c] = h;
      v[c] = f;
    }
  }
  for (size_t c = 0; c <= n; ++c) { H_out[c] = s[c]; F_out[c] = v[c]; }
  free(s);
  free(v);
}
