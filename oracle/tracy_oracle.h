/*
 * tracy_oracle.h -- CPU restatement of the tracy hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle: a plain-C restatement of the reference algorithms in
 * /root/reference/src/{align,gotoh,needle,profile,decompose,abif,fmindex}.h.  It is NOT part of the
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * PARITY STATUS
 *   - basecall / iupac / trimmedSeq (abif.h): PINNED against the reference's own code compiled from
 *     /root/reference/src/abif.h (std-only header) -> oracle/_ref/ref_abif (see oracle/Makefile).
 *   - gotoh / gotohScore / needle / profile / decompose / trimReferenceSlice: PARITY UNPINNED by
 *     reference execution.  align.h / gotoh.h / profile.h / decompose.h need Boost (multi_array,
 *     dynamic_bitset) and fmindex.h needs htslib + sdsl-lite; none is installed in this image and the
 *     reference ships no tests, golden vectors or fixtures.  The restatement follows the source text
 *     line by line (citations on every function) and is cross-checked against an independent
 *     full-matrix numpy formulation (tests/test_oracle_*.py) and hand-derived known answers.
 */
#ifndef TRACY_ORACLE_H
#define TRACY_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* DnaScore<int>, align.h:11-32.  inf is fixed at 1000000 (align.h:26,30). */
typedef struct {
  int32_t match;
  int32_t mismatch;
  int32_t go;
  int32_t ge;
} orc_score;

#define ORC_INF 1000000

/* ---- Gotoh, string x string (gotoh.h:12-68, 71-174; _score align.h:96-101) ------------------ */
int32_t orc_gotoh_score_str(const char* s1, size_t m, const char* s2, size_t n,
                            int hfree, int vfree, const orc_score* sc);
/* btr receives the trace in push order (i.e. back-to-front, gotoh.h:143-167); capacity >= m+n. */
int32_t orc_gotoh_str(const char* s1, size_t m, const char* s2, size_t n,
                      int hfree, int vfree, const orc_score* sc, char* btr, size_t* btr_len);

/* ---- Gotoh, float profile x float profile (p[k][j] at k*len + j, 6 rows) (align.h:103-118) --- */
int32_t orc_gotoh_score_prof(const float* p1, size_t m, const float* p2, size_t n,
                             int hfree, int vfree, const orc_score* sc);
int32_t orc_gotoh_prof(const float* p1, size_t m, const float* p2, size_t n,
                       int hfree, int vfree, const orc_score* sc, char* btr, size_t* btr_len);

/* ---- Needleman-Wunsch, linear gaps (needle.h:12-57, 59-138).  Profiles are float inputs copied
 *      into double profiles (needle.h:26, align.h:183-194). ------------------------------------- */
int32_t orc_needle_score_str(const char* s1, size_t m, const char* s2, size_t n,
                             int hfree, int vfree, const orc_score* sc);
int32_t orc_needle_str(const char* s1, size_t m, const char* s2, size_t n,
                       int hfree, int vfree, const orc_score* sc, char* btr, size_t* btr_len);
int32_t orc_needle_score_prof(const float* p1, size_t m, const float* p2, size_t n,
                              int hfree, int vfree, const orc_score* sc);
int32_t orc_needle_prof(const float* p1, size_t m, const float* p2, size_t n,
                        int hfree, int vfree, const orc_score* sc, char* btr, size_t* btr_len);

/* ---- alignment materialisation (align.h:196-223, 254-293).  row0/row1 capacity >= btr_len. ---- */
void orc_create_alignment_str(const char* btr, size_t btr_len, const char* s1, const char* s2,
                              char* row0, char* row1);
void orc_create_alignment_prof(const char* btr, size_t btr_len, const float* p1, size_t m,
                               const float* p2, size_t n, char* row0, char* row1);
char orc_profile_cons_char(const float* p, size_t len, size_t pos); /* align.h:254-270 */

/* ---- profiles -------------------------------------------------------------------------------- */
/* _createProfile(std::string), align.h:121-136: one-hot float[6][n]. */
void orc_create_profile_str(const char* s, size_t n, float* p);
/* createProfile(Trace, BaseCalls, p, trimleft, trimright), profile.h:21-52.
 * trace: 4 channels (A,C,G,T) of nsamples int32 each, channel k at trace + k*nsamples.
 * Returns the number of profile columns written (sz); p must hold 6*nbc floats. */
int32_t orc_create_profile_trace(const int32_t* trace, size_t nsamples, const int32_t* bcpos,
                                 const char* primary, const char* secondary, size_t nbc,
                                 int32_t trimleft, int32_t trimright, float* p);
/* reverseComplementProfile, profile.h:74-90 */
void orc_revcomp_profile(const float* p, size_t n, float* out);

/* ---- trimReferenceSlice, fmindex.h:429-463.  Works on the 2-row alignment; returns the new
 *      offset/size and the position update.  ------------------------------------------------------ */
typedef struct {
  uint32_t ri;       /* offset into the old refslice */
  uint32_t risize;   /* size of the new refslice (before substr clamping) */
  uint32_t pos_add;  /* amount added to rs.pos */
  int32_t warn_negative_offset;
} orc_trim_result;
void orc_trim_reference_slice(const char* row0, const char* row1, size_t L, uint32_t trimLeft,
                              uint32_t trimRight, size_t refslice_size, int forward,
                              orc_trim_result* out);

#ifdef __cplusplus
}
#endif
#endif
