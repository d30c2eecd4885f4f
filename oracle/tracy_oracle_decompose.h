/*
 * tracy_oracle_decompose.h -- oracle declarations for decompose.h / abif.h helpers.
 * TEST INFRASTRUCTURE ONLY (see tracy_oracle.h for the parity status).
 */
#ifndef TRACY_ORACLE_DECOMPOSE_H
#define TRACY_ORACLE_DECOMPOSE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* TraceBreakpoint, fmindex.h:51-56 */
typedef struct {
  int32_t indelshift;
  int32_t traceleft;
  uint32_t breakpoint;
  float bestDiff;
} orc_breakpoint;

/* the IndigoConfig fields decomposeAlleles reads (indigo.h:16-40): trimLeft, trimRight, maxindel, madc */
typedef struct {
  int32_t trimLeft;
  int32_t trimRight;
  int32_t maxindel;
  int32_t madc;
} orc_decomp_cfg;

enum { ORC_DECOMP_SIMPLE = 0, ORC_DECOMP_COMPLEX = 1, ORC_DECOMP_NONE = 2 };
typedef struct {
  int32_t kind;    /* which stdout line decomposeAlleles printed: none / :315 / :327 */
  int32_t bestIns; /* only meaningful for ORC_DECOMP_COMPLEX / NONE */
  int32_t bestDel;
  int32_t bestFR;
} orc_decomp_status;

char orc_iupac2(char one, char two);
void orc_trimmed_seq(size_t size, uint32_t ltrim, uint32_t rtrim, size_t* off, size_t* len);
void orc_find_breakpoint(const float* ptrace, size_t ncol, orc_breakpoint* bp);
int orc_find_homozygous_breakpoint(const char* row0, const char* row1, size_t L, orc_breakpoint* bp);
/* primary/secondary (nbc chars) are rewritten in place; dcp_* need capacity 2*maxindel+2 */
int orc_decompose_alleles(const orc_decomp_cfg* c, const char* row0, const char* row1, size_t L,
                          char* primary, char* secondary, size_t nbc, orc_breakpoint bp,
                          size_t refslice_size, int32_t* dcp_indel, int32_t* dcp_err, size_t* dcp_n,
                          orc_decomp_status* st);
void orc_generate_secondary_decomposed(const int32_t* trace, size_t nsamples, const int32_t* bcpos,
                                       const char* primary, const char* secondary, size_t nbc,
                                       char* secdecomp);
void orc_allelic_fraction(const int32_t* trace, size_t nsamples, const int32_t* bcpos,
                          const char* primary_full, const char* secdecomp_full, size_t nbc,
                          uint32_t trimLeft, uint32_t trimRight, double* outI, double* outJ);

/* basecall(), abif.h:408-511 (without estimateQualities).  basecallpos has npos entries.  Output
 * arrays need capacity npos; returns the number of calls made (windows skipped by peak() :81 are
 * dropped). */
size_t orc_basecall(const int32_t* trace, size_t nsamples, const int32_t* basecallpos, size_t npos,
                    float sigratio, char* primary, char* secondary, char* consensus, int32_t* bcpos);

#ifdef __cplusplus
}
#endif
#endif
