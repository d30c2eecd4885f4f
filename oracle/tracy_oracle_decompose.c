/*
 * tracy_oracle_decompose.c -- CPU restatement of decompose.h (TEST INFRASTRUCTURE ONLY).
 * PARITY UNPINNED by reference execution (decompose.h needs Boost + fmindex.h types; see
 * tracy_oracle.h).  Follows /root/reference/src/decompose.h and abif.h:116-161 line by line,
 * including the unsigned wrap-arounds and float/double mixes.
 */
#include "tracy_oracle_decompose.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* iupac(TMountains), abif.h:116-133, for the 2-element case produced by iupac(char,char) */
static char iupac_pair(int32_t a, int32_t b) {
  if ((a == 0) && (b == 2)) return 'R';
  else if ((a == 1) && (b == 3)) return 'Y';
  else if ((a == 1) && (b == 2)) return 'S';
  else if ((a == 0) && (b == 3)) return 'W';
  else if ((a == 2) && (b == 3)) return 'K';
  else if ((a == 0) && (b == 1)) return 'M';
  return 'N';
}

/* iupac(char one, char two), abif.h:142-161: unknown letters keep index 0 ('A') */
char orc_iupac2(char one, char two) {
  int32_t p0 = 0, p1 = 0;
  if (one == 'A') p0 = 0; else if (one == 'C') p0 = 1; else if (one == 'G') p0 = 2; else if (one == 'T') p0 = 3;
  if (two == 'A') p1 = 0; else if (two == 'C') p1 = 1; else if (two == 'G') p1 = 2; else if (two == 'T') p1 = 3;
  if (p1 < p0) { int32_t t = p0; p0 = p1; p1 = t; }
  return iupac_pair(p0, p1);
}

/* isAmbiguous, abif.h:135-139 */
static int is_ambiguous(char n) { return !((n == 'A') || (n == 'C') || (n == 'G') || (n == 'T')); }

/* trimmedSeq, abif.h:68-75.  Returns the offset and length of the kept substring. */
void orc_trimmed_seq(size_t size, uint32_t ltrim, uint32_t rtrim, size_t* off, size_t* len) {
  if ((size_t)(uint32_t)(ltrim + rtrim + 1) >= size) { *off = 0; *len = size; }
  else { *off = ltrim; *len = (uint32_t)(size - ltrim - rtrim); }
}

/* findBreakpoint, decompose.h:7-56 */
void orc_find_breakpoint(const float* ptrace, size_t ncol, orc_breakpoint* bp) {
  double* sigratio = (double*)malloc(sizeof(double) * (ncol ? ncol : 1));
  for (uint32_t j = 0; j < ncol; ++j) {
    double best = 0.001;
    double sndBest = 0.001;
    for (uint32_t i = 0; i < 6; ++i) {
      float v = ptrace[(size_t)i * ncol + j];
      if (v > best) { sndBest = best; best = v; }
      else if (v > sndBest) { sndBest = v; }
    }
    sigratio[j] = best - sndBest;
  }
  bp->bestDiff = 0;
  bp->traceleft = 1;
  bp->breakpoint = 0;
  uint32_t minWindow = 25;
  if (minWindow < ncol) {
    for (uint32_t i = minWindow; i < ncol - minWindow; ++i) {
      double leftSum = 0;
      for (uint32_t k = i - minWindow; k < i; ++k) leftSum += sigratio[k];
      double left = leftSum / (double)minWindow;
      double rightSum = 0;
      for (uint32_t k = i; k < i + minWindow; ++k) rightSum += sigratio[k];
      double right = rightSum / (double)minWindow;
      double diff = fabs(right - left);
      if (diff > bp->bestDiff) { /* bestDiff is a float field (fmindex.h:55) */
        bp->breakpoint = i;
        bp->bestDiff = (float)diff;
        bp->traceleft = (left < right) ? 0 : 1;
      }
    }
  }
  bp->indelshift = 1;
  if (bp->bestDiff < 0.25) {
    bp->indelshift = 0;
    bp->breakpoint = (uint32_t)ncol;
    bp->traceleft = 1;
    bp->bestDiff = 0;
  }
  free(sigratio);
}

/* findHomozygousBreakpoint, decompose.h:59-128.  Returns 1 on success, 0 = "No valid alignment",
 * -1 = "Alignment too short" (both return false in the reference). */
int orc_find_homozygous_breakpoint(const char* row0, const char* row1, size_t L, orc_breakpoint* bp) {
  int64_t alignStart = 0, alignEnd = 0, varIndex = 0;
  for (int64_t j = 0; j < (int64_t)L; ++j) {
    if ((row0[j] != '-') && (row1[j] != '-')) { alignStart = j; break; }
    if (row0[j] != '-') ++varIndex;
  }
  for (int32_t j = (int32_t)(L - 1); j >= 0; --j) {
    if ((row0[j] != '-') && (row1[j] != '-')) { alignEnd = j; break; }
  }
  if (alignStart >= alignEnd) return 0;
  bp->bestDiff = 0;
  bp->traceleft = 1;
  bp->breakpoint = 0;
  uint32_t minWindow = 25;
  if (alignEnd < alignStart + (int64_t)(2 * minWindow)) return -1;
  for (uint32_t i = (uint32_t)alignStart; (int64_t)i < alignStart + (int64_t)minWindow; ++i) {
    if (row0[i] != '-') ++varIndex;
  }
  for (uint32_t i = (uint32_t)(alignStart + minWindow); (int64_t)i < alignEnd - (int64_t)minWindow; ++i) {
    if (row0[i] != '-') ++varIndex;
    double leftSum = 0;
    for (uint32_t k = i - minWindow; k < i; ++k) if (row0[k] != row1[k]) ++leftSum;
    double left = leftSum / (double)minWindow;
    double rightSum = 0;
    for (uint32_t k = i; k < i + minWindow; ++k) if (row0[k] != row1[k]) ++rightSum;
    double right = rightSum / (double)minWindow;
    double diff = fabs(right - left);
    if (diff > bp->bestDiff) {
      bp->breakpoint = (uint32_t)varIndex;
      bp->bestDiff = (float)diff;
      bp->traceleft = (left < right) ? 1 : 0;
    }
  }
  bp->indelshift = 1;
  if (bp->bestDiff < 0.25) {
    bp->indelshift = 0;
    bp->breakpoint = (uint32_t)varIndex;
    bp->traceleft = 1;
    bp->bestDiff = 0;
  }
  return 1;
}

/* getMedian, decompose.h:129-135: value at sorted position size/2 */
static int cmp_i32(const void* a, const void* b) {
  int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
  return (x > y) - (x < y);
}
static int32_t median_i32(const int32_t* v, size_t n) {
  int32_t* c = (int32_t*)malloc(sizeof(int32_t) * (n ? n : 1));
  memcpy(c, v, sizeof(int32_t) * n);
  qsort(c, n, sizeof(int32_t), cmp_i32);
  int32_t med = c[n / 2];
  free(c);
  return med;
}
/* getMAD, decompose.h:137-145 */
static int32_t mad_i32(const int32_t* v, size_t n, int32_t median) {
  int32_t* d = (int32_t*)malloc(sizeof(int32_t) * (n ? n : 1));
  for (size_t i = 0; i < n; ++i) d[i] = abs(v[i] - median);
  int32_t mad = median_i32(d, n);
  free(d);
  return mad;
}

/* phaseRefAllele, decompose.h:147-175 */
static char phase_ref_allele(const char* primary, const char* secondary, char r, uint32_t vi) {
  char s = secondary[vi], p = primary[vi];
  if ((r == '-') || (s == 'N')) return 'N';
  else if (s == r) return p;
  else {
    if (s == 'R') { if (r == 'A') return orc_iupac2(p, 'G'); else if (r == 'G') return orc_iupac2(p, 'A'); }
    else if (s == 'Y') { if (r == 'C') return orc_iupac2(p, 'T'); else if (r == 'T') return orc_iupac2(p, 'C'); }
    else if (s == 'S') { if (r == 'C') return orc_iupac2(p, 'G'); else if (r == 'G') return orc_iupac2(p, 'C'); }
    else if (s == 'W') { if (r == 'A') return orc_iupac2(p, 'T'); else if (r == 'T') return orc_iupac2(p, 'A'); }
    else if (s == 'K') { if (r == 'G') return orc_iupac2(p, 'T'); else if (r == 'T') return orc_iupac2(p, 'G'); }
    else if (s == 'M') { if (r == 'A') return orc_iupac2(p, 'C'); else if (r == 'C') return orc_iupac2(p, 'A'); }
    else return 'N';
  }
  return 'N';
}

/* the shift-scan inner loop shared by decompose.h:214-224, 251-261, 293-313 */
static int32_t count_failed(const char* row1, size_t L, const char* primary, const char* secondary,
                            size_t vend, uint32_t jstart, uint32_t vi) {
  int32_t failedref = 0;
  for (uint32_t j = jstart; ((j < L) && (vi < vend)); ++j, ++vi) {
    if (row1[j] != primary[vi]) {
      if (phase_ref_allele(primary, secondary, row1[j], vi) == 'N') ++failedref;
    }
  }
  return failedref;
}
/* the rewrite loop shared by decompose.h:317-326, 351-359, 363-371 */
static void apply_phase(const char* row1, size_t L, char* primary, char* secondary, size_t vend,
                        uint32_t jstart, uint32_t vi) {
  for (uint32_t j = jstart; ((j < L) && (vi < vend)); ++j, ++vi) {
    if (row1[j] != primary[vi]) {
      char sec = phase_ref_allele(primary, secondary, row1[j], vi);
      if (sec != 'N') { primary[vi] = row1[j]; secondary[vi] = sec; }
    }
  }
}

/* decomposeAlleles, decompose.h:179-376 */
int orc_decompose_alleles(const orc_decomp_cfg* c, const char* row0, const char* row1, size_t L,
                          char* primary, char* secondary, size_t nbc, orc_breakpoint bp,
                          size_t refslice_size, int32_t* dcp_indel, int32_t* dcp_err, size_t* dcp_n,
                          orc_decomp_status* st) {
  int32_t ltrim = c->trimLeft;
  int32_t rtrim = c->trimRight;
  uint32_t varIndex = 0, refPointer = 0, alignIndex = 0;
  uint32_t vi = (uint32_t)ltrim;
  bp.breakpoint += (uint32_t)ltrim;
  for (uint32_t j = 0; j < L; ++j) {
    if (row0[j] != '-') {
      if (row1[j] != primary[vi]) {
        char sec = phase_ref_allele(primary, secondary, row1[j], vi);
        if (sec != 'N') { primary[vi] = row1[j]; secondary[vi] = sec; }
      }
      ++vi;
      if (vi == bp.breakpoint) { alignIndex = j; varIndex = vi; break; }
    }
    if (row1[j] != '-') ++refPointer;
  }
  const size_t vend = nbc - (size_t)rtrim; /* bc.consensus.size() - rtrim, size_t arithmetic */

  /* deletion scan, decompose.h:210-225 */
  uint32_t maxdel = 2;
  if (refslice_size > (size_t)(uint32_t)(refPointer + (uint32_t)rtrim + 2))
    maxdel = (uint32_t)(refslice_size - (size_t)(uint32_t)(refPointer + (uint32_t)rtrim));
  size_t cap = (size_t)c->maxindel + 1;
  int32_t* fref = (int32_t*)malloc(sizeof(int32_t) * cap);
  size_t nfref = 0;
  for (uint32_t del = 0; ((del < (uint32_t)c->maxindel) && (del < maxdel / 2)); ++del)
    fref[nfref++] = count_failed(row1, L, primary, secondary, vend, alignIndex + del + 1, varIndex);

  /* cutoffs, decompose.h:227-234 */
  int32_t med = median_i32(fref, nfref);
  int32_t mad = mad_i32(fref, nfref, med);
  int32_t thres = 0;
  if (med > c->madc * mad) thres = med - c->madc * mad;
  if (thres < 10) thres = 10;

  /* deletion picks, decompose.h:237-245 */
  int32_t* deldecomp = (int32_t*)malloc(sizeof(int32_t) * cap);
  size_t ndel = 0;
  for (uint32_t i = 0; i < nfref; ++i) {
    if (fref[i] < thres) {
      if ((i + 1 < nfref) && (2 * fref[i] < fref[i + 1])) deldecomp[ndel++] = (int32_t)i;
      else if ((i > 0) && (2 * fref[i] < fref[i - 1])) deldecomp[ndel++] = (int32_t)i;
      else if ((i == 0) && (i + 2 < nfref) && (2 * fref[i] < fref[i + 2])) deldecomp[ndel++] = (int32_t)i;
    }
  }

  /* insertion scan, decompose.h:247-262 */
  int32_t* fins = (int32_t*)malloc(sizeof(int32_t) * cap);
  size_t nfins = 0;
  fins[nfins++] = fref[0];
  uint32_t maxins = (uint32_t)((int32_t)nbc - (int32_t)((uint32_t)rtrim + bp.breakpoint));
  for (uint32_t ins = 1; ((ins < (uint32_t)c->maxindel) && (ins < maxins / 2)); ++ins)
    fins[nfins++] = count_failed(row1, L, primary, secondary, vend, alignIndex + 1, varIndex + ins);

  /* insertion picks, decompose.h:264-271 */
  int32_t* insdecomp = (int32_t*)malloc(sizeof(int32_t) * cap);
  size_t nins = 0;
  for (uint32_t i = 0; i < nfins; ++i) {
    if (fins[i] < thres) {
      if ((i + 1 < nfins) && (2 * fins[i] < fins[i + 1])) insdecomp[nins++] = (int32_t)i;
      else if ((i > 0) && (2 * fins[i] < fins[i - 1])) insdecomp[nins++] = (int32_t)i;
      else if ((i == 0) && (i + 2 < nfins) && (2 * fins[i] < fins[i + 2])) insdecomp[nins++] = (int32_t)i;
    }
  }

  /* decomposition table, decompose.h:273-285 */
  int32_t defins = 15;
  if ((ndel == 0) && (nins == 0)) defins = 50;
  for (uint32_t i = 0; i < nins; ++i) if (insdecomp[i] + 15 > defins) defins = insdecomp[i] + 15;
  if (defins > (int32_t)nfins) defins = (int32_t)nfins;
  int32_t defdel = 15;
  if ((ndel == 0) && (nins == 0)) defdel = 50;
  for (uint32_t i = 0; i < ndel; ++i) if (deldecomp[i] + 15 > defdel) defdel = deldecomp[i] + 15;
  if (defdel > (int32_t)nfref) defdel = (int32_t)nfref;
  size_t nd = 0;
  for (int32_t i = defdel - 1; i >= 0; --i) { dcp_indel[nd] = -1 * i; dcp_err[nd] = fref[i]; ++nd; }
  for (int32_t i = 1; i < defins; ++i) { dcp_indel[nd] = i; dcp_err[nd] = fins[i]; ++nd; }
  *dcp_n = nd;

  /* actual decomposition, decompose.h:287-374 */
  st->kind = ORC_DECOMP_SIMPLE;
  st->bestIns = 0; st->bestDel = 0; st->bestFR = 1000;
  if ((ndel == 0) && (nins == 0)) {
    int32_t bestIns = 0, bestDel = 0, bestFR = 1000;
    for (uint32_t ins = 0; ((ins < (uint32_t)c->maxindel) && (ins < maxins / 2)); ++ins) {
      int32_t prevFailedRef = 0;
      for (uint32_t del = 0; ((del < (uint32_t)c->maxindel) && (del < maxdel / 2)); ++del) {
        int32_t failedref = count_failed(row1, L, primary, secondary, vend, alignIndex + del + 1, varIndex + ins);
        if (2 * failedref < prevFailedRef) {
          if (failedref < bestFR) { bestIns = (int32_t)ins; bestDel = (int32_t)del; bestFR = failedref; }
        }
        prevFailedRef = failedref;
      }
    }
    st->bestIns = bestIns; st->bestDel = bestDel; st->bestFR = bestFR;
    if (bestFR != 1000) {
      st->kind = ORC_DECOMP_COMPLEX; /* "Complex mutation, decomposition: ins: .. del: .. error: .." :315 */
      apply_phase(row1, L, primary, secondary, vend, alignIndex + (uint32_t)bestDel + 1, varIndex + (uint32_t)bestIns);
    } else {
      st->kind = ORC_DECOMP_NONE; /* "No InDel detected, traverse the whole alignment." :327 */
      vi = (uint32_t)ltrim;
      for (uint32_t j = 0; j < L; ++j) {
        if (row0[j] != '-') {
          if (row1[j] != primary[vi]) {
            char sec = phase_ref_allele(primary, secondary, row1[j], vi);
            if (sec != 'N') { primary[vi] = row1[j]; secondary[vi] = sec; }
          }
          ++vi;
        }
      }
    }
  } else {
    if (ndel != 0) { /* deldecomp is already ascending (pushed in index order); std::sort is a no-op */
      apply_phase(row1, L, primary, secondary, vend, alignIndex + (uint32_t)deldecomp[0] + 1, varIndex);
    } else {
      apply_phase(row1, L, primary, secondary, vend, alignIndex + 1, varIndex + (uint32_t)insdecomp[0]);
    }
  }
  free(fref); free(fins); free(deldecomp); free(insdecomp);
  return 1;
}

/* generateSecondaryDecomposed, decompose.h:378-410 */
void orc_generate_secondary_decomposed(const int32_t* trace, size_t nsamples, const int32_t* bcpos,
                                       const char* primary, const char* secondary, size_t nbc,
                                       char* secdecomp) {
  const int32_t* A = trace;
  const int32_t* C = trace + nsamples;
  const int32_t* G = trace + 2 * nsamples;
  const int32_t* T = trace + 3 * nsamples;
  for (uint32_t i = 0; i < nbc; ++i) {
    if (primary[i] == secondary[i]) secdecomp[i] = primary[i];
    else if (!is_ambiguous(secondary[i])) secdecomp[i] = secondary[i];
    else {
      uint32_t tp = (uint32_t)bcpos[i];
      char s = secondary[i];
      if (s == 'R') secdecomp[i] = (A[tp] > G[tp]) ? 'A' : 'G';
      else if (s == 'Y') secdecomp[i] = (C[tp] > T[tp]) ? 'C' : 'T';
      else if (s == 'S') secdecomp[i] = (C[tp] > G[tp]) ? 'C' : 'G';
      else if (s == 'W') secdecomp[i] = (A[tp] > T[tp]) ? 'A' : 'T';
      else if (s == 'K') secdecomp[i] = (G[tp] > T[tp]) ? 'G' : 'T';
      else if (s == 'M') secdecomp[i] = (A[tp] > C[tp]) ? 'A' : 'C';
      else secdecomp[i] = 'N';
    }
  }
}

/* allelicFraction, decompose.h:412-621.  primary/secdecomp are the untrimmed strings. */
void orc_allelic_fraction(const int32_t* trace, size_t nsamples, const int32_t* bcpos,
                          const char* primary_full, const char* secdecomp_full, size_t nbc,
                          uint32_t trimLeft, uint32_t trimRight, double* outI, double* outJ) {
  size_t off, len;
  orc_trimmed_seq(nbc, trimLeft, trimRight, &off, &len);
  const char* pri = primary_full + off;
  const char* sec = secdecomp_full + off;
  uint32_t diffnuc = 0;
  for (uint32_t i = 0; i < len; ++i) if (pri[i] != sec[i]) ++diffnuc;
  double bestI = 0.5, bestJ = 0.5;
  if (diffnuc) {
    double bestSSE = 0, bestK = 0, bestL = 0;
    size_t d = diffnuc;
    double* tp = (double*)calloc(4 * d, sizeof(double));
    double* prip = (double*)calloc(4 * d, sizeof(double));
    double* secp = (double*)calloc(4 * d, sizeof(double));
    double* terp = (double*)calloc(4 * d, sizeof(double));
    double* quap = (double*)calloc(4 * d, sizeof(double));
    uint32_t nucpos = 0;
    for (uint32_t i = 0; i < len; ++i) {
      if (pri[i] != sec[i]) {
        uint32_t tpos = (uint32_t)bcpos[i + trimLeft];
        int32_t sg[4];
        for (int k = 0; k < 4; ++k) sg[k] = trace[(size_t)k * nsamples + tpos];
        double sigsum = sg[0] + sg[1] + sg[2] + sg[3]; /* int sum, then double */
        for (int k = 0; k < 4; ++k) tp[(size_t)k * d + nucpos] = (double)(sg[k]) / sigsum;
        /* the 12 ordered (pri,sec) cases, decompose.h:448-569: the two remaining channels a<b in
         * ACGT order; tertiary = a if sig[a] > sig[b] (strict) else b */
        int pi = -1, si = -1;
        if (pri[i] == 'A') pi = 0; else if (pri[i] == 'C') pi = 1; else if (pri[i] == 'G') pi = 2; else if (pri[i] == 'T') pi = 3;
        if (sec[i] == 'A') si = 0; else if (sec[i] == 'C') si = 1; else if (sec[i] == 'G') si = 2; else if (sec[i] == 'T') si = 3;
        if ((pi >= 0) && (si >= 0) && (pi != si)) {
          int rest[2], nr = 0;
          for (int k = 0; k < 4; ++k) if ((k != pi) && (k != si)) rest[nr++] = k;
          prip[(size_t)pi * d + nucpos] = 1;
          secp[(size_t)si * d + nucpos] = 1;
          if (sg[rest[0]] > sg[rest[1]]) { terp[(size_t)rest[0] * d + nucpos] = 1; quap[(size_t)rest[1] * d + nucpos] = 1; }
          else { terp[(size_t)rest[1] * d + nucpos] = 1; quap[(size_t)rest[0] * d + nucpos] = 1; }
        }
        ++nucpos;
      }
    }
    for (uint32_t m = 0; m < 4; ++m) {
      for (uint32_t n = 0; n < diffnuc; ++n) {
        double pred = bestI * prip[m * d + n] + bestJ * secp[m * d + n] + bestK * terp[m * d + n] + bestL * quap[m * d + n];
        bestSSE += (pred - tp[m * d + n]) * (pred - tp[m * d + n]);
      }
    }
    for (double i = 0; i <= 1; i += 0.01) {
      for (double j = 0; j <= 1; j += 0.01) {
        if (i + j <= 1) {
          for (double k = 0; k <= 1; k += 0.01) {
            if (i + j + k <= 1) {
              double l = 1 - (i + j + k);
              double sse = 0;
              for (uint32_t m = 0; m < 4; ++m) {
                for (uint32_t n = 0; n < diffnuc; ++n) {
                  double pred = i * prip[m * d + n] + j * secp[m * d + n] + k * terp[m * d + n] + l * quap[m * d + n];
                  sse += (pred - tp[m * d + n]) * (pred - tp[m * d + n]);
                  if (sse >= bestSSE) break;
                }
              }
              if (sse < bestSSE) { bestSSE = sse; bestL = l; bestK = k; bestJ = j; bestI = i; }
            }
          }
        }
      }
    }
    (void)bestK; (void)bestL;
    free(tp); free(prip); free(secp); free(terp); free(quap);
  }
  *outI = bestI;
  *outJ = bestJ;
}
