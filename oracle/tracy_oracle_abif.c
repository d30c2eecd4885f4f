/*
 * tracy_oracle_abif.c -- CPU restatement of basecall() (abif.h:408-511) and peak() (abif.h:77-97).
 * TEST INFRASTRUCTURE ONLY.  PINNED: tests/test_oracle_abif_ref.py compares this restatement with the
 * reference's own abif.h compiled into oracle/_ref/libref_abif.so (see oracle/Makefile, ref_abif.cpp).
 * estimateQualities() (abif.h:232-253) is not restated: nothing on the alignment/decomposition path
 * reads estQual.
 */
#include "tracy_oracle_decompose.h"

#include <math.h>
#include <stdlib.h>

/* peak(), abif.h:77-97 */
static int peak4(const int32_t* trace, size_t nsamples, float s, float e, int32_t* pVal, int32_t* pIdx) {
  if ((int32_t)(floorf(s)) == (int32_t)(floorf(e))) return 0;
  for (uint32_t k = 0; k < 4; ++k) {
    const int32_t* t = trace + (size_t)k * nsamples;
    int32_t bestIdx = (int32_t)(floorf(s));
    int32_t bestVal = 0;
    int32_t lo = (int32_t)floorf(s);
    if (lo < 1) lo = 1;
    int32_t hi = (int32_t)floorf(e);
    if ((int32_t)(nsamples - 1) < hi) hi = (int32_t)(nsamples - 1);
    for (int32_t i = lo; i < hi; ++i) {
      if (((t[i - 1] <= t[i]) && (t[i] > t[i + 1])) || ((t[i - 1] < t[i]) && (t[i] >= t[i + 1]))) {
        if (t[i] > bestVal) { bestIdx = i; bestVal = t[i]; }
      }
    }
    pVal[k] = bestVal;
    pIdx[k] = bestIdx;
  }
  return 1;
}

static char base_letter(int32_t k) { return k == 0 ? 'A' : k == 1 ? 'C' : k == 2 ? 'G' : 'T'; }

size_t orc_basecall(const int32_t* trace, size_t nsamples, const int32_t* basecallpos, size_t npos,
                    float sigratio, char* primary, char* secondary, char* consensus, int32_t* bcpos) {
  if (npos == 0) return 0;
  float* st = (float*)malloc(sizeof(float) * npos);
  float* ed = (float*)malloc(sizeof(float) * npos);
  int32_t oldVal = 0, lastDiff = 0;
  size_t ned = 0;
  for (uint32_t i = 0; i < npos; ++i) { /* abif.h:417-423 */
    lastDiff = basecallpos[i] - oldVal;
    st[i] = (float)((float)basecallpos[i] - 0.5 * (float)lastDiff);
    if (i > 0) ed[ned++] = (float)((float)basecallpos[i - 1] + 0.5 * (float)lastDiff);
    oldVal = basecallpos[i];
  }
  ed[ned++] = (float)(basecallpos[npos - 1] + 0.5 * lastDiff);

  size_t nc = 0;
  for (uint32_t i = 0; i < npos; ++i) {
    int32_t pVal[4], pIdx[4];
    if (!peak4(trace, nsamples, st[i], ed[i], pVal, pIdx)) continue;
    int32_t midpoint = (int32_t)((st[i] + ed[i]) / 2.0);
    if (midpoint >= floorf(ed[i])) midpoint = (int32_t)floorf(st[i]);
    int32_t estVal = 1;
    for (uint32_t k = 0; k < 4; ++k)
      if (trace[(size_t)k * nsamples + midpoint] > estVal) estVal = trace[(size_t)k * nsamples + midpoint];
    int32_t threshold = (int32_t)(sigratio * estVal);
    if ((pVal[0] <= threshold) && (pVal[1] <= threshold) && (pVal[2] <= threshold) && (pVal[3] <= threshold)) {
      for (uint32_t k = 0; k < 4; ++k) { pIdx[k] = midpoint; pVal[k] = trace[(size_t)k * nsamples + midpoint]; }
    }
    int32_t maxVal = 1;
    for (uint32_t k = 0; k < 4; ++k) if (pVal[k] > maxVal) maxVal = pVal[k];
    float srat[4];
    for (uint32_t k = 0; k < 4; ++k) srat[k] = (float)pVal[k] / (float)maxVal;
    float bestRat = sigratio;
    int32_t selACGT = -1;
    int32_t selPos = pIdx[0];
    int32_t validBases = 0;
    for (uint32_t k = 0; k < 4; ++k) {
      if (srat[k] >= sigratio) {
        ++validBases;
        if (srat[k] >= bestRat) { bestRat = srat[k]; selPos = pIdx[k]; selACGT = (int32_t)k; }
      }
    }
    bcpos[nc] = selPos;
    if ((validBases == 4) || (selACGT == -1)) {
      primary[nc] = 'N'; secondary[nc] = 'N'; consensus[nc] = 'N';
    } else if (validBases > 1) {
      primary[nc] = base_letter(selACGT);
      int32_t left[4] = {0, 0, 0, 0}, nl = 0;
      for (int32_t k = 0; k < 4; ++k) if ((k != selACGT) && (srat[k] >= sigratio)) left[nl++] = k;
      if (nl == 1) secondary[nc] = base_letter(left[0]);       /* iupac(TMountains) size 1, abif.h:119-123 */
      else secondary[nc] = orc_iupac2(base_letter(left[0]), base_letter(left[1])); /* size 2, :124-131 */
      consensus[nc] = 'N';
    } else {
      primary[nc] = secondary[nc] = consensus[nc] = base_letter(selACGT);
    }
    ++nc;
  }
  free(st); free(ed);
  return nc;
}
