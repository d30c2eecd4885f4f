// ref_abif.cpp -- thin extern "C" driver over the REFERENCE's own abif.h, compiled where it lies
// (-I/root/reference/src, never copied).  abif.h is the only header on the hot path whose includes
// are all present in this image (std only); everything DP/decompose needs Boost/htslib/sdsl and is
// unbuildable here (no stand-ins are written).  Output: oracle/_ref/libref_abif.so (git-ignored).
// TEST INFRASTRUCTURE ONLY: used by tests/ to pin oracle/tracy_oracle_abif.c and orc_iupac2/trimmed_seq.
#include <cstring>
#include "abif.h"

extern "C" {

size_t ref_basecall(const int32_t* trace, size_t nsamples, const int32_t* basecallpos, size_t npos,
                    float sigratio, char* primary, char* secondary, char* consensus, int32_t* bcpos) {
  tracy::Trace tr;
  tr.traceACGT.resize(4);
  for (int k = 0; k < 4; ++k) tr.traceACGT[k].assign(trace + k * nsamples, trace + (k + 1) * nsamples);
  tr.basecallpos.assign(basecallpos, basecallpos + npos);
  tracy::BaseCalls bc;
  tracy::basecall(tr, bc, sigratio);
  size_t n = bc.primary.size();
  std::memcpy(primary, bc.primary.data(), n);
  std::memcpy(secondary, bc.secondary.data(), n);
  std::memcpy(consensus, bc.consensus.data(), n);
  for (size_t i = 0; i < n; ++i) bcpos[i] = bc.bcPos[i];
  return n;
}

char ref_iupac2(char one, char two) { return tracy::iupac(one, two); }

int ref_is_ambiguous(char c) { return tracy::isAmbiguous(c) ? 1 : 0; }

size_t ref_trimmed_seq(const char* s, size_t n, uint32_t ltrim, uint32_t rtrim, char* out) {
  std::string r = tracy::trimmedSeq(std::string(s, n), ltrim, rtrim);
  std::memcpy(out, r.data(), r.size());
  return r.size();
}

}
