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

size_t ref_basecall_qual(const int32_t* trace, size_t nsamples, const int32_t* basecallpos, size_t npos,
                         float sigratio, char* primary, char* secondary, char* consensus, int32_t* bcpos, uint8_t* estqual) {
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
  std::memcpy(estqual, bc.estQual.data(), n);
  return n;
}

void* ref_trace_read(const char* path) {
  tracy::Trace* tr = new tracy::Trace();
  if (!tracy::readab(path, *tr)) { delete tr; return nullptr; }
  return tr;
}
void ref_trace_dims(const void* h, uint64_t* nsamples, uint64_t* ncalls) {
  const tracy::Trace* tr = static_cast<const tracy::Trace*>(h);
  size_t ns = 0;
  for (auto const& c : tr->traceACGT) ns = std::max(ns, c.size());
  *nsamples = ns;
  *ncalls = tr->basecallpos.size();
}
void ref_trace_get(const void* h, int32_t* signal, int32_t* basecallpos, char* basecalls1, char* basecalls2, uint8_t* qual) {
  const tracy::Trace* tr = static_cast<const tracy::Trace*>(h);
  uint64_t ns, nc;
  ref_trace_dims(h, &ns, &nc);
  for (size_t k = 0; k < 4; ++k)
    for (size_t i = 0; i < ns; ++i) signal[k * ns + i] = (k < tr->traceACGT.size() && i < tr->traceACGT[k].size()) ? tr->traceACGT[k][i] : 0;
  for (size_t i = 0; i < nc; ++i) basecallpos[i] = tr->basecallpos[i];
  std::memcpy(basecalls1, tr->basecalls1.data(), std::min<size_t>(nc, tr->basecalls1.size()));
  std::memcpy(basecalls2, tr->basecalls2.data(), std::min<size_t>(nc, tr->basecalls2.size()));
  std::memcpy(qual, tr->qual.data(), std::min<size_t>(nc, tr->qual.size()));
}
void ref_trace_free(void* h) { delete static_cast<tracy::Trace*>(h); }

int32_t ref_trace_txt(const char* outfile, const int32_t* trace, size_t nsamples, const int32_t* basecallpos, size_t npos,
                      float sigratio, uint32_t left_trim, uint32_t right_trim) {
  tracy::Trace tr;
  tr.traceACGT.resize(4);
  for (int k = 0; k < 4; ++k) tr.traceACGT[k].assign(trace + k * nsamples, trace + (k + 1) * nsamples);
  tr.basecallpos.assign(basecallpos, basecallpos + npos);
  tracy::BaseCalls bc;
  tracy::basecall(tr, bc, sigratio);
  if (bc.bcPos.empty()) return -1;
  tracy::traceTxtOut(outfile, bc, tr, left_trim, right_trim);
  return 0;
}

}
