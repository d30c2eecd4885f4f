// msa_capi.cpp -- C wrappers over msa.hpp (libtracy_msa.so, linked against libtracy_hip.so) for the Python tests
// and tools: the progressive multiple alignment of `tracy assemble` with its DPs on the device.
#include <cstring>

#include "msa.hpp"

using namespace tracy_amd;

namespace {
void unpack(const float* data, const uint64_t* off, const uint32_t* len, uint32_t n, std::vector<Profile>& out) {
  out.resize(n);
  for (uint32_t i = 0; i < n; ++i) {
    out[i].resize(len[i]);
    std::memcpy(out[i].v.data(), data + off[i], sizeof(float) * 6 * len[i]);
  }
}
}  // namespace

extern "C" {

// the pair list of distanceMatrix (msa.h:33-42) as msa() hands it to the device: i1 / i2 of capacity n (n - 1) / 2
uint64_t tracymsa_pair_list(uint32_t n, uint32_t* i1, uint32_t* i2) {
  std::vector<uint32_t> a, b;
  pairList((int32_t)n, a, b);
  if (i1 && i2) {
    std::memcpy(i1, a.data(), sizeof(uint32_t) * a.size());
    std::memcpy(i2, b.data(), sizeof(uint32_t) * b.size());
  }
  return a.size();
}

// msa() with the all-pairs distance matrix spread over the devices of a group (group may be null)
int64_t tracymsa_msa_group(tracyhip_ctx* ctx, tracyhip_group* group, const tracyhip_params* prm, const float* data, const uint64_t* off,
                           const uint32_t* len, uint32_t n, char* rows, uint64_t rows_cap, uint32_t* seqidx, uint32_t* nrows) {
  std::vector<Profile> sps;
  unpack(data, off, len, n, sps);
  CharAlign align;
  std::vector<uint32_t> sidx;
  const int rc = msa(ctx, *prm, sps, align, sidx, group);
  if (rc != TRACYHIP_OK) return rc;
  const uint64_t ncol = align.empty() ? 0 : align[0].size();
  if (ncol * align.size() > rows_cap) return -100;
  for (std::size_t i = 0; i < align.size(); ++i) std::memcpy(rows + i * ncol, align[i].data(), ncol);
  for (std::size_t i = 0; i < sidx.size(); ++i) seqidx[i] = sidx[i];
  *nrows = (uint32_t)align.size();
  return (int64_t)ncol;
}

// msa(): rows (nseq x ncols bytes, row-major) into `rows` (capacity rows_cap), seqidx[nseq]; returns ncols or < 0
int64_t tracymsa_msa(tracyhip_ctx* ctx, const tracyhip_params* prm, const float* data, const uint64_t* off, const uint32_t* len, uint32_t n,
                     char* rows, uint64_t rows_cap, uint32_t* seqidx, uint32_t* nrows) {
  std::vector<Profile> sps;
  unpack(data, off, len, n, sps);
  CharAlign align;
  std::vector<uint32_t> sidx;
  const int rc = msa(ctx, *prm, sps, align, sidx);
  if (rc != TRACYHIP_OK) return rc;
  const uint64_t ncol = align.empty() ? 0 : align[0].size();
  if (ncol * align.size() > rows_cap) return -100;
  for (std::size_t i = 0; i < align.size(); ++i) std::memcpy(rows + i * ncol, align[i].data(), ncol);
  for (std::size_t i = 0; i < sidx.size(); ++i) seqidx[i] = sidx[i];
  *nrows = (uint32_t)align.size();
  return (int64_t)ncol;
}

// consensus(): gapped[ncols], cs / qstr (capacity ncols); returns the consensus length
int64_t tracymsa_consensus(float fraction_called, const char* rows, uint32_t nrows, uint64_t ncols, int32_t ignore_last, char* gapped,
                           char* cs, char* qstr) {
  CharAlign align(nrows);
  for (uint32_t i = 0; i < nrows; ++i) align[i].assign(rows + (uint64_t)i * ncols, ncols);
  std::string g, c, q;
  consensus(fraction_called, align, g, c, q, ignore_last != 0);
  std::memcpy(gapped, g.data(), g.size());
  std::memcpy(cs, c.data(), c.size());
  std::memcpy(qstr, q.data(), q.size());
  return (int64_t)c.size();
}

// _createProfile(char MSA): out[6][ncols]
void tracymsa_profile_of_alignment(const char* rows, uint32_t nrows, uint64_t ncols, float* out) {
  CharAlign align(nrows);
  for (uint32_t i = 0; i < nrows; ++i) align[i].assign(rows + (uint64_t)i * ncols, ncols);
  Profile p;
  createProfile(align, p);
  std::memcpy(out, p.v.data(), sizeof(float) * 6 * ncols);
}

// revSeqBasedOnDist(): profiles are rewritten in place (same lengths), fwd[n] flipped where the strand changed
int32_t tracymsa_rev_seq(tracyhip_ctx* ctx, const tracyhip_params* prm, float* data, const uint64_t* off, const uint32_t* len, uint32_t n,
                         uint8_t* fwd) {
  std::vector<Profile> seq;
  unpack(data, off, len, n, seq);
  std::vector<bool> f(n);
  for (uint32_t i = 0; i < n; ++i) f[i] = fwd[i] != 0;
  const int rc = revSeqBasedOnDist(ctx, *prm, seq, f);
  if (rc != TRACYHIP_OK) return rc;
  for (uint32_t i = 0; i < n; ++i) {
    std::memcpy(data + off[i], seq[i].v.data(), sizeof(float) * 6 * len[i]);
    fwd[i] = f[i] ? 1 : 0;
  }
  return 0;
}

}  // extern "C"
