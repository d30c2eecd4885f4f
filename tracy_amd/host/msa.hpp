// msa.hpp -- progressive multiple alignment of trace profiles (`tracy assemble`, SURVEY.md section 8(f) rank 3,
// BASELINE configs[4]) with every dynamic program on the device through the C ABI:
//   * the all-pairs distance matrix is ONE tracyhip_gotoh_score call (profile x profile, AlignConfig<true,true>),
//   * the guide tree (UPGMA) is host bookkeeping,
//   * the profile-to-profile alignments of the tree are batched by tree height: all nodes whose children are
//     finished go into one tracyhip_gotoh_align call.
//
// Mirrors of /root/reference/src (names, argument meaning):
//   _profileConsChar                     align.h:254-270
//   _createProfile(char MSA)             align.h:138-180
//   distanceMatrix, closestPair, updateDistanceMatrix, upgma      msa.h:33-91
//   palign (recursive there, by tree height here; same result)    msa.h:93-163
//   consensus                            msa.h:165-254
//   revSeqBasedOnDist                    msa.h:258-323
//   msa                                  msa.h:326-368
// PARITY UNPINNED (msa.h needs Boost); cross-checked against an independent Python restatement over the oracle.
#ifndef TRACY_AMD_MSA_HPP
#define TRACY_AMD_MSA_HPP

#include <algorithm>
#include <array>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/tracy_hip.h"
#include "tracy_host.hpp"

namespace tracy_amd {

typedef std::vector<std::string> CharAlign;  // boost::multi_array<char,2>: one string per row, equal lengths

inline char profileConsChar(Profile const& p, std::size_t pos) {  // align.h:254-270
  uint32_t maxidx = 0;
  double maxval = p(0, pos);
  for (uint32_t k = 1; k < 6; ++k)
    if (p(k, pos) > maxval) { maxval = p(k, pos); maxidx = k; }
  static const char letters[6] = {'A', 'C', 'G', 'T', 'N', 'N'};  // never '-': it would create gap-to-gap columns
  return letters[maxidx];
}

// _createProfile(char MSA), align.h:138-180: column frequencies over the rows that span the column (a row counts
// from its first to its last non-gap character; a row of gaps only counts everywhere)
inline void createProfile(CharAlign const& a, Profile& p) {
  const std::size_t nrow = a.size(), ncol = nrow ? a[0].size() : 0;
  p.resize(ncol);
  std::vector<int64_t> first(nrow, -1), last(nrow, (int64_t)ncol);
  for (std::size_t i = 0; i < nrow; ++i)
    for (std::size_t j = 0; j < ncol; ++j)
      if (a[i][j] != '-') {
        if (first[i] == -1) first[i] = (int64_t)j;
        last[i] = (int64_t)j;
      }
  for (std::size_t j = 0; j < ncol; ++j) {
    int sum = 0;
    float cnt[6] = {0, 0, 0, 0, 0, 0};
    for (std::size_t i = 0; i < nrow; ++i) {
      if (!(first[i] <= (int64_t)j && (int64_t)j <= last[i])) continue;
      ++sum;
      switch (a[i][j]) {
        case 'A': case 'a': cnt[0] += 1; break;
        case 'C': case 'c': cnt[1] += 1; break;
        case 'G': case 'g': cnt[2] += 1; break;
        case 'T': case 't': cnt[3] += 1; break;
        case 'N': case 'n': cnt[4] += 1; break;
        case '-': cnt[5] += 1; break;
        default: --sum; break;
      }
    }
    for (int k = 0; k < 6; ++k) p(k, j) = sum > 0 ? cnt[k] / sum : cnt[k];
  }
}

namespace detail {

struct ProfilePack {  // a list of profiles as one TRACYHIP_SEQ_PROFILE set
  std::vector<float> data;
  std::vector<uint64_t> off;
  std::vector<uint32_t> len;
  void add(Profile const& p) {
    off.push_back(data.size());
    len.push_back((uint32_t)p.cols);
    data.insert(data.end(), p.v.begin(), p.v.end());
  }
  tracyhip_seqset set() {
    if (data.empty()) data.push_back(0.0f);
    return tracyhip_seqset{TRACYHIP_SEQ_PROFILE, data.data(), off.data(), len.data(), (uint32_t)off.size()};
  }
};

}  // namespace detail

// the pair list of distanceMatrix (msa.h:33-42): i < j, row-major
inline void pairList(int32_t num, std::vector<uint32_t>& i1, std::vector<uint32_t>& i2) {
  i1.clear();
  i2.clear();
  for (int32_t i = 0; i < num; ++i)
    for (int32_t j = i + 1; j < num; ++j) { i1.push_back((uint32_t)i); i2.push_back((uint32_t)j); }
}

// gotohScore(a1[i1[k]], a2[i2[k]], AlignConfig<true,true>) for every k, on the device -- on all devices of `group` when one
// is given (the pair list is cut into slices of equal cell count, the profiles are replicated: tracyhip_group_gotoh_score)
inline int scorePairs(tracyhip_ctx* ctx, tracyhip_params prm, std::vector<Profile> const& a1, std::vector<Profile> const& a2,
                      std::vector<uint32_t> const& i1, std::vector<uint32_t> const& i2, std::vector<int32_t>& scores,
                      tracyhip_group* group = nullptr) {
  scores.assign(i1.size(), 0);
  if (i1.empty()) return TRACYHIP_OK;
  detail::ProfilePack p1, p2;
  for (auto const& p : a1) p1.add(p);
  for (auto const& p : a2) p2.add(p);
  tracyhip_pairs pr{};
  pr.npairs = (uint32_t)i1.size();
  pr.a1 = p1.set();
  pr.a2 = p2.set();
  pr.a1_index = i1.data();
  pr.a2_index = i2.data();
  prm.hfree = 1;
  prm.vfree = 1;
  if (group) return tracyhip_group_gotoh_score(group, &pr, &prm, scores.data());
  return tracyhip_gotoh_score(ctx, &pr, &prm, TRACYHIP_MEM_HOST, scores.data());
}

// closestPair / updateDistanceMatrix / upgma, msa.h:44-91.  d is (2n+1) x (2n+1), upper triangle, -1 = retired;
// p[node] = {parent, left, right}.  Returns the root.
inline int32_t upgma(std::vector<std::vector<int32_t>>& d, std::vector<std::array<int32_t, 3>>& p, int32_t num) {
  int32_t nn = num;
  for (; nn < 2 * num + 1; ++nn) {
    int32_t best = -1, dI = 0, dJ = 0;
    for (int32_t i = 0; i < nn; ++i)
      for (int32_t j = i + 1; j < nn; ++j)
        if (d[i][j] > best) { best = d[i][j]; dI = i; dJ = j; }
    if (best == -1) break;
    p[dI][0] = nn;
    p[dJ][0] = nn;
    p[nn][1] = dI;
    p[nn][2] = dJ;
    for (int32_t i = 0; i < nn; ++i)
      if (p[i][0] == -1) d[i][nn] = ((dI < i ? d[dI][i] : d[i][dI]) + (dJ < i ? d[dJ][i] : d[i][dJ])) / 2;
    for (int32_t i = 0; i < dI; ++i) d[i][dI] = -1;
    for (int32_t i = dI + 1; i < nn + 1; ++i) d[dI][i] = -1;
    for (int32_t i = 0; i < dJ; ++i) d[i][dJ] = -1;
    for (int32_t i = dJ + 1; i < nn + 1; ++i) d[dJ][i] = -1;
  }
  return nn > 0 ? nn - 1 : 0;
}

// msa(), msa.h:326-368: distance matrix -> UPGMA -> progressive alignment.  align: one row per sequence in the order
// seqidx gives (seqidx[row] = index into sps).
inline int msa(tracyhip_ctx* ctx, tracyhip_params prm, std::vector<Profile> const& sps, CharAlign& align, std::vector<uint32_t>& seqidx,
               tracyhip_group* group = nullptr) {
  const int32_t num = (int32_t)sps.size();
  align.clear();
  seqidx.clear();
  if (num == 0) return TRACYHIP_OK;
  const int32_t dim = 2 * num + 1;
  std::vector<std::vector<int32_t>> d(dim, std::vector<int32_t>(dim, -1));
  {
    std::vector<uint32_t> i1, i2;
    pairList(num, i1, i2);
    std::vector<int32_t> sc;
    const int rc = scorePairs(ctx, prm, sps, sps, i1, i2, sc, group);
    if (rc != TRACYHIP_OK) return rc;
    for (std::size_t k = 0; k < i1.size(); ++k) d[i1[k]][i2[k]] = sc[k];
  }
  std::vector<std::array<int32_t, 3>> p(dim, std::array<int32_t, 3>{-1, -1, -1});
  const int32_t root = upgma(d, p, num);

  // progressive alignment by tree height (palign, msa.h:93-163, is the post-order recursion of the same tree)
  struct Node { CharAlign align; Profile prof; std::vector<uint32_t> sidx; bool done = false; };
  std::vector<Node> node(dim);
  std::vector<int32_t> height(dim, -1);
  for (int32_t i = 0; i < dim; ++i) {
    if (p[i][1] == -1 && p[i][2] == -1 && i < num) {
      Node& n = node[i];
      n.align.assign(1, std::string(sps[i].cols, 'N'));
      for (std::size_t c = 0; c < sps[i].cols; ++c) n.align[0][c] = profileConsChar(sps[i], c);
      n.prof = sps[i];
      n.sidx.assign(1, (uint32_t)i);
      n.done = true;
      height[i] = 0;
    }
  }
  int32_t maxh = 0;
  for (int32_t i = num; i <= root; ++i) {  // children always have smaller indices than their parent
    if (p[i][1] < 0 || p[i][2] < 0) continue;
    height[i] = std::max(height[p[i][1]], height[p[i][2]]) + 1;
    maxh = std::max(maxh, height[i]);
  }
  prm.hfree = 1;
  prm.vfree = 1;
  for (int32_t h = 1; h <= maxh; ++h) {
    std::vector<int32_t> todo;
    for (int32_t i = num; i <= root; ++i)
      if (height[i] == h) todo.push_back(i);
    if (todo.empty()) continue;
    detail::ProfilePack p1, p2;
    std::vector<uint64_t> ooff;
    uint64_t ocap = 0;
    for (int32_t i : todo) {
      p1.add(node[p[i][1]].prof);
      p2.add(node[p[i][2]].prof);
      ooff.push_back(ocap);
      ocap += node[p[i][1]].prof.cols + node[p[i][2]].prof.cols;
    }
    tracyhip_pairs pr{};
    pr.npairs = (uint32_t)todo.size();
    pr.a1 = p1.set();
    pr.a2 = p2.set();
    std::vector<uint8_t> ops(ocap ? ocap : 1);
    std::vector<uint32_t> olen(todo.size());
    const int rc = tracyhip_gotoh_align(ctx, &pr, &prm, TRACYHIP_MEM_HOST, nullptr, ops.data(), ooff.data(), olen.data());
    if (rc != TRACYHIP_OK) return rc;
    for (std::size_t t = 0; t < todo.size(); ++t) {
      Node& n = node[todo[t]];
      Node& l = node[p[todo[t]][1]];
      Node& r = node[p[todo[t]][2]];
      const uint32_t ncol = olen[t];
      const std::size_t n1 = l.align.size(), n2 = r.align.size();
      n.align.assign(n1 + n2, std::string(ncol, '-'));
      uint32_t a1p = 0, a2p = 0;
      for (uint32_t j = 0; j < ncol; ++j) {
        const uint8_t op = ops[ooff[t] + (ncol - 1 - j)];  // ops are in push order (end of the alignment first)
        if (op != 'h') {  // alignNew[0][j] != '-'
          for (std::size_t k = 0; k < n1; ++k) n.align[k][j] = l.align[k][a1p];
          ++a1p;
        }
        if (op != 'v') {  // alignNew[1][j] != '-'
          for (std::size_t k = 0; k < n2; ++k) n.align[n1 + k][j] = r.align[k][a2p];
          ++a2p;
        }
      }
      createProfile(n.align, n.prof);
      n.sidx = l.sidx;
      n.sidx.insert(n.sidx.end(), r.sidx.begin(), r.sidx.end());
      n.done = true;
      l = Node();  // children are no longer needed
      r = Node();
    }
  }
  align = node[root].align;
  seqidx = node[root].sidx;
  return TRACYHIP_OK;
}

// consensus(), msa.h:165-254: majority letter per column over the rows covering it; qualities 47 + 10*count/rows
inline void consensus(float fractionCalled, CharAlign const& align, std::string& gapped, std::string& cs, std::string& qstr, bool ignoreLast) {
  const int64_t rows = (int64_t)align.size() - (ignoreLast ? 1 : 0);
  const std::size_t ncol = align.empty() ? 0 : align[0].size();
  std::vector<std::vector<bool>> fl((std::size_t)std::max<int64_t>(rows, 0), std::vector<bool>(ncol, false));
  std::vector<int32_t> cov(ncol, 0);
  for (int64_t i = 0; i < rows; ++i) {
    int start = 0, end = -1;
    for (std::size_t j = 0; j < ncol; ++j) {
      if (align[i][j] != '-') end = (int)j;
      else if (end == -1) start = (int)j + 1;
    }
    for (int j = start; j <= end; ++j) { ++cov[j]; fl[i][j] = true; }
  }
  const int32_t covThreshold = (int32_t)(fractionCalled * (float)(std::size_t)rows);  // float x size_t, msa.h:196
  const int32_t totCount = (int32_t)rows;
  std::string cons(ncol, '-'), qual(ncol, '#');
  int32_t qualval = 33;
  for (std::size_t j = 0; j < ncol; ++j) {
    int32_t maxIdx = 4, maxCount = 0;
    if (cov[j] >= 1 && cov[j] >= covThreshold) {
      int32_t count[5] = {0, 0, 0, 0, 0};
      for (int64_t i = 0; i < rows; ++i) {
        if (!fl[i][j]) continue;
        switch (align[i][j]) {
          case 'A': case 'a': ++count[0]; break;
          case 'C': case 'c': ++count[1]; break;
          case 'G': case 'g': ++count[2]; break;
          case 'T': case 't': ++count[3]; break;
          default: ++count[4]; break;
        }
      }
      maxIdx = 0;
      maxCount = count[0];
      for (int k = 1; k < 5; ++k)
        if (count[k] > maxCount) { maxCount = count[k]; maxIdx = k; }
      qualval = 47 + maxCount * 10 / totCount;
    }
    if (maxIdx < 4) { cons[j] = "ACGT"[maxIdx]; qual[j] = (char)qualval; }
  }
  gapped = cons;
  for (std::size_t j = 0; j < ncol; ++j)
    if (cons[j] != '-') { cs.push_back(cons[j]); qstr.push_back(qual[j]); }
}

// revSeqBasedOnDist, msa.h:258-323: greedy strand assignment -- flip a sequence when the summed all-vs-one score of
// its reverse complement is at least as good; repeat while the total improves.  Scores on the device, in batches.
inline int revSeqBasedOnDist(tracyhip_ctx* ctx, tracyhip_params prm, std::vector<Profile>& seq, std::vector<bool>& fwd) {
  const int32_t num = (int32_t)seq.size();
  std::vector<std::vector<int32_t>> d(num, std::vector<int32_t>(num, 0));
  int32_t totalScore = 0;
  {
    std::vector<uint32_t> i1, i2;
    for (int32_t i = 0; i < num; ++i)
      for (int32_t j = i + 1; j < num; ++j) { i1.push_back((uint32_t)i); i2.push_back((uint32_t)j); }
    std::vector<int32_t> sc;
    const int rc = scorePairs(ctx, prm, seq, seq, i1, i2, sc);
    if (rc != TRACYHIP_OK) return rc;
    for (std::size_t k = 0; k < i1.size(); ++k) { d[i1[k]][i2[k]] = d[i2[k]][i1[k]] = sc[k]; totalScore += sc[k]; }
  }
  bool iterate = true;
  while (iterate) {
    std::vector<std::pair<int32_t, int32_t>> quality;
    for (int32_t i = 0; i < num; ++i) {
      int32_t rowSum = 0;
      for (int32_t j = 0; j < num; ++j) rowSum += d[i][j];
      quality.push_back(std::make_pair(rowSum, i));
    }
    std::sort(quality.begin(), quality.end());  // worst sequence first
    for (auto const& q : quality) {
      const int32_t who = q.second;
      std::vector<Profile> flipped(1);
      reverseComplementProfile(seq[who], flipped[0]);
      std::vector<uint32_t> i1, i2;
      for (int32_t i = 0; i < num; ++i)
        if (i != who) { i1.push_back((uint32_t)i); i2.push_back(0); }
      std::vector<int32_t> sc;
      const int rc = scorePairs(ctx, prm, seq, flipped, i1, i2, sc);
      if (rc != TRACYHIP_OK) return rc;
      int32_t scoreSum = 0, oldScoreSum = 0;
      for (std::size_t k = 0; k < i1.size(); ++k) { scoreSum += sc[k]; oldScoreSum += d[i1[k]][who]; }
      if (scoreSum >= oldScoreSum) {
        seq[who] = flipped[0];
        fwd[who] = !fwd[who];
        for (std::size_t k = 0; k < i1.size(); ++k) d[i1[k]][who] = d[who][i1[k]] = sc[k];
        d[who][who] = 0;  // the reference writes newD[who] = 0 onto the diagonal
      }
      std::cout << "." << std::flush;
    }
    int32_t updated = 0;
    for (int32_t i = 0; i < num; ++i)
      for (int32_t j = 0; j < num; ++j) updated += d[i][j];
    if (totalScore < updated) totalScore = updated;
    else iterate = false;
  }
  std::cout << std::endl;
  return TRACYHIP_OK;
}

}  // namespace tracy_amd
#endif
