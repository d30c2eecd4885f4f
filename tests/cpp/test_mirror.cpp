// test_mirror.cpp -- the reference-shaped C++ calls (tracy_amd.hpp) against the oracle, on the GPU.
// Reads like a reference call site: gotoh(a1, a2, align, semiglobal, sc).
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../oracle/tracy_oracle.h"
#include "../../tracy_amd/host/tracy_amd.hpp"

using namespace tracy_amd;

static std::string rnd(unsigned& s, size_t n) {
  std::string r(n, 'A');
  for (auto& c : r) { s = s * 1664525u + 1013904223u; c = "ACGT"[(s >> 24) & 3]; }
  return r;
}

int main() {
  unsigned seed = 7;
  DnaScore<int32_t> sc(3, -5, -10, -4);
  AlignConfig<true, false> semiglobal;
  AlignConfig<false, false> global;
  orc_score os{3, -5, -10, -4};
  int fails = 0;
  for (int it = 0; it < 6; ++it) {
    std::string ref = rnd(seed, 300 + 97 * it), q = ref.substr(40, 150 + 20 * it);
    q[10] = 'N'; q.erase(60, 3);
    Alignment align;
    int s = gotoh(q, ref, align, semiglobal, sc);
    std::vector<char> btr(q.size() + ref.size() + 1), r0(btr.size()), r1(btr.size());
    size_t bl = 0;
    int want = orc_gotoh_str(q.data(), q.size(), ref.data(), ref.size(), 1, 0, &os, btr.data(), &bl);
    orc_create_alignment_str(btr.data(), bl, q.data(), ref.data(), r0.data(), r1.data());
    if (s != want || align[0] != std::string(r0.data(), bl) || align[1] != std::string(r1.data(), bl) || align.shape(1) != bl) ++fails;
    if (gotohScore(q, ref, semiglobal, sc) != want) ++fails;
    if (gotohScore(q, ref, global, sc) != orc_gotoh_score_str(q.data(), q.size(), ref.data(), ref.size(), 0, 0, &os)) ++fails;
    // profile call shape of sage.h:239-258
    Profile pq, pr;
    createProfile(q, pq);
    createProfile(ref, pr);
    int sp = gotoh(pq, pr, align, semiglobal, sc);
    std::vector<float> f1(pq.v), f2(pr.v);
    int wantp = orc_gotoh_prof(f1.data(), q.size(), f2.data(), ref.size(), 1, 0, &os, btr.data(), &bl);
    if (sp != wantp || align.shape(1) != bl) ++fails;
  }
  std::printf("mirror test: %d failures\n", fails);
  return fails ? 1 : 0;
}
