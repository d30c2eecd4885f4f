// test_mirror.cpp -- the reference-shaped C++ calls (tracy_amd.hpp) against the oracle, on the GPU.
// Reads like a reference call site: gotoh(a1, a2, align, semiglobal, sc).
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../oracle/tracy_oracle.h"
#include "../../oracle/tracy_oracle_decompose.h"
#include "../../tracy_amd/host/tracy_amd.hpp"

extern "C" uint32_t tracyhost_synth_decompose(uint64_t seed, uint32_t n, uint32_t mf, uint32_t maxlen, int kind, double frac1, uint8_t* ref_out,
                                              int32_t* signal, uint32_t nsamples_cap, int32_t* basecallpos, int32_t* indel_out);

using namespace tracy_amd;

static std::string rnd(unsigned& s, size_t n) {
  std::string r(n, 'A');
  for (auto& c : r) { s = s * 1664525u + 1013904223u; c = "ACGT"[(s >> 24) & 3]; }
  return r;
}

// The hot section of indigo.h:314-387 written as the reference writes it, against tracy_amd.hpp, beside the same chain
// through the oracle's C functions.
struct IndigoConfig {  // the fields the hot section reads (indigo.h:16-40)
  uint16_t trimLeft = 50, trimRight = 50, maxindel = 1000, madc = 5;
  DnaScore<int32_t> aliscore = DnaScore<int32_t>(3, -5, -10, -4);
};

static int decompose_half() {
  int fails = 0;
  IndigoConfig c;
  orc_score os{3, -5, -10, -4};
  orc_decomp_cfg oc{c.trimLeft, c.trimRight, c.maxindel, c.madc};
  for (int it = 0; it < 6; ++it) {
    const uint32_t n = 1600, mf = 520, ns = 12 * mf + 12;
    std::vector<uint8_t> refb(n);
    std::vector<int32_t> sig(4 * (size_t)ns), pos(mf + 64);
    int32_t indel = 0;
    const int kind = (it == 4) ? 2 : (it == 5) ? 3 : 0;  // het indels, one homozygous indel, one without variant
    const uint32_t npos = tracyhost_synth_decompose(900 + it, n, mf, 20, kind | ((it & 1) ? 16 : 0), 0.6, refb.data(), sig.data(), ns, pos.data(), &indel);
    Trace tr;
    tr.traceACGT.resize(4);
    for (int k = 0; k < 4; ++k) tr.traceACGT[k].assign(sig.begin() + (size_t)k * ns, sig.begin() + (size_t)(k + 1) * ns);
    tr.basecallpos.assign(pos.begin(), pos.begin() + npos);
    BaseCalls bc;
    basecall(tr, bc, 0.33f);
    ReferenceSlice rs;
    rs.refslice.assign(refb.begin(), refb.end());
    rs.forward = true;
    rs.pos = 0;
    rs.filetype = 1;
    // orientation as indigo.h:235-247
    Profile ptrace, prefslice, prevslice;
    createProfile(tr, bc, ptrace, c.trimLeft, c.trimRight);
    createProfile(rs.refslice, prefslice);
    reverseComplementProfile(prefslice, prevslice);
    AlignConfig<true, false> semiglobal;
    if (!(gotohScore(ptrace, prefslice, semiglobal, c.aliscore) > gotohScore(ptrace, prevslice, semiglobal, c.aliscore))) {
      std::string rc(rs.refslice.rbegin(), rs.refslice.rend());
      for (auto& ch : rc) ch = ch == 'A' ? 'T' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : ch == 'T' ? 'A' : ch;
      rs.refslice = rc;
      rs.forward = false;
      prefslice = prevslice;
    }
    TraceBreakpoint bp;
    findBreakpoint(ptrace, bp);
    Alignment align;
    gotoh(ptrace, prefslice, align, semiglobal, c.aliscore);
    // ---- oracle side, same inputs ----
    BaseCalls obc = bc;
    orc_breakpoint obp{bp.indelshift ? 1 : 0, bp.traceleft ? 1 : 0, bp.breakpoint, bp.bestDiff};
    const size_t L = align.shape(1);
    int orc_ok = 1;
    if (!obp.indelshift) orc_ok = orc_find_homozygous_breakpoint(align[0].data(), align[1].data(), L, &obp);
    // ---- indigo.h:314-317 ----
    bool ok = true;
    if (!bp.indelshift) ok = findHomozygousBreakpoint(align, bp);
    if (ok != (orc_ok == 1)) { ++fails; continue; }
    if (!ok) continue;
    if (bp.indelshift != (obp.indelshift != 0) || bp.traceleft != (obp.traceleft != 0) || bp.breakpoint != obp.breakpoint || bp.bestDiff != obp.bestDiff) ++fails;
    // ---- indigo.h:340-350 ----
    typedef std::vector<std::pair<int32_t, int32_t>> TDecomposition;
    TDecomposition dcp;
    if (!decomposeAlleles(c, align, bc, bp, rs, dcp)) ++fails;
    generateSecondaryDecomposed(tr, bc);
    std::pair<double, double> a1a2 = allelicFraction(c, tr, bc);
    std::vector<int32_t> di(2 * c.maxindel + 4), de(2 * c.maxindel + 4);
    size_t dn = 0;
    orc_decomp_status ost;
    std::string opri = obc.primary, osec = obc.secondary;
    orc_decompose_alleles(&oc, align[0].data(), align[1].data(), L, &opri[0], &osec[0], opri.size(), obp, rs.refslice.size(), di.data(), de.data(), &dn, &ost);
    std::string osd(osec.size(), 'N');
    orc_generate_secondary_decomposed(sig.data(), ns, obc.bcPos.data(), opri.data(), osec.data(), opri.size(), &osd[0]);
    double oi = 0, oj = 0;
    orc_allelic_fraction(sig.data(), ns, obc.bcPos.data(), opri.data(), osd.data(), opri.size(), c.trimLeft, c.trimRight, &oi, &oj);
    if (bc.primary != opri || bc.secondary != osec || bc.secDecompose != osd) ++fails;
    if (dcp.size() != dn) ++fails;
    for (size_t k = 0; k < dn && k < dcp.size(); ++k)
      if (dcp[k].first != di[k] || dcp[k].second != de[k]) { ++fails; break; }
    if (a1a2.first != oi || a1a2.second != oj) ++fails;
    // ---- indigo.h:355-387: allele-specific alignments ----
    for (int allele = 0; allele < 2; ++allele) {
      const std::string seq = trimmedSeq(allele == 0 ? bc.primary : bc.secDecompose, c.trimLeft, c.trimRight);
      Alignment first, final;
      gotoh(seq, rs.refslice, first, semiglobal, c.aliscore);
      ReferenceSlice trimmed(rs);
      trimReferenceSlice(c, first, trimmed);
      const int sc = gotoh(seq, trimmed.refslice, final, semiglobal, c.aliscore);
      orc_trim_result tr_;
      orc_trim_reference_slice(first[0].data(), first[1].data(), first.shape(1), c.trimLeft, c.trimRight, rs.refslice.size(), rs.forward ? 1 : 0, &tr_);
      std::string want_slice = tr_.ri <= rs.refslice.size() ? rs.refslice.substr(tr_.ri, tr_.risize) : std::string();
      if (trimmed.refslice != want_slice || trimmed.pos != rs.pos + tr_.pos_add) ++fails;
      std::vector<char> btr(seq.size() + want_slice.size() + 1);
      size_t bl = 0;
      if (sc != orc_gotoh_str(seq.data(), seq.size(), want_slice.data(), want_slice.size(), 1, 0, &os, btr.data(), &bl) || final.shape(1) != bl) ++fails;
    }
    AlignConfig<false, false> global;
    Alignment final3;
    const std::string pri = trimmedSeq(bc.primary, c.trimLeft, c.trimRight), sec = trimmedSeq(bc.secDecompose, c.trimLeft, c.trimRight);
    std::vector<char> btr(pri.size() + sec.size() + 1);
    size_t bl = 0;
    if (gotoh(pri, sec, final3, global, c.aliscore) != orc_gotoh_str(pri.data(), pri.size(), sec.data(), sec.size(), 0, 0, &os, btr.data(), &bl)) ++fails;
  }
  return fails;
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
  fails += decompose_half();
  std::printf("mirror test: %d failures\n", fails);
  return fails ? 1 : 0;
}
