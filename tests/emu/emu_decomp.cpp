// emu_decomp.cpp -- host execution of the decompose phase functions (TEST INFRASTRUCTURE ONLY): the
// same tracy_amd/csrc/decompose_kernels.h code the HIP kernel runs, with the lanes looped on the CPU.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../tracy_amd/csrc/decompose_kernels.h"
#include "../../tracy_amd/csrc/decompose_wave.h"

using namespace tracyhip;

#include "host_wave.h"

extern "C" {

int emu_decompose(const uint8_t* row0, const uint8_t* row1, uint32_t L, uint8_t* primary, uint8_t* secondary, uint32_t nbc,
                  uint32_t breakpoint, uint32_t refslice_len, int32_t trimLeft, int32_t trimRight, int32_t maxindel,
                  int32_t madc, int32_t* dcp_indel, int32_t* dcp_err, int32_t* out6, int32_t force_bytewise) {
  DecompDesc d{0, 0, 0, L, nbc, refslice_len, breakpoint};
  DecompOut out{};
  DecompArgs a{};
  a.desc = &d; a.rows0 = row0; a.rows1 = row1; a.primary = primary; a.secondary = secondary;
  a.dcp_indel = dcp_indel; a.dcp_err = dcp_err; a.out = &out;
  a.prm = DecompParams{trimLeft, trimRight, maxindel, madc};
  a.ntraces = 1;
  static DecompShared sh;
  std::memset(&sh, 0, sizeof(sh));
  for (int st = 0; st < kDecompSteps; ++st) {
    if (decomp_step_all_lanes(st)) {
      // every lane keeps its own copy of `out` on the device; lane 0's copy is the one that is stored
      DecompOut scratch = out;
      for (uint32_t l = 1; l < 64; ++l) { DecompOut o = scratch; decomp_step(st, a, d, sh, o, l); }
      decomp_step(st, a, d, sh, out, 0);
    } else {
      decomp_step(st, a, d, sh, out, 0);
    }
    if (force_bytewise && st == 4) sh.exotic = 1;
  }
  out6[0] = out.kind; out6[1] = out.bestIns; out6[2] = out.bestDel; out6[3] = out.bestFR; out6[4] = (int32_t)out.dcp_n;
  return 0;
}

// the one-wave body of decompose_wave.h on the 64-fiber host wave; returns 1 when the body left the trace to decompose_kernel (todo)
int emu_decompose_wave(const uint8_t* row0, const uint8_t* row1, uint32_t L, uint8_t* primary, uint8_t* secondary, uint32_t nbc,
                       uint32_t breakpoint, uint32_t refslice_len, int32_t trimLeft, int32_t trimRight, int32_t maxindel,
                       int32_t madc, int32_t* dcp_indel, int32_t* dcp_err, int32_t* out6, uint32_t capL, uint32_t capB, uint32_t capI, uint32_t capF) {
  DecompDesc d{0, 0, 0, L, nbc, refslice_len, breakpoint};
  BreakpointOut bp{};
  bp.breakpoint = breakpoint;
  DecompOut out{};
  DecompWaveArgs wa{};
  wa.a.desc = &d; wa.a.rows0 = row0; wa.a.rows1 = row1; wa.a.primary = primary; wa.a.secondary = secondary;
  wa.a.dcp_indel = dcp_indel; wa.a.dcp_err = dcp_err; wa.a.out = &out;
  wa.a.prm = DecompParams{trimLeft, trimRight, maxindel, madc};
  wa.a.ntraces = 1;
  wa.bps = &bp;
  alignas(8) uint8_t lut[kLutBytes];
  decomp_lut_build(lut);
  wa.lut = lut;
  uint32_t todo = 7;
  wa.todo = &todo;
  wa.caps = DecompWaveCaps{capL, capB, capI, capF};
  if (!decomp_wave_caps_ok(wa.caps)) return -1;
  WaveShared sh;
  sh.lds.assign(decomp_wave_layout(wa.caps).total + 64, (char)0x5a);  // (stale LDS: the body must initialise what it reads)
  sh.run([&](uint32_t l) { HostWave w{l, &sh}; decomp_wave_body(w, wa, 0); });
  out6[0] = out.kind; out6[1] = out.bestIns; out6[2] = out.bestDel; out6[3] = out.bestFR; out6[4] = (int32_t)out.dcp_n; out6[5] = (int32_t)decomp_wave_layout(wa.caps).total;
  return (int)todo;
}

void emu_find_breakpoint(const float* prof, uint32_t stride, uint32_t ncol, int32_t* out4, float* bestdiff) {
  std::vector<double> sig(ncol + 1), diff(ncol + 1);
  std::vector<uint8_t> ltr(ncol + 1);
  for (uint32_t j = 0; j < ncol; ++j) sig[j] = signal_ratio(prof, stride, j);
  if (25 < ncol)
    for (uint32_t i = 25; i < ncol - 25; ++i) {
      double l, r;
      diff[i] = window_diff(sig.data(), i, &l, &r);
      ltr[i] = l < r;
    }
  BreakpointOut bp;
  breakpoint_select(diff.data(), ltr.data(), ncol, bp);
  out4[0] = bp.indelshift; out4[1] = bp.traceleft; out4[2] = (int32_t)bp.breakpoint;
  *bestdiff = bp.bestDiff;
}

int emu_homozygous(const uint8_t* row0, const uint8_t* row1, uint32_t L, int32_t* out4, float* bestdiff) {
  BreakpointOut bp{};
  const int rc = homozygous_breakpoint(row0, row1, L, bp);
  out4[0] = bp.indelshift; out4[1] = bp.traceleft; out4[2] = (int32_t)bp.breakpoint;
  *bestdiff = bp.bestDiff;
  return rc;
}

uint8_t emu_secdecomp(uint8_t p, uint8_t s, const int32_t* trace, uint64_t nsamples, int32_t pos) {
  return secondary_decomposed(p, s, trace, nsamples, pos);
}
char emu_phase(char p, char s, char r) { return phase_ref_allele(p, s, r); }
}
