// emu_wave.cpp -- host execution of the wave-level kernel bodies (TEST INFRASTRUCTURE ONLY).
// Runs tracy_amd/csrc/dp_kernels.h as a 64-lane "wave": one fiber (ucontext) per lane on ONE thread, every cross-lane
// shift a barrier at which the lane yields to the next one -- so that the index math, tag arithmetic, traceback layout and
// walker can be checked against the oracle in the CPU-only container.  (Lanes run round-robin from barrier to barrier: all
// lanes of a wave pass the same sequence of barriers, which is all a barrier promises.  One std::thread per lane did the same
// with 64 futex waits per barrier: six minutes of system time for 47 s of arithmetic.)  Never linked into the product library.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#include "../../tracy_amd/csrc/band16.h"
#include "../../tracy_amd/csrc/dp_kernels.h"
#include "../../tracy_amd/csrc/front.h"

using namespace tracyhip;

#include "host_wave.h"

namespace {
template <int K, int MODE, bool TRACE, bool NEEDLE, bool NARROW = false, bool COMPACT = false>
void run_wave(const DpArgs& a) {
  WaveShared sh;
  sh.lds.assign((qp_like(MODE) ? std::max(lds_bytes(MODE_QP, K), lds_bytes_sweep16(K, false)) : needle_lds_bytes(MODE_PROF, K)) + 64, 0);
  sh.run([&](uint32_t l) {
    HostWave w{l, &sh};
    if constexpr (NEEDLE) {
      if constexpr (MODE != MODE_QP) needle_body<HostWave, K, MODE, TRACE>(w, a, 0);
    } else {
      gotoh_body<HostWave, K, MODE, TRACE, NARROW, false, 0, COMPACT>(w, a, 0);
    }
  });
}

template <int K>
void dispatch_narrow(int mode, const DpArgs& a) {
  if (mode == MODE_CHAR) run_wave<K, MODE_CHAR, false, false, true>(a);
  else if (mode == MODE_CQ) {  // strings through the table: both forms of the 16-bit sweep
    run_wave<K, MODE_CQ, false, false, true, true>(a);
    run_wave<K, MODE_CQ, false, false, true, false>(a);
  } else if (mode == MODE_PROF) {  // the profile x profile score kernel with 16-bit cells
    if constexpr (K == 4 || K == 8) run_wave<K, MODE_PROF, false, false, false, true>(a);
  } else {  // both forms of the 16-bit query-profile sweep, as the library launches them: exactly one of them takes the pair
    run_wave<K, MODE_QP, false, false, true, true>(a);
    run_wave<K, MODE_QP, false, false, true, false>(a);
  }
}

template <int K, bool NEEDLE>
void dispatch(int mode, bool trace, const DpArgs& a) {
  if (mode == MODE_CHAR) trace ? run_wave<K, MODE_CHAR, true, NEEDLE>(a) : run_wave<K, MODE_CHAR, false, NEEDLE>(a);
  else if (mode == MODE_QP) trace ? run_wave<K, MODE_QP, true, NEEDLE>(a) : run_wave<K, MODE_QP, false, NEEDLE>(a);
  else if (mode == MODE_CQ) { if constexpr (!NEEDLE) run_wave<K, MODE_CQ, true, false>(a); }  // tracebacks only
  else trace ? run_wave<K, MODE_PROF, true, NEEDLE>(a) : run_wave<K, MODE_PROF, false, NEEDLE>(a);
}
}  // namespace

template <int K, int MODE, bool NARROW>
void run_ckpt_pair(const DpArgs& a, const WalkArgs& wa) {
  for (int form = 0; form < ((NARROW && MODE == MODE_QP) ? 2 : 1); ++form) {  // checkpointed score pass (both forms of the 16-bit sweep)
    WaveShared sh;
    sh.lds.assign(std::max(lds_bytes(MODE_QP, K), lds_bytes_sweep16(K, false)) + 64, 0);
    sh.run([&](uint32_t l) {
        HostWave w{l, &sh};
        if (form == 0) gotoh_body<HostWave, K, MODE, false, NARROW, true, 0, NARROW && MODE == MODE_QP>(w, a, 0);
        else gotoh_body<HostWave, K, MODE, false, NARROW, true, 0, false>(w, a, 0);
      });
  }
  {  // band traceback
    WaveShared sh;
    sh.lds.assign(lds_bytes(MODE_QP, K) + 64, 0);
    sh.run([&](uint32_t l) { HostWave w{l, &sh}; gotoh_band_trace_body<HostWave, K, MODE>(w, a, wa, 0); });
  }
}

template <int K, int TABLE = 0, int NC = 6>
void run_origin_wave(const DpArgs& a) {
  WaveShared sh;
  sh.lds.assign(lds_bytes(MODE_CQ, K) + 64, 0);
  sh.run([&](uint32_t l) { HostWave w{l, &sh}; gotoh_origin_body<HostWave, K, TABLE, NC>(w, a, 0); });
}
// MODE_CQ: case-sensitive codes of a string, padded like the library's code buffers
static std::vector<uint8_t> cq_codes(const void* a2, size_t bytes) {
  std::vector<uint8_t> v(bytes + 256, 5);
  const uint8_t* p = static_cast<const uint8_t*>(a2);
  for (size_t i = 0; i < bytes; ++i) v[128 + i] = (uint8_t)cq_code(p[i]);
  return v;
}

template <int K, int GL = kPrefixLanes>
void run_prefix_wave(const DpArgs& a, uint32_t npairs) {
  for (int form = 0; form < 2; ++form) {  // both forms, as the library launches them: every group is worked on in one of them
    WaveShared sh;
    sh.lds.assign(lds_bytes_prefix(K, false) + 64, 0);
    sh.run([&](uint32_t l) {
        HostWave w{l, &sh};
        if (form == 0) gotoh_prefix_body<HostWave, K, GL, true>(w, a, 0, npairs);
        else gotoh_prefix_body<HostWave, K, GL, false>(w, a, 0, npairs);
      });
  }
}

// MODE_QP kernels read the code buffer without clamping (idle lanes, look-ahead): the library pads it (kCodePad in
// capi_internal.h) with code 5 (an all-zero column); the emulator does the same.  Idle lanes of the steady-state steps of the
// 16-bit sweep really look such a byte up in the table (and discard the result), so it has to be a code.
static std::vector<uint8_t> padded_codes(const void* a2, size_t bytes) {
  std::vector<uint8_t> v(bytes + 256, 5);
  if (bytes) std::memcpy(v.data() + 128, a2, bytes);
  for (auto& c : v) if (c > 5) c = 5;  // as the library's encoders (dp_code)
  return v;
}

// DpArgs::special_blocks of a padded code buffer (codes start at v[128])
static std::vector<uint8_t> special_blocks_of(const std::vector<uint8_t>& v, size_t n) {
  std::vector<uint8_t> b((n >> 8) + 2, 0);
  for (size_t i = 0; i < n; ++i) if (v[128 + i] >= 4) b[i >> 8] = 1;
  return b;
}

// The screened profile x profile substitution score (SubProf::screen) against the exact chain (SubProf::prepare) on n column
// pairs (a, b: n x 5 floats).  counts[0] = strips the screen could not prove, counts[1] = proven strips whose int differs
// from the exact one (must stay 0).
template <int NT>
static void screen_check(const float* a, const float* b, uint64_t n, float fmatch, float fmis, uint64_t* counts) {
  for (uint64_t j = 0; j < n; ++j) {
    SubProf<1, NT> s;
    for (int k = 0; k < 5; ++k) { s.a[0][k] = a[5 * j + k]; s.b[k] = b[5 * j + k]; }
    s.fmatch = fmatch; s.fmis = fmis; s.shift = 0;
    s.screen_setup();
    const int32_t bad = s.screen();
    const int32_t fast = s.sv[0];
    s.prepare();
    if (bad < 0) ++counts[0];
    else if (fast != s.sv[0]) ++counts[1];
  }
}
extern "C" {
// prefix bound of up to 64 / kPrefixLanes pairs in one wave: pair i has profile a1 + a1_off[i] (m[i] columns, row stride
// m[i]) and reference codes a2 + a2_off[i] (n[i] bytes, already encoded with base_code); flags[i] & 1 = reverse complement
int emu_prefix(int K, uint32_t npairs, const float* a1, const uint64_t* a1_off, const uint32_t* m, const uint8_t* a2,
               const uint64_t* a2_off, const uint32_t* n, const uint32_t* flags, int32_t match, int32_t mismatch, int32_t go, int32_t ge,
               int32_t* out, int32_t* err_out) {
  std::vector<PairDesc> d(npairs);
  for (uint32_t i = 0; i < npairs; ++i) {
    d[i] = PairDesc{};
    d[i].a1_off = a1_off[i]; d[i].a2_off = a2_off[i]; d[i].m = m[i]; d[i].n = n[i]; d[i].a1_stride = m[i]; d[i].a2_stride = n[i];
    d[i].flags = flags[i]; d[i].out = i;
  }
  int32_t errw[kErrWords] = {0};
  int32_t& err = errw[0];
  DpArgs a{};
  uint64_t extent = 0;
  for (uint32_t i = 0; i < npairs; ++i) extent = std::max<uint64_t>(extent, a2_off[i] + n[i]);
  const std::vector<uint8_t> padded = padded_codes(a2, extent);  // (the kernel looks ahead of and behind a window, as in the library's buffer)
  a.pairs = d.data(); a.a1 = a1; a.a2 = padded.data() + 128; a.scores = out; a.err = &err;
  a.match = match; a.mismatch = mismatch; a.go = go; a.ge = ge; a.hfree = 1; a.vfree = 0;
  std::vector<uint8_t> special((extent >> 8) + 2, 0);
  for (uint64_t i = 0; i < extent; ++i) if (a2[i] >= 4) special[i >> 8] = 1;
  a.special_blocks = special.data();
  switch (K) {
    case 4: run_prefix_wave<4>(a, npairs); break;
    case 8: run_prefix_wave<8>(a, npairs); break;
    case 15: run_prefix_wave<15>(a, npairs); break;
    case 16: run_prefix_wave<16>(a, npairs); break;
    default: return -1;
  }
  if (err_out) *err_out = err;
  return 0;
}

// origin-tracking sweep of one string x string pair (AlignConfig<true,false>, single pass): score and the two ends
int emu_origin(int K, const uint8_t* a1, uint32_t m, const uint8_t* a2, uint32_t n, uint32_t flags, int32_t match, int32_t mismatch,
               int32_t go, int32_t ge, int32_t* score, uint32_t* ends) {
  PairDesc d{};
  d.m = m; d.n = n; d.a1_stride = m; d.a2_stride = n; d.flags = flags; d.out = 0;
  int32_t errw[kErrWords] = {0};
  int32_t& err = errw[0];
  DpArgs a{};
  a.pairs = &d; a.a1 = a1; a.a2 = a2; a.scores = score; a.err = &err; a.ends = ends;
  a.match = match; a.mismatch = mismatch; a.go = go; a.ge = ge; a.hfree = 1; a.vfree = 0;
  std::vector<uint8_t> codes;
  if (flags & 0x200u) {  // table form (MODE_CQ): a2 as case-sensitive codes
    d.flags &= 0xffu;
    codes = cq_codes(a2, n);
    a.a2 = codes.data() + 128;
    int ncodes = 4;  // the smallest table the columns allow, as the library chooses
    for (uint32_t j = 0; j < n; ++j) ncodes = std::max(ncodes, codes[128 + j] >= 5 ? 6 : codes[128 + j] == 4 ? 5 : 4);
    switch (K) {
      case 4: if (ncodes == 4) run_origin_wave<4, 1, 4>(a); else if (ncodes == 5) run_origin_wave<4, 1, 5>(a); else run_origin_wave<4, 1, 6>(a); break;
      case 8: if (ncodes == 4) run_origin_wave<8, 1, 4>(a); else if (ncodes == 5) run_origin_wave<8, 1, 5>(a); else run_origin_wave<8, 1, 6>(a); break;
      case 12: if (ncodes == 4) run_origin_wave<12, 1, 4>(a); else if (ncodes == 5) run_origin_wave<12, 1, 5>(a); else run_origin_wave<12, 1, 6>(a); break;
      case 15: if (ncodes == 4) run_origin_wave<15, 1, 4>(a); else if (ncodes == 5) run_origin_wave<15, 1, 5>(a); else run_origin_wave<15, 1, 6>(a); break;
      case 16: if (ncodes == 4) run_origin_wave<16, 1, 4>(a); else if (ncodes == 5) run_origin_wave<16, 1, 5>(a); else run_origin_wave<16, 1, 6>(a); break;
      default: return -1;
    }
    return 0;
  }
  switch (K) {
    case 4: run_origin_wave<4>(a); break;
    case 8: run_origin_wave<8>(a); break;
    case 12: run_origin_wave<12>(a); break;
    case 15: run_origin_wave<15>(a); break;
    case 16: run_origin_wave<16>(a); break;
    default: return -1;
  }
  return 0;
}

// checkpointed score pass + band traceback of one pair (single pass: m <= 64*K)
int emu_band(int mode, int K, int narrow, uint32_t B, const void* a1, uint32_t m, uint32_t a1_stride, const void* a2, uint32_t n,
             uint32_t flags, int32_t match, int32_t mismatch, int32_t go, int32_t ge, int32_t hfree, int32_t vfree,
             int32_t* score, uint8_t* ops, uint32_t* ops_len, int32_t* err_out) {
  PairDesc d{};
  d.m = m; d.n = n; d.a1_stride = a1_stride; d.a2_stride = n; d.flags = flags;
  const uint32_t steps = n + 64;
  std::vector<int32_t> ckpt((size_t)(steps / B + 2) * ckpt_fields(K) * 64, 0x7f7f7f7f);
  std::vector<int32_t> lastrow(2 * (size_t)(n + 2), 0x7f7f7f7f);
  std::vector<uint64_t> band((size_t)B * 64 + 64, 0xDEADBEEFDEADBEEFull);
  int32_t errw[kErrWords] = {0};
  int32_t& err = errw[0];
  DpArgs a{};
  a.pairs = &d; a.a1 = a1; a.a2 = a2; a.scores = score; a.err = &err;
  std::vector<uint8_t> codes;
  std::vector<uint8_t> special;
  if (mode == MODE_QP) { codes = padded_codes(a2, n); a.a2 = codes.data() + 128; special = special_blocks_of(codes, n); a.special_blocks = special.data(); }
  a.match = match; a.mismatch = mismatch; a.go = go; a.ge = ge; a.hfree = hfree; a.vfree = vfree;
  a.ckpt = ckpt.data(); a.lastrow = lastrow.data(); a.band = band.data(); a.ckpt_B = B; a.ckpt_narrow = narrow ? 1 : 0;
  uint64_t off = 0;
  WalkArgs wa{};
  wa.pairs = &d; wa.ops = ops; wa.ops_off = &off; wa.ops_len = ops_len; wa.err = &err; wa.npairs = 1; wa.K = K;
  *score = 0x7fffffff;
#define EMU_CK(KK)                                                                                     \
  case KK:                                                                                              \
    if (mode == MODE_CHAR) { if (narrow) run_ckpt_pair<KK, MODE_CHAR, true>(a, wa); else run_ckpt_pair<KK, MODE_CHAR, false>(a, wa); } \
    else { if (narrow) run_ckpt_pair<KK, MODE_QP, true>(a, wa); else run_ckpt_pair<KK, MODE_QP, false>(a, wa); }                        \
    break;
  switch (K) {
    EMU_CK(4) EMU_CK(8) EMU_CK(15) EMU_CK(16)
    default: return -1;
  }
#undef EMU_CK
  if (err_out) *err_out = err;
  return 0;
}

// origin-tracking sweep with profile rows (MODE_QP): a1 = float[6][stride] profile, a2 = reference characters
int emu_origin_qp(int K, const float* a1, uint32_t m, uint32_t stride, const uint8_t* a2, uint32_t n, uint32_t flags, int32_t match,
                  int32_t mismatch, int32_t go, int32_t ge, int32_t* score, uint32_t* ends) {
  PairDesc d{};
  d.m = m; d.n = n; d.a1_stride = stride; d.a2_stride = n; d.flags = flags & 1u; d.out = 0;
  int32_t errw[kErrWords] = {0};
  DpArgs a{};
  a.pairs = &d; a.a1 = a1; a.scores = score; a.err = errw; a.ends = ends;
  a.match = match; a.mismatch = mismatch; a.go = go; a.ge = ge; a.hfree = 1; a.vfree = 0;
  a.qlimit = std::max(std::abs(match), std::abs(mismatch));
  std::vector<uint8_t> codes = padded_codes(a2, n);
  a.a2 = codes.data() + 128;
  switch (K) {
    case 4: run_origin_wave<4, 2, 6>(a); break;
    case 8: run_origin_wave<8, 2, 6>(a); break;
    case 12: run_origin_wave<12, 2, 6>(a); break;
    case 15: run_origin_wave<15, 2, 6>(a); break;
    case 16: run_origin_wave<16, 2, 6>(a); break;
    default: return -1;
  }
  return 0;
}

int emu_screen_check(const float* a, const float* b, uint64_t n, int32_t match, int32_t mismatch, int nt, uint64_t* counts) {
  counts[0] = counts[1] = 0;
  if (nt == 4) screen_check<4>(a, b, n, (float)match, (float)mismatch, counts);
  else screen_check<5>(a, b, n, (float)match, (float)mismatch, counts);
  return 0;
}

// One pair through the kernel bodies.  a1/a2: bytes (CHAR) or float[6][len] (PROFILE side of the mode).
// Returns 0 on success; score/ops (push order)/ops_len are outputs; ops may be null for score-only.
int emu_dp(int needle, int mode, int K, int trace, const void* a1, uint32_t m, uint32_t a1_stride, const void* a2,
           uint32_t n, uint32_t a2_stride, uint32_t flags, int32_t match, int32_t mismatch, int32_t go, int32_t ge,
           int32_t hfree, int32_t vfree, int32_t* score, uint8_t* ops, uint32_t* ops_len, int32_t* err_out) {
  PairDesc d{};
  d.m = m; d.n = n; d.a1_stride = a1_stride; d.a2_stride = a2_stride; d.flags = flags & 0xffffu;
  if (flags & 0x400u) {  // traceback on the diagonal band of half-width flags >> 16 (PAIR_BANDED, as the align pipeline sets it up)
    const int32_t W = (int32_t)(flags >> 16), over = (int32_t)n - (int32_t)m;
    d.flags = (d.flags & ~0x400u) | PAIR_BANDED;
    d.ckpt_off = band_pack(-W - (over < 0 ? -over : 0), W + (over > 0 ? over : 0));
  }
  const uint32_t P = num_passes(m ? m : 1, K);
  std::vector<uint64_t> bits((size_t)P * steps_per_pass(n) * 64 + 64, 0xDEADBEEFDEADBEEFull);
  std::vector<int32_t> scratch(2 * (size_t)(n + 2), 0);
  int32_t errw[kErrWords] = {0};
  int32_t& err = errw[0];
  DpArgs a{};
  a.pairs = &d; a.a1 = a1; a.a2 = a2;
  std::vector<uint8_t> codes;
  std::vector<uint8_t> special;
  if (mode == MODE_QP && !needle) { codes = padded_codes(a2, n); a.a2 = codes.data() + 128; special = special_blocks_of(codes, n); a.special_blocks = special.data(); }
  if (mode == MODE_CQ) { codes = cq_codes(a2, n); a.a2 = codes.data() + 128; special = special_blocks_of(codes, n); a.special_blocks = special.data(); }
  a.bits = bits.data(); a.bits32 = reinterpret_cast<uint32_t*>(bits.data());
  a.scratch = scratch.data(); a.scores = score; a.err = &err;
  a.match = match; a.mismatch = mismatch; a.go = go; a.ge = ge; a.hfree = hfree; a.vfree = vfree;
  *score = 0x7fffffff;
  std::vector<uint8_t> colclass;
  if (flags & 0x200u) {  // profile x profile: screened substitution scores
    a.screen = 1;
    d.flags &= ~0x200u;
    colclass.resize(n + 1);
    for (uint32_t j = 0; j < n; ++j) colclass[j] = (uint8_t)column_class(static_cast<const float*>(a2), a2_stride, j);
    a.colcode = colclass.data();
  }
  if (flags & 0x100u) {  // 16-bit score-only kernel
    d.flags &= 0xffu;
    switch (K) {
      case 4: dispatch_narrow<4>(mode, a); break;
      case 8: dispatch_narrow<8>(mode, a); break;
      case 12: dispatch_narrow<12>(mode, a); break;
      case 15: dispatch_narrow<15>(mode, a); break;
      case 16: dispatch_narrow<16>(mode, a); break;
      default: return -1;
    }
    if (err_out) *err_out = err;
    return 0;
  }
  switch (K) {
    case 4: needle ? dispatch<4, true>(mode, trace, a) : dispatch<4, false>(mode, trace, a); break;
    case 8: needle ? dispatch<8, true>(mode, trace, a) : dispatch<8, false>(mode, trace, a); break;
    case 12: if (needle) return -1; dispatch<12, false>(mode, trace, a); break;
    case 15: if (needle) return -1; dispatch<15, false>(mode, trace, a); break;
    case 16: needle ? dispatch<16, true>(mode, trace, a) : dispatch<16, false>(mode, trace, a); break;
    default: return -1;
  }
  if (trace && ops) {
    uint64_t off = 0;
    WalkArgs wa{};
    wa.pairs = &d; wa.bits = bits.data(); wa.ops = ops; wa.ops_off = &off; wa.ops_len = ops_len; wa.err = &err;
    wa.npairs = 1; wa.K = K;
    if (needle) needle_walk_one(wa, reinterpret_cast<uint32_t*>(bits.data()), 0);
    else {
      // both walkers: the one-lane reference walk and the wave-cooperative one must agree
      gotoh_walk_one(wa, 0);
      std::vector<uint8_t> ref_ops(ops, ops + *ops_len);
      const uint32_t ref_len = *ops_len;
      std::memset(ops, 0, (size_t)m + n);
      WaveShared sh;
      sh.run([&](uint32_t l) { HostWave w{l, &sh}; gotoh_walk_wave<HostWave>(w, wa, 0); });
      if (*ops_len != ref_len || std::memcmp(ops, ref_ops.data(), ref_len) != 0) err |= 0x100;
    }
  }
  if (err_out) *err_out = err;
  return 0;
}

// Up to four pairs through the band kernels (band16.h) as one wave: a1 = concatenated strings (bytes) or float[6][stride] profiles,
// a2 = reference characters; dmin / dmax = the band's diagonals per pair.  kind 0: traceback (scores, ops in push order at
// ops + i * ops_cap, ops_len); kind 1: origin-tracking sweep (scores, ends).
int emu_band16(int K, int kind, int strings, uint32_t npairs, const void* a1, const uint64_t* a1_off, const uint32_t* a1_stride,
               const uint32_t* m, const uint8_t* a2, const uint64_t* a2_off, const uint32_t* n, const uint32_t* flags, const int32_t* dmin,
               const int32_t* dmax, int32_t match, int32_t mismatch, int32_t go, int32_t ge, int32_t hfree, int32_t* scores, uint32_t* ends,
               uint8_t* ops, uint64_t ops_cap, uint32_t* ops_len, int32_t* err_out) {
  const bool quads = K == 44;  // strip height 4 swept by four lanes per pair (band16_body P = 4): up to sixteen pairs
  if (quads) K = 4;
  if (npairs == 0 || npairs > (quads ? 16u : 4u)) return -1;
  const int shift = kTagShift;  // one table for both kinds, as the library builds it
  std::vector<PairDesc> d(npairs);
  std::vector<int16_t> qp;
  std::vector<uint8_t> codes(128, 5);  // (the library's code buffers carry kCodePad = 128 spare bytes on both sides)
  std::vector<uint64_t> ops_off(npairs);
  uint64_t bits_total = 0;
  uint32_t nmax = 0;
  for (uint32_t i = 0; i < npairs; ++i) {
    PairDesc& p = d[i];
    p = PairDesc{};
    p.m = m[i]; p.n = n[i]; p.flags = flags[i] & PAIR_A2_REVCOMP; p.out = i;
    p.ckpt_off = band_pack(dmin[i], dmax[i]);
    const uint32_t stride = b16_table_stride(m[i]);
    p.a1_off = qp.size(); p.a1_stride = stride;
    qp.resize(qp.size() + (size_t)kB16Codes * stride, 0);
    for (uint32_t r = 0; r < m[i]; ++r) {
      int32_t q[kB16Codes];
      b16_table_row(a1, strings != 0, a1_off[i], a1_stride[i], r, match, mismatch, q);
      for (uint32_t b = 0; b < kB16Codes; ++b) qp[p.a1_off + (size_t)b * stride + r] = (int16_t)((uint32_t)q[b] << shift);
    }
    p.a2_off = codes.size() - 128;
    for (uint32_t c = 0; c < n[i]; ++c) codes.push_back((uint8_t)(strings ? cq_code(a2[a2_off[i] + c]) : dp_code(a2[a2_off[i] + c])));
    p.bits_off = bits_total;
    bits_total += (b16_store_words(m[i], n[i], K, dmin[i], dmax[i]) * b16_word_bytes(K) + 7u) & ~7ull;
    ops_off[i] = (uint64_t)i * ops_cap;
    nmax = std::max(nmax, n[i]);
  }
  codes.insert(codes.end(), 128, 5);
  std::vector<uint8_t> bits(bits_total + 64, 0xEE);
  int32_t errw[kErrWords] = {0};
  Band16Args a{};
  a.pairs = d.data(); a.npairs = npairs; a.qp = qp.data(); a.codes = codes.data() + 128; a.bits = bits.data(); a.scores = scores; a.ends = ends;
  a.err = errw; a.go = go; a.ge = ge; a.hfree = hfree; a.code_cap = (nmax + 3u) & ~3u; a.ops = ops; a.ops_off = ops_off.data(); a.ops_len = ops_len;
  WaveShared sh;
  sh.lds.assign((quads ? 16u * b16_packed_row(a.code_cap) : 4u * a.code_cap) + b16_table_bytes(K) + 64, 0);
  if (quads) {
    for (uint32_t i = 0; i < npairs; ++i)
      if (!b16_narrow_ok(dmin[i], dmax[i])) return -1;
    sh.run([&](uint32_t l) { HostWave w{l, &sh}; if (kind == 0) band16_body<HostWave, 4, 0, false, 4>(w, a, 0); else band16_body<HostWave, 4, 1, false, 4>(w, a, 0); });
    if (err_out) *err_out = errw[0];
    return 0;
  }
#define EMU_B16(KK)                                                                                                     \
  case KK:                                                                                                              \
    sh.run([&](uint32_t l) { HostWave w{l, &sh}; if (kind == 0) band16_body<HostWave, KK, 0>(w, a, 0); else band16_body<HostWave, KK, 1>(w, a, 0); }); \
    break;
  switch (K) {
    EMU_B16(4) EMU_B16(8) EMU_B16(12)
    default: return -1;
  }
#undef EMU_B16
  if (err_out) *err_out = errw[0];
  return 0;
}
// The pruned orientation sweep (front.h) of up to four traces as the library runs it: prefix rows 1 .. GLp Kp over all columns (GLp = 8
// or 16 lanes per pair) with the row kept (gotoh_prefix_body, PAIR_KEEP_ROW), front_place, the band kernels below that row (band16_body CONT, strip height Kb),
// front_certify.  a1 = float [6][m] profiles (row stride m), a2 = reference characters.  Per pair: out[0..5] = {vmax, c*, shift,
// ok, score, c_e (window column)}; rows (may be null) receives the kept rows, pair i at rows + i * rows_cap.
int emu_front(int Kp, int GLp, int Kb, uint32_t npairs, const float* a1, const uint64_t* a1_off, const uint32_t* m, const uint8_t* a2,
              const uint64_t* a2_off, const uint32_t* n, const uint32_t* flags, int32_t halfw, int32_t match, int32_t mismatch, int32_t go,
              int32_t ge, int32_t* out, uint32_t* rows, uint64_t rows_cap, int32_t* err_out, int32_t second_bound) {
  if (npairs == 0 || npairs > 4) return -1;
  if (GLp != 8 && GLp != 16) return -1;
  const uint32_t R = (uint32_t)GLp * (uint32_t)Kp;
  uint64_t extent = 0;
  for (uint32_t i = 0; i < npairs; ++i) extent = std::max<uint64_t>(extent, a2_off[i] + n[i]);
  std::vector<uint8_t> raw(extent);
  for (uint64_t i = 0; i < extent; ++i) raw[i] = (uint8_t)dp_code(a2[i]);
  std::vector<uint8_t> codes = padded_codes(raw.data(), extent);
  std::vector<uint8_t> special = special_blocks_of(codes, extent);
  // ---- prefix sweeps, rows kept ----
  std::vector<PairDesc> d(npairs);
  std::vector<int32_t> lastrow;
  for (uint32_t i = 0; i < npairs; ++i) {
    if (m[i] <= R + (uint32_t)Kb) return -2;
    d[i] = PairDesc{};
    d[i].a1_off = a1_off[i]; d[i].a2_off = a2_off[i]; d[i].m = m[i]; d[i].n = n[i]; d[i].a1_stride = m[i]; d[i].a2_stride = n[i];
    d[i].flags = (flags[i] & PAIR_A2_REVCOMP) | PAIR_KEEP_ROW; d[i].out = i;
    d[i].lastrow_off = lastrow.size();
    lastrow.resize(lastrow.size() + n[i] + 1, 0x7fff7fff);
  }
  std::vector<int32_t> pmax(npairs, 0);
  int32_t errw[kErrWords] = {0};
  DpArgs pa{};
  pa.pairs = d.data(); pa.a1 = a1; pa.a2 = codes.data() + 128; pa.scores = pmax.data(); pa.err = errw; pa.lastrow = lastrow.data();
  pa.match = match; pa.mismatch = mismatch; pa.go = go; pa.ge = ge; pa.hfree = 1; pa.vfree = 0; pa.qlimit = 1 << 20;
  pa.special_blocks = special.data();
  switch (Kp * 100 + GLp) {
    case 408: run_prefix_wave<4>(pa, npairs); break;
    case 808: run_prefix_wave<8>(pa, npairs); break;
    case 1508: run_prefix_wave<15>(pa, npairs); break;
    case 416: run_prefix_wave<4, 16>(pa, npairs); break;
    case 816: run_prefix_wave<8, 16>(pa, npairs); break;
    default: return -1;
  }
  // ---- tables of the whole profiles, the rest bound ----
  std::vector<int16_t> qp;
  std::vector<FrontDesc> fd(npairs);
  for (uint32_t i = 0; i < npairs; ++i) {
    const uint32_t stride = b16_table_stride(m[i]);
    const uint64_t off = qp.size();
    qp.resize(qp.size() + (size_t)kB16Codes * stride, 0);
    int32_t rest = 0, rest1 = 0;
    for (uint32_t r = 0; r < m[i]; ++r) {
      int32_t q[kB16Codes];
      b16_table_row(a1, false, a1_off[i], m[i], r, match, mismatch, q);
      int32_t best = INT32_MIN;
      for (uint32_t b = 0; b < kB16Codes; ++b) {
        qp[off + (size_t)b * stride + r] = (int16_t)((uint32_t)q[b] << kTagShift);
        if (b < 5 && q[b] > best) best = q[b];
      }
      if (r >= R) { rest += best > 0 ? best : 0; rest1 += best > -1 ? best : -1; }
    }
    FrontDesc& f = fd[i];
    f = FrontDesc{};
    f.row_off = d[i].lastrow_off; f.a2_off = a2_off[i]; f.tab_off = off + R; f.tab_stride = stride; f.m_rest = m[i] - R; f.n = n[i];
    f.flags = flags[i] & PAIR_A2_REVCOMP; f.out = i; f.R = R; f.rest = rest;
    f.tight = (ge <= -2 && (second_bound & 1)) ? (uint32_t)(rest - rest1) + 1u : 0u;  // (as pipeline.hip fills it)
  }
  const uint32_t* rowp = reinterpret_cast<const uint32_t*>(lastrow.data());
  std::vector<PairDesc> bp(npairs);
  std::vector<FrontOut> fo(npairs);
  for (uint32_t i = 0; i < npairs; ++i) {
    WaveShared sh;
    sh.lds.assign(64, 0);
    sh.run([&](uint32_t l) { HostWave w{l, &sh}; front_place_body(w, fd[i], rowp, go + ge, halfw, &bp[i], &fo[i]); });
  }
  // ---- the band below the kept row ----
  std::vector<int32_t> scores(npairs, 0);
  std::vector<uint32_t> ends(2 * npairs, 0);
  uint32_t nmax = 0;
  for (uint32_t i = 0; i < npairs; ++i) nmax = std::max(nmax, bp[i].n);
  Band16Args a{};
  a.pairs = bp.data(); a.npairs = npairs; a.qp = qp.data(); a.codes = codes.data() + 128; a.scores = scores.data(); a.ends = ends.data();
  a.err = errw; a.go = go; a.ge = ge; a.hfree = 1; a.code_cap = (nmax + 3u) & ~3u; a.row = rowp;
  {
    WaveShared sh;
    const bool quad = (second_bound & 4) != 0;  // ... in the quad form (four lanes per pair, strip height 4, bands of at most 12 diagonals)
    sh.lds.assign(std::max<uint32_t>(4u * a.code_cap + b16_table_bytes(Kb) + 4u * 2u * kB16RowCap * 4u, b16_cont_quad_lds(a.code_cap)) + 64, 0);
    const bool cont16 = (second_bound & 2) != 0;  // the band on the 16-bit cells (band16_cont16_body)
    if (quad) {
      if (Kb != 4 || 2 * halfw + 1 > 12) return -1;
      sh.run([&](uint32_t l) { HostWave w{l, &sh}; band16_cont16_body<HostWave, 4, 4>(w, a, 0); });
    } else
    switch (Kb + (cont16 ? 100 : 0)) {
      case 4: sh.run([&](uint32_t l) { HostWave w{l, &sh}; band16_body<HostWave, 4, 1, true>(w, a, 0); }); break;
      case 8: sh.run([&](uint32_t l) { HostWave w{l, &sh}; band16_body<HostWave, 8, 1, true>(w, a, 0); }); break;
      case 12: sh.run([&](uint32_t l) { HostWave w{l, &sh}; band16_body<HostWave, 12, 1, true>(w, a, 0); }); break;
      case 104: sh.run([&](uint32_t l) { HostWave w{l, &sh}; band16_cont16_body<HostWave, 4>(w, a, 0); }); break;
      case 108: sh.run([&](uint32_t l) { HostWave w{l, &sh}; band16_cont16_body<HostWave, 8>(w, a, 0); }); break;
      case 112: sh.run([&](uint32_t l) { HostWave w{l, &sh}; band16_cont16_body<HostWave, 12>(w, a, 0); }); break;
      default: return -1;
    }
  }
  for (uint32_t i = 0; i < npairs; ++i) {
    WaveShared sh;
    sh.lds.assign(64, 0);
    sh.run([&](uint32_t l) { HostWave w{l, &sh}; front_certify_body(w, fd[i], rowp, go, ge, halfw, scores[i], ends[2 * i + 1], &fo[i]); });
    out[6 * i + 0] = fo[i].vmax; out[6 * i + 1] = (int32_t)fo[i].cstar; out[6 * i + 2] = (int32_t)fo[i].shift; out[6 * i + 3] = (int32_t)fo[i].ok;
    out[6 * i + 4] = scores[i]; out[6 * i + 5] = (int32_t)(ends[2 * i + 1] ? ends[2 * i + 1] + fo[i].shift : 0u);
    if (rows) for (uint32_t c = 0; c <= n[i] && c < rows_cap; ++c) rows[(uint64_t)i * rows_cap + c] = rowp[d[i].lastrow_off + c];
  }
  if (err_out) *err_out = errw[0];
  return 0;
}
// the substitution scores of a profile's rows against the six column codes (b16_table_row): out[r * 6 + b]
int emu_table_rows(const float* a1, uint64_t a1_off, uint32_t a1_stride, uint32_t m, int32_t match, int32_t mismatch, int32_t* out) {
  for (uint32_t r = 0; r < m; ++r) b16_table_row(a1, false, a1_off, a1_stride, r, match, mismatch, out + (size_t)r * kB16Codes);
  return 0;
}
}
