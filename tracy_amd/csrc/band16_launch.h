// band16_launch.h -- host-visible side of the band kernels (band16.hip)
#ifndef TRACY_AMD_BAND16_LAUNCH_H
#define TRACY_AMD_BAND16_LAUNCH_H

#include <hip/hip_runtime.h>

#include "band16.h"

namespace tracyhip {

struct B16TableDesc {
  uint64_t a1_off;     // source: bytes (strings) or floats (&p[0][first row]) into the a1 payload
  uint64_t out_off;    // destination: int16 units into the table buffer
  uint32_t a1_stride;  // profiles: distance between profile rows k
  uint32_t m;          // rows of the sequence
  uint32_t stride;     // rows per code of its table (b16_table_stride(m))
  uint32_t pad;
};

hipError_t launch_b16_tables(const B16TableDesc* d_desc, uint32_t nseq, const void* a1, bool strings, int32_t match, int32_t mismatch,
                             int32_t qlimit, int shift, int16_t* out, int32_t* err, hipStream_t s);
// kind 0: traceback words + walk; kind 1: origin-tracking sweep
hipError_t launch_band16(int K, int kind, const Band16Args& a, hipStream_t s);
// the three strip heights in one launch (small jobs); the jobs share one code_cap
hipError_t launch_band16_multi(int kind, const Band16Args& a12, const Band16Args& a8, const Band16Args& a4, hipStream_t s);
// the three strip heights of a job whose lists and sizes are on the device (Band16Args::index / count; npairs = capacity of each list)
// Side streams of a context: launches that do not depend on each other run side by side -- the voted strand's chain beside the other
// strand's sweeps, the tallest strips of a band stage beside the other lists, allelicFraction beside the allele stages (stream.hip).
// `forked` makes a side stream wait for what the call's stream has queued, `joined[i]` the call's stream for side stream i.
// side[3] is a low-priority stream where the device has priorities (allelicFraction fills what the allele stages leave idle; the fills and
// copies of the call's stream do not queue behind its hundred thousand waves).  A high priority for side[0] was measured: the voted strand's
// prefixes finish in 10 ms instead of 20, but the band tiers behind them ask for 15-20 KB of LDS per workgroup and find no room while the
// full sweeps' 7.5 KB workgroups fill the CUs, whatever the priority -- they run when the sweeps drain, as before.
struct B16Fork {
  static constexpr int kSide = 4;  // three for the lists of a band stage (and the voted strand's chain), one for work that runs beside whole stages
  hipStream_t side[kSide] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t forked = nullptr, joined[kSide] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ready[2] = {nullptr, nullptr};  // points of the call's stream a side stream waits for later on (recorded once, waited for after more has been queued)
  hipError_t create();
  void destroy();
};
// aq: the pairs of narrow bands (b16_narrow_ok) swept four lanes to a pair; aq.npairs = 0 when the launch's code_cap leaves no room (b16_quad_lds)
hipError_t launch_band16_counted(int kind, const Band16Args& a12, const Band16Args& a8, const Band16Args& a4, const Band16Args& aq, hipStream_t s,
                                 const B16Fork* fork = nullptr);
// the sweep below a stored prefix row (Band16Args::row; K = 8 or 12), and the two kernels of front.h around it
// narrow: every DP value of the launch fits int16 (narrow_ok for the tallest pair, rows above the stored one included): the 16-bit cells
hipError_t launch_band16_cont(int K, const Band16Args& a, hipStream_t s, bool narrow = false);
// ... on 16-bit cells in the quad form (strip height 4, bands of at most 12 diagonals: b16_narrow_ok)
hipError_t launch_band16_cont_quad(const Band16Args& a, hipStream_t s);
struct FrontDesc;
struct FrontOut;
// d_prev (or null): verdicts of an earlier tier over the same descriptors; what certified there is skipped
// d_index / d_count (or null): workgroup i takes unit d_index[i], and only *d_count of them do (a list laid out on the device, n = its
// worst case); results stay in the units' own slots
hipError_t launch_front_place(const FrontDesc* d_desc, uint32_t n, const uint32_t* row, int32_t goe, int32_t halfw, PairDesc* d_pairs, FrontOut* d_fo,
                              hipStream_t s, const FrontOut* d_prev = nullptr, const uint32_t* d_index = nullptr, const uint32_t* d_count = nullptr);
hipError_t launch_front_certify(const FrontDesc* d_desc, uint32_t n, const uint32_t* row, int32_t go, int32_t ge, int32_t halfw, const int32_t* d_scores,
                                const uint32_t* d_ends, FrontOut* d_fo, hipStream_t s, const FrontOut* d_prev = nullptr, const uint32_t* d_index = nullptr,
                                const uint32_t* d_count = nullptr);

}  // namespace tracyhip
#endif
