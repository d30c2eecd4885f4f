// launch.h -- host-visible launcher declarations (kernels live in dp_kernels.hip).
#ifndef TRACY_AMD_LAUNCH_H
#define TRACY_AMD_LAUNCH_H

#include <hip/hip_runtime.h>

#include "dp_kernels.h"

namespace tracyhip {

struct RowsArgs {
  const PairDesc* pairs;
  const void* a1;
  const void* a2;
  int32_t a1_profile, a2_profile;
  int32_t a2_revcomp_flag;  // honour PairDesc::flags & PAIR_A2_REVCOMP when reading a2 characters
  int32_t a2_onehot;  // a2 is a string standing for its one-hot profile: show _profileConsChar of that profile
  const uint8_t* ops;
  const uint64_t* ops_off;  // indexed by PairDesc::out
  const uint32_t* ops_len;
  uint8_t* rows0;
  uint8_t* rows1;
  uint32_t npairs;
  // or null: int32 per PairDesc::out with this stride in int32s -- non-zero: nobody will read the CHARACTERS of the pair's row 0, only
  // where its gaps are (`tracy decompose`: a trace with a shift skips findHomozygousBreakpoint, decomposeAlleles looks at row0 != '-'):
  // 'N' stands for every base and the six profile reads per column are left out
  const int32_t* row0_gaps_only;
  uint32_t row0_gaps_only_stride;
};

// narrow: use the 16-bit score-only kernel (caller has checked the value range, see narrow_ok)
// rawtab (trace, MODE_QP): the table holds unshifted scores (scorings whose entries x 32 leave int16)
hipError_t launch_gotoh(int mode, int K, bool trace, bool narrow, const DpArgs& a, uint32_t npairs, hipStream_t s, bool rawtab = false);
// profile x profile with the substitution-term count fixed per launch: row4_zero = every pair of the launch carries PAIR_ROW4_ZERO
// arith16 (score only): 16-bit cells -- the caller has checked the value range (arith16_ok)
hipError_t launch_gotoh_front_prefix_cq(const DpArgs& a, uint32_t npairs, hipStream_t s);
hipError_t launch_gotoh_prof(int K, bool trace, bool row4_zero, bool arith16, const DpArgs& a, uint32_t npairs, hipStream_t s);
// checkpointed score pass / band traceback (single-pass problems, MODE_CHAR or MODE_QP)
hipError_t launch_gotoh_ckpt(int mode, int K, bool narrow, const DpArgs& a, uint32_t npairs, hipStream_t s);
hipError_t launch_band_trace(int mode, int K, const DpArgs& a, const WalkArgs& wa, uint32_t npairs, hipStream_t s);
// prefix bound (max of H, F over row kPrefixLanes*K) of profile x code pairs, AlignConfig<true,false>, 16-bit domain
hipError_t launch_gotoh_origin(int K, int table, int codes, const DpArgs& a, uint32_t npairs, hipStream_t s);  // table: 0 strings by byte compare, 1 MODE_CQ, 2 MODE_QP (profile rows); codes (MODE_CQ): 4 = columns of A C G T only, 5 = with N, 6 = any
hipError_t launch_gotoh_ckpt_prefix(int K, const DpArgs& full, uint32_t nfull, const DpArgs& pre, uint32_t npre, hipStream_t s);
hipError_t launch_gotoh_prefix(int K, const DpArgs& a, uint32_t npairs, hipStream_t s);
hipError_t launch_gotoh_ckpt_front(int K, const DpArgs& full, uint32_t nfull, const DpArgs& pre, uint32_t npre, hipStream_t s);
hipError_t launch_needle(int mode, int K, bool trace, const DpArgs& a, uint32_t npairs, hipStream_t s);
hipError_t launch_gotoh_walk(const WalkArgs& a, hipStream_t s);
hipError_t launch_needle_walk(const WalkArgs& a, const uint32_t* bits32, hipStream_t s);
hipError_t launch_alignment_rows(const RowsArgs& a, hipStream_t s);

}  // namespace tracyhip
#endif
