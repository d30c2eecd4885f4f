// decompose_launch.h -- device-level launchers of the deconvolution kernels (device pointers only);
// shared by the per-function C-ABI entry points and the tracyhip_decompose_traces pipeline.
#ifndef TRACY_AMD_DECOMPOSE_LAUNCH_H
#define TRACY_AMD_DECOMPOSE_LAUNCH_H

#include "capi_internal.h"
#include "decompose_kernels.h"

namespace tracyhip {

struct BpDesc { uint64_t off; uint32_t stride; uint32_t ncol; };     // profile view: p[k][j] at off + k*stride + j
struct RowsDesc { uint64_t off; uint32_t L; uint32_t pad; };
struct BcDesc { uint64_t sig_off; uint64_t bc_off; uint32_t nsamples; uint32_t nbc; };

int launch_breakpoint(tracyhip_ctx* ctx, const BpDesc* d_desc, uint32_t n, uint32_t maxcol, const float* d_prof, BreakpointOut* d_out);
int launch_homozygous(tracyhip_ctx* ctx, const RowsDesc* d_desc, const uint8_t* d_rows0, const uint8_t* d_rows1, uint32_t n,
                      BreakpointOut* d_bps, int32_t* d_status, const uint32_t* d_lens = nullptr);  // d_lens: overrides RowsDesc::L
// work_cells / work_bytes: what the kernel timers (tracyhip_timing_get) account for the launch -- alignment columns walked and
// algorithmic bytes (alignment rows + basecalls read, basecalls rewritten); 0 = not accounted
// maxbc: the longest trace (basecalls) of the batch; it and prm.maxindel pick the size class of the LDS-resident scan state
int decompose_limits(int32_t maxindel, uint32_t maxbc);  // TRACYHIP_OK, or ERR_RANGE beyond the larger size class
int launch_decompose(tracyhip_ctx* ctx, const DecompArgs& a, const BreakpointOut* d_bps, uint32_t maxbc, uint64_t work_cells = 0, uint64_t work_bytes = 0);
// the peak table (decompose_kernels.hip): d_peaks[4 * (bc_off + i) + k] = channel k at bcPos[i]; built from the chromatogram when the caller
// passes none (tracyhip_basecalls::peaks)
int launch_peaks(tracyhip_ctx* ctx, const BcDesc* d_desc, uint32_t n, uint32_t maxbc, const int32_t* d_sig, const int32_t* d_pos, int32_t* d_peaks);
int launch_secdecomp(tracyhip_ctx* ctx, const BcDesc* d_desc, uint32_t n, uint32_t maxbc, const int32_t* d_peaks, const uint8_t* d_pri, const uint8_t* d_sec,
                     uint8_t* d_out);
int launch_allelic_fraction(tracyhip_ctx* ctx, const BcDesc* d_desc, uint32_t n, uint32_t maxbc, const int32_t* d_peaks, const uint8_t* d_pri, const uint8_t* d_sec,
                            uint32_t trim_left, uint32_t trim_right, double* d_out, uint64_t work_bytes = 0, uint64_t bext = 0);  // bext: extent of the basecall arrays (max of
                            // bc_off + nbc; 0 = not known): with it the two-launch form runs (af_prepare_kernel / af_search_kernel, scratch of 36 bext bytes)

}  // namespace tracyhip
#endif
