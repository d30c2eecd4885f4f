// pipe_internal.h -- what pipeline.hip (the pipelines planned by the host between launches) and stream.hip (the same stage orders
// planned on the device, one host synchronisation per call) know of each other.
#ifndef TRACY_AMD_PIPE_INTERNAL_H
#define TRACY_AMD_PIPE_INTERNAL_H

#include "capi_internal.h"

namespace tracyhip {

// internal status of stream_align / stream_decompose: the batch is outside what the stream-ordered pass is built for (an option
// selects another tier, caller-oriented windows, a wildtype-trace reference, scorings or shapes beyond the 16-bit / band forms) or
// a launch reported values outside its proven range; nothing was written, the host-planned pipeline takes the whole call
constexpr int kStreamNo = 3;

// sage.h:191-311 / indigo.h:190-388 for one context (no lanes), planned by the host between launches: every tier, every fallback
int align_traces_legacy(tracyhip_ctx* ctx, const tracyhip_align_job* job, const tracyhip_params* prm, int mem, const tracyhip_align_result* out);
int decompose_traces_legacy(tracyhip_ctx* ctx, const tracyhip_decompose_job* job, const tracyhip_params* prm, int mem,
                            const tracyhip_decompose_result* out);
// the same stage orders with the planning on the device; traces whose tier the device cannot give them go through the functions
// above afterwards.  TRACYHIP_OK, an error, or kStreamNo.
int stream_align(tracyhip_ctx* ctx, const tracyhip_align_job* job, const tracyhip_params* prm, int mem, const tracyhip_align_result* out);
int stream_decompose(tracyhip_ctx* ctx, const tracyhip_decompose_job* job, const tracyhip_params* prm, int mem,
                     const tracyhip_decompose_result* out);

}  // namespace tracyhip
#endif
