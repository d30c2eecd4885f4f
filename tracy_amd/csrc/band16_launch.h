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

}  // namespace tracyhip
#endif
