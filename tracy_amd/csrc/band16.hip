// band16.hip -- gfx950 kernels of the band sweeps (band16.h): four pairs per 64-lane workgroup, sixteen lanes per pair; the
// hand-over between strips is a DPP row rotate (row_ror:1).  Plus the kernel that writes the substitution tables they read.
#include <hip/hip_runtime.h>

#include "band16.h"
#include "band16_launch.h"
#include "front.h"

namespace tracyhip {

struct DeviceWave16 {
  __device__ __forceinline__ uint32_t lane() const { return threadIdx.x; }
  // lane j of a row of 16 <- lane j - 1 of the same row, lane 0 <- lane 15 (row_ror:1)
  __device__ __forceinline__ int32_t rot16(int32_t x) const { return __builtin_amdgcn_update_dpp(0, x, 0x121, 0xf, 0xf, false); }
  __device__ __forceinline__ int32_t rot(int32_t x, std::integral_constant<int, 16>) const { return rot16(x); }
  // lane j of a quad <- lane j - 1 of the same quad, lane 0 <- lane 3 (quad_perm:[3,0,1,2])
  __device__ __forceinline__ int32_t rot(int32_t x, std::integral_constant<int, 4>) const { return __builtin_amdgcn_update_dpp(0, x, 0x93, 0xf, 0xf, false); }
  __device__ __forceinline__ uint64_t ballot(bool p) const { return __ballot(p); }
  __device__ __forceinline__ uint32_t bcast(uint32_t x, uint32_t src_lane) const { return (uint32_t)__shfl((int)x, (int)src_lane, 64); }
  __device__ __forceinline__ void sync() const { __syncthreads(); }
  __device__ __forceinline__ void sync_global() const {  // words written by lanes of this wave, read by others of it (dp_kernels.hip)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  __device__ __forceinline__ char* lds() const {
    extern __shared__ __attribute__((aligned(16))) char tracy_smem16[];
    return tracy_smem16;
  }
};

// (register caps were measured and lose: amdgpu_waves_per_eu 3 / 4 for K = 12 / 8 spills 24 / 14 registers inside the blocks)
template <int K, int KIND>
__global__ __launch_bounds__(64) void band16_kernel(Band16Args a) {
  DeviceWave16 w;
  band16_body<DeviceWave16, K, KIND>(w, a, blockIdx.x);
}
// One launch for the three strip heights of a small job: blocks [0, w12) sweep a12's pairs on K = 12 strips, the next w8 a8's on
// K = 8, the rest a4's on K = 4.  A job of 10 000 pairs is 2 500 waves -- fewer than the device holds -- so its launches are as long
// as one wave takes, and three of them in a row take three times that.
template <int KIND>
__global__ __launch_bounds__(64) void band16_multi_kernel(Band16Args a12, uint32_t w12, Band16Args a8, uint32_t w8, Band16Args a4) {
  DeviceWave16 w;
  if (blockIdx.x < w12) band16_body<DeviceWave16, 12, KIND>(w, a12, blockIdx.x);
  else if (blockIdx.x < w12 + w8) band16_body<DeviceWave16, 8, KIND>(w, a8, blockIdx.x - w12);
  else band16_body<DeviceWave16, 4, KIND>(w, a4, blockIdx.x - w12 - w8);
}

// The same with the three jobs' sizes on the device (Band16Args::count; lists laid out by stream.hip's planning kernels): the grid holds
// the worst case, a block finds its job from the counts and the blocks past the last job leave at once.
// (aq: the pairs of the quad form, sixteen to a block; the four jobs share one code_cap and the LDS block is laid out per form)
// (aq: the pairs of the quad form, sixteen to a block; the four jobs share one code_cap and the LDS block is laid out per form)
template <int KIND>
__global__ __launch_bounds__(64) void band16_multi_counted_kernel(Band16Args a12, Band16Args a8, Band16Args a4, Band16Args aq) {
  DeviceWave16 w;
  const uint32_t w12 = (*a12.count + 3u) / 4u, w8 = (*a8.count + 3u) / 4u, w4 = (*a4.count + 3u) / 4u, wq = (*aq.count + 15u) / 16u;
  if (blockIdx.x < w12) band16_body<DeviceWave16, 12, KIND>(w, a12, blockIdx.x);
  else if (blockIdx.x < w12 + w8) band16_body<DeviceWave16, 8, KIND>(w, a8, blockIdx.x - w12);
  else if (blockIdx.x < w12 + w8 + w4) band16_body<DeviceWave16, 4, KIND>(w, a4, blockIdx.x - w12 - w8);
  else if (blockIdx.x < w12 + w8 + w4 + wq) band16_body<DeviceWave16, 4, KIND, false, 4>(w, aq, blockIdx.x - w12 - w8 - w4);
}

// the same without the tallest strips (whose 195 registers hold every workgroup of the kernel above at two waves per SIMD): large jobs
// give strip height 12 -- the shortest list -- a launch of its own
template <int KIND>
__global__ __launch_bounds__(64) void band16_multi3_counted_kernel(Band16Args a8, Band16Args a4, Band16Args aq) {
  DeviceWave16 w;
  const uint32_t w8 = (*a8.count + 3u) / 4u, w4 = (*a4.count + 3u) / 4u, wq = (*aq.count + 15u) / 16u;
  if (blockIdx.x < w8) band16_body<DeviceWave16, 8, KIND>(w, a8, blockIdx.x);
  else if (blockIdx.x < w8 + w4) band16_body<DeviceWave16, 4, KIND>(w, a4, blockIdx.x - w8);
  else if (blockIdx.x < w8 + w4 + wq) band16_body<DeviceWave16, 4, KIND, false, 4>(w, aq, blockIdx.x - w8 - w4);
}

template <int K>
__global__ __launch_bounds__(64) void band16_cont_kernel(Band16Args a) {
  DeviceWave16 w;
  band16_body<DeviceWave16, K, 1, true>(w, a, blockIdx.x);
}

template <int K>
__global__ __launch_bounds__(64) void band16_cont16_kernel(Band16Args a) {
  DeviceWave16 w;
  band16_cont16_body<DeviceWave16, K>(w, a, blockIdx.x);
}

// the narrow first tier of the pruned sweeps: strips of four rows, four lanes per pair
__global__ __launch_bounds__(64) void band16_cont16_quad_kernel(Band16Args a) {
  DeviceWave16 w;
  band16_cont16_body<DeviceWave16, 4, 4>(w, a, blockIdx.x);
}

// front.h: one wave per pair
// prev (or null): the verdicts of an earlier, narrower tier over the same descriptors -- what certified there is an empty slot here
__global__ __launch_bounds__(64) void front_place_kernel(const FrontDesc* __restrict__ desc, const uint32_t* __restrict__ row, int32_t goe, int32_t halfw,
                                                         PairDesc* __restrict__ pairs, FrontOut* __restrict__ fo, const FrontOut* __restrict__ prev,
                                                         const uint32_t* __restrict__ index, const uint32_t* __restrict__ count) {
  DeviceWave16 w;
  // (a later tier's units as a list laid out on the device: the grid holds the worst case, the workgroups past the count leave)
  uint32_t u = blockIdx.x;
  if (index) {
    if (u >= *count) return;
    u = index[u];
  }
  FrontDesc f = desc[u];
  if (prev && prev[u].ok) f.flags |= PAIR_SKIP;
  // (an earlier tier placed the pair on this very row: its maximum and column are taken over, the row is not scanned again -- unless
  // that tier left the slot empty, cstar = 0)
  const FrontOut* known = (prev && !(f.flags & PAIR_SKIP) && prev[u].cstar != 0u) ? prev + u : nullptr;
  front_place_body(w, f, row, goe, halfw, pairs + u, fo + u, known);
}
__global__ __launch_bounds__(64) void front_certify_kernel(const FrontDesc* __restrict__ desc, const uint32_t* __restrict__ row, int32_t go, int32_t ge,
                                                           int32_t halfw, const int32_t* __restrict__ scores, const uint32_t* __restrict__ ends,
                                                           FrontOut* __restrict__ fo, const FrontOut* __restrict__ prev,
                                                           const uint32_t* __restrict__ index, const uint32_t* __restrict__ count) {
  DeviceWave16 w;
  uint32_t u = blockIdx.x;
  if (index) {
    if (u >= *count) return;
    u = index[u];
  }
  const FrontDesc f = desc[u];
  if (prev && prev[u].ok) return;
  front_certify_body(w, f, row, go, ge, halfw, scores[f.out], ends[2 * f.out + 1], fo + u);
}

// Substitution tables: int16 [6 codes][stride] per sequence.  Profile rows: the int of the fp32 chain of align.h:112-117 against the
// one-hot column of base b (onehot_score) for b = A C G T N, 0 for '-' / any other letter (an all-zero column); string rows:
// match / mismatch by byte equality (align.h:96-101), mismatch for a column no row letter can equal.  Rows past the sequence
// hold 0.  Entries are stored << shift (the traceback kernels keep scores x 32).
__global__ __launch_bounds__(256) void b16_table_kernel(const B16TableDesc* __restrict__ desc, const void* __restrict__ a1, int strings,
                                                        int32_t match, int32_t mismatch, int32_t qlimit, int shift, int16_t* __restrict__ out,
                                                        int32_t* __restrict__ err) {
  const B16TableDesc d = desc[blockIdx.x];
  bool overflow = false;
  int32_t qabs = 0;
  for (uint32_t r = threadIdx.x; r < d.stride; r += blockDim.x) {
    const bool real = r < d.m;
    int32_t q[kB16Codes] = {0, 0, 0, 0, 0, 0};
    if (real) {
      b16_table_row(a1, strings != 0, d.a1_off, d.a1_stride, r, match, mismatch, q);
      if (!strings)
        for (uint32_t b = 0; b < 5; ++b) qabs = imax(qabs, q[b] < 0 ? -q[b] : q[b]);
    }
#pragma unroll
    for (uint32_t b = 0; b < kB16Codes; ++b) {
      const int32_t v = (int32_t)((uint32_t)q[b] << shift);
      overflow |= v > 32767 || v < -32768 || q[b] > 32767 || q[b] < -32768;
      out[d.out_off + (uint64_t)b * d.stride + r] = (int16_t)v;
    }
  }
  if (overflow) flag_error(err, 1);
  if (!strings && qabs > qlimit) flag_max(err, 1, qabs);  // un-normalised profile: the host re-checks the value range
}

hipError_t launch_b16_tables(const B16TableDesc* d_desc, uint32_t nseq, const void* a1, bool strings, int32_t match, int32_t mismatch,
                             int32_t qlimit, int shift, int16_t* out, int32_t* err, hipStream_t s) {
  if (nseq == 0) return hipSuccess;
  hipLaunchKernelGGL(b16_table_kernel, dim3(nseq), dim3(256), 0, s, d_desc, a1, strings ? 1 : 0, match, mismatch, qlimit, shift, out, err);
  return hipGetLastError();
}

hipError_t launch_band16(int K, int kind, const Band16Args& a, hipStream_t s) {
  if (a.npairs == 0) return hipSuccess;
  const dim3 grid((a.npairs + 3u) / 4u);
  const uint32_t lds = 4u * a.code_cap + b16_table_bytes(K);
#define TRACY_B16(KK)                                                                                     \
  case KK:                                                                                                \
    if (kind == 0) hipLaunchKernelGGL((band16_kernel<KK, 0>), grid, dim3(64), lds, s, a);                 \
    else hipLaunchKernelGGL((band16_kernel<KK, 1>), grid, dim3(64), lds, s, a);                           \
    break;
  switch (K) {
    TRACY_B16(4) TRACY_B16(8) TRACY_B16(12)
    default: return hipErrorInvalidValue;
  }
#undef TRACY_B16
  return hipGetLastError();
}

hipError_t launch_band16_multi(int kind, const Band16Args& a12, const Band16Args& a8, const Band16Args& a4, hipStream_t s) {
  const uint32_t w12 = (a12.npairs + 3u) / 4u, w8 = (a8.npairs + 3u) / 4u, w4 = (a4.npairs + 3u) / 4u;
  if (w12 + w8 + w4 == 0) return hipSuccess;
  // (the callers give the three jobs one code_cap: the LDS layout is [codes of four pairs][tables])
  const uint32_t lds = 4u * a12.code_cap + b16_table_bytes(12);
  if (kind == 0) hipLaunchKernelGGL((band16_multi_kernel<0>), dim3(w12 + w8 + w4), dim3(64), lds, s, a12, w12, a8, w8, a4);
  else hipLaunchKernelGGL((band16_multi_kernel<1>), dim3(w12 + w8 + w4), dim3(64), lds, s, a12, w12, a8, w8, a4);
  return hipGetLastError();
}

// jobs whose sizes are on the device (a.count != null, a.npairs = the most pairs the job can hold): one launch per strip height for
// large batches (a K = 4 workgroup then asks for its own, smaller LDS block), one for all three below 24 576 pairs
hipError_t B16Fork::create() {
  hipError_t e;
  int least = 0, greatest = 0;  // (numerically: greatest = the highest priority)
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = greatest = 0; }
  const bool flat = getenv("TRACYHIP_NO_STREAM_PRIORITY") != nullptr;
  for (int i = 0; i < kSide; ++i) {
    const int prio = (!flat && i == 3) ? least : 0;
    (void)greatest;
    if ((e = hipStreamCreateWithPriority(&side[i], hipStreamNonBlocking, prio)) != hipSuccess) return e;
    if ((e = hipEventCreateWithFlags(&joined[i], hipEventDisableTiming)) != hipSuccess) return e;
  }
  for (auto& r : ready)
    if ((e = hipEventCreateWithFlags(&r, hipEventDisableTiming)) != hipSuccess) return e;
  return hipEventCreateWithFlags(&forked, hipEventDisableTiming);
}
void B16Fork::destroy() {
  for (int i = 0; i < kSide; ++i) {
    if (side[i]) (void)hipStreamDestroy(side[i]);
    if (joined[i]) (void)hipEventDestroy(joined[i]);
    side[i] = nullptr; joined[i] = nullptr;
  }
  for (auto& r : ready) { if (r) (void)hipEventDestroy(r); r = nullptr; }
  if (forked) (void)hipEventDestroy(forked);
  forked = nullptr;
}

hipError_t launch_band16_counted(int kind, const Band16Args& a12, const Band16Args& a8, const Band16Args& a4, const Band16Args& aq, hipStream_t s,
                                 const B16Fork* fork) {
  uint32_t most = a12.npairs;
  for (uint32_t x : {a8.npairs, a4.npairs, aq.npairs}) most = x > most ? x : most;
  if (most == 0) return hipSuccess;
  hipError_t e;
  // Small jobs: one launch for the four lists (a launch lasts at least as long as one of its waves; 10 000 pairs are fewer waves than
  // the device holds, so a second launch costs more than the registers of the tallest strips cost the others: 1.25 / 1.45 ms for the
  // final alignments of 10 000 traces).  Large jobs: strip height 12 -- 195 registers, the shortest list -- gets a launch of its
  // own beside the other three (side stream), which then run three waves per SIMD instead of two.  Measured per decompose step of
  // 12 500 / 25 000 / 100 000 traces: one launch 20.0 / 35.9-36.8 / 131.8-132.0 ms, three + one 20.8 (two stages of it) / 35.5-35.7 /
  // 129.9-130.6 ms, four launches side by side 20.8 / 37.3 / 132.9 ms.
#ifndef TRACY_B16_MULTI_MAX
#define TRACY_B16_MULTI_MAX 49152u
#endif
  if (most <= TRACY_B16_MULTI_MAX) {
    const uint32_t lds16 = 4u * a12.code_cap + b16_table_bytes(12), ldsq = aq.npairs ? b16_quad_lds(aq.code_cap) : 0u;
    const uint32_t lds = lds16 > ldsq ? lds16 : ldsq;
    const dim3 grid((most + 3u) / 4u + 4u);  // (the four jobs together hold at most `most` pairs: every pair is in one of them)
    if (kind == 0) hipLaunchKernelGGL((band16_multi_counted_kernel<0>), grid, dim3(64), lds, s, a12, a8, a4, aq);
    else hipLaunchKernelGGL((band16_multi_counted_kernel<1>), grid, dim3(64), lds, s, a12, a8, a4, aq);
    return hipGetLastError();
  }
  const uint32_t lds16 = 4u * a8.code_cap + b16_table_bytes(8), ldsq = aq.npairs ? b16_quad_lds(aq.code_cap) : 0u;
  const uint32_t lds = lds16 > ldsq ? lds16 : ldsq;
  const dim3 grid((most + 3u) / 4u + 3u);
  if (fork) {
    if ((e = hipEventRecord(fork->forked, s)) != hipSuccess) return e;
    if ((e = hipStreamWaitEvent(fork->side[2], fork->forked, 0)) != hipSuccess) return e;
  }
  if (kind == 0) hipLaunchKernelGGL((band16_multi3_counted_kernel<0>), grid, dim3(64), lds, s, a8, a4, aq);
  else hipLaunchKernelGGL((band16_multi3_counted_kernel<1>), grid, dim3(64), lds, s, a8, a4, aq);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  if ((e = launch_band16(12, kind, a12, fork ? fork->side[2] : s)) != hipSuccess) return e;
  if (fork) {
    if ((e = hipEventRecord(fork->joined[2], fork->side[2])) != hipSuccess) return e;
    if ((e = hipStreamWaitEvent(s, fork->joined[2], 0)) != hipSuccess) return e;
  }
  return hipSuccess;
}

hipError_t launch_band16_cont_quad(const Band16Args& a, hipStream_t s) {
  if (a.npairs == 0) return hipSuccess;
  hipLaunchKernelGGL(band16_cont16_quad_kernel, dim3((a.npairs + 15u) / 16u), dim3(64), b16_cont_quad_lds(a.code_cap), s, a);
  return hipGetLastError();
}

hipError_t launch_band16_cont(int K, const Band16Args& a, hipStream_t s, bool narrow) {
  if (a.npairs == 0) return hipSuccess;
  const dim3 grid((a.npairs + 3u) / 4u);
  const uint32_t lds = narrow ? b16_cont16_lds(K, a.code_cap) : 4u * a.code_cap + b16_table_bytes(K) + 4u * 2u * kB16RowCap * 4u;
  if (narrow) {
    switch (K) {
      case 12: hipLaunchKernelGGL((band16_cont16_kernel<12>), grid, dim3(64), lds, s, a); break;
      case 8: hipLaunchKernelGGL((band16_cont16_kernel<8>), grid, dim3(64), lds, s, a); break;
      case 4: hipLaunchKernelGGL((band16_cont16_kernel<4>), grid, dim3(64), lds, s, a); break;
      default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
  }
  switch (K) {
    case 12: hipLaunchKernelGGL((band16_cont_kernel<12>), grid, dim3(64), lds, s, a); break;
    case 8: hipLaunchKernelGGL((band16_cont_kernel<8>), grid, dim3(64), lds, s, a); break;
    case 4: hipLaunchKernelGGL((band16_cont_kernel<4>), grid, dim3(64), lds, s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_front_place(const FrontDesc* d_desc, uint32_t n, const uint32_t* row, int32_t goe, int32_t halfw, PairDesc* d_pairs, FrontOut* d_fo,
                              hipStream_t s, const FrontOut* d_prev, const uint32_t* d_index, const uint32_t* d_count) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(front_place_kernel, dim3(n), dim3(64), 0, s, d_desc, row, goe, halfw, d_pairs, d_fo, d_prev, d_index, d_count);
  return hipGetLastError();
}

hipError_t launch_front_certify(const FrontDesc* d_desc, uint32_t n, const uint32_t* row, int32_t go, int32_t ge, int32_t halfw, const int32_t* d_scores,
                                const uint32_t* d_ends, FrontOut* d_fo, hipStream_t s, const FrontOut* d_prev, const uint32_t* d_index,
                                const uint32_t* d_count) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(front_certify_kernel, dim3(n), dim3(64), 0, s, d_desc, row, go, ge, halfw, d_scores, d_ends, d_fo, d_prev, d_index, d_count);
  return hipGetLastError();
}

}  // namespace tracyhip
