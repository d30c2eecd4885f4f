// pack.hip -- tracyhip_pack_ragged: the used parts of fixed-stride result regions (traceback strings at ops_offset[i] = i * cap,
// rows of a decomposition table, rewritten basecalls) back to back, in trace order.  What a rank does to its variable-length results
// before the second half of the final gather ("... followed by a variable-length gather of op strings / decomposition
// tables"): HBM-bound byte work, one pass -- a scan of the lengths by one workgroup, then one workgroup per region copying its bytes.
#include <hip/hip_runtime.h>

#include <cstring>

#include "../../include/tracy_hip.h"
#include "capi_internal.h"

using namespace tracyhip;

#define HIP_TRY(expr)                                                                               \
  do {                                                                                              \
    hipError_t _e = (expr);                                                                         \
    if (_e != hipSuccess)                                                                           \
      return set_error(_e == hipErrorOutOfMemory ? TRACYHIP_ERR_OOM : TRACYHIP_ERR_HIP, "%s failed: %s (%s:%d)", \
                       #expr, hipGetErrorString(_e), __FILE__, __LINE__);                           \
  } while (0)

namespace {
constexpr uint32_t kScanThreads = 1024;

// exclusive scan of the regions' byte counts: off[i] = bytes before region i, off[n] = total
__device__ __forceinline__ unsigned long long region_bytes(const uint32_t* __restrict__ lens, uint32_t lens_stride, uint32_t i, uint32_t elem, unsigned long long clamp) {
  const unsigned long long b = (unsigned long long)lens[(size_t)i * lens_stride] * elem;
  return b < clamp ? b : clamp;  // (a strided region holds at most its stride, whatever its length word says)
}
__global__ __launch_bounds__(kScanThreads) void pack_scan_kernel(const uint32_t* __restrict__ lens, uint32_t lens_stride, uint32_t n, uint32_t elem, unsigned long long clamp,
                                                                 unsigned long long* __restrict__ off) {
  __shared__ unsigned long long s[kScanThreads];
  const uint32_t tid = threadIdx.x;
  const uint32_t per = (n + kScanThreads - 1) / kScanThreads;
  const uint32_t lo = (uint64_t)tid * per < n ? tid * per : n, hi = lo + per < n ? lo + per : n;
  unsigned long long mine = 0;
  for (uint32_t i = lo; i < hi; ++i) mine += region_bytes(lens, lens_stride, i, elem, clamp);
  s[tid] = mine;
  __syncthreads();
  for (uint32_t d = 1; d < kScanThreads; d <<= 1) {
    const unsigned long long v = tid >= d ? s[tid - d] : 0ull;
    __syncthreads();
    s[tid] += v;
    __syncthreads();
  }
  unsigned long long at = s[tid] - mine;
  for (uint32_t i = lo; i < hi; ++i) {
    off[i] = at;
    at += region_bytes(lens, lens_stride, i, elem, clamp);
  }
  if (tid == kScanThreads - 1) off[n] = s[tid];
}

// region i: bytes [i * stride, i * stride + len_i) of src (or from src_off[i]) to dst + off[i].  Four regions per workgroup of 256: one
// wave each.  Sixteen bytes per lane and step where source and destination are aligned alike, bytes otherwise (a wave's 64 bytes are
// one request either way).
__global__ __launch_bounds__(256) void pack_copy_kernel(const uint8_t* __restrict__ src, unsigned long long stride, const unsigned long long* __restrict__ src_off,
                                                        const uint32_t* __restrict__ lens, uint32_t lens_stride, uint32_t n, uint32_t elem, unsigned long long clamp,
                                                        const unsigned long long* __restrict__ off, uint8_t* __restrict__ dst) {
  const uint32_t i = blockIdx.x * 4u + (threadIdx.x >> 6), L = threadIdx.x & 63u;
  if (i >= n) return;
  const unsigned long long bytes = region_bytes(lens, lens_stride, i, elem, clamp);
  const uint8_t* s = src + (src_off ? src_off[i] : (unsigned long long)i * stride);
  uint8_t* d = dst + off[i];
  unsigned long long at = 0;
  if (((reinterpret_cast<uintptr_t>(s) ^ reinterpret_cast<uintptr_t>(d)) & 15u) == 0) {
    const unsigned long long head = (16u - (reinterpret_cast<uintptr_t>(s) & 15u)) & 15u;
    const unsigned long long h = head < bytes ? head : bytes;
    if (L < h) d[L] = s[L];
    at = h;
    const unsigned long long vecs = (bytes - at) >> 4;
    const uint4* sv = reinterpret_cast<const uint4*>(s + at);
    uint4* dv = reinterpret_cast<uint4*>(d + at);
    for (unsigned long long v = L; v < vecs; v += 64) dv[v] = sv[v];
    at += vecs << 4;
  }
  for (unsigned long long b = at + L; b < bytes; b += 64) d[b] = s[b];
}
}  // namespace

extern "C" {

int tracyhip_pack_ragged(tracyhip_ctx* ctx, const void* src, uint64_t stride_bytes, const uint64_t* src_offset, uint32_t elem_bytes, const uint32_t* lens,
                         uint32_t lens_stride, uint32_t n, void* dst, uint64_t dst_cap, uint64_t* total_bytes) {
  int rc = ctx_begin(ctx);
  if (rc) return rc;
  if (!total_bytes || (n && (!src || !lens)) || elem_bytes == 0 || lens_stride == 0) return set_error(TRACYHIP_ERR_ARG, "tracyhip_pack_ragged: null pointer or zero element size");
  *total_bytes = 0;
  if (n == 0) return TRACYHIP_OK;
  hipStream_t st = ctx->stream;
  DevBuf& scr = ctx->d_tmp[7];
  const size_t off_bytes = sizeof(unsigned long long) * ((size_t)n + 1);
  HIP_TRY(scr.ensure(off_bytes + (src_offset ? sizeof(unsigned long long) * (size_t)n : 0)));
  unsigned long long* d_off = static_cast<unsigned long long*>(scr.p);
  unsigned long long* d_soff = nullptr;
  if (src_offset) {  // (a host array, as every offset array of the ABI)
    HIP_TRY(ctx->h_off.ensure(sizeof(uint64_t) * (size_t)n));
    std::memcpy(ctx->h_off.p, src_offset, sizeof(uint64_t) * (size_t)n);
    d_soff = d_off + n + 1;
    HIP_TRY(hipMemcpyAsync(d_soff, ctx->h_off.p, sizeof(uint64_t) * (size_t)n, hipMemcpyHostToDevice, st));
  }
  const unsigned long long clamp = src_offset ? ~0ull : (unsigned long long)stride_bytes;
  hipLaunchKernelGGL(pack_scan_kernel, dim3(1), dim3(kScanThreads), 0, st, lens, lens_stride, n, elem_bytes, clamp, d_off);
  HIP_TRY(hipGetLastError());
  HIP_TRY(ctx->h_res.ensure(sizeof(unsigned long long)));
  unsigned long long* h_tot = static_cast<unsigned long long*>(ctx->h_res.p);
  HIP_TRY(hipMemcpyAsync(h_tot, d_off + n, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  if (dst) {
    // a region holds at most its stride: with strided regions and a capacity for all of them the copy is queued at once; otherwise the
    // total is read first
    const bool safe = !src_offset && dst_cap / n >= stride_bytes;
    if (!safe) {
      HIP_TRY(ctx_sync(ctx));
      if (*h_tot > dst_cap) {
        *total_bytes = *h_tot;
        return set_error(TRACYHIP_ERR_ARG, "tracyhip_pack_ragged: %llu bytes to pack, capacity %llu", *h_tot, (unsigned long long)dst_cap);
      }
    }
    hipLaunchKernelGGL(pack_copy_kernel, dim3((n + 3) / 4), dim3(256), 0, st, static_cast<const uint8_t*>(src), (unsigned long long)stride_bytes, d_soff, lens, lens_stride, n,
                       elem_bytes, clamp, d_off, static_cast<uint8_t*>(dst));
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(ctx_sync(ctx));
  *total_bytes = *h_tot;
  return TRACYHIP_OK;
}

}  // extern "C"
