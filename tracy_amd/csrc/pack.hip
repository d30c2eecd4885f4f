// pack.hip -- tracyhip_pack_ragged / tracyhip_pack_ragged_multi: the used parts of fixed-stride result regions (traceback strings at
// ops_offset[i] = i * cap, rows of a decomposition table, rewritten basecalls) back to back, in trace order.  What a rank does to its
// variable-length results before the second half of the final gather ("... followed by a variable-length gather of op
// strings / decomposition tables"): HBM-bound byte work -- a scan of the lengths, then one wave per region copying its bytes, for every payload kind of a
// call in ONE scan, ONE copy launch and ONE synchronisation.
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
constexpr uint32_t kMaxKinds = 16;

struct PackKind {  // one payload kind: region i = bytes [i * stride, i * stride + min(len_i * elem, stride)) of src
  const uint8_t* src;
  unsigned long long stride;
  const uint32_t* lens;
  uint32_t lens_stride, elem;
};
struct PackArgs {
  PackKind k[kMaxKinds];
  uint32_t nkinds, n;
};

__device__ __forceinline__ unsigned long long region_bytes(const PackKind& k, uint32_t i) {
  const unsigned long long b = (unsigned long long)k.lens[(size_t)i * k.lens_stride] * k.elem;
  return b < k.stride ? b : k.stride;  // (a strided region holds at most its stride, whatever its length word says)
}

// Exclusive scan over the nkinds * n regions in output order (kind-major: all regions of kind 0, then kind 1, ...), in three launches:
// (1) one workgroup per 256 regions of a kind (grid.y = the kind: no division by n; a kind's regions are padded to whole workgroups):
//     the bytes of its regions, their exclusive scan within the workgroup (local[]) and the workgroup's total (bsum[]);
// (2) one workgroup scans the totals in place (bsum[w] = bytes before workgroup w) and writes the bytes up to the end of every kind;
// (3) the copy kernel adds the two.
constexpr uint32_t kScanBlock = 256;
__global__ __launch_bounds__(kScanBlock) void pack_local_kernel(PackArgs a, unsigned long long* __restrict__ local, unsigned long long* __restrict__ bsum) {
  __shared__ unsigned long long s[kScanBlock];
  const uint32_t tid = threadIdx.x, i = blockIdx.x * kScanBlock + tid;
  const unsigned long long mine = i < a.n ? region_bytes(a.k[blockIdx.y], i) : 0ull;
  s[tid] = mine;
  __syncthreads();
  for (uint32_t d = 1; d < kScanBlock; d <<= 1) {
    const unsigned long long x = tid >= d ? s[tid - d] : 0ull;
    __syncthreads();
    s[tid] += x;
    __syncthreads();
  }
  const size_t w = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
  local[w * kScanBlock + tid] = s[tid] - mine;
  if (tid == kScanBlock - 1) bsum[w] = s[tid];
}
__global__ __launch_bounds__(kScanThreads) void pack_top_kernel(unsigned long long* __restrict__ bsum, uint32_t nw, uint32_t w_per_kind, uint32_t nkinds,
                                                                unsigned long long* __restrict__ kind_end) {
  __shared__ unsigned long long s[kScanThreads];
  const uint32_t tid = threadIdx.x;
  const uint32_t per = (nw + kScanThreads - 1) / kScanThreads;
  const uint32_t lo = (uint64_t)tid * per < nw ? tid * per : nw, hi = lo + per < nw ? lo + per : nw;
  unsigned long long mine = 0;
  for (uint32_t w = lo; w < hi; ++w) mine += bsum[w];
  s[tid] = mine;
  __syncthreads();
  for (uint32_t d = 1; d < kScanThreads; d <<= 1) {
    const unsigned long long x = tid >= d ? s[tid - d] : 0ull;
    __syncthreads();
    s[tid] += x;
    __syncthreads();
  }
  unsigned long long at = s[tid] - mine;
  for (uint32_t w = lo; w < hi; ++w) {
    const unsigned long long x = bsum[w];
    bsum[w] = at;
    at += x;
    if ((w + 1u) % w_per_kind == 0u) kind_end[w / w_per_kind] = at;
  }
}

// One wave per region (four to a workgroup).  The destination is written in aligned dwords: a lane's dword is cut out of the two
// aligned source dwords that hold it (v_alignbyte), so a wave moves 256 bytes per step whatever the two alignments are; the bytes
// before the first and behind the last aligned dword of the destination go one by one.
__global__ __launch_bounds__(256) void pack_copy_kernel(PackArgs a, const unsigned long long* __restrict__ local, const unsigned long long* __restrict__ bsum,
                                                        uint32_t w_per_kind, uint8_t* __restrict__ dst) {
  const uint32_t i = blockIdx.x * 4u + (threadIdx.x >> 6), L = threadIdx.x & 63u;  // (grid.y = the kind)
  if (i >= a.n) return;
  const PackKind& k = a.k[blockIdx.y];
  const unsigned long long bytes = region_bytes(k, i);
  if (bytes == 0) return;
  const size_t w = (size_t)blockIdx.y * w_per_kind + i / kScanBlock;
  const uint8_t* s = k.src + (unsigned long long)i * k.stride;
  uint8_t* d = dst + bsum[w] + local[w * kScanBlock + i % kScanBlock];
  const unsigned long long head0 = (4u - (reinterpret_cast<uintptr_t>(d) & 3u)) & 3u;
  const unsigned long long head = head0 < bytes ? head0 : bytes;
  if (L < head) d[L] = s[L];
  const unsigned long long words = (bytes - head) >> 2;
  const uint8_t* sb = s + head;
  const uint32_t r = (uint32_t)(reinterpret_cast<uintptr_t>(sb) & 3u);
  const uint32_t* sw = reinterpret_cast<const uint32_t*>(sb - r);  // aligned dwords of the source from the one that holds sb[0]
  uint32_t* dw = reinterpret_cast<uint32_t*>(d + head);
  if (r == 0) {
    for (unsigned long long w = L; w < words; w += 64) dw[w] = sw[w];
  } else {
    // (r != 0: every destination word needs bytes of both source dwords, so sw[w + 1] holds bytes of the region -- an aligned dword
    // that overlaps the caller's buffer is inside its allocation)
    for (unsigned long long w = L; w < words; w += 64) dw[w] = __builtin_amdgcn_alignbyte(sw[w + 1], sw[w], r);
  }
  const unsigned long long done = head + (words << 2);
  if (done + L < bytes) d[done + L] = s[done + L];  // (at most three bytes)
}

int pack_multi(tracyhip_ctx* ctx, const tracyhip_ragged_src* kinds, uint32_t nkinds, uint32_t n, void* dst, uint64_t dst_cap, uint64_t* kind_bytes) {
  if (nkinds == 0 || nkinds > kMaxKinds || !kinds || !kind_bytes) return set_error(TRACYHIP_ERR_ARG, "tracyhip_pack_ragged: 1 .. %u payload kinds", kMaxKinds);
  for (uint32_t k = 0; k < nkinds; ++k) kind_bytes[k] = 0;
  if (n == 0) return TRACYHIP_OK;
  PackArgs a{};
  a.nkinds = nkinds; a.n = n;
  uint64_t worst = 0;
  for (uint32_t k = 0; k < nkinds; ++k) {
    const tracyhip_ragged_src& q = kinds[k];
    if (!q.src || !q.lens || q.elem_bytes == 0 || q.lens_stride == 0) return set_error(TRACYHIP_ERR_ARG, "tracyhip_pack_ragged: null pointer or zero element size (kind %u)", k);
    a.k[k] = PackKind{static_cast<const uint8_t*>(q.src), (unsigned long long)q.stride_bytes, q.lens, q.lens_stride, q.elem_bytes};
    worst += (uint64_t)n * q.stride_bytes;
  }
  hipStream_t st = ctx->stream;
  const uint32_t wpk = (n + kScanBlock - 1) / kScanBlock, nw = wpk * nkinds;  // scan workgroups per kind, in all
  HIP_TRY(ctx->d_tmp[7].ensure(sizeof(unsigned long long) * ((size_t)nw * kScanBlock + nw + kMaxKinds)));
  unsigned long long* d_local = static_cast<unsigned long long*>(ctx->d_tmp[7].p);
  unsigned long long* d_bsum = d_local + (size_t)nw * kScanBlock;
  unsigned long long* d_kend = d_bsum + nw;
  hipLaunchKernelGGL(pack_local_kernel, dim3(wpk, nkinds), dim3(kScanBlock), 0, st, a, d_local, d_bsum);
  hipLaunchKernelGGL(pack_top_kernel, dim3(1), dim3(kScanThreads), 0, st, d_bsum, nw, wpk, nkinds, d_kend);
  HIP_TRY(hipGetLastError());
  HIP_TRY(ctx->h_res.ensure(sizeof(unsigned long long) * kMaxKinds));
  unsigned long long* h_kend = static_cast<unsigned long long*>(ctx->h_res.p);
  HIP_TRY(hipMemcpyAsync(h_kend, d_kend, sizeof(unsigned long long) * nkinds, hipMemcpyDeviceToHost, st));
  auto report = [&]() {
    for (uint32_t k = 0; k < nkinds; ++k) kind_bytes[k] = h_kend[k] - (k ? h_kend[k - 1] : 0ull);
  };
  if (dst) {
    // a region holds at most its stride: with a capacity for every region in full the copy is queued at once, otherwise the total is read first
    if (dst_cap < worst) {
      HIP_TRY(hipStreamSynchronize(st));  // (not ctx_sync: tracyhip_call_stats::host_syncs counts the pipeline calls' own)
      report();
      if (h_kend[nkinds - 1] > dst_cap)
        return set_error(TRACYHIP_ERR_ARG, "tracyhip_pack_ragged: %llu bytes to pack, capacity %llu", h_kend[nkinds - 1], (unsigned long long)dst_cap);
    }
    hipLaunchKernelGGL(pack_copy_kernel, dim3((n + 3) / 4, nkinds), dim3(256), 0, st, a, d_local, d_bsum, wpk, static_cast<uint8_t*>(dst));
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipStreamSynchronize(st));  // (not ctx_sync: tracyhip_call_stats::host_syncs counts the pipeline calls' own)
  report();
  return TRACYHIP_OK;
}
}  // namespace

extern "C" {

int tracyhip_pack_ragged_multi(tracyhip_ctx* ctx, const tracyhip_ragged_src* kinds, uint32_t nkinds, uint32_t n, void* dst, uint64_t dst_cap, uint64_t* kind_bytes) {
  int rc = ctx_begin(ctx);
  if (rc) return rc;
  return pack_multi(ctx, kinds, nkinds, n, dst, dst_cap, kind_bytes);
}

int tracyhip_pack_ragged(tracyhip_ctx* ctx, const void* src, uint64_t stride_bytes, uint32_t elem_bytes, const uint32_t* lens, uint32_t lens_stride, uint32_t n, void* dst,
                         uint64_t dst_cap, uint64_t* total_bytes) {
  int rc = ctx_begin(ctx);
  if (rc) return rc;
  if (!total_bytes) return set_error(TRACYHIP_ERR_ARG, "tracyhip_pack_ragged: null total");
  const tracyhip_ragged_src one{src, stride_bytes, elem_bytes, lens, lens_stride};
  return pack_multi(ctx, &one, 1, n, dst, dst_cap, total_bytes);
}

}  // extern "C"
