// pipe_kernels.h -- the small kernels around the DP launches of the two pipelines (trimReferenceSlice, row-m ends, row maxima,
// reference encoders, the orientation vote).  Included by pipeline.hip (pipelines planned by the host) and stream.hip (planned on the
// device); everything sits in an anonymous namespace, so each translation unit carries its own copy.
#ifndef TRACY_AMD_PIPE_KERNELS_H
#define TRACY_AMD_PIPE_KERNELS_H

#include <hip/hip_runtime.h>

#include "capi_internal.h"

namespace {
using namespace tracyhip;

struct TrimOut {
  uint32_t ri;       // offset of the trimmed slice in the oriented reference
  uint32_t len;      // its length after std::string::substr clamping
  uint32_t pos;      // rs.pos after the update (rs.pos starts at 0)
  uint32_t pad;
};

// the widening / clamping / rs.pos part of trimReferenceSlice (fmindex.h:443-461)
__device__ inline TrimOut trim_finish(uint32_t ri, uint32_t risize, uint32_t n, uint32_t trim_left, uint32_t trim_right, bool forward) {
  if (ri >= trim_left) { ri -= trim_left; risize += trim_left; }
  if ((uint32_t)(ri + risize + trim_right) < n) risize += trim_right;
  TrimOut r;
  r.ri = ri;
  r.len = (ri <= n) ? ((risize < n - ri) ? risize : n - ri) : 0;  // substr(ri, risize)
  r.pos = 0;
  if (forward) r.pos = ri;
  else {
    const int32_t offset = (int32_t)n - (int32_t)ri - (int32_t)risize;
    if (offset >= 0) r.pos = (uint32_t)offset;  // negative: the reference only warns (fmindex.h:457-459)
  }
  r.pad = 0;
  return r;
}

// trimReferenceSlice (fmindex.h:429-463) evaluated directly on the traceback string.  ops are in push
// order (end -> start); alignment column j (forward) is ops[L-1-j].  Row 0 holds a trace base unless
// the op is 'h', row 1 holds a reference base unless the op is 'v' (align.h:204-214).
// The reference scans for s = first column with a trace base and e = last such column + 1, then counts
// reference bases before s (ri) and inside [s, e) (risize).  Every column before s and from e on is an
// 'h' (a reference base), and the alignment consumes all n reference bases, so ri = s and
// risize = n - s - (L - e): only the two ends of the string have to be looked at.  One wave per trace.
__global__ __launch_bounds__(64) void trim_kernel(const uint8_t* __restrict__ ops, const uint64_t* __restrict__ ops_off,
                                                  const uint32_t* __restrict__ ops_len, const uint32_t* __restrict__ ref_len,
                                                  const uint8_t* __restrict__ forward, uint32_t trim_left,
                                                  uint32_t trim_right, uint32_t ntraces, TrimOut* __restrict__ out) {
  const uint32_t t = blockIdx.x;
  if (t >= ntraces) return;
  const uint8_t* o = ops + ops_off[t];
  const uint32_t L = ops_len[t];
  const uint32_t lane = threadIdx.x;
  // s: first forward column that is not 'h'  <=>  scanning the push-order string from its end
  int32_t s = -1, e = -1;
  for (uint32_t base = 0; base < L; base += 64) {
    const uint32_t j = base + lane;
    const bool hit = (j < L) && (o[L - 1 - j] != 'h');
    const unsigned long long m = __ballot(hit);
    if (m) { s = (int32_t)(base + (uint32_t)__builtin_ctzll(m)); break; }
  }
  if (s >= 0) {  // e: last forward column that is not 'h', + 1  <=>  scanning the push-order string from its start
    for (uint32_t base = 0; base < L; base += 64) {
      const uint32_t q = base + lane;  // push-order index q <-> forward column L-1-q
      const bool hit = (q < L) && (o[q] != 'h');
      const unsigned long long m = __ballot(hit);
      if (m) { e = (int32_t)(L - (base + (uint32_t)__builtin_ctzll(m))); break; }
    }
  }
  if (lane != 0) return;
  const uint32_t n = ref_len[t];
  uint32_t ri, risize;
  if (s < 0) {  // no trace base at all: every column counts towards ri (fmindex.h:435-441), the span is empty
    uint32_t cnt = 0;
    for (uint32_t j = 0; j < L; ++j) cnt += (o[j] != 'v');
    ri = cnt;
    risize = 0;
  } else {
    ri = (uint32_t)s;
    // reference bases inside [s, e): all n bases minus the leading s columns and the trailing L - e columns
    uint32_t inside = 0;
    const uint32_t lead = (uint32_t)s, trail = L - (uint32_t)e;
    // columns before s and from e on are 'h' only when the string really is a complete alignment; count exactly
    // when the totals do not add up (defensive: degenerate inputs)
    uint32_t refcols = 0;
    if (lead + trail <= n) inside = n - lead - trail;
    else { for (int32_t j = s; j < e; ++j) refcols += (o[L - 1 - j] != 'v'); inside = refcols; }
    risize = inside;
  }
  out[t] = trim_finish(ri, risize, n, trim_left, trim_right, forward[t] != 0);
}

// trimReferenceSlice from the two ends of the alignment alone (origin-tracking sweep, dp_kernels.h gotoh_origin_body):
// ends[2t] = leading 'h' columns, ends[2t+1] = last column that is not a trailing 'h'; every reference base in between
// belongs to the slice.
__global__ void trim_from_ends_kernel(const uint32_t* __restrict__ ends, const uint32_t* __restrict__ ref_len,
                                      const uint8_t* __restrict__ forward, uint32_t trim_left, uint32_t trim_right, uint32_t ntraces,
                                      TrimOut* __restrict__ out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ntraces) return;
  const uint32_t lead = ends[2 * t], ce = ends[2 * t + 1];
  out[t] = trim_finish(lead, ce >= lead ? ce - lead : 0u, ref_len[t], trim_left, trim_right, forward[t] != 0);
}

// c_e of a pair from the row-m values the 16-bit sweep left behind ({H, E - goe} per column): the last column whose H(m, c) is
// strictly greater than E(m, c) -- where the reference's traceback leaves the trailing run of row m.  One wave per pair.
struct RowEndDesc { uint64_t off; uint32_t n, pad; };
__global__ __launch_bounds__(64) void row_m_end_kernel(const RowEndDesc* __restrict__ desc, const int32_t* __restrict__ lastrow, int32_t goe,
                                                       uint32_t* __restrict__ ce) {
  const RowEndDesc d = desc[blockIdx.x];
  const int32_t* lr = lastrow + d.off;
  uint32_t found = 0;
  for (int64_t base = d.n; base >= 1 && !found; base -= 64) {
    const int64_t c = base - threadIdx.x;
    bool hit = false;
    if (c >= 1) { const int32_t v = lr[c]; hit = sext16(v) > (v >> 16) + goe; }
    const unsigned long long mask = __ballot(hit);
    if (mask) found = (uint32_t)(base - __builtin_ctzll(mask));  // lane 0 holds the largest column
  }
  if (threadIdx.x == 0) ce[blockIdx.x] = found;
}
__global__ void ends_shift_kernel(uint32_t* __restrict__ ends, const uint32_t* __restrict__ shift, uint32_t ntraces) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < ntraces) { ends[2 * t] += shift[t]; ends[2 * t + 1] += shift[t]; }
}

// trimReferenceSlice (fmindex.h:429-463) on the two alignment rows themselves, as the reference scans them: s / e = first / last + 1
// column holding a trace base, ri = reference bases before s, risize = reference bases in [s, e).  One wave per trace.
struct TrimRowsDesc { uint64_t off; uint32_t L, n; uint8_t forward, pad[7]; };
__global__ __launch_bounds__(64) void trim_rows_kernel(const TrimRowsDesc* __restrict__ desc, const uint8_t* __restrict__ rows0,
                                                       const uint8_t* __restrict__ rows1, uint32_t trim_left, uint32_t trim_right,
                                                       uint32_t ntraces, TrimOut* __restrict__ out) {
  const uint32_t t = blockIdx.x;
  if (t >= ntraces) return;
  const TrimRowsDesc d = desc[t];
  const uint8_t* r0 = rows0 + d.off;
  const uint8_t* r1 = rows1 + d.off;
  const uint32_t lane = threadIdx.x, L = d.L;
  int32_t s = -1, e = -1;
  uint32_t ri = 0;
  for (uint32_t base = 0; base < L && s < 0; base += 64) {  // first column with a trace base; reference bases before it
    const uint32_t j = base + lane;
    const bool tb = (j < L) && (r0[j] != '-');
    const bool rb = (j < L) && (r1[j] != '-');
    const unsigned long long mt = __ballot(tb), mr = __ballot(rb);
    if (mt) {
      const uint32_t first = (uint32_t)__builtin_ctzll(mt);
      s = (int32_t)(base + first);
      ri += (uint32_t)__popcll(mr & ((1ull << first) - 1ull));
    } else {
      ri += (uint32_t)__popcll(mr);
    }
  }
  uint32_t risize = 0;
  if (s >= 0) {
    for (uint32_t base = 0; base < L; base += 64) {  // last column with a trace base, scanning from the end
      const uint32_t q = base + lane;
      const bool tb = (q < L) && (r0[L - 1 - q] != '-');
      const unsigned long long m = __ballot(tb);
      if (m) { e = (int32_t)(L - (base + (uint32_t)__builtin_ctzll(m))); break; }
    }
    for (uint32_t base = (uint32_t)s; base < (uint32_t)e; base += 64) {
      const uint32_t j = base + lane;
      risize += (uint32_t)__popcll(__ballot(j < (uint32_t)e && r1[j] != '-'));
    }
  }
  if (lane == 0) out[t] = trim_finish(ri, risize, d.n, trim_left, trim_right, d.forward != 0);
}

// (loadSingleFasta hands over upper-case [ACGTN] only (fasta.h:54-95); anything else makes the string and profile reverse
// complements (fmindex.h:8-24 vs profile.h:74-90) disagree, so it is rejected: encode_codes_kernel's verr.)

// upper bound for what rows [first, m) of a trimmed profile view can still add to a semiglobal score: every row
// adds at most max(0, its best one-hot substitution score) (gaps cost <= 0 when go <= 0 and ge < 0)
struct RowMaxDesc { uint64_t off; uint32_t stride, m, first; };
// out1 (or null): the same sum with the rows clamped at -1 instead of 0 (the allowance of front.h's second certificate)
__global__ __launch_bounds__(64) void rowmax_rest_kernel(const RowMaxDesc* desc, const float* prof, float fmatch, float fmis, int32_t* out,
                                                         int32_t* out1 = nullptr) {
  const RowMaxDesc d = desc[blockIdx.x];
  int32_t sum = 0, sum1 = 0;
  for (uint32_t r = d.first + threadIdx.x; r < d.m; r += 64) {
    float pr[5];
    for (int k = 0; k < 5; ++k) pr[k] = prof[d.off + (uint64_t)k * d.stride + r];
    int32_t best = INT32_MIN;
    for (uint32_t b = 0; b < 5; ++b) {
      const int32_t q = onehot_score(pr, b, fmatch, fmis);
      best = q > best ? q : best;
    }
    sum += best > 0 ? best : 0;
    sum1 += best > -1 ? best : -1;
  }
  for (int o = 32; o > 0; o >>= 1) { sum += __shfl_down(sum, o, 64); sum1 += __shfl_down(sum1, o, 64); }
  if (threadIdx.x == 0) {
    out[blockIdx.x] = sum;
    if (out1) out1[blockIdx.x] = sum1;
  }
}

// MODE_CQ (strings scored through the query-profile table): case-sensitive column codes, and the test that row strings hold
// nothing but the five letters the table has entries for
// flag |= 2 where a column is none of A C G T N, |= 4 where it is N (both rare): the origin sweeps size their table by it
// special (or null): the block map of the codes written (DpArgs::special_blocks: one byte per 256 code bytes, set where a block holds
// anything but A C G T)
// (sixteen bytes per thread, as encode_codes_kernel: a byte per thread made 3 * 10^8 threads of the windows of a 100 000-trace batch;
// flags through one atomic per wave that has something to report)
constexpr uint32_t kCqBytesPerThread = 16;
__global__ void encode_cq_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n, int32_t* flag, uint8_t* __restrict__ special) {
  const uint64_t i0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * kCqBytesPerThread;
  int32_t f = 0;
  if (i0 < n) {
    const uint32_t cnt = n - i0 < kCqBytesPerThread ? (uint32_t)(n - i0) : kCqBytesPerThread;
    uint8_t b[kCqBytesPerThread];
    if (cnt == kCqBytesPerThread) __builtin_memcpy(b, in + i0, kCqBytesPerThread);  // (any alignment)
    else for (uint32_t j = 0; j < cnt; ++j) b[j] = in[i0 + j];
    bool any = false;
#pragma unroll
    for (uint32_t j = 0; j < kCqBytesPerThread; ++j) {
      const uint32_t c = j < cnt ? cq_code(b[j]) : 0u;
      b[j] = (uint8_t)c;
      if (c >= 4u) { f |= c >= 5u ? 2 : 4; any = true; }
    }
    if (cnt == kCqBytesPerThread) __builtin_memcpy(out + i0, b, kCqBytesPerThread);
    else for (uint32_t j = 0; j < cnt; ++j) out[i0 + j] = b[j];
    if (any && special) { special[i0 >> 8] = 1; special[(i0 + cnt - 1) >> 8] = 1; }  // (sixteen bytes lie in at most two blocks of 256)
  }
  if (f) atomicOr(flag, f);
}
__global__ void cq_rows_kernel(const uint8_t* __restrict__ in, uint64_t n, int32_t* flag) {
  const uint64_t i0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * kCqBytesPerThread;
  if (i0 >= n) return;
  const uint32_t cnt = n - i0 < kCqBytesPerThread ? (uint32_t)(n - i0) : kCqBytesPerThread;
  uint8_t b[kCqBytesPerThread];
  if (cnt == kCqBytesPerThread) __builtin_memcpy(b, in + i0, kCqBytesPerThread);
  else for (uint32_t j = 0; j < cnt; ++j) b[j] = in[i0 + j];
  bool bad = false;
#pragma unroll
  for (uint32_t j = 0; j < kCqBytesPerThread; ++j)
    if (j < cnt && !cq_row_char(b[j])) bad = true;
  if (bad) atomicOr(flag, 1);
}

// reference characters -> profile-row codes (align.h:121-136), sixteen bytes per thread.  special: one byte per 256 code bytes, set
// where a block holds an N or '-' / other code.  verr (or null): |= 4 when a byte is not one of A C G T N (the validation
// verdict, folded into the same pass).
__device__ __forceinline__ uint32_t encode_word(uint32_t w, uint32_t cnt, bool& any_special, bool& invalid) {
  uint32_t codes = 0;
#pragma unroll
  for (uint32_t j = 0; j < 4u; ++j) {
    const uint8_t ch = (uint8_t)(w >> (8 * j));
    // A C G T N in either case -> 0..4, '-' / anything else -> 5 (dp_code), without a branch per byte
    const uint8_t up = ch & 0xdfu;
    const uint32_t c = up == 'A' ? 0u : up == 'C' ? 1u : up == 'G' ? 2u : up == 'T' ? 3u : up == 'N' ? 4u : 5u;
    if (j < cnt) {
      invalid |= !(ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T' || ch == 'N');
      any_special |= c >= 4u;
    }
    codes |= c << (8 * j);
  }
  return codes;
}
__global__ __launch_bounds__(256) void encode_codes_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n,
                                                           uint8_t* __restrict__ special, int32_t* __restrict__ verr) {
  const uint64_t i0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16u;
  if (i0 >= n) return;
  bool any_special = false, invalid = false;
  if (n - i0 >= 16u) {
    uint32_t w[4];
    __builtin_memcpy(w, in + i0, 16);  // (unaligned: the payload may start anywhere)
#pragma unroll
    for (int q = 0; q < 4; ++q) w[q] = encode_word(w[q], 4u, any_special, invalid);
    __builtin_memcpy(out + i0, w, 16);
  } else {
    for (uint64_t i = i0; i < n; i += 4) {
      const uint32_t cnt = (n - i < 4u) ? (uint32_t)(n - i) : 4u;
      uint32_t w = 0;
      for (uint32_t j = 0; j < cnt; ++j) w |= (uint32_t)in[i + j] << (8 * j);
      const uint32_t c = encode_word(w, cnt, any_special, invalid);
      for (uint32_t j = 0; j < cnt; ++j) out[i + j] = (uint8_t)(c >> (8 * j));
    }
  }
  if (any_special) special[i0 >> 8] = 1;  // rare; sixteen bytes from a 16-byte boundary lie in one 256-byte block
  if (verr && invalid) atomicOr(verr, 4);
}

// Orientation vote: shared 11-mers between the trace (consensus base per profile column) and its reference window, read
// forward and as the reverse complement.  Only a GUESS of which strand to sweep first -- the strand is decided by the
// scores and the certificate below, a wrong or missing vote costs time, never the result.  One wave per trace; two
// hashed bitmaps of the trace's k-mers (as they are / reverse-complemented) in LDS, the window's k-mers probe both.
struct VoteDesc { uint64_t a1_off, a2_off; uint32_t stride, m, n, pad; };
constexpr int kVoteK = 11;
constexpr uint32_t kVoteBits = 1u << 16;
constexpr uint32_t kVotePiece = 65, kVoteTile = 64 * kVotePiece;  // window positions per lane and per LDS tile
__device__ inline uint32_t vote_hash(uint32_t kmer) { return (kmer * 0x9E3779B1u) >> 16; }
__global__ __launch_bounds__(64) void kmer_vote_kernel(const VoteDesc* __restrict__ desc, const float* __restrict__ prof,
                                                       const uint8_t* __restrict__ codes, uint32_t* __restrict__ votes) {
  __shared__ uint32_t bm[2][kVoteBits / 32];
  __shared__ uint8_t cons[1040];
  __shared__ __attribute__((aligned(16))) uint8_t win[kVoteTile + 16];
  const VoteDesc d = desc[blockIdx.x];
  const uint32_t lane = threadIdx.x;
  uint32_t hf = 0, hr = 0;
  const uint32_t m = d.m < 1024u ? d.m : 1024u;
  if (m >= (uint32_t)kVoteK && d.n >= (uint32_t)kVoteK) {
    for (uint32_t i = lane; i < kVoteBits / 32; i += 64) { bm[0][i] = 0; bm[1][i] = 0; }
    // the four rows of eight rounds of columns requested together (clamped columns, selects): a round of 64 columns per wait was 16 memory
    // round trips per trace, and the kernel runs beside the peak table's and the score tables' streaming passes
    constexpr uint32_t kCols = 8;
    for (uint32_t j0 = 0; j0 < m; j0 += 64 * kCols) {
      float v[kCols][4];
#pragma unroll
      for (uint32_t u = 0; u < kCols; ++u) {
        const uint32_t j = j0 + 64 * u + lane, jc = j < m ? j : m - 1;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) v[u][k] = prof[d.a1_off + (uint64_t)k * d.stride + jc];
      }
#pragma unroll
      for (uint32_t u = 0; u < kCols; ++u) {
        const uint32_t j = j0 + 64 * u + lane;
        uint32_t best = 0;
        float bv = v[u][0];
#pragma unroll
        for (uint32_t k = 1; k < 4; ++k) {
          const bool g = v[u][k] > bv;
          bv = g ? v[u][k] : bv;
          best = g ? k : best;
        }
        if (j < m) cons[j] = (uint8_t)best;
      }
    }
    __syncthreads();
    constexpr uint32_t mask = (1u << (2 * kVoteK)) - 1u;
    for (uint32_t i = lane; i + kVoteK <= m; i += 64) {
      uint32_t f = 0, r = 0;
      for (int j = 0; j < kVoteK; ++j) {
        const uint32_t c = cons[i + j];
        f = (f << 2) | c;
        r |= (3u - c) << (2 * j);  // reverse complement: complemented bases in reverse order
      }
      const uint32_t h0 = vote_hash(f & mask), h1 = vote_hash(r & mask);
      atomicOr(&bm[0][h0 >> 5], 1u << (h0 & 31));
      atomicOr(&bm[1][h1 >> 5], 1u << (h1 & 31));
    }
    __syncthreads();
    // The window goes through LDS in tiles (coalesced loads); within a tile every lane rolls over its own contiguous piece
    // (kVoteK - 1 bases of overlap with the next piece).  Pieces of kVotePiece = 65 positions: an odd stride, so the byte
    // reads of the 64 lanes spread over the banks.  (Rolling straight from global memory -- one dependent, uncoalesced byte
    // load per position -- took five times as long.)
    const uint32_t npos = d.n - kVoteK + 1;
    for (uint32_t tile = 0; tile < npos; tile += kVoteTile) {
      const uint32_t tn = (npos - tile < kVoteTile) ? npos - tile : kVoteTile;  // positions of this tile
      const uint32_t nbytes = tn + kVoteK - 1;
      __syncthreads();
      // the tile by ALIGNED dwords, eight rounds of 256 bytes per wait: the tile lies in LDS at the misalignment `skew` it has in memory
      // (the up to three bytes in front of it and behind it are other codes or the code buffer's spare bytes, kCodePad)
      const uint8_t* src = codes + d.a2_off + tile;
      const uint32_t skew = (uint32_t)(reinterpret_cast<uintptr_t>(src) & 3u);
      {
        const uint32_t* src32 = reinterpret_cast<const uint32_t*>(src - skew);
        uint32_t* win32 = reinterpret_cast<uint32_t*>(win);
        const uint32_t nd = (skew + nbytes + 3u) >> 2;
        constexpr uint32_t kRounds = 8;
        for (uint32_t w0 = 0; w0 < nd; w0 += 64 * kRounds) {
          uint32_t x[kRounds];
#pragma unroll
          for (uint32_t u = 0; u < kRounds; ++u) {
            const uint32_t wi = w0 + 64 * u + lane;
            x[u] = src32[wi < nd ? wi : nd - 1];
          }
#pragma unroll
          for (uint32_t u = 0; u < kRounds; ++u) {
            const uint32_t wi = w0 + 64 * u + lane;
            if (wi < nd) win32[wi] = x[u];
          }
        }
      }
      __syncthreads();
      // every lane rolls over its piece; no branch: the bitmaps are read for every position and a position without eleven valid bases
      // before it counts nothing
      const uint32_t lo = lane * kVotePiece, hi = (lo + kVotePiece < tn) ? lo + kVotePiece : tn;
      const uint32_t end = lo < hi ? hi + kVoteK - 1 : lo;
      uint32_t k = 0, valid = 0;
#pragma unroll 5
      for (uint32_t q = 0; q < kVotePiece + kVoteK - 1; ++q) {
        const uint32_t p = lo + q;
        const uint32_t c0 = win[skew + p < kVoteTile + 15u ? skew + p : kVoteTile + 15u];
        const uint32_t c = p < end ? c0 : 4u;
        k = ((k << 2) | (c & 3u)) & mask;
        valid = c < 4u ? valid + 1u : 0u;
        const uint32_t h = vote_hash(k);
        const uint32_t hit = valid >= (uint32_t)kVoteK ? 1u : 0u;
        hf += (bm[0][h >> 5] >> (h & 31)) & hit;
        hr += (bm[1][h >> 5] >> (h & 31)) & hit;
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) { hf += __shfl_down(hf, o, 64); hr += __shfl_down(hr, o, 64); }
  if (lane == 0) { votes[2 * blockIdx.x] = hf; votes[2 * blockIdx.x + 1] = hr; }
}

}  // namespace
#endif
