// seg_guard.hip -- content fingerprints: the guard of everything the host side derives from an index and keeps.
//
// The reference derives nothing from its inputs and keeps nothing between calls (csrc/gather_scatter.cpp:25-34 reads the
// caller's tensors afresh every time), so its result always follows the bytes it is handed.  The host side here keeps
// products of index tensors - the slab plan (seg_plan.hip), the edge list sorted by source for the backward pass, the stable
// sort of an index with descents, widened int32 indices, expanded CSR row ids - under the tensors' identity and version
// counter.  A write that goes around the version counter (`.data`, DLPack, another extension's kernel) leaves such a product
// describing bytes that are no longer there.  This kernel is what closes that: a 128-bit position-sensitive fingerprint of the
// inputs is stored with the product when it is made, and every later use re-reads the inputs (8-16 bytes per edge, streamed:
// 115 M edges of two int64 indices = 1.8 GB = ~0.3 ms, against the 8 ms the call takes), compares ON THE DEVICE and publishes
// a verdict word into pinned host memory, which the host reads when the call's own read-back completes (the row rule) - a
// mismatch drops the product and the call runs again from the caller's bytes.
//
// Not cryptographic: two 64-bit sums of keyed 32 x 32 products (NH-style), the keys a function of the word's position.  A product is used wrongly
// only if the stale content collides in both sums.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "geot_hip.h"
#include "internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxSeg = 8;
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

struct FpJob {
  const void *ptr[kMaxSeg];
  unsigned long long words[kMaxSeg]; // words of `unit` bytes
  int unit[kMaxSeg];                 // 16, 8, 4 or 2 bytes per word
  int n;
  unsigned long long total;
  unsigned long long *acc;  // device: [0] ticket (zero on entry, zero again on exit), [2 + 2 b], [3 + 2 b] sums of workgroup b
  unsigned long long *fp;   // device: the product's fingerprint (stored, or compared with)
  int compare;
  long long *verdict;       // pinned host: [0] = 1 match / 2 mismatch, [1] = seq (written last)
  long long seq;
};

// The kernel must stay under the HBM stream it rides on, and 64-bit integer products are slow here (a full splitmix64
// finaliser per word measured 3.7 TB/s, one 64 x 64 product per word 5.5 TB/s: ALU-bound both).  So: NH-style, two
// 32 x 32 -> 64 multiply-accumulates per 8-byte word (v_mad_u64_u32), the word's halves offset by KEYS first.  The keys come
// from a xorshift32 state per lane, seeded from the lane's global id and the segment and advanced once per word - a function
// of the word's position alone, since the launch geometry is a function of the job's sizes alone.
__device__ __forceinline__ void absorb(unsigned long long v, unsigned &s, unsigned long long &h0, unsigned long long &h1) {
  s ^= s << 13;
  s ^= s >> 17;
  s ^= s << 5;
  const unsigned t = __builtin_rotateleft32(s, 15) ^ 0x85EBCA6Bu;
  const unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
  h0 += (unsigned long long)(lo + s) * (hi + t);
  h1 += (unsigned long long)((lo ^ t) + 0x27D4EB2Fu) * ((hi + s) | 1u); // (an index below 2^32 has hi = 0: both factors stay keyed)
}

__device__ __forceinline__ unsigned lane_seed(unsigned long long gtid, int seg) {
  unsigned s = ((unsigned)gtid * 0x9E3779B1u) ^ ((unsigned)(gtid >> 32) * 0x7FEB352Du) ^ ((unsigned)(seg + 1) * 0x85EBCA77u);
  s ^= s >> 15;
  s *= 0x2C1B3C6Du;
  s ^= s >> 12;
  s *= 0x297A2D39u;
  s ^= s >> 15;
  return s ? s : 0x6B43A9B5u; // (xorshift has no way out of zero)
}

__device__ __forceinline__ void absorb_word(u64x2 v, unsigned &s, unsigned long long &h0, unsigned long long &h1) {
  absorb(v.x, s, h0, h1);
  absorb(v.y, s, h0, h1);
}
template <typename W> __device__ __forceinline__ void absorb_word(W v, unsigned &s, unsigned long long &h0, unsigned long long &h1) {
  absorb((unsigned long long)v, s, h0, h1);
}

// One segment: every workgroup takes ONE contiguous span of it (a lane striding over the whole buffer - 4 MB between its
// loads - measured 5.1 TB/s; spans, 8 x 4 KB per round, stream like the tile kernel's reads), eight loads in flight per lane.
template <typename W>
__device__ __forceinline__ void absorb_segment(const W *__restrict__ p, unsigned long long n, unsigned &s, unsigned long long &h0,
                                               unsigned long long &h1) {
  constexpr unsigned long long kRound = 8ull * kThreads;
  const unsigned long long span = ((n + gridDim.x - 1) / gridDim.x + kRound - 1) / kRound * kRound;
  const unsigned long long b0 = (unsigned long long)blockIdx.x * span;
  if (b0 >= n) return;
  const unsigned long long b1 = b0 + span < n ? b0 + span : n;
  unsigned long long i = b0 + threadIdx.x;
  for (; i + 7 * kThreads < b1; i += kRound) {
    W v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + i + u * kThreads);
#pragma unroll
    for (int u = 0; u < 8; ++u) absorb_word(v[u], s, h0, h1);
  }
  for (; i < b1; i += kThreads) absorb_word(__builtin_nontemporal_load(p + i), s, h0, h1);
}

__global__ __launch_bounds__(kThreads) void fingerprint_kernel(FpJob j) {
  unsigned long long h0 = 0, h1 = 0;
  const unsigned long long gtid = (unsigned long long)blockIdx.x * kThreads + threadIdx.x;
  for (int sg = 0; sg < j.n; ++sg) {
    const unsigned long long n = j.words[sg];
    unsigned s = lane_seed(gtid, sg);
    switch (j.unit[sg]) {
    case 16: absorb_segment(static_cast<const u64x2 *>(j.ptr[sg]), n, s, h0, h1); break;
    case 8: absorb_segment(static_cast<const unsigned long long *>(j.ptr[sg]), n, s, h0, h1); break;
    case 4: absorb_segment(static_cast<const unsigned int *>(j.ptr[sg]), n, s, h0, h1); break;
    default: absorb_segment(static_cast<const unsigned short *>(j.ptr[sg]), n, s, h0, h1); break;
    }
  }
  // wave, then workgroup, then device
  for (int d = 32; d > 0; d >>= 1) {
    h0 += __shfl_down(h0, d, 64);
    h1 += __shfl_down(h1, d, 64);
  }
  __shared__ unsigned long long part[2][kThreads / 64];
  __shared__ bool last;
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    part[0][wave] = h0;
    part[1][wave] = h1;
  }
  __syncthreads();
  // (same-address atomics cost ~15 ns each: 4096 workgroups adding two sums and a ticket measured 0.22 ms of nothing else -
  // the sums go to the workgroup's own slot, one ticket per workgroup, the last one adds the slots up)
  if (threadIdx.x == 0) {
    unsigned long long s0 = 0, s1 = 0;
    for (int w = 0; w < kThreads / 64; ++w) {
      s0 += part[0][w];
      s1 += part[1][w];
    }
    __hip_atomic_store(j.acc + 2 + 2 * blockIdx.x, s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(j.acc + 3 + 2 * blockIdx.x, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    last = atomicAdd(j.acc, 1ull) == (unsigned long long)gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  h0 = h1 = 0;
  for (unsigned b = threadIdx.x; b < gridDim.x; b += kThreads) {
    h0 += __hip_atomic_load(j.acc + 2 + 2 * b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    h1 += __hip_atomic_load(j.acc + 3 + 2 * b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  for (int d = 32; d > 0; d >>= 1) {
    h0 += __shfl_down(h0, d, 64);
    h1 += __shfl_down(h1, d, 64);
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    part[0][wave] = h0;
    part[1][wave] = h1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long f0 = 0, f1 = 0;
    for (int w = 0; w < kThreads / 64; ++w) {
      f0 += part[0][w];
      f1 += part[1][w];
    }
    f1 ^= j.total; // (the lengths folded in)
    __hip_atomic_store(j.acc, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long long v = 1;
    if (j.compare) {
      v = (__hip_atomic_load(j.fp + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == f0 &&
           __hip_atomic_load(j.fp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == f1) ? 1 : 2;
    } else {
      j.fp[0] = f0;
      j.fp[1] = f1;
    }
    if (j.verdict) {
      __hip_atomic_store(j.verdict, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __threadfence_system();
      __hip_atomic_store(j.verdict + 1, j.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

constexpr int kMaxGrid = 1024; // 4 workgroups per CU

} // namespace

extern "C" size_t geot_content_fingerprint_scratch_bytes(void) { return (size_t)(2 + 2 * kMaxGrid) * sizeof(unsigned long long); }

extern "C" int geot_content_fingerprint(const void *const *bufs, const size_t *bytes, int nbufs, unsigned long long *fp, int compare,
                                        int64_t *verdict2, int64_t seq, void *scratch, void *stream) {
  if (nbufs < 1 || nbufs > 4 || !bufs || !bytes || !fp || !scratch) return geot_internal_fail(GEOT_EINVAL, "geot_content_fingerprint: 1..4 buffers, fp and scratch required");
  FpJob j{};
  unsigned long long total_bytes = 0;
  for (int b = 0; b < nbufs; ++b) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(bufs[b]);
    size_t left = bytes[b];
    total_bytes = total_bytes * 0x100000001B3ull + left;
    if (left == 0) continue;
    if (!bufs[b]) return geot_internal_fail(GEOT_EINVAL, "geot_content_fingerprint: null buffer");
    if ((a & 1) || (left & 1)) return geot_internal_fail(GEOT_EINVAL, "geot_content_fingerprint: buffers of whole 2-byte words, 2-byte aligned");
    // the body in the widest words its address and length allow, the rest behind it in narrower ones
    const char *p = static_cast<const char *>(bufs[b]);
    const int body_unit = (a % 16 == 0 && left >= 16) ? 16 : (a % 8 == 0 && left >= 8) ? 8 : (a % 4 == 0 && left >= 4) ? 4 : 2;
    const size_t body = left / body_unit * body_unit;
    j.ptr[j.n] = p;
    j.unit[j.n] = body_unit;
    j.words[j.n] = body / body_unit;
    ++j.n;
    left -= body;
    if (left) { // (< 16 bytes: 2-byte words)
      j.ptr[j.n] = p + body;
      j.unit[j.n] = 2;
      j.words[j.n] = left / 2;
          ++j.n;
    }
  }
  j.total = total_bytes;
  j.acc = static_cast<unsigned long long *>(scratch);
  j.fp = fp;
  j.compare = compare;
  j.verdict = reinterpret_cast<long long *>(verdict2);
  j.seq = seq;
  // enough workgroups to fill the chip for the large jobs, one for the small ones (a lane takes 128 bytes per round)
  unsigned long long work = 0;
  for (int s = 0; s < j.n; ++s) work += j.words[s];
  int grid = (int)((work + (unsigned long long)kThreads * 8 - 1) / ((unsigned long long)kThreads * 8));
  if (grid < 1) grid = 1;
  if (grid > kMaxGrid) grid = kMaxGrid;
  hipLaunchKernelGGL(fingerprint_kernel, dim3(grid), dim3(kThreads), 0, static_cast<hipStream_t>(stream), j);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return geot_internal_fail(GEOT_ELAUNCH, hipGetErrorString(e));
  return GEOT_OK;
}
