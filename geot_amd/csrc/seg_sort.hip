// seg_sort.hip -- the stable sort the unsorted path reduces over.
//
// An index with descents (the reference's scatter_reduce_kernel case, csrc/cuda/index_scatter_kernel.cuh:204-263: one
// atomicAdd per edge) is reduced here over its stable sort by the atomic-free gather-mode kernels.  The sort is the
// price of the first sighting of such an index, so it only moves the bits the keys use: the probe
// (geot_index_probe_range) returns the key range with the row rule, keys below 2^32 are narrowed to 32 bits next to a
// 32-bit position, rocPRIM's LSD radix sort (stable) runs over bits [0, bit_width(max)) only, and one last pass widens
// (keys, positions) to the int64 the kernels read.  1 M keys = 20 bits = 3 passes over 8 bytes per edge, where a
// generic 64-bit sort of (int64, int64) pairs runs 8 passes over 16 bytes per edge.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include <cstdint>

#include "geot_hip.h"
#include "internal.h"

namespace {

constexpr int kThreads = 256;

inline size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }

#define SORT_TRY(expr)                                                                                                   \
  do {                                                                                                                   \
    const hipError_t e_ = (expr);                                                                                        \
    if (e_ != hipSuccess) return geot_internal_fail(GEOT_ELAUNCH, hipGetErrorString(e_));                                   \
  } while (0)

// (VEC: every pointer 16-byte aligned - two elements per lane per access)
template <bool VEC>
__global__ __launch_bounds__(kThreads) void narrow_kernel(const int64_t *__restrict__ index, uint32_t *__restrict__ k32,
                                                          uint32_t *__restrict__ v32, int64_t nnz) {
  const int64_t i = ((int64_t)blockIdx.x * kThreads + threadIdx.x) * 2;
  if (VEC && i + 1 < nnz) {
    const longlong2 k = *reinterpret_cast<const longlong2 *>(index + i);
    *reinterpret_cast<uint2 *>(k32 + i) = make_uint2((uint32_t)k.x, (uint32_t)k.y);
    *reinterpret_cast<uint2 *>(v32 + i) = make_uint2((uint32_t)i, (uint32_t)i + 1u);
  } else {
    for (int64_t j = i; j < i + 2 && j < nnz; ++j) {
      k32[j] = (uint32_t)index[j];
      v32[j] = (uint32_t)j;
    }
  }
}

template <bool VEC>
__global__ __launch_bounds__(kThreads) void widen_kernel(const uint32_t *__restrict__ k32, const uint32_t *__restrict__ v32,
                                                         int64_t *__restrict__ keys, int64_t *__restrict__ perm, int64_t nnz) {
  const int64_t i = ((int64_t)blockIdx.x * kThreads + threadIdx.x) * 2;
  if (VEC && i + 1 < nnz) {
    const uint2 k = *reinterpret_cast<const uint2 *>(k32 + i), v = *reinterpret_cast<const uint2 *>(v32 + i);
    *reinterpret_cast<longlong2 *>(keys + i) = make_longlong2((long long)k.x, (long long)k.y);
    *reinterpret_cast<longlong2 *>(perm + i) = make_longlong2((long long)v.x, (long long)v.y);
  } else {
    for (int64_t j = i; j < i + 2 && j < nnz; ++j) {
      keys[j] = k32[j];
      perm[j] = v32[j];
    }
  }
}

// out4 = {index[nnz-1], descents, min, max}; out4[1] zeroed, out4[2] / out4[3] set to +/- extremes by the launcher
__global__ __launch_bounds__(kThreads) void index_range_kernel(const int64_t *__restrict__ index, int64_t nnz, int64_t *__restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  long long bad = 0, lo = INT64_MAX, hi = INT64_MIN;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nnz; i += stride) {
    const long long a = index[i];
    if (i + 1 < nnz) bad += a > index[i + 1];
    lo = a < lo ? a : lo;
    hi = a > hi ? a : hi;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    bad += __shfl_xor(bad, d, 64);
    const long long l2 = __shfl_xor(lo, d, 64), h2 = __shfl_xor(hi, d, 64);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
  }
  __shared__ long long part[3][kThreads / 64];
  if ((threadIdx.x & 63) == 0) {
    part[0][threadIdx.x >> 6] = bad;
    part[1][threadIdx.x >> 6] = lo;
    part[2][threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kThreads / 64; ++w) {
      bad += part[0][w];
      lo = part[1][w] < lo ? part[1][w] : lo;
      hi = part[2][w] > hi ? part[2][w] : hi;
    }
    if (gridDim.x == 1) { // small index: one workgroup, plain stores, no memset before the launch
      out[1] = bad;
      out[2] = lo;
      out[3] = hi;
    } else {
      if (bad) atomicAdd(reinterpret_cast<unsigned long long *>(out + 1), (unsigned long long)bad);
      atomicMin(reinterpret_cast<long long *>(out + 2), lo);
      atomicMax(reinterpret_cast<long long *>(out + 3), hi);
    }
    if (blockIdx.x == 0) out[0] = index[nnz - 1];
  }
}

__global__ void range_init_kernel(int64_t *out) {
  out[1] = 0;
  out[2] = INT64_MAX;
  out[3] = INT64_MIN;
}

struct SortLayout {
  size_t k0, k1, v0, v1, tmp, total;
};
SortLayout sort_layout(int64_t nnz, size_t rocprim_bytes) {
  SortLayout L;
  const size_t arr = up256((size_t)(nnz > 0 ? nnz : 1) * sizeof(uint32_t));
  L.k0 = 0;
  L.k1 = arr;
  L.v0 = 2 * arr;
  L.v1 = 3 * arr;
  L.tmp = 4 * arr;
  L.total = 4 * arr + up256(rocprim_bytes);
  return L;
}

hipError_t radix_pairs(void *tmp, size_t &bytes, uint32_t *k0, uint32_t *k1, uint32_t *v0, uint32_t *v1, int64_t nnz, int bits,
                       hipStream_t st, uint32_t **k_out, uint32_t **v_out) {
  rocprim::double_buffer<uint32_t> keys(k0, k1), vals(v0, v1);
  const hipError_t e = rocprim::radix_sort_pairs(tmp, bytes, keys, vals, (size_t)nnz, 0u, (unsigned)bits, st);
  if (k_out) *k_out = keys.current();
  if (v_out) *v_out = vals.current();
  return e;
}

} // namespace

extern "C" {

int geot_index_probe_range(const int64_t *index, int64_t nnz, int64_t *out4, void *stream) {
  if (nnz <= 0 || !index || !out4) return geot_internal_fail(GEOT_EINVAL, "index_probe_range: needs a non-empty index and an output");
  hipStream_t st = static_cast<hipStream_t>(stream);
  int64_t blocks = (nnz + kThreads * 4 - 1) / (kThreads * 4);
  if (nnz <= 32768) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  if (blocks > 1) hipLaunchKernelGGL(range_init_kernel, dim3(1), dim3(1), 0, st, out4);
  hipLaunchKernelGGL(index_range_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, st, index, nnz, out4);
  SORT_TRY(hipGetLastError());
  return GEOT_OK;
}

int geot_sort_supported(int64_t nnz, int64_t key_min, int64_t key_max) {
  return nnz > 0 && nnz < ((int64_t)1 << 32) && key_min >= 0 && key_max < ((int64_t)1 << 32);
}

size_t geot_sort_workspace_bytes(int64_t nnz) {
  size_t bytes = 0;
  if (nnz <= 0) return 256;
  if (radix_pairs(nullptr, bytes, nullptr, nullptr, nullptr, nullptr, nnz, 32, nullptr, nullptr, nullptr) != hipSuccess) return 0;
  return sort_layout(nnz, bytes).total;
}

int geot_sort_index(const int64_t *index, int64_t nnz, int64_t key_max, int64_t *keys_out, int64_t *perm_out, void *ws, size_t ws_bytes,
                    void *stream) {
  if (nnz <= 0) return GEOT_OK;
  if (!index || !keys_out || !perm_out || !ws) return geot_internal_fail(GEOT_EINVAL, "sort_index: null pointer");
  if (!geot_sort_supported(nnz, 0, key_max)) return geot_internal_fail(GEOT_EINVAL, "sort_index: keys must lie in [0, 2^32) and nnz < 2^32");
  hipStream_t st = static_cast<hipStream_t>(stream);
  size_t rp_bytes = 0;
  SORT_TRY(radix_pairs(nullptr, rp_bytes, nullptr, nullptr, nullptr, nullptr, nnz, 32, nullptr, nullptr, nullptr));
  const SortLayout L = sort_layout(nnz, rp_bytes);
  if (ws_bytes < L.total) return geot_internal_fail(GEOT_EWORKSPACE, "sort_index: workspace too small (geot_sort_workspace_bytes)");
  if (reinterpret_cast<uintptr_t>(ws) & 255) return geot_internal_fail(GEOT_EINVAL, "sort_index: workspace must be 256-byte aligned");
  char *base = static_cast<char *>(ws);
  uint32_t *k0 = reinterpret_cast<uint32_t *>(base + L.k0), *k1 = reinterpret_cast<uint32_t *>(base + L.k1);
  uint32_t *v0 = reinterpret_cast<uint32_t *>(base + L.v0), *v1 = reinterpret_cast<uint32_t *>(base + L.v1);
  int bits = 1;
  while (bits < 32 && (key_max >> bits) != 0) ++bits;
  const unsigned blocks = (unsigned)((nnz + 2 * kThreads - 1) / (2 * kThreads));
  if ((reinterpret_cast<uintptr_t>(index) & 15) == 0) hipLaunchKernelGGL(narrow_kernel<true>, dim3(blocks), dim3(kThreads), 0, st, index, k0, v0, nnz);
  else hipLaunchKernelGGL(narrow_kernel<false>, dim3(blocks), dim3(kThreads), 0, st, index, k0, v0, nnz);
  SORT_TRY(hipGetLastError());
  uint32_t *ks = nullptr, *vs = nullptr;
  size_t bytes = rp_bytes;
  SORT_TRY(radix_pairs(base + L.tmp, bytes, k0, k1, v0, v1, nnz, bits, st, &ks, &vs));
  if (((reinterpret_cast<uintptr_t>(keys_out) | reinterpret_cast<uintptr_t>(perm_out)) & 15) == 0)
    hipLaunchKernelGGL(widen_kernel<true>, dim3(blocks), dim3(kThreads), 0, st, ks, vs, keys_out, perm_out, nnz);
  else hipLaunchKernelGGL(widen_kernel<false>, dim3(blocks), dim3(kThreads), 0, st, ks, vs, keys_out, perm_out, nnz);
  SORT_TRY(hipGetLastError());
  return GEOT_OK;
}

} // extern "C"
