// ref_harness.cpp -- C-ABI doorway into the REFERENCE's own CPU index_scatter.
//
// TEST INFRASTRUCTURE ONLY (see oracle/README.md).  This file is ours; it is compiled
// TOGETHER WITH the reference source file csrc/cpu/index_scatter_cpu.cpp, taken from
// where it lies under /root/reference (never copied into this repo), into
// oracle/_ref/libgeot_ref*.so by oracle/Makefile.  It wraps raw pointers in at::Tensor
// views and calls the reference entry point
//     at::Tensor index_scatter_cpu(self, dim, index, src, reduce, sorted)
// declared at csrc/cpu/index_scatter_cpu.h:4-6 and defined at
// csrc/cpu/index_scatter_cpu.cpp:136-155.  The zero-fill + "rows = index[-1]+1" rule of
// the dispatcher shim (csrc/index_scatter.cpp:11-24) is restated here because that file
// also references the CUDA entry point, which cannot be built in this image.
#include "cpu/index_scatter_cpu.h"

#include <ATen/Parallel.h>
#include <cstdint>
#include <cstring>
#include <string>

static thread_local std::string g_err;

extern "C" {

const char *geot_ref_last_error() { return g_err.c_str(); }

int geot_ref_num_threads() { return at::get_num_threads(); }

void geot_ref_set_num_threads(int n) {
  if (n > 0) at::set_num_threads(n);
}

// dtype: 0 = float32, 1 = float64, 2 = float16, 3 = bfloat16 (the set the reference's CPU path
// dispatches, csrc/cpu/index_scatter_cpu.cpp:127-133).  out must hold K*F elements; it is zero-filled here
// exactly as torch::zeros does in csrc/index_scatter.cpp:21.
int geot_ref_index_scatter_cpu(const int64_t *index, const void *src, void *out, int64_t nnz,
                               int64_t F, int64_t K, int dtype, const char *reduce,
                               int sorted) {
  try {
    auto st = dtype == 0 ? at::kFloat : dtype == 1 ? at::kDouble : dtype == 2 ? at::kHalf : at::kBFloat16;
    size_t esz = dtype == 0 ? 4 : dtype == 1 ? 8 : 2;
    auto opts = at::TensorOptions().dtype(st).device(at::kCPU);
    at::Tensor idx = at::from_blob(const_cast<int64_t *>(index), {nnz},
                                   at::TensorOptions().dtype(at::kLong));
    at::Tensor s = at::from_blob(const_cast<void *>(src), {nnz, F}, opts);
    at::Tensor o = at::from_blob(out, {K, F}, opts);
    std::memset(out, 0, static_cast<size_t>(K * F) * esz);
    index_scatter_cpu(o, /*dim=*/0, idx, s, reduce, sorted != 0);
    return 0;
  } catch (const std::exception &e) {
    g_err = e.what();
    return -1;
  }
}

} // extern "C"
