// torch_shim.cpp -- optional PyTorch dispatcher plugin (`geot_amd/_C.so`) over the C ABI.
//
// This is the C++ form of the drop-in: it registers, in the dispatcher namespace `geot`, exactly the
// schemas the reference's `_C` extension registers and implements their CUDA (= ROCm GPU) key by calling
// libgeot_hip.so.  With it, the REFERENCE's unmodified Python package (geot/*.py: torch.ops.load_library
// on `_C`, geot/__init__.py:12-19) runs on MI355X.  It replaces:
//   csrc/index_scatter.cpp:26-56, csrc/gather_scatter.cpp:13-34,114-117,
//   csrc/gather_weight_scatter.cpp:11-49, csrc/mh_spmm.cpp:10-23, csrc/csr_gws.cpp:11-60
// (the shims) together with csrc/cuda/*_cuda.cu (the entry points they call).
// It must NOT be loaded next to geot_amd/ops.py, which registers the same schemas from Python.
//
// Build (Makefile target `shim`): g++ -shared -fPIC torch_shim.cpp -I<torch>/include -I/opt/rocm/include
//        -D__HIP_PLATFORM_AMD__ -L<repo>/geot_amd -lgeot_hip -ltorch -ltorch_cpu -lc10 -lc10_hip
#include <ATen/ATen.h>
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <map>

#include "geot_hip.h"

namespace {

int dtype_code(const at::Tensor &t, const char *op) {
  switch (t.scalar_type()) {
  case at::kFloat: return GEOT_F32;
  case at::kDouble: return GEOT_F64;
  case at::kHalf: return GEOT_F16;
  case at::kBFloat16: return GEOT_BF16;
  default: TORCH_CHECK(false, "\"", op, "\" not implemented for '", toString(t.scalar_type()), "'");
  }
}

void *current_stream(const at::Tensor &t) {
  return c10::hip::getCurrentHIPStream(t.device().index()).stream();
}

// one zero-initialised workspace per (device, stream), grown on demand: the ABI allows a workspace to be
// used by one stream at a time (control words and carries live in it) and wants the control words zero once
at::Tensor &workspace(const at::Tensor &like, size_t bytes) {
  static thread_local std::map<std::pair<int, void *>, at::Tensor> ws;
  auto &w = ws[{(int)like.device().index(), current_stream(like)}];
  if (!w.defined() || (size_t)w.numel() < bytes)
    w = at::zeros({(int64_t)std::max<size_t>(bytes, 1 << 20)}, like.options().dtype(at::kByte));
  return w;
}

// the library launches on the HIP current device: make it the tensors' device for the duration of the call
#define GEOT_DEVICE_GUARD(t) const c10::hip::HIPGuard geot_device_guard_((t).device())

void check_weight(const at::Tensor &weight, const at::Tensor &src) {
  TORCH_CHECK(weight.scalar_type() == src.scalar_type(), "expected weight of dtype ", toString(src.scalar_type()),
              " but found ", toString(weight.scalar_type()));
}

int reduce_code(c10::string_view reduce) { // csrc/reduceutils.h:5-22
  if (reduce == "max" || reduce == "amax") return GEOT_REDUCE_MAX;
  if (reduce == "mean") return GEOT_REDUCE_MEAN;
  if (reduce == "min" || reduce == "amin") return GEOT_REDUCE_MIN;
  if (reduce == "sum") return GEOT_REDUCE_SUM;
  if (reduce == "prod") return GEOT_REDUCE_PROD;
  TORCH_CHECK(false, "reduce argument must be either sum, prod, mean, amax or amin, got ", reduce);
}

int64_t rows_from_last(const at::Tensor &index) { return index[-1].item<int64_t>() + 1; } // csrc/index_scatter.cpp:30

at::Tensor index_scatter_impl(const int64_t dim, at::Tensor index, at::Tensor src, const c10::string_view reduce,
                              const bool sorted) {
  TORCH_CHECK(dim >= 0 && dim < src.dim(), "dim must be non-negative and less than input dimensions");
  GEOT_DEVICE_GUARD(src);
  TORCH_CHECK(index.dim() == 1, "index must be 1 dimensional");
  TORCH_CHECK(src.size(dim) == index.size(0), "index length must be equal to src dimension size");
  const int red = reduce_code(reduce);
  at::Tensor moved = (dim == 0 ? src : src.movedim(dim, 0)).contiguous();
  index = index.contiguous();
  int64_t rows;
  bool ascending = sorted;
  if (sorted) {
    rows = rows_from_last(index);
  } else {
    // sorted=False promises nothing: one pass over the index yields the row count and the number of
    // descents in the same 16-byte read-back; an ascending index is served by the atomic-free kernels
    // (the reference's own test and benchmark pass sorted=False with a sorted index)
    TORCH_CHECK(index.numel() > 0, "index out of range: index[-1] of an empty index");
    at::Tensor probe = at::empty({2}, index.options());
    int prc = geot_index_probe(index.data_ptr<int64_t>(), index.numel(), probe.data_ptr<int64_t>(), current_stream(src));
    TORCH_CHECK(prc == GEOT_OK, geot_last_error());
    at::Tensor host = probe.cpu();
    rows = host.data_ptr<int64_t>()[0] + 1;
    ascending = host.data_ptr<int64_t>()[1] == 0;
  }
  TORCH_CHECK(ascending || red == GEOT_REDUCE_SUM, "an unsorted index supports reduce='sum' only");
  auto shape = moved.sizes().vec();
  shape[0] = rows;
  at::Tensor out = at::empty(shape, moved.options());
  const int64_t nnz = index.numel(), feat = nnz ? moved.numel() / nnz : 0;
  const int dt = dtype_code(src, "index_scatter_sorted");
  auto &ws = workspace(src, geot_workspace_bytes(nnz, feat, rows, dt));
  int rc = red == GEOT_REDUCE_SUM
               ? geot_index_scatter(index.data_ptr<int64_t>(), moved.data_ptr(), out.data_ptr(), nnz, feat, rows, dt,
                                    ascending, ws.data_ptr(), ws.numel(), current_stream(src))
               : geot_index_scatter_reduce(index.data_ptr<int64_t>(), moved.data_ptr(), out.data_ptr(), nnz, feat,
                                           rows, dt, red, ws.data_ptr(), ws.numel(), current_stream(src));
  TORCH_CHECK(rc == GEOT_OK, geot_last_error());
  return dim == 0 ? out : out.movedim(0, dim);
}

void check_gather(const at::Tensor &si, const at::Tensor &di, const at::Tensor &src, int64_t ndim) {
  TORCH_CHECK(si.dim() == di.dim() && si.dim() == 1, "src_index and dst_index must be 1 dimensional");
  TORCH_CHECK(src.dim() == ndim, "src must be ", ndim, " dimensional");
}

at::Tensor gather_scatter_impl(at::Tensor si, at::Tensor di, at::Tensor src) {
  check_gather(si, di, src, 2);
  GEOT_DEVICE_GUARD(src);
  const int64_t rows = rows_from_last(di), nnz = di.numel(), feat = src.size(1);
  si = si.contiguous(); di = di.contiguous(); src = src.contiguous();
  at::Tensor out = at::empty({rows, feat}, src.options());
  const int dt = dtype_code(src, "gather_scatter_sorted");
  auto &ws = workspace(src, geot_workspace_bytes(nnz, feat, rows, dt));
  int rc = geot_gather_scatter(si.data_ptr<int64_t>(), di.data_ptr<int64_t>(), src.data_ptr(), out.data_ptr(), nnz, feat,
                               src.size(0), rows, dt, ws.data_ptr(), ws.numel(), current_stream(src));
  TORCH_CHECK(rc == GEOT_OK, geot_last_error());
  return out;
}

at::Tensor gather_weight_scatter_impl(at::Tensor si, at::Tensor di, at::Tensor weight, at::Tensor src) {
  check_gather(si, di, src, 2);
  check_weight(weight, src);
  GEOT_DEVICE_GUARD(src);
  const int64_t rows = rows_from_last(di), nnz = di.numel(), feat = src.size(1);
  si = si.contiguous(); di = di.contiguous(); src = src.contiguous(); weight = weight.contiguous();
  at::Tensor out = at::empty({rows, feat}, src.options());
  const int dt = dtype_code(src, "gather_weight_scatter_sorted");
  auto &ws = workspace(src, geot_workspace_bytes(nnz, feat, rows, dt));
  int rc = geot_gather_weight_scatter(si.data_ptr<int64_t>(), di.data_ptr<int64_t>(), weight.data_ptr(), src.data_ptr(),
                                      out.data_ptr(), nnz, feat, src.size(0), rows, dt, ws.data_ptr(), ws.numel(),
                                      current_stream(src));
  TORCH_CHECK(rc == GEOT_OK, geot_last_error());
  return out;
}

at::Tensor sddmm_coo_impl(at::Tensor si, at::Tensor di, at::Tensor m1, at::Tensor m2) {
  GEOT_DEVICE_GUARD(m1);
  // the reference's Python wrapper hands over int32 indices (geot/gather_weight_scatter.py:10-11)
  si = si.to(at::kLong).contiguous(); di = di.to(at::kLong).contiguous();
  m1 = m1.contiguous(); m2 = m2.contiguous();
  at::Tensor out = at::empty({di.size(0)}, m1.options());
  int rc = geot_sddmm_coo(si.data_ptr<int64_t>(), di.data_ptr<int64_t>(), m1.data_ptr(), m2.data_ptr(), out.data_ptr(),
                          di.size(0), m1.size(1), m1.size(0), m2.size(0), dtype_code(m1, "sddmm_coo"), current_stream(m1));
  TORCH_CHECK(rc == GEOT_OK, geot_last_error());
  return out;
}

at::Tensor csr_gws_impl(at::Tensor indptr, at::Tensor indices, at::Tensor weight, at::Tensor src) {
  check_weight(weight, src);
  GEOT_DEVICE_GUARD(src);
  indptr = indptr.to(at::kLong).contiguous(); indices = indices.to(at::kLong).contiguous();
  weight = weight.contiguous(); src = src.contiguous();
  const int64_t rows = indptr.size(0), nnz = indices.size(0), feat = src.size(1); // csrc/csr_gws.cpp:29-31
  at::Tensor out = at::empty({rows, feat}, src.options());
  const int dt = dtype_code(src, "csr_gws");
  auto &ws = workspace(src, geot_csr_workspace_bytes(nnz, feat, rows, dt));
  int rc = geot_csr_gws(indptr.data_ptr<int64_t>(), indices.data_ptr<int64_t>(), weight.data_ptr(), src.data_ptr(),
                        out.data_ptr(), rows - 1, nnz, feat, src.size(0), rows, dt, ws.data_ptr(), ws.numel(),
                        current_stream(src));
  TORCH_CHECK(rc == GEOT_OK, geot_last_error());
  return out;
}

at::Tensor mh_spmm_impl(at::Tensor si, at::Tensor di, at::Tensor weight, at::Tensor src, const c10::string_view reduce) {
  check_gather(si, di, src, 3);
  check_weight(weight, src);
  GEOT_DEVICE_GUARD(src);
  TORCH_CHECK(reduce_code(reduce) == GEOT_REDUCE_SUM, "mh_spmm: only 'sum' is implemented");
  const int64_t nnz = si.size(0), rows = rows_from_last(di);
  int layout;
  if (weight.dim() == 2 && weight.size(0) == nnz && weight.size(1) == src.size(1)) layout = GEOT_W_EDGE_MAJOR;
  else if (weight.dim() == 2 && weight.size(1) == nnz && weight.size(0) == src.size(1)) layout = GEOT_W_HEAD_MAJOR;
  else throw std::runtime_error("Invalid weight size"); // csrc/cuda/wrapper/mh_spmm_base.h:49
  si = si.contiguous(); di = di.contiguous(); weight = weight.contiguous(); src = src.contiguous();
  at::Tensor out = at::empty({rows, src.size(1), src.size(2)}, src.options());
  const int dt = dtype_code(src, "mh_spmm_sorted");
  auto &ws = workspace(src, geot_mh_workspace_bytes(nnz, src.size(1), src.size(2), rows, dt));
  int rc = geot_mh_spmm(si.data_ptr<int64_t>(), di.data_ptr<int64_t>(), weight.data_ptr(), src.data_ptr(), out.data_ptr(),
                        nnz, src.size(1), src.size(2), src.size(0), rows, layout, dt, ws.data_ptr(), ws.numel(),
                        current_stream(src));
  TORCH_CHECK(rc == GEOT_OK, geot_last_error());
  return out;
}

} // namespace

// same schema strings as csrc/index_scatter.cpp:43-47, gather_scatter.cpp:16-17, gather_weight_scatter.cpp:12-16,
// csr_gws.cpp:12-13; mh_spmm is a catch-all def with an inferred schema like csrc/mh_spmm.cpp:23
TORCH_LIBRARY_FRAGMENT(geot, m) {
  m.def("index_scatter(int dim, Tensor index, Tensor src, str reduce, bool sorted)->Tensor ");
  m.def("gather_scatter_impl(Tensor src_index, Tensor dst_index, Tensor src) -> Tensor");
  m.def("gather_weight_scatter_impl(Tensor src_index, Tensor dst_index, Tensor weight, Tensor src) -> Tensor");
  m.def("sddmm_coo_impl(Tensor src_index, Tensor dst_index, Tensor mat_1, Tensor mat_2) -> Tensor");
  m.def("csr_gws_impl(Tensor indptr, Tensor indices, Tensor weight, Tensor src) -> Tensor");
  m.def("mh_spmm", mh_spmm_impl);
}

TORCH_LIBRARY_IMPL(geot, CUDA, m) {
  m.impl("index_scatter", index_scatter_impl);
  m.impl("gather_scatter_impl", gather_scatter_impl);
  m.impl("gather_weight_scatter_impl", gather_weight_scatter_impl);
  m.impl("sddmm_coo_impl", sddmm_coo_impl);
  m.impl("csr_gws_impl", csr_gws_impl);
}
