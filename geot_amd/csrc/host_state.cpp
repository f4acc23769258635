// host_state.cpp -- options, statistics, small helpers, the per-(device, stream) workspace and the pinned read-back slot of the
// dispatcher plugin (see host.h for the map of the translation units).
#include "host.h"

namespace geot_host {

Options g_opt;
Stats g_stats;
std::mutex g_mu;
thread_local bool tl_capturing = false;

// ---- small helpers -------------------------------------------------------------------------------------------------------
int dtype_code(const at::Tensor &t, const char *op) {
  switch (t.scalar_type()) {
  case at::kFloat: return GEOT_F32;
  case at::kDouble: return GEOT_F64;
  case at::kHalf: return GEOT_F16;
  case at::kBFloat16: return GEOT_BF16;
  default: TORCH_CHECK(false, "\"", op, "\" not implemented for '", toString(t.scalar_type()), "'");
  }
}

int reduce_code(c10::string_view reduce, bool pyg_add) { // csrc/reduceutils.h:5-22 (+ PyG's 'add' for the gather ops)
  if (reduce == "max" || reduce == "amax") return GEOT_REDUCE_MAX;
  if (reduce == "mean") return GEOT_REDUCE_MEAN;
  if (reduce == "min" || reduce == "amin") return GEOT_REDUCE_MIN;
  if (reduce == "sum" || (pyg_add && reduce == "add")) return GEOT_REDUCE_SUM;
  if (reduce == "prod") return GEOT_REDUCE_PROD;
  TORCH_CHECK(false, "reduce argument must be either sum, prod, mean, amax or amin, got ", reduce);
}

void require_gpu(const char *op, std::initializer_list<const at::Tensor *> ts) {
  const at::Tensor *first = nullptr;
  for (const at::Tensor *t : ts) {
    if (!t || !t->defined()) continue;
    TORCH_CHECK(t->is_cuda(), "geot::", op, ": CPU tensors are not supported by geot_amd (MI355X-only package, no CPU "
                "fallback).  Move the tensors to the GPU.");
    if (!first) first = t;
    TORCH_CHECK(t->device() == first->device(), "all tensors must be on the same device");
  }
}

// (a ROCm build of PyTorch calls the GPU "cuda": the guard / stream types that accept that device type)
void *stream_of(const at::Tensor &t) { return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream(); }


// one zero-initialised workspace per (device, stream), grown on demand (the ABI: one stream at a time per workspace)
at::Tensor workspace(const at::Tensor &like, size_t bytes) {
  static thread_local std::map<std::pair<int, void *>, at::Tensor> ws;
  const std::pair<int, void *> key{(int)like.device().index(), stream_of(like)};
  if (tl_capturing) {
    auto it = ws.find(key);
    if (it != ws.end() && it->second.defined() && (size_t)it->second.numel() >= bytes) return it->second;   // made before the capture (warm-up on this stream)
    return at::zeros({(int64_t)std::max<size_t>(bytes, 1 << 20)}, like.options().dtype(at::kByte));          // this call's own; not kept
  }
  auto &w = ws[key];
  if (!w.defined() || (size_t)w.numel() < bytes)
    w = at::zeros({(int64_t)std::max<size_t>(bytes, 1 << 20)}, like.options().dtype(at::kByte));
  return w;
}

Slot &slot_for(int device) {
  static thread_local std::map<int, Slot> slots;
  Slot &s = slots[device];
  if (!s.host) {
    // fine-grained (coherent) pinned memory: a running kernel's stores become visible to the spinning host
    if (hipHostMalloc(reinterpret_cast<void **>(&s.host), kSlotWords * sizeof(int64_t), hipHostMallocCoherent) != hipSuccess) {
      (void)hipGetLastError();
      TORCH_CHECK(hipHostMalloc(reinterpret_cast<void **>(&s.host), kSlotWords * sizeof(int64_t), hipHostMallocDefault) == hipSuccess,
                  "hipHostMalloc failed");
    }
    std::memset(s.host, 0, kSlotWords * sizeof(int64_t));
    TORCH_CHECK(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming) == hipSuccess, "hipEventCreate failed");
  }
  return s;
}

} // namespace geot_host
