// kexp6 -- does a hipMemsetAsync node of a captured HIP graph still zero its buffer on the 2nd, 3rd ... replay?  Round 6 found the first
// 64 KB of geot_slab_spmm's output (rows without edges: they rely on the call's memset) holding a repeating 16-byte pattern that looked
// like another kernel's argument block from the second replay of a captured call on - with an ordinary kernel launched between replays.
// hipcc -O3 --offload-arch=gfx950 tools/kexp6.hip -o tools/kexp6
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
__global__ void fill(unsigned short *p, size_t n, unsigned short v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void touch(float *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void zero16(uint4 *p, size_t n16) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = uint4{0, 0, 0, 0};
}
int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  float *aux; CK(hipMalloc(&aux, 64)); CK(hipMemset(aux, 0, 64));
  for (size_t bytes : {(size_t)4, (size_t)16640, (size_t)65536, (size_t)1280000, (size_t)(64u << 20)})
    for (int own_kernel = 0; own_kernel < 2; ++own_kernel) {
      unsigned short *buf; CK(hipMalloc(&buf, bytes + 64));
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      if (own_kernel && bytes % 16 == 0) zero16<<<256, 256, 0, st>>>(reinterpret_cast<uint4 *>(buf), bytes / 16);
      else CK(hipMemsetAsync(buf, 0, bytes, st));
      touch<<<1, 64, 0, st>>>(aux);
      CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      std::vector<unsigned short> h(bytes / 2 ? bytes / 2 : 1);
      printf("%9zu bytes, %s:", bytes, own_kernel && bytes % 16 == 0 ? "own zero kernel " : "hipMemsetAsync  ");
      for (int rep = 0; rep < 4; ++rep) {
        fill<<<512, 256, 0, st>>>(buf, bytes / 2 ? bytes / 2 : 1, (unsigned short)0x40E0);   // an ordinary kernel between replays
        CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(h.data(), buf, bytes >= 2 ? bytes : 2, hipMemcpyDeviceToHost));
        size_t bad = 0, first = (size_t)-1;
        for (size_t i = 0; i < bytes / 2; ++i) if (h[i] != 0) { if (first == (size_t)-1) first = i; ++bad; }
        if (bytes < 2) bad = (h[0] & 0xff) != 0;
        printf("  replay %d: %zu non-zero halfwords%s", rep, bad, bad ? "" : "");
        if (bad) printf(" (first at %zu: 0x%04x 0x%04x 0x%04x 0x%04x)", first, h[first], h[first + 1 < h.size() ? first + 1 : first], h[first + 2 < h.size() ? first + 2 : first], h[first + 3 < h.size() ? first + 3 : first]);
      }
      printf("\n");
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipFree(buf));
    }
  printf("# done\n");
  return 0;
}
