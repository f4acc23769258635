// kexp3 -- can gfx950 do 16-byte global loads / stores at addresses that are only 4-byte (or 2-byte) aligned, and at what rate?
// (rows whose width is not a whole 16-byte vector: F = 601 / 602 floats.)   hipcc -O3 --offload-arch=gfx950 tools/kexp3.hip -o tools/kexp3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void copy_rows(const float *src, float *dst, long rows, long F, int vec) {
  // one wave per row; lane c moves elements [4c, 4c+4) with ONE 16-byte access (whatever the row's alignment), tail lane clamped
  const long row = (long)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
  const int c = threadIdx.x & 63;
  if (row >= rows) return;
  for (long f0 = (long)c * 4; f0 < F; f0 += 256) {
    long off = f0;
    int shift = 0;
    if (f0 + 4 > F) { shift = (int)(f0 + 4 - F); off = F - 4; }
    if (vec) {
      const f4 x = *reinterpret_cast<const f4 *>(src + row * F + off);
      if (shift == 0) *reinterpret_cast<f4 *>(dst + row * F + off) = x;
      else for (int i = shift; i < 4; ++i) dst[row * F + off + i] = x[i];
    } else {
      for (int i = 0; i < 4; ++i) if (f0 + i < F) dst[row * F + f0 + i] = src[row * F + f0 + i];
    }
  }
}
int main() {
  for (long F : {600L, 601L, 602L, 603L, 608L, 130L, 129L}) {
    const long rows = 2000000000L / (F * 4) / 4;
    float *s, *d;
    hipMalloc(&s, rows * F * 4 + 64); hipMalloc(&d, rows * F * 4 + 64);
    std::vector<float> h(rows * F);
    for (long i = 0; i < rows * F; ++i) h[i] = (float)(i % 9973) * 0.5f;
    hipMemcpy(s, h.data(), rows * F * 4, hipMemcpyHostToDevice);
    for (int vec = 1; vec >= 0; --vec) {
      hipMemset(d, 0, rows * F * 4);
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      copy_rows<<<(rows + 3) / 4, 256>>>(s, d, rows, F, vec);
      hipEventRecord(a);
      for (int it = 0; it < 5; ++it) copy_rows<<<(rows + 3) / 4, 256>>>(s, d, rows, F, vec);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
      std::vector<float> o(rows * F);
      hipMemcpy(o.data(), d, rows * F * 4, hipMemcpyDeviceToHost);
      long bad = 0; for (long i = 0; i < rows * F; ++i) bad += o[i] != h[i];
      printf("F=%ld rows=%ld %s: %.3f ms = %.2f TB/s (read+write), mismatches %ld, hip error %d\n", F, rows, vec ? "16-byte accesses at 4-byte alignment" : "4-byte accesses",
             ms, 2.0 * rows * F * 4 / ms / 1e9, bad, (int)hipGetLastError());
    }
    hipFree(s); hipFree(d);
  }
  return 0;
}
