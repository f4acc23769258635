// probe: v_dot2c_f32_bf16 / v_dot2c_f32_f16 semantics on gfx950 against fp32 arithmetic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
__global__ void k(const uint32_t *a, const uint32_t *b, const float *c, float *o_bf, float *o_h, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float acc = c[i];
  asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc) : "v"(a[i]), "v"(b[i]));
  o_bf[i] = acc;
  float acc2 = c[i];
  asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(acc2) : "v"(a[i]), "v"(b[i]));
  o_h[i] = acc2;
}
static float bf(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static float hf(uint16_t h) { _Float16 x; memcpy(&x, &h, 2); return (float)x; }
int main() {
  const int n = 1 << 16;
  std::vector<uint32_t> a(n), b(n); std::vector<float> c(n), ob(n), oh(n);
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
  for (int i = 0; i < n; ++i) {
    // moderate exponents so that products are normal numbers; a few special cases at the front
    auto mk = [&]() { uint32_t r = rnd(); uint16_t m = r & 0x7f, e = 120 + (r >> 8) % 16, sg = (r >> 20) & 1; return (uint16_t)((sg << 15) | (e << 7) | m); };
    a[i] = mk() | ((uint32_t)mk() << 16); b[i] = mk() | ((uint32_t)mk() << 16);
    uint32_t r = rnd(); float f; uint32_t u = ((r & 1) << 31) | ((120 + (r >> 4) % 16) << 23) | (rnd() & 0x7fffff); memcpy(&f, &u, 4); c[i] = f;
  }
  a[0] = 0x7f803f80; b[0] = 0x00003f80; c[0] = 1.f;     // a = (1.0, +Inf), b = (1.0, 0): Inf * 0
  a[1] = 0x00013f80; b[1] = 0x3f803f80; c[1] = 0.f;     // a.hi = smallest bf16 denormal
  a[2] = 0x3f803f80; b[2] = 0x3f803f80; c[2] = 16777216.f; // 1 + 1 + 2^24: rounding of the sum
  uint32_t *da, *db; float *dc, *dob, *doh;
  hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dob, n * 4); hipMalloc(&doh, n * 4);
  hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, db, dc, dob, doh, n);
  hipMemcpy(ob.data(), dob, n * 4, hipMemcpyDeviceToHost); hipMemcpy(oh.data(), doh, n * 4, hipMemcpyDeviceToHost);
  printf("special: inf*0 -> %g ; denormal a.hi*1 -> %g (exact %g) ; 1+1+2^24 -> %.1f\n", ob[0], ob[1], 1.f + bf(1), ob[2]);
  int exact_seq = 0, exact_fused = 0, exact_pairfirst = 0; double maxrel = 0, maxn_dot = 0, maxn_seq = 0;
  for (int i = 3; i < n; ++i) {
    float a0 = bf(a[i] & 0xffff), a1 = bf(a[i] >> 16), b0 = bf(b[i] & 0xffff), b1 = bf(b[i] >> 16);
    float seq = fmaf(a1, b1, fmaf(a0, b0, c[i]));                 // what the kernels do today (lo first)
    double ex = (double)a0 * b0 + (double)a1 * b1 + (double)c[i];
    float fused = (float)ex;                                       // one rounding of the exact sum
    float pf = (float)((double)a0 * b0 + (double)a1 * b1) + c[i];  // products summed (rounded), then + c
    exact_seq += ob[i] == seq; exact_fused += ob[i] == fused; exact_pairfirst += ob[i] == pf;
    maxrel = fmax(maxrel, fabs((double)ob[i] - ex) / fmax(1e-30, fabs(ex)));
    const double scale = fmax(fmax(fabs((double)a0 * b0), fabs((double)a1 * b1)), fabs((double)c[i]));
    maxn_dot = fmax(maxn_dot, fabs((double)ob[i] - ex) / scale);
    maxn_seq = fmax(maxn_seq, fabs((double)seq - ex) / scale);
  }
  printf("bf16: max |err| / max(|p0|,|p1|,|c|): dot2c %.3g, sequential fma %.3g (2^-24 = %.3g)\n", maxn_dot, maxn_seq, 1.0 / 16777216.0);
  printf("bf16: of %d: == sequential fma %d, == single rounding of exact %d, == (p0+p1 rounded)+c %d ; max rel err vs exact %.3g\n", n - 3, exact_seq, exact_fused, exact_pairfirst, maxrel);
  int hseq = 0; double hmax = 0;
  for (int i = 3; i < n; ++i) {
    float a0 = hf(a[i] & 0xffff), a1 = hf(a[i] >> 16), b0 = hf(b[i] & 0xffff), b1 = hf(b[i] >> 16);
    if (!std::isfinite(a0) || !std::isfinite(a1) || !std::isfinite(b0) || !std::isfinite(b1)) continue;
    float seq = fmaf(a1, b1, fmaf(a0, b0, c[i]));
    double ex = (double)a0 * b0 + (double)a1 * b1 + (double)c[i];
    hseq += oh[i] == seq; hmax = fmax(hmax, fabs((double)oh[i] - ex) / fmax(1e-30, fabs(ex)));
  }
  printf("f16: == sequential fma %d ; max rel err vs exact %.3g\n", hseq, hmax);
  return 0;
}
