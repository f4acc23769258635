// kexp5 -- what does ONE wave-wide row-gather instruction cost a CU, whatever memory does with it?  Round 6's knock-out of the
// matrix-core SpMM (gathers dropped by the range check, no LDS writes, no transposed reads, no MFMAs) still ran 3.9 of 4.2 ms: the
// skeleton - one v_readlane + one buffer_load_dwordx2 per edge - is the floor of every "one row per wave-instruction" kernel of
// csrc/seg_slab.hip.  This program prices that instruction by itself:
//   width  8 / 16 bytes per lane (512-byte row per instruction / two rows per instruction)
//   table  dropped (descriptor with zero records: no memory traffic at all) | 1 MB (L1 / L2 resident) | 119 MB (configs[3]'s table)
//   waves  4 / 8 / 12 per CU, 8 or 16 loads in flight per lane
// and prints cycles per instruction per CU and the implied bytes per clock.
// hipcc -O3 --offload-arch=gfx950 tools/kexp5.hip -o tools/kexp5
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

template <int BYTES, int U>
__global__ __launch_bounds__(256) void rows(const void *table, uint32_t records, uint32_t row_mask, int steps, uint32_t *sink) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(table), 0, (int)records, 0x00020000);
  const int lane = threadIdx.x & 63;
  uint32_t x = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  uint32_t acc = 0;
  // a lane's offset inside the instruction's bytes: 8 B/lane = one 512-byte row per instruction; 16 B/lane = two rows (lanes 0..31 | 32..63)
  const uint32_t lane_off = BYTES == 8 ? lane * 8u : (lane & 31) * 16u;
  for (int s = 0; s < steps; ++s) {
    x = x * 1664525u + 1013904223u;                        // 64 fresh row numbers per step, one per lane
    const uint32_t my_off = ((x >> 8) & row_mask) << 9;    // row * 512
    for (int b = 0; b < 64; b += U) {
      if constexpr (BYTES == 8) {
        u2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
          v[u] = __builtin_bit_cast(u2, __builtin_amdgcn_raw_buffer_load_b64(rs, lane_off, (uint32_t)__builtin_amdgcn_readlane(my_off, (b + u) & 63), 0));
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u][0] + v[u][1];
      } else {
        u4 v[U / 2];
#pragma unroll
        for (int u = 0; u < U / 2; ++u) {                   // rows b + 2u (lanes 0..31) and b + 2u + 1 (lanes 32..63): a per-lane row offset
          const uint32_t ra = (uint32_t)__builtin_amdgcn_readlane(my_off, (b + 2 * u) & 63), rb = (uint32_t)__builtin_amdgcn_readlane(my_off, (b + 2 * u + 1) & 63);
          v[u] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off + (lane < 32 ? ra : rb), 0, 0));
        }
#pragma unroll
        for (int u = 0; u < U / 2; ++u) acc ^= v[u][0] + v[u][1] + v[u][2] + v[u][3];
      }
    }
  }
  if (acc == 0x9e3779b9u) sink[0] = acc;
}

int main() {
  void *table; uint32_t *sink;
  const size_t big = (size_t)232965 * 512;
  CK(hipMalloc(&table, big)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(table, 1, big));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const double ghz = prop.clockRate / 1e6;
  printf("# %s, %d CUs, %.2f GHz (reported)\n", prop.name, prop.multiProcessorCount, ghz);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  struct T { const char *name; uint32_t records, mask; } tables[] = {{"dropped (0 records)", 0, 0x3ffff}, {"1 MB table", 1u << 20, 2047}, {"119 MB table", (uint32_t)big, 0x1ffff}};
  for (auto &t : tables)
    for (int wg : {1, 2, 3})
      for (int cfg = 0; cfg < 4; ++cfg) {
        const int steps = 400;
        const int grid = 256 * wg;
        float best = 1e30f;
        for (int it = 0; it < 4; ++it) {
          CK(hipEventRecord(a));
          if (cfg == 0) rows<8, 8><<<grid, 256>>>(table, t.records, t.mask, steps, sink);
          else if (cfg == 1) rows<8, 16><<<grid, 256>>>(table, t.records, t.mask, steps, sink);
          else if (cfg == 2) rows<16, 8><<<grid, 256>>>(table, t.records, t.mask, steps, sink);
          else rows<16, 16><<<grid, 256>>>(table, t.records, t.mask, steps, sink);
          CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
          float ms; CK(hipEventElapsedTime(&ms, a, b));
          if (it > 0 && ms < best) best = ms;
        }
        CK(hipGetLastError());
        const int bytes = cfg < 2 ? 8 : 16, U = (cfg & 1) ? 16 : 8;
        const double instr_per_cu = (double)wg * 4 * steps * (bytes == 8 ? 64 : 32);          // wave-instructions per CU
        const double rows_per_cu = (double)wg * 4 * steps * 64;
        const double cyc = best * 1e-3 * 2.4e9;
        printf("%-20s %2d waves/CU  %2d B/lane U=%2d: %7.3f ms  %6.1f cycles per load instruction per CU (at 2.4 GHz)  %6.1f B/clk/CU  %5.2f TB/s chip-wide\n", t.name, wg * 4, bytes, U, best,
               cyc / instr_per_cu, rows_per_cu * 512 / cyc, rows_per_cu * 512 * 256 / (best * 1e-3) / 1e12);
      }
  printf("# done\n");
  return 0;
}
