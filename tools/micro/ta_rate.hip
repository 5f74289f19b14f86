// Microbenchmark (GPU box): how fast does a CU's vector-memory path deliver scattered 16-byte loads from an L1-resident table,
// and does that rate depend on the number of ACTIVE lanes of the wave? (What bounds the lean BVH walks: 64 B per lane-step in
// four 16-byte requests.)  hipcc --offload-arch=gfx950 -O3 -o ta_rate ta_rate.hip && ./ta_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// mode: which lanes are active. tableBytes: size of the table the loads scatter over. pairBytes: 64 = four loads of one 64-B record,
// 16 = one load per step.
template <int LOADS>
__global__ void __launch_bounds__(256) k_scatter(const uint4* __restrict__ table, uint32_t mask, uint32_t steps, unsigned long long laneMask,
                                                 uint32_t* out) {
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
  uint32_t acc = 0;
  if ((laneMask >> lane) & 1ull) {
    for (uint32_t s = 0; s < steps; s++) {
      x = x * 1664525u + 1013904223u;
      const uint32_t rec = ((x >> 8) & mask) * 4u;         // 64-byte records
      uint4 v[LOADS];
#pragma unroll
      for (int k = 0; k < LOADS; k++) v[k] = table[rec + k];
#pragma unroll
      for (int k = 0; k < LOADS; k++) acc += v[k].x ^ v[k].w;
      x += acc & 1u;                                       // (dependent: one step's loads complete before the next address)
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double clk = prop.clockRate * 1e3;       // Hz
  printf("%s: %d CUs, %.0f MHz\n", prop.name, cus, clk * 1e-6);
  uint32_t* out; CK(hipMalloc(&out, 4));
  const size_t maxBytes = 64u << 20;
  uint4* table; CK(hipMalloc(&table, maxBytes)); CK(hipMemset(table, 1, maxBytes));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  struct M { const char* name; unsigned long long m; };
  const M masks[] = {{"64 lanes", ~0ull}, {"32 lanes (even)", 0x5555555555555555ull}, {"32 lanes (low half)", 0xffffffffull},
                     {"16 lanes (every 4th)", 0x1111111111111111ull}, {"16 lanes (low quarter)", 0xffffull}, {"8 lanes (every 8th)", 0x0101010101010101ull}};
  const size_t sizes[] = {16u << 10, 2u << 20, 24u << 20};
  for (size_t tb : sizes) {
    const uint32_t recMask = uint32_t(tb / 64 - 1);
    for (int wavesPerSimd : {4, 7}) {
      for (const M& m : masks) {
        const uint32_t steps = 4000;
        const int blocks = cus * wavesPerSimd;           // 256 threads = 4 waves: one per SIMD
        hipLaunchKernelGGL(k_scatter<4>, dim3(blocks), dim3(256), 0, 0, table, recMask, 200u, m.m, out);
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(k_scatter<4>, dim3(blocks), dim3(256), 0, 0, table, recMask, steps, m.m, out);
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
        const double waveSteps = double(blocks) * 4 * steps;
        const double clkPerStepPerCU = ms * 1e-3 * clk / (waveSteps / cus);
        const int active = __builtin_popcountll(m.m);
        printf("table %6zu KB, %d waves/SIMD, %-24s: %7.2f ms, %6.1f clk per wave-step per CU, %6.1f B/clk/CU (active lanes' bytes)\n",
               tb >> 10, wavesPerSimd, m.name, ms, clkPerStepPerCU, active * 64.0 / clkPerStepPerCU);
      }
    }
  }
  return 0;
}
