// Microbenchmark (GPU box), companion of ta_rate.hip: scattered 64-byte records read as four 16-byte loads per lane from LDS
// instead of the vector L1, and a mix (a share of the lanes from LDS, the others from an L1-resident global table).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// ldsShare256: lanes whose per-step random byte is below it read from LDS, the others from the global table
__global__ void __launch_bounds__(256) k_mix(const uint4* __restrict__ table, uint32_t gmask, uint32_t steps, uint32_t ldsShare256,
                                             uint32_t ldsRecs, uint32_t* out) {
  extern __shared__ uint4 cache[];
  for (uint32_t i = threadIdx.x; i < ldsRecs * 4u; i += blockDim.x) cache[i] = table[i];
  __syncthreads();
  uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
  uint32_t acc = 0;
  for (uint32_t s = 0; s < steps; s++) {
    x = x * 1664525u + 1013904223u;
    uint4 v0, v1, v2, v3;
    if (((x >> 24) & 255u) < ldsShare256) {
      const uint32_t rec = ((x >> 8) % ldsRecs) * 4u;
      v0 = cache[rec]; v1 = cache[rec + 1]; v2 = cache[rec + 2]; v3 = cache[rec + 3];
    } else {
      const uint32_t rec = ((x >> 8) & gmask) * 4u;
      v0 = table[rec]; v1 = table[rec + 1]; v2 = table[rec + 2]; v3 = table[rec + 3];
    }
    acc += v0.x ^ v1.w ^ v2.y ^ v3.z;
    x += acc & 1u;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double clk = prop.clockRate * 1e3;
  uint32_t* out; CK(hipMalloc(&out, 4));
  uint4* table; CK(hipMalloc(&table, 64u << 20)); CK(hipMemset(table, 1, 64u << 20));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const uint32_t steps = 4000;
  for (size_t tb : {size_t(2u << 20), size_t(24u << 20)}) {
    for (uint32_t ldsKB : {16u, 32u}) {
      const int blocksPerCU = ldsKB == 16 ? 7 : 4;
      for (uint32_t share : {0u, 64u, 102u, 128u, 192u, 256u}) {
        const int blocks = cus * blocksPerCU;
        const uint32_t recs = ldsKB * 1024u / 64u;
        hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(256), ldsKB * 1024, 0, table, uint32_t(tb / 64 - 1), 100u, share, recs, out);
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(256), ldsKB * 1024, 0, table, uint32_t(tb / 64 - 1), steps, share, recs, out);
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
        const double clkPerStepPerCU = ms * 1e-3 * clk / (double(blocks) * 4 * steps / cus);
        printf("global table %5zu KB, LDS cache %2u KB x %d blocks/CU, LDS share %.2f: %7.2f ms, %6.1f clk per wave-step per CU\n",
               tb >> 10, ldsKB, blocksPerCU, share / 256.0, ms, clkPerStepPerCU);
      }
    }
  }
  return 0;
}
