// VALU issue-rate probe for gfx950: how many cycles does one wave64 integer VALU instruction occupy a SIMD?
// Build: hipcc --offload-arch=gfx950 -O3 scripts/microbench/valu_peak.hip -o gpurun_out/valu_peak ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int CHAINS, int KIND>
__global__ void __launch_bounds__(1024) k_probe(uint32_t *out, int iters, uint32_t seed) {
  uint32_t x0[CHAINS], x1[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) { x0[c] = seed + threadIdx.x + c; x1[c] = seed * 3u + blockIdx.x + c; }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        if (KIND == 0) {  // Threefry round: add, rotate, xor
          x0[c] += x1[c];
          x1[c] = __builtin_amdgcn_alignbit(x1[c], x1[c], 32 - 13);
          x1[c] ^= x0[c];
        } else if (KIND == 1) {  // adds only
          x0[c] += x1[c]; x1[c] += x0[c]; x0[c] += x1[c];
        } else {  // xors/ands
          x0[c] ^= x1[c]; x1[c] = (x1[c] & x0[c]) ^ 0x9E3779B9u; x0[c] ^= x1[c] >> 3;
        }
      }
    }
  }
  uint32_t acc = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) acc ^= x0[c] ^ x1[c];
  if (acc == 0x12345678u) out[0] = acc;
}

template <int CHAINS, int KIND>
static void run(const char *name, int blocks, int threads, int ops_per_round) {
  uint32_t *d; hipMalloc(&d, 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 2000;
  hipLaunchKernelGGL((k_probe<CHAINS, KIND>), dim3(blocks), dim3(threads), 0, 0, d, 10, 1u);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((k_probe<CHAINS, KIND>), dim3(blocks), dim3(threads), 0, 0, d, iters, 1u);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double waves = (double)blocks * threads / 64.0;
  const double insts = waves * iters * 8.0 * CHAINS * ops_per_round;
  const double per_simd_per_s = insts / (ms * 1e-3) / 1024.0;
  printf("%-28s chains %d blocks %4d x %4d: %8.3f ms  %.3e wave-insts/s  -> %.2f cycles per wave-inst per SIMD at 2.4 GHz\n",
         name, CHAINS, blocks, threads, ms, insts / (ms * 1e-3), 2.4e9 / per_simd_per_s);
  hipFree(d);
}

int main() {
  run<4, 0>("threefry round", 512, 1024, 3);
  run<1, 0>("threefry round", 512, 1024, 3);
  run<4, 0>("threefry round (1 wave/SIMD)", 256, 256, 3);
  run<1, 0>("threefry round (1 wave/SIMD)", 256, 256, 3);
  run<4, 1>("adds", 512, 1024, 3);
  run<4, 2>("xor/and/shift", 512, 1024, 4);
  return 0;
}
