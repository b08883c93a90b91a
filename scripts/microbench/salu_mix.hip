// How do scalar instructions share issue with vector ones on gfx950?  Each wave runs ITER iterations of
// NV full-rate v_add_u32 (8 independent chains) + NS s_add_u32 (4 independent chains, asm volatile), 8 waves per SIMD.
// If the scalar unit had issue bandwidth of its own, time would not move with NS until it saturates.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/microbench/salu_mix.hip -o scripts/microbench/salu_mix.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define ITER 2048
template <int NV, int NS, int KIND>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t s0) {
  uint32_t a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i * 40503u + s0;
  uint32_t s[4] = {s0, s0 + 1, s0 + 2, s0 + 3};
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int r = 0; r < (NV > NS ? NV : NS); ++r) {
      if (r < NV) {
        if (KIND == 0) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[r & 7]) : "v"(a[(r + 3) & 7]));
        else asm volatile("v_alignbit_b32 %0, %0, %1, 19" : "+v"(a[r & 7]) : "v"(a[(r + 3) & 7]));
      }
      if (r < NS) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s[r & 3]) : "s"(s[(r + 1) & 3]) : "scc");
    }
  }
  uint32_t r = s[0] ^ s[1] ^ s[2] ^ s[3];
#pragma unroll
  for (int i = 0; i < 8; ++i) r ^= a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int NV, int NS, int KIND>
void run(uint32_t *d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NV, NS, KIND><<<2048, 256>>>(d, 1); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) k<NV, NS, KIND><<<2048, 256>>>(d, 1);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double iters_per_simd = 2048.0 * 4 * ITER / 1024.0;  // wave-iterations per SIMD
  printf("%s NV %2d NS %2d: %.4f ms  %.1f ns per wave-iteration per SIMD  (%.2f ns per VALU, %.2f ns per SALU if alone)\n",
         KIND ? "alignbit" : "v_add   ", NV, NS, ms, ms * 1e6 / iters_per_simd, NV ? ms * 1e6 / iters_per_simd / NV : 0.0, NS ? ms * 1e6 / iters_per_simd / NS : 0.0);
}
int main() {
  uint32_t *d; hipMalloc(&d, 2048 * 256 * 4);
  run<16, 0, 0>(d); run<16, 2, 0>(d); run<16, 4, 0>(d); run<16, 8, 0>(d); run<16, 16, 0>(d); run<0, 16, 0>(d); run<0, 32, 0>(d);
  run<16, 0, 1>(d); run<16, 4, 1>(d); run<16, 8, 1>(d); run<16, 16, 1>(d); run<8, 16, 1>(d);
  return 0;
}
