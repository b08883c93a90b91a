// Does the chip run a short burst after an idle gap as fast as sustained work?  VALU-bound kernel of ~50 us, launched
// back to back n times after an idle gap; per-kernel durations from HIP events.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <unistd.h>
#include <vector>
__global__ void __launch_bounds__(1024) k_burn(uint32_t *out, int iters) {
  uint32_t x0 = threadIdx.x, x1 = blockIdx.x * 3u + 1u;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r) { x0 += x1; x1 = __builtin_amdgcn_alignbit(x1, x1, 19); x1 ^= x0; }
  }
  if ((x0 ^ x1) == 0x12345678u) out[0] = x0;
}
int main() {
  uint32_t *d; (void)hipMalloc(&d, 4);
  const int n = 400;
  std::vector<hipEvent_t> ev(n + 1);
  for (auto &e : ev) (void)hipEventCreate(&e);
  for (int idle_us : {0, 50, 1000, 100000}) {
    hipLaunchKernelGGL(k_burn, dim3(512), dim3(1024), 0, 0, d, 40); (void)hipDeviceSynchronize();
    for (int w = 0; w < 2000; ++w) hipLaunchKernelGGL(k_burn, dim3(512), dim3(1024), 0, 0, d, 40);  // sustained load first
    (void)hipDeviceSynchronize();
    usleep(idle_us);
    (void)hipEventRecord(ev[0]);
    for (int i = 0; i < n; ++i) { hipLaunchKernelGGL(k_burn, dim3(512), dim3(1024), 0, 0, d, 40); (void)hipEventRecord(ev[i + 1]); }
    (void)hipDeviceSynchronize();
    printf("idle %6d us before the burst: kernel time (us) #1..8:", idle_us);
    float ms;
    for (int i = 0; i < 8; ++i) { (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]); printf(" %.1f", ms * 1e3); }
    double a = 0, b = 0, c = 0;
    for (int i = 10; i < 30; ++i) { (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]); a += ms; }
    for (int i = 100; i < 120; ++i) { (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]); b += ms; }
    for (int i = 380; i < 400; ++i) { (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]); c += ms; }
    printf("  | avg #10-30: %.1f  #100-120: %.1f  #380-400: %.1f\n", a / 20 * 1e3, b / 20 * 1e3, c / 20 * 1e3);
  }
  return 0;
}
