// cumask_probe.hip - which compute units does a stream created with hipExtStreamCreateWithCUMask use, and do two
// streams with complementary masks run side by side?  (the hard-row lane on its own CUs.)
//   hipcc --offload-arch=gfx950 -O2 scripts/microbench/cumask_probe.hip -o scripts/microbench/cumask_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_where(uint32_t *hist /* [8 xcc][64 slots] */, int spin) {
  uint32_t xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  // HW_ID (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
  const uint32_t cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
  if (threadIdx.x == 0) atomicAdd(&hist[(xcc & 7u) * 64u + ((se * 2u + sh) * 16u + cu) % 64u], 1u);
  uint32_t v = threadIdx.x;
  for (int i = 0; i < spin; ++i) v = v * 1664525u + 1013904223u;
  if (v == 0x12345u) hist[0] = v;
}

__global__ void k_busy(uint32_t *sink, int spin) {
  uint32_t v = threadIdx.x + blockIdx.x;
  for (int i = 0; i < spin; ++i) v = v * 1664525u + 1013904223u;
  if (v == 0x12345u) sink[0] = v;
}

static int count_cus(const std::vector<uint32_t> &h) { int n = 0; for (uint32_t v : h) n += v != 0; return n; }

int main() {
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  printf("device %s, %d CUs\n", pr.name, pr.multiProcessorCount);
  uint32_t *d; CK(hipMalloc(&d, 512 * 4));
  std::vector<uint32_t> h(512);
  auto run = [&](hipStream_t s, const char *what) {
    CK(hipMemsetAsync(d, 0, 512 * 4, s));
    hipLaunchKernelGGL(k_where, dim3(8192), dim3(256), 0, s, d, 2000);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), d, 512 * 4, hipMemcpyDeviceToHost));
    printf("%-40s distinct (xcc, se, sh, cu) slots used: %d; per XCC:", what, count_cus(h));
    for (int x = 0; x < 8; ++x) { int n = 0; for (int i = 0; i < 64; ++i) n += h[x * 64 + i] != 0; printf(" %d", n); }
    printf("\n");
  };
  hipStream_t s0; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  run(s0, "plain stream");
  const int words = (pr.multiProcessorCount + 31) / 32;
  for (int variant = 0; variant < 4; ++variant) {
    std::vector<uint32_t> m(words, 0u);
    char name[64];
    if (variant == 0) { for (int i = 0; i < 32; ++i) m[i / 32] |= 1u << (i % 32); snprintf(name, 64, "mask bits 0..31"); }
    if (variant == 1) { for (int i = 224; i < 256; ++i) m[i / 32] |= 1u << (i % 32); snprintf(name, 64, "mask bits 224..255"); }
    if (variant == 2) { for (int i = 0; i < 256; i += 8) m[i / 32] |= 1u << (i % 32); snprintf(name, 64, "mask bits 0, 8, 16, ... (32 bits)"); }
    if (variant == 3) { for (int i = 32; i < 256; ++i) m[i / 32] |= 1u << (i % 32); snprintf(name, 64, "mask bits 32..255"); }
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, m.data());
    if (e != hipSuccess) { printf("hipExtStreamCreateWithCUMask failed: %s\n", hipGetErrorString(e)); return 0; }
    run(s, name);
    CK(hipStreamDestroy(s));
  }
  // side by side: a long chip-filling kernel on the 224-CU stream, a small latency-bound one on the 32-CU stream
  std::vector<uint32_t> ma(words, 0u), mb(words, 0u);
  for (int i = 0; i < 32; ++i) ma[i / 32] |= 1u << (i % 32);
  for (int i = 32; i < 256; ++i) mb[i / 32] |= 1u << (i % 32);
  hipStream_t sa, sb, sp;
  CK(hipExtStreamCreateWithCUMask(&sa, (uint32_t)words, ma.data()));
  CK(hipExtStreamCreateWithCUMask(&sb, (uint32_t)words, mb.data()));
  CK(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto small_alone = [&](hipStream_t s) {
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL(k_busy, dim3(64), dim3(256), 0, s, d, 20000);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1000.f;
  };
  auto small_beside = [&](hipStream_t big, hipStream_t small) {
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_busy, dim3(256 * 64), dim3(1024), 0, big, d, 20000);  // several chip-fulls
    auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < 200.0) {}
    CK(hipEventRecord(e0, small));
    hipLaunchKernelGGL(k_busy, dim3(64), dim3(256), 0, small, d, 20000);
    CK(hipEventRecord(e1, small)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipDeviceSynchronize());
    return ms * 1000.f;
  };
  for (int rep = 0; rep < 3; ++rep) {
    printf("small kernel (64 blocks x 20000 iterations): alone plain %.1f us, alone masked(32) %.1f us | beside a chip-filling kernel: both plain %.1f us, big on 224 CUs + small on its own 32 CUs %.1f us, big plain + small masked %.1f us\n",
           small_alone(sp), small_alone(sa), small_beside(s0, sp), small_beside(sb, sa), small_beside(s0, sa));
  }
  return 0;
}
