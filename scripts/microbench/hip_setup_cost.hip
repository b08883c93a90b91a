// What the per-handle set-up calls cost on this box (a fresh handle's first call took ~80 ms).
// hipcc --offload-arch=gfx950 -O2 hip_setup_cost.hip -o hip_setup_cost.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_nop(int *p) { if (p) p[0] = 1; }
int main() {
  hipSetDevice(0);
  hipFree(nullptr);
  hipStream_t s0;
  hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
  hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s0, nullptr);
  hipStreamSynchronize(s0);
  for (int rep = 0; rep < 2; ++rep) {
    double t = now();
    std::vector<hipStream_t> st(32);
    for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    printf("32 x hipStreamCreateWithFlags: %.0f us\n", now() - t);
    t = now();
    std::vector<hipEvent_t> ev(64);
    for (auto &e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    printf("64 x hipEventCreateWithFlags: %.0f us\n", now() - t);
    t = now();
    std::vector<void *> a(32), b(32);
    for (auto &q : a) hipMalloc(&q, 16640);
    printf("32 x hipMalloc(16 KB): %.0f us\n", now() - t);
    t = now();
    for (auto &q : b) hipMalloc(&q, 16 << 20);
    printf("32 x hipMalloc(16 MB): %.0f us\n", now() - t);
    t = now();
    void *big = nullptr;
    hipMalloc(&big, 512 << 20);
    printf("1 x hipMalloc(512 MB): %.0f us\n", now() - t);
    t = now();
    for (auto &q : a) { hipMemset(q, 0, 16640); hipMemset((char *)q + 8192, 0xFF, 4); hipMemset((char *)q + 16000, 0xFF, 4); }
    printf("96 x hipMemset (null stream): %.0f us\n", now() - t);
    t = now();
    for (auto &q : a) hipMemsetAsync(q, 0, 16640, s0);
    hipStreamSynchronize(s0);
    printf("32 x hipMemsetAsync + sync: %.0f us\n", now() - t);
    std::vector<char> host(32 * 16640, 0);
    void *one = nullptr;
    t = now();
    hipMalloc(&one, host.size());
    hipMemcpy(one, host.data(), host.size(), hipMemcpyHostToDevice);
    printf("1 x hipMalloc(520 KB) + hipMemcpy H2D: %.0f us\n", now() - t);
    t = now();
    for (auto &s : st) { hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s, nullptr); }
    for (auto &s : st) hipStreamSynchronize(s);
    printf("first kernel on each of 32 new streams + sync: %.0f us\n", now() - t);
    t = now();
    for (auto &q : a) hipFree(q);
    for (auto &q : b) hipFree(q);
    hipFree(big); hipFree(one);
    for (auto &s : st) hipStreamDestroy(s);
    for (auto &e : ev) hipEventDestroy(e);
    printf("free everything: %.0f us\n", now() - t);
  }
  return 0;
}
