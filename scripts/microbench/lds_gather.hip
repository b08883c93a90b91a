// lds_gather.hip - what a per-lane table read from LDS costs on gfx950, by width and address pattern: the column reads of the
// sparse-column evaluation (tsim_kernel4w.hip.h / tsim_wide.hip.h) are 16-byte reads at a random entry per lane.
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench/lds_gather.hip -o scripts/microbench/lds_gather.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
extern __shared__ uint32_t lds[];
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// MODE 0: b128, 1: b64, 2: b32.  `addr` = per-lane byte offsets (8 per lane, cycled), n_iter reads of 8 each.
template <int MODE>
__global__ void __launch_bounds__(1024) k_gather(const uint32_t *addr, uint32_t *sink, int n_iter, int table_bytes) {
  for (int i = threadIdx.x; i < table_bytes / 4; i += blockDim.x) lds[i] = i * 2654435761u;
  __syncthreads();
  uint32_t a[8];
  for (int k = 0; k < 8; ++k) a[k] = addr[(blockIdx.x * blockDim.x + threadIdx.x) * 8 + k];
  uint32_t acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
  const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds;
  const uint32_t mask = (uint32_t)table_bytes - (MODE == 0 ? 16u : MODE == 1 ? 8u : 4u), step = 7u * 64u * 16u;
  for (int it = 0; it < n_iter; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t ad = base + ((a[k] + (uint32_t)it * step) & mask);  // every lane moves by the same amount: the pattern stays what it is
      if (MODE == 0) {
        const u32x4 v = *(const __attribute__((address_space(3))) u32x4 *)(uintptr_t)ad;
        acc0 ^= v.x; acc1 ^= v.y; acc2 ^= v.z; acc3 ^= v.w;
      } else if (MODE == 1) {
        const u32x2 v = *(const __attribute__((address_space(3))) u32x2 *)(uintptr_t)ad;
        acc0 ^= v.x; acc1 ^= v.y;
      } else {
        acc0 ^= *(const __attribute__((address_space(3))) uint32_t *)(uintptr_t)ad;
      }
    }
  }
  if ((acc0 ^ acc1 ^ acc2 ^ acc3) == 0x12345u) sink[0] = acc0;
}

int main() {
  const int blocks = 256, threads = 1024, table = 32 * 1024, iters = 2000;
  uint32_t *d_addr, *d_sink;
  CK(hipMalloc(&d_addr, (size_t)blocks * threads * 8 * 4));
  CK(hipMalloc(&d_sink, 64));
  std::vector<uint32_t> h((size_t)blocks * threads * 8);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char *pat_name[] = {"all lanes one address (broadcast)", "random entry per lane", "random entry per lane, half the lanes on one shared entry", "lane-linear (conflict-free)"};
  for (int mode = 0; mode < 3; ++mode) {
    const int width = mode == 0 ? 16 : mode == 1 ? 8 : 4;
    for (int pat = 0; pat < 4; ++pat) {
      srand(1);
      for (size_t i = 0; i < h.size(); ++i) {
        const int lane = (int)((i / 8) % 64);
        uint32_t e = (uint32_t)(rand() % (table / width));
        if (pat == 0) e = 7;
        if (pat == 2 && (rand() & 1)) e = 200;
        if (pat == 3) e = (uint32_t)(lane + 64 * (int)(i % 8));
        h[i] = e * (uint32_t)width;
      }
      CK(hipMemcpy(d_addr, h.data(), h.size() * 4, hipMemcpyHostToDevice));
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(k_gather<0>, dim3(blocks), dim3(threads), table, 0, d_addr, d_sink, iters, table);
        if (mode == 1) hipLaunchKernelGGL(k_gather<1>, dim3(blocks), dim3(threads), table, 0, d_addr, d_sink, iters, table);
        if (mode == 2) hipLaunchKernelGGL(k_gather<2>, dim3(blocks), dim3(threads), table, 0, d_addr, d_sink, iters, table);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
      }
      // per CU: 16 waves x iters x 8 reads
      const double reads = 16.0 * iters * 8;
      printf("ds_read_b%-3d %-62s %7.1f us  -> %6.1f ns per wave-read per CU (%.1f cycles at 2.1 GHz), %6.1f B/clk/CU\n", width * 8, pat_name[pat], best * 1e3,
             best * 1e6 / reads, best * 1e6 / reads * 2.1, 64.0 * width / (best * 1e6 / reads * 2.1));
    }
  }
  return 0;
}
