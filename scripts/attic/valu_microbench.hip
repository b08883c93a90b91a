// Micro-benchmark: sustained wave64 issue rate of the integer VALU ops the sampling kernel uses
// (gfx950).  Build: hipcc --offload-arch=gfx950 -O3 scripts/valu_microbench.hip -o gpurun_out/valu_mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define ITER 2048
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t s0, uint32_t s1) {
  uint32_t a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i * 40503u + s0;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      uint32_t x = a[i], y = a[(i + 3) & 7];
      if (OP == 0) asm volatile("v_and_b32 %0, %1, %2" : "=v"(x) : "s"(s1), "v"(x));
      if (OP == 1) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(x));
      if (OP == 2) asm volatile("v_bcnt_u32_b32 %0, %1, %2" : "=v"(x) : "v"(x), "s"(s1));
      if (OP == 3) asm volatile("v_add_u32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(x));
      if (OP == 4) asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(x) : "v"(y), "v"(x), "s"(s1));
      if (OP == 5) asm volatile("v_ashrrev_i32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(x));
      if (OP == 6) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(x));
      if (OP == 7) asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(x) : "v"(y), "s"(s1), "v"(x));
      if (OP == 8) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x) : "v"(y), "v"(x));
      if (OP == 9) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x) : "v"(y), "v"(x), "v"(x));
      if (OP == 10) asm volatile("v_or3_b32 %0, %1, %2, %3" : "=v"(x) : "v"(y), "v"(x), "s"(s1));
      if (OP == 11) asm volatile("v_lshlrev_b32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(x));
      if (OP == 12) asm volatile("v_sub_u32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(x));
      if (OP == 13) asm volatile("v_cmp_ne_u32 vcc, %1, %2\n v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x) : "v"(y), "v"(x) : "vcc");
      if (OP == 14) asm volatile("v_xad_u32 %0, %1, %2, %3" : "=v"(x) : "v"(y), "v"(x), "s"(s1));
      if (OP == 15) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(x) : "v"(y), "s"(s1), "v"(x));
      if (OP == 16) asm volatile("v_and_b32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(x));
      if (OP == 17) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "s"(s1), "v"(x));
      if (OP == 18) asm volatile("v_bcnt_u32_b32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y));
      if (OP == 19) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "s"(s1));
      if (OP == 20) asm volatile("v_lshrrev_b32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(x));
      if (OP == 21) asm volatile("v_bfe_u32 %0, %1, %2, 1" : "=v"(x) : "v"(x), "v"(y));
      if (OP == 22) asm volatile("v_and_b32 %0, 1, %1" : "=v"(x) : "v"(x));
      if (OP == 23) asm volatile("v_add_u32 %0, %1, %2" : "=v"(x) : "s"(s1), "v"(x));
      if (OP == 24) asm volatile("v_lshlrev_b32 %0, 1, %1" : "=v"(x) : "v"(x));
      if (OP == 25) asm volatile("v_ashrrev_i32 %0, 1, %1" : "=v"(x) : "v"(x));
      if (OP == 26) asm volatile("v_or_b32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(x));
      if (OP == 27) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(y));
      if (OP == 28) asm volatile("v_pk_add_u16 %0, %1, %2" : "=v"(x) : "v"(y), "v"(x));
      if (OP == 29) asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(x) : "v"(y), "v"(x));
      if (OP == 30) asm volatile("v_subrev_u32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(x));
      if (OP == 31) asm volatile("v_max_i32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(x));
      if (OP == 32) asm volatile("v_add_u32 %0, 5, %1" : "=v"(x) : "v"(x));
      if (OP == 33) asm volatile("v_xor_b32 %0, 0x12345678, %1" : "=v"(x) : "v"(x));
      if (OP == 34) asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(x) : "v"(y), "v"(x), "v"(y));
      if (OP == 35) asm volatile("v_and_b32 %0, %1, %2\n v_xor_b32 %0, %0, %3" : "=&v"(x) : "s"(s1), "v"(x), "v"(y));
      if (OP == 40) asm volatile("v_bcnt_u32_b32 %0, %1, 0\n v_xor_b32 %0, %0, %2" : "=&v"(x) : "v"(x), "v"(y));
      if (OP == 41) asm volatile("v_and_b32 %0, %1, %2\n v_and_b32 %0, %3, %0" : "=&v"(x) : "s"(s1), "v"(x), "s"(s0));
      if (OP == 42) asm volatile("v_add3_u32 %0, %1, %2, %3\n v_xor_b32 %0, %0, %2" : "=&v"(x) : "v"(y), "v"(x), "v"(y));
      if (OP == 43) { uint32_t t; asm volatile("v_and_b32 %1, %2, %3\n v_and_b32 %0, %4, %5\n v_xor_b32 %0, %0, %1\n v_bcnt_u32_b32 %0, %0, 0\n v_and_b32 %0, 1, %0\n v_add_u32 %0, %0, %5" : "=&v"(x), "=&v"(t) : "s"(s1), "v"(x), "s"(s0), "v"(y)); }
      if (OP == 44) { uint32_t t; asm volatile("v_and_b32 %1, %2, %3\n v_bitop3_b32 %0, %4, %5, %1 bitop3:0x78\n v_bcnt_u32_b32 %0, %0, 0\n v_and_b32 %0, 1, %0\n v_add_u32 %0, %0, %5" : "=&v"(x), "=&v"(t) : "s"(s1), "v"(x), "s"(s0), "v"(y)); }
      if (OP == 45) asm volatile("v_bcnt_u32_b32 %0, %1, 0\n v_bcnt_u32_b32 %0, %2, %0" : "=&v"(x) : "v"(x), "v"(y));
      if (OP == 46) asm volatile("v_mad_u32_u24 %0, %1, %2, %3\n v_xor_b32 %0, %0, %1" : "=&v"(x) : "v"(y), "s"(s1), "v"(x));
      if (OP == 47) asm volatile("v_lshlrev_b32 %0, 1, %1\n v_xor_b32 %0, %0, %2" : "=&v"(x) : "v"(x), "v"(y));
      a[i] = x;
    }
  }
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r ^= a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
void run(const char* name, uint32_t* d, int insts_per_iter = 8) {
  dim3 grid(256 * 8), block(256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<grid, block>>>(d, 1, 0x55555555u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) k<OP><<<grid, block>>>(d, 1, 0x55555555u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double waveinst = 5.0 * grid.x * (block.x / 64) * (double)ITER * insts_per_iter;
  double laneops = waveinst * 64;
  // cycles per wave-instruction per SIMD at 2.4 GHz: 1024 SIMDs
  double cyc = (ms * 1e-3 * 2.4e9) * 1024 / waveinst;
  printf("%-16s %8.3f ms  %7.2f Tlane-op/s  %5.2f cycles/wave-inst/SIMD (@2.4GHz)\n", name, ms / 5, laneops / (ms * 1e-3) / 1e12, cyc);
}

int main() {
  uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  run<0>("v_and_b32(s,v)", d); run<1>("v_xor_b32", d); run<2>("v_bcnt_u32_b32", d); run<3>("v_add_u32", d);
  run<4>("v_add3_u32", d); run<5>("v_ashrrev_i32", d); run<6>("v_mul_lo_u32", d); run<7>("v_mad_u32_u24", d);
  run<8>("v_cndmask_b32", d); run<9>("v_fma_f32", d); run<10>("v_or3_b32", d); run<11>("v_lshlrev_b32", d);
  run<12>("v_sub_u32", d); run<13>("v_cmp+v_cndmask", d, 16); run<14>("v_xad_u32", d); run<15>("v_and_or_b32", d);
  run<16>("v_and_b32(v,v)", d); run<17>("v_xor_b32(s,v)", d); run<18>("v_bcnt(v,v)", d); run<19>("v_mov_b32(s)", d);
  run<20>("v_lshrrev(v,v)", d); run<21>("v_bfe_u32", d); run<22>("v_and_b32(1,v)", d); run<23>("v_add_u32(s,v)", d);
  run<24>("v_lshlrev(1,v)", d); run<25>("v_ashrrev(1,v)", d); run<26>("v_or_b32(v,v)", d); run<27>("v_mov_b32(v)", d);
  run<28>("v_pk_add_u16", d); run<29>("v_mul_u32_u24", d); run<30>("v_subrev_u32", d); run<31>("v_max_i32", d);
  run<40>("bcnt+xor", d, 16); run<41>("and(s)+and(s)", d, 16); run<42>("add3+xor", d, 16); run<43>("rowB 6 ops", d, 48);
  run<44>("rowB bitop3 5ops", d, 40); run<45>("bcnt+bcnt", d, 16); run<46>("mad24(s)+xor", d, 16); run<47>("lshl+xor", d, 16);
  run<32>("v_add_u32(5,v)", d); run<33>("v_xor(lit,v)", d); run<34>("v_bitop3_b32", d); run<35>("and(s)+xor pair", d, 16);
  return 0;
}
