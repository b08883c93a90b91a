// VALU issue-rate table for gfx950 (round 3): every instruction the first pass issues, and every candidate for its
// Threefry rotate, as `asm volatile` so that the instruction measured is the instruction written (the ISA of this
// file is dumped next to the timings: scripts/microbench/run_valu_table.sh).
//
// Each kernel keeps 8 independent accumulators per lane and issues the op on them round-robin, 8 waves per SIMD
// (2048 blocks x 256 threads = 32 waves per CU), so neither dependency latency nor occupancy limits the rate.
// The sustained shader clock is measured inside every kernel: s_memtime counts shader cycles, s_memrealtime a
// constant 100 MHz clock; cycles per wave-instruction per SIMD are reported at the MEASURED clock.
//
// Build: hipcc --offload-arch=gfx950 -O3 scripts/microbench/valu_table.hip -o scripts/microbench/valu_table.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define ITER 1024

struct Clk {
  unsigned long long cyc0, cyc1, rt0, rt1;
};

#define KERNEL_BEGIN(NAME)                                                                                  \
  __global__ void __launch_bounds__(256) NAME(uint32_t *out, Clk *clk, uint32_t s0, uint32_t s1) {          \
    uint32_t a[8];                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i * 40503u + s0;       \
    unsigned long long c0 = 0, r0 = 0;                                                                      \
    if (threadIdx.x == 0) { c0 = __builtin_readcyclecounter(); r0 = wall_clock64(); }                      \
    for (int it = 0; it < ITER; ++it) {                                                                     \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                       \
        uint32_t x = a[i], y = a[(i + 3) & 7], z = a[(i + 5) & 7];                                          \
        (void)y; (void)z;

#define KERNEL_END                                                                                          \
        a[i] = x;                                                                                           \
      }                                                                                                     \
    }                                                                                                       \
    if (threadIdx.x == 0 && blockIdx.x == 0) {                                                              \
      clk->cyc0 = c0; clk->rt0 = r0; clk->cyc1 = __builtin_readcyclecounter(); clk->rt1 = wall_clock64();  \
    }                                                                                                       \
    uint32_t r = 0;                                                                                         \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) r ^= a[i];                                                \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                                         \
  }

#define OP1(NAME, ASM, ...) KERNEL_BEGIN(NAME) asm volatile(ASM : "+v"(x) : __VA_ARGS__); KERNEL_END

// ---- plain two-operand ops, VGPR operands
OP1(k_add_vv, "v_add_u32 %0, %1, %0", "v"(y))
OP1(k_sub_vv, "v_sub_u32 %0, %1, %0", "v"(y))
OP1(k_xor_vv, "v_xor_b32 %0, %1, %0", "v"(y))
OP1(k_and_vv, "v_and_b32 %0, %1, %0", "v"(y))
OP1(k_or_vv, "v_or_b32 %0, %1, %0", "v"(y))
OP1(k_mov_v, "v_mov_b32 %0, %1", "v"(y))
OP1(k_lshr_imm, "v_lshrrev_b32 %0, 19, %0", "v"(y))
OP1(k_lshr_vv, "v_lshrrev_b32 %0, %1, %0", "v"(y))
OP1(k_lshl_imm, "v_lshlrev_b32 %0, 13, %0", "v"(y))
OP1(k_lshl_vv, "v_lshlrev_b32 %0, %1, %0", "v"(y))
OP1(k_ashr_imm, "v_ashrrev_i32 %0, 3, %0", "v"(y))
OP1(k_add_imm, "v_add_u32 %0, 5, %0", "v"(y))
OP1(k_add_lit, "v_add_u32 %0, 0x1BD11BDA, %0", "v"(y))
OP1(k_xor_lit, "v_xor_b32 %0, 0x12345678, %0", "v"(y))
OP1(k_or_lit, "v_or_b32 %0, 0x3f800000, %0", "v"(y))
OP1(k_bcnt_v0, "v_bcnt_u32_b32 %0, %0, 0", "v"(y))
OP1(k_bcnt_vv, "v_bcnt_u32_b32 %0, %0, %1", "v"(y))
OP1(k_mul_lo, "v_mul_lo_u32 %0, %1, %0", "v"(y))
OP1(k_mul_hi, "v_mul_hi_u32 %0, %1, %0", "v"(y))
OP1(k_mul_u24, "v_mul_u32_u24 %0, %1, %0", "v"(y))
OP1(k_min_u32, "v_min_u32 %0, %1, %0", "v"(y))
OP1(k_max_f32, "v_max_f32 %0, %1, %0", "v"(y))
OP1(k_sub_f32, "v_sub_f32 %0, %0, %1", "v"(y))
OP1(k_add_f32_imm, "v_add_f32 %0, -1.0, %0", "v"(y))
OP1(k_ffbl, "v_ffbl_b32 %0, %0", "v"(y))
OP1(k_ffbh, "v_ffbh_u32 %0, %0", "v"(y))
OP1(k_not, "v_not_b32 %0, %0", "v"(y))
OP1(k_bfrev, "v_bfrev_b32 %0, %0", "v"(y))
OP1(k_pk_add_u16, "v_pk_add_u16 %0, %1, %0", "v"(y))
// ---- SGPR operand in a two-operand op
OP1(k_add_sv, "v_add_u32 %0, %1, %0", "s"(s1))
OP1(k_xor_sv, "v_xor_b32 %0, %1, %0", "s"(s1))
OP1(k_and_sv, "v_and_b32 %0, %1, %0", "s"(s1))
OP1(k_mov_s, "v_mov_b32 %0, %1", "s"(s1))
OP1(k_lshr_sv, "v_lshrrev_b32 %0, %1, %0", "s"(s0))
// ---- rotate candidates
OP1(k_alignbit_xx_imm, "v_alignbit_b32 %0, %0, %0, 19", "v"(y))
OP1(k_alignbit_xy_imm, "v_alignbit_b32 %0, %0, %1, 19", "v"(y))
OP1(k_alignbit_xx_v, "v_alignbit_b32 %0, %0, %0, %1", "v"(y))
OP1(k_alignbit_xx_s, "v_alignbit_b32 %0, %0, %0, %1", "s"(s0))
OP1(k_alignbyte_xx_imm, "v_alignbyte_b32 %0, %0, %0, 1", "v"(y))
OP1(k_alignbyte_xy_imm, "v_alignbyte_b32 %0, %0, %1, 2", "v"(y))
OP1(k_perm_xx_v, "v_perm_b32 %0, %0, %0, %1", "v"(y))
OP1(k_perm_xx_s, "v_perm_b32 %0, %0, %0, %1", "s"(s1))
OP1(k_perm_xy_v, "v_perm_b32 %0, %0, %1, %2", "v"(y), "v"(z))
OP1(k_lshl_or_imm, "v_lshl_or_b32 %0, %0, 13, %1", "v"(y))
OP1(k_lshl_add_imm, "v_lshl_add_u32 %0, %0, 13, %1", "v"(y))
OP1(k_add_lshl_imm, "v_add_lshl_u32 %0, %0, %1, 13", "v"(y))
OP1(k_bfi, "v_bfi_b32 %0, %1, %0, %2", "v"(y), "v"(z))
OP1(k_bfe_imm, "v_bfe_u32 %0, %0, 3, 7", "v"(y))
OP1(k_bfe_vv, "v_bfe_u32 %0, %0, %1, %2", "v"(y), "v"(z))
OP1(k_xor_sdwa_w1, "v_xor_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1", "v"(y))
OP1(k_xor_sdwa_dstw1, "v_xor_b32_sdwa %0, %1, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_0", "v"(y))
// ---- three-operand ops, VGPR only and with scalar / constant operands
OP1(k_add3_vvv, "v_add3_u32 %0, %0, %1, %2", "v"(y), "v"(z))
OP1(k_add3_vvs, "v_add3_u32 %0, %0, %1, %2", "v"(y), "s"(s1))
OP1(k_add3_vvimm, "v_add3_u32 %0, %0, %1, 4", "v"(y))
OP1(k_add3_vslit, "v_add3_u32 %0, %0, %1, 4", "s"(s1))
OP1(k_bitop3_vvv, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96", "v"(y), "v"(z))
OP1(k_bitop3_vvs, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96", "v"(y), "s"(s1))
OP1(k_bitop3_vsv, "v_bitop3_b32 %0, %1, %0, %2 bitop3:0x6a", "s"(s1), "v"(y))
OP1(k_xad_vvv, "v_xad_u32 %0, %0, %1, %2", "v"(y), "v"(z))
OP1(k_and_or_vvv, "v_and_or_b32 %0, %0, %1, %2", "v"(y), "v"(z))
OP1(k_or3_vvv, "v_or3_b32 %0, %0, %1, %2", "v"(y), "v"(z))
OP1(k_mad_u24_vvv, "v_mad_u32_u24 %0, %0, %1, %2", "v"(y), "v"(z))
OP1(k_fma_f32, "v_fma_f32 %0, %0, %1, %2", "v"(y), "v"(z))
OP1(k_med3_u32, "v_med3_u32 %0, %0, %1, %2", "v"(y), "v"(z))
// ---- compares and selects (an SGPR pair as the mask: no VCC serialisation)
KERNEL_BEGIN(k_cmp_lt_f32)
  asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" : : "v"(x), "v"(y) : "s20", "s21");
KERNEL_END
KERNEL_BEGIN(k_cmp_gt_u32)
  asm volatile("v_cmp_gt_u32 s[20:21], %0, %1" : : "v"(x), "v"(y) : "s20", "s21");
KERNEL_END
KERNEL_BEGIN(k_cmp_gt_u32_s)
  asm volatile("v_cmp_gt_u32 s[20:21], %0, %1" : : "v"(x), "s"(s1) : "s20", "s21");
KERNEL_END
KERNEL_BEGIN(k_cndmask_sgpr)
  asm volatile("v_cndmask_b32 %0, %0, %1, s[22:23]" : "+v"(x) : "v"(y) : "s22", "s23");
KERNEL_END
KERNEL_BEGIN(k_cndmask_vcc)
  asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(y) : "vcc");
KERNEL_END
KERNEL_BEGIN(k_cndmask_imm)
  asm volatile("v_cndmask_b32 %0, 0, 1, s[22:23]" : "=v"(x) : : "s22", "s23");
KERNEL_END
KERNEL_BEGIN(k_cmp_cnd_pair)
  asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(x) : "v"(y) : "s20", "s21");
KERNEL_END
KERNEL_BEGIN(k_readfirstlane)
  { uint32_t s; asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s) : "v"(x)); (void)s; }
KERNEL_END
KERNEL_BEGIN(k_mbcnt_lo)
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, %0" : "+v"(x));
KERNEL_END
// ---- 64-bit candidates on a register pair
KERNEL_BEGIN(k_lshl_b64_pair)
  { unsigned long long q = ((unsigned long long)x << 32) | y; asm volatile("v_lshlrev_b64 %0, 13, %0" : "+v"(q)); x = (uint32_t)(q >> 32); }
KERNEL_END
KERNEL_BEGIN(k_lshr_b64_pair)
  { unsigned long long q = ((unsigned long long)x << 32) | y; asm volatile("v_lshrrev_b64 %0, 19, %0" : "+v"(q)); x = (uint32_t)q; }
KERNEL_END
KERNEL_BEGIN(k_mad_u64_u32)
  { unsigned long long q = y; asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(q) : "v"(x), "v"(y) : "s20", "s21"); x = (uint32_t)q ^ (uint32_t)(q >> 32); }
KERNEL_END

// ---- one Threefry round (x0 += x1; x1 = rotl(x1, r) ^ x0) in the candidate instruction selections; the pair
// (a[i], a[i+4]) is one chain, four chains per lane
#define ROUND_KERNEL(NAME, BODY)                                                                            \
  __global__ void __launch_bounds__(256) NAME(uint32_t *out, Clk *clk, uint32_t s0, uint32_t s1) {          \
    uint32_t a[8];                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i * 40503u + s0;       \
    unsigned long long c0 = 0, r0 = 0;                                                                      \
    if (threadIdx.x == 0) { c0 = __builtin_readcyclecounter(); r0 = wall_clock64(); }                      \
    for (int it = 0; it < ITER; ++it) {                                                                     \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                       \
        uint32_t x0 = a[i], x1 = a[i + 4], t, u; (void)t; (void)u;                                          \
        BODY                                                                                                \
        a[i] = x0; a[i + 4] = x1;                                                                           \
      }                                                                                                     \
    }                                                                                                       \
    if (threadIdx.x == 0 && blockIdx.x == 0) {                                                              \
      clk->cyc0 = c0; clk->rt0 = r0; clk->cyc1 = __builtin_readcyclecounter(); clk->rt1 = wall_clock64();  \
    }                                                                                                       \
    uint32_t r = 0;                                                                                         \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) r ^= a[i];                                                \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                                         \
  }

// rotl 13 = alignbit by 19
ROUND_KERNEL(r_alignbit, asm volatile("v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 19\n v_xor_b32 %1, %1, %0" : "+v"(x0), "+v"(x1));)
ROUND_KERNEL(r_shr_shl_bitop3, asm volatile("v_add_u32 %0, %0, %1\n v_lshrrev_b32 %2, 19, %1\n v_lshlrev_b32 %3, 13, %1\n v_bitop3_b32 %1, %2, %3, %0 bitop3:0x96" : "+v"(x0), "+v"(x1), "=&v"(t), "=&v"(u));)
ROUND_KERNEL(r_shr_lshlor_xor, asm volatile("v_add_u32 %0, %0, %1\n v_lshrrev_b32 %2, 19, %1\n v_lshl_or_b32 %1, %1, 13, %2\n v_xor_b32 %1, %1, %0" : "+v"(x0), "+v"(x1), "=&v"(t));)
ROUND_KERNEL(r_shr_lshladd_xor, asm volatile("v_add_u32 %0, %0, %1\n v_lshrrev_b32 %2, 19, %1\n v_lshl_add_u32 %1, %1, 13, %2\n v_xor_b32 %1, %1, %0" : "+v"(x0), "+v"(x1), "=&v"(t));)
ROUND_KERNEL(r_mul_shr_bitop3, asm volatile("v_add_u32 %0, %0, %1\n v_lshrrev_b32 %2, 19, %1\n v_mul_u32_u24 %3, 0x2000, %1\n v_bitop3_b32 %1, %2, %3, %0 bitop3:0x96" : "+v"(x0), "+v"(x1), "=&v"(t), "=&v"(u));)
// rotl 16 candidates
ROUND_KERNEL(r16_alignbit, asm volatile("v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 16\n v_xor_b32 %1, %1, %0" : "+v"(x0), "+v"(x1));)
ROUND_KERNEL(r16_perm, asm volatile("v_add_u32 %0, %0, %1\n v_perm_b32 %1, %1, %1, %2\n v_xor_b32 %1, %1, %0" : "+v"(x0), "+v"(x1) : "v"(0x01000302u));)
ROUND_KERNEL(r16_alignbyte, asm volatile("v_add_u32 %0, %0, %1\n v_alignbyte_b32 %1, %1, %1, 2\n v_xor_b32 %1, %1, %0" : "+v"(x0), "+v"(x1));)
ROUND_KERNEL(r16_sdwa, asm volatile("v_add_u32 %0, %0, %1\n v_xor_b32_sdwa %2, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_xor_b32_sdwa %2, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_0\n v_mov_b32 %1, %2" : "+v"(x0), "+v"(x1), "=&v"(t));)
// what the compiler makes of the C form
ROUND_KERNEL(r_c_form, x0 += x1; x1 = (x1 << 13) | (x1 >> 19); x1 ^= x0;)

typedef void (*kern_t)(uint32_t *, Clk *, uint32_t, uint32_t);
struct Row { const char *name; kern_t fn; int insts; };

int main(int argc, char **argv) {
  uint32_t *d; Clk *dclk;
  hipMalloc(&d, 2048 * 256 * 4);
  hipMalloc(&dclk, sizeof(Clk));
  const Row rows[] = {
#define R(k, n) {#k, k, n}
    R(k_add_vv, 8), R(k_sub_vv, 8), R(k_xor_vv, 8), R(k_and_vv, 8), R(k_or_vv, 8), R(k_mov_v, 8),
    R(k_lshr_imm, 8), R(k_lshr_vv, 8), R(k_lshl_imm, 8), R(k_lshl_vv, 8), R(k_ashr_imm, 8),
    R(k_add_imm, 8), R(k_add_lit, 8), R(k_xor_lit, 8), R(k_or_lit, 8),
    R(k_bcnt_v0, 8), R(k_bcnt_vv, 8), R(k_mul_lo, 8), R(k_mul_hi, 8), R(k_mul_u24, 8), R(k_min_u32, 8),
    R(k_max_f32, 8), R(k_sub_f32, 8), R(k_add_f32_imm, 8), R(k_ffbl, 8), R(k_ffbh, 8), R(k_not, 8), R(k_bfrev, 8),
    R(k_pk_add_u16, 8),
    R(k_add_sv, 8), R(k_xor_sv, 8), R(k_and_sv, 8), R(k_mov_s, 8), R(k_lshr_sv, 8),
    R(k_alignbit_xx_imm, 8), R(k_alignbit_xy_imm, 8), R(k_alignbit_xx_v, 8), R(k_alignbit_xx_s, 8),
    R(k_alignbyte_xx_imm, 8), R(k_alignbyte_xy_imm, 8),
    R(k_perm_xx_v, 8), R(k_perm_xx_s, 8), R(k_perm_xy_v, 8),
    R(k_lshl_or_imm, 8), R(k_lshl_add_imm, 8), R(k_add_lshl_imm, 8), R(k_bfi, 8), R(k_bfe_imm, 8), R(k_bfe_vv, 8),
    R(k_xor_sdwa_w1, 8), R(k_xor_sdwa_dstw1, 8),
    R(k_add3_vvv, 8), R(k_add3_vvs, 8), R(k_add3_vvimm, 8), R(k_add3_vslit, 8),
    R(k_bitop3_vvv, 8), R(k_bitop3_vvs, 8), R(k_bitop3_vsv, 8),
    R(k_xad_vvv, 8), R(k_and_or_vvv, 8), R(k_or3_vvv, 8), R(k_mad_u24_vvv, 8), R(k_fma_f32, 8), R(k_med3_u32, 8),
    R(k_cmp_lt_f32, 8), R(k_cmp_gt_u32, 8), R(k_cmp_gt_u32_s, 8), R(k_cndmask_sgpr, 8), R(k_cndmask_vcc, 8),
    R(k_cndmask_imm, 8), R(k_cmp_cnd_pair, 16), R(k_readfirstlane, 8), R(k_mbcnt_lo, 8),
    R(k_lshl_b64_pair, 8), R(k_lshr_b64_pair, 8), R(k_mad_u64_u32, 8),
    R(r_alignbit, 12), R(r_shr_shl_bitop3, 16), R(r_shr_lshlor_xor, 16), R(r_shr_lshladd_xor, 16),
    R(r_mul_shr_bitop3, 16), R(r16_alignbit, 12), R(r16_perm, 12), R(r16_alignbyte, 12), R(r16_sdwa, 16), R(r_c_form, 12),
#undef R
  };
  const char *only = argc > 1 ? argv[1] : nullptr;
  printf("%-22s %9s %9s %12s %10s %10s\n", "kernel", "ms", "clock_GHz", "winst/s", "cyc@clock", "cyc@2.4GHz");
  for (const Row &r : rows) {
    if (only && !strstr(r.name, only)) continue;
    dim3 grid(2048), block(256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    r.fn<<<grid, block>>>(d, dclk, 1, 0x55555555u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 5;
    for (int k = 0; k < reps; ++k) r.fn<<<grid, block>>>(d, dclk, 1, 0x55555555u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    Clk c; hipMemcpy(&c, dclk, sizeof c, hipMemcpyDeviceToHost);
    const double ghz = (double)(c.cyc1 - c.cyc0) / ((double)(c.rt1 - c.rt0) * 10.0);  // 100 MHz real-time ticks
    const double winst = (double)reps * grid.x * (block.x / 64) * (double)ITER * r.insts;
    const double per_simd = winst / (ms * 1e-3) / 1024.0;
    printf("%-22s %9.4f %9.3f %12.3e %10.2f %10.2f\n", r.name, ms / reps, ghz, winst / (ms * 1e-3), ghz * 1e9 / per_simd,
           2.4e9 / per_simd);
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
  return 0;
}
