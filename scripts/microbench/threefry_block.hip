// Threefry-2x32-20 block, instruction selections compared on gfx950 (round 3).
//
// The first pass draws one uniform per compiled output and shot: jax.random.bernoulli's stream
// (/root/reference/src/tsim/sampler.py:74-75) = threefry2x32(subkey, (0, shot)), x0 ^ x1.  Five blocks per shot for
// the 35-qubit shape, 385 of the 523 VALU instructions a wave issues.  Every variant below computes the SAME bits
// (checked against the host loop in main) - they differ only in which instructions form the round.
//
// Build: hipcc --offload-arch=gfx950 -O3 scripts/microbench/threefry_block.hip -o scripts/microbench/threefry_block.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

enum { V_C = 0, V_ALIGNBIT_ASM, V_SHR_SHL_BITOP3, V_SHR_LSHLOR, V_SHR_LSHLADD, V_PERM_BYTES, V_PERM_BYTES_BITOP, V_VGPR_KEYS_C,
       V_VGPR_KEYS_BITOP3, V_FOLD_INJECT, V_FOLD_ASM, V_FOLD_ASM_X2, V_FOLD_ASM_X3, V_FOLD_ASM_X5, N_VARIANTS };
static const char *VNAME[N_VARIANTS] = {"c_form(compiler)", "alignbit+xor asm", "lshr+lshl+bitop3", "lshr+lshl_or+xor",
                                        "lshr+lshl_add+xor", "perm(16,24)+alignbit", "perm(16,24)+shifts+bitop3",
                                        "c_form, keys in VGPRs", "shifts+bitop3, keys in VGPRs", "c_form, injections folded into add3",
                                        "asm block, folded injections (kernel form)", "asm, 2 draws interleaved", "asm, 3+2 draws interleaved", "asm, 5 draws interleaved"};

__host__ __device__ inline uint32_t rotl_c(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

template <int V, int R>
__device__ __forceinline__ void tf_round(uint32_t &x0, uint32_t &x1, uint32_t perm16, uint32_t perm24) {
  if constexpr (V == V_C || V == V_VGPR_KEYS_C || V == V_FOLD_INJECT) {
    x0 += x1; x1 = rotl_c(x1, R); x1 ^= x0;
  } else if constexpr (V == V_ALIGNBIT_ASM) {
    asm volatile("v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, %2\n v_xor_b32 %1, %1, %0" : "+v"(x0), "+v"(x1) : "n"(32 - R));
  } else if constexpr (V == V_SHR_SHL_BITOP3 || V == V_VGPR_KEYS_BITOP3) {
    uint32_t t, u;
    asm volatile("v_add_u32 %0, %0, %1\n v_lshrrev_b32 %2, %4, %1\n v_lshlrev_b32 %3, %5, %1\n v_bitop3_b32 %1, %2, %3, %0 bitop3:0x96"
                 : "+v"(x0), "+v"(x1), "=&v"(t), "=&v"(u) : "n"(32 - R), "n"(R));
  } else if constexpr (V == V_SHR_LSHLOR) {
    uint32_t t;
    asm volatile("v_add_u32 %0, %0, %1\n v_lshrrev_b32 %2, %3, %1\n v_lshl_or_b32 %1, %1, %4, %2\n v_xor_b32 %1, %1, %0"
                 : "+v"(x0), "+v"(x1), "=&v"(t) : "n"(32 - R), "n"(R));
  } else if constexpr (V == V_SHR_LSHLADD) {
    uint32_t t;
    asm volatile("v_add_u32 %0, %0, %1\n v_lshrrev_b32 %2, %3, %1\n v_lshl_add_u32 %1, %1, %4, %2\n v_xor_b32 %1, %1, %0"
                 : "+v"(x0), "+v"(x1), "=&v"(t) : "n"(32 - R), "n"(R));
  } else if constexpr (V == V_PERM_BYTES || V == V_PERM_BYTES_BITOP) {
    if constexpr (R == 16) {
      asm volatile("v_add_u32 %0, %0, %1\n v_perm_b32 %1, %1, %1, %2\n v_xor_b32 %1, %1, %0" : "+v"(x0), "+v"(x1) : "s"(perm16));
    } else if constexpr (R == 24) {
      asm volatile("v_add_u32 %0, %0, %1\n v_perm_b32 %1, %1, %1, %2\n v_xor_b32 %1, %1, %0" : "+v"(x0), "+v"(x1) : "s"(perm24));
    } else if constexpr (V == V_PERM_BYTES) {
      asm volatile("v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, %2\n v_xor_b32 %1, %1, %0" : "+v"(x0), "+v"(x1) : "n"(32 - R));
    } else {
      uint32_t t, u;
      asm volatile("v_add_u32 %0, %0, %1\n v_lshrrev_b32 %2, %4, %1\n v_lshlrev_b32 %3, %5, %1\n v_bitop3_b32 %1, %2, %3, %0 bitop3:0x96"
                   : "+v"(x0), "+v"(x1), "=&v"(t), "=&v"(u) : "n"(32 - R), "n"(R));
    }
  }
}

template <int V>
__device__ __forceinline__ uint32_t tf_bits(uint32_t k0, uint32_t k1, uint32_t ctr, uint32_t perm16, uint32_t perm24) {
  const uint32_t k2 = k0 ^ k1 ^ 0x1BD11BDAu;
  uint32_t x0, x1;
  if constexpr (V == V_VGPR_KEYS_C || V == V_VGPR_KEYS_BITOP3) {
    // keys as vector registers: an add with a scalar operand issues at half rate
    uint32_t vk0, vk1, vk2;
    asm volatile("v_mov_b32 %0, %1" : "=v"(vk0) : "s"(k0));
    asm volatile("v_mov_b32 %0, %1" : "=v"(vk1) : "s"(k1));
    asm volatile("v_mov_b32 %0, %1" : "=v"(vk2) : "s"(k2));
    x0 = vk0; x1 = ctr + vk1;
#define RR(r) tf_round<V, r>(x0, x1, perm16, perm24);
    RR(13) RR(15) RR(26) RR(6)
    x0 += vk1; x1 += vk2; x1 += 1u;
    RR(17) RR(29) RR(16) RR(24)
    x0 += vk2; x1 += vk0; x1 += 2u;
    RR(13) RR(15) RR(26) RR(6)
    x0 += vk0; x1 += vk1; x1 += 3u;
    RR(17) RR(29) RR(16) RR(24)
    x0 += vk1; x1 += vk2; x1 += 4u;
    RR(13) RR(15) RR(26) RR(6)
    x0 += vk2; x1 += vk0; x1 += 5u;
    return x0 ^ x1;
  } else if constexpr (V == V_FOLD_INJECT) {
    // x0 += ka; x1 += kb + i; x0 += x1  ==  x1 += (kb + i) [scalar sum]; x0 = x0 + x1 + ka [one add3]
#define RN(r) x1 = rotl_c(x1, r); x1 ^= x0;
#define INJ(ka, kb, i) x1 += (kb) + (i); x0 = x0 + x1 + (ka);
    x1 = ctr + k1; x0 = k0 + x1;
    RN(13) RR(15) RR(26) RR(6)
    INJ(k1, k2, 1u) RN(17) RR(29) RR(16) RR(24)
    INJ(k2, k0, 2u) RN(13) RR(15) RR(26) RR(6)
    INJ(k0, k1, 3u) RN(17) RR(29) RR(16) RR(24)
    INJ(k1, k2, 4u) RN(13) RR(15) RR(26) RR(6)
    x0 += k2; x1 += k0 + 5u;
#undef RN
#undef INJ
    return x0 ^ x1;
  } else {
    x0 = k0; x1 = ctr + k1;
    RR(13) RR(15) RR(26) RR(6)
    x0 += k1; x1 += k2 + 1u;
    RR(17) RR(29) RR(16) RR(24)
    x0 += k2; x1 += k0 + 2u;
    RR(13) RR(15) RR(26) RR(6)
    x0 += k0; x1 += k1 + 3u;
    RR(17) RR(29) RR(16) RR(24)
    x0 += k1; x1 += k2 + 4u;
    RR(13) RR(15) RR(26) RR(6)
    x0 += k2; x1 += k0 + 5u;
#undef RR
    return x0 ^ x1;
  }
}


#include "threefry_gen.inc"

struct Keys { uint32_t k[10]; };

// one lane = `per_lane` shots in turn, five draws per shot (five subkeys), as the first pass does
template <int V>
__global__ void __launch_bounds__(1024) k_blocks(uint32_t *out, Keys K, uint32_t n_shots, uint32_t perm16, uint32_t perm24) {
  uint32_t acc = 0;
  for (uint32_t shot = blockIdx.x * blockDim.x + threadIdx.x; shot < n_shots; shot += gridDim.x * blockDim.x) {
    if constexpr (V == V_FOLD_ASM) {
      uint32_t b;
#pragma unroll
      for (int j = 0; j < 5; ++j) { threefry_bits32_x1(K.k[2 * j], K.k[2 * j + 1], shot, b); acc ^= b >> j; }
    } else if constexpr (V == V_FOLD_ASM_X2) {
      uint32_t b0, b1, b2, b3, b4;
      threefry_bits32_x2(K.k[0], K.k[1], K.k[2], K.k[3], shot, b0, b1);
      threefry_bits32_x2(K.k[4], K.k[5], K.k[6], K.k[7], shot, b2, b3);
      threefry_bits32_x1(K.k[8], K.k[9], shot, b4);
      acc ^= b0 ^ (b1 >> 1) ^ (b2 >> 2) ^ (b3 >> 3) ^ (b4 >> 4);
    } else if constexpr (V == V_FOLD_ASM_X3) {
      uint32_t b0, b1, b2, b3, b4;
      threefry_bits32_x3(K.k[0], K.k[1], K.k[2], K.k[3], K.k[4], K.k[5], shot, b0, b1, b2);
      threefry_bits32_x2(K.k[6], K.k[7], K.k[8], K.k[9], shot, b3, b4);
      acc ^= b0 ^ (b1 >> 1) ^ (b2 >> 2) ^ (b3 >> 3) ^ (b4 >> 4);
    } else if constexpr (V == V_FOLD_ASM_X5) {
      uint32_t b0, b1, b2, b3, b4;
      threefry_bits32_x5(K.k[0], K.k[1], K.k[2], K.k[3], K.k[4], K.k[5], K.k[6], K.k[7], K.k[8], K.k[9], shot, b0, b1, b2, b3, b4);
      acc ^= b0 ^ (b1 >> 1) ^ (b2 >> 2) ^ (b3 >> 3) ^ (b4 >> 4);
    } else {
#pragma unroll
      for (int j = 0; j < 5; ++j) acc ^= tf_bits<V>(K.k[2 * j], K.k[2 * j + 1], shot, perm16, perm24) >> j;
    }
  }
  // one word per wave would hide a wrong lane: keep every lane's word
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

static void host_tf(uint32_t k0, uint32_t k1, uint32_t &x0, uint32_t &x1) {
  const uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
  static const int R[8] = {13, 15, 26, 6, 17, 29, 16, 24};
  x0 += ks[0]; x1 += ks[1];
  for (int g = 0; g < 5; ++g) {
    for (int r = 0; r < 4; ++r) { x0 += x1; x1 = rotl_c(x1, R[(g & 1) * 4 + r]); x1 ^= x0; }
    x0 += ks[(g + 1) % 3]; x1 += ks[(g + 2) % 3] + (uint32_t)(g + 1);
  }
}

typedef void (*kern_t)(uint32_t *, Keys, uint32_t, uint32_t, uint32_t);

int main(int argc, char **argv) {
  const uint32_t n_shots = 1u << 24;
  const int blocks = 512, threads = 1024;  // one chip-full, as the first pass launches
  uint32_t *d; hipMalloc(&d, (size_t)blocks * threads * 4);
  Keys K;
  for (int i = 0; i < 10; ++i) K.k[i] = 0x9E3779B9u * (uint32_t)(i + 1) + 12345u;
  // host reference on a sample of lanes
  std::vector<uint32_t> want((size_t)blocks * threads, 0u);
  const int check_lanes[6] = {0, 1, 63, 1024, 77777, blocks * threads - 1};
  for (int li = 0; li < 6; ++li) {
    uint32_t acc = 0;
    for (uint32_t shot = (uint32_t)check_lanes[li]; shot < n_shots; shot += (uint32_t)(blocks * threads))
      for (int j = 0; j < 5; ++j) { uint32_t x0 = 0, x1 = shot; host_tf(K.k[2 * j], K.k[2 * j + 1], x0, x1); acc ^= (x0 ^ x1) >> j; }
    want[check_lanes[li]] = acc;
  }
  const kern_t fns[N_VARIANTS] = {k_blocks<0>, k_blocks<1>, k_blocks<2>, k_blocks<3>, k_blocks<4>, k_blocks<5>, k_blocks<6>, k_blocks<7>, k_blocks<8>, k_blocks<9>, k_blocks<10>, k_blocks<11>, k_blocks<12>, k_blocks<13>};
  printf("%-44s %9s %14s %16s %6s\n", "variant", "ms", "blocks/s", "us/1e6 shots", "ok");
  std::vector<uint32_t> got((size_t)blocks * threads);
  for (int v = 0; v < N_VARIANTS; ++v) {
    if (argc > 1 && atoi(argv[1]) != v) continue;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    fns[v]<<<blocks, threads>>>(d, K, n_shots, 0x01000302u, 0x00030201u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 5;
    for (int r = 0; r < reps; ++r) fns[v]<<<blocks, threads>>>(d, K, n_shots, 0x01000302u, 0x00030201u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(got.data(), d, got.size() * 4, hipMemcpyDeviceToHost);
    bool ok = true;
    for (int li = 0; li < 6; ++li) ok = ok && got[check_lanes[li]] == want[check_lanes[li]];
    const double per = ms / reps;
    printf("%-44s %9.4f %14.4e %16.2f %6s\n", VNAME[v], per, 5.0 * n_shots / (per * 1e-3), per * 1e3 / (n_shots / 1e6),
           ok ? "yes" : "NO");
  }
  return 0;
}
