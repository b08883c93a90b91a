// tsim_lw_fast.hip.h - the fused register first pass for the common program shape: ONE compiled component of at most
// 8 outputs (the BASELINE circuits: 5 logical observables in one component), k_sample_lw_fast.
//
// What the generic pass (k_sample_lw_multi) spends outside its Threefry blocks was measured by leaving parts out
// (scripts/lwm_probe.py, profiles/r03/first_pass_breakdown.txt): of 17.4 us per 10^6 shots the five draws are 8.6 -
// the chip's floor for them - and NO memory access matters (f loads, threshold gathers, stores: 0.8-1.0 us each,
// 2.1 together); the other 6.4 us were ~125 vector, ~200 scalar and ~60 scalar-memory instructions per 64 shots:
// 64-bit address arithmetic for every table and key, per-output position loads, divergent while-loops over the
// lane's set bits, select chains over scalar operands.  A scalar instruction costs a SIMD as much issue time as a
// half-rate vector one (4.4 cycles, scripts/microbench/salu_mix.hip).  This kernel removes them instead of tuning them:
//   * everything wave-uniform and loop-invariant is loaded ONCE per block into LDS (rank table, output look-up
//     table, direct-output runs, pattern bases) or once per wave into SGPRs (selection masks, table descriptor);
//   * memory goes through buffer descriptors with 32-bit byte offsets (no 64-bit address arithmetic at all);
//   * the colex rank of the lane's error pattern is a UNIFORM loop over ordinals (trip count = heaviest lane of the
//     wave, no exec masking): per ordinal the lowest set bit of the masked f words, one LDS read of
//     RANK[ordinal][bit position] = C(position inside f_sel, ordinal + 1), precomputed per program;
//   * the n_out sampled bits are placed with ONE look-up LUT[leaf] -> (word 0, word 1) instead of a shift per output;
//   * n_out is a template parameter: the threshold walk is straight-line code.
// Same thresholds, same draws, same hard-row protocol as k_sample_lw_reg / k_sample_lw_multi: bit-identical results
// (tests/test_gpu_steps.py runs every shape through both).
#pragma once
#include "tsim_lw_multi.hip.h"

namespace tsimk {

// header of the fast record in the program image (uint32 words, 64-byte aligned), followed by its tables
enum { LWF_NRUNS = 0, LWF_FLIP0, LWF_FLIP1, LWF_RUNS /* image offset: n_runs x (ctl, mask0, mask1, 0) */,
       LWF_RANK /* image offset: [8][128] words */, LWF_LUT /* image offset: [2^n_out][2] words */, LWF_NOUT, LWF_WORDS = 16 };
#define TSIMK_LWF_MAX_RUNS 32
#define TSIMK_LWF_MAX_NOUT 8

// u < t as integers: see bernoulli_threshold
__device__ __forceinline__ uint32_t threefry_bits32_lo(uint32_t k0, uint32_t k1, uint32_t k0hi, uint32_t lo) {
  // threefry_bits32 for a counter (hi, lo) whose high word is wave-uniform: k0hi = k0 + hi formed on the scalar unit
  const uint32_t k2 = k0 ^ k1 ^ 0x1BD11BDAu;
  uint32_t x0, x1, t;
#define TF_RN(r) "v_alignbit_b32 %[x1], %[x1], %[x1], " #r "\n v_xor_b32 %[x1], %[x1], %[x0]\n"
#define TF_RA(r) "v_add_u32 %[x0], %[x0], %[x1]\n" TF_RN(r)
#define TF_INJ(kb, i, ka) "s_add_i32 %[t], %[" #kb "], " #i "\n v_add_u32 %[x1], %[t], %[x1]\n v_add3_u32 %[x0], %[x0], %[x1], %[" #ka "]\n"
  asm("v_add_u32 %[x1], %[k1], %[lo]\n v_add_u32 %[x0], %[k0hi], %[x1]\n"
      TF_RN(19) TF_RA(17) TF_RA(6) TF_RA(26)
      TF_INJ(k2, 1, k1) TF_RN(15) TF_RA(3) TF_RA(16) TF_RA(8)
      TF_INJ(k0, 2, k2) TF_RN(19) TF_RA(17) TF_RA(6) TF_RA(26)
      TF_INJ(k1, 3, k0) TF_RN(15) TF_RA(3) TF_RA(16) TF_RA(8)
      TF_INJ(k2, 4, k1) TF_RN(19) TF_RA(17) TF_RA(6) TF_RA(26)
      "s_add_i32 %[t], %[k0], 5\n v_add_u32 %[x0], %[k2], %[x0]\n v_add_u32 %[x1], %[t], %[x1]\n v_xor_b32 %[x0], %[x0], %[x1]\n"
      : [x0] "=&v"(x0), [x1] "=&v"(x1), [t] "=&s"(t)
      : [lo] "v"(lo), [k0] "s"(k0), [k1] "s"(k1), [k2] "s"(k2), [k0hi] "s"(k0hi)
      : "scc");
#undef TF_INJ
#undef TF_RA
#undef TF_RN
  return x0;
}

template <int WF32, int NOUT>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(TSIMK_LW_SGPRS))) k_sample_lw_fast(LwMultiArgs M) {
  typedef const __attribute__((address_space(4))) uint8_t *cbytes;
  typedef const __attribute__((address_space(4))) LwStep *cstep;
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  constexpr int NPOS = 32 * WF32;                       // f-row bit positions
  constexpr int RSTR = NPOS + 1;                        // a rank row: NPOS entries and a zero (what an exhausted lane reads)
  constexpr int L_RANK = 0;                             // [8][RSTR]
  constexpr int L_LUT = (L_RANK + 8 * RSTR + 1) & ~1;   // [2^NOUT][2], 8-byte aligned
  constexpr int L_RUNS = L_LUT + (2 << NOUT);           // [MAX_RUNS][4]
  constexpr int L_BASES = L_RUNS + 4 * TSIMK_LWF_MAX_RUNS;  // [72]: bases[cnt], cnt <= 64; zeros above wmax
  constexpr int L_WORDS = L_BASES + 72;
  __shared__ uint32_t lds[L_WORDS];
  const int nthr = blockDim.x;
  cptr img = (cptr)(uintptr_t)M.img;
  cptr rec = img + M.lw_off;     // the one component's LW record
  cptr fr = img + M.lwf_off;     // its fast record
  // ---- once per block: tables into LDS
  {
    const uint32_t *g = M.img;
    const uint32_t rank_off = fr[LWF_RANK], lut_off = fr[LWF_LUT], runs_off = fr[LWF_RUNS], n_runs = fr[LWF_NRUNS];
    for (int i = threadIdx.x; i < 8 * RSTR; i += nthr)
      lds[L_RANK + i] = (i % RSTR) < NPOS ? g[rank_off + (uint32_t)(i / RSTR) * 128u + (uint32_t)(i % RSTR)] : 0u;
    for (int i = threadIdx.x; i < (2 << NOUT); i += nthr) lds[L_LUT + i] = g[lut_off + i];
    for (int i = threadIdx.x; i < 4 * TSIMK_LWF_MAX_RUNS; i += nthr) lds[L_RUNS + i] = (uint32_t)i < 4u * n_runs ? g[runs_off + i] : 0u;
    if (threadIdx.x < 72) lds[L_BASES + threadIdx.x] = threadIdx.x < 8 ? g[M.lw_off + LW_BASES_INLINE + threadIdx.x] : 0u;
    __syncthreads();
  }
  // ---- once per wave: scalars
  const uint32_t n_runs = fr[LWF_NRUNS], flip0 = fr[LWF_FLIP0], flip1 = fr[LWF_FLIP1];
  uint32_t sel[4];
#pragma unroll
  for (int w = 0; w < 4; ++w) sel[w] = w < WF32 ? rec[LW_SEL_INLINE + w] : 0u;
  const uint32_t wmax = rec[LW_WMAX];
  const uint32_t tab_byte = rec[LW_TAB] * 4u;
  const uint32_t keybase = rec[LW_KEYBASE];
  // the descriptor ends with the tables: a lane whose pattern index means nothing (hard rows) reads zeros, never beyond
  const __amdgpu_buffer_rsrc_t r_tab = __builtin_amdgcn_make_buffer_rsrc((void *)M.tab, 0, M.tab_bytes, 0x00020000);
  cstep steps = (cstep)((cbytes)__builtin_amdgcn_kernarg_segment_ptr() + __builtin_offsetof(LwMultiArgs, step));
  const uint32_t so_lo = (uint32_t)M.shot_offset, so_hi = (uint32_t)((unsigned long long)M.shot_offset >> 32);

  const uint32_t bps = (uint32_t)M.blocks_per_step;
  const uint32_t total = bps * (uint32_t)M.n_steps;
  uint32_t vb = blockIdx.x;
  if (vb >= total) return;
  uint32_t step = vb / bps, rb = vb - step * bps;
  const uint32_t Bu = (uint32_t)M.B;
  // the first row's f words; the NEXT row's are requested at the top of every iteration
  uint32_t n[4] = {0u, 0u, 0u, 0u};
  auto load_f = [&](uint32_t st, uint32_t rbk) {
    // the descriptor ends with the batch: rows beyond it (the last block's idle lanes) read zeros
    const __amdgpu_buffer_rsrc_t r_f = __builtin_amdgcn_make_buffer_rsrc((void *)steps[st].f, 0, Bu * (uint32_t)(4 * WF32), 0x00020000);
    const uint32_t row = rbk * (uint32_t)nthr + threadIdx.x;
    const uint32_t off = row * (uint32_t)(4 * WF32);
    if (TSIMK_LWM_SKIP & 64) { n[0] ^= (row * 0x9E3779B9u) & (row * 0x85EBCA6Bu) & (row * 0xC2B2AE35u) & 0x11111111u; n[1] = 0; return; }
    if constexpr (WF32 == 2) {
      const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r_f, off, 0, 0);
      n[0] = v.x; n[1] = v.y;
    } else {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r_f, off, 0, 0);
      n[0] = v.x; n[1] = v.y; n[2] = v.z; n[3] = v.w;
    }
  };
  load_f(step, rb);
  for (;;) {
    cstep S = steps + step;
    const uint32_t row = rb * (uint32_t)nthr + threadIdx.x;
    const bool active = row < Bu;
    uint32_t f[4] = {n[0], n[1], n[2], n[3]};
    uint32_t vb_n = vb + gridDim.x, step_n = step, rb_n = rb + gridDim.x;
    while (rb_n >= bps) { rb_n -= bps; ++step_n; }
    const bool more = vb_n < total;
    if (more) load_f(step_n, rb_n);
    if (rb == 0u && threadIdx.x <= TSIMK_LW_LISTS)  // reset the slot's other counter set (nobody else touches it now)
      S->ctl_next[32u * threadIdx.x] = (threadIdx.x == TSIMK_LW_LISTS) ? 0xFFFFFFFFu : 0u;
    // ---- K14: direct outputs f[idx] ^ flip (sampler.py:140-145): bit-field runs, rotate and mask
    uint32_t o0 = 0u, o1 = 0u;
    for (uint32_t r = 0; r < ((TSIMK_LWM_SKIP & 2) ? 0u : n_runs); ++r) {
      const u32x4 run = *reinterpret_cast<const u32x4 *>(&lds[L_RUNS + 4u * r]);  // same address in every lane: broadcast
      const uint32_t sw = (uint32_t)__builtin_amdgcn_readfirstlane((int)run.x) >> 8;
      uint32_t src = f[0];
#pragma unroll
      for (int w = 1; w < WF32; ++w) src = (sw == (uint32_t)w) ? f[w] : src;
      const uint32_t rot = __builtin_amdgcn_alignbit(src, src, run.x);  // rotate right by ctl & 31
      o0 |= rot & run.y;
      o1 |= rot & run.z;
    }
    o0 ^= flip0;
    o1 ^= flip1;
    // ---- the component: weight test and colex rank of the masked f words (sampler.py:48 without the gather)
    uint32_t m[4] = {0u, 0u, 0u, 0u};
    uint32_t cnt = 0u;
#pragma unroll
    for (int w = 0; w < WF32; ++w) {
      m[w] = f[w] & sel[w];
      cnt += (uint32_t)__builtin_popcount(m[w]);
    }
    bool hard = cnt > wmax;
    if (M.has_check && rb == 0u && threadIdx.x == 0u) {  // the normalisation-check row (sampler.py:66-72): always hard
      hard = true;
      S->ctl[32 * TSIMK_LW_LISTS] = row;
    }
    hard = hard && active;
    const bool easy = active && !hard;
    uint32_t pat = lds[L_BASES + cnt];
    const uint32_t live = easy ? cnt : 0u;
    // One ordinal: the lowest set bit of the masked words leaves them, its RANK entry joins the index.  An exhausted
    // lane finds position 0xFFFFFFFF (v_ffbl_b32 of 0 is -1, `| 32 w` keeps it) -> the zero at the end of the row.
    auto ordinal = [&](uint32_t k) -> uint32_t {
      uint32_t c[4], t[4];
#pragma unroll
      for (int w = 0; w < WF32; ++w) {
        uint32_t fb;
        asm("v_ffbl_b32 %0, %1" : "=v"(fb) : "v"(m[w]));
        c[w] = w ? (fb | (32u * (uint32_t)w)) : fb;
        t[w] = m[w] & (m[w] - 1u);
      }
      uint32_t p = c[0];
#pragma unroll
      for (int w = 1; w < WF32; ++w) p = p < c[w] ? p : c[w];
      bool lower_zero = m[0] == 0u;
      m[0] = t[0];
#pragma unroll
      for (int w = 1; w < WF32; ++w) {
        const bool z = m[w] == 0u;
        m[w] = lower_zero ? t[w] : m[w];
        lower_zero = lower_zero && z;
      }
      p = p < (uint32_t)NPOS ? p : (uint32_t)NPOS;
      return lds[(uint32_t)L_RANK + k * (uint32_t)RSTR + p];
    };
    if (!(TSIMK_LWM_SKIP & 4)) {
      // The first two ordinals without asking (98 % of the waves hold a lane of weight 2), the rest while a lane
      // needs them.  Lanes that are not `easy` run along: whatever they add up is never used (their table reads stay
      // inside the descriptor).
      const uint32_t r0 = ordinal(0u);
      const uint32_t r1 = ordinal(1u);
      pat += r0 + r1;
      for (uint32_t k = 2u; __builtin_amdgcn_ballot_w64(live > k) != 0ull; ++k) pat += ordinal(k);
    }
    // ---- thresholds of the pattern's prefix tree and the draws (sampler.py:62-79 with the thresholds tabulated)
    const uint32_t thr = tab_byte + (pat << (NOUT + 2));  // byte offset of the pattern's row
    const uint32_t slo = so_lo + row;  // the host launches this kernel only when shot_offset + B stays below the next 2^32
    cptr kp = (cptr)((cbytes)S + __builtin_offsetof(LwStep, keys)) + 2u * keybase;
    auto draw = [&](int o) -> uint32_t {
      const uint32_t k0 = kp[2 * o], k1 = kp[2 * o + 1];
      if (TSIMK_LWM_SKIP & 1) return (slo * 0x9E3779B9u + k0 + k1) >> 9;  // diagnostic: no Threefry
      return threefry_bits32_lo(k0, k1, k0 + so_hi, slo) >> 9;
    };
    uint32_t node = 1u;
    int i = 0;
    if (TSIMK_LWM_SKIP & 32) {  // diagnostic: thresholds without memory
#pragma unroll
      for (; i < NOUT; ++i) node = 2u * node + (draw(i) < ((thr * 2654435761u + node * 40503u) & 0x7FFFFFu) ? 1u : 0u);
    }
#pragma unroll
    for (; i + 3 <= NOUT; i += 3) {
      const uint32_t t0 = __builtin_amdgcn_raw_buffer_load_b32(r_tab, thr + 4u * node, 0, 0);
      const u32x2 t1 = __builtin_amdgcn_raw_buffer_load_b64(r_tab, thr + 8u * node, 0, 0);
      const u32x4 t2 = __builtin_amdgcn_raw_buffer_load_b128(r_tab, thr + 16u * node, 0, 0);
      const uint32_t d0 = draw(i), d1 = draw(i + 1), d2 = draw(i + 2);
      const bool b0 = d0 < t0;
      const bool b1 = d1 < (b0 ? t1.y : t1.x);
      const uint32_t lo = b1 ? t2.y : t2.x, hi = b1 ? t2.w : t2.z;
      const bool b2 = d2 < (b0 ? hi : lo);
      node = 8u * node + (b0 ? 4u : 0u) + (b1 ? 2u : 0u) + (b2 ? 1u : 0u);
    }
    if (TSIMK_LWM_SKIP & 32) {
    } else if constexpr (NOUT % 3 == 2) {
      const uint32_t t0 = __builtin_amdgcn_raw_buffer_load_b32(r_tab, thr + 4u * node, 0, 0);
      const u32x2 t1 = __builtin_amdgcn_raw_buffer_load_b64(r_tab, thr + 8u * node, 0, 0);
      const uint32_t d0 = draw(i), d1 = draw(i + 1);
      const bool b0 = d0 < t0;
      const bool b1 = d1 < (b0 ? t1.y : t1.x);
      node = 4u * node + (b0 ? 2u : 0u) + (b1 ? 1u : 0u);
    } else if constexpr (NOUT % 3 == 1) {
      const uint32_t t0 = __builtin_amdgcn_raw_buffer_load_b32(r_tab, thr + 4u * node, 0, 0);
      node = 2u * node + (draw(i) < t0 ? 1u : 0u);
    }
    // ---- place the sampled bits (one look-up: leaf -> the two output words), store the row
    const u32x2 placed = *reinterpret_cast<const u32x2 *>(&lds[L_LUT + 2u * (node & ((1u << NOUT) - 1u))]);
    o0 |= placed.x;
    o1 |= placed.y;
    if (TSIMK_LWM_SKIP & 16) n[0] ^= (o0 ^ o1) & 0x10101010u;  // diagnostic: results stay live without stores
    if (easy && !(TSIMK_LWM_SKIP & 16)) {
      uint64_t *out = S->out;
      uint8_t *oc = S->out_compact;
      if (out) {
        const __amdgpu_buffer_rsrc_t r_o = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, 0xFFFFFFFF, 0x00020000);
        u32x2 v;
        v.x = o0; v.y = o1;
        __builtin_amdgcn_raw_buffer_store_b64(v, r_o, row * 8u, 0, 0);
      }
      if (oc) {
        const __amdgpu_buffer_rsrc_t r_c = __builtin_amdgcn_make_buffer_rsrc((void *)oc, 0, 0xFFFFFFFF, 0x00020000);
        // out_rb bytes at row * out_rb (any alignment: the device runs in unaligned-access mode): as few stores as the
        // size allows - 4-byte pieces, then 2, then 1 (20 outputs: a short and a byte instead of three bytes)
        const uint32_t off = row * (uint32_t)M.out_rb;
        const int rb8 = M.out_rb;
        if (rb8 >= 4) __builtin_amdgcn_raw_buffer_store_b32(o0, r_c, off, 0, 0);
        if (rb8 == 8) __builtin_amdgcn_raw_buffer_store_b32(o1, r_c, off, 4, 0);
        else {
          const uint32_t w = rb8 >= 4 ? o1 : o0;  // the word the remaining 1..3 bytes come from
          const uint32_t at = rb8 >= 4 ? off + 4u : off;
          const int rem = rb8 & 3;
          if (rem >= 2) __builtin_amdgcn_raw_buffer_store_b16((uint16_t)w, r_c, at, 0, 0);
          if (rem & 1) __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(w >> (rem == 3 ? 16 : 0)), r_c, at + (rem == 3 ? 2u : 0u), 0, 0);
        }
      }
    }
    // ---- wave-aggregated append of the hard rows to this batch's lists
    const unsigned long long hm = __builtin_amdgcn_ballot_w64(hard);
    if (hm != 0ull) {
      const int lane = (int)(threadIdx.x & 63u);
      const int leader = __builtin_ctzll(hm);
      uint32_t basei = 0;
      const uint32_t k = rb & (uint32_t)(M.n_lists - 1);  // this row block's sub-list (n_lists is a power of two)
      uint32_t *ctl = S->ctl;
      if (lane == leader) basei = atomicAdd(&ctl[32u * k], (uint32_t)__popcll(hm));
      basei = (uint32_t)__shfl((int)basei, leader, 64);
      if (hard) S->hard_index[(size_t)k * M.list_cap + basei + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull))] = row;
    }
    if (!more) break;
    vb = vb_n;
    step = step_n;
    rb = rb_n;
  }
}

}  // namespace tsimk
