// tsim_lw_multi.hip.h - the register first pass over SEVERAL batches in one grid (k_sample_lw_multi).
//
// Why.  A 10^6-shot batch is ~15 wave-rows per SIMD: a first pass of its own lives for two rounds of waves, a third of
// which is ramp (kernel arguments, the binomial table, the first f rows: every wave of the chip waits for HBM at the
// same time) and tail; its issue-bound core - five Threefry-2x32-20 blocks per shot, 8.6 us per 10^6 shots on this
// chip whatever the instruction selection (profiles/r03/threefry_block.txt) - is about half of the 19.6 us the kernel
// takes alone.  Consecutive batches of a caller's loop (sampler.py:340-420: one sample_program per batch, the key
// split once per batch) are independent, so `tsim_sample_steps_device` enqueues up to TSIMK_LWM_MAX_STEPS of them as
// ONE grid of chip-resident blocks that stride over (batch, 1024-row block) pairs: one ramp and one tail per group,
// no kernel boundary and no host call between the batches, and the next row's f words are requested before the
// current row's draws start.  Each batch keeps its own subkeys, f / output buffers, hard-row lists and counters:
// results are those of TSIMK_LWM_MAX_STEPS separate launches, bit for bit (tests/test_gpu_steps.py).
#pragma once
#include "tsim_lw_pass.hip.h"

namespace tsimk {

// diagnostic builds only (scripts/lwm_probe.py): leave parts of the pass out to see what each costs (wrong results)
#ifndef TSIMK_LWM_SKIP
#define TSIMK_LWM_SKIP 0
#endif

#define TSIMK_LWM_MAX_STEPS 16
#define TSIMK_LWM_KEYS 16  // compiled outputs per program the fused launch carries subkeys for

struct LwStep {
  const uint64_t *f;      // [B, WF] packed error-mechanism rows of this batch
  uint64_t *out;          // [B] padded output words, or nullptr
  uint8_t *out_compact;   // [B, out_rb] bit_packed rows, or nullptr
  uint32_t *hard_index;   // this batch's hard-row lists
  uint32_t *ctl;          // its counters: ctl[32 k] = entries of list k, ctl[32 LISTS] = check row
  uint32_t *ctl_next;     // the counter set of the slot's NEXT launch: reset here
  uint32_t keys[2 * TSIMK_LWM_KEYS];  // per-output subkeys of this batch (sampler.py:74,147-148), host-computed
};

struct LwMultiArgs {
  const uint32_t *img;
  const uint32_t *tab;    // integer thresholds, all components
  long long B;            // rows per batch
  long long shot_offset;  // in-batch index of row 0 (the same for every batch of the group)
  int n_steps, blocks_per_step;
  int n_comp, lw_off, direct_rot, binom_off;
  int has_check, list_cap, n_lists, out_rb;
  int lwf_off;            // image offset of the fast record (k_sample_lw_fast), 0 = none
  uint32_t tab_bytes;     // size of the pattern tables when below 4 GB (k_sample_lw_fast: range-checked reads)
  int partial;            // k_sample_lw_fastm: hard rows are stored too - direct outputs and the tabulated components' bits - and
                          // their list entries carry the mask of the components that are NOT tabulated in bits 28.. (k_sample_hw then
                          // evaluates exactly those, one block per (row, component), and ORs their bits in)
  LwStep step[TSIMK_LWM_MAX_STEPS];
};

template <int WF32>
__device__ __forceinline__ void lwm_load_f(const uint64_t *f, long long row, bool active, uint32_t &f0, uint32_t &f1, uint32_t &f2,
                                           uint32_t &f3) {
  f0 = f1 = f2 = f3 = 0u;
  if (TSIMK_LWM_SKIP & 64) {  // diagnostic: no f load
    f0 = ((uint32_t)row * 0x9E3779B9u) & ((uint32_t)row * 0x85EBCA6Bu) & ((uint32_t)row * 0xC2B2AE35u) & 0x11111111u;
    return;
  }
  if (active) {
    const uint64_t *frow = f + row * (WF32 / 2);
    const uint64_t v0 = frow[0];
    f0 = (uint32_t)v0;
    f1 = (uint32_t)(v0 >> 32);
    if constexpr (WF32 == 4) {
      const uint64_t v1 = frow[1];
      f2 = (uint32_t)v1;
      f3 = (uint32_t)(v1 >> 32);
    }
  }
}

template <int WF32>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(TSIMK_LW_SGPRS))) k_sample_lw_multi(LwMultiArgs M) {
  typedef const __attribute__((address_space(4))) uint8_t *cbytes;
  typedef const __attribute__((address_space(4))) LwStep *cstep;
  const int nthr = blockDim.x;  // 1024
  cptr img = (cptr)(uintptr_t)M.img;
  // C(b, k + 1) for b < 64, k < 8: every wave writes the whole 2 KB table itself and reads only after its own
  // stores (k_sample_lw_reg does the same): no block barrier
  __shared__ uint4 binom_lds4[128];
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(M.img + M.binom_off);
    const int lane = threadIdx.x & 63;
    binom_lds4[lane] = src[lane];
    binom_lds4[64 + lane] = src[64 + lane];
    __builtin_amdgcn_wave_barrier();
  }
  const uint32_t *binom_lds = reinterpret_cast<const uint32_t *>(binom_lds4);
  // the batches' records sit in the kernel-argument segment: wave-uniform index -> scalar loads
  cstep steps = (cstep)((cbytes)__builtin_amdgcn_kernarg_segment_ptr() + __builtin_offsetof(LwMultiArgs, step));

  const uint32_t bps = (uint32_t)M.blocks_per_step;
  const uint32_t total = bps * (uint32_t)M.n_steps;
  uint32_t vb = blockIdx.x;
  if (vb >= total) return;
  uint32_t step = vb / bps, rb = vb - step * bps;  // once per block; afterwards by increments
  // the first row's f words: requested here, consumed in the loop; the NEXT row's are requested at the top of
  // every iteration (software prefetch: the HBM latency runs under this row's draws)
  uint32_t n0, n1, n2, n3;
  uint32_t dbg_acc = 0u;
  (void)dbg_acc;
  {
    const long long row = (long long)rb * nthr + threadIdx.x;
    lwm_load_f<WF32>(steps[step].f, row, row < M.B, n0, n1, n2, n3);
  }
  for (;;) {
    cstep S = steps + step;
    const long long row = (long long)rb * nthr + threadIdx.x;
    const bool active = row < M.B;
    const unsigned long long shot = (unsigned long long)(M.shot_offset + row);
    const uint32_t f0 = n0, f1 = n1, f2 = n2, f3 = n3;
    // next (batch, row block) of this block
    uint32_t vb_n = vb + gridDim.x, step_n = step, rb_n = rb + gridDim.x;
    while (rb_n >= bps) { rb_n -= bps; ++step_n; }
    const bool more = vb_n < total;
    if (more) {
      const long long row_n = (long long)rb_n * nthr + threadIdx.x;
      lwm_load_f<WF32>(steps[step_n].f, row_n, row_n < M.B, n0, n1, n2, n3);
    }
    if (rb == 0u && threadIdx.x <= TSIMK_LW_LISTS)  // the block that owns a batch's first rows resets the other counter set
      S->ctl_next[32u * threadIdx.x] = (threadIdx.x == TSIMK_LW_LISTS) ? 0xFFFFFFFFu : 0u;  // last: "no check row"
    bool hard = false;
    if (active) {
      hard = M.has_check && row == 0;
      if (hard) S->ctl[32 * TSIMK_LW_LISTS] = (uint32_t)row;
      uint32_t o0 = 0, o1 = 0;
      if (!(TSIMK_LWM_SKIP & 2)) lw_direct_reg<WF32>(img + M.direct_rot, f0, f1, f2, f3, o0, o1);  // K14, sampler.py:140-145
      for (int ci = 0; ci < M.n_comp; ++ci) {
        cptr rec = img + M.lw_off + ci * LW_WORDS;
        cptr sel = rec + LW_SEL_INLINE;
        const uint32_t m0 = f0 & sel[0], m1 = f1 & sel[1];
        uint32_t m2 = 0u, m3 = 0u;
        uint32_t cnt = (uint32_t)__builtin_popcount(m0) + (uint32_t)__builtin_popcount(m1);
        if constexpr (WF32 == 4) {
          m2 = f2 & sel[2];
          m3 = f3 & sel[3];
          cnt += (uint32_t)__builtin_popcount(m2) + (uint32_t)__builtin_popcount(m3);
        }
        if (cnt > rec[LW_WMAX]) hard = true;
        if (hard) continue;  // needs the full kernel: nothing of this row is written here
        uint32_t pat = rec[LW_BASES_INLINE];
#pragma unroll
        for (uint32_t w = 1; w <= TSIMK_LW_MAX_WEIGHT; ++w) {
          const uint32_t bw = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec[LW_BASES_INLINE + w]);
          pat = (cnt == w) ? bw : pat;
        }
        uint32_t ord = 0;
        if (!(TSIMK_LWM_SKIP & 4)) {
        lw_rank_word(m0, sel[0], 0u, binom_lds, ord, pat);
        lw_rank_word(m1, sel[1], sel[5], binom_lds, ord, pat);
        if constexpr (WF32 == 4) {
          lw_rank_word(m2, sel[2], sel[6], binom_lds, ord, pat);
          lw_rank_word(m3, sel[3], sel[7], binom_lds, ord, pat);
        }
        }
        const uint32_t n_out = rec[LW_NOUT];
        cptr outpos = img + rec[LW_OUTPOS];
        const uint32_t keybase = rec[LW_KEYBASE];
        const uint32_t *thr = M.tab + rec[LW_TAB] + ((size_t)pat << n_out);
        cptr kp = (cptr)((cbytes)S + __builtin_offsetof(LwStep, keys));
        lw_walk_keys(kp, thr, n_out, keybase, shot, [&](uint32_t i, bool bit) {
          const uint32_t dst = outpos[i];
          const uint32_t v = (bit ? 1u : 0u) << (dst & 31u);
          if ((dst >> 5) == 0u) o0 |= v;
          else o1 |= v;
        });
      }
      if (TSIMK_LWM_SKIP & 16) dbg_acc ^= o0 ^ o1;
      if (!hard && !(TSIMK_LWM_SKIP & (8 | 16))) {
        uint64_t *out = S->out;
        uint8_t *oc = S->out_compact;
        if (out) out[row] = (uint64_t)o0 | ((uint64_t)o1 << 32);  // WO == 1
        if (oc) {  // at most 8 bytes here (WO == 1)
          uint8_t *dst = oc + row * M.out_rb;
          for (int k = 0; k < M.out_rb; ++k) dst[k] = (uint8_t)(((k < 4) ? o0 : o1) >> (8 * (k & 3)));
        }
      }
    }
    // wave-aggregated append of the hard rows to this batch's lists
    const unsigned long long hm = __ballot(hard ? 1 : 0);
    if (hm != 0ull) {
      const int lane = (int)(threadIdx.x & 63u);
      const int leader = __builtin_ctzll(hm);
      uint32_t basei = 0;
      const uint32_t k = rb % (uint32_t)M.n_lists;  // this row block's sub-list
      uint32_t *ctl = S->ctl;
      if (lane == leader) basei = atomicAdd(&ctl[32u * k], (uint32_t)__popcll(hm));
      basei = (uint32_t)__shfl((int)basei, leader, 64);
      if (hard)
        S->hard_index[(size_t)k * M.list_cap + basei + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull))] = (uint32_t)row;
    }
    if (!more) {
      if ((TSIMK_LWM_SKIP & 16) && S->out_compact) S->out_compact[threadIdx.x] = (uint8_t)dbg_acc;
      break;
    }
    vb = vb_n;
    step = step_n;
    rb = rb_n;
  }
}

}  // namespace tsimk
