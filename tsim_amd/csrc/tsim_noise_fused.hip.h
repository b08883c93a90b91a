// tsim_noise_fused.hip.h - device noise and the first pass in ONE kernel (round 6): k_noise_sample_fast.
//
// The resident pipeline with device noise was two kernels per batch: the tile sampler writes packed f rows to HBM (8 bytes
// per shot of C2's 11 algorithmic bytes), the first pass reads them back - 10.1 + 17.8 us per 10^6 shots, one after the
// other (beside each other on two streams they were slower: the first pass is a chip-full of 1024-thread blocks, a noise
// block of 32 KB waits for one of them to end).  Here a block of 16 waves alternates: it builds a tile of f rows in LDS
// (noise_wave_tile, tsim_noise.hip.h - a latency-bound phase: prefix sums, LDS atomics), copies the tile to the batch's f
// buffer (the hard-row kernels and the caller read it there: same bytes as k_noise_wave writes for the same key) and then
// runs k_sample_lw_fast's row code (tsim_lw_fast.hip.h - an issue-bound phase: the draws) on the tile's rows FROM LDS.
// Two blocks share a CU and drift apart, so one block's draws fill the issue slots the other's noise phase leaves empty.
// Same thresholds, draws, lists and counters as k_sample_lw_fast behind k_noise_wave: bit-identical results
// (tests/test_gpu_noise_fused.py).  One component of at most 8 outputs, f rows of at most 128 bits - the BASELINE
// distillation shapes; every other program takes k_noise_wave + its own first pass (tsim_sample.hip).
#pragma once
#include "tsim_lw_fast.hip.h"
#include "tsim_noise_wave.hip.h"

namespace tsimk {

struct NoiseFusedArgs {
  LwMultiArgs M;
  NoiseWaveArgs N;                                // (f, B, k0, k1 unused: per step below)
  uint32_t nkeys[2 * TSIMK_LWM_MAX_STEPS];        // the batches' noise keys
};

template <int WF32, int NOUT>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(TSIMK_LW_SGPRS))) k_noise_sample_fast(NoiseFusedArgs F) {
  typedef const __attribute__((address_space(4))) uint8_t *cbytes;
  typedef const __attribute__((address_space(4))) LwStep *cstep;
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const LwMultiArgs &M = F.M;
  constexpr int NPOS = 32 * WF32;
  constexpr int RSTR = NPOS + 1;
  constexpr int L_RANK = 0;
  constexpr int L_LUT = (L_RANK + 8 * RSTR + 1) & ~1;
  constexpr int L_RUNS = L_LUT + (2 << NOUT);
  constexpr int L_BASES = L_RUNS + 4 * TSIMK_LWF_MAX_RUNS;
  constexpr int L_WORDS = (L_BASES + 72 + 3) & ~3;  // (16-byte multiple: the dynamic segment behind it holds 8-byte rows)
  __shared__ uint32_t lds[L_WORDS];
  extern __shared__ unsigned long long noise_rows[];  // [tile][WF32 / 2], then the channel records
  const int nthr = blockDim.x;
  cptr img = (cptr)(uintptr_t)M.img;
  cptr rec = img + M.lw_off;
  cptr fr = img + M.lwf_off;
  uint32_t *chrec = reinterpret_cast<uint32_t *>(noise_rows + (size_t)F.N.tile * F.N.WF);
  {
    const uint32_t *g = M.img;
    const uint32_t rank_off = fr[LWF_RANK], lut_off = fr[LWF_LUT], runs_off = fr[LWF_RUNS], n_runs = fr[LWF_NRUNS];
    for (int i = threadIdx.x; i < 8 * RSTR; i += nthr)
      lds[L_RANK + i] = (i % RSTR) < NPOS ? g[rank_off + (uint32_t)(i / RSTR) * 128u + (uint32_t)(i % RSTR)] : 0u;
    for (int i = threadIdx.x; i < (2 << NOUT); i += nthr) lds[L_LUT + i] = g[lut_off + i];
    for (int i = threadIdx.x; i < 4 * TSIMK_LWF_MAX_RUNS; i += nthr) lds[L_RUNS + i] = (uint32_t)i < 4u * n_runs ? g[runs_off + i] : 0u;
    if (threadIdx.x < 72) lds[L_BASES + threadIdx.x] = threadIdx.x < 8 ? g[M.lw_off + LW_BASES_INLINE + threadIdx.x] : 0u;
    noise_wave_records(F.N, chrec);
    __syncthreads();
  }
  const uint32_t n_runs = fr[LWF_NRUNS], flip0 = fr[LWF_FLIP0], flip1 = fr[LWF_FLIP1];
  uint32_t sel[4];
#pragma unroll
  for (int w = 0; w < 4; ++w) sel[w] = w < WF32 ? rec[LW_SEL_INLINE + w] : 0u;
  const uint32_t wmax = rec[LW_WMAX];
  const uint32_t tab_byte = rec[LW_TAB] * 4u;
  const uint32_t keybase = rec[LW_KEYBASE];
  const __amdgpu_buffer_rsrc_t r_tab = __builtin_amdgcn_make_buffer_rsrc((void *)M.tab, 0, M.tab_bytes, 0x00020000);
  cstep steps = (cstep)((cbytes)__builtin_amdgcn_kernarg_segment_ptr() + __builtin_offsetof(NoiseFusedArgs, M) + __builtin_offsetof(LwMultiArgs, step));
  const uint32_t so_lo = (uint32_t)M.shot_offset, so_hi = (uint32_t)((unsigned long long)M.shot_offset >> 32);
  const uint32_t Bu = (uint32_t)M.B;
  const uint32_t tile = (uint32_t)F.N.tile, rpt = tile / (uint32_t)nthr;  // row blocks (of nthr rows) per tile
  const uint32_t tps = (Bu + tile - 1u) / tile;                          // tiles per batch
  const uint32_t total = tps * (uint32_t)M.n_steps;
  const uint32_t *rows32 = reinterpret_cast<const uint32_t *>(noise_rows);

  for (uint32_t t = blockIdx.x; t < total; t += gridDim.x) {
    const uint32_t step = t / tps, ts = t - step * tps;
    cstep S = steps + step;
    const int rows = (int)min(tile, Bu - ts * tile);
    // ---- phase A: the tile's f rows (k_noise_wave's code and stream), then their copy in the batch's f buffer
    noise_wave_tile(F.N, noise_rows, chrec, ts, rows, F.nkeys[2u * step], F.nkeys[2u * step + 1u]);
    {
      unsigned long long *dst = const_cast<unsigned long long *>(reinterpret_cast<const unsigned long long *>(S->f)) + (size_t)ts * tile * F.N.WF;
      for (int i = threadIdx.x; i < rows * F.N.WF; i += nthr) dst[i] = noise_rows[i];
    }
    // ---- phase B: k_sample_lw_fast's row code on the tile's rows, row block after row block
    for (uint32_t r = 0; r < rpt; ++r) {
      const uint32_t rb = ts * rpt + r;
      const uint32_t row = rb * (uint32_t)nthr + threadIdx.x;
      if (rb * (uint32_t)nthr >= Bu) break;  // block-uniform
      const bool active = row < Bu;
      const uint32_t lrow = r * (uint32_t)nthr + threadIdx.x;  // row inside the tile
      uint32_t f[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int w = 0; w < WF32; ++w) f[w] = active ? rows32[lrow * (uint32_t)WF32 + (uint32_t)w] : 0u;
      if (rb == 0u && threadIdx.x <= TSIMK_LW_LISTS)  // reset the slot's other counter set (nobody else touches it now)
        S->ctl_next[32u * threadIdx.x] = (threadIdx.x == TSIMK_LW_LISTS) ? 0xFFFFFFFFu : 0u;
      // K14: direct outputs f[idx] ^ flip (sampler.py:140-145)
      uint32_t o0 = 0u, o1 = 0u;
      for (uint32_t q = 0; q < n_runs; ++q) {
        const u32x4 run = *reinterpret_cast<const u32x4 *>(&lds[L_RUNS + 4u * q]);
        const uint32_t sw = (uint32_t)__builtin_amdgcn_readfirstlane((int)run.x) >> 8;
        uint32_t src = f[0];
#pragma unroll
        for (int w = 1; w < WF32; ++w) src = (sw == (uint32_t)w) ? f[w] : src;
        const uint32_t rot = __builtin_amdgcn_alignbit(src, src, run.x);
        o0 |= rot & run.y;
        o1 |= rot & run.z;
      }
      o0 ^= flip0;
      o1 ^= flip1;
      // the component: weight test and colex rank of the masked f words (sampler.py:48 without the gather)
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
      auto ordinal = [&](uint32_t k) -> uint32_t {
        uint32_t c[4], tt[4];
#pragma unroll
        for (int w = 0; w < WF32; ++w) {
          uint32_t fb;
          asm("v_ffbl_b32 %0, %1" : "=v"(fb) : "v"(m[w]));
          c[w] = w ? (fb | (32u * (uint32_t)w)) : fb;
          tt[w] = m[w] & (m[w] - 1u);
        }
        uint32_t pp = c[0];
#pragma unroll
        for (int w = 1; w < WF32; ++w) pp = pp < c[w] ? pp : c[w];
        bool lower_zero = m[0] == 0u;
        m[0] = tt[0];
#pragma unroll
        for (int w = 1; w < WF32; ++w) {
          const bool z = m[w] == 0u;
          m[w] = lower_zero ? tt[w] : m[w];
          lower_zero = lower_zero && z;
        }
        pp = pp < (uint32_t)NPOS ? pp : (uint32_t)NPOS;
        return lds[(uint32_t)L_RANK + k * (uint32_t)RSTR + pp];
      };
      {
        const uint32_t r0 = ordinal(0u);
        const uint32_t r1 = ordinal(1u);
        pat += r0 + r1;
        for (uint32_t k = 2u; __builtin_amdgcn_ballot_w64(live > k) != 0ull; ++k) pat += ordinal(k);
      }
      // thresholds of the pattern's prefix tree and the draws (sampler.py:62-79 with the thresholds tabulated)
      const uint32_t thr = tab_byte + (pat << (NOUT + 2));
      const uint32_t slo = so_lo + row;
      cptr kp = (cptr)((cbytes)S + __builtin_offsetof(LwStep, keys)) + 2u * keybase;
      auto draw = [&](int o) -> uint32_t {
        const uint32_t k0 = kp[2 * o], k1 = kp[2 * o + 1];
        return threefry_bits32_lo(k0, k1, k0 + so_hi, slo) >> 9;
      };
      uint32_t node = 1u;
      int i = 0;
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
      if constexpr (NOUT % 3 == 2) {
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
      const u32x2 placed = *reinterpret_cast<const u32x2 *>(&lds[L_LUT + 2u * (node & ((1u << NOUT) - 1u))]);
      o0 |= placed.x;
      o1 |= placed.y;
      if (easy) {
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
          const uint32_t off = row * (uint32_t)M.out_rb;
          const int rb8 = M.out_rb;
          if (rb8 >= 4) __builtin_amdgcn_raw_buffer_store_b32(o0, r_c, off, 0, 0);
          if (rb8 == 8) __builtin_amdgcn_raw_buffer_store_b32(o1, r_c, off, 4, 0);
          else {
            const uint32_t w = rb8 >= 4 ? o1 : o0;
            const uint32_t at = rb8 >= 4 ? off + 4u : off;
            const int rem = rb8 & 3;
            if (rem >= 2) __builtin_amdgcn_raw_buffer_store_b16((uint16_t)w, r_c, at, 0, 0);
            if (rem & 1) __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(w >> (rem == 3 ? 16 : 0)), r_c, at + (rem == 3 ? 2u : 0u), 0, 0);
          }
        }
      }
      // wave-aggregated append of the hard rows to this batch's lists
      const unsigned long long hm = __builtin_amdgcn_ballot_w64(hard);
      if (hm != 0ull) {
        const int lane = (int)(threadIdx.x & 63u);
        const int leader = __builtin_ctzll(hm);
        uint32_t basei = 0;
        const uint32_t k = rb & (uint32_t)(M.n_lists - 1);
        uint32_t *ctl = S->ctl;
        if (lane == leader) basei = atomicAdd(&ctl[32u * k], (uint32_t)__popcll(hm));
        basei = (uint32_t)__shfl((int)basei, leader, 64);
        if (hard) S->hard_index[(size_t)k * M.list_cap + basei + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull))] = row;
      }
    }
    __syncthreads();  // every row of the tile has been read: the next tile may zero the buffer
  }
}

}  // namespace tsimk
