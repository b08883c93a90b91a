// tsim_lw_fastm.hip.h - k_sample_lw_fast (tsim_lw_fast.hip.h) for programs of 2..4 compiled components of at most 8
// outputs each (the cultivation shape: components of 1, 1 and 6 outputs).  The general fused pass (k_sample_lw_multi)
// serves them with per-row scalar loads and divergent loops: 31.7 us per 10^6 shots of C4 where its eight Threefry
// draws cost 13.8.  Here every component gets what the single-component kernel has - rank table, base table and
// placement table in LDS, buffer-descriptor loads, uniform ordinal loop, straight-line threshold walk (one
// instantiation per output count, chosen by a wave-uniform switch inside the component loop) - with the tables'
// LDS offsets fixed by the packer (tsim_program.hip: "fastm" record).  A row is finished here when EVERY component's
// error pattern is tabulated.  Same thresholds, draws and hard-row protocol: bit-identical results.
#pragma once
#include "tsim_lw_fast.hip.h"

namespace tsimk {

#define TSIMK_LWFM_MAX_COMP 4

// one component of one row: weight test, colex rank, threshold walk, placement.  Returns "pattern not tabulated".
template <int WF32, int NOUT>
__device__ __forceinline__ bool lwfm_component(const uint32_t *lds, const uint32_t (&f)[4], cptr rec, const __amdgpu_buffer_rsrc_t &r_tab,
                                               cptr keys, uint32_t so_hi, uint32_t slo, uint32_t l_rank, uint32_t l_bases, uint32_t l_lut,
                                               bool active, uint32_t &o0, uint32_t &o1) {
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  constexpr int NPOS = 32 * WF32, RSTR = NPOS + 1;
  const uint32_t wmax = rec[LW_WMAX], tab_byte = rec[LW_TAB] * 4u, keybase = rec[LW_KEYBASE];
  uint32_t m[4] = {0u, 0u, 0u, 0u};
  uint32_t cnt = 0u;
#pragma unroll
  for (int w = 0; w < WF32; ++w) {
    m[w] = f[w] & rec[LW_SEL_INLINE + w];
    cnt += (uint32_t)__builtin_popcount(m[w]);
  }
  const bool hard = cnt > wmax;
  uint32_t pat = lds[l_bases + (cnt < 71u ? cnt : 71u)];
  const uint32_t live = (active && !hard) ? cnt : 0u;
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
    return lds[l_rank + k * (uint32_t)RSTR + p];
  };
  for (uint32_t k = 0u; __builtin_amdgcn_ballot_w64(live > k) != 0ull; ++k) pat += ordinal(k);
  const uint32_t thr = tab_byte + (pat << (NOUT + 2));
  cptr kp = keys + 2u * keybase;
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
  const u32x2 placed = *reinterpret_cast<const u32x2 *>(&lds[l_lut + 2u * (node & ((1u << NOUT) - 1u))]);
  o0 |= hard ? 0u : placed.x;  // (a heavier pattern's walk went through some other pattern's tree)
  o1 |= hard ? 0u : placed.y;
  return hard;
}

template <int WF32>
__global__ void __launch_bounds__(1024) k_sample_lw_fastm(LwMultiArgs M) {
  typedef const __attribute__((address_space(4))) uint8_t *cbytes;
  typedef const __attribute__((address_space(4))) LwStep *cstep;
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  constexpr int NPOS = 32 * WF32, RSTR = NPOS + 1;
  uint32_t *lds = tsimk_lds;  // dynamic: 4 * MAX_RUNS words of runs, then per component RANK[8][RSTR], BASES[72], LUT[2^n_out][2]
  const int nthr = blockDim.x;
  cptr img = (cptr)(uintptr_t)M.img;
  cptr fr = img + M.lwf_off;  // the "fastm" record
  const uint32_t n_comp = fr[4];
  {
    const uint32_t *g = M.img;
    const uint32_t runs_off = fr[3], n_runs = fr[0];
    for (int i = threadIdx.x; i < 4 * TSIMK_LWF_MAX_RUNS; i += nthr) lds[i] = (uint32_t)i < 4u * n_runs ? g[runs_off + i] : 0u;
    uint32_t l = 4u * TSIMK_LWF_MAX_RUNS;  // LDS layout: the same running sum in the row loop below and on the host (lwfm_lds_words)
    for (uint32_t c = 0; c < n_comp; ++c) {
      cptr hc = fr + 16 + 8 * c;
      const uint32_t rank_off = hc[0], lut_off = hc[1], nout = hc[2];
      const uint32_t l_rank = l, l_bases = l + 8u * (uint32_t)RSTR, l_lut = (l_bases + 72u + 1u) & ~1u;
      l = l_lut + (2u << nout);
      for (int i = threadIdx.x; i < 8 * RSTR; i += nthr)
        lds[l_rank + i] = (i % RSTR) < NPOS ? g[rank_off + (uint32_t)(i / RSTR) * 128u + (uint32_t)(i % RSTR)] : 0u;
      for (int i = threadIdx.x; i < (int)(2u << nout); i += nthr) lds[l_lut + i] = g[lut_off + i];
      if (threadIdx.x < 72) lds[l_bases + threadIdx.x] = threadIdx.x < 8 ? g[M.lw_off + c * LW_WORDS + LW_BASES_INLINE + threadIdx.x] : 0u;
    }
    __syncthreads();
  }
  const uint32_t n_runs = fr[0], flip0 = fr[1], flip1 = fr[2];
  const __amdgpu_buffer_rsrc_t r_tab = __builtin_amdgcn_make_buffer_rsrc((void *)M.tab, 0, M.tab_bytes, 0x00020000);
  cstep steps = (cstep)((cbytes)__builtin_amdgcn_kernarg_segment_ptr() + __builtin_offsetof(LwMultiArgs, step));
  const uint32_t so_lo = (uint32_t)M.shot_offset, so_hi = (uint32_t)((unsigned long long)M.shot_offset >> 32);
  const uint32_t bps = (uint32_t)M.blocks_per_step;
  const uint32_t total = bps * (uint32_t)M.n_steps;
  uint32_t vb = blockIdx.x;
  if (vb >= total) return;
  uint32_t step = vb / bps, rb = vb - step * bps;
  const uint32_t Bu = (uint32_t)M.B;
  uint32_t n[4] = {0u, 0u, 0u, 0u};
  auto load_f = [&](uint32_t st, uint32_t rbk) {
    const __amdgpu_buffer_rsrc_t r_f = __builtin_amdgcn_make_buffer_rsrc((void *)steps[st].f, 0, Bu * (uint32_t)(4 * WF32), 0x00020000);
    const uint32_t off = (rbk * (uint32_t)nthr + threadIdx.x) * (uint32_t)(4 * WF32);
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
    const uint32_t f[4] = {n[0], n[1], n[2], n[3]};
    uint32_t vb_n = vb + gridDim.x, step_n = step, rb_n = rb + gridDim.x;
    while (rb_n >= bps) { rb_n -= bps; ++step_n; }
    const bool more = vb_n < total;
    if (more) load_f(step_n, rb_n);
    if (rb == 0u && threadIdx.x <= TSIMK_LW_LISTS)
      S->ctl_next[32u * threadIdx.x] = (threadIdx.x == TSIMK_LW_LISTS) ? 0xFFFFFFFFu : 0u;
    // ---- K14: direct outputs (sampler.py:140-145)
    uint32_t o0 = 0u, o1 = 0u;
    for (uint32_t r = 0; r < n_runs; ++r) {
      const u32x4 run = *reinterpret_cast<const u32x4 *>(&lds[4u * r]);
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
    // ---- the components, in processing order (sampler.py:147-148): every one runs, a row is finished here when none is hard
    bool hard = false;
    uint32_t hmask = 0u;  // the components this row's pattern is not tabulated for
    const uint32_t slo = so_lo + row;
    cptr keys = (cptr)((cbytes)S + __builtin_offsetof(LwStep, keys));
    uint32_t l = 4u * TSIMK_LWF_MAX_RUNS;
    for (uint32_t c = 0; c < n_comp; ++c) {
      cptr rec = img + M.lw_off + c * LW_WORDS;
      const uint32_t nout = fr[16 + 8 * c + 2];
      const uint32_t l_rank = l, l_bases = l + 8u * (uint32_t)RSTR, l_lut = (l_bases + 72u + 1u) & ~1u;
      l = l_lut + (2u << nout);
      bool h = false;
      switch (nout) {  // wave-uniform
#define TSIM_FM(N) case N: h = lwfm_component<WF32, N>(lds, f, rec, r_tab, keys, so_hi, slo, l_rank, l_bases, l_lut, active, o0, o1); break;
        TSIM_FM(1) TSIM_FM(2) TSIM_FM(3) TSIM_FM(4) TSIM_FM(5) TSIM_FM(6) TSIM_FM(7) TSIM_FM(8)
#undef TSIM_FM
        default: break;
      }
      hard = hard || h;
      hmask |= h ? (1u << c) : 0u;
    }
    if (M.has_check && rb == 0u && threadIdx.x == 0u) {  // the normalisation-check row (sampler.py:66-72): always hard
      hard = true;
      hmask = (1u << n_comp) - 1u;  // every component is evaluated again, with trial bit 0 beside it
      S->ctl[32 * TSIMK_LW_LISTS] = row;
    }
    hard = hard && active;
    const bool easy = active && !hard;
    if (easy || (hard && M.partial)) {
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
    const unsigned long long hm = __builtin_amdgcn_ballot_w64(hard);
    if (hm != 0ull) {
      const int lane = (int)(threadIdx.x & 63u);
      const int leader = __builtin_ctzll(hm);
      uint32_t basei = 0;
      const uint32_t k = rb & (uint32_t)(M.n_lists - 1);
      uint32_t *ctl = S->ctl;
      if (lane == leader) basei = atomicAdd(&ctl[32u * k], (uint32_t)__popcll(hm));
      basei = (uint32_t)__shfl((int)basei, leader, 64);
      if (hard) S->hard_index[(size_t)k * M.list_cap + basei + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull))] = M.partial ? (row | (hmask << 28)) : row;
    }
    if (!more) break;
    vb = vb_n;
    step = step_n;
    rb = rb_n;
  }
}

}  // namespace tsimk
