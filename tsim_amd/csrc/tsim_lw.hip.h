// tsim_lw.hip.h - low-weight error-pattern tables for the autoregressive sampler (gfx950).
//
// Everything `_sample_component` (reference: src/tsim/sampler.py:28-81) computes for a shot -
// prev = |amp_0(f_sel)|, p1 = |amp_i(f_sel, m_0..m_{i-1}, 1)|, the Bernoulli threshold p1/prev and
// the chain rule prev <- bit ? p1 : prev - p1 - is a function of (f_sel, m_0..m_{i-1}) only.  The
// shot index enters through the uniform draw alone.  Error bits are sparse (the regime the
// reference's ChannelSampler is built for, noise/channels.py:624-658), so most shots of a batch
// carry one of very few f_sel patterns: weight 0, 1, 2 or 3.  For those patterns the packer
// tabulates the threshold of every node of the prefix tree once per program (k_lw_build, with the
// SAME device arithmetic as the sampling kernels: eval_any -> cabs32 -> __fdiv_rn/__fsub_rn), and
//
//   pass 1 (k_sample_lw): every shot: direct bits, f_sel gather, weight test.  Shots whose every
//           component has a tabulated pattern finish here: per output one Threefry draw and one
//           4-byte table gather (`u < thr[pattern][prefix node]`).  The others are appended to the
//           "hard" list.
//   pass 2: the full kernel (k_sample4 / k_sample) on the hard list only (row indirection).
//
// Results are bit-identical to running the full kernel on every shot: same thresholds (floats
// produced by the same instruction sequence), same draws (Threefry counter = in-batch shot index).
#pragma once
#include "tsim_kernels.hip.h"

namespace tsimk {

// per-component record of the pattern tables (uint32 words, in the program image)
enum {
  LW_NOUT = 0, LW_F, LW_FSEL, LW_OUTPOS, LW_KEYBASE, LW_WMAX, LW_TAB /* float offset into tab */,
  LW_OFF2 /* index of the first weight-2 pattern */, LW_OFF3, LW_NPAT, LW_WORDS = 16
};
#define TSIMK_LW_MAX_NOUT 10
#define TSIMK_LW_MAX_WEIGHT 3

// colex rank of a pattern with sorted set-bit positions b0 < b1 < b2 (missing ones passed as 0)
__host__ __device__ __forceinline__ uint32_t lw_binom2(uint32_t b) { return (b * (b - 1u)) >> 1; }
__host__ __device__ __forceinline__ uint32_t lw_binom3(uint32_t b) {
  // b(b-1)(b-2)/6 for b <= 64: the product is a multiple of 6 below 2^18, and
  // floor(v * 43691 / 2^18) == v / 6 for multiples of 6 up to 2^17 * 6 (error q * 2^-17 < 1)
  const uint32_t v = b * (b - 1u) * (b - 2u);
  return (uint32_t)(((unsigned long long)v * 43691ull) >> 18);
}

// ---------------------------------------------------------------------------
// table build: one lane per (pattern, full assignment m of the n outputs).  The lane walks the
// levels exactly like run_component with the outcome bits forced to m and records the threshold
// of every node on its path (lanes sharing a prefix write the same value).
// ---------------------------------------------------------------------------
struct LwBuildArgs {
  const uint32_t *img;
  const unsigned long long *patbits;  // [npat] f_sel bit patterns in table order
  float *tab;                         // this component's table: [npat << n_out]
  int comp_off;                       // row-layout component record (C_*)
  int npat;
};

template <int W, bool FAST>
__global__ void __launch_bounds__(256) k_lw_build(LwBuildArgs A) {
  cptr img = (cptr)(uintptr_t)A.img;
  cptr comp = img + A.comp_off;
  const uint32_t n_out = comp[C_NOUT], F = comp[C_F];
  cptr levels = img + comp[C_LEVELS];
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ((long long)A.npat << n_out)) return;
  const uint32_t pat = (uint32_t)(t >> n_out), m = (uint32_t)t & ((1u << n_out) - 1u);
  const unsigned long long pb = A.patbits[pat];
  uint32_t x[W];
#pragma unroll
  for (int w = 0; w < W; ++w) x[w] = (w < 2) ? (uint32_t)(pb >> (32 * w)) : 0u;
  float re, im;
  eval_any<W, FAST>(A.img, img, levels, x, re, im, nullptr);  // sampler.py:54
  float prev = cabs32(re, im);
  float *row = A.tab + ((size_t)pat << n_out);
  if (m == 0u) row[0] = 0.0f;  // node 0 is unused
  uint32_t node = 1u;
  for (uint32_t i = 0; i < n_out; ++i) {
    cptr lvl = levels + (i + 1) * L_WORDS;
    const uint32_t bitpos = F + i;
    const uint32_t wi = bitpos >> 5, bm = 1u << (bitpos & 31u);
#pragma unroll
    for (int w = 0; w < W; ++w)
      if ((uint32_t)w == wi) x[w] |= bm;
    eval_any<W, FAST>(A.img, img, lvl, x, re, im, nullptr);  // sampler.py:65
    const float p1 = cabs32(re, im);
    row[node] = __fdiv_rn(p1, prev);                          // sampler.py:75
    const bool bit = ((m >> (n_out - 1u - i)) & 1u) != 0u;    // prefix bits, first output first
#pragma unroll
    for (int w = 0; w < W; ++w)
      if ((uint32_t)w == wi) x[w] = bit ? (x[w] | bm) : (x[w] & ~bm);
    prev = bit ? p1 : __fsub_rn(prev, p1);                    // sampler.py:79
    node = 2u * node + (bit ? 1u : 0u);
  }
}

// ---------------------------------------------------------------------------
// pass 1
// ---------------------------------------------------------------------------
struct LwArgs {
  SampleArgs s;           // row_index/row_count: optional INPUT list (device-side post-selection)
  const float *tab;       // thresholds, all components
  int lw_off;             // image offset of the LW component records
  int has_check;          // the first slot's row is the normalisation-check row: always "hard"
  uint32_t *hard_index;   // [B] out: rows that need the full kernel (unordered)
  uint32_t *ctl;          // ctl[0] = hard count (zeroed by the caller), ctl[1] = check row
};

__global__ void __launch_bounds__(256) k_sample_lw(LwArgs L) {
  const SampleArgs &A = L.s;
  const int nthr = blockDim.x;
  const long long slot = (long long)blockIdx.x * nthr + threadIdx.x;
  long long n_rows = A.B;
  if (A.row_index) n_rows = (long long)*A.row_count;
  const bool active = slot < n_rows;
  long long row = slot;
  if (A.row_index) row = active ? (long long)A.row_index[slot] : 0;
  const unsigned long long shot = (unsigned long long)(A.shot_offset + row);
  cptr img = (cptr)(uintptr_t)A.img;

  const int WF32 = 2 * A.WF, WO32 = 2 * A.WO;
  uint32_t *lds_f = tsimk_lds + threadIdx.x;                // [WF32][nthr]
  uint32_t *lds_o = tsimk_lds + WF32 * nthr + threadIdx.x;  // [WO32][nthr]

  bool hard = false;
  if (active) {
    const uint64_t *frow = A.f + row * A.WF;
    for (int w = 0; w < A.WF; ++w) {
      const uint64_t v = frow[w];
      lds_f[(2 * w) * nthr] = (uint32_t)v;
      lds_f[(2 * w + 1) * nthr] = (uint32_t)(v >> 32);
    }
    for (int w = 0; w < WO32; ++w) lds_o[w * nthr] = 0u;
    hard = L.has_check && slot == 0;
    if (hard) L.ctl[1] = (uint32_t)row;

    // f_sel weight test of every component first: a hard shot skips all the rest of this pass
    for (int ci = 0; ci < A.n_comp && !hard; ++ci) {
      cptr rec = img + L.lw_off + ci * LW_WORDS;
      const uint32_t F = rec[LW_F];
      cptr fsel = img + rec[LW_FSEL];
      uint32_t cnt = 0;
      for (uint32_t j = 0; j < F; ++j) {
        const uint32_t src = fsel[j];
        cnt += (lds_f[(src >> 5) * nthr] >> (src & 31u)) & 1u;
      }
      hard = cnt > rec[LW_WMAX];
    }
  }

  if (active && !hard) {
    // K14: direct outputs f[idx] ^ flip (sampler.py:140-145)
    cptr dt = img + A.direct_off;
    for (int j = 0; j < A.n_direct; ++j) {
      const uint32_t s = dt[2 * j], dst = dt[2 * j + 1];
      const uint32_t src = s & 0x7FFFFFFFu;
      const uint32_t bit = ((lds_f[(src >> 5) * nthr] >> (src & 31u)) ^ (s >> 31)) & 1u;
      lds_o[(dst >> 5) * nthr] |= bit << (dst & 31u);
    }
    for (int ci = 0; ci < A.n_comp; ++ci) {
      cptr rec = img + L.lw_off + ci * LW_WORDS;
      const uint32_t n_out = rec[LW_NOUT], F = rec[LW_F];
      cptr fsel = img + rec[LW_FSEL];
      cptr outpos = img + rec[LW_OUTPOS];
      const uint32_t *keys = A.subkeys + 2 * rec[LW_KEYBASE];
      // positions (within f_sel) of the at most three set bits, ascending
      uint32_t b0 = 0, b1 = 0, b2 = 0, cnt = 0;
      for (uint32_t j = 0; j < F; ++j) {
        const uint32_t src = fsel[j];
        const bool set = ((lds_f[(src >> 5) * nthr] >> (src & 31u)) & 1u) != 0u;
        if (set) {
          if (cnt == 0) b0 = j; else if (cnt == 1) b1 = j; else b2 = j;
          ++cnt;
        }
      }
      const uint32_t base = (cnt == 0) ? 0u : (cnt == 1) ? 1u : (cnt == 2) ? rec[LW_OFF2] : rec[LW_OFF3];
      const uint32_t pat = base + b0 + (cnt >= 2 ? lw_binom2(b1) : 0u) + (cnt >= 3 ? lw_binom3(b2) : 0u);
      const float *thr = L.tab + rec[LW_TAB] + ((size_t)pat << n_out);
      uint32_t node = 1u;
      for (uint32_t i = 0; i < n_out; ++i) {
        const float u = uniform01(keys[2 * i], keys[2 * i + 1], shot);  // sampler.py:74-75
        const bool bit = u < thr[node];
        node = 2u * node + (bit ? 1u : 0u);
        const uint32_t dst = outpos[i];
        lds_o[(dst >> 5) * nthr] |= (bit ? 1u : 0u) << (dst & 31u);
      }
    }
    uint64_t *orow = A.out + row * A.WO;
    for (int w = 0; w < A.WO; ++w)
      orow[w] = (uint64_t)lds_o[(2 * w) * nthr] | ((uint64_t)lds_o[(2 * w + 1) * nthr] << 32);
  }

  // wave-aggregated append of the hard rows
  const unsigned long long hm = __ballot(hard ? 1 : 0);
  if (hm != 0ull) {
    const int lane = (int)(threadIdx.x & 63u);
    const int leader = __builtin_ctzll(hm);
    uint32_t basei = 0;
    if (lane == leader) basei = atomicAdd(&L.ctl[0], (uint32_t)__popcll(hm));
    basei = (uint32_t)__shfl((int)basei, leader, 64);
    if (hard) L.hard_index[basei + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull))] = (uint32_t)row;
  }
}

}  // namespace tsimk
