// tsim_lw.hip.h - low-weight error-pattern tables for the autoregressive sampler (gfx950).
//
// Everything `_sample_component` (reference: src/tsim/sampler.py:28-81) computes for a shot -
// prev = |amp_0(f_sel)|, p1 = |amp_i(f_sel, m_0..m_{i-1}, 1)|, the Bernoulli threshold p1/prev and
// the chain rule prev <- bit ? p1 : prev - p1 - is a function of (f_sel, m_0..m_{i-1}) only.  The
// shot index enters through the uniform draw alone.  Error bits are sparse (the regime the
// reference's ChannelSampler is built for, noise/channels.py:624-658), so most shots of a batch
// carry one of very few f_sel patterns: weight 0 to 5.  For those patterns the packer
// tabulates the threshold of every node of the prefix tree once per program (k_lw_build, with the
// SAME device arithmetic as the sampling kernels: eval_any -> cabs32 -> __fdiv_rn/__fsub_rn), and
//
//   pass 1 (k_sample_lw): every shot: direct bits, f_sel gather, weight test.  Shots whose every
//           component has a tabulated pattern finish here: per output one Threefry draw and one
//           4-byte table gather (`u < thr[pattern][prefix node]`).  The others are appended to the
//           "hard" list.
//   pass 2: the hard rows only (row indirection): k_sample4h (tsim_kernel4h.hip.h) for the first 256
//           slots of each of the 64 lists, k_sample4 / k_sample for the rest.
//
// Results are bit-identical to running the full kernel on every shot: same thresholds (floats
// produced by the same instruction sequence), same draws (Threefry counter = in-batch shot index).
#pragma once
#include "tsim_kernels.hip.h"

namespace tsimk {

// per-component record of the pattern tables (uint32 words, in the program image)
enum {
  LW_NOUT = 0, LW_F, LW_FSELP /* gather program f row -> x */, LW_OUTPOS, LW_KEYBASE, LW_WMAX,
  LW_TAB /* float offset into tab */, LW_BASES /* image offset of 8 words: index of the first pattern of weight w */,
  LW_FMT /* 0: thr[pattern][2^n_out]; 1: chunked prefix tree (tsim_trie.hip.h) */, LW_NPAT, LW_FSELN /* chunks of the gather program */,
  LW_CHUNKS /* trie: chunks the component's table holds */,
  LW_NPAT_OK /* patterns below this index are complete (trie: the budget may end a build early; dense: = LW_NPAT) */,
  LW_SELMASK /* image offset of 4 selection-mask words + 4 prefix counts (register first pass), 0 = none */,
  // The masks and the bases live INSIDE the record (LW_SELMASK / LW_BASES point at these words): a wave has
  // everything it needs of a component after one 128-byte scalar load - no dependent load behind the record.
  LW_SEL_INLINE = 16 /* 8 words */, LW_BASES_INLINE = 24 /* 8 words */,
  LW_WORDS = 32
};
// A gather program moves bit fields of the packed f row to a destination bit vector.  It is a
// list of 4-word runs [ctl, mask, flip, 0], four runs per 64-byte chunk (one s_load_dwordx16; the
// last chunk is padded with mask = 0 runs):
//   ctl = src_shift | dst_shift << 8 | dst_word << 16 | src_word << 24
//   dst_word[dst_shift ..] |= ((f32[src_word] >> src_shift) & mask) ^ flip
#define TSIMK_LW_MAX_NOUT 12
#define TSIMK_LW_MAX_WEIGHT 7
#define TSIMK_LW_LISTS 64   // hard-row sub-lists (one atomic counter each, 128 bytes apart)
// Wide components (more than 64 parameters; first pass k_sample_lw<true> in front of the sparse-column kernel):
// f_sel positions below 256, patterns to weight 4, rank from a [4][256] binomial table; the table-build kernel
// gets no pattern list but unranks the pattern index itself.
#define TSIMK_LWW_MAX_WEIGHT 4
#define TSIMK_LWW_MAX_F 511   // (255 until round 5: positions were bytes; k_sample_wide<.., P16> keeps them in 16 bits)

// colex rank of a pattern with sorted set-bit positions b0 < b1 < b2 < b3 < b4 (missing ones passed as 0)
__host__ __device__ __forceinline__ uint32_t lw_binom2(uint32_t b) { return (b * (b - 1u)) >> 1; }
__host__ __device__ __forceinline__ uint32_t lw_binom3(uint32_t b) {
  // b(b-1)(b-2)/6 for b <= 64: the product is a multiple of 6 below 2^18, and
  // floor(v * 43691 / 2^18) == v / 6 for multiples of 6 up to 2^17 * 6 (error q * 2^-17 < 1)
  const uint32_t v = b * (b - 1u) * (b - 2u);
  return (uint32_t)(((unsigned long long)v * 43691ull) >> 18);
}
// C(b,4) = C(b,3) (b-3) / 4, exact; 0 for b < 4 (C(b,3) = 0 below 3, the factor is 0 at 3)
__host__ __device__ __forceinline__ uint32_t lw_binom4(uint32_t b) { return (lw_binom3(b) * (b - 3u)) >> 2; }
// C(b,5) = C(b,4) (b-4) / 5, exact (the product stays below 2^26 for b <= 64); 0 for b < 5
__host__ __device__ __forceinline__ uint32_t lw_binom5(uint32_t b) {
  const unsigned long long v = (unsigned long long)lw_binom4(b) * (unsigned long long)(b - 4u);  // 0 when C(b,4) == 0
  return (uint32_t)((v * 3435973837ull) >> 34);  // v / 5 for v < 2^32 (magic 0xCCCCCCCD)
}

// C(b,6) = C(b,5) (b-5) / 6 and C(b,7) = C(b,6) (b-6) / 7 for b <= 64: the products are exact multiples (< 2^32),
// so the division is a multiplication by the inverse modulo 2^32 (of 3 after halving, of 7)
__host__ __device__ __forceinline__ uint32_t lw_binom6(uint32_t b) {
  return b < 6u ? 0u : ((lw_binom5(b) * (b - 5u)) >> 1) * 0xAAAAAAABu;
}
__host__ __device__ __forceinline__ uint32_t lw_binom7(uint32_t b) { return b < 7u ? 0u : (lw_binom6(b) * (b - 6u)) * 0xB6DB6DB7u; }
// C(b, i + 1): the term set bit number i (in ascending order) at f_sel position b adds to the colex rank
__host__ __device__ __forceinline__ uint32_t lw_rank_term(int i, uint32_t b) {
  return (i == 0) ? b : (i == 1) ? lw_binom2(b) : (i == 2) ? lw_binom3(b) : (i == 3) ? lw_binom4(b)
       : (i == 4) ? lw_binom5(b) : (i == 5) ? lw_binom6(b) : lw_binom7(b);
}

// ---------------------------------------------------------------------------
// table build: one lane per (pattern, full assignment m of the n outputs).  The lane walks the
// levels exactly like run_component with the outcome bits forced to m and records the threshold
// of every node on its path (lanes sharing a prefix write the same value).
// ---------------------------------------------------------------------------
struct LwBuildArgs {
  const uint32_t *img;
  const unsigned long long *patbits;  // [npat] f_sel bit patterns in table order, or nullptr: the lane unranks its pattern
  uint32_t *tab;                      // this component's table: [npat << n_out] integer thresholds (bernoulli_threshold)
  int comp_off;                       // row-layout component record (C_*)
  int npat;
  // patbits == nullptr: weight class from `bases`, then the set bits from the largest down - the largest b with
  // C(b, i + 1) <= r.  Wide components (wide_binom_off != 0): binary search in the [4][256] table; narrow ones: the same
  // search on lw_rank_term (no 14-million-entry list built on the host and copied)
  int wide_binom_off, bases_off, wmax;
  int binom_stride;                   // words per row of the wide binomial table (256; 512 for components beyond 255 selected bits)
  float *p1;                          // scratch [npat << n_out]: |amp| of every node (k_lw_nodes), node 0 = the normalisation
  int depth;                          // k_lw_nodes: -1 = the normalisation level, d = the nodes with d prefix bits, -2 = all of them
  int pat_begin, pat_count;           // this launch serves patterns [pat_begin, pat_begin + pat_count) (pat_count 0: to npat) - a build in slices
  // chunked prefix tree (tsim_trie.hip.h): `tab` = the component's chunks, `p1` = scratch: a header of TH_WORDS words, then one
  // TrieMeta per chunk
  int trie;                           // 1: build that format
  int trie_level;                     // chunk level of this launch (outputs 3 L .. 3 L + 2)
  uint32_t trie_cap;                  // chunks the table holds
  int comp4, nch;                     // the component's chunk-table record (C4_*) and the chunks per tile, 0 = none: the prefix-tree nodes on the LDS chunk tables (tsim_build4.hip)
};

// the pattern of table row `pat` as f_sel-position bits
template <int W>
__device__ __forceinline__ void lw_pattern_bits(const LwBuildArgs &A, cptr img, uint32_t F, uint32_t pat, uint32_t (&x)[W]) {
  if (A.patbits) {
    const unsigned long long pb = A.patbits[pat];
#pragma unroll
    for (int w = 0; w < W; ++w) x[w] = (w < 2) ? (uint32_t)(pb >> (32 * w)) : 0u;
    return;
  }
#pragma unroll
  for (int w = 0; w < W; ++w) x[w] = 0u;
  cptr bases = img + A.bases_off;
  int wt = 0;
  for (int k = 1; k <= A.wmax; ++k)
    if (pat >= bases[k]) wt = k;
  uint32_t r = pat - bases[wt];
  int hi = (int)F;  // the next set bit lies below this position
  for (int i = wt - 1; i >= 0; --i) {
    int lo = i, up = hi - 1;  // C(i, i + 1) = 0 <= r: position i always qualifies
    uint32_t at_lo = 0u;
    if (A.wide_binom_off) {
      cptr bn = img + A.wide_binom_off;
      while (lo < up) {
        const int mid = (lo + up + 1) >> 1;
        if (bn[i * A.binom_stride + mid] <= r) lo = mid;
        else up = mid - 1;
      }
      at_lo = bn[i * A.binom_stride + lo];
    } else {
      while (lo < up) {
        const int mid = (lo + up + 1) >> 1;
        if (lw_rank_term(i, (uint32_t)mid) <= r) lo = mid;
        else up = mid - 1;
      }
      at_lo = lw_rank_term(i, (uint32_t)lo);
    }
    r -= at_lo;
#pragma unroll
    for (int w = 0; w < W; ++w)
      if (w == (lo >> 5)) x[w] |= 1u << (lo & 31);
    hi = lo;
  }
}

// ---------------------------------------------------------------------------
// Table build, round 3: the |amp| of a prefix-tree node depends on (pattern, prefix bits, trial bit) only - not on the
// chain of `prev` values above it - so every node is evaluated ONCE: k_lw_nodes, one launch per depth (all lanes of a
// launch evaluate the same level), one lane per (pattern, prefix); then k_lw_finish forms the thresholds p1 / prev
// with prev from the chain rule over the node's ancestors (sampler.py:75-79).  The first version gave a lane to every
// (pattern, full assignment) and let it walk all n + 1 levels of its path: (n + 1) x the evaluations (6 x for the
// distillation shapes: 24 -> 4 ms for C3's weight-5 tables, 180 -> 30 for weight 6).  Same device arithmetic as the
// sampling kernels (eval_any -> cabs32 -> __fdiv_rn / __fsub_rn), same values.
// ---------------------------------------------------------------------------
template <int W, bool FAST>
__global__ void __launch_bounds__(256) k_lw_nodes(LwBuildArgs A) {
  cptr img = (cptr)(uintptr_t)A.img;
  cptr comp = img + A.comp_off;
  const uint32_t n_out = comp[C_NOUT], F = comp[C_F];
  cptr levels = img + comp[C_LEVELS];
  int d = A.depth;
  const long long tl = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long np = (long long)(A.pat_count ? A.pat_count : A.npat - A.pat_begin);
  uint32_t pat, prefix;
  if (d == -2) {
    // every node of every pattern in ONE launch, node-major (a wave's lanes evaluate the same level but for its ends), the deepest
    // nodes first: one launch per depth was seven launches one behind the other, each as long as ONE lane's walk over the graphs
    // of its level when the patterns are few (C4 at weight 3: 10 701 patterns, 2.4 ms of a fresh handle's 7)
    if (tl >= (np << n_out)) return;
    const uint32_t node = ((1u << n_out) - 1u) - (uint32_t)(tl / np);
    pat = (uint32_t)A.pat_begin + (uint32_t)(tl % np);
    d = node == 0u ? -1 : 31 - __builtin_clz(node);
    prefix = d < 0 ? 0u : node - (1u << d);
  } else {
    const int dd = d < 0 ? 0 : d;
    if (tl >= (np << dd)) return;
    const long long t = ((long long)A.pat_begin << dd) + tl;
    pat = d < 0 ? (uint32_t)t : (uint32_t)(t >> d);
    prefix = d < 0 ? 0u : ((uint32_t)t & ((1u << d) - 1u));
  }
  uint32_t x[W];
  lw_pattern_bits<W>(A, img, F, pat, x);
  float re, im;
  float *row = A.p1 + ((size_t)pat << n_out);
  if (d < 0) {
    eval_any<W, FAST>(A.img, img, levels, x, re, im, nullptr);  // sampler.py:54
    row[0] = cabs32(re, im);
    return;
  }
  for (int i = 0; i <= d; ++i) {  // prefix bits (first output first), then the trial bit of output d
    const uint32_t bitpos = F + (uint32_t)i;
    const bool on = i == d ? true : (((prefix >> (d - 1 - i)) & 1u) != 0u);
#pragma unroll
    for (int w = 0; w < W; ++w)
      if ((uint32_t)w == (bitpos >> 5) && on) x[w] |= 1u << (bitpos & 31u);
  }
  eval_any<W, FAST>(A.img, img, levels + (d + 1) * L_WORDS, x, re, im, nullptr);  // sampler.py:65
  row[(1u << d) + prefix] = cabs32(re, im);
}

// thresholds from the node values: lane = (pattern, node)
template <bool FAST>
__global__ void __launch_bounds__(256) k_lw_finish(LwBuildArgs A) {
  cptr img = (cptr)(uintptr_t)A.img;
  const uint32_t n_out = (img + A.comp_off)[C_NOUT];
  const long long tl = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tl >= ((long long)(A.pat_count ? A.pat_count : A.npat - A.pat_begin) << n_out)) return;
  const long long t = ((long long)A.pat_begin << n_out) + tl;
  const uint32_t node = (uint32_t)t & ((1u << n_out) - 1u);
  const float *row = A.p1 + (((size_t)t >> n_out) << n_out);
  if (node == 0u) {
    A.tab[t] = 0u;  // node 0 is unused
    return;
  }
  const int d = 31 - __builtin_clz(node);  // depth: the node's prefix has d bits
  float prev = row[0];
  for (int i = 0; i < d; ++i) {  // the chain rule along the ancestors (sampler.py:79)
    const uint32_t anc = node >> (d - i);
    const bool bit = ((node >> (d - 1 - i)) & 1u) != 0u;
    const float p1 = row[anc];
    prev = bit ? p1 : __fsub_rn(prev, p1);
  }
  A.tab[t] = bernoulli_threshold(__fdiv_rn(row[node], prev));  // sampler.py:75: bit = u < p1 / prev
}

struct LwArgs {
  SampleArgs s;           // row_index/row_count: optional INPUT list (device-side post-selection)
  const uint32_t *tab;    // integer thresholds (bernoulli_threshold), all components
  int lw_off;             // image offset of the LW component records
  int direct_prog;        // image offset of the direct-output gather program (64-byte aligned)
  int direct_chunks;
  int direct_rot;         // image offset of the rotate-and-mask form of the same program (register first pass)
  int has_check;          // the first slot's row is the normalisation-check row: always "hard"
  uint32_t *hard_index;   // out: n_lists sub-lists of list_cap rows that need the full kernel
  uint32_t *ctl;          // ctl[32 k] = entries of list k (zeroed by the caller), ctl[32 LISTS] = check row
  int list_cap;
  uint32_t *ctl_next;     // the counter set of the NEXT launch: reset here (nobody else touches it now)
  int binom_off;          // image offset of the binomial table C(b, k + 1), [8][64] words (register first pass)
  int n_lists;            // sub-lists in use (power of two <= TSIMK_LW_LISTS): few hard rows -> few lists,
                          // so that the 64-row blocks of the second pass are well filled
};

}  // namespace tsimk
