// tsim_build4.hip - the pattern-table build on the LDS chunk tables (round 6).
//
// k_lw_nodes (tsim_lw.hip.h) gives every node of a pattern's prefix tree a lane that walks the graphs of the node's level row by
// row - the row kernel's arithmetic, ~100 dot products per graph.  Programs that have chunk tables (p->v4: the "Four Russians"
// layout of tsim_kernel4.hip.h) can evaluate the same node with one 16-byte LDS read per 4-bit chunk of x and graph: the g.f
// kernel's eval_level4 - same exact sums, same float conversion, same |amp| (every sampling kernel agrees with the oracle bit
// for bit; tests/test_gpu_pattern_tables.py compares the tables of both builders word for word).  A block serves ONE level
// (its tiles stream through LDS once per block), the levels' blocks lie one behind the other in the grid, deepest first.
// k_lw_finish (thresholds from the node values) is unchanged.
#include "tsim_internal.hip.h"
#include "tsim_kernel4.hip.h"
#include "tsim_trie.hip.h"

using namespace tsimk;

namespace tsimk {

template <int GT, int NCH>
__global__ void __launch_bounds__(256) k_lw_nodes4(LwBuildArgs A, int comp4) {
  cptr img = (cptr)(uintptr_t)A.img;
  cptr comp = img + comp4;  // the component's C4 record: words 0..7 = its C record
  const uint32_t n_out = comp[C_NOUT], F = comp[C_F];
  cptr levels = img + comp[C4_LEVELS];
  const long long np = (long long)(A.pat_count ? A.pat_count : A.npat - A.pat_begin);
  // block -> level (block-uniform): level L = 0 is the normalisation, L = d + 1 holds the 2^d prefixes of d bits
  long long b = blockIdx.x;
  int L = (int)n_out;
  long long lanes = 0;
  for (; L >= 0; --L) {
    lanes = np << (L > 0 ? L - 1 : 0);
    const long long nb = (lanes + 255) / 256;
    if (b < nb) break;
    b -= nb;
  }
  if (L < 0) return;
  const long long tl = b * 256 + threadIdx.x;
  const bool active = tl < lanes;
  const int d = L - 1;
  const uint32_t prefix = active ? (uint32_t)(tl / np) : 0u;  // node-major: neighbours share the prefix
  const uint32_t pat = (uint32_t)A.pat_begin + (active ? (uint32_t)(tl % np) : 0u);
  constexpr int XW = NCH > 24 ? 4 : NCH > 16 ? 3 : 2;  // words of x (sample4_block)
  uint32_t x[XW];
  lw_pattern_bits<XW>(A, img, F, pat, x);
  for (int i = 0; i <= d; ++i) {  // prefix bits (first output first), then the trial bit of output d
    const uint32_t bitpos = F + (uint32_t)i;
    const bool on = i == d ? true : (((prefix >> (d - 1 - i)) & 1u) != 0u);
#pragma unroll
    for (int w = 0; w < XW; ++w)
      if ((uint32_t)w == (bitpos >> 5) && on) x[w] |= 1u << (bitpos & 31u);
  }
  if (!active) {
#pragma unroll
    for (int w = 0; w < XW; ++w) x[w] = 0u;
  }
  uint32_t en[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const uint32_t w = x[(c >> 3) < XW ? (c >> 3) : XW - 1];
    en[c] = ((w >> (4 * (c & 7))) & 15u) * 16u + c * Tile4<GT>::kChunkBytes;
  }
  cptr lvl = levels + L * L4_WORDS;
  uint8_t *lds_tab = reinterpret_cast<uint8_t *>(tsimk_lds);
  float re, im;
  if (lvl[L4_FLAGS] & TSIMK_LFLAG_FIXED) eval_level4<GT, NCH, true, 256>(A.img, img, lvl, en, NCH * Tile4<GT>::kChunkBytes, lvl[L4_TABLES], lds_tab, re, im);
  else eval_level4<GT, NCH, false, 256>(A.img, img, lvl, en, NCH * Tile4<GT>::kChunkBytes, lvl[L4_TABLES], lds_tab, re, im);
  if (!active) return;
  float *row = A.p1 + ((size_t)pat << n_out);
  row[d < 0 ? 0u : (1u << d) + prefix] = cabs32(re, im);
}

// The prefix-tree builder's node pass (k_trie_nodes, tsim_trie.hip.h) on the chunk tables: the chunk nodes at local depth A.depth
// of chunk level A.trie_level - one level of the component per launch, so a block's lanes share the tiles.  Same bookkeeping, same
// values; the loop's trip count is block-uniform (eval_level4 has barriers), idle lanes carry x = 0.
template <int GT, int NCH>
__global__ void __launch_bounds__(256) k_trie_nodes4(LwBuildArgs A) {
  cptr img = (cptr)(uintptr_t)A.img;
  cptr comp = img + A.comp4;
  const uint32_t F = comp[C_F];
  const int n_out = (int)comp[C_NOUT];
  cptr levels = img + comp[C4_LEVELS];
  const uint32_t *h = reinterpret_cast<const uint32_t *>(A.p1);
  const TrieMeta *meta = reinterpret_cast<const TrieMeta *>(h + TH_WORDS);
  const uint32_t begin = h[TH_BEGIN], end = h[TH_END];
  const int L = A.trie_level, d = A.depth, dd = d < 0 ? 0 : d;
  const long long items = (long long)(end - begin) << dd;
  const int np = trie_first_output(n_out, L);
  cptr lvl = levels + (d < 0 ? 0 : np + d + 1) * L4_WORDS;
  const bool fixed = (lvl[L4_FLAGS] & TSIMK_LFLAG_FIXED) != 0;
  uint8_t *lds_tab = reinterpret_cast<uint8_t *>(tsimk_lds);
  constexpr int XW = NCH > 24 ? 4 : NCH > 16 ? 3 : 2;
  for (long long base = (long long)blockIdx.x * blockDim.x; base < items; base += (long long)gridDim.x * blockDim.x) {
    const long long t = base + threadIdx.x;
    const bool active = t < items;
    const uint32_t chunk = begin + (active ? (uint32_t)(t >> dd) : 0u);
    const uint32_t loc = active ? ((uint32_t)t & ((1u << dd) - 1u)) : 0u;  // the path inside the chunk, its first output most significant
    uint32_t *cw = A.tab + (size_t)chunk * 8u;
    uint32_t pat = chunk, pre_lo = 0u, pre_hi = 0u;
    float prev = 0.0f;
    bool reach = active;  // can a draw take this path?  (the thresholds of the nodes above, as k_trie_finish forms them)
    if (active) {
      if (L == 0) {
        if (d >= 0) prev = __uint_as_float(cw[0]);
      } else {
        const TrieMeta m = meta[chunk];
        pat = m.pat;
        prev = m.prev;
        pre_lo = m.pre_lo;
        pre_hi = m.pre_hi;
      }
      float pv = prev;
      uint32_t node = 1u;
      for (int k = 0; k < d; ++k) {
        const float p1 = __uint_as_float(cw[node]);
        const uint32_t T = bernoulli_threshold(__fdiv_rn(p1, pv));
        const bool bit = ((loc >> (d - 1 - k)) & 1u) != 0u;
        if (bit ? T == 0u : T == (1u << 23)) reach = false;
        pv = bit ? p1 : __fsub_rn(pv, p1);
        node = 2u * node + (bit ? 1u : 0u);
      }
    }
    uint32_t x[XW];
#pragma unroll
    for (int w = 0; w < XW; ++w) x[w] = 0u;
    if (reach) {
      lw_pattern_bits<XW>(A, img, F, pat, x);
      if (d >= 0) {
        auto set_bit = [&](uint32_t bitpos) {
#pragma unroll
          for (int w = 0; w < XW; ++w)
            if ((uint32_t)w == (bitpos >> 5)) x[w] |= 1u << (bitpos & 31u);
        };
        for (int i = 0; i < np; ++i) {
          const int sh = np - 1 - i;
          const bool on = sh >= 32 ? ((pre_hi >> (sh - 32)) & 1u) != 0u : ((pre_lo >> sh) & 1u) != 0u;
          if (on) set_bit(F + (uint32_t)i);
        }
        for (int k = 0; k < d; ++k)
          if ((loc >> (d - 1 - k)) & 1u) set_bit(F + (uint32_t)(np + k));
        set_bit(F + (uint32_t)(np + d));  // the trial bit (sampler.py:65)
      }
    }
    uint32_t en[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const uint32_t w = x[(c >> 3) < XW ? (c >> 3) : XW - 1];
      en[c] = ((w >> (4 * (c & 7))) & 15u) * 16u + c * Tile4<GT>::kChunkBytes;
    }
    float re, im;
    if (fixed) eval_level4<GT, NCH, true, 256>(A.img, img, lvl, en, NCH * Tile4<GT>::kChunkBytes, lvl[L4_TABLES], lds_tab, re, im);
    else eval_level4<GT, NCH, false, 256>(A.img, img, lvl, en, NCH * Tile4<GT>::kChunkBytes, lvl[L4_TABLES], lds_tab, re, im);
    if (!active) continue;
    const uint32_t slot = d < 0 ? 0u : (1u << d) + loc;
    cw[slot] = reach ? __float_as_uint(cabs32(re, im)) : 0u;
  }
}

}  // namespace tsimk

int tsim_launch_trie_nodes4(const LwBuildArgs &a, unsigned grid, hipStream_t s) {
  constexpr int GT = 4;
  const size_t lds = 2 * (size_t)a.nch * Tile4<GT>::kChunkBytes;
  switch (a.nch) {
#define TSIM_T4(N) case N: hipLaunchKernelGGL((k_trie_nodes4<GT, N>), dim3(grid), dim3(256), lds, s, a); break;
    TSIM_T4(2) TSIM_T4(4) TSIM_T4(6) TSIM_T4(8) TSIM_T4(10) TSIM_T4(12) TSIM_T4(14) TSIM_T4(16) TSIM_T4(20) TSIM_T4(32)
#undef TSIM_T4
    default: return tsim_fail(TSIM_ESTATE, "bad chunk count %d", a.nch);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// every node of the slice's trees, then the thresholds (the protocol of tsimrows::lw_build: `a.p1` scratch, `a.tab` table)
int tsim_launch_lw_build4(tsim_program *p, int ci, const LwBuildArgs &a, int n_out, hipStream_t s) {
  constexpr int GT = 4;
  const long long np = (long long)(a.pat_count ? a.pat_count : a.npat - a.pat_begin);
  if (np <= 0) return 0;
  long long blocks = 0;
  for (int L = 0; L <= n_out; ++L) blocks += ((np << (L > 0 ? L - 1 : 0)) + 255) / 256;
  if (blocks > 0x7FFFFFFFll) return tsim_fail(TSIM_ENOTSUP, "pattern tables: slice too large");
  const int comp4 = p->comp4_off + ci * C4_WORDS;
  const int nch = p->v4_max_nch;
  const size_t lds = 2 * (size_t)nch * Tile4<GT>::kChunkBytes;
  switch (nch) {
#define TSIM_B4(N)                                                                                                     \
  case N: {                                                                                                            \
    auto kfn = k_lw_nodes4<GT, N>;                                                                                     \
    hipLaunchKernelGGL(kfn, dim3((unsigned)blocks), dim3(256), lds, s, a, comp4);                                      \
  } break;
    TSIM_B4(2) TSIM_B4(4) TSIM_B4(6) TSIM_B4(8) TSIM_B4(10) TSIM_B4(12) TSIM_B4(14) TSIM_B4(16) TSIM_B4(20) TSIM_B4(32)
#undef TSIM_B4
    default: return tsim_fail(TSIM_ESTATE, "bad chunk count %d", nch);
  }
  HIP_TRY(hipGetLastError());
  const long long lanes = np << n_out;
  hipLaunchKernelGGL((k_lw_finish<true>), dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, s, a);
  HIP_TRY(hipGetLastError());
  return 0;
}
