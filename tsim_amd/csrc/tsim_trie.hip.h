// tsim_trie.hip.h - pattern tables as a pruned, chunked prefix tree (round 6).
//
// The dense format (tsim_lw.hip.h) stores thr[pattern][2^n_out]: every node of the outcome tree of every tabulated f_sel
// pattern.  That is what stops a component at 12 outputs (a pattern of n11 is 8 KB, of n16 256 KB) although the
// reference loops over any number of levels (src/tsim/sampler.py:62) and a node whose Bernoulli threshold is 0 or 1 has
// only ONE reachable child: real detector bits are mostly fixed by f and the few genuinely random outcomes before them.
//
// Here a pattern's tree is cut into CHUNKS of three levels (the ROOT chunk takes the n_out mod 3 odd ones: every other chunk
// is full, and a full tree of n16 is 9 363 chunks instead of 37 449) - 8 words, 32 bytes, one memory access of the first pass:
//   word 0      index of the first child chunk (the children of the reachable leaves lie side by side)
//   word 1      T(node 1) | child-present mask << 24     (T <= 2^23: bernoulli_threshold)
//   words 2..7  T of the chunk's nodes 2..7 (heap order: node 2 b + bit below node b)
// A leaf that no draw can reach (T = 0 kills the 1-branch, T = 2^23 the 0-branch) gets no child, and its subtree is never
// evaluated: the table is linear in the LIVE nodes.  Chunk `pat` (< npat) is the root of pattern `pat`; the others are
// handed out by an atomic counter while the tree is built breadth first, chunk level by chunk level, weight class after
// weight class (the slices of tsim_tables.hip), until the component's budget is spent - a reachable leaf without a child
// then sends the row to the hard list like a pattern beyond the table depth does.  Same arithmetic as the dense builder
// (eval_any -> cabs32 -> __fdiv_rn / __fsub_rn -> bernoulli_threshold): bit-identical thresholds.
#pragma once
#include "tsim_lw.hip.h"

namespace tsimk {

struct TrieMeta {  // what the build knows of a chunk that is not a root
  uint32_t pat;    // its pattern (root chunk)
  float prev;      // the chain-rule value at its top (sampler.py:79)
  uint32_t pre_lo, pre_hi;  // the 3 L outcome bits above it, first output most significant
};
// scratch header (uint32 words in front of the TrieMeta array)
enum { TH_NEXT = 0 /* next free chunk */, TH_BEGIN, TH_END /* chunks of the level being built */, TH_LOST /* children refused: budget */,
       TH_NEXT_BEGIN, TH_VALID_END /* chunks from here on were never assigned */, TH_WORDS = 16 };

// chunk level L holds the outputs [trie_first_output(L), + trie_outputs(L)): the root chunk the n_out mod 3 odd ones (3 if none)
__host__ __device__ __forceinline__ int trie_root_outputs(int n_out) { return n_out % 3 ? n_out % 3 : (n_out < 3 ? n_out : 3); }
__host__ __device__ __forceinline__ int trie_outputs(int n_out, int L) { return L == 0 ? trie_root_outputs(n_out) : 3; }
__host__ __device__ __forceinline__ int trie_first_output(int n_out, int L) { return L == 0 ? 0 : trie_root_outputs(n_out) + 3 * (L - 1); }
__host__ __device__ __forceinline__ int trie_levels(int n_out) { return n_out <= 0 ? 0 : 1 + (n_out - trie_root_outputs(n_out)) / 3; }

template <bool FAST>
__global__ void k_trie_begin(LwBuildArgs A) {
  uint32_t *h = reinterpret_cast<uint32_t *>(A.p1);
  if (A.pat_begin == 0) {
    h[TH_NEXT] = (uint32_t)A.npat;
    h[TH_LOST] = 0u;
    h[TH_VALID_END] = A.trie_cap;
  }
  h[TH_BEGIN] = (uint32_t)A.pat_begin;
  h[TH_END] = (uint32_t)(A.pat_begin + (A.pat_count ? A.pat_count : A.npat - A.pat_begin));
  h[TH_NEXT_BEGIN] = h[TH_NEXT] < h[TH_VALID_END] ? h[TH_NEXT] : h[TH_VALID_END];
}
template <bool FAST>
__global__ void k_trie_advance(LwBuildArgs A) {
  uint32_t *h = reinterpret_cast<uint32_t *>(A.p1);
  const uint32_t e = h[TH_NEXT] < h[TH_VALID_END] ? h[TH_NEXT] : h[TH_VALID_END];
  h[TH_BEGIN] = h[TH_NEXT_BEGIN];
  h[TH_END] = e;
  h[TH_NEXT_BEGIN] = e;
}

// |amp| of the chunk nodes at local depth A.depth (-1: the normalisation of root chunks), as float bits in the chunk's words
template <int W, bool FAST>
__global__ void __launch_bounds__(256) k_trie_nodes(LwBuildArgs A) {
  cptr img = (cptr)(uintptr_t)A.img;
  cptr comp = img + A.comp_off;
  const uint32_t F = comp[C_F];
  const int n_out = (int)comp[C_NOUT];
  cptr levels = img + comp[C_LEVELS];
  const uint32_t *h = reinterpret_cast<const uint32_t *>(A.p1);
  const TrieMeta *meta = reinterpret_cast<const TrieMeta *>(h + TH_WORDS);
  const uint32_t begin = h[TH_BEGIN], end = h[TH_END];
  const int L = A.trie_level, d = A.depth, dd = d < 0 ? 0 : d;
  const long long items = (long long)(end - begin) << dd;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < items; t += (long long)gridDim.x * blockDim.x) {
    const uint32_t chunk = begin + (uint32_t)(t >> dd);
    const uint32_t loc = (uint32_t)t & ((1u << dd) - 1u);  // the path inside the chunk, its first output most significant
    uint32_t *cw = A.tab + (size_t)chunk * 8u;
    uint32_t pat = chunk, pre_lo = 0u, pre_hi = 0u;
    float prev = 0.0f;
    if (L == 0) {
      if (d >= 0) prev = __uint_as_float(cw[0]);
    } else {
      const TrieMeta m = meta[chunk];
      pat = m.pat;
      prev = m.prev;
      pre_lo = m.pre_lo;
      pre_hi = m.pre_hi;
    }
    bool reach = true;  // can a draw take this path?  (the thresholds of the nodes above, as k_trie_finish forms them)
    {
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
    const uint32_t slot = d < 0 ? 0u : (1u << d) + loc;
    if (!reach) {
      cw[slot] = 0u;
      continue;
    }
    uint32_t x[W];
    lw_pattern_bits<W>(A, img, F, pat, x);
    float re, im;
    if (d < 0) {
      eval_any<W, FAST>(A.img, img, levels, x, re, im, nullptr);  // sampler.py:54
      cw[0] = __float_as_uint(cabs32(re, im));
      continue;
    }
    const int np = trie_first_output(n_out, L);
    auto set_bit = [&](uint32_t bitpos) {
#pragma unroll
      for (int w = 0; w < W; ++w)
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
    eval_any<W, FAST>(A.img, img, levels + (np + d + 1) * L_WORDS, x, re, im, nullptr);
    cw[slot] = __float_as_uint(cabs32(re, im));
  }
}

// node values -> thresholds, reachable leaves -> child chunks (one lane per chunk)
template <bool FAST>
__global__ void __launch_bounds__(256) k_trie_finish(LwBuildArgs A) {
  cptr img = (cptr)(uintptr_t)A.img;
  const int n_out = (int)(img + A.comp_off)[C_NOUT];
  uint32_t *h = reinterpret_cast<uint32_t *>(A.p1);
  TrieMeta *meta = reinterpret_cast<TrieMeta *>(h + TH_WORDS);
  const uint32_t begin = h[TH_BEGIN], end = h[TH_END];
  const int L = A.trie_level;
  const int rem = trie_outputs(n_out, L);
  const bool last = L + 1 >= trie_levels(n_out);
  for (uint32_t chunk = begin + blockIdx.x * blockDim.x + threadIdx.x; chunk < end; chunk += gridDim.x * blockDim.x) {
    uint32_t *cw = A.tab + (size_t)chunk * 8u;
    uint32_t pat = chunk, pre_lo = 0u, pre_hi = 0u;
    float prev0;
    if (L == 0) prev0 = __uint_as_float(cw[0]);
    else {
      const TrieMeta m = meta[chunk];
      pat = m.pat;
      prev0 = m.prev;
      pre_lo = m.pre_lo;
      pre_hi = m.pre_hi;
    }
    float p[8];
#pragma unroll
    for (int j = 1; j < 8; ++j) p[j] = __uint_as_float(cw[j]);
    uint32_t T[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    // the chain rule down the chunk (sampler.py:75-79); pv[k] / rc[k]: prev and reachability of the node / leaf with heap index k
    float pv[16];
    bool rc[16];
    pv[1] = prev0;
    rc[1] = true;
#pragma unroll
    for (int node = 1; node < 8; ++node) {
      const int dep = node < 2 ? 0 : (node < 4 ? 1 : 2);
      if (dep < rem) {
        T[node] = rc[node] ? bernoulli_threshold(__fdiv_rn(p[node], pv[node])) : 0u;
        pv[2 * node] = __fsub_rn(pv[node], p[node]);
        pv[2 * node + 1] = p[node];
        rc[2 * node] = rc[node] && T[node] != (1u << 23);
        rc[2 * node + 1] = rc[node] && T[node] != 0u;
      } else {
        pv[2 * node] = pv[2 * node + 1] = 0.0f;
        rc[2 * node] = rc[2 * node + 1] = false;
      }
    }
    uint32_t mask = 0u, base = 0u;
    if (!last) {
      // the chunk's leaves: heap nodes 2^rem .. 2^(rem + 1) - 1 (rem = 3 but for the root chunk)
#pragma unroll
      for (int leaf = 0; leaf < 8; ++leaf) {
        const bool r = rem == 3 ? rc[8 + leaf] : (rem == 2 ? (leaf < 4 && rc[4 + (leaf & 3)]) : (leaf < 2 && rc[2 + (leaf & 1)]));
        mask |= r ? (1u << leaf) : 0u;
      }
      const uint32_t cnt = (uint32_t)__builtin_popcount(mask);
      base = atomicAdd(&h[TH_NEXT], cnt);
      if (base + cnt > A.trie_cap || base + cnt < base) {  // the budget is spent: rows that come this way are hard rows
        atomicMin(&h[TH_VALID_END], base < A.trie_cap ? base : A.trie_cap);
        atomicAdd(&h[TH_LOST], cnt);
        mask = 0u;
        base = 0u;
      } else {
        uint32_t at = base;
#pragma unroll
        for (int leaf = 0; leaf < 8; ++leaf)
          if ((mask >> leaf) & 1u) {
            TrieMeta m;
            m.pat = pat;
            m.prev = rem == 3 ? pv[8 + leaf] : (rem == 2 ? pv[4 + (leaf & 3)] : pv[2 + (leaf & 1)]);
            m.pre_lo = (pre_lo << rem) | (uint32_t)leaf;  // (rem < 3 only in the root chunk: nothing above it)
            m.pre_hi = (pre_hi << rem) | (rem == 3 ? (pre_lo >> 29) : 0u);
            meta[at++] = m;
          }
      }
    }
    cw[0] = base;
    cw[1] = T[1] | (mask << 24);
#pragma unroll
    for (int j = 2; j < 8; ++j) cw[j] = T[j];
  }
}

}  // namespace tsimk
