// tsim_lw_pass.hip.h - the first pass of a two-pass launch (k_sample_lw); see tsim_lw.hip.h.
#pragma once
#include "tsim_lw.hip.h"

namespace tsimk {

// Optional timeline of the register first pass (build with -DTSIMK_LW_TRACE, scripts/lw_trace.py): wave 0 of the first
// blocks stamps s_memtime at the phase boundaries of its first two rows.  Compiled out by default.
#ifdef TSIMK_LW_TRACE
__device__ unsigned long long tsimk_lw_trace[16 * 32];
#define LW_STAMP(k)                                                                                           \
  do {                                                                                                        \
    if (threadIdx.x == 0 && blockIdx.x < 16 && tr_it < 2) {                                                   \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                             \
      tsimk_lw_trace[blockIdx.x * 32 + tr_it * 12 + (k)] = __builtin_readcyclecounter();                      \
    }                                                                                                         \
  } while (0)
#define LW_STAMP_ABS(slot)                                                                                    \
  do {                                                                                                        \
    if (threadIdx.x == 0 && blockIdx.x < 16) {                                                                \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                             \
      tsimk_lw_trace[blockIdx.x * 32 + (slot)] = __builtin_readcyclecounter();                                \
    }                                                                                                         \
  } while (0)
#else
#define LW_STAMP(k) do { } while (0)
#define LW_STAMP_ABS(slot) do { } while (0)
#endif

// ---------------------------------------------------------------------------
// The threshold walk of one component: for output i, bit_i = u_i < thr[node], node <- 2 node + bit_i
// (sampler.py:74-79 with the thresholds tabulated).  Walking node by node is a chain of n_out DEPENDENT 4-byte
// gathers - five memory latencies per shot for C2, most of a wave's lifetime.  The tree of a pattern is one
// contiguous row (2^n_out floats, level k at [2^k, 2^(k+1))), so the next THREE levels below a node are one
// 4-byte, one 8-byte and one 16-byte word (thr + node, thr + 2 node, thr + 4 node): they are loaded together,
// the three Threefry blocks run while they are in flight, and the bits are picked with selects - one latency
// per three outputs, the same comparisons on the same floats.
// ---------------------------------------------------------------------------
template <class Key, class Emit>
__device__ __forceinline__ void lw_walk_impl(Key key, const uint32_t *thr, uint32_t n_out, uint32_t keybase,
                                             unsigned long long shot, Emit emit, int tr_it) {
  (void)tr_it;
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  // the draw of output o as the 23 bits the reference turns into its uniform (bit = m < T, see bernoulli_threshold)
#if defined(TSIMK_LWM_SKIP) && (TSIMK_LWM_SKIP & 1)
  auto draw = [&](uint32_t o) { return ((uint32_t)shot * 0x9E3779B9u + key(o, 0u)) >> 9; };  // diagnostic: no Threefry
#else
  auto draw = [&](uint32_t o) { return threefry_bits32(key(o, 0u), key(o, 1u), shot) >> 9; };
#endif
  uint32_t node = 1u, i = 0u;
#if defined(TSIMK_LWM_SKIP) && (TSIMK_LWM_SKIP & 32)
  {  // diagnostic: thresholds without memory
    const uint32_t base = (uint32_t)(uintptr_t)thr;
    for (; i < n_out; ++i) {
      const bool b = draw(keybase + i) < ((base * 2654435761u + node * 40503u) & 0x7FFFFFu);
      node = 2u * node + (b ? 1u : 0u);
      emit(i, b);
    }
    return;
  }
#endif
  for (; i + 3u <= n_out; i += 3u) {
    const uint32_t t0 = thr[node];
    const u32x2 t1 = *reinterpret_cast<const u32x2 *>(thr + 2u * node);
    const u32x4 t2 = *reinterpret_cast<const u32x4 *>(thr + 4u * node);
    const uint32_t m0 = draw(keybase + i), m1 = draw(keybase + i + 1u), m2 = draw(keybase + i + 2u);
#ifdef TSIMK_LW_TRACE
    if (threadIdx.x == 0 && blockIdx.x < 16 && tr_it < 2) {  // draws done, loads possibly still in flight
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      asm volatile("" :: "v"(m0), "v"(m1), "v"(m2));
      tsimk_lw_trace[blockIdx.x * 32 + tr_it * 12 + 5] = __builtin_readcyclecounter();
    }
#endif
    const bool b0 = m0 < t0;
    const bool b1 = m1 < (b0 ? t1.y : t1.x);
    const uint32_t lo = b1 ? t2.y : t2.x, hi = b1 ? t2.w : t2.z;
    const bool b2 = m2 < (b0 ? hi : lo);
    node = 8u * node + (b0 ? 4u : 0u) + (b1 ? 2u : 0u) + (b2 ? 1u : 0u);
    emit(i, b0);
    emit(i + 1u, b1);
    emit(i + 2u, b2);
    LW_STAMP(6);
  }
  if (n_out - i == 2u) {
    const uint32_t t0 = thr[node];
    const u32x2 t1 = *reinterpret_cast<const u32x2 *>(thr + 2u * node);
    const uint32_t m0 = draw(keybase + i), m1 = draw(keybase + i + 1u);
    const bool b0 = m0 < t0;
    const bool b1 = m1 < (b0 ? t1.y : t1.x);
    emit(i, b0);
    emit(i + 1u, b1);
    LW_STAMP(7);
  } else if (n_out - i == 1u) {
    const uint32_t m0 = draw(keybase + i);
    emit(i, m0 < thr[node]);
  }
}

// subkeys from the launch's SampleArgs (inline in the kernel arguments or the k_keygen buffer) ...
template <class Emit>
__device__ __forceinline__ void lw_walk(const SampleArgs &A, const uint32_t *thr, uint32_t n_out, uint32_t keybase,
                                        unsigned long long shot, Emit emit, int tr_it = 0) {
  lw_walk_impl([&](uint32_t o, uint32_t j) { return subkey(A, o, j); }, thr, n_out, keybase, shot, emit, tr_it);
}
// ... or from a wave-uniform key array [output][2] (the fused multi-batch pass: one array per batch)
template <class Emit>
__device__ __forceinline__ void lw_walk_keys(cptr kp, const uint32_t *thr, uint32_t n_out, uint32_t keybase,
                                             unsigned long long shot, Emit emit) {
  lw_walk_impl([&](uint32_t o, uint32_t j) { return kp[2u * o + j]; }, thr, n_out, keybase, shot, emit, 0);
}

// the lane's set bits of one masked f word: each adds C(position inside f_sel, ordinal + 1) to the colex rank
template <int STRIDE = 64>
__device__ __forceinline__ void lw_rank_word(uint32_t mw, uint32_t sw, uint32_t base, const uint32_t *binom_lds, uint32_t &ord,
                                             uint32_t &pat) {
  while (mw) {
    const uint32_t p = (uint32_t)__builtin_ctz(mw);
    const uint32_t b = base + (uint32_t)__builtin_popcount(sw & ((1u << p) - 1u));
    pat += binom_lds[ord * (uint32_t)STRIDE + b];
    ++ord;
    mw &= mw - 1u;
  }
}


// ---------------------------------------------------------------------------
// pass 1
// ---------------------------------------------------------------------------
// WIDE = false: components of at most 64 parameters (f_sel gathered into a 64-bit word, rank by arithmetic).
// WIDE = true : components of up to 255 f_sel bits with ascending f_selection (the first pass in front of the
//   sparse-column kernel): weight and rank straight from the f row through the component's selection masks, as the
//   register form does, the binomials C(b, k + 1), b < 256, k < 4 from an LDS table behind the staging columns.
template <bool WIDE>
__global__ void __launch_bounds__(1024) k_sample_lw(LwArgs L) {
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

  if (blockIdx.x == 0 && threadIdx.x <= TSIMK_LW_LISTS)
    L.ctl_next[32u * threadIdx.x] = (threadIdx.x == TSIMK_LW_LISTS) ? 0xFFFFFFFFu : 0u;  // last: "no check row"
  uint32_t *binom_lds = tsimk_lds + (WF32 + WO32) * nthr;  // WIDE: [4][256] words; every wave writes all of it
  if constexpr (WIDE) {                                   // itself and reads only after its own stores: no barrier
    const uint4 *src = reinterpret_cast<const uint4 *>(A.img + L.binom_off);
    uint4 *dst = reinterpret_cast<uint4 *>(binom_lds);
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 4; ++k) dst[64 * k + lane] = src[64 * k + lane];
    __builtin_amdgcn_wave_barrier();
  }
  bool hard = false;
  uint32_t o0 = 0, o1 = 0;  // output words 0 and 1 (the LDS column holds the others)
  if (active) {
    stage_f_row(A.f + row * A.WF, A.WF, lds_f, nthr);
    for (int w = 0; w < WO32; ++w) lds_o[w * nthr] = 0u;
    hard = L.has_check && slot == 0;
    if (hard) L.ctl[32 * TSIMK_LW_LISTS] = (uint32_t)row;

    // K14: direct outputs f[idx] ^ flip (sampler.py:140-145), as bit-field moves
    gather_runs(img + L.direct_prog, (uint32_t)L.direct_chunks, lds_f, lds_o, nthr, o0, o1);

    for (int ci = 0; ci < A.n_comp; ++ci) {
      cptr rec = img + L.lw_off + ci * LW_WORDS;
      uint32_t pat;
      if constexpr (!WIDE) {
        // f_sel gather (sampler.py:48) -> x, then the weight test
        uint32_t x0 = 0, x1 = 0;
        gather_runs(img + rec[LW_FSELP], rec[LW_FSELN], lds_f, nullptr, nthr, x0, x1);
        unsigned long long xf = ((unsigned long long)x1 << 32) | x0;
        const uint32_t cnt = (uint32_t)__popcll(xf);
        if (cnt > rec[LW_WMAX]) hard = true;
        if (hard) continue;  // needs the full kernel: nothing of this row is written here
        // colex rank of the pattern: set bits in ascending order, bit number i at position b adds C(b, i + 1)
        pat = (img + rec[LW_BASES])[cnt];
#pragma unroll
        for (int i = 0; i < TSIMK_LW_MAX_WEIGHT; ++i) {
          if (cnt > (uint32_t)i) {
            pat += lw_rank_term(i, (uint32_t)__builtin_ctzll(xf));
            xf &= xf - 1ull;
          }
        }
      } else {
        // 16 mask words (f bits 0..511) and 16 prefix counts: two 64-byte scalar loads, then static indices only
        // (a scalar load per word inside the loops made a chain of ~60 dependent scalar-memory latencies per wave)
        const lw_u32x16 selm = *(lw_cptr16)(img + rec[LW_SELMASK]);
        const lw_u32x16 selp = *(lw_cptr16)(img + rec[LW_SELMASK] + 16u);
        const int nw = WF32 < 16 ? WF32 : 16;
        uint32_t mw[16];
        uint32_t cnt = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
          mw[w] = (w < nw) ? (lds_f[w * nthr] & selm[w]) : 0u;
          cnt += (uint32_t)__builtin_popcount(mw[w]);
        }
        if (cnt > rec[LW_WMAX]) hard = true;
        if (hard) continue;
        pat = rec[LW_BASES_INLINE];
#pragma unroll
        for (uint32_t w = 1; w <= TSIMK_LWW_MAX_WEIGHT; ++w) {
          const uint32_t bw = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec[LW_BASES_INLINE + w]);
          pat = (cnt == w) ? bw : pat;
        }
        uint32_t ord = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w)
          if (w < nw) lw_rank_word<256>(mw[w], selm[w], selp[w], binom_lds, ord, pat);
      }
      const uint32_t n_out = rec[LW_NOUT];
      cptr outpos = img + rec[LW_OUTPOS];
      const uint32_t keybase = rec[LW_KEYBASE];
      const uint32_t *thr = L.tab + rec[LW_TAB] + ((size_t)pat << n_out);
      lw_walk(A, thr, n_out, keybase, shot, [&](uint32_t i, bool bit) {
        const uint32_t dst = outpos[i];
        const uint32_t v = (bit ? 1u : 0u) << (dst & 31u);
        if ((dst >> 5) == 0u) o0 |= v;
        else if ((dst >> 5) == 1u) o1 |= v;
        else lds_o[(dst >> 5) * nthr] |= v;
      });
    }
    if (!hard) {
      if (A.out) {
        uint64_t *orow = A.out + row * A.WO;
        if (A.WO > 0) orow[0] = (uint64_t)o0 | ((uint64_t)o1 << 32);
        for (int w = 1; w < A.WO; ++w)
          orow[w] = (uint64_t)lds_o[(2 * w) * nthr] | ((uint64_t)lds_o[(2 * w + 1) * nthr] << 32);
      }
      // words 0 and 1 live in registers here
      store_compact_words(A, row, [&](int w) { return (w == 0) ? o0 : (w == 1) ? o1 : lds_o[w * nthr]; });
    }
  }

  // wave-aggregated append of the hard rows
  const unsigned long long hm = __ballot(hard ? 1 : 0);
  if (hm != 0ull) {
    const int lane = (int)(threadIdx.x & 63u);
    const int leader = __builtin_ctzll(hm);
    uint32_t basei = 0;
    const uint32_t k = blockIdx.x % (uint32_t)L.n_lists;  // this block's sub-list
    if (lane == leader) basei = atomicAdd(&L.ctl[32u * k], (uint32_t)__popcll(hm));
    basei = (uint32_t)__shfl((int)basei, leader, 64);
    if (hard)
      L.hard_index[(size_t)k * L.list_cap + basei + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull))] = (uint32_t)row;
  }
}

// ---------------------------------------------------------------------------
// pass 1, register form: narrow programs (f rows of at most 128 bits, at most 64 outputs, every component's
// f_selection ascending - what the reference's compiler emits, pipeline.py:141-143).  Nothing is staged in LDS:
//   * the f row lives in WF32 VGPRs; direct outputs are rotate-and-mask runs on those registers;
//   * f_sel is never materialised: the weight test is popcount(f & selection mask) per word, and the colex rank
//     of the pattern is accumulated from the set bits directly - the position of a set bit inside f_sel is the
//     number of selected bits below it, popcount(mask & (2^p - 1)) plus the word's prefix count (both wave-uniform
//     operands).  Lanes loop over THEIR set bits only (mean weight 0.6 at the benchmark's noise level) instead
//     of every lane running the whole gather program (one run per contiguous bit field, ~16 for C2).
// Same thresholds, same Threefry draws, same hard-row protocol as k_sample_lw.
// ---------------------------------------------------------------------------
// Direct outputs from registers, rotate-and-mask form (tsim_pack.hip: emit_rotmask_program): the loops over source
// and destination word are compile-time, a run is v_alignbit + v_and_or.
template <int WF32>
__device__ __forceinline__ void lw_direct_reg(cptr prog, uint32_t f0, uint32_t f1, uint32_t f2, uint32_t f3,
                                              uint32_t &o0, uint32_t &o1) {
  typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
  typedef const __attribute__((address_space(4))) u32x8 *cptr8;
#pragma unroll
  for (int s = 0; s < WF32; ++s) {
    const uint32_t src = (s == 0) ? f0 : (s == 1) ? f1 : (s == 2) ? f2 : f3;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const uint32_t groups = prog[s * 2 + d];
      cptr q = prog + prog[8 + s * 2 + d];
      uint32_t acc = 0u;
      for (uint32_t g = 0; g < groups; ++g) {
        const u32x8 v = *(cptr8)(q + 8u * g);
        acc |= __builtin_amdgcn_alignbit(src, src, v[0]) & v[1];
        acc |= __builtin_amdgcn_alignbit(src, src, v[2]) & v[3];
        acc |= __builtin_amdgcn_alignbit(src, src, v[4]) & v[5];
        acc |= __builtin_amdgcn_alignbit(src, src, v[6]) & v[7];
      }
      if (d == 0) o0 |= acc;
      else o1 |= acc;
    }
  }
  o0 ^= prog[16];
  o1 ^= prog[17];
}

#ifndef TSIMK_LW_SGPRS
#define TSIMK_LW_SGPRS 80
#endif
template <int WF32>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(TSIMK_LW_SGPRS))) k_sample_lw_reg(LwArgs L) {
  const SampleArgs &A = L.s;
  const int nthr = blockDim.x;
  LW_STAMP_ABS(24);
  long long n_rows = A.B;
  if (A.row_index) n_rows = (long long)*A.row_count;
  cptr img = (cptr)(uintptr_t)A.img;

  if (blockIdx.x == 0 && threadIdx.x <= TSIMK_LW_LISTS)
    L.ctl_next[32u * threadIdx.x] = (threadIdx.x == TSIMK_LW_LISTS) ? 0xFFFFFFFFu : 0u;  // last: "no check row"
  // C(b, k + 1) for b < 64, k < 8 (packer: tsim_program.hip).  Every WAVE writes the whole 2 KB table itself (all
  // waves store the same values) and only ever reads after its own stores - LDS operations of one wave are in
  // order - so no block barrier is needed: a wave starts as soon as ITS two 16-byte loads are back.
  __shared__ uint4 binom_lds4[128];
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(A.img + L.binom_off);
    const int lane = threadIdx.x & 63;
    binom_lds4[lane] = src[lane];
    binom_lds4[64 + lane] = src[64 + lane];
    __builtin_amdgcn_wave_barrier();
  }
  const uint32_t *binom_lds = reinterpret_cast<const uint32_t *>(binom_lds4);
  int tr_it = 0;
  (void)tr_it;
  LW_STAMP_ABS(25);
  // Grid-stride over the rows: the launch puts as many blocks on the chip as fit AT ONCE (tsim_sample.hip) and every
  // wave takes several rows per lane in turn - one round of waves instead of several, each with its ramp and
  // tail, and the per-wave preamble (kernel arguments, binomial table) paid once.  The trip count is uniform
  // over a block, so the wave-wide append at the end of the body is executed by every lane.
  for (long long base = (long long)blockIdx.x * nthr; base < n_rows; base += (long long)gridDim.x * nthr, ++tr_it) {
  LW_STAMP(0);
  const long long slot = base + threadIdx.x;
  const bool active = slot < n_rows;
  long long row = slot;
  if (A.row_index) row = active ? (long long)A.row_index[slot] : 0;
  const unsigned long long shot = (unsigned long long)(A.shot_offset + row);
  bool hard = false;
  if (active) {
    uint32_t f0, f1, f2 = 0u, f3 = 0u;  // scalars, not an array: a dynamically indexed one would be parked in LDS
    {
      const uint64_t *frow = A.f + row * (WF32 / 2);
      const uint64_t v0 = frow[0];
      f0 = (uint32_t)v0;
      f1 = (uint32_t)(v0 >> 32);
      if constexpr (WF32 == 4) {
        const uint64_t v1 = frow[1];
        f2 = (uint32_t)v1;
        f3 = (uint32_t)(v1 >> 32);
      }
    }
    hard = L.has_check && slot == 0;
    if (hard) L.ctl[32 * TSIMK_LW_LISTS] = (uint32_t)row;
    LW_STAMP(1);
    uint32_t o0 = 0, o1 = 0;
    lw_direct_reg<WF32>(img + L.direct_rot, f0, f1, f2, f3, o0, o1);  // K14, sampler.py:140-145
#ifdef TSIMK_LW_TRACE
    asm volatile("" :: "v"(o0), "v"(o1));
#endif
    LW_STAMP(2);

    for (int ci = 0; ci < A.n_comp; ++ci) {
      cptr rec = img + L.lw_off + ci * LW_WORDS;
      cptr sel = rec + LW_SEL_INLINE;  // sel[0..3]: masks, sel[4..7]: selected bits in lower words (inside the record)
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
      // colex rank = sum over the set bits, in ascending order, of C(position inside f_sel, ordinal + 1): each lane
      // walks ITS OWN set bits word by word; the binomials come from a 2 KB LDS table (no per-lane multiplications)
      // index of the first pattern of weight cnt: a select chain over the record's 8 base words (scalars) - a
      // gather from the image here would be one more memory latency in front of the threshold reads
      // (readfirstlane pins each word as a scalar: otherwise the selects are folded into ONE select of addresses
      // and a gather again)
      uint32_t pat = rec[LW_BASES_INLINE];
#pragma unroll
      for (uint32_t w = 1; w <= TSIMK_LW_MAX_WEIGHT; ++w) {
        const uint32_t bw = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec[LW_BASES_INLINE + w]);
        pat = (cnt == w) ? bw : pat;
      }
      uint32_t ord = 0;
      lw_rank_word(m0, sel[0], 0u, binom_lds, ord, pat);
      lw_rank_word(m1, sel[1], sel[5], binom_lds, ord, pat);
      if constexpr (WF32 == 4) {
        lw_rank_word(m2, sel[2], sel[6], binom_lds, ord, pat);
        lw_rank_word(m3, sel[3], sel[7], binom_lds, ord, pat);
      }
      const uint32_t n_out = rec[LW_NOUT];
      cptr outpos = img + rec[LW_OUTPOS];
      const uint32_t keybase = rec[LW_KEYBASE];
      const uint32_t *thr = L.tab + rec[LW_TAB] + ((size_t)pat << n_out);
#ifdef TSIMK_LW_TRACE
      asm volatile("" :: "v"(pat));
#endif
      LW_STAMP(3);
      lw_walk(A, thr, n_out, keybase, shot, [&](uint32_t i, bool bit) {
        const uint32_t dst = outpos[i];
        const uint32_t v = (bit ? 1u : 0u) << (dst & 31u);
        if ((dst >> 5) == 0u) o0 |= v;
        else o1 |= v;
      }, tr_it);
    }
    if (!hard) {
      if (A.out) A.out[row] = (uint64_t)o0 | ((uint64_t)o1 << 32);  // WO == 1
      if (A.out_compact) {  // at most 8 bytes here (WO == 1)
        uint8_t *dst = A.out_compact + row * A.out_rb;
        for (int k = 0; k < A.out_rb; ++k) dst[k] = (uint8_t)(((k < 4) ? o0 : o1) >> (8 * (k & 3)));
      }
    }
  }
  LW_STAMP(8);

  // wave-aggregated append of the hard rows
  const unsigned long long hm = __ballot(hard ? 1 : 0);
  if (hm != 0ull) {
    const int lane = (int)(threadIdx.x & 63u);
    const int leader = __builtin_ctzll(hm);
    uint32_t basei = 0;
    const uint32_t k = blockIdx.x % (uint32_t)L.n_lists;  // this block's sub-list
    if (lane == leader) basei = atomicAdd(&L.ctl[32u * k], (uint32_t)__popcll(hm));
    basei = (uint32_t)__shfl((int)basei, leader, 64);
    if (hard)
      L.hard_index[(size_t)k * L.list_cap + basei + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull))] = (uint32_t)row;
  }
  LW_STAMP(9);
  }  // rows of this block
  LW_STAMP_ABS(26);
}

}  // namespace tsimk
