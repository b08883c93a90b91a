// tsim_kernel_hw.hip.h - the hard rows, one WAVE per row (k_sample_hw).
//
// The rows the first pass cannot finish from its tables (error patterns heavier than the tables go, the
// normalisation-check row) are few - tens per 10^6 shots - and each needs the whole sample_program
// (src/tsim/sampler.py:117-167): every level's sum over the stabiliser terms (compile/evaluate.py:15-59).  One lane
// per shot (k_sample4h: 64 rows per block, every block streaming all chunk tables of all levels through one CU's LDS)
// takes 33-45 us for them however few they are - the tail of every timed region - 200 us for the cultivation shape,
// and runs the order-dependent float32 sum of the approximate branch (evaluate.py:56-59, the branch the real
// distillation circuits take) on one wave of eight.  Here the parallel axis is the one the reference's formula has:
//
//     amp(x) = sum_g term_g(x)          lane = graph g (g = lane, lane + 64, ...), the row's x is wave-uniform
//
// * every lane evaluates ITS graphs from the fast row layout (eval_graph_fast: the per-shot kernels' own code; the
//   parameter words are scalar operands here, the rows per-lane loads that hit L2);
// * fixed-frame levels: the int32 partial sums meet in a butterfly (order-free, exact);
//   other exact levels and the approximate branch: the lanes' terms are added IN GRAPH ORDER on the uniform path
//   (v_readlane), exactly the sequential scans of exact_scalar.py:173-189 / evaluate.py:56-59;
// * |amp|, p1 / prev, the Threefry draw and the chain rule are wave-uniform; the normalisation check
//   (sampler.py:66-72) is a second evaluation with trial bit 0 for the one row that carries it.
// Waves are independent: a batch of hard rows takes the time of ONE row (~10-15 us for the 35-qubit shape) whatever
// their number up to a chip-full, and no LDS at all.  Same values as every other kernel: same tables, same float
// epilogue, same draws (tests/test_gpu_hard_wave.py).
#pragma once
#include "tsim_kernels.hip.h"
#include "tsim_kernel4.hip.h"   // over4_rows: the per-shot workers behind long lists

namespace tsimk {

#define TSIMK_HW_MAX_CTX 8
struct HwMulti {
  int n_ctx, waves_per_list, max_lists;
  int comp_par;                       // 1, or the program's component count: the lists carry component masks (LwMultiArgs.partial) and
                                      // block (slot, c) evaluates component c of its row alone
  int par_words;                      // LDS words per bit array: one parity bit per row of the longest level stream, + spare (8 arrays per block)
  uint32_t slot_cap;                  // the block-per-row blocks serve the first slot_cap slots of every list (0: all); the worker blocks the rest
  uint32_t hw_blocks;                 // blocks of the block-per-row part; the grid's further blocks are per-shot workers (over4_rows)
  int comp4_off;
  uint32_t *feedback;                 // launch-plan feedback of the first context (see sample4h_rows)
  SampleArgs ctx[TSIMK_HW_MAX_CTX];
};

__device__ __forceinline__ int hw_sum_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ uint32_t hw_or_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v |= (uint32_t)__shfl_xor((int)v, o, 64);
  return v;
}

// n <= 32 consecutive bits of the wave's parity array, starting at bit `pos` (lanes ask for different positions)
__device__ __forceinline__ uint32_t hw_bits(const uint32_t *par, uint32_t pos, uint32_t n) {
  const uint32_t w = pos >> 5, sh = pos & 31u;
  const uint32_t lo = par[w], hi = par[w + 1u];  // (the array has a spare word at the end)
  const uint32_t v = __builtin_amdgcn_alignbit(hi, lo, sh);  // (hi:lo) >> sh
  return n >= 32u ? v : (v & ((1u << n) - 1u));
}

// evaluate() of one level for ONE parameter row x (wave-uniform), in two phases that run on DIFFERENT waves:
//   phase 1, lane = ROW: the level's uniform-stride row stream, 64 rows per step, fully coalesced, eight steps' loads
//            in flight; the parities ((popcount(row & x) + const) & 1) of a step are one ballot, kept in an LDS bit array;
//   phase 2, lane = GRAPH (g = lane, lane + 64, ...), leading wave only: the graph's counts and exponent bits are bit
//            fields of that array (GraphBits), its value one table gather (graph_fast_value) - then the sum over the lanes.
// The levels of a component form a chain only through the SAMPLED BIT (sampler.py:74-79): level k + 1 needs b_k in its
// x.  A row's parity for b_k = 0 is its parity for b_k = 1 XOR its own bit at that position, so phase 1 of level k + 1
// does not wait for b_k: the helper waves run it WHILE the leading wave is in phase 2 of level k, for both values
// (two ballots per step instead of one, into two bit arrays), and the leading wave picks the array once it has drawn the
// bit.  The same trick gives the normalisation-check row (sampler.py:66-72) its trial-bit-0 evaluation: four arrays.
// A level then costs the leading wave's phase 2 alone - the row pass, its global-load latency and its barrier are off
// the chain.
struct HwLevelOut {
  float re, im, re0, im0;
};

// variant v of a level's bit array: bit 0 of v = "previous sampled bit is 0", bit 1 = "trial bit is 0"
template <int W>
__device__ __forceinline__ void hw_phase1(const uint32_t *gimg, cptr lvl, const uint32_t (&x)[W], bool hasA, uint32_t flipA, bool hasB,
                                          uint32_t flipB, uint32_t *par, uint32_t pw, uint32_t wv, uint32_t nwv) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t n_rows = lvl[L_HWN];
  const uint32_t *rows = gimg + lvl[L_HWROWS];
  const uint32_t steps = (n_rows + 63u) >> 6;
  const uint32_t aw = flipA >> 5, as = flipA & 31u, bw = flipB >> 5, bs = flipB & 31u;
#pragma unroll 8
  for (uint32_t st = wv; st < steps; st += nwv) {
    const uint32_t r = st * 64u + lane;
    uint32_t pb = 0u, fa = 0u, fb = 0u;
    if (r < n_rows) {
      const uint32_t *q = rows + (size_t)r * (W + 1);
      uint32_t rw[W + 1];
#pragma unroll
      for (int w = 0; w <= W; ++w) rw[w] = q[w];
      uint32_t t = rw[W] & x[W - 1];
#pragma unroll
      for (int w = W - 2; w >= 0; --w) t = and_xor(rw[1 + w], x[w], t);
      pb = ((uint32_t)__builtin_popcount(t) + rw[0]) & 1u;
      uint32_t wa = rw[1], wb = rw[1];
#pragma unroll
      for (int w = 1; w < W; ++w) {
        wa = (aw == (uint32_t)w) ? rw[1 + w] : wa;
        wb = (bw == (uint32_t)w) ? rw[1 + w] : wb;
      }
      fa = (wa >> as) & 1u;
      fb = (wb >> bs) & 1u;
    }
    const unsigned long long m0 = __builtin_amdgcn_ballot_w64(pb != 0u);
    unsigned long long m1 = 0ull, m2 = 0ull, m3 = 0ull;
    if (hasA) m1 = __builtin_amdgcn_ballot_w64((pb ^ fa) != 0u);
    if (hasB) m2 = __builtin_amdgcn_ballot_w64((pb ^ fb) != 0u);
    if (hasA && hasB) m3 = __builtin_amdgcn_ballot_w64((pb ^ fa ^ fb) != 0u);
    if (lane == 0u) {
      par[2u * st] = (uint32_t)m0;
      par[2u * st + 1u] = (uint32_t)(m0 >> 32);
      if (hasA) { par[pw + 2u * st] = (uint32_t)m1; par[pw + 2u * st + 1u] = (uint32_t)(m1 >> 32); }
      if (hasB) { par[2u * pw + 2u * st] = (uint32_t)m2; par[2u * pw + 2u * st + 1u] = (uint32_t)(m2 >> 32); }
      if (hasA && hasB) { par[3u * pw + 2u * st] = (uint32_t)m3; par[3u * pw + 2u * st + 1u] = (uint32_t)(m3 >> 32); }
    }
  }
  if (lane == 0u && wv == (steps % nwv)) {  // the spare word hw_bits may touch (written once, by whichever wave)
    par[2u * steps] = 0u;
    par[pw + 2u * steps] = 0u;
    par[2u * pw + 2u * steps] = 0u;
    par[3u * pw + 2u * steps] = 0u;
  }
}

// a lane's graph record (16 words).  Records do not depend on sampled bits: the leading wave asks for the NEXT level's
// while it works on the current one (one global-load latency less on every level's chain).
typedef uint32_t hw_u32x4 __attribute__((ext_vector_type(4)));
struct HwRec {
  hw_u32x4 r0, r1, r2, r3;
};
__device__ __forceinline__ HwRec hw_load_rec(const uint32_t *gimg, cptr lvl, uint32_t g) {
  HwRec R;
  R.r0 = R.r1 = R.r2 = R.r3 = hw_u32x4{0u, 0u, 0u, 0u};
  if (g < lvl[L_G]) {
    const hw_u32x4 *q4 = reinterpret_cast<const hw_u32x4 *>(gimg + lvl[L_GRAPHS] + (size_t)g * G_WORDS);  // 64-byte aligned records
    R.r0 = q4[0]; R.r1 = q4[1]; R.r2 = q4[2]; R.r3 = q4[3];
  }
  return R;
}

// phase 2 on the calling wave: pa = the bit array for trial bit 1, pa0 (DUAL) = for trial bit 0; `first` = the records of
// graphs 0..63 (hw_load_rec(…, lane), requested earlier)
template <int W, bool DUAL>
__device__ __forceinline__ HwLevelOut hw_phase2(const uint32_t *gimg, cptr img, cptr lvl, const uint32_t *pa, const uint32_t *pa0, const HwRec &first) {
  const uint32_t G = lvl[L_G];
  const bool approx = (lvl[L_FLAGS] & TSIMK_LFLAG_APPROX) != 0;
  const bool fixed = (lvl[L_FLAGS] & TSIMK_LFLAG_FIXED) != 0;
  const uint32_t graphs = lvl[L_GRAPHS];
  const uint32_t lane = threadIdx.x & 63u;
  hw_u32x4 rec0 = first.r0, rec1 = first.r1, rec2 = first.r2, rec3 = first.r3;
  auto load_rec = [&](uint32_t g) {
    const hw_u32x4 *q4 = reinterpret_cast<const hw_u32x4 *>(gimg + graphs + (size_t)g * G_WORDS);
    rec0 = q4[0]; rec1 = q4[1]; rec2 = q4[2]; rec3 = q4[3];
  };
  LevelSum S, S0;
  for (uint32_t g0 = 0; g0 < G; g0 += 64u) {
    const uint32_t g = g0 + lane;
    const bool mine = g < G;
    int a = 0, b = 0, c = 0, d = 0, p = 0, a0 = 0, b0 = 0, c0 = 0, d0 = 0, p0 = 0;
    float tr = 0.0f, ti = 0.0f, tr0 = 0.0f, ti0 = 0.0f;
    if (mine && g0 > 0u) load_rec(g);
    if (mine) {
      const uint32_t n01 = rec0.x, n3h = rec0.y, flags = rec0.z, nD = rec0.w;
      const uint32_t start = rec2.x;  // GF_HWROW = 8
      static_assert(GF_N01 == 0 && GF_N3H == 1 && GF_FLAGS == 2 && GF_ND == 3 && GF_TBL == 5 && GF_N1 == 6 && GF_TBL2 == 7 && GF_HWROW == 8 &&
                        GF_APRE == 11 && GF_APIM == 12 && G_WORDS == 16, "record words are taken from the four 16-byte loads by position");
      GraphRec R;
      R.flags = flags; R.nD = nD; R.n1 = rec1.z; R.tbl = rec1.y; R.tbl2 = rec1.w;  // GF_N1 = 6, GF_TBL = 5, GF_TBL2 = 7
      const uint32_t apre = rec2.w, apim = rec3.x;                                  // GF_APRE = 11, GF_APIM = 12
      auto bits_of = [&](const uint32_t *pq) {
        GraphBits q;
        uint32_t pos = start;
        // (usually at most 30 rows each; more when the terms are deltas - level_fast_eligible: components of many deterministic outputs)
        const uint32_t n0 = n01 & 0xFFFFu, n1 = n01 >> 16, n3 = n3h & 0xFFFFu;
        auto count = [&](uint32_t n) -> uint32_t {
          uint32_t m = (uint32_t)__builtin_popcount(hw_bits(pq, pos, n));
          for (uint32_t t0 = 32u; t0 < n; t0 += 32u) m += (uint32_t)__builtin_popcount(hw_bits(pq, pos + t0, n - t0));
          pos += n;
          return m;
        };
        q.m0 = count(n0);
        q.m1 = count(n1);
        q.m3 = count(n3);
        q.dbits = 0u;
        for (uint32_t t0 = 0; t0 < nD; t0 += 16u) {  // (pa, pb) per term, first term most significant
          const uint32_t nt = min(16u, nD - t0);
          const uint32_t bits = hw_bits(pq, pos, 2u * nt);
          for (uint32_t t = 0; t < nt; ++t) q.dbits = (q.dbits << 2) | ((bits >> (2u * t)) & 3u);
          pos += 2u * nt;
        }
        q.lam = 0u;
        q.e = 0u;
        if (flags & TSIMK_GFLAG_LAM) { q.lam = hw_bits(pq, pos, 1u); pos += 1u; }
        if (flags & TSIMK_GFLAG_LIN) { q.e = hw_bits(pq, pos, 1u); pos += 1u; }
        const uint32_t nH = n3h >> 16;
        for (uint32_t t0 = 0; t0 < nH; t0 += 16u) {  // XOR_s <u_s,x><v_s,x>: u at even, v at odd positions
          const uint32_t nt = min(16u, nH - t0);
          const uint32_t bits = hw_bits(pq, pos, 2u * nt);
          q.e ^= (uint32_t)__builtin_popcount(bits & (bits >> 1) & 0x55555555u);
          pos += 2u * nt;
        }
        return q;
      };
      const GraphBits q1 = bits_of(pa);
      graph_fast_value(gimg, R, fixed, q1, a, b, c, d, p);
      if (approx) level_term_approx(apre, apim, a, b, c, d, p, tr, ti);
      if constexpr (DUAL) {
        const GraphBits q0 = bits_of(pa0);
        graph_fast_value(gimg, R, fixed, q0, a0, b0, c0, d0, p0);
        if (approx) level_term_approx(apre, apim, a0, b0, c0, d0, p0, tr0, ti0);
      }
    }
    if (fixed) {
      S.sa += a; S.sb += b; S.sc += c; S.sd += d;  // lanes without a graph add zeros
      if constexpr (DUAL) { S0.sa += a0; S0.sb += b0; S0.sc += c0; S0.sd += d0; }
    } else {
      // in graph order on the uniform path: lane l of this round holds graph g0 + l
      const uint32_t n = min(64u, G - g0);
      for (uint32_t l = 0; l < n; ++l) {
        if (approx) {
          S.fre = __fadd_rn(S.fre, __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(tr), (int)l)));
          S.fim = __fadd_rn(S.fim, __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(ti), (int)l)));
          if constexpr (DUAL) {
            S0.fre = __fadd_rn(S0.fre, __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(tr0), (int)l)));
            S0.fim = __fadd_rn(S0.fim, __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(ti0), (int)l)));
          }
        } else {
          level_sum_exact(S, __builtin_amdgcn_readlane(a, (int)l), __builtin_amdgcn_readlane(b, (int)l), __builtin_amdgcn_readlane(c, (int)l),
                          __builtin_amdgcn_readlane(d, (int)l), __builtin_amdgcn_readlane(p, (int)l));
          if constexpr (DUAL)
            level_sum_exact(S0, __builtin_amdgcn_readlane(a0, (int)l), __builtin_amdgcn_readlane(b0, (int)l), __builtin_amdgcn_readlane(c0, (int)l),
                            __builtin_amdgcn_readlane(d0, (int)l), __builtin_amdgcn_readlane(p0, (int)l));
        }
      }
    }
  }
  if (fixed) {
    S.sa = hw_sum_i32(S.sa); S.sb = hw_sum_i32(S.sb); S.sc = hw_sum_i32(S.sc); S.sd = hw_sum_i32(S.sd);
    if constexpr (DUAL) { S0.sa = hw_sum_i32(S0.sa); S0.sb = hw_sum_i32(S0.sb); S0.sc = hw_sum_i32(S0.sc); S0.sd = hw_sum_i32(S0.sd); }
  }
  HwLevelOut o;
  o.re = o.im = o.re0 = o.im0 = 0.0f;
  level_finish(S, lvl, approx, fixed, o.re, o.im, nullptr);
  if constexpr (DUAL) level_finish(S0, lvl, approx, fixed, o.re0, o.im0, nullptr);
  return o;
}

// one component of one row: _sample_component (sampler.py:28-81), W = the component's own parameter-row width.
// LDS: two buffers (levels alternate) of four bit arrays of `pw` words each, then the word that carries the sampled bit.
template <int W, class FBit>
__device__ __forceinline__ void hw_component(const SampleArgs &A, cptr img, cptr comp, int ci, FBit fbit, unsigned long long shot, bool check,
                                             uint32_t *par, uint32_t pw, uint32_t (&out_w)[4]) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
  const bool lead = wv == 0u;
  const uint32_t n_out = comp[C_NOUT], F = comp[C_F];
  cptr fsel = img + comp[C_FSEL];
  cptr levels = img + comp[C_LEVELS];
  cptr outpos = img + comp[C_OUTPOS];
  const uint32_t keybase = comp[C_KEYBASE];
  uint32_t *bit_word = par + 8u * pw;  // the sampled bit, leading wave -> the others
  // K1: the component's f bits (sampler.py:48): lane j fetches bit j, the ballot is the packed word pair
  uint32_t x[W];
#pragma unroll
  for (int w2 = 0; w2 < W; w2 += 2) {
    const uint32_t j = 32u * (uint32_t)w2 + lane;
    const uint32_t raw = fbit(fsel[j < F ? j : 0u]);
    const unsigned long long m = __builtin_amdgcn_ballot_w64(j < F && raw != 0u);
    x[w2] = (uint32_t)m;
    if (w2 + 1 < W) x[w2 + 1] = (uint32_t)(m >> 32);
  }
  // the component's draws (sampler.py:74-75), all at once: lane i computes output i's uniform
  uint32_t u_bits = 0u;
  if (lead && n_out > 0u) {  // (a component without outputs has no subkeys at all)
    const uint32_t o = keybase + (lane < n_out ? lane : 0u);
    uint32_t x0 = (uint32_t)(shot >> 32), x1 = (uint32_t)shot;
    threefry2x32(subkey(A, o, 0), subkey(A, o, 1), x0, x1);
    u_bits = __float_as_uint(__uint_as_float(((x0 ^ x1) >> 9) | 0x3F800000u) - 1.0f);
  }
  auto set_bit = [&](uint32_t pos, bool on) {
    const uint32_t wi = pos >> 5, bm = 1u << (pos & 31u);
#pragma unroll
    for (int w = 0; w < W; ++w)
      if ((uint32_t)w == wi) x[w] = on ? (x[w] | bm) : (x[w] & ~bm);
  };
  // ---- level 0 (normalisation, sampler.py:54): its row pass by every wave (the leading wave's records are on their way)
  HwRec cur = lead ? hw_load_rec(A.img, levels, lane) : HwRec{};
  hw_phase1<W>(A.img, levels, x, false, 0u, false, 0u, par, pw, wv, nwv);
  __syncthreads();
  // ---- level 0's phase 2 (leading wave) beside level 1's row pass (the others): x = f | trial bit 1
  float prev = 0.0f, maxdev = 0.0f;
  if (n_out > 0u) set_bit(F, true);
  if (lead) {
    const HwRec nxt = n_out > 0u ? hw_load_rec(A.img, levels + L_WORDS, lane) : HwRec{};
    const HwLevelOut n0 = hw_phase2<W, false>(A.img, img, levels, par, par, cur);
    prev = cabs32(n0.re, n0.im);
    cur = nxt;
  } else if (n_out > 0u) {
    hw_phase1<W>(A.img, levels + L_WORDS, x, false, 0u, check, F, par + 4u * pw, pw, wv - 1u, nwv - 1u);
  }
  __syncthreads();
  uint32_t prev_bit = 1u;  // (level 1 has no previous bit: its arrays are the "bit = 1" ones)
  for (uint32_t i = 0; i < n_out; ++i) {
    // x holds f | b_0 .. b_(i-1) | trial bit i = 1; level i + 1's arrays are in buffer (i + 1) & 1, variant by b_(i-1)
    cptr lvl = levels + (i + 1) * L_WORDS;
    const uint32_t bitpos = F + i;
    uint32_t *buf = par + (((i + 1u) & 1u) ? 4u * pw : 0u);
    if (lead) {
      const uint32_t *pa = buf + (prev_bit ? 0u : pw);
      const HwRec nxt = i + 1u < n_out ? hw_load_rec(A.img, lvl + L_WORDS, lane) : HwRec{};
      float p1;
      if (check) {  // the check row also evaluates trial bit = 0 (sampler.py:66-72), from the same row pass
        const HwLevelOut o = hw_phase2<W, true>(A.img, img, lvl, pa, pa + 2u * pw, cur);
        p1 = cabs32(o.re, o.im);
        const float p0 = cabs32(o.re0, o.im0);
        const float norm = __fdiv_rn(__fadd_rn(p0, p1), prev);
        maxdev = nanmax(maxdev, fabsf(__fsub_rn(norm, 1.0f)));
      } else {
        const HwLevelOut o = hw_phase2<W, false>(A.img, img, lvl, pa, pa, cur);
        p1 = cabs32(o.re, o.im);
      }
      // sampler.py:74-79
      const float u = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)u_bits, (int)i));
      const bool bit = u < __fdiv_rn(p1, prev);
      prev = bit ? p1 : __fsub_rn(prev, p1);
      if (lane == 0u) *bit_word = bit ? 1u : 0u;
      cur = nxt;
    } else if (i + 1u < n_out) {
      // the next level's row pass, for both values of the bit being drawn: x = f | b_0..b_(i-1) | b_i = 1 | trial bit i + 1
      uint32_t xn[W];
#pragma unroll
      for (int w = 0; w < W; ++w) xn[w] = x[w];
      const uint32_t np = bitpos + 1u, wi = np >> 5, bm = 1u << (np & 31u);
#pragma unroll
      for (int w = 0; w < W; ++w)
        if ((uint32_t)w == wi) xn[w] |= bm;
      hw_phase1<W>(A.img, lvl + L_WORDS, xn, true, bitpos, check, np, par + (((i + 2u) & 1u) ? 4u * pw : 0u), pw, wv - 1u, nwv - 1u);
    }
    __syncthreads();  // the bit is drawn, the next level's arrays are complete (and this level's are free again)
    const bool bit = *bit_word != 0u;  // (rewritten only behind the next barrier)
    prev_bit = bit ? 1u : 0u;
    set_bit(bitpos, bit);
    if (i + 1u < n_out) set_bit(bitpos + 1u, true);  // trial bit of the next level
    const uint32_t dst = outpos[i];  // K15: the final column (sampler.py:164-166)
    if (lane == 0u) {
#pragma unroll
      for (int w = 0; w < 4; ++w)
        if ((dst >> 5) == (uint32_t)w) out_w[w] |= (bit ? 1u : 0u) << (dst & 31u);
    }
  }
  if (check && A.norm_dev && threadIdx.x == 0u) A.norm_dev[ci] = maxdev;
  __syncthreads();  // the next component's level 0 overwrites buffer 0
}

// (The lookahead form of round 5 - levels in groups of three, every node of a group evaluated at once - was measured slower beside a
// first pass and is gone: profiles/r05/hard_tree.txt, HISTORY.md.)

// sample_program (sampler.py:117-167) for one row on one wave
// only_comp >= 0: that component alone, its bits ORed into the row the first pass stored (direct outputs and the other
// components' bits are there already)
template <int WMAX>
__device__ __forceinline__ void hw_row(const SampleArgs &A, long long row, bool check, uint32_t *par, uint32_t pw, int only_comp) {
  cptr img = (cptr)(uintptr_t)A.img;
  const uint32_t lane = threadIdx.x & 63u;
  const unsigned long long shot = (unsigned long long)(A.shot_offset + row);
  const int WF32 = 2 * A.WF;
  // the packed f row: lane w holds word w (WF32 <= 64)
  const uint32_t *frow = reinterpret_cast<const uint32_t *>(A.f + row * A.WF);
  const uint32_t fw = (int)lane < WF32 ? frow[lane] : 0u;
  auto fbit = [&](uint32_t src) -> uint32_t {  // bit `src` of the row, src differs per lane: the owning lane's word
    const uint32_t wv = (uint32_t)__shfl((int)fw, (int)(src >> 5), 64);
    return (wv >> (src & 31u)) & 1u;
  };
  // K14: direct outputs f[idx] ^ flip (sampler.py:140-145): lane j moves direct output j, j + 64, ...
  uint32_t out_w[4] = {0u, 0u, 0u, 0u};  // output words (num_outputs <= 128 here)
  if (only_comp < 0) {
    cptr dt = img + A.direct_off;
    for (int j0 = 0; j0 < A.n_direct; j0 += 64) {
      const int j = j0 + (int)lane;
      const bool mine = j < A.n_direct;
      const int jc = mine ? j : 0;
      const uint32_t s = dt[2 * jc], dst = mine ? dt[2 * jc + 1] : 0u;
      const uint32_t raw = fbit(s & 0x7FFFFFFFu);  // every lane takes part in the exchange (the owners of the words must)
      const uint32_t bit = mine ? (raw ^ (s >> 31)) : 0u;
#pragma unroll
      for (int w = 0; w < 4; ++w)
        if ((dst >> 5) == (uint32_t)w) out_w[w] |= bit << (dst & 31u);
    }
  }
  for (int ci = only_comp < 0 ? 0 : only_comp; ci < (only_comp < 0 ? A.n_comp : only_comp + 1); ++ci) {
    cptr comp = img + A.comp_off + ci * C_WORDS;
    switch (comp[C_W]) {  // rows are packed with the component's own word count
      case 1:
        hw_component<1>(A, img, comp, ci, fbit, shot, check, par, pw, out_w);
        break;
      case 2:
        if constexpr (WMAX >= 2) {
          hw_component<2>(A, img, comp, ci, fbit, shot, check, par, pw, out_w);
        }
        break;
      case 3:
      case 4:  // (parameter rows of 65..128 bits: the overflow / check rows of the sparse-column path, tsim_sample.hip: launch_sample)
        if constexpr (WMAX >= 4) {
          if (comp[C_W] == 3u) hw_component<3>(A, img, comp, ci, fbit, shot, check, par, pw, out_w);
          else hw_component<4>(A, img, comp, ci, fbit, shot, check, par, pw, out_w);
        }
        break;
      case 6:
      case 8:  // (129..256 bits: the check / overflow rows of wide components with many graphs, class F140)
        if constexpr (WMAX >= 8) {
          if (comp[C_W] == 6u) hw_component<6>(A, img, comp, ci, fbit, shot, check, par, pw, out_w);
          else hw_component<8>(A, img, comp, ci, fbit, shot, check, par, pw, out_w);
        }
        break;
      default: __builtin_trap();  // the host launches this kernel for programs of at most 256 parameters only
    }
  }
#pragma unroll
  for (int w = 0; w < 4; ++w) out_w[w] = hw_or_u32(out_w[w]);
  if (threadIdx.x == 0u && only_comp >= 0) {
    // (every word this row's bytes touch: the neighbours' bytes get zeros ORed in; all stores of the first pass are complete)
    if (A.out) {
      uint32_t *orow = reinterpret_cast<uint32_t *>(A.out + row * A.WO);
#pragma unroll
      for (int w = 0; w < 4; ++w)
        if (w < 2 * A.WO && out_w[w]) atomicOr(orow + w, out_w[w]);
    }
    if (A.out_compact) {
      // the row as a little-endian byte string of out_rb <= 16 bytes, shifted to its place in the aligned words
      const unsigned long long at = (unsigned long long)row * (unsigned long long)A.out_rb;
      uint32_t *base = reinterpret_cast<uint32_t *>(A.out_compact + (at & ~3ull));
      const uint32_t sh = 8u * (uint32_t)(at & 3ull);
#pragma unroll
      for (int w = 0; w < 5; ++w) {
        const uint32_t cur = w < 4 ? out_w[w < 4 ? w : 0] : 0u, below = w > 0 ? out_w[w > 0 ? w - 1 : 0] : 0u;
        const uint32_t v = sh ? ((cur << sh) | (below >> (32u - sh))) : cur;
        if (v) atomicOr(base + w, v);
      }
    }
  } else if (threadIdx.x == 0u) {
    if (A.out) {
      uint64_t *orow = A.out + row * A.WO;
      for (int w = 0; w < A.WO; ++w) orow[w] = (uint64_t)out_w[2 * w] | ((uint64_t)out_w[2 * w + 1] << 32);
    }
    if (A.out_compact) {
      uint8_t *dst = A.out_compact + row * A.out_rb;
      for (int k = 0; k < A.out_rb; ++k) dst[k] = (uint8_t)(out_w[k >> 2] >> (8 * (k & 3)));
    }
  }
}

// Every BLOCK (4 waves on one row at a time) serves ONE list of ONE launch: block j of a list takes its slots j,
// j + waves_per_list, ...
// NCH: chunks per tile of the program's chunk tables (the workers' kernel), 0: a grid without workers
template <int W, int NCH>
__global__ void __launch_bounds__(256) k_sample_hw(HwMulti M) {
  if constexpr (NCH > 0) {
    if (blockIdx.x >= M.hw_blocks) {  // block-uniform
      over4_rows<4, NCH>(M.ctx, M.n_ctx, M.comp4_off, M.slot_cap, M.comp_par > 1, blockIdx.x - M.hw_blocks, gridDim.x - M.hw_blocks);
      return;
    }
  }
  __builtin_amdgcn_s_setprio(3);  // a few hundred latency-bound waves beside a chip-full of issue-bound ones: issue when ready
  const uint32_t cp = (uint32_t)M.comp_par;
  const uint32_t wv = blockIdx.x / cp;
  const int only_comp = cp > 1u ? (int)(blockIdx.x - wv * cp) : -1;
  const uint32_t per_ctx = (uint32_t)(M.max_lists * M.waves_per_list);
  const uint32_t c = wv / per_ctx;
  if (c >= (uint32_t)M.n_ctx) return;
  const uint32_t r = wv - c * per_ctx;
  const SampleArgs &A = M.ctx[c];
  const uint32_t k = r / (uint32_t)M.waves_per_list, j = r - k * (uint32_t)M.waves_per_list;
  const uint32_t lane = threadIdx.x & 63u;
  // feedback to the host (mapped pinned memory, read at later launches to choose the launch plan): total and longest
  // hard-row list of the first launch of this batch
  if (M.feedback && blockIdx.x == 0u && threadIdx.x < 64u) {
    uint32_t cn = (int)lane < A.row_lists ? A.row_count[32u * lane] : 0u, mx = cn;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      cn += (uint32_t)__shfl_xor((int)cn, o, 64);
      mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
    }
    if (lane == 0u) {
      M.feedback[0] = cn;
      M.feedback[1] = mx;
      M.feedback[2] = (uint32_t)min(A.B, 0xFFFFFFFFll);
    }
  }
  if ((int)k >= A.row_lists) return;
  uint32_t n = A.row_count[32u * k];
  if (M.slot_cap) n = min(n, M.slot_cap);
  const uint32_t check_row = (A.no_check || !A.check_row) ? 0xFFFFFFFFu : *A.check_row;
  for (uint32_t slot = j; slot < n; slot += (uint32_t)M.waves_per_list) {  // block-uniform
    const uint32_t entry = A.row_index[(size_t)k * A.row_list_cap + slot];
    if (only_comp >= 0) {
      if (!((entry >> (28 + only_comp)) & 1u)) continue;  // block-uniform: this component's bits came from the tables
      const uint32_t row = entry & 0x0FFFFFFFu;
      hw_row<W>(A, (long long)row, row == check_row, tsimk_lds, (uint32_t)M.par_words, only_comp);
    } else {
      hw_row<W>(A, (long long)entry, entry == check_row, tsimk_lds, (uint32_t)M.par_words, -1);
    }
  }
}

}  // namespace tsimk
