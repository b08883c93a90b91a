// tsim_noise_wave.hip.h - the tile code of the device channel sampler (k_noise_wave, tsim_noise.hip.h), shared with the fused
// noise + first pass kernel (tsim_noise_fused.hip.h): argument record and device functions only - no kernel lives here.
#pragma once
#include "tsim_kernels.hip.h"

namespace tsimk {

// ---------------------------------------------------------------------------
// Round 6: the tile sampler with GROUPS OF LANES per (channel, tile) instead of one thread per (channel,
// sub-segment).  k_noise_tile above was the most expensive kernel of the resident pipeline - 22.9 us per 10^6 shots of the
// benchmark's noise model against 15.8 for the sampling pass it feeds - although it runs fewer Threefry blocks per shot (0.9
// against 5): every thread walked its own chain of geometric gaps (a divergent loop of ~6 dependent draws, each behind a
// v_log_f32 and an LDS atomic; the wave waits for its longest lane).  Here the g lanes of a group (8, 16, 32 or 64: about one
// and a half times the fires a channel expects in a tile) draw 2 g gaps AT ONCE - one Threefry block per lane, both words
// are gaps - a segmented prefix sum over the group turns them into positions, and every position inside the tile is a
// fire: no chain, no divergence, one more round only while the last position is still inside the tile.  Outcomes of
// multi-outcome channels come from a second block (both words used, one per gap).  Same statistics as k_noise_tile: gaps
// floor(log2 u / log2(1 - p)) + 1 from a 24-bit uniform, outcomes resolved to 2^-32 against the float64 conditional CDF.
// Stream: key ^ channel, counter = (tile, round * 64 + lane of the group [| 2^31 for the outcome block]).
// ---------------------------------------------------------------------------
struct NoiseWaveArgs {
  const float *inv_log2_1mp;  // [n_ch]
  const uint32_t *cdf_off;    // [n_ch + 1]
  const uint32_t *cdf;
  const uint32_t *pw_off;     // [total outcomes + 1]: the outcome's non-zero pattern words, as (word index, mask lo, mask hi, 0) records
  const uint32_t *pw;
  unsigned long long *f;      // [B, WF]
  long long B;
  int n_ch, WF, tile;
  int g;                      // lanes per (channel, tile) group: 8 .. 64
  uint32_t k0, k1;
};

// The channel records of the tile sampler in LDS (once per block): (inv, c0, c1, first pattern word | entries << 16, mask lo, mask hi)
__device__ __forceinline__ void noise_wave_records(const NoiseWaveArgs &A, uint32_t *chrec) {
  for (int c = threadIdx.x; c < A.n_ch; c += blockDim.x) {
    const uint32_t c0 = A.cdf_off[c], c1 = A.cdf_off[c + 1];
    const uint32_t e0 = A.pw_off[c0], e1 = A.pw_off[c0 + 1u];
    chrec[6 * c] = __float_as_uint(A.inv_log2_1mp[c]);
    chrec[6 * c + 1] = c0;
    chrec[6 * c + 2] = c1;
    chrec[6 * c + 3] = (e1 > e0 ? A.pw[4u * e0] : 0u) | ((e1 - e0) << 16);
    chrec[6 * c + 4] = e1 > e0 ? A.pw[4u * e0 + 1u] : 0u;
    chrec[6 * c + 5] = e1 > e0 ? A.pw[4u * e0 + 2u] : 0u;
  }
}

// One tile: `rows` packed f rows of WF 64-bit words in noise_rows (LDS; zeroed here), tile number `tile_idx` of the batch whose
// noise key is (k0, k1).  The whole block takes part; the rows are complete behind the barrier at the end.  Shared by
// k_noise_wave and the fused noise + first pass (tsim_noise_fused.hip.h): the same key gives the same rows in both.
__device__ __forceinline__ void noise_wave_tile(const NoiseWaveArgs &A, unsigned long long *noise_rows, const uint32_t *chrec, uint32_t tile_idx,
                                                int rows, uint32_t k0, uint32_t k1) {
  for (int i = threadIdx.x; i < rows * A.WF; i += blockDim.x) noise_rows[i] = 0ull;
  __syncthreads();
  const uint32_t g = (uint32_t)A.g, lane = threadIdx.x & 63u;
  const uint32_t gl = threadIdx.x & (g - 1u);            // lane inside its group
  const uint32_t grp = threadIdx.x / g, ngrp = blockDim.x / g;
  const uint32_t lane0 = lane & ~(g - 1u);               // the group's first lane inside the wave
  const uint32_t rounds_ch = ((uint32_t)A.n_ch + ngrp - 1u) / ngrp;
  for (uint32_t it = 0; it < rounds_ch; ++it) {
    const uint32_t ch = it * ngrp + grp;
    const bool live = ch < (uint32_t)A.n_ch;
    const uint32_t chc = live ? ch : 0u;
    const float inv = __uint_as_float(chrec[6u * chc]);
    const uint32_t c0 = chrec[6u * chc + 1u], c1 = chrec[6u * chc + 2u];
    const uint32_t rec3 = chrec[6u * chc + 3u];
    const unsigned long long m_first = (unsigned long long)chrec[6u * chc + 4u] | ((unsigned long long)chrec[6u * chc + 5u] << 32);
    const bool multi = c1 - c0 > 1u;
    const bool simple = !multi && (rec3 >> 16) <= 1u;  // one outcome of at most one pattern word: everything is in the record
    int base = live ? -1 : rows;  // the last position drawn so far
    for (uint32_t round = 0; __builtin_amdgcn_ballot_w64(base < rows - 1) != 0ull; ++round) {
      const bool on = base < rows - 1;
      uint32_t x0 = tile_idx, x1 = round * 64u + gl;
      threefry2x32(k0 ^ chc * 0x9E3779B9u, k1, x0, x1);
      auto gap_of = [&](uint32_t w) -> int {
        if (!(inv < 0.0f)) return 1;  // the channel always fires
        const float u = (float)((w >> 8) + 1u) * (1.0f / 16777216.0f);  // (0, 1]
        const float gf = floorf(__log2f(u) * inv);
        return gf >= 1048576.0f ? 1048576 : (int)gf + 1;  // (beyond any tile: the sums below stay far from 2^31)
      };
      const int ga = gap_of(x0), gb = gap_of(x1);
      // inclusive prefix sum of (ga + gb) over the group's lanes
      int sum = ga + gb;
      for (uint32_t o = 1u; o < g; o <<= 1) {
        const int up = __shfl_up(sum, (int)o, 64);
        if (gl >= o) sum += up;
      }
      const int pos2 = base + sum, pos1 = pos2 - gb;
      uint32_t oa = c0, ob = c0;
      if (__builtin_amdgcn_ballot_w64(on && multi && pos1 < rows) != 0ull) {
        uint32_t y0 = tile_idx, y1 = (round * 64u + gl) | 0x80000000u;
        threefry2x32(k0 ^ chc * 0x9E3779B9u, k1, y0, y1);
        if (multi) {
          while (oa + 1u < c1 && A.cdf[oa] <= y0) ++oa;  // u = y / 2^32 against the float64 CDF rounded up to 2^-32
          while (ob + 1u < c1 && A.cdf[ob] <= y1) ++ob;
        }
      }
      auto apply = [&](int pos, uint32_t o) {
        if (simple) {
          if (rec3 >> 16) atomicXor(&noise_rows[(size_t)pos * A.WF + (rec3 & 0xFFFFu)], m_first);
          return;
        }
        for (uint32_t e = A.pw_off[o]; e < A.pw_off[o + 1u]; ++e) {
          const uint32_t w = A.pw[4u * e];
          const unsigned long long m = (unsigned long long)A.pw[4u * e + 1u] | ((unsigned long long)A.pw[4u * e + 2u] << 32);
          atomicXor(&noise_rows[(size_t)pos * A.WF + w], m);
        }
      };
      if (on && pos1 < rows) apply(pos1, oa);
      if (on && pos2 < rows) apply(pos2, ob);
      const int last = __shfl(pos2, (int)(lane0 + g - 1u), 64);
      base = on ? (last < rows ? last : rows) : base;
    }
  }
  __syncthreads();
}

}  // namespace tsimk
