// tsim_noise.hip.h - device-side noise sampling: error channels -> packed f rows (gfx950).
//
// Statistically equivalent replacement of the reference's host-side
// ChannelSampler.sample (src/tsim/noise/channels.py:624-658): the same geometric-skip idea -
// per (simplified) channel the positions of the firing shots are cumulative Geometric(p_fire)
// gaps, the outcome of a fire is drawn from the conditional CDF, and the outcome's XOR pattern is
// applied to the shot's f row - but parallel: one thread owns one (channel, shot-segment) pair
// (the geometric distribution is memoryless, so segments are independent), draws from a
// counter-based Threefry-2x32 stream keyed by (batch key; channel, segment, draw index) and applies
// patterns with 64-bit atomic XORs on the packed rows.  Work is O(number of fires), not O(B * num_f).
// The numpy PCG64 stream of the reference is NOT reproduced (it is inherently sequential); tests
// validate the distribution instead (tests/test_gpu_noise.py).
#pragma once
#include "tsim_kernels.hip.h"

namespace tsimk {

struct NoiseArgs {
  const double *log1m_p;     // [n_ch] log(1 - p_fire)  (0 -> channel always fires)
  const uint32_t *cdf_off;   // [n_ch + 1] offsets into cdf / pattern tables
  const float *cdf;          // conditional CDF over the non-identity outcomes
  const uint64_t *patterns;  // [total outcomes, WF] packed XOR patterns
  unsigned long long *f;     // [B, WF] packed rows, zeroed by the caller
  long long B;
  int n_ch, WF;
  int seg;                   // shots per segment
  long long n_seg;
  uint32_t k0, k1;
};

__global__ void __launch_bounds__(256) k_noise(NoiseArgs A) {
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)A.n_ch * A.n_seg;
  if (tid >= total) return;
  // consecutive threads take consecutive segments of one channel (coalesced-ish atomics)
  const int ch = (int)(tid / A.n_seg);
  const long long sg = tid - (long long)ch * A.n_seg;
  const long long lo = sg * A.seg, hi = min(lo + (long long)A.seg, A.B);
  const double l1p = A.log1m_p[ch];
  const uint32_t c0 = A.cdf_off[ch], c1 = A.cdf_off[ch + 1];
  long long pos = lo - 1;
  for (uint32_t draw = 0;; ++draw) {
    uint32_t x0 = (uint32_t)tid, x1 = draw;
    {
      // fold the high half of the thread index into the key so that > 2^32 threads stay distinct
      threefry2x32(A.k0 ^ (uint32_t)((unsigned long long)tid >> 32), A.k1, x0, x1);
    }
    // gap ~ Geometric(p) on {1, 2, ...}: floor(log(u) / log(1 - p)) + 1, u uniform in (0, 1]
    long long gap = 1;
    if (l1p < 0.0) {
      const double u = ((double)x0 + 1.0) * (1.0 / 4294967296.0);
      const double g = floor(log(u) / l1p);
      gap = (g >= 9.0e18) ? (long long)9.0e18 : (long long)g + 1;
    }
    pos += gap;
    if (pos >= hi || pos < lo) break;
    // outcome: first index with cdf > u2
    const float u2 = (float)(x1 >> 8) * (1.0f / 16777216.0f);
    uint32_t o = c0;
    while (o + 1 < c1 && A.cdf[o] <= u2) ++o;
    const uint64_t *pat = A.patterns + (size_t)o * A.WF;
    for (int w = 0; w < A.WF; ++w) {
      const uint64_t v = pat[w];
      if (v) atomicXor(&A.f[pos * A.WF + w], (unsigned long long)v);
    }
  }
}

}  // namespace tsimk
