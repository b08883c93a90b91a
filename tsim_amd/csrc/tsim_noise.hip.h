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
  const float *inv_log2_1mp; // [n_ch] 1 / log2(1 - p_fire) (negative; 0 -> channel always fires): k_noise_tile
  int tile, n_tiles;         // k_noise_tile: shots per block (seg divides it)
  const double *log1m_p;     // [n_ch] log(1 - p_fire)  (0 -> channel always fires)
  const uint32_t *cdf_off;   // [n_ch + 1] offsets into cdf / pattern tables
  const uint32_t *cdf;       // conditional CDF over the non-identity outcomes as 32-bit thresholds: ceil(cdf * 2^32), the last clamped
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
    uint32_t o = c0;
    while (o + 1 < c1 && A.cdf[o] <= x1) ++o;  // searchsorted(cdf, u), u = x1 / 2^32 (channels.py:641-656): outcomes resolved to 2^-32
    const uint64_t *pat = A.patterns + (size_t)o * A.WF;
    for (int w = 0; w < A.WF; ++w) {
      const uint64_t v = pat[w];
      if (v) atomicXor(&A.f[pos * A.WF + w], (unsigned long long)v);
    }
  }
}

// The same sampler, one block per TILE of shots (round 3).  k_noise applies every fire with a device-scope 64-bit
// atomic on the row in HBM behind a full memset: 1.3 atomics per shot at the benchmark's noise level are what it
// waits for (62 us per 10^6 shots; its arithmetic is a few microseconds).  Here a block owns the rows
// [tile * TILE, (tile + 1) * TILE): they live in LDS - zeroed there, patterns applied with LDS atomics by the
// (channel, sub-segment) pairs of the tile, then written ONCE, coalesced; no memset, no global atomic.  The gap of the
// geometric skip is float32: floor(log2(u) / log2(1 - p)) + 1 with u a 24-bit uniform in (0, 1] and v_log_f32 (1 ulp)
// - P(gap >= k) is reproduced to a relative ~1e-6, gaps beyond 2^24 / p shots are cut off (channels with
// p < 1e-6 per shot lose ~1e-7 of their fires); single-outcome channels use both Threefry words of a block for gaps.
__global__ void __launch_bounds__(256) k_noise_tile(NoiseArgs A) {
  extern __shared__ unsigned long long noise_rows[];  // [TILE][WF]
  const long long t_lo = (long long)blockIdx.x * A.tile;
  const int rows = (int)min((long long)A.tile, A.B - t_lo);
  for (int i = threadIdx.x; i < rows * A.WF; i += blockDim.x) noise_rows[i] = 0ull;
  __syncthreads();
  const int nsub = A.tile / A.seg;
  const int pairs = A.n_ch * nsub;
  for (int pr = threadIdx.x; pr < pairs; pr += blockDim.x) {
    const int ch = pr / nsub, sub = pr - ch * nsub;
    const int lo = sub * A.seg, hi = min(lo + A.seg, rows);
    if (lo >= rows) continue;
    const float inv = A.inv_log2_1mp[ch];
    const uint32_t c0 = A.cdf_off[ch], c1 = A.cdf_off[ch + 1];
    const bool single = c1 - c0 <= 1u;
    // stream of this (channel, global segment): key folded with the channel, counter = (segment, draw)
    const uint32_t gseg = (uint32_t)((t_lo / A.seg) + sub);
    int pos = lo - 1;
    uint32_t spare = 0u;
    bool have_spare = false;
    for (uint32_t draw = 0;; ++draw) {
      uint32_t w_gap, w_out = 0u;
      if (single && have_spare) {
        w_gap = spare;
        have_spare = false;
      } else {
        uint32_t x0 = gseg, x1 = draw;
        threefry2x32(A.k0 ^ (uint32_t)ch * 0x9E3779B9u, A.k1, x0, x1);
        w_gap = x0;
        if (single) { spare = x1; have_spare = true; }
        else w_out = x1;
      }
      int gap = 1;
      if (inv < 0.0f) {
        const float u = (float)((w_gap >> 8) + 1u) * (1.0f / 16777216.0f);  // (0, 1]
        const float g = floorf(__log2f(u) * inv);
        gap = g >= 1.0e9f ? 1000000000 : (int)g + 1;
      }
      if (gap >= hi - pos) break;  // (no overflow: pos < hi)
      pos += gap;
      uint32_t o = c0;
      if (!single) {
        while (o + 1 < c1 && A.cdf[o] <= w_out) ++o;  // u = w_out / 2^32 against the float64 CDF rounded up to 2^-32
      }
      const uint64_t *pat = A.patterns + (size_t)o * A.WF;
      for (int w = 0; w < A.WF; ++w) {
        const uint64_t v = pat[w];
        if (v) atomicXor(&noise_rows[pos * A.WF + w], (unsigned long long)v);
      }
    }
  }
  __syncthreads();
  unsigned long long *dst = A.f + t_lo * A.WF;
  for (int i = threadIdx.x; i < rows * A.WF; i += blockDim.x) dst[i] = noise_rows[i];
}

}  // namespace tsimk
#include "tsim_noise_wave.hip.h"
namespace tsimk {

__global__ void __launch_bounds__(1024) k_noise_wave(NoiseWaveArgs A) {
  extern __shared__ unsigned long long noise_rows[];  // [TILE][WF], then the channel records
  const long long t_lo = (long long)blockIdx.x * A.tile;
  const int rows = (int)min((long long)A.tile, A.B - t_lo);
  uint32_t *chrec = reinterpret_cast<uint32_t *>(noise_rows + (size_t)A.tile * A.WF);
  // A 245-block grid (10^6 shots in tiles of 4096) is one block per CU: what a block waits for IS the kernel's time.  The
  // channel tables - five dependent global loads per fire in the first version (23 us per 10^6 shots, no faster than
  // k_noise_tile) - are copied to LDS once; a single-outcome channel's whole pattern usually is one (word, mask) record.
  noise_wave_records(A, chrec);
  noise_wave_tile(A, noise_rows, chrec, (uint32_t)blockIdx.x, rows, A.k0, A.k1);  // (its first barrier orders the records too)
  unsigned long long *dst = A.f + t_lo * A.WF;
  for (int i = threadIdx.x; i < rows * A.WF; i += blockDim.x) dst[i] = noise_rows[i];
}

}  // namespace tsimk
