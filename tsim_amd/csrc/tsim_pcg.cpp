// tsim_pcg.cpp - the reference's host-side noise sampler at native speed, on numpy's own random stream.
//
// ChannelSampler.sample (reference src/tsim/noise/channels.py:624-658) draws, per simplified channel,
// `n_draws` geometric gaps and one uniform per fired row from a numpy Generator (PCG64).  That stream IS the
// drop-in contract for a fixed seed - so this file restates the three numpy pieces it consumes, bit for bit:
//   * PCG64 (XSL-RR 128/64, O'Neill 2014): state = state * MULT + inc; output = rotr64(hi ^ lo, hi >> 58);
//     next_double = (next_uint64 >> 11) * 2^-53;
//   * standard_exponential: 256-strip ziggurat (Marsaglia & Tsang 2000) with numpy's table constants
//     (tsim_zig_tables.h, measured from numpy - see scripts/numpy_ziggurat_tables.py);
//   * geometric(p): p >= 1/3 -> search on one uniform; else ceil(-standard_exponential / log1p(-p)).
// numpy is a pinned dependency of the reference (uv.lock: numpy 2.2.6), not part of /root/reference; the
// equivalence is tested draw for draw against the installed numpy (tests/test_pcg_native.py) and re-checked at
// run time before the native engine is trusted (tsim_amd/channels.py).
//
// Host only - no HIP in this translation unit.  Rows come out PACKED (uint64 little-endian bit rows), i.e. in
// the layout the sampling kernels read: generation is one sequential pass over the stream (the RNG is a serial
// dependency chain), the XOR scatter is tiled over shot ranges and threaded.
#include "../../include/tsim_hip.h"
#include "tsim_zig_tables.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

int tsim_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

namespace {

typedef unsigned __int128 u128;

struct Pcg {
  u128 state, inc;
};

const u128 kMult = ((u128)0x2360ED051FC65DA4ull << 64) | 0x4385DF649FCCF645ull;

// One PCG64 step + output.  (Producing the raw outputs in blocks with several interleaved LCG chains was
// measured and dropped: the 128-bit multiply chain already overlaps with the per-draw work of its consumers,
// and the extra buffer traffic made the exponential 40 % slower.)
inline uint64_t next64(Pcg &g) {
  g.state = g.state * kMult + g.inc;
  const uint64_t hi = (uint64_t)(g.state >> 64), lo = (uint64_t)g.state;
  const unsigned rot = (unsigned)(hi >> 58);
  const uint64_t x = hi ^ lo;
  return (x >> rot) | (x << ((64u - rot) & 63u));
}

inline double next_double(Pcg &g) { return (double)(next64(g) >> 11) * (1.0 / 9007199254740992.0); }

// strip data side by side: one cache line serves the common path
struct Strip {
  uint64_t ke;
  double we;
};
struct StripTable {
  Strip s[256];
  StripTable() {
    for (int i = 0; i < 256; ++i) s[i] = Strip{kZigKe[i], kZigWe[i]};
  }
};
const StripTable kStrips;

inline double standard_exponential(Pcg &g) {
  for (;;) {
    uint64_t ri = next64(g) >> 3;
    const unsigned idx = (unsigned)(ri & 0xFFu);
    ri >>= 8;
    const Strip &st = kStrips.s[idx];
    const double x = (double)ri * st.we;
    if (ri < st.ke) return x;  // ~98.9 % of the draws
    if (idx == 0) return kZigExpR - log1p(-next_double(g));  // the tail beyond r
    if ((kZigFe[idx - 1] - kZigFe[idx]) * next_double(g) + kZigFe[idx] < exp(-x)) return x;  // the wedge
  }
}

inline int64_t geometric_search(Pcg &g, double p) {
  int64_t X = 1;
  double sum = p, prod = p;
  const double q = 1.0 - p, U = next_double(g);
  while (U > sum) {
    prod *= q;
    sum += prod;
    ++X;
  }
  return X;
}

inline int64_t geometric_inversion(Pcg &g, double log1m_p) {
  const double z = ceil(-standard_exponential(g) / log1m_p);
  if (z >= 9.223372036854776e+18) return INT64_MAX;
  return (int64_t)z;
}

inline Pcg load(const tsim_pcg64 *s) {
  Pcg g;
  g.state = ((u128)s->state_hi << 64) | s->state_lo;
  g.inc = ((u128)s->inc_hi << 64) | s->inc_lo;
  return g;
}

inline void store(tsim_pcg64 *s, const Pcg &g) {
  s->state_hi = (uint64_t)(g.state >> 64);
  s->state_lo = (uint64_t)g.state;
}

// g jumps `delta` raw outputs ahead in O(log delta) (Brown, "Random number generation with arbitrary strides")
inline void advance(Pcg &g, uint64_t delta) {
  u128 acc_mult = 1, acc_plus = 0, cur_mult = kMult, cur_plus = g.inc;
  while (delta) {
    if (delta & 1u) {
      acc_mult *= cur_mult;
      acc_plus = acc_plus * cur_mult + cur_plus;
    }
    cur_plus = (cur_mult + 1) * cur_plus;
    cur_mult *= cur_mult;
    delta >>= 1;
  }
  g.state = acc_mult * g.state + acc_plus;
}

// scratch that survives between calls (one sampler thread per process is the rule; thread_local keeps others safe)
struct Scratch {
  std::vector<uint32_t> row;      // fired rows, channel after channel
  std::vector<uint16_t> outcome;  // outcome index per fired row (multi-outcome channels only)
};
thread_local Scratch g_scratch;

}  // namespace

extern "C" int tsim_pcg_draw(tsim_pcg64 *rng, int32_t kind, double p, int64_t n, void *out) {
  if (!rng || (n > 0 && !out) || n < 0) return tsim_fail(TSIM_EINVAL, "bad argument");
  Pcg g = load(rng);
  switch (kind) {
    case TSIM_PCG_RAW:
      for (int64_t i = 0; i < n; ++i) ((uint64_t *)out)[i] = next64(g);
      break;
    case TSIM_PCG_DOUBLE:
      for (int64_t i = 0; i < n; ++i) ((double *)out)[i] = next_double(g);
      break;
    case TSIM_PCG_EXPONENTIAL:
      for (int64_t i = 0; i < n; ++i) ((double *)out)[i] = standard_exponential(g);
      break;
    case TSIM_PCG_GEOMETRIC: {
      if (!(p > 0.0) || p > 1.0) return tsim_fail(TSIM_EINVAL, "geometric: p = %g outside (0, 1]", p);
      if (p >= 0.333333333333333333333333) {
        for (int64_t i = 0; i < n; ++i) ((int64_t *)out)[i] = geometric_search(g, p);
      } else {
        const double l = log1p(-p);
        for (int64_t i = 0; i < n; ++i) ((int64_t *)out)[i] = geometric_inversion(g, l);
      }
      break;
    }
    default: return tsim_fail(TSIM_EINVAL, "unknown draw kind %d", kind);
  }
  store(rng, g);
  return TSIM_OK;
}

extern "C" int tsim_pcg_sample_channels(tsim_pcg64 *rng, int32_t n_channels, const double *p_fire,
                                        const int32_t *n_outcomes, const double *cond_cdf, const uint64_t *patterns,
                                        int32_t words, int64_t num_samples, uint64_t *rows, int32_t threads) {
  if (!rng || n_channels < 0 || words < 1 || num_samples < 0) return tsim_fail(TSIM_EINVAL, "bad argument");
  if (num_samples >= (1ll << 32)) return tsim_fail(TSIM_ENOTSUP, "more than 2^32 - 1 rows per call");
  if (num_samples > 0 && !rows) return tsim_fail(TSIM_EINVAL, "rows is NULL");
  if (n_channels > 0 && (!p_fire || !n_outcomes || !cond_cdf || !patterns)) return tsim_fail(TSIM_EINVAL, "NULL channel table");
  if (num_samples == 0) return TSIM_OK;
  static const bool timing = getenv("TSIM_PCG_TIMING") != nullptr;
  const auto t_start = std::chrono::steady_clock::now();
  memset(rows, 0, (size_t)num_samples * words * 8);
  const auto t_zero = std::chrono::steady_clock::now();
  Pcg g = load(rng);
  // ---- pass 1 (sequential: the stream is one dependency chain): per channel the fired rows and their outcomes
  Scratch &S = g_scratch;
  std::vector<size_t> chan_begin((size_t)n_channels + 1, 0), table_off((size_t)n_channels + 1, 0);
  {
    double cap = 0;
    for (int c = 0; c < n_channels; ++c) {
      const double e = (double)num_samples * p_fire[c];
      cap += e + 7.0 * sqrt(e) + 101.0;
      table_off[c + 1] = table_off[c] + (size_t)std::max(0, n_outcomes[c]);
      if (n_outcomes[c] > 65535) return tsim_fail(TSIM_ENOTSUP, "channel %d has more than 65535 outcomes", c);
    }
    if (S.row.size() < (size_t)cap + 16) S.row.resize((size_t)cap + 16);
    if (S.outcome.size() < S.row.size()) S.outcome.resize(S.row.size());
  }
  uint32_t *const frow = S.row.data();
  uint16_t *const fout = S.outcome.data();
  size_t n_fires = 0;
  for (int c = 0; c < n_channels; ++c) {
    const double p = p_fire[c];
    const int nout = n_outcomes[c];
    if (!(p > 0.0) || p > 1.0 || nout < 1) return tsim_fail(TSIM_EINVAL, "channel %d: p_fire = %g, outcomes = %d", c, p, nout);
    const double *cdf = cond_cdf + table_off[c];
    // n_draws = int(expected + 7 sigma) + 100 (channels.py:641-644), in the same double operations
    const double expected = (double)num_samples * p;
    const double sigma = sqrt(expected * (1.0 - p));
    const int64_t n_draws = (int64_t)(expected + 7.0 * sigma) + 100;
    chan_begin[c] = n_fires;
    // positions = cumsum(gaps) - 1, kept while < num_samples.  Every one of the n_draws gaps is drawn (the stream
    // position depends on it); positions only grow, so the store below is unconditional and the count advances
    // only while the position is still inside the batch.
    size_t n = n_fires;
    const size_t room = S.row.size() - 1;
    int64_t pos = -1;
    if (p >= 0.333333333333333333333333) {
      for (int64_t i = 0; i < n_draws; ++i) {
        pos += geometric_search(g, p);
        frow[n] = (uint32_t)pos;
        n += (size_t)((pos < num_samples) & (n < room));
      }
    } else {
      // gap = ceil(-e / log1p(-p)) with the exact division (a multiply by the reciprocal plus a closeness check
      // was measured: slower - the divide is off the dependency chain).  A position that has left the batch stays
      // outside (clamped: numpy's int64 cumsum cannot wrap back below 2^32 rows either).
      const double l = log1p(-p);
      const int64_t far = num_samples + 1;
      for (int64_t i = 0; i < n_draws; ++i) {
        const int64_t gap = geometric_inversion(g, l);
        pos = (gap >= far || pos + gap >= far) ? far : pos + gap;
        frow[n] = (uint32_t)pos;
        n += (size_t)((pos < num_samples) & (n < room));
      }
    }
    // one uniform per fired row; outcome = searchsorted(cdf, u) (first entry >= u; cdf[-1] == 1 > u).  A channel
    // with a single non-identity outcome needs no value - only the stream position: jump over its uniforms.
    if (nout == 1) {
      advance(g, (uint64_t)(n - n_fires));
    } else {
      for (size_t k = n_fires; k < n; ++k) {
        const double u = next_double(g);
        int o = 0;
        while (o < nout - 1 && cdf[o] < u) ++o;
        fout[k] = (uint16_t)o;
      }
    }
    n_fires = n;
  }
  chan_begin[n_channels] = n_fires;
  store(rng, g);
  const auto t_pass1 = std::chrono::steady_clock::now();
  // ---- pass 2: XOR the patterns in, tiled over shot ranges (a tile of rows stays cache resident while every
  //      channel's cursor sweeps it) and threaded over tiles
  const int64_t tile_rows = std::max<int64_t>(1024, (256 * 1024) / (8 * (int64_t)words));
  const int64_t n_tiles = (num_samples + tile_rows - 1) / tile_rows;
  int T = threads > 0 ? threads : (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()));
  T = (int)std::max<int64_t>(1, std::min<int64_t>(T, n_tiles));
  if (n_fires < 16384) T = 1;
  auto work = [&](int t) {
    std::vector<size_t> cur((size_t)n_channels);
    const int64_t tile_lo = n_tiles * t / T, tile_hi = n_tiles * (t + 1) / T;
    const uint32_t first_row = (uint32_t)std::min<int64_t>(num_samples, tile_lo * tile_rows);
    for (int c = 0; c < n_channels; ++c)  // first fire of this thread's range in every channel
      cur[c] = (size_t)(std::lower_bound(frow + chan_begin[c], frow + chan_begin[c + 1], first_row) - frow);
    for (int64_t tile = tile_lo; tile < tile_hi; ++tile) {
      const uint32_t end_row = (uint32_t)std::min<int64_t>(num_samples, (tile + 1) * tile_rows);
      for (int c = 0; c < n_channels; ++c) {
        size_t k = cur[c];
        const size_t stop = chan_begin[c + 1];
        const uint64_t *pat = patterns + table_off[c] * (size_t)words;
        if (words == 1 && n_outcomes[c] == 1) {
          const uint64_t v = pat[0];
          for (; k < stop && frow[k] < end_row; ++k) rows[frow[k]] ^= v;
        } else if (words == 1) {
          for (; k < stop && frow[k] < end_row; ++k) rows[frow[k]] ^= pat[fout[k]];
        } else {
          for (; k < stop && frow[k] < end_row; ++k) {
            uint64_t *dst = rows + (size_t)frow[k] * words;
            const uint64_t *src = pat + (n_outcomes[c] == 1 ? 0 : (size_t)fout[k] * words);
            for (int w = 0; w < words; ++w) dst[w] ^= src[w];
          }
        }
        cur[c] = k;
      }
    }
  };
  if (T <= 1) {
    work(0);
  } else {
    std::vector<std::thread> pool;
    for (int t = 1; t < T; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto &th : pool) th.join();
  }
  const size_t fires_total = n_fires;
  if (timing)
    fprintf(stderr, "tsim_pcg_sample_channels: %lld rows, %zu fires: zero %.2f ms, draws %.2f ms, scatter (%d threads) %.2f ms\n",
            (long long)num_samples, fires_total, std::chrono::duration<double, std::milli>(t_zero - t_start).count(),
            std::chrono::duration<double, std::milli>(t_pass1 - t_zero).count(), T,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_pass1).count());
  return TSIM_OK;
}
