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
#include <memory>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <type_traits>
#include <vector>

#include <immintrin.h>

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

// ---------------------------------------------------------------------------
// Blocked stream (AVX-512 machines).  The generator is an LCG: s_{n+4} = s_n * M^4 + c (M^3 + M^2 + M + 1), so four
// interleaved chains produce the SAME outputs in the same order with four multiplies in flight instead of one
// dependent multiply per output.  Outputs go to a small buffer; the geometric draws below consume it eight at a
// time.  The stream position is a plain count: the state handed back to the caller is the initial state advanced
// by the number of outputs consumed (Brown's jump), skipping outputs (single-outcome channels) moves the count.
// ---------------------------------------------------------------------------
struct Stream {
  static constexpr int kBlock = 4096;
  Pcg start;                 // state before output 0 of the call
  u128 m4, c4;               // the four-step LCG
  u128 chain[4];             // states whose outputs are next in line: buf[len + j] = out(chain[j])
  uint64_t consumed = 0;     // outputs consumed before buf[pos]... i.e. stream position of buf[0] is `origin`
  uint64_t origin = 0;       // stream index of buf[0]
  int pos = 0, len = 0;
  alignas(64) uint64_t buf[kBlock + 16];

  static inline uint64_t out(u128 st) {
    const uint64_t hi = (uint64_t)(st >> 64), lo = (uint64_t)st;
    const unsigned rot = (unsigned)(hi >> 58);
    const uint64_t x = hi ^ lo;
    return (x >> rot) | (x << ((64u - rot) & 63u));
  }
  void seed_chains(u128 st) {  // st = state BEFORE the next output
    for (int j = 0; j < 4; ++j) {
      st = st * kMult + start.inc;
      chain[j] = st;
    }
  }
  explicit Stream(const Pcg &g) : start(g) {
    const u128 m2 = kMult * kMult;
    m4 = m2 * m2;
    c4 = start.inc * (kMult * m2 + m2 + kMult + 1);
    seed_chains(g.state);
  }
  // make at least `need` (<= 16) outputs available at buf[pos..]
  inline void ensure(int need) {
    if (len - pos >= need) return;
    const int left = len - pos;
    for (int k = 0; k < left; ++k) buf[k] = buf[pos + k];
    origin += (uint64_t)pos;
    pos = 0;
    len = left;
    u128 s0 = chain[0], s1 = chain[1], s2 = chain[2], s3 = chain[3];
    uint64_t *dst = buf + len;
    const int n4 = (kBlock - len) / 4;
    for (int t = 0; t < n4; ++t) {
      dst[0] = out(s0); dst[1] = out(s1); dst[2] = out(s2); dst[3] = out(s3);
      s0 = s0 * m4 + c4; s1 = s1 * m4 + c4; s2 = s2 * m4 + c4; s3 = s3 * m4 + c4;
      dst += 4;
    }
    chain[0] = s0; chain[1] = s1; chain[2] = s2; chain[3] = s3;
    len += 4 * n4;
  }
  inline uint64_t next64() {
    ensure(1);
    return buf[pos++];
  }
  inline double next_double() { return (double)(next64() >> 11) * (1.0 / 9007199254740992.0); }
  uint64_t position() const { return origin + (uint64_t)pos; }
  void skip(uint64_t n) {
    if (n <= (uint64_t)(len - pos)) {
      pos += (int)n;
      return;
    }
    const uint64_t target = position() + n;
    Pcg g = start;
    advance(g, target);
    seed_chains(g.state);
    origin = target;
    pos = len = 0;
  }
  Pcg finish() const {
    Pcg g = start;
    advance(g, position());
    return g;
  }
};

inline double standard_exponential(Stream &g) {
  for (;;) {
    uint64_t ri = g.next64() >> 3;
    const unsigned idx = (unsigned)(ri & 0xFFu);
    ri >>= 8;
    const Strip &st = kStrips.s[idx];
    const double x = (double)ri * st.we;
    if (ri < st.ke) return x;
    if (idx == 0) return kZigExpR - log1p(-g.next_double());
    if ((kZigFe[idx - 1] - kZigFe[idx]) * g.next_double() + kZigFe[idx] < exp(-x)) return x;
  }
}

inline int64_t geometric_search(Stream &g, double p) {
  int64_t X = 1;
  double sum = p, prod = p;
  const double q = 1.0 - p, U = g.next_double();
  while (U > sum) {
    prod *= q;
    sum += prod;
    ++X;
  }
  return X;
}

inline int64_t geometric_inversion(Stream &g, double log1m_p) {
  const double z = ceil(-standard_exponential(g) / log1m_p);
  if (z >= 9.223372036854776e+18) return INT64_MAX;
  return (int64_t)z;
}

// Eight geometric(p < 1/3) draws at a time: the common path of the ziggurat (one output per draw, 98.9 %) for eight
// consecutive outputs at once.  k = how many leading draws took it: the caller consumes exactly k outputs and handles
// the next draw (a wedge or tail case, which reads further outputs) with the scalar code.  Same operations as the
// scalar path lane by lane - shift / mask, exact u64 -> f64, one multiply, one divide, ceil, truncation, all
// correctly rounded - so the bits are the same.  The positions (running sum of the gaps, clamped at `far` once
// they leave the batch, exactly like the scalar code) come from an in-register prefix sum; they only grow, so the
// k values are stored unconditionally and the count of those inside the batch advances the output cursor.
struct Geo8 {
  int k;          // draws taken
  int inside;     // of those, positions < num_samples
  int64_t pos;    // position after the k-th draw
};
__attribute__((target("avx512f,avx512dq,avx512vl"))) inline Geo8 geometric8(const uint64_t *raw, double log1m_p, int64_t pos,
                                                                          int64_t far, int64_t num_samples, uint32_t *dst) {
  const __m512i v = _mm512_loadu_si512((const void *)raw);
  __m512i ri = _mm512_srli_epi64(v, 3);
  const __m512i idx = _mm512_and_si512(ri, _mm512_set1_epi64(0xFF));
  ri = _mm512_srli_epi64(ri, 8);
  const __m512i ke = _mm512_i64gather_epi64(idx, (const void *)kZigKe, 8);
  const __m512d we = _mm512_i64gather_pd(idx, (const void *)kZigWe, 8);
  const __m512d x = _mm512_mul_pd(_mm512_cvtepu64_pd(ri), we);
  const __mmask8 ok = _mm512_cmplt_epu64_mask(ri, ke);
  const __m512d nx = _mm512_xor_pd(x, _mm512_set1_pd(-0.0));  // unary minus, as the scalar code
  const __m512d z = _mm512_roundscale_pd(_mm512_div_pd(nx, _mm512_set1_pd(log1m_p)), _MM_FROUND_TO_POS_INF | _MM_FROUND_NO_EXC);
  __m512i p = _mm512_cvttpd_epi64(z);
  const __m512i zero = _mm512_setzero_si512();
  p = _mm512_add_epi64(p, _mm512_alignr_epi64(p, zero, 7));
  p = _mm512_add_epi64(p, _mm512_alignr_epi64(p, zero, 6));
  p = _mm512_add_epi64(p, _mm512_alignr_epi64(p, zero, 4));
  p = _mm512_min_epi64(_mm512_add_epi64(p, _mm512_set1_epi64(pos)), _mm512_set1_epi64(far));
  Geo8 r;
  r.k = __builtin_ctz(~(unsigned)ok | 0x100u);
  const __mmask8 taken = (__mmask8)((1u << r.k) - 1u);
  r.inside = __builtin_popcount((unsigned)(_mm512_cmplt_epi64_mask(p, _mm512_set1_epi64(num_samples)) & taken));
  _mm256_storeu_si256((__m256i *)dst, _mm512_cvtepi64_epi32(p));  // 8 values; only the first `inside` are kept
  alignas(64) int64_t tmp[8];
  _mm512_store_si512((void *)tmp, p);
  r.pos = r.k ? tmp[r.k - 1] : pos;
  return r;
}

// The eight-wide loop of one channel (in a function of its own so that geometric8 - compiled for AVX-512 - inlines).
__attribute__((target("avx512f,avx512dq,avx512vl"))) void geometric_run(Stream &g, double l, int64_t n_draws, int64_t far,
                                                                      int64_t num_samples, uint32_t *frow, size_t room,
                                                                      int64_t &i, int64_t &pos, size_t &n) {
  while (i + 8 <= n_draws && n + 8 <= room) {
    g.ensure(8);
    const Geo8 r = geometric8(g.buf + g.pos, l, pos, far, num_samples, frow + n);
    n += (size_t)r.inside;
    pos = r.pos;
    g.pos += r.k;
    i += r.k;
    if (r.k < 8) {  // this draw needs the wedge / tail code (it reads further outputs)
      const int64_t gap = geometric_inversion(g, l);
      pos = (gap >= far || pos + gap >= far) ? far : pos + gap;
      frow[n] = (uint32_t)pos;
      n += (size_t)((pos < num_samples) & (n < room));
      ++i;
    }
  }
}


bool have_avx512();

// ---------------------------------------------------------------------------
// Worker pool (persistent: a call per 10^6-row batch every millisecond or two cannot afford thread creation).
// run(n, fn) hands fn(0..n-1) to the workers and returns at once; wait() joins the job.  One job at a time.
// ---------------------------------------------------------------------------
class Pool {
 public:
  explicit Pool(int n_threads) {
    for (int t = 0; t < n_threads; ++t) threads_.emplace_back([this] { loop(); });
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto &t : threads_) t.join();
  }
  int size() const { return (int)threads_.size(); }
  void run(int n_items, std::function<void(int)> fn) {
    std::lock_guard<std::mutex> lk(m_);
    fn_ = std::move(fn);
    n_items_ = n_items;
    next_.store(0, std::memory_order_relaxed);
    active_ = (int)threads_.size();
    ++generation_;
    cv_.notify_all();
  }
  // the caller takes items too, then waits for the workers to leave the job
  void help_and_wait() {
    work();
    std::unique_lock<std::mutex> lk(m_);
    done_cv_.wait(lk, [this] { return active_ == 0; });
  }

 private:
  void work() {
    for (;;) {
      const int i = next_.fetch_add(1, std::memory_order_relaxed);
      if (i >= n_items_) break;
      fn_(i);
    }
  }
  void loop() {
    unsigned long long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return stop_ || generation_ != seen; });
        if (stop_) return;
        seen = generation_;
      }
      work();
      {
        std::lock_guard<std::mutex> lk(m_);
        if (--active_ == 0) done_cv_.notify_all();
      }
    }
  }
  std::vector<std::thread> threads_;
  std::mutex m_;
  std::condition_variable cv_, done_cv_;
  std::function<void(int)> fn_;
  std::atomic<int> next_{0};
  int n_items_ = 0, active_ = 0;
  unsigned long long generation_ = 0;
  bool stop_ = false;
};

Pool &pool(int want_threads) {  // want_threads - 1 workers beside the caller (created once; the first caller decides)
  static Pool p(std::max(0, std::min(31, want_threads - 1)));
  return p;
}

// ---------------------------------------------------------------------------
// Prefetched stream.  The raw PCG64 stream is a pure function of the index (Brown's jump), and so is the COMMON path of
// the ziggurat on an output (x = (out >> 11) * we[idx], taken when (out >> 11) < ke[idx]: 98.9 %): only what a channel
// does with them - divide by ITS log1p(-p), accumulate positions, decide where its draws end - is sequential.  So
// the workers fill, block by block and ahead of the consumer,
//     raw[i] = output i            ex[i] = the exponential a draw starting at output i returns on the common path,
//                                          -1 where that draw needs the wedge / tail code (which reads further outputs)
// and the consumer walks the channels over those arrays: eight divides, a ceil, a prefix sum and a store per eight
// draws - a fifth of the sequential work (the generator's 128-bit multiplies and the two table gathers are gone from it).
// Same values in the same order as the blocked Stream above, hence as numpy.
// ---------------------------------------------------------------------------
struct PStream {
  static constexpr uint64_t kBlk = 16384;
  Pcg start;
  u128 m4, c4;
  std::vector<uint64_t> *raw_v;
  std::vector<double> *ex_v;
  std::vector<uint8_t> *extra_v;
  uint64_t *buf = nullptr;   // raw outputs (the name Stream uses)
  double *ex = nullptr;      // the exponential a draw starting at this output returns ...
  uint8_t *extra = nullptr;  // ... and how many FURTHER outputs it consumes (0: common path; 255: not precomputed)
  uint64_t pos = 0;          // stream index of the next output
  uint64_t n_blocks = 0;     // blocks handed to the workers
  uint64_t ready_upto = 0;   // outputs [0, ready_upto) are known to be complete
  std::unique_ptr<std::atomic<uint8_t>[]> done;  // per block
  bool job_open = false;
  Pool *pl = nullptr;

  PStream(const Pcg &g, uint64_t expect_outputs, Pool *p, std::vector<uint64_t> &rv, std::vector<double> &ev, std::vector<uint8_t> &xv)
      : start(g), raw_v(&rv), ex_v(&ev), extra_v(&xv), pl(p) {
    const u128 m2 = kMult * kMult;
    m4 = m2 * m2;
    c4 = start.inc * (kMult * m2 + m2 + kMult + 1);
    n_blocks = (expect_outputs + kBlk - 1) / kBlk + 1;
    const size_t cap = (size_t)(n_blocks * kBlk + 64);
    if (rv.size() < cap) rv.resize(cap);
    if (ev.size() < cap) ev.resize(cap);
    if (xv.size() < cap) xv.resize(cap);
    buf = rv.data();
    ex = ev.data();
    extra = xv.data();
    done.reset(new std::atomic<uint8_t>[n_blocks]);
    for (uint64_t b = 0; b < n_blocks; ++b) done[b].store(0, std::memory_order_relaxed);
    job_open = true;
    pl->run((int)n_blocks, [this](int b) { fill_block((uint64_t)b); done[b].store(1, std::memory_order_release); });
  }
  ~PStream() { close(); }
  void close() {
    if (job_open) {
      pl->help_and_wait();
      job_open = false;
    }
  }
  static inline uint64_t out(u128 st) {
    const uint64_t hi = (uint64_t)(st >> 64), lo = (uint64_t)st;
    const unsigned rot = (unsigned)(hi >> 58);
    const uint64_t x = hi ^ lo;
    return (x >> rot) | (x << ((64u - rot) & 63u));
  }
  void fill_block(uint64_t b);
  // outputs [pos, pos + need) complete?  (waits for the workers; beyond the prepared range: more blocks, filled here)
  inline void ensure(uint64_t need) {
    if (pos + need <= ready_upto) return;
    ensure_slow(need);
  }
  double waited_ms = 0;
  void ensure_slow(uint64_t need) {
    const auto t0 = std::chrono::steady_clock::now();
    struct Acc { double &d; std::chrono::steady_clock::time_point t; ~Acc() { d += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); } } acc{waited_ms, t0};
    const uint64_t want = pos + need;
    while (ready_upto < want) {
      const uint64_t b = ready_upto / kBlk;
      if (b < n_blocks) {
        while (!done[b].load(std::memory_order_acquire)) _mm_pause();
        ready_upto = (b + 1) * kBlk;
        continue;
      }
      // the estimate of the stream length was short (it never is for the reference's n_draws rule): grow, fill here
      close();
      const uint64_t nb = n_blocks + std::max<uint64_t>(8, n_blocks / 4);
      raw_v->resize((size_t)(nb * kBlk + 64));
      ex_v->resize((size_t)(nb * kBlk + 64));
      extra_v->resize((size_t)(nb * kBlk + 64));
      buf = raw_v->data();
      ex = ex_v->data();
      extra = extra_v->data();
      for (uint64_t k = n_blocks; k < nb; ++k) fill_block(k);
      std::unique_ptr<std::atomic<uint8_t>[]> d2(new std::atomic<uint8_t>[nb]);
      for (uint64_t k = 0; k < nb; ++k) d2[k].store(1, std::memory_order_relaxed);
      done.swap(d2);
      n_blocks = nb;
    }
  }
  inline uint64_t next64() {
    ensure(1);
    return buf[pos++];
  }
  inline double next_double() { return (double)(next64() >> 11) * (1.0 / 9007199254740992.0); }
  uint64_t position() const { return pos; }
  void skip(uint64_t n) { pos += n; }
  Pcg finish() {
    close();
    Pcg g = start;
    advance(g, pos);
    return g;
  }
};

__attribute__((target("avx512f,avx512dq,avx512vl"))) static void zig_common_avx512(const uint64_t *raw, double *ex, uint8_t *extra, uint64_t n) {
  for (uint64_t i = 0; i < n; i += 8) {
    const __m512i v = _mm512_loadu_si512((const void *)(raw + i));
    __m512i ri = _mm512_srli_epi64(v, 3);
    const __m512i idx = _mm512_and_si512(ri, _mm512_set1_epi64(0xFF));
    ri = _mm512_srli_epi64(ri, 8);
    const __m512i ke = _mm512_i64gather_epi64(idx, (const void *)kZigKe, 8);
    const __m512d we = _mm512_i64gather_pd(idx, (const void *)kZigWe, 8);
    const __m512d x = _mm512_mul_pd(_mm512_cvtepu64_pd(ri), we);
    const __mmask8 ok = _mm512_cmplt_epu64_mask(ri, ke);
    _mm512_storeu_pd(ex + i, x);
    // 0 where the common path holds, 255 ("not resolved yet") elsewhere: one byte per output
    const __m128i bytes = _mm512_cvtepi64_epi8(_mm512_maskz_set1_epi64((__mmask8)~ok, 0xFF));
    _mm_storel_epi64((__m128i *)(extra + i), bytes);
  }
}

void PStream::fill_block(uint64_t b) {
  constexpr int kTail = 64;  // outputs beyond the block, for wedge / tail draws that start near its end
  Pcg g = start;
  advance(g, b * kBlk);
  u128 st = g.state, s[4];
  for (int j = 0; j < 4; ++j) {
    st = st * kMult + start.inc;
    s[j] = st;
  }
  uint64_t *dst = buf + b * kBlk;
  u128 s0 = s[0], s1 = s[1], s2 = s[2], s3 = s[3];
  for (uint64_t t = 0; t < kBlk / 4; ++t) {
    dst[0] = out(s0); dst[1] = out(s1); dst[2] = out(s2); dst[3] = out(s3);
    s0 = s0 * m4 + c4; s1 = s1 * m4 + c4; s2 = s2 * m4 + c4; s3 = s3 * m4 + c4;
    dst += 4;
  }
  double *e = ex + b * kBlk;
  uint8_t *xt = extra + b * kBlk;
  const uint64_t *r = buf + b * kBlk;
  if (have_avx512()) {
    zig_common_avx512(r, e, xt, kBlk);
  } else {
    for (uint64_t i = 0; i < kBlk; ++i) {
      uint64_t ri = r[i] >> 3;
      const unsigned idx = (unsigned)(ri & 0xFFu);
      ri >>= 8;
      e[i] = (double)ri * kZigWe[idx];
      xt[i] = ri < kZigKe[idx] ? 0 : 255;
    }
  }
  // the other 1.1 %: the wedge / tail draw itself (standard_exponential above, on the outputs that follow), so that the
  // consumer finds every draw's value and length ready - a draw it had to redo cost it ~200 ns (exp, log1p, a
  // mispredicted branch out of the eight-wide loop)
  uint64_t tail[kTail];
  bool have_tail = false;
  auto raw_at = [&](uint64_t j) -> uint64_t {  // output j of this block's range, j < kBlk + kTail
    if (j < kBlk) return r[j];
    if (!have_tail) {
      for (int t = 0; t < kTail; t += 4) {
        tail[t] = out(s0); tail[t + 1] = out(s1); tail[t + 2] = out(s2); tail[t + 3] = out(s3);
        s0 = s0 * m4 + c4; s1 = s1 * m4 + c4; s2 = s2 * m4 + c4; s3 = s3 * m4 + c4;
      }
      have_tail = true;
    }
    return tail[j - kBlk];
  };
  for (uint64_t i0 = 0; i0 < kBlk; i0 += 8) {
    uint64_t m;
    memcpy(&m, xt + i0, 8);
    while (m) {
      const unsigned lane = (unsigned)__builtin_ctzll(m) >> 3;
      m &= ~(0xFFull << (8 * lane));
      const uint64_t i = i0 + lane;
      uint64_t j = i;
      double val = 0.0;
      bool resolved = false;
      while (j + 2 < kBlk + kTail) {
        uint64_t ri = raw_at(j++) >> 3;
        const unsigned idx = (unsigned)(ri & 0xFFu);
        ri >>= 8;
        const Strip &sp = kStrips.s[idx];
        const double x = (double)ri * sp.we;
        if (ri < sp.ke) { val = x; resolved = true; break; }
        const double u = (double)(raw_at(j++) >> 11) * (1.0 / 9007199254740992.0);
        if (idx == 0) { val = kZigExpR - log1p(-u); resolved = true; break; }
        if ((kZigFe[idx - 1] - kZigFe[idx]) * u + kZigFe[idx] < exp(-x)) { val = x; resolved = true; break; }
      }
      if (resolved && j - i - 1 < 255) {
        e[i] = val;
        xt[i] = (uint8_t)(j - i - 1);
      }  // else: stays 255, the consumer redoes it
    }
  }
}

inline double standard_exponential(PStream &g) {
  for (;;) {
    uint64_t ri = g.next64() >> 3;
    const unsigned idx = (unsigned)(ri & 0xFFu);
    ri >>= 8;
    const Strip &st = kStrips.s[idx];
    const double x = (double)ri * st.we;
    if (ri < st.ke) return x;
    if (idx == 0) return kZigExpR - log1p(-g.next_double());
    if ((kZigFe[idx - 1] - kZigFe[idx]) * g.next_double() + kZigFe[idx] < exp(-x)) return x;
  }
}
inline int64_t geometric_search(PStream &g, double p) {
  int64_t X = 1;
  double sum = p, prod = p;
  const double q = 1.0 - p, U = g.next_double();
  while (U > sum) {
    prod *= q;
    sum += prod;
    ++X;
  }
  return X;
}
inline int64_t geometric_inversion(PStream &g, double log1m_p) {
  const double z = ceil(-standard_exponential(g) / log1m_p);
  if (z >= 9.223372036854776e+18) return INT64_MAX;
  return (int64_t)z;
}

// geometric8 on the prefetched exponentials: the per-channel part - quotient, ceil, running sum.
//   gap = ceil(-e / log1p(-p)) must be numpy's, bit for bit, but the DIVIDE is now the consumer's bottleneck (one
//   512-bit vdivpd per eight draws: 16 cycles on Skylake-X, ~10 on Zen 4).  q = e * (-1 / l) differs from the exact
//   quotient by at most 2 ulp, so ceil(q) is the exact ceil unless an integer lies within that distance of q: the
//   group then (never, in practice: e is continuous) takes the divide.  Everything but `pos = min(pos + total, far)`
//   is off the dependency chain: the prefix sum is local to the group, its last lane is the group's total.
struct GeoCtx {
  double l, inv;  // log1p(-p), -1 / l
};
// Up to eight draws of one channel from the prefetched arrays, starting at output `at`.  Lanes after the first wedge /
// tail draw belong to outputs that draw consumed: the group ends with it (`taken` draws, `advance` outputs).
struct GeoP {
  int taken, inside, redo;  // redo: the draw after the taken ones was not precomputed (extra == 255): scalar code
  uint64_t advance;
  int64_t pos;
};
__attribute__((target("avx512f,avx512dq,avx512vl"))) inline GeoP geometric8_pre(const double *ex, const uint8_t *extra, const GeoCtx &c,
                                                                              int64_t pos, int64_t far, int64_t num_samples, uint32_t *dst) {
  const __m512d x = _mm512_loadu_pd(ex);
  uint64_t xm;
  memcpy(&xm, extra, 8);
  const __m512d q = _mm512_mul_pd(x, _mm512_set1_pd(c.inv));
  __m512d z = _mm512_roundscale_pd(q, _MM_FROUND_TO_POS_INF | _MM_FROUND_NO_EXC);
  // an integer within 4 ulp of q (either side)?  |q - rint(q)| <= q * 2^-50
  const __m512d near = _mm512_roundscale_pd(q, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
  const __m512d dist = _mm512_abs_pd(_mm512_sub_pd(q, near));
  const __mmask8 close = _mm512_cmp_pd_mask(dist, _mm512_mul_pd(q, _mm512_set1_pd(8.8817841970012523e-16)), _CMP_LE_OQ);
  if (__builtin_expect(close != 0, 0)) {
    const __m512d nx = _mm512_xor_pd(x, _mm512_set1_pd(-0.0));
    z = _mm512_roundscale_pd(_mm512_div_pd(nx, _mm512_set1_pd(c.l)), _MM_FROUND_TO_POS_INF | _MM_FROUND_NO_EXC);
  }
  // numpy: z >= 9.22e18 -> INT64_MAX; here every gap saturates at `far` (> num_samples) - a position that left the batch
  // stays outside either way
  z = _mm512_min_pd(z, _mm512_set1_pd((double)far));
  __m512i p = _mm512_cvttpd_epi64(z);
  const __m512i zero = _mm512_setzero_si512();
  p = _mm512_add_epi64(p, _mm512_alignr_epi64(p, zero, 7));
  p = _mm512_add_epi64(p, _mm512_alignr_epi64(p, zero, 6));
  p = _mm512_add_epi64(p, _mm512_alignr_epi64(p, zero, 4));
  GeoP r;
  r.redo = 0;
  if (__builtin_expect(xm == 0, 1)) {
    const int64_t total = _mm_extract_epi64(_mm512_extracti64x2_epi64(p, 3), 1);
    p = _mm512_min_epi64(_mm512_add_epi64(p, _mm512_set1_epi64(pos)), _mm512_set1_epi64(far));
    r.taken = 8;
    r.advance = 8;
    r.inside = __builtin_popcount((unsigned)_mm512_cmplt_epi64_mask(p, _mm512_set1_epi64(num_samples)));
    _mm256_storeu_si256((__m256i *)dst, _mm512_cvtepi64_epi32(p));
    r.pos = pos + total >= far ? far : pos + total;
    return r;
  }
  const int lane = __builtin_ctzll(xm) >> 3;  // the first wedge / tail draw
  const unsigned ext = (unsigned)((xm >> (8 * lane)) & 0xFFu);
  r.redo = ext == 255u;
  r.taken = r.redo ? lane : lane + 1;
  r.advance = (uint64_t)r.taken + (r.redo ? 0u : ext);
  p = _mm512_min_epi64(_mm512_add_epi64(p, _mm512_set1_epi64(pos)), _mm512_set1_epi64(far));
  const __mmask8 taken = (__mmask8)((1u << r.taken) - 1u);
  r.inside = __builtin_popcount((unsigned)(_mm512_cmplt_epi64_mask(p, _mm512_set1_epi64(num_samples)) & taken));
  _mm256_storeu_si256((__m256i *)dst, _mm512_cvtepi64_epi32(p));
  alignas(64) int64_t tmp[8];
  _mm512_store_si512((void *)tmp, p);
  r.pos = r.taken ? tmp[r.taken - 1] : pos;
  return r;
}

__attribute__((target("avx512f,avx512dq,avx512vl"))) void geometric_run(PStream &g, double l, int64_t n_draws, int64_t far,
                                                                      int64_t num_samples, uint32_t *frow, size_t room, int64_t &i_io,
                                                                      int64_t &pos_io, size_t &n_io) {
  const GeoCtx c{l, -1.0 / l};
  int64_t i = i_io, pos = pos_io;
  size_t n = n_io;
  uint64_t at = g.pos;
  while (i + 8 <= n_draws && n + 8 <= room) {
    if (__builtin_expect(at + 8 > g.ready_upto, 0)) {
      g.pos = at;
      g.ensure(8);
    }
    // the lines were written by other cores: ask for them early
    _mm_prefetch((const char *)(g.ex + at + 512), _MM_HINT_T0);
    _mm_prefetch((const char *)(g.extra + at + 1024), _MM_HINT_T0);
    const GeoP r = geometric8_pre(g.ex + at, g.extra + at, c, pos, far, num_samples, frow + n);
    n += (size_t)r.inside;
    pos = r.pos;
    at += r.advance;
    i += r.taken;
    if (__builtin_expect(r.redo, 0)) {  // (a draw the fill could not finish inside its look-ahead: never seen)
      g.pos = at;
      const int64_t gap = geometric_inversion(g, l);
      at = g.pos;
      pos = (gap >= far || pos + gap >= far) ? far : pos + gap;
      frow[n] = (uint32_t)pos;
      n += (size_t)((pos < num_samples) & (n < room));
      ++i;
    }
  }
  g.pos = at;
  i_io = i;
  pos_io = pos;
  n_io = n;
}

bool have_avx512() {
  static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512vl");
  return ok;
}

// scratch that survives between calls (one sampler thread per process is the rule; thread_local keeps others safe)
struct Scratch {
  std::vector<uint32_t> row;      // fired rows, channel after channel
  std::vector<uint16_t> outcome;  // outcome index per fired row (multi-outcome channels only)
  std::vector<uint64_t> raw;      // prefetched stream: outputs, ...
  std::vector<double> ex;         // ... the exponentials of draws starting there ...
  std::vector<uint8_t> extra;     // ... and their lengths
};
// One process-wide scratch (tens of MB for a 10^6-row batch: a sampler thread that a caller creates per sample() call must
// not pay for allocating and faulting it in every time), one call at a time.
Scratch g_scratch;
std::mutex g_scratch_mutex;

// generator-agnostic helpers of pass 1
inline void skip_outputs(Pcg &g, uint64_t n) { advance(g, n); }
inline void skip_outputs(Stream &g, uint64_t n) { g.skip(n); }
inline double draw_double(Pcg &g) { return next_double(g); }
inline double draw_double(Stream &g) { return g.next_double(); }
inline void skip_outputs(PStream &g, uint64_t n) { g.skip(n); }
inline double draw_double(PStream &g) { return g.next_double(); }

// pass 1 of tsim_pcg_sample_channels for either generator form
template <class G>
int draw_channels(G &g, int32_t n_channels, const double *p_fire, const int32_t *n_outcomes, const double *cond_cdf,
                  int64_t num_samples, Scratch &S, std::vector<size_t> &chan_begin, std::vector<size_t> &table_off,
                  size_t &n_fires_out) {
  {
    double cap = 0;
    for (int c = 0; c < n_channels; ++c) {
      const double e = (double)num_samples * p_fire[c];
      cap += e + 7.0 * sqrt(e) + 101.0;
      table_off[c + 1] = table_off[c] + (size_t)std::max(0, n_outcomes[c]);
      if (n_outcomes[c] > 65535) return tsim_fail(TSIM_ENOTSUP, "channel %d has more than 65535 outcomes", c);
    }
    if (S.row.size() < (size_t)cap + 32) S.row.resize((size_t)cap + 32);
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
      int64_t i = 0;
      if constexpr (std::is_same<G, Stream>::value || std::is_same<G, PStream>::value) {
        // eight draws per step while the common path holds; the scalar loop below finishes the rest (the draw
        // that left the common path, and tiny p whose gaps may not fit an int64)
        if (l < -1e-15) {
          geometric_run(g, l, n_draws, far, num_samples, frow, room, i, pos, n);
        }
      }
      for (; i < n_draws; ++i) {
        const int64_t gap = geometric_inversion(g, l);
        pos = (gap >= far || pos + gap >= far) ? far : pos + gap;
        frow[n] = (uint32_t)pos;
        n += (size_t)((pos < num_samples) & (n < room));
      }
    }
    // one uniform per fired row; outcome = searchsorted(cdf, u) (first entry >= u; cdf[-1] == 1 > u).  A channel
    // with a single non-identity outcome needs no value - only the stream position: jump over its uniforms.
    if (nout == 1) {
      skip_outputs(g, (uint64_t)(n - n_fires));
    } else {
      for (size_t k = n_fires; k < n; ++k) {
        const double u = draw_double(g);
        int o = 0;
        while (o < nout - 1 && cdf[o] < u) ++o;
        fout[k] = (uint16_t)o;
      }
    }
    n_fires = n;
  }
  n_fires_out = n_fires;
  return TSIM_OK;
}

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
  static const bool timing = getenv("TSIM_AMD_DEBUG") != nullptr && strstr(getenv("TSIM_AMD_DEBUG"), "pcg") != nullptr;
  const bool force_scalar = getenv("TSIM_PCG_SCALAR") != nullptr, force_serial = getenv("TSIM_PCG_SERIAL") != nullptr;
  const auto t_start = std::chrono::steady_clock::now();
  int T = threads > 0 ? threads : (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()));
  if (const char *e = getenv("TSIM_PCG_THREADS")) T = std::max(1, atoi(e));
  Pool &pl = pool(T);
  T = std::min(T, pl.size() + 1);
  // ---- pass 1 (the stream is one dependency chain): per channel the fired rows and their outcomes
  std::lock_guard<std::mutex> scratch_lock(g_scratch_mutex);
  Scratch &S = g_scratch;
  std::vector<size_t> chan_begin((size_t)n_channels + 1, 0), table_off((size_t)n_channels + 1, 0);
  size_t n_fires = 0;
  Pcg g = load(rng);
  // how many outputs the call will consume, about: per channel n_draws (+ 1.2 % wedge / tail extras) and one per fire
  double expect = 4096.0;
  for (int c = 0; c < n_channels; ++c) {
    const double e = (double)num_samples * p_fire[c];
    const double nd = e + 7.0 * sqrt(std::max(0.0, e * (1.0 - p_fire[c]))) + 100.0;
    expect += nd * 1.02 + 64.0 + e + 8.0 * sqrt(std::max(0.0, e)) + 16.0;
  }
  const bool prefetch = have_avx512() && !force_scalar && !force_serial && T > 1 && expect >= 65536.0 && expect < 1.5e8;
  if (prefetch) {
    PStream st(g, (uint64_t)expect, &pl, S.raw, S.ex, S.extra);
    const int r = draw_channels(st, n_channels, p_fire, n_outcomes, cond_cdf, num_samples, S, chan_begin, table_off, n_fires);
    if (timing)
      fprintf(stderr, "  consumer done at %.2f ms, %.2f ms of them waiting for blocks (position %llu of %llu prepared)\n",
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(), st.waited_ms,
              (unsigned long long)st.position(), (unsigned long long)(st.n_blocks * PStream::kBlk));
    g = st.finish();
    if (r) return r;
  } else if (have_avx512() && !force_scalar) {
    Stream st(g);
    if (int r = draw_channels(st, n_channels, p_fire, n_outcomes, cond_cdf, num_samples, S, chan_begin, table_off, n_fires)) return r;
    g = st.finish();
  } else {
    if (int r = draw_channels(g, n_channels, p_fire, n_outcomes, cond_cdf, num_samples, S, chan_begin, table_off, n_fires)) return r;
  }
  uint32_t *const frow = S.row.data();
  uint16_t *const fout = S.outcome.data();
  chan_begin[n_channels] = n_fires;
  store(rng, g);
  const auto t_pass1 = std::chrono::steady_clock::now();
  // ---- pass 2: zero the rows and XOR the patterns in, tile by tile over shot ranges (a tile of rows is cleared and
  //      stays cache resident while every channel's cursor sweeps it), the tile ranges shared out over the pool
  const int64_t tile_rows = std::max<int64_t>(1024, (256 * 1024) / (8 * (int64_t)words));
  const int64_t n_tiles = (num_samples + tile_rows - 1) / tile_rows;
  const int parts = (int)std::max<int64_t>(1, std::min<int64_t>(n_fires < 16384 && num_samples < 65536 ? 1 : 4 * T, n_tiles));
  auto work = [&](int t) {
    std::vector<size_t> cur((size_t)n_channels);
    const int64_t tile_lo = n_tiles * t / parts, tile_hi = n_tiles * (t + 1) / parts;
    const uint32_t first_row = (uint32_t)std::min<int64_t>(num_samples, tile_lo * tile_rows);
    for (int c = 0; c < n_channels; ++c)  // first fire of this range in every channel
      cur[c] = (size_t)(std::lower_bound(frow + chan_begin[c], frow + chan_begin[c + 1], first_row) - frow);
    for (int64_t tile = tile_lo; tile < tile_hi; ++tile) {
      const uint32_t begin_row = (uint32_t)std::min<int64_t>(num_samples, tile * tile_rows);
      const uint32_t end_row = (uint32_t)std::min<int64_t>(num_samples, (tile + 1) * tile_rows);
      memset(rows + (size_t)begin_row * words, 0, (size_t)(end_row - begin_row) * words * 8);
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
  if (parts <= 1) {
    work(0);
  } else {
    pl.run(parts, work);
    pl.help_and_wait();
  }
  const size_t fires_total = n_fires;
  if (timing)
    fprintf(stderr, "tsim_pcg_sample_channels: %lld rows, %zu fires: draws (%s) %.2f ms, zero + scatter (%d threads) %.2f ms\n",
            (long long)num_samples, fires_total, prefetch ? "prefetched stream" : "serial stream",
            std::chrono::duration<double, std::milli>(t_pass1 - t_start).count(), T,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_pass1).count());
  return TSIM_OK;
}
