// standalone: how fast are (a) the 4-chain generation, (b) the 8-wide consumer on this CPU?
#include <cstdio>
#include <cstdint>
#include <chrono>
#include <cmath>
#include <immintrin.h>
#include "../../tsim_amd/csrc/tsim_zig_tables.h"
typedef unsigned __int128 u128;
static const u128 kMult = ((u128)0x2360ED051FC65DA4ull << 64) | 0x4385DF649FCCF645ull;
static inline uint64_t outp(u128 st) { uint64_t hi = st >> 64, lo = (uint64_t)st; unsigned rot = hi >> 58; uint64_t x = hi ^ lo; return (x >> rot) | (x << ((64u - rot) & 63u)); }
template <int K> double gen(uint64_t *buf, int n, int reps) {
  u128 inc = 12345 * 2 + 1, s[K]; u128 st = 42;
  u128 mk = 1, ck = 0; for (int i = 0; i < K; ++i) { ck = ck * kMult + inc; mk *= kMult; }
  for (int j = 0; j < K; ++j) { st = st * kMult + inc; s[j] = st; }
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r)
    for (int t = 0; t < n; t += K)
      for (int j = 0; j < K; ++j) { buf[t + j] = outp(s[j]); s[j] = s[j] * mk + ck; }
  return std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / ((double)n * reps);
}
__attribute__((target("avx512f,avx512dq,avx512vl"))) double consume(const uint64_t *buf, int n, int reps, uint32_t *dst) {
  auto t0 = std::chrono::steady_clock::now();
  int64_t pos = 0; long acc = 0;
  for (int r = 0; r < reps; ++r)
    for (int t = 0; t + 8 <= n; t += 8) {
      const __m512i v = _mm512_loadu_si512((const void *)(buf + t));
      __m512i ri = _mm512_srli_epi64(v, 3);
      const __m512i idx = _mm512_and_si512(ri, _mm512_set1_epi64(0xFF));
      ri = _mm512_srli_epi64(ri, 8);
      const __m512i ke = _mm512_i64gather_epi64(idx, (const void *)kZigKe, 8);
      const __m512d we = _mm512_i64gather_pd(idx, (const void *)kZigWe, 8);
      const __m512d x = _mm512_mul_pd(_mm512_cvtepu64_pd(ri), we);
      const __mmask8 ok = _mm512_cmplt_epu64_mask(ri, ke);
      const __m512d z = _mm512_roundscale_pd(_mm512_div_pd(_mm512_xor_pd(x, _mm512_set1_pd(-0.0)), _mm512_set1_pd(-0.0202)), _MM_FROUND_TO_POS_INF | _MM_FROUND_NO_EXC);
      __m512i p = _mm512_cvttpd_epi64(z);
      const __m512i zero = _mm512_setzero_si512();
      p = _mm512_add_epi64(p, _mm512_alignr_epi64(p, zero, 7));
      p = _mm512_add_epi64(p, _mm512_alignr_epi64(p, zero, 6));
      p = _mm512_add_epi64(p, _mm512_alignr_epi64(p, zero, 4));
      p = _mm512_min_epi64(_mm512_add_epi64(p, _mm512_set1_epi64(pos)), _mm512_set1_epi64(1000001));
      _mm256_storeu_si256((__m256i *)(dst + (t & 1023)), _mm512_cvtepi64_epi32(p));
      alignas(64) int64_t tmp[8]; _mm512_store_si512((void *)tmp, p);
      int k = __builtin_ctz(~(unsigned)ok | 0x100u);
      pos = tmp[k ? k - 1 : 0] & 1023; acc += k;
    }
  double ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / ((double)n * reps);
  if (acc == 1) printf("x");
  return ns;
}
int main() {
  static uint64_t buf[4096 + 16]; static uint32_t dst[2048];
  printf("generation ns/output: 1 chain %.2f, 2 chains %.2f, 4 chains %.2f, 8 chains %.2f\n", gen<1>(buf, 4096, 2000), gen<2>(buf, 4096, 2000), gen<4>(buf, 4096, 2000), gen<8>(buf, 4096, 2000));
  printf("8-wide consumer ns/draw: %.2f\n", consume(buf, 4096, 2000, dst));
}
