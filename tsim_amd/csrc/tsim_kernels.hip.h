// tsim_kernels.hip.h - gfx950 device code of the stabilizer-rank sampling engine.
//
// One lane = one shot.  The whole autoregressive loop of a batch
// (reference: src/tsim/sampler.py:28-167) runs in ONE kernel: the shot's packed
// f row is staged in LDS, every level's GF(2) rows are wave-uniform and come
// in through the scalar data cache (s_load -> SGPR operands of v_and/v_bcnt),
// parities feed the exact Z[omega]*2^k arithmetic held in VGPRs
// (reference: src/tsim/core/exact_scalar.py), the marginal is formed in float32
// and the Bernoulli draw uses an in-kernel Threefry-2x32 keyed by the global
// in-batch shot index, so results do not depend on how a batch is sharded.
//
// Integer semantics deliberately mirror the reference's int32 arithmetic
// (wrap-around, one /2 step per multiply/add, fix-point at the end of each
// scan) so that integer amplitudes are bit-identical, not just equal in value.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tsimk {

// program image is read through the constant address space => scalar loads
typedef const __attribute__((address_space(4))) uint32_t *cptr;

// ---- image layout constants (uint32 words) --------------------------------
// component record
enum { C_NOUT = 0, C_F, C_W, C_FSEL, C_LEVELS, C_OUTPOS, C_KEYBASE, C_NLEVELS, C_WORDS = 8 };
// level record
enum { L_G = 0, L_GRAPHS, L_FLAGS, L_NPARAMS, L_FRAME, L_HWROWS /* fast layout: image offset of the level's uniform-stride row stream */,
       L_HWN /* rows in it */, L_WORDS = 8 };
// graph record
enum {
  G_NA = 0, G_NB, G_NC, G_ND, G_ROWS, G_PHASE, G_FFA, G_FFB, G_FFC, G_FFD, G_POW2,
  G_APRE, G_APIM, G_FLAGS, G_WORDS = 16
};
#define TSIMK_LFLAG_APPROX 1u
#define TSIMK_LFLAG_FIXED 2u   // fast layout: table entries pre-shifted to the power L_FRAME
#define TSIMK_GFLAG_LAM 1u
#define TSIMK_GFLAG_LIN 2u
#define TSIMK_GFLAG_D_TABLED 4u
#define TSIMK_GFLAG_FF_IS_ONE 1u

struct SampleArgs {
  const uint32_t *img;      // program image
  const uint64_t *f;        // [B, WF] packed error-mechanism rows
  uint64_t *out;            // [B, WO] packed outputs (nullptr when only out_compact is wanted)
  const uint32_t *subkeys;  // [total compiled outputs, 2] Threefry subkeys
  float *norm_dev;          // [n_components] or nullptr
  long long B;              // rows in this launch
  long long shot_offset;    // in-batch index of row 0
  int WF, WO;               // 64-bit words per f row / out row
  int n_direct, direct_off; // direct table: (src | flip<<31, dst) pairs
  // the same moves as a gather program (bit-field runs, see gather_runs below): image offset (64-byte aligned) and
  // number of 16-word chunks; direct_chunks == 0: walk the pair table instead
  int direct_prog, direct_chunks;
  int n_comp, comp_off;
  // optional row indirection (device-side post-selection): launch slot i handles row
  // row_index[i] for i < *row_count; the Threefry counter stays the row's own in-batch index,
  // so results do not depend on the order of the list.  nullptr = identity.
  const uint32_t *row_index;
  const uint32_t *row_count;
  // row_lists > 1 (two-pass launches): row_lists sub-lists of capacity row_list_cap each; list k
  // holds row_count[32 * k] entries at row_index + k * row_list_cap and is served by the blocks with
  // blockIdx % row_lists == k (several counters instead of one: same-address atomics serialise)
  int row_lists, row_list_cap;
  // list mode only: this launch serves the slots [row_slot_begin, row_slot_end) of every list
  // (row_slot_end == 0: to the end) - lets two kernels split the lists between them
  int row_slot_begin, row_slot_end;
  // normalisation-check row (sampler.py:66-72).  Default: in-batch shot 0, or the first listed row
  // when a row list is given.  check_row != nullptr: the listed row whose index equals *check_row
  // (two-pass launches: the list is unordered).  no_check != 0: nobody.
  const uint32_t *check_row;
  int no_check;
  // per-output subkeys computed by the host and passed in the kernel arguments (programs with at
  // most TSIMK_INLINE_KEYS compiled outputs): no key-generation kernel, no shared key buffer
  int n_inline_keys;
  uint32_t inline_keys[2 * 32];
  // optional second output: the same rows as ceil(num_outputs/8)-byte strings (the reference's
  // bit_packed layout, sampler.py:665-669) - what a gather moves; nullptr = not wanted
  uint8_t *out_compact;
  int out_rb;
  // byte offset of THIS struct inside the kernel-argument segment (0 for every kernel whose first
  // argument starts with it; the multi-launch hard-row kernel carries an array of them)
  int kernarg_off;
};
#define TSIMK_INLINE_KEYS 32

// the lane's output row, word w at lds_o[w * stride], as out_rb bytes of the compact output
// Stage a shot's packed f row into its LDS column ([word][lane], stride `stride`).  The first eight 64-bit words are
// loaded by a fully unrolled, guarded sequence - all loads in flight before the first LDS store - instead of a
// load-wait-store loop (one HBM latency per word); longer rows finish in a plain loop.
__device__ __forceinline__ void stage_f_row(const uint64_t *frow, int WF, uint32_t *lds_f, int stride) {
  uint64_t v[8];
#pragma unroll
  for (int w = 0; w < 8; ++w) v[w] = (w < WF) ? frow[w] : 0ull;
#pragma unroll
  for (int w = 0; w < 8; ++w)
    if (w < WF) {
      lds_f[(2 * w) * stride] = (uint32_t)v[w];
      lds_f[(2 * w + 1) * stride] = (uint32_t)(v[w] >> 32);
    }
  for (int w = 8; w < WF; ++w) {
    const uint64_t x = frow[w];
    lds_f[(2 * w) * stride] = (uint32_t)x;
    lds_f[(2 * w + 1) * stride] = (uint32_t)(x >> 32);
  }
}

// One bit_packed row (out_rb bytes at out_compact + row * out_rb); word w of the row comes from `word(w)`.  Rows of
// a multiple of four bytes in a 4-byte aligned buffer are written as dwords (16-byte rows: one store), the others
// byte by byte.  The choice is uniform over the launch.
template <class Word>
__device__ __forceinline__ void store_compact_words(const SampleArgs &A, long long row, Word word) {
  if (!A.out_compact) return;
  uint8_t *dst = A.out_compact + row * A.out_rb;
  const uintptr_t base = (uintptr_t)A.out_compact;
  if ((A.out_rb & 15) == 0 && (base & 15u) == 0u) {
    for (int k = 0; k < A.out_rb; k += 16) {
      const int w = k >> 2;
      *reinterpret_cast<uint4 *>(dst + k) = make_uint4(word(w), word(w + 1), word(w + 2), word(w + 3));
    }
  } else if ((A.out_rb & 3) == 0 && (base & 3u) == 0u) {
    for (int k = 0; k < A.out_rb; k += 4) *reinterpret_cast<uint32_t *>(dst + k) = word(k >> 2);
  } else {
    for (int k = 0; k < A.out_rb; ++k) dst[k] = (uint8_t)(word(k >> 2) >> (8 * (k & 3)));
  }
}
__device__ __forceinline__ void store_compact_row(const SampleArgs &A, long long row, const uint32_t *lds_o, int stride) {
  store_compact_words(A, row, [&](int w) { return lds_o[w * stride]; });
}

// A gather program moves bit fields of the packed f row to a destination bit vector.  It is a list of 4-word runs
// [ctl, mask, flip, 0], four runs per 64-byte chunk (one s_load_dwordx16; the last chunk is padded with mask = 0
// runs), built by the packer from (source bit, destination bit, flip) moves that are contiguous in both words:
//   ctl = src_shift | dst_shift << 8 | dst_word << 16 | src_word << 24
//   dst_word[dst_shift ..] |= ((f32[src_word] >> src_shift) & mask) ^ flip
// `lds_f` is the lane's LDS column of f words ([word][lane], stride nthr).  Destination words 0 and 1 are register
// accumulators, higher ones (more than 64 outputs) go to the lane's LDS column `lds_hi`.
typedef uint32_t lw_u32x16 __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(4))) lw_u32x16 *lw_cptr16;

__device__ __forceinline__ void gather_runs(const __attribute__((address_space(4))) uint32_t *prog, uint32_t nchunks,
                                            const uint32_t *lds_f, uint32_t *lds_hi, int nthr, uint32_t &a0, uint32_t &a1) {
  for (uint32_t c = 0; c < nchunks; ++c) {
    const lw_u32x16 q = *(lw_cptr16)(prog + 16u * c);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t ctl = q[4 * k];
      const uint32_t fw = lds_f[(ctl >> 24) * nthr];
      const uint32_t v = (((fw >> (ctl & 31u)) & q[4 * k + 1]) ^ q[4 * k + 2]) << ((ctl >> 8) & 31u);
      const uint32_t dw = (ctl >> 16) & 255u;
      if (dw == 0u) a0 |= v;
      else if (dw == 1u) a1 |= v;
      else lds_hi[dw * nthr] |= v;
    }
  }
}

// subkey word j (0/1) of compiled output `o` (sampler.py:74,147-148)
__device__ __forceinline__ uint32_t subkey(const SampleArgs &A, uint32_t o, uint32_t j) {
  // Wave-uniform, read through the constant address space (scalar loads): the inline keys sit in the
  // kernel-argument segment (at kernarg_off: SampleArgs is the first member of the first argument of
  // the sampling kernels, or an element of k_sample4h_multi's array) and the k_keygen buffer was
  // written by an earlier kernel, so both are read-only here.
  typedef const __attribute__((address_space(4))) uint8_t *cbytes;
  cptr kp = A.n_inline_keys
                ? (cptr)((cbytes)__builtin_amdgcn_kernarg_segment_ptr() + A.kernarg_off +
                         __builtin_offsetof(SampleArgs, inline_keys))
                : (cptr)(uintptr_t)A.subkeys;
  return kp[2u * o + j];
}

struct EvalArgs {
  const uint32_t *img;
  const uint32_t *x;        // [B, W] packed params (32-bit words)
  float *re, *im;           // [B]
  float *abs;               // [B] or nullptr
  int *exact;               // [B,5] or nullptr
  long long B;
  int level_off;            // offset of the level record
  int W;
};

// ---------------------------------------------------------------------------
// exact scalar helpers (int32, wrap-around == XLA)
// ---------------------------------------------------------------------------

// exact_scalar.py:42-49
__device__ __forceinline__ void reduce1(int &a, int &b, int &c, int &d, int &p) {
  int o = a | b | c | d;
  int sh = (((o & 1) == 0) & (o != 0)) ? 1 : 0;
  a >>= sh; b >>= sh; c >>= sh; d >>= sh;
  p += sh;
}

// fix-point of reduce1 (exact_scalar.py:119-136): strip all common factors of 2
__device__ __forceinline__ void canon(int &a, int &b, int &c, int &d, int &p) {
  int o = a | b | c | d;
  int tz = (o != 0) ? __builtin_ctz((unsigned)o) : 0;
  a >>= tz; b >>= tz; c >>= tz; d >>= tz;
  p += tz;
}

// exact_scalar.py:19-39 on basis (1, w, i, conj w)
__device__ __forceinline__ void zmul(int &a1, int &b1, int &c1, int &d1, int a2, int b2, int c2,
                                     int d2) {
  unsigned ua1 = a1, ub1 = b1, uc1 = c1, ud1 = d1, ua2 = a2, ub2 = b2, uc2 = c2, ud2 = d2;
  unsigned A = ua1 * ua2 + ub1 * ud2 - uc1 * uc2 + ud1 * ub2;
  unsigned B = ua1 * ub2 + ub1 * ua2 + uc1 * ud2 + ud1 * uc2;
  unsigned C = ua1 * uc2 + ub1 * ub2 + uc1 * ua2 - ud1 * ud2;
  unsigned D = ua1 * ud2 - ub1 * uc2 - uc1 * ub2 + ud1 * ua2;
  a1 = (int)A; b1 = (int)B; c1 = (int)C; d1 = (int)D;
}

// x << s with XLA semantics (s >= 32 -> 0), s >= 0
__device__ __forceinline__ int shl_sat(int x, int s) {
  return (s >= 32) ? 0 : (int)((unsigned)x << (s & 31));
}

// float32 constants of exact_scalar.py:15-16
#define TSIMK_E4 0.70710677f

// exact_scalar.py:87-89,218-222: one rounding per operation, no FMA
__device__ __forceinline__ void to_complex(int a, int b, int c, int d, int p, float &re, float &im) {
  float fa = (float)a, fb = (float)b, fc = (float)c, fd = (float)d;
  float r = __fadd_rn(__fadd_rn(fa, __fmul_rn(fb, TSIMK_E4)), __fmul_rn(fd, TSIMK_E4));
  float i = __fadd_rn(__fadd_rn(__fmul_rn(fb, TSIMK_E4), fc), __fmul_rn(fd, -TSIMK_E4));
  float s = ldexpf(1.0f, p);
  re = __fmul_rn(r, s);
  im = __fmul_rn(i, s);
}

// jnp.abs(complex64): max * sqrt(1 + (min/max)^2)
__device__ __forceinline__ float cabs32(float re, float im) {
  // No lane of the wave has an imaginary part (probability models: the exact sums are real): the general formula
  // gives max = |re|, min/max = 0 (or the mx == 0 / inf / NaN cases below) -> |re| bit for bit, without the
  // division and the square root (about 25 VALU operations per level).
  if (__builtin_amdgcn_ballot_w64(im != 0.0f) == 0ull) return fabsf(re);
  float ar = fabsf(re), ai = fabsf(im);
  float mx = fmaxf(ar, ai), mn = fminf(ar, ai);
  float r = __fdiv_rn(mn, mx);
  // sqrtf is correctly rounded; __fsqrt_rn is a bare v_sqrt_f32 (1 ulp) on gfx950
  float v = __fmul_rn(mx, sqrtf(__fadd_rn(1.0f, __fmul_rn(r, r))));
  if (mx == 0.0f) v = 0.0f;
  if (isinf(mx)) v = INFINITY;
  if (isnan(re) || isnan(im)) v = NAN;
  return v;
}

// ---------------------------------------------------------------------------
// Threefry-2x32-20 (jax.random, threefry_partitionable)
// ---------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

__host__ __device__ __forceinline__ void threefry2x32(uint32_t k0, uint32_t k1, uint32_t &x0, uint32_t &x1) {
  const uint32_t k2 = k0 ^ k1 ^ 0x1BD11BDAu;
  x0 += k0; x1 += k1;
#define TF_R(r) x0 += x1; x1 = rotl32(x1, r); x1 ^= x0;
  TF_R(13) TF_R(15) TF_R(26) TF_R(6)
  x0 += k1; x1 += k2 + 1u;
  TF_R(17) TF_R(29) TF_R(16) TF_R(24)
  x0 += k2; x1 += k0 + 2u;
  TF_R(13) TF_R(15) TF_R(26) TF_R(6)
  x0 += k0; x1 += k1 + 3u;
  TF_R(17) TF_R(29) TF_R(16) TF_R(24)
  x0 += k1; x1 += k2 + 4u;
  TF_R(13) TF_R(15) TF_R(26) TF_R(6)
  x0 += k2; x1 += k0 + 5u;
#undef TF_R
}

// The 32 random bits jax.random.bits(key, (B,))[s] = x0 ^ x1 of threefry2x32(key, (s >> 32, s)) - the block of every
// Bernoulli draw (sampler.py:74-75).  Same function as threefry2x32 above; the instructions are selected by hand,
// because this block is most of what the first pass issues (measured per instruction class in
// profiles/r03/valu_table.txt):
//   * full rate (2 cycles per wave64 instruction and SIMD): v_add_u32 / v_xor_b32 on vector registers;
//     half rate: v_alignbit_b32 (like every other way to rotate or shift left), v_add3_u32, any SGPR operand;
//   * a round is v_add_u32, v_alignbit_b32, v_xor_b32 - written as asm so that the compiler cannot reassociate
//     the additions of a block (it turned the 32 adds of a block into 40-60 when it could see them);
//   * a key injection `x0 += ka; x1 += kb + i` and the add of the following round are two instructions instead of
//     three: x1 += (kb + i) with the sum formed on the scalar unit, then x0 = x0 + x1 + ka as one v_add3_u32.
// The asm is not volatile: the compiler may still interleave the blocks of different draws.
__device__ __forceinline__ uint32_t threefry_bits32(uint32_t k0, uint32_t k1, unsigned long long s) {
  const uint32_t k2 = k0 ^ k1 ^ 0x1BD11BDAu;
  const uint32_t j1 = k2 + 1u, j2 = k0 + 2u, j3 = k1 + 3u, j4 = k2 + 4u, j5 = k0 + 5u;
  const uint32_t hi = (uint32_t)(s >> 32), lo = (uint32_t)s;
  uint32_t x0, x1;
#define TF_RN(r) "v_alignbit_b32 %1, %1, %1, " #r "\n v_xor_b32 %1, %1, %0\n"  /* rotate left by 32 - r, xor */
#define TF_RA(r) "v_add_u32 %0, %0, %1\n" TF_RN(r)
  asm("v_add_u32 %1, %5, %2\n v_add3_u32 %0, %3, %1, %4\n"  // x1 = lo + k1; x0 = hi + k0, then round 1's add
      TF_RN(19) TF_RA(17) TF_RA(6) TF_RA(26)
      "v_add_u32 %1, %7, %1\n v_add3_u32 %0, %0, %1, %5\n" TF_RN(15) TF_RA(3) TF_RA(16) TF_RA(8)
      "v_add_u32 %1, %8, %1\n v_add3_u32 %0, %0, %1, %6\n" TF_RN(19) TF_RA(17) TF_RA(6) TF_RA(26)
      "v_add_u32 %1, %9, %1\n v_add3_u32 %0, %0, %1, %4\n" TF_RN(15) TF_RA(3) TF_RA(16) TF_RA(8)
      "v_add_u32 %1, %10, %1\n v_add3_u32 %0, %0, %1, %5\n" TF_RN(19) TF_RA(17) TF_RA(6) TF_RA(26)
      "v_add_u32 %0, %6, %0\n v_add_u32 %1, %11, %1\n v_xor_b32 %0, %0, %1\n"
      : "=&v"(x0), "=&v"(x1)
      : "v"(lo), "v"(hi), "s"(k0), "s"(k1), "s"(k2), "s"(j1), "s"(j2), "s"(j3), "s"(j4), "s"(j5));
#undef TF_RA
#undef TF_RN
  return x0;
}

// jax.random.uniform(key, (B,), float32)[s]
__device__ __forceinline__ float uniform01(uint32_t k0, uint32_t k1, unsigned long long s) {
  const uint32_t bits = threefry_bits32(k0, k1, s);
  return __uint_as_float((bits >> 9) | 0x3F800000u) - 1.0f;  // in [0, 1): no clamp needed
}

// Bernoulli thresholds as integers.  u = (bits >> 9) / 2^23 exactly, so for a float threshold t
//   u < t  <=>  (bits >> 9) < T(t),  T = 0 for t <= 0 or NaN, 2^23 for t >= 1, ceil(t * 2^23) otherwise
// (t * 2^23 is exact, its ceiling too): the same bit for every draw, one shift and one compare instead of the
// conversion to float.  The pattern tables of the first pass hold T.
__device__ __forceinline__ uint32_t bernoulli_threshold(float t) {
  if (!(t > 0.0f)) return 0u;
  if (t >= 1.0f) return 1u << 23;
  return (uint32_t)ceilf(t * 8388608.0f);
}

// ---------------------------------------------------------------------------
// GF(2) row parity: popcount(AND) over W 32-bit words (+ constant)
// ---------------------------------------------------------------------------
// (a & b) ^ c in ONE full-rate VALU op (v_bitop3_b32, truth table 0x6A)
__device__ __forceinline__ uint32_t and_xor(uint32_t a, uint32_t b, uint32_t c) {
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x6A);
}

template <int W>
__device__ __forceinline__ uint32_t row_par(cptr r, const uint32_t (&x)[W], uint32_t c) {
  uint32_t t = r[W - 1] & x[W - 1];
#pragma unroll
  for (int w = W - 2; w >= 0; --w) t = and_xor(r[w], x[w], t);
  return (uint32_t)__builtin_popcount(t) + c;  // caller masks bit 0
}

// ---------------------------------------------------------------------------
// evaluate() of one level for one shot (compile/evaluate.py:15-59)
// ---------------------------------------------------------------------------
template <int W>
__device__ __forceinline__ void eval_level(cptr img, cptr lvl, const uint32_t (&x)[W], float &out_re,
                                           float &out_im, int *exact5) {
  const uint32_t G = lvl[L_G];
  const bool approx = (lvl[L_FLAGS] & TSIMK_LFLAG_APPROX) != 0;
  cptr gr = img + lvl[L_GRAPHS];

  int sa = 0, sb = 0, sc = 0, sd = 0, sp = 0;  // running exact sum
  float fre = 0.0f, fim = 0.0f;                // running float sum (approximate branch)

  for (uint32_t g = 0; g < G; ++g, gr += G_WORDS) {
    const uint32_t nA = gr[G_NA], nB = gr[G_NB], nC = gr[G_NC], nD = gr[G_ND];
    cptr row = img + gr[G_ROWS];

    // ---- NodePhases (terms.py:56-73): prod_t (1 + w^(4 par + phase)) ----
    int a = 1, b = 0, c = 0, d = 0, p = 0;
#pragma unroll 4
    for (uint32_t t = 0; t < nA; ++t) {
      const uint32_t ph = row[0];
      const int par = (int)(row_par<W>(row + 1, x, ph >> 2) & 1u);
      row += 1 + W;
      // acc *= (1 + s w^j), j = ph & 3 (wave-uniform), s = par ? -1 : +1
      int ra, rb, rc, rd;
      switch (ph & 3u) {
        case 0: ra = a; rb = b; rc = c; rd = d; break;
        case 1: ra = d; rb = a; rc = b; rd = -c; break;
        case 2: ra = -c; rb = d; rc = a; rd = -b; break;
        default: ra = -b; rb = -c; rc = d; rd = -a; break;
      }
      const int nm = -par;
      a += (ra ^ nm) - nm; b += (rb ^ nm) - nm; c += (rc ^ nm) - nm; d += (rd ^ nm) - nm;
      if (t > 0) reduce1(a, b, c, d, p);  // the scan's first element is taken as is
    }
    canon(a, b, c, d, p);

    // ---- HalfPiPhases (terms.py:94-107) + static phase: exponent of w ----
    uint32_t k = gr[G_PHASE];
#pragma unroll 8
    for (uint32_t t = 0; t < nB; ++t) {
      const uint32_t coeff = row[0];
      k += (row_par<W>(row + 1, x, 0) & 1u) * coeff;
      row += 1 + W;
    }

    // ---- PiProducts (terms.py:125-144): (-1)^(sum psi*phi) ----
    uint32_t e = 0;
#pragma unroll 4
    for (uint32_t t = 0; t < nC; ++t) {
      const uint32_t cc = row[0];
      const uint32_t psi = row_par<W>(row + 1, x, cc & 1u);
      const uint32_t phi = row_par<W>(row + 1 + W, x, cc >> 1);
      e ^= psi & phi;
      row += 1 + 2 * W;
    }
    k += (e & 1u) << 2;

    // ---- PhasePairs (terms.py:164-187): prod_t (1 + w^al + w^be - w^(al+be)) ----
    if (nD) {
      int da = 1, db = 0, dc = 0, dd = 0, dp = 0;
      for (uint32_t t = 0; t < nD; ++t) {
        const uint32_t pa = row_par<W>(row + 4, x, 0) & 1u;
        const uint32_t pb = row_par<W>(row + 4 + W, x, 0) & 1u;
        const uint32_t w0 = pa ? row[1] : row[0];
        const uint32_t w1 = pa ? row[3] : row[2];
        const uint32_t tw = pb ? w1 : w0;  // packed int8 x4 term for (pa, pb)
        row += 4 + 2 * W;
        const int ta = (int)(int8_t)(tw), tb = (int)(int8_t)(tw >> 8), tc = (int)(int8_t)(tw >> 16),
                  td = (int)(int8_t)(tw >> 24);
        if (t == 0) {
          da = ta; db = tb; dc = tc; dd = td;
        } else {
          zmul(da, db, dc, dd, ta, tb, tc, td);
          reduce1(da, db, dc, dd, dp);
        }
      }
      canon(da, db, dc, dd, dp);
      zmul(a, b, c, d, da, db, dc, dd);
      p += dp;
    }

    // ---- static prefactor (evaluate.py:37-50): * w^phase * floatfactor ----
    if (!(gr[G_FLAGS] & TSIMK_GFLAG_FF_IS_ONE))
      zmul(a, b, c, d, (int)gr[G_FFA], (int)gr[G_FFB], (int)gr[G_FFC], (int)gr[G_FFD]);
    // rotate by w^k (k differs per lane)
    {
      const bool k1 = (k & 1u) != 0, k2 = (k & 2u) != 0;
      int t0 = k1 ? d : a, t1 = k1 ? a : b, t2 = k1 ? b : c, t3 = k1 ? -c : d;
      a = k2 ? -t2 : t0; b = k2 ? t3 : t1; c = k2 ? t0 : t2; d = k2 ? -t1 : t3;
      const int nm = -(int)((k >> 2) & 1u);
      a = (a ^ nm) - nm; b = (b ^ nm) - nm; c = (c ^ nm) - nm; d = (d ^ nm) - nm;
    }

    if (!approx) {
      // ---- exact sum over graphs (exact_scalar.py:74-84,173-189) ----
      p += (int)gr[G_POW2];
      if (g == 0) {
        sa = a; sb = b; sc = c; sd = d; sp = p;
      } else {
        const int d1 = max(sp - p, 0), d2 = max(p - sp, 0);
        sa = (int)((unsigned)shl_sat(sa, d1) + (unsigned)shl_sat(a, d2));
        sb = (int)((unsigned)shl_sat(sb, d1) + (unsigned)shl_sat(b, d2));
        sc = (int)((unsigned)shl_sat(sc, d1) + (unsigned)shl_sat(c, d2));
        sd = (int)((unsigned)shl_sat(sd, d1) + (unsigned)shl_sat(d, d2));
        sp = min(sp, p);
        reduce1(sa, sb, sc, sd, sp);
      }
    } else {
      // ---- approximate branch (evaluate.py:56-59), sequential in g ----
      float zr, zi;
      to_complex(a, b, c, d, p, zr, zi);
      const float ar = __uint_as_float(gr[G_APRE]), ai = __uint_as_float(gr[G_APIM]);
      const float tr = __fsub_rn(__fmul_rn(zr, ar), __fmul_rn(zi, ai));
      const float ti = __fadd_rn(__fmul_rn(zr, ai), __fmul_rn(zi, ar));
      const float s = ldexpf(1.0f, (int)gr[G_POW2]);
      fre = __fadd_rn(fre, __fmul_rn(tr, s));
      fim = __fadd_rn(fim, __fmul_rn(ti, s));
    }
  }

  if (!approx) {
    canon(sa, sb, sc, sd, sp);
    to_complex(sa, sb, sc, sd, sp, out_re, out_im);
    if (exact5) { exact5[0] = sa; exact5[1] = sb; exact5[2] = sc; exact5[3] = sd; exact5[4] = sp; }
  } else {
    out_re = fre; out_im = fim;
    if (exact5) { exact5[0] = exact5[1] = exact5[2] = exact5[3] = exact5[4] = 0; }
  }
}

// ---------------------------------------------------------------------------
// evaluate() of one level, "fast exact" formulation.
//
// Same exact Z[w]*2^k value as eval_level for every graph, hence - the canonical form being
// unique - the same summed (a,b,c,d,power) and the same float32 amplitude whenever the
// reference's own int32 arithmetic does not wrap (the packer only selects this path when
// nA <= 30 per graph, for which the NodePhases scan provably cannot wrap).  What changes is how
// the per-graph value is obtained:
//   * NodePhases rows are grouped by phase class j = phase & 3 at pack time.  With m_j the number
//     of class-j rows whose (parity ^ phase>>2) is 1,
//         prod_t (1 + w^(4 par_t + phase_t))
//           = [m_0 == 0] * BASE * (-i)^(m_1+m_2+m_3) * (sqrt2 + 1)^(m_3 - m_1),
//     because (1-i) = (1+i)(-i), (1-w) = (1+w)(-i)(sqrt2-1), (1-w^3) = (1+w^3)(-i)(sqrt2+1).
//     The lane only COUNTS parities (2 VALU ops per row after the parity) and fetches
//     TABLE[m_3 - m_1] = canon(BASE * (sqrt2+1)^(m_3-m_1) * floatfactor * w^static_phase),
//     built exactly at pack time; (-i)^M joins the per-lane phase exponent as 6*M.
//   * HalfPi rows are grouped by coefficient (2, 4, 6): two counters and one XOR accumulator.
// PiProducts, PhasePairs, the per-lane w^k rotation, the graph sum and the float epilogue are
// the same code as the faithful path.
// ---------------------------------------------------------------------------
enum {
  GF_N01 = 0,   // n0 | n1 << 16   NodePhases rows of phase class 0 / 1 (counted)
  GF_N3H = 1,   // n3 | h << 16    class-3 rows (counted) / number of product pairs
  GF_FLAGS = 2, // TSIMK_GFLAG_LAM | _LIN | _D_TABLED
  GF_ND = 3, GF_ROWS = 4, GF_TBL = 5, GF_N1 = 6, GF_TBL2 = 7,
  GF_HWROW = 8, // index of the graph's first row in the level's uniform-stride stream (tsim_kernel_hw.hip.h)
  GF_APRE = 11, GF_APIM = 12
};
#define TSIMK_ZERO_POWER (1 << 20)  // power given to an exactly-zero term: a no-op in the aligned add

template <int W>
__device__ __forceinline__ uint32_t rows_count(cptr &row, uint32_t n, const uint32_t (&x)[W]) {
  // sum over n rows [const, w0..] of ((popcount(row & x) + const) & 1)
  uint32_t cnt = 0;
#pragma unroll 4
  for (uint32_t t = 0; t < n; ++t) {
    cnt += row_par<W>(row + 1, x, row[0]) & 1u;
    row += 1 + W;
  }
  return cnt;
}

template <int W>
__device__ __forceinline__ uint32_t rows_count_nometa(cptr &row, uint32_t n, const uint32_t (&x)[W]) {
  uint32_t cnt = 0;
#pragma unroll 4
  for (uint32_t t = 0; t < n; ++t) {
    cnt += row_par<W>(row, x, 0) & 1u;
    row += W;
  }
  return cnt;
}

// One graph (stabiliser term) of a fast-layout level: its exact value (a, b, c, d) * 2^p - in fixed-frame levels
// already shifted to the level's frame power (p unused) - from the graph record `gr` and the parameter words x.
// `gr` may be wave-uniform (the per-shot kernels: rows arrive through scalar loads) or differ per lane (the
// wave-per-row kernel, tsim_kernel_hw.hip.h: lane = graph): the code is the same, the compiler picks the loads.
// What the rows of one fast-layout graph contribute, as numbers: the counts of minus signs in the counted
// NodePhases classes, the PhasePairs index bits (two per term, first term most significant), the lambda bit and the
// bit e = <lin,x> ^ XOR_s <u_s,x><v_s,x> of the exponent.
struct GraphBits {
  uint32_t m0, m1, m3, dbits, lam, e;
};

// ... and the graph's exact value from them: (a, b, c, d) * 2^p - in fixed-frame levels already shifted to the
// level's frame power (p unused).  The record words it needs are passed in (scalar loads of a wave-uniform record in
// the per-shot kernels, the lane's own record in the wave-per-row kernel).
struct GraphRec {
  uint32_t flags, nD, n1, tbl, tbl2;
};
__device__ __forceinline__ GraphRec graph_rec_of(cptr gr) {
  GraphRec r;
  r.flags = gr[GF_FLAGS]; r.nD = gr[GF_ND]; r.n1 = gr[GF_N1]; r.tbl = gr[GF_TBL]; r.tbl2 = gr[GF_TBL2];
  return r;
}
__device__ __forceinline__ void graph_fast_value(const uint32_t *gimg, const GraphRec &R, bool fixed, const GraphBits &q, int &a, int &b, int &c,
                                                 int &d, int &p) {
  const uint32_t flags = R.flags, nD = R.nD;
  const bool d_tabled = (flags & TSIMK_GFLAG_D_TABLED) != 0;
  uint32_t idx = q.m3 - q.m1 + R.n1;
  int da = 1, db = 0, dc = 0, dd = 0, dp = 0;
  if (d_tabled) {
    idx = (idx << (2u * nD)) | q.dbits;
  } else if (nD) {
    // a separate PhasePairs table (the combined one would be too large)
    const uint32_t *td = gimg + R.tbl2 + 8u * q.dbits;
    const uint4 dv = *reinterpret_cast<const uint4 *>(td);
    da = (int)dv.x; db = (int)dv.y; dc = (int)dv.z; dd = (int)dv.w;
    dp = (int)td[4];
  }
  // entry 0 of every table is the exact zero (a vanished NodePhases product: some 1 + w^4 factor)
  idx = (q.m0 != 0) ? 0u : idx + 1u;
  // per-lane gather of the tabulated term (global memory, L1/L2 resident).  Fixed-frame levels
  // hold four pre-rotated copies per entry (value * i^r), selected by the exponent below.
  const uint32_t *te = gimg + R.tbl + (fixed ? 16u : 8u) * idx;
  // exponent of w: k = 2 <lam,x> + 4 e (k0 is in the table)
  const uint32_t k = ((q.lam & 1u) << 1) + ((q.e & 1u) << 2);
  const uint4 tv = *reinterpret_cast<const uint4 *>(te + (fixed ? (k << 1) : 0u));  // r = k/2, 4 words each
  a = (int)tv.x; b = (int)tv.y; c = (int)tv.z; d = (int)tv.w; p = 0;
  if (!fixed) p = (int)te[4];
  if (!d_tabled && nD) {
    zmul(a, b, c, d, da, db, dc, dd);
    p += dp;
  }
  if (!fixed) {  // rotate by w^k, k in {0, 2, 4, 6} (differs per lane)
    const bool k2 = (k & 2u) != 0;
    const int t0 = k2 ? -c : a, t1 = k2 ? d : b, t2 = k2 ? a : c, t3 = k2 ? -b : d;
    const int nm = -(int)((k >> 2) & 1u);
    a = (t0 ^ nm) - nm; b = (t1 ^ nm) - nm; c = (t2 ^ nm) - nm; d = (t3 ^ nm) - nm;
  }
}

// One graph (stabiliser term) of a fast-layout level from its rows (wave-uniform `gr`: the rows arrive through
// scalar loads) and the lane's parameter words x.
template <int W>
__device__ __forceinline__ void eval_graph_fast(const uint32_t *gimg, cptr img, cptr gr, const uint32_t (&x)[W], bool fixed,
                                                int &a, int &b, int &c, int &d, int &p) {
  const uint32_t n01 = gr[GF_N01], n3h = gr[GF_N3H], flags = gr[GF_FLAGS];
  cptr row = img + gr[GF_ROWS];
  GraphBits q;
  // ---- NodePhases: count minus signs in the phase classes that need an integer count ----
  q.m0 = rows_count<W>(row, n01 & 0xFFFFu, x);
  q.m1 = rows_count<W>(row, n01 >> 16, x);
  q.m3 = rows_count<W>(row, n3h & 0xFFFFu, x);
  // ---- PhasePairs: two index bits per term (into the combined table or the separate one) ----
  const uint32_t nD = gr[GF_ND];
  q.dbits = 0;
  for (uint32_t t = 0; t < nD; ++t) {
    const uint32_t pa = row_par<W>(row, x, 0) & 1u;
    const uint32_t pb = row_par<W>(row + W, x, 0) & 1u;
    q.dbits = (q.dbits << 2) | pa | (pb << 1);
    row += 2 * W;
  }
  // ---- exponent of w: k = 2 <lam,x> + 4 ( <lin,x> ^ XOR_s <u_s,x><v_s,x> )
  q.lam = 0;
  q.e = 0;
  if (flags & TSIMK_GFLAG_LAM) { q.lam = row_par<W>(row, x, 0) & 1u; row += W; }
  if (flags & TSIMK_GFLAG_LIN) { q.e = row_par<W>(row, x, 0); row += W; }
  const uint32_t nH = n3h >> 16;
#pragma unroll 4
  for (uint32_t t = 0; t < nH; ++t) {
    const uint32_t pu = row_par<W>(row, x, 0);
    const uint32_t pv = row_par<W>(row + W, x, 0);
    q.e = and_xor(pu, pv, q.e);
    row += 2 * W;
  }
  graph_fast_value(gimg, graph_rec_of(gr), fixed, q, a, b, c, d, p);
}

// the running sum of a level takes one more graph (evaluate.py:53-59), in graph order
struct LevelSum {
  int sa = 0, sb = 0, sc = 0, sd = 0, sp = TSIMK_ZERO_POWER;
  float fre = 0.0f, fim = 0.0f;
};
__device__ __forceinline__ void level_sum_exact(LevelSum &S, int a, int b, int c, int d, int p) {
  // exactly-zero terms get a huge power, which makes the aligned add below a no-op for them
  if ((a | b | c | d) == 0) p = TSIMK_ZERO_POWER;
  const int d1 = max(S.sp - p, 0), d2 = max(p - S.sp, 0);
  S.sa = (int)((unsigned)shl_sat(S.sa, d1) + (unsigned)shl_sat(a, d2));
  S.sb = (int)((unsigned)shl_sat(S.sb, d1) + (unsigned)shl_sat(b, d2));
  S.sc = (int)((unsigned)shl_sat(S.sc, d1) + (unsigned)shl_sat(c, d2));
  S.sd = (int)((unsigned)shl_sat(S.sd, d1) + (unsigned)shl_sat(d, d2));
  S.sp = min(S.sp, p);
  reduce1(S.sa, S.sb, S.sc, S.sd, S.sp);
}
// the float32 term of the approximate branch: complex64(term) * approx (the table power already contains power2)
__device__ __forceinline__ void level_term_approx(uint32_t apre, uint32_t apim, int a, int b, int c, int d, int p, float &tr, float &ti) {
  float zr, zi;
  to_complex(a, b, c, d, p, zr, zi);
  const float ar = __uint_as_float(apre), ai = __uint_as_float(apim);
  tr = __fsub_rn(__fmul_rn(zr, ar), __fmul_rn(zi, ai));
  ti = __fadd_rn(__fmul_rn(zr, ai), __fmul_rn(zi, ar));
}
__device__ __forceinline__ void level_finish(const LevelSum &S0, cptr lvl, bool approx, bool fixed, float &out_re, float &out_im, int *exact5) {
  if (!approx) {
    int sa = S0.sa, sb = S0.sb, sc = S0.sc, sd = S0.sd, sp = S0.sp;
    if (fixed) sp = (int)lvl[L_FRAME];
    canon(sa, sb, sc, sd, sp);
    if ((sa | sb | sc | sd) == 0) sp = 0;
    to_complex(sa, sb, sc, sd, sp, out_re, out_im);
    if (exact5) { exact5[0] = sa; exact5[1] = sb; exact5[2] = sc; exact5[3] = sd; exact5[4] = sp; }
  } else {
    out_re = S0.fre; out_im = S0.fim;
    if (exact5) { exact5[0] = exact5[1] = exact5[2] = exact5[3] = exact5[4] = 0; }
  }
}

template <int W>
__device__ __forceinline__ void eval_level_fast(const uint32_t *gimg, cptr img, cptr lvl, const uint32_t (&x)[W],
                                                float &out_re, float &out_im, int *exact5) {
  const uint32_t G = lvl[L_G];
  const bool approx = (lvl[L_FLAGS] & TSIMK_LFLAG_APPROX) != 0;
  const bool fixed = (lvl[L_FLAGS] & TSIMK_LFLAG_FIXED) != 0;
  cptr gr = img + lvl[L_GRAPHS];
  LevelSum S;
  for (uint32_t g = 0; g < G; ++g, gr += G_WORDS) {
    int a, b, c, d, p;
    eval_graph_fast<W>(gimg, img, gr, x, fixed, a, b, c, d, p);
    if (fixed) {
      // every table entry of this level is pre-shifted to the level's frame power: plain adds,
      // the pack-time bound guarantees no int32 overflow (see pack_level_fast)
      S.sa += a; S.sb += b; S.sc += c; S.sd += d;
    } else if (!approx) {
      level_sum_exact(S, a, b, c, d, p);
    } else {
      float tr, ti;
      level_term_approx(gr[GF_APRE], gr[GF_APIM], a, b, c, d, p, tr, ti);
      S.fre = __fadd_rn(S.fre, tr);
      S.fim = __fadd_rn(S.fim, ti);
    }
  }
  level_finish(S, lvl, approx, fixed, out_re, out_im, exact5);
}

template <int W, bool FAST>
__device__ __forceinline__ void eval_any(const uint32_t *gimg, cptr img, cptr lvl, const uint32_t (&x)[W],
                                         float &re, float &im, int *exact5) {
  if constexpr (FAST) eval_level_fast<W>(gimg, img, lvl, x, re, im, exact5);
  else eval_level<W>(img, lvl, x, re, im, exact5);
}

// jnp.maximum semantics (NaN propagates)
__device__ __forceinline__ float nanmax(float a, float b) {
  return (isnan(a) || isnan(b)) ? NAN : fmaxf(a, b);
}

// K14: direct outputs f[idx] ^ flip (sampler.py:140-145) from the LDS-staged f row into the (zeroed) LDS output row:
// bit-field runs when the packer emitted them (a read-modify-write of LDS per BIT made this loop the longest
// dependency chain of a block for programs with ~100 direct detectors), else the pair table.
__device__ __forceinline__ void direct_outputs(const SampleArgs &A, cptr img, const uint32_t *lds_f, uint32_t *lds_o, int nthr) {
  if (A.direct_chunks > 0) {
    uint32_t o0 = 0, o1 = 0;
    gather_runs(img + A.direct_prog, (uint32_t)A.direct_chunks, lds_f, lds_o, nthr, o0, o1);
    if (A.WO > 0) {
      lds_o[0] |= o0;
      lds_o[nthr] |= o1;
    }
    return;
  }
  cptr dt = img + A.direct_off;
  for (int j = 0; j < A.n_direct; ++j) {
    const uint32_t s = dt[2 * j], dst = dt[2 * j + 1];
    const uint32_t src = s & 0x7FFFFFFFu;
    const uint32_t bit = ((lds_f[(src >> 5) * nthr] >> (src & 31u)) ^ (s >> 31)) & 1u;
    lds_o[(dst >> 5) * nthr] |= bit << (dst & 31u);
  }
}

// ---------------------------------------------------------------------------
// one component: _sample_component (sampler.py:28-81)
// ---------------------------------------------------------------------------
template <int W, bool FAST>
__device__ __forceinline__ void run_component(const SampleArgs &A, cptr img, cptr comp, const uint32_t *lds_f,
                                              uint32_t *lds_o, int nthr, unsigned long long shot,
                                              bool check_lane, int comp_index) {
  const uint32_t n_out = comp[C_NOUT], F = comp[C_F];
  cptr fsel = img + comp[C_FSEL];
  cptr levels = img + comp[C_LEVELS];
  cptr outpos = img + comp[C_OUTPOS];
  const uint32_t keybase = comp[C_KEYBASE];

  // K1: gather the component's f bits (sampler.py:48) from the LDS-staged row
  uint32_t x[W];
#pragma unroll
  for (int w = 0; w < W; ++w) {
    uint32_t v = 0;
    const int lo = w * 32;
    const int hi = min((int)F, lo + 32);
    for (int j = lo; j < hi; ++j) {
      const uint32_t src = fsel[j];
      v |= ((lds_f[(src >> 5) * nthr] >> (src & 31u)) & 1u) << (j - lo);
    }
    x[w] = v;
  }

  float re, im;
  eval_any<W, FAST>(A.img, img, levels, x, re, im, nullptr);  // normalisation (sampler.py:54)
  float prev = cabs32(re, im);
  float maxdev = 0.0f;

  for (uint32_t i = 0; i < n_out; ++i) {
    cptr lvl = levels + (i + 1) * L_WORDS;
    const uint32_t bitpos = F + i;
    const uint32_t wi = bitpos >> 5, bm = 1u << (bitpos & 31u);
    // trial bit = 1 (sampler.py:65); the check lane also evaluates trial bit = 0 (:66)
    float pr[2] = {0.0f, 0.0f};
    const int npass = check_lane ? 2 : 1;
    for (int pass = 0; pass < npass; ++pass) {
#pragma unroll
      for (int w = 0; w < W; ++w)
        if ((uint32_t)w == wi) x[w] = (pass == 0) ? (x[w] | bm) : (x[w] & ~bm);
      eval_any<W, FAST>(A.img, img, lvl, x, re, im, nullptr);
      const float v = cabs32(re, im);
      if (pass == 0) pr[0] = v; else pr[1] = v;
    }
    const float p1 = pr[0];
    if (check_lane) {
      const float norm = __fdiv_rn(__fadd_rn(pr[1], p1), prev);  // sampler.py:71
      maxdev = nanmax(maxdev, fabsf(__fsub_rn(norm, 1.0f)));      // sampler.py:72
    }
    // sampler.py:74-79
    const float u = uniform01(subkey(A, keybase + i, 0), subkey(A, keybase + i, 1), shot);
    const bool bit = u < __fdiv_rn(p1, prev);
#pragma unroll
    for (int w = 0; w < W; ++w)
      if ((uint32_t)w == wi) x[w] = bit ? (x[w] | bm) : (x[w] & ~bm);
    prev = bit ? p1 : __fsub_rn(prev, p1);
    // K15: scatter to the final column (sampler.py:164-166)
    const uint32_t dst = outpos[i];
    lds_o[(dst >> 5) * nthr] |= (bit ? 1u : 0u) << (dst & 31u);
  }
  if (check_lane && A.norm_dev) A.norm_dev[comp_index] = maxdev;
}

// ---------------------------------------------------------------------------
// sample_program for one batch (sampler.py:117-167), one lane per shot
// ---------------------------------------------------------------------------
extern __shared__ uint32_t tsimk_lds[];

// WMAX = largest per-component word count in the program: the kernel is instantiated per
// WMAX so that narrow programs are not charged the registers of the wide variants.
template <int WMAX, bool FAST>
__global__ void __launch_bounds__(256) k_sample(SampleArgs A) {
  const int nthr = blockDim.x;
  long long slot = (long long)blockIdx.x * nthr + threadIdx.x;
  long long row = slot;
  if (A.row_index) {
    if (A.row_lists > 1) {
      const uint32_t k = blockIdx.x % (uint32_t)A.row_lists;
      slot = (long long)(blockIdx.x / (uint32_t)A.row_lists) * nthr + threadIdx.x + A.row_slot_begin;
      if (slot >= (long long)A.row_count[32u * k]) return;
      row = A.row_index[(size_t)k * A.row_list_cap + slot];
    } else {
      if (slot >= (long long)*A.row_count) return;
      row = A.row_index[slot];
    }
  } else if (slot >= A.B) {
    return;  // no barriers below: every lane owns its LDS columns
  }
  const unsigned long long shot = (unsigned long long)(A.shot_offset + row);
  // normalisation check (sampler.py:66-72): in-batch shot 0, or the first listed survivor
  const bool check_lane = A.no_check ? false
                          : A.check_row ? (row == (long long)*A.check_row)
                          : A.row_index ? (slot == 0) : (shot == 0ull);
  cptr img = (cptr)(uintptr_t)A.img;

  const int WF32 = 2 * A.WF, WO32 = 2 * A.WO;
  uint32_t *lds_f = tsimk_lds + threadIdx.x;                // [WF32][nthr]
  uint32_t *lds_o = tsimk_lds + WF32 * nthr + threadIdx.x;  // [WO32][nthr]

  // stage this shot's packed f row (the only per-shot HBM read)
  stage_f_row(A.f + row * A.WF, A.WF, lds_f, nthr);
  for (int w = 0; w < WO32; ++w) lds_o[w * nthr] = 0u;

  // K14: direct outputs f[idx] ^ flip (sampler.py:140-145)
  direct_outputs(A, img, lds_f, lds_o, nthr);

  // compiled components, in processing order (sampler.py:147-162)
  for (int ci = 0; ci < A.n_comp; ++ci) {
    cptr comp = img + A.comp_off + ci * C_WORDS;
    switch (comp[C_W]) {
#define TSIMK_CASE(WV)                                                                        \
  case WV:                                                                                    \
    if constexpr (WV <= WMAX) run_component<WV, FAST>(A, img, comp, lds_f, lds_o, nthr, shot, check_lane, ci); \
    break;
      TSIMK_CASE(1) TSIMK_CASE(2) TSIMK_CASE(3) TSIMK_CASE(4) TSIMK_CASE(6) TSIMK_CASE(8)
      TSIMK_CASE(12) TSIMK_CASE(16) TSIMK_CASE(24) TSIMK_CASE(32) TSIMK_CASE(48) TSIMK_CASE(64)
#undef TSIMK_CASE
      default: __builtin_trap();  // the packer only emits the widths instantiated here (tsimhost::kWVariants)
    }
  }

  // the only per-shot HBM write: the packed output row
  if (A.out) {  // nullptr: the caller wants the bit_packed rows only
    uint64_t *orow = A.out + row * A.WO;
    for (int w = 0; w < A.WO; ++w)
      orow[w] = (uint64_t)lds_o[(2 * w) * nthr] | ((uint64_t)lds_o[(2 * w + 1) * nthr] << 32);
  }
  store_compact_row(A, row, lds_o, nthr);
}

// ---------------------------------------------------------------------------
// evaluate() seam (compile/evaluate.py:16)
// ---------------------------------------------------------------------------
template <int W, bool FAST>
__global__ void __launch_bounds__(256) k_evaluate(EvalArgs A) {
  const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= A.B) return;
  cptr img = (cptr)(uintptr_t)A.img;
  uint32_t x[W];
#pragma unroll
  for (int w = 0; w < W; ++w) x[w] = A.x[row * W + w];
  float re, im;
  int ex[5];
  eval_any<W, FAST>(A.img, img, img + A.level_off, x, re, im, ex);
  A.re[row] = re;
  A.im[row] = im;
  if (A.abs) A.abs[row] = cabs32(re, im);
  if (A.exact) {
#pragma unroll
    for (int i = 0; i < 5; ++i) A.exact[row * 5 + i] = ex[i];
  }
}

}  // namespace tsimk
