/*
 * oracle.c - ORACLE (plain C): TEST INFRASTRUCTURE ONLY.
 *
 * A scalar CPU restatement of the reference hot path, one shot at a time, used
 *   (1) by tests/ as the checker at sizes the numpy oracle cannot reach, and
 *   (2) by bench.py's `cpu_baseline` leg (kind "port"), timed on the host cores.
 * It is never linked into, imported by, or called from the product path
 * (tsim_amd/), and shares no code with it: it walks the reference's PADDED
 * byte-per-bit arrays slot by slot, uses the general 16-multiply Z[omega]
 * product for every term and the literal one-step-reduce + fix-point loops.
 * Rows are bit-packed once at load time into 64-bit words purely as a CPU
 * speed-up of the parity (the float32 GEMM "% 2" of utils/linalg.py:81-102
 * computes the same parity bit).
 *
 * Reference anchors (under /root/reference/src/tsim/):
 *   sampler.py:28-81     _sample_component      -> orc_sample_component
 *   sampler.py:117-167   sample_program         -> orc_sample_program
 *   compile/evaluate.py:15-59  evaluate         -> eval_level
 *   compile/terms.py:56-73,94-107,125-144,164-187  the four families
 *   core/exact_scalar.py:19-49,52-84,98-137,218-222  exact scalar arithmetic
 *   jax.random threefry2x32 (partitionable)      -> threefry2x32 / uniform01
 *
 * PARITY PINNED through oracle_np.py, which reproduces the reference's seeded
 * KATs; tests/test_oracle_c.py requires this file to agree with oracle_np.py
 * bit for bit (samples, exact integers, float32 amplitudes) on seeded programs
 * and to reproduce the same KATs directly.
 * PARITY UNPINNED: the same float32/XLA ulp-level caveats as oracle_np.py.
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp -ffp-contract=off -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
  int32_t num_graphs, n_params, ta, tb, tc, td;
  const uint8_t *a_phases, *a_params;
  const int32_t *a_counts;
  const uint8_t *b_coeffs, *b_params;
  const uint8_t *c_psi_const, *c_psi_params, *c_phi_const, *c_phi_params;
  const uint8_t *d_alpha, *d_alpha_params, *d_beta, *d_beta_params;
  const int32_t *d_counts;
  const uint8_t *phase_indices;
  const int32_t *floatfactor;
  const int32_t *power2;
  const float *approx; /* [G,2] */
  int32_t has_approx;
} orc_level_desc;

typedef struct {
  orc_level_desc d; /* scalar fields + small arrays are deep-copied below */
  int W;            /* 64-bit words per packed row */
  uint64_t *a_rows, *b_rows, *psi_rows, *phi_rows, *al_rows, *be_rows;
  uint8_t *a_phases, *b_coeffs, *psi_const, *phi_const, *d_alpha, *d_beta, *phase_indices;
  int32_t *a_counts, *d_counts, *floatfactor, *power2;
  float *approx;
} orc_level;

typedef struct {
  int n_out, F, n_levels;
  int32_t *output_indices, *f_selection;
  orc_level *levels;
  int levels_added;
} orc_component;

typedef struct {
  int num_outputs, n_direct, n_comp, cap_comp;
  int32_t *direct_f, *output_order, *reindex;
  uint8_t *direct_flips;
  orc_component *comps;
  int identity_order;
} orc_program;

/* ---------------------------------------------------------------- utils */
static void *dup_mem(const void *src, size_t n) {
  void *p = malloc(n ? n : 1);
  if (n && src) memcpy(p, src, n);
  else if (n) memset(p, 0, n);
  return p;
}

static uint64_t *pack_rows(const uint8_t *bits, size_t nrows, int P, int W) {
  uint64_t *out = (uint64_t *)calloc(nrows * (size_t)W + 1, 8);
  for (size_t r = 0; r < nrows; ++r)
    for (int i = 0; i < P; ++i)
      if (bits[r * (size_t)P + i] & 1) out[r * W + (i >> 6)] |= 1ull << (i & 63);
  return out;
}

/* ------------------------------------------------------- program builder */
orc_program *orc_program_new(int32_t num_outputs, int32_t n_direct, const int32_t *direct_f,
                             const uint8_t *direct_flips, const int32_t *output_order) {
  orc_program *p = (orc_program *)calloc(1, sizeof *p);
  p->num_outputs = num_outputs;
  p->n_direct = n_direct;
  p->direct_f = (int32_t *)dup_mem(direct_f, 4 * (size_t)n_direct);
  p->direct_flips = (uint8_t *)dup_mem(direct_flips, (size_t)n_direct);
  p->output_order = (int32_t *)dup_mem(output_order, 4 * (size_t)num_outputs);
  /* output_reindex = argsort(output_order) (pipeline.py:90-92) */
  p->reindex = (int32_t *)malloc(4 * (size_t)(num_outputs ? num_outputs : 1));
  p->identity_order = 1;
  for (int i = 0; i < num_outputs; ++i) {
    p->reindex[output_order[i]] = i;
    if (output_order[i] != i) p->identity_order = 0;
  }
  return p;
}

int orc_program_add_component(orc_program *p, int32_t n_out, const int32_t *output_indices, int32_t F,
                              const int32_t *f_selection, int32_t n_levels) {
  if (p->n_comp == p->cap_comp) {
    p->cap_comp = p->cap_comp ? 2 * p->cap_comp : 4;
    p->comps = (orc_component *)realloc(p->comps, sizeof(orc_component) * (size_t)p->cap_comp);
  }
  orc_component *c = &p->comps[p->n_comp];
  memset(c, 0, sizeof *c);
  c->n_out = n_out;
  c->F = F;
  c->n_levels = n_levels;
  c->output_indices = (int32_t *)dup_mem(output_indices, 4 * (size_t)n_out);
  c->f_selection = (int32_t *)dup_mem(f_selection, 4 * (size_t)F);
  c->levels = (orc_level *)calloc((size_t)n_levels, sizeof(orc_level));
  return p->n_comp++;
}

int orc_program_add_level(orc_program *p, int32_t comp, const orc_level_desc *d) {
  orc_component *c = &p->comps[comp];
  if (c->levels_added >= c->n_levels) return -1;
  orc_level *L = &c->levels[c->levels_added++];
  L->d = *d;
  const size_t G = (size_t)d->num_graphs;
  const int P = d->n_params;
  L->W = (P + 63) / 64;
  if (L->W == 0) L->W = 1;
  L->a_rows = pack_rows(d->a_params, G * d->ta, P, L->W);
  L->b_rows = pack_rows(d->b_params, G * d->tb, P, L->W);
  L->psi_rows = pack_rows(d->c_psi_params, G * d->tc, P, L->W);
  L->phi_rows = pack_rows(d->c_phi_params, G * d->tc, P, L->W);
  L->al_rows = pack_rows(d->d_alpha_params, G * d->td, P, L->W);
  L->be_rows = pack_rows(d->d_beta_params, G * d->td, P, L->W);
  L->a_phases = (uint8_t *)dup_mem(d->a_phases, G * d->ta);
  L->b_coeffs = (uint8_t *)dup_mem(d->b_coeffs, G * d->tb);
  L->psi_const = (uint8_t *)dup_mem(d->c_psi_const, G * d->tc);
  L->phi_const = (uint8_t *)dup_mem(d->c_phi_const, G * d->tc);
  L->d_alpha = (uint8_t *)dup_mem(d->d_alpha, G * d->td);
  L->d_beta = (uint8_t *)dup_mem(d->d_beta, G * d->td);
  L->phase_indices = (uint8_t *)dup_mem(d->phase_indices, G);
  L->a_counts = (int32_t *)dup_mem(d->ta ? d->a_counts : NULL, 4 * G);
  L->d_counts = (int32_t *)dup_mem(d->td ? d->d_counts : NULL, 4 * G);
  L->floatfactor = (int32_t *)dup_mem(d->floatfactor, 16 * G);
  L->power2 = (int32_t *)dup_mem(d->power2, 4 * G);
  L->approx = (float *)dup_mem(d->approx, 8 * G);
  if (!d->approx)
    for (size_t g = 0; g < G; ++g) L->approx[2 * g] = 1.0f;
  return 0;
}

void orc_program_free(orc_program *p) {
  if (!p) return;
  for (int ci = 0; ci < p->n_comp; ++ci) {
    orc_component *c = &p->comps[ci];
    for (int k = 0; k < c->levels_added; ++k) {
      orc_level *L = &c->levels[k];
      free(L->a_rows); free(L->b_rows); free(L->psi_rows); free(L->phi_rows); free(L->al_rows); free(L->be_rows);
      free(L->a_phases); free(L->b_coeffs); free(L->psi_const); free(L->phi_const); free(L->d_alpha);
      free(L->d_beta); free(L->phase_indices); free(L->a_counts); free(L->d_counts); free(L->floatfactor);
      free(L->power2); free(L->approx);
    }
    free(c->levels); free(c->output_indices); free(c->f_selection);
  }
  free(p->comps); free(p->direct_f); free(p->direct_flips); free(p->output_order); free(p->reindex);
  free(p);
}

/* ------------------------------------------------ exact scalar arithmetic */
typedef struct { int32_t c[4]; int32_t p; } es_t;

static _Thread_local int g_overflow; /* set when an int32 operation of the reference would wrap */

static inline int32_t wrap32(__int128 v) {
  if (v > INT32_MAX || v < INT32_MIN) g_overflow = 1;
  return (int32_t)(uint32_t)(uint64_t)v;
}

/* exact_scalar.py:19-39 */
static inline void scalar_mul(const int32_t *x, const int32_t *y, int32_t *o) {
  const __int128 a1 = x[0], b1 = x[1], c1 = x[2], d1 = x[3], a2 = y[0], b2 = y[1], c2 = y[2], d2 = y[3];
  int32_t A = wrap32(a1 * a2 + b1 * d2 - c1 * c2 + d1 * b2);
  int32_t B = wrap32(a1 * b2 + b1 * a2 + c1 * d2 + d1 * c2);
  int32_t C = wrap32(a1 * c2 + b1 * b2 + c1 * a2 - d1 * d2);
  int32_t D = wrap32(a1 * d2 - b1 * c2 - c1 * b2 + d1 * a2);
  o[0] = A; o[1] = B; o[2] = C; o[3] = D;
}

/* exact_scalar.py:42-49; returns 1 if it reduced.  Python % and // on negative even ints:
 * x % 2 == 0 iff low bit clear; x // 2 is the arithmetic shift. */
static inline int reduce_step(es_t *s) {
  int all_even = 1, any_nz = 0;
  for (int i = 0; i < 4; ++i) { all_even &= ((s->c[i] & 1) == 0); any_nz |= (s->c[i] != 0); }
  if (all_even && any_nz) {
    for (int i = 0; i < 4; ++i) s->c[i] = s->c[i] >> 1;
    s->p = wrap32((__int128)s->p + 1);
    return 1;
  }
  return 0;
}

/* exact_scalar.py:52-71 */
static inline void mul_with_power(es_t *acc, const es_t *y) {
  int32_t o[4];
  scalar_mul(acc->c, y->c, o);
  memcpy(acc->c, o, sizeof o);
  acc->p = wrap32((__int128)acc->p + y->p);
  reduce_step(acc);
}

/* jnp.left_shift(1, s) in int32 with XLA semantics (s >= 32 -> 0; s == 31 -> INT32_MIN) */
static inline int32_t shl_one(int64_t s) {
  if (s >= 32) return 0;
  return (int32_t)(1u << s);
}

/* exact_scalar.py:74-84 */
static inline void add_with_power(es_t *acc, const es_t *y) {
  const int64_t d1 = (int64_t)acc->p - y->p, d2 = (int64_t)y->p - acc->p;
  const int64_t g1 = d1 > 0 ? d1 : 0, g2 = d2 > 0 ? d2 : 0;
  const int32_t s1 = shl_one(g1), s2 = shl_one(g2);
  if (g1 >= 31 || g2 >= 31) {
    /* the true scale 2^gap does not fit int32: flag unless the scaled operand is zero */
    const int32_t *z = g1 >= 31 ? acc->c : y->c;
    if (z[0] | z[1] | z[2] | z[3]) g_overflow = 1;
  }
  for (int i = 0; i < 4; ++i) {
    /* wrap-around multiply/add exactly as int32 XLA ops */
    uint32_t v = (uint32_t)acc->c[i] * (uint32_t)s1 + (uint32_t)y->c[i] * (uint32_t)s2;
    __int128 t = (__int128)acc->c[i] * ((__int128)1 << (g1 < 100 ? g1 : 100)) +
                 (__int128)y->c[i] * ((__int128)1 << (g2 < 100 ? g2 : 100));
    if (g1 < 31 && g2 < 31 && (t > INT32_MAX || t < INT32_MIN)) g_overflow = 1;
    acc->c[i] = (int32_t)v;
  }
  acc->p = acc->p < y->p ? acc->p : y->p;
  reduce_step(acc);
}

/* exact_scalar.py:119-136 */
static inline void fixpoint(es_t *s) { while (reduce_step(s)) {} }

static const float E4 = 0.70710677f; /* exp(+-i pi/4) in complex64 (exact_scalar.py:15-16) */

/* exact_scalar.py:87-89,218-222 - float32, one rounding per operation */
static inline void to_complex(const es_t *s, float *re, float *im) {
  const float a = (float)s->c[0], b = (float)s->c[1], c = (float)s->c[2], d = (float)s->c[3];
  volatile float t1 = b * E4, t3 = d * E4, u1 = b * E4, u3 = d * (-E4);
  volatile float r = a + t1; r = r + t3;
  volatile float i = u1 + c; i = i + u3;
  const float sc = ldexpf(1.0f, s->p);
  *re = r * sc;
  *im = i * sc;
}

/* jnp.abs(complex64) */
static inline float cabs32(float re, float im) {
  const float ar = fabsf(re), ai = fabsf(im);
  if (isnan(re) || isnan(im)) return NAN;
  const float mx = ar > ai ? ar : ai, mn = ar > ai ? ai : ar;
  if (isinf(mx)) return INFINITY;
  if (mx == 0.0f) return 0.0f;
  volatile float r = mn / mx;
  volatile float q = r * r;
  volatile float w = 1.0f + q;
  return mx * sqrtf(w);
}

static const int32_t UNIT[8][4] = {{1, 0, 0, 0}, {0, 1, 0, 0},  {0, 0, 1, 0},  {0, 0, 0, -1},
                                   {-1, 0, 0, 0}, {0, -1, 0, 0}, {0, 0, -1, 0}, {0, 0, 0, 1}};

static inline int parity(const uint64_t *row, const uint64_t *x, int W) {
  uint64_t t = 0;
  for (int w = 0; w < W; ++w) t ^= row[w] & x[w];
  return __builtin_parityll(t);
}

/* compile/evaluate.py:15-59 for one shot; x holds the level's n_params bits */
static void eval_level(const orc_level *L, const uint64_t *x, float *re, float *im, es_t *exact) {
  const int G = L->d.num_graphs, W = L->W;
  const int TA = L->d.ta, TB = L->d.tb, TC = L->d.tc, TD = L->d.td;
  es_t sum = {{0, 0, 0, 0}, 0};
  float fre = 0.0f, fim = 0.0f;
  if (exact) memset(exact, 0, sizeof *exact);
  if (G == 0) { *re = 0.0f; *im = 0.0f; return; } /* evaluate.py:34-35 */
  for (int g = 0; g < G; ++g) {
    /* NodePhases (terms.py:56-73): masked product over ALL padded slots */
    es_t A = {{1, 0, 0, 0}, 0};
    for (int t = 0; t < TA; ++t) {
      es_t term = {{1, 0, 0, 0}, 0};
      if (t < L->a_counts[g]) {
        const int par = parity(L->a_rows + ((size_t)g * TA + t) * W, x, W);
        const int idx = (4 * par + L->a_phases[(size_t)g * TA + t]) % 8;
        memcpy(term.c, UNIT[idx], sizeof term.c);
        term.c[0] += 1;
      }
      if (t == 0) A = term; else mul_with_power(&A, &term);
    }
    if (TA > 0) fixpoint(&A);
    /* HalfPiPhases (terms.py:94-107) */
    int k = 0;
    for (int t = 0; t < TB; ++t)
      k += (parity(L->b_rows + ((size_t)g * TB + t) * W, x, W) * L->b_coeffs[(size_t)g * TB + t]) % 8;
    k %= 8;
    /* PiProducts (terms.py:125-144) */
    int e = 0;
    for (int t = 0; t < TC; ++t) {
      const int psi = (L->psi_const[(size_t)g * TC + t] + parity(L->psi_rows + ((size_t)g * TC + t) * W, x, W)) % 2;
      const int phi = (L->phi_const[(size_t)g * TC + t] + parity(L->phi_rows + ((size_t)g * TC + t) * W, x, W)) % 2;
      e += (psi * phi) % 2;
    }
    e %= 2;
    /* PhasePairs (terms.py:164-187) */
    es_t D = {{1, 0, 0, 0}, 0};
    for (int t = 0; t < TD; ++t) {
      es_t term = {{1, 0, 0, 0}, 0};
      if (t < L->d_counts[g]) {
        const int ra = parity(L->al_rows + ((size_t)g * TD + t) * W, x, W);
        const int rb = parity(L->be_rows + ((size_t)g * TD + t) * W, x, W);
        const int al = (L->d_alpha[(size_t)g * TD + t] + 4 * ra) % 8;
        const int be = (L->d_beta[(size_t)g * TD + t] + 4 * rb) % 8;
        const int ga = (al + be) % 8;
        for (int i = 0; i < 4; ++i) term.c[i] = (i == 0) + UNIT[al][i] + UNIT[be][i] - UNIT[ga][i];
      }
      if (t == 0) D = term; else mul_with_power(&D, &term);
    }
    if (TD > 0) fixpoint(&D);
    /* product A * B * C * D * static * floatfactor, no reduction (evaluate.py:40-50) */
    es_t tot = A;
    int32_t o[4], fac[4];
    memcpy(fac, UNIT[k], sizeof fac);
    scalar_mul(tot.c, fac, o); memcpy(tot.c, o, sizeof o);
    fac[0] = 1 - 2 * e; fac[1] = fac[2] = fac[3] = 0;
    scalar_mul(tot.c, fac, o); memcpy(tot.c, o, sizeof o);
    scalar_mul(tot.c, D.c, o); memcpy(tot.c, o, sizeof o);
    tot.p = wrap32((__int128)tot.p + D.p);
    memcpy(fac, UNIT[L->phase_indices[g] % 8], sizeof fac);
    scalar_mul(tot.c, fac, o); memcpy(tot.c, o, sizeof o);
    scalar_mul(tot.c, L->floatfactor + 4 * (size_t)g, o); memcpy(tot.c, o, sizeof o);
    if (!L->d.has_approx) {
      tot.p = wrap32((__int128)tot.p + L->power2[g]); /* evaluate.py:53 */
      if (g == 0) sum = tot; else add_with_power(&sum, &tot); /* exact_scalar.py:173-189 */
    } else {
      /* evaluate.py:56-59, sequential in g */
      float zr, zi;
      to_complex(&tot, &zr, &zi);
      const float ar = L->approx[2 * (size_t)g], ai = L->approx[2 * (size_t)g + 1];
      volatile float m1 = zr * ar, m2 = zi * ai, m3 = zr * ai, m4 = zi * ar;
      volatile float tr = m1 - m2, ti = m3 + m4;
      const float sc = ldexpf(1.0f, L->power2[g]);
      volatile float pr = tr * sc, pi = ti * sc;
      fre = fre + pr;
      fim = fim + pi;
    }
  }
  if (!L->d.has_approx) {
    fixpoint(&sum);
    to_complex(&sum, re, im);
    if (exact) *exact = sum;
  } else {
    *re = fre;
    *im = fim;
  }
}

/* ------------------------------------------------------------- threefry */
static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

static void threefry2x32(uint32_t k0, uint32_t k1, uint32_t *x0, uint32_t *x1) {
  static const int R[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
  const uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
  uint32_t a = *x0 + ks[0], b = *x1 + ks[1];
  for (int blk = 0; blk < 5; ++blk) {
    for (int i = 0; i < 4; ++i) { a += b; b = rotl32(b, R[blk & 1][i]); b ^= a; }
    a += ks[(blk + 1) % 3];
    b += ks[(blk + 2) % 3] + (uint32_t)(blk + 1);
  }
  *x0 = a; *x1 = b;
}

static inline float uniform01(uint32_t k0, uint32_t k1, uint64_t s) {
  uint32_t a = (uint32_t)(s >> 32), b = (uint32_t)s;
  threefry2x32(k0, k1, &a, &b);
  const uint32_t bits = ((a ^ b) >> 9) | 0x3F800000u;
  float f;
  memcpy(&f, &bits, 4);
  f -= 1.0f;
  return f > 0.0f ? f : 0.0f;
}

/* ------------------------------------------------------ the sampling loop */
#define ORC_MAXW 64

/* sampler.py:117-167.  out: uint8 [B, num_outputs]; devs: float [n_comp] (may be NULL);
 * returns 1 if any int32 operation of the reference would have wrapped, else 0. */
int orc_sample_program(const orc_program *p, const uint8_t *f, int64_t B, int32_t num_f, uint32_t key_hi,
                       uint32_t key_lo, int64_t shot_offset, uint8_t *out, float *devs, int32_t nthreads) {
  const int n_out_total = p->num_outputs;
  if (n_out_total == 0 || B == 0) return 0;
  /* per-output subkeys: key, subkey = split(key) (sampler.py:74), threaded through components */
  int total = 0;
  for (int ci = 0; ci < p->n_comp; ++ci) total += p->comps[ci].n_out;
  uint32_t *sub = (uint32_t *)malloc(8 * (size_t)(total ? total : 1));
  {
    uint32_t k0 = key_hi, k1 = key_lo;
    for (int i = 0; i < total; ++i) {
      uint32_t a0 = 0, a1 = 0, b0 = 0, b1 = 1;
      threefry2x32(k0, k1, &a0, &a1);
      threefry2x32(k0, k1, &b0, &b1);
      sub[2 * i] = b0; sub[2 * i + 1] = b1;
      k0 = a0; k1 = a1;
    }
  }
  if (devs) for (int ci = 0; ci < p->n_comp; ++ci) devs[ci] = 0.0f;
  int overflow = 0;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(static) reduction(| : overflow)
  for (int64_t s = 0; s < B; ++s) {
    const uint8_t *frow = f + (size_t)s * num_f;
    uint8_t cat[n_out_total]; /* concatenated, pre-reindex (VLA: released every iteration) */
    g_overflow = 0;
    int pos = 0;
    for (int j = 0; j < p->n_direct; ++j) /* sampler.py:140-145 */
      cat[pos++] = (uint8_t)((frow[p->direct_f[j]] != 0) ^ (p->direct_flips[j] != 0));
    int kb = 0;
    for (int ci = 0; ci < p->n_comp; ++ci) {
      const orc_component *c = &p->comps[ci];
      uint64_t x[ORC_MAXW];
      memset(x, 0, sizeof x);
      for (int j = 0; j < c->F; ++j) /* sampler.py:48 */
        if (frow[c->f_selection[j]]) x[j >> 6] |= 1ull << (j & 63);
      float re, im;
      eval_level(&c->levels[0], x, &re, &im, NULL); /* sampler.py:54 */
      float prev = cabs32(re, im);
      float maxdev = 0.0f;
      const int is_check = (shot_offset + s == 0);
      for (int i = 0; i < c->n_out; ++i) {
        const int bit = c->F + i;
        x[bit >> 6] |= 1ull << (bit & 63); /* trial bit 1 (sampler.py:65) */
        eval_level(&c->levels[i + 1], x, &re, &im, NULL);
        const float p1 = cabs32(re, im);
        if (is_check) { /* sampler.py:66-72 */
          x[bit >> 6] &= ~(1ull << (bit & 63));
          eval_level(&c->levels[i + 1], x, &re, &im, NULL);
          const float p0 = cabs32(re, im);
          x[bit >> 6] |= 1ull << (bit & 63);
          volatile float sm = p0 + p1;
          volatile float norm = sm / prev;
          volatile float dv = fabsf(norm - 1.0f);
          maxdev = (isnan(maxdev) || isnan(dv)) ? NAN : (dv > maxdev ? dv : maxdev);
        }
        const float u = uniform01(sub[2 * (kb + i)], sub[2 * (kb + i) + 1], (uint64_t)(shot_offset + s));
        volatile float pr = p1 / prev;
        const int b = u < pr; /* NaN -> 0 */
        if (!b) x[bit >> 6] &= ~(1ull << (bit & 63));
        volatile float dif = prev - p1;
        prev = b ? p1 : dif; /* sampler.py:79 */
        cat[pos++] = (uint8_t)b;
      }
      kb += c->n_out;
      if (is_check && devs) devs[ci] = maxdev;
    }
    uint8_t *orow = out + (size_t)s * n_out_total;
    if (p->identity_order) memcpy(orow, cat, (size_t)n_out_total);
    else for (int i = 0; i < n_out_total; ++i) orow[i] = cat[p->reindex[i]]; /* sampler.py:164-166 */
    overflow |= g_overflow;
  }
  free(sub);
  return overflow;
}

/* evaluate() seam: params uint8 [B, n_params] -> re/im float [B], exact int32 [B,5] (may be NULL) */
int orc_evaluate(const orc_program *p, int32_t comp, int32_t level, const uint8_t *params, int64_t B,
                 float *re, float *im, int32_t *exact5) {
  const orc_level *L = &p->comps[comp].levels[level];
  const int P = L->d.n_params;
  g_overflow = 0;
  for (int64_t s = 0; s < B; ++s) {
    uint64_t x[ORC_MAXW];
    memset(x, 0, sizeof x);
    for (int i = 0; i < P; ++i)
      if (params[(size_t)s * P + i]) x[i >> 6] |= 1ull << (i & 63);
    es_t ex;
    eval_level(L, x, &re[s], &im[s], &ex);
    if (exact5) {
      for (int i = 0; i < 4; ++i) exact5[5 * s + i] = ex.c[i];
      exact5[5 * s + 4] = ex.p;
    }
  }
  return g_overflow;
}

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
