// tsim_pack.hip - the packer: reference-layout level descriptions -> device image pieces (host only).
//
//   pack_level       faithful layout: the reference's rows, one meta word each, exact no-ops dropped
//   pack_level_fast  exact-value formulation: NodePhases class counting, the phase exponent as a
//                    Dickson-reduced GF(2) quadratic form, pack-time Z[w] term tables
//   emit_level4      4-bit chunk tables / sparse-f column tables of the LDS kernel (k_sample4)
//   emit_gather_program  bit-field runs of the first pass (k_sample_lw)
#include "tsim_internal.hip.h"

using namespace tsimk;

namespace tsimhost {

// pack one byte-per-bit row into W 32-bit words appended to `dst`; returns true if any bit set
// eight 0/1 bytes -> eight bits (byte k -> bit k); reference bit-matrices hold 0/1 (compile.py:40-238)
static inline uint32_t bits_of_8_bytes(const uint8_t *b) {
  uint64_t x;
  memcpy(&x, b, 8);
  return (uint32_t)(((x & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56);
}
static bool pack_row(std::vector<uint32_t> &dst, const uint8_t *bits, int P, int W) {
  uint32_t any = 0;
  const size_t base = dst.size();
  dst.resize(base + W, 0u);
  uint32_t *o = &dst[base];
  int i = 0;
  for (; i + 8 <= P; i += 8) {
    const uint32_t b = bits_of_8_bytes(bits + i);
    o[i >> 5] |= b << (i & 31);
    any |= b;
  }
  for (; i < P; ++i)
    if (bits[i] & 1) {
      o[i >> 5] |= 1u << (i & 31);
      any = 1;
    }
  return any != 0;
}

static const int8_t kUnit[8][4] = {{1, 0, 0, 0}, {0, 1, 0, 0},  {0, 0, 1, 0},  {0, 0, 0, -1},
                                   {-1, 0, 0, 0}, {0, -1, 0, 0}, {0, 0, -1, 0}, {0, 0, 0, 1}};

// Build graph records + rows for a level with W words per row.
void pack_level(HostLevel &h, int W) {
  const int G = h.G, P = h.P;
  const tsim_level_desc &d = h.d;
  h.graph_rec.assign((size_t)G * G_WORDS, 0u);
  h.rows.clear();
  h.n_rows = 0;
  std::vector<uint32_t> tmp;
  for (int g = 0; g < G; ++g) {
    uint32_t *rec = &h.graph_rec[(size_t)g * G_WORDS];
    rec[G_ROWS] = (uint32_t)h.rows.size();
    // A: the first counts[g] slots are real (terms.py:70-71)
    const int nA = d.ta ? h.i32[0][g] : 0;
    for (int t = 0; t < nA; ++t) {
      h.rows.push_back((uint32_t)(h.u8[0][(size_t)g * d.ta + t] & 7));
      pack_row(h.rows, &h.u8[1][((size_t)g * d.ta + t) * P], P, W);
    }
    rec[G_NA] = (uint32_t)nA;
    // B: zero coefficient or empty parity contributes exponent 0 (terms.py:104-106) -> dropped
    int nB = 0;
    for (int t = 0; t < d.tb; ++t) {
      const uint32_t coeff = h.u8[2][(size_t)g * d.tb + t] & 7u;  // (rowsum*coeff) % 8
      if (!coeff) continue;
      tmp.clear();
      if (!pack_row(tmp, &h.u8[3][((size_t)g * d.tb + t) * P], P, W)) continue;
      h.rows.push_back(coeff);
      h.rows.insert(h.rows.end(), tmp.begin(), tmp.end());
      ++nB;
    }
    rec[G_NB] = (uint32_t)nB;
    // C: a slot whose psi or phi is identically 0 contributes (-1)^0 (terms.py:136-141) -> dropped
    int nC = 0;
    for (int t = 0; t < d.tc; ++t) {
      const uint32_t pc = h.u8[4][(size_t)g * d.tc + t] & 1u, qc = h.u8[6][(size_t)g * d.tc + t] & 1u;
      std::vector<uint32_t> r1, r2;
      const bool any1 = pack_row(r1, &h.u8[5][((size_t)g * d.tc + t) * P], P, W);
      const bool any2 = pack_row(r2, &h.u8[7][((size_t)g * d.tc + t) * P], P, W);
      if ((!any1 && !pc) || (!any2 && !qc)) continue;
      h.rows.push_back(pc | (qc << 1));
      h.rows.insert(h.rows.end(), r1.begin(), r1.end());
      h.rows.insert(h.rows.end(), r2.begin(), r2.end());
      ++nC;
    }
    rec[G_NC] = (uint32_t)nC;
    // D: first counts[g] slots are real; the four possible term values are tabulated
    const int nD = d.td ? h.i32[1][g] : 0;
    for (int t = 0; t < nD; ++t) {
      const int al = h.u8[8][(size_t)g * d.td + t] & 7, be = h.u8[10][(size_t)g * d.td + t] & 7;
      for (int idx = 0; idx < 4; ++idx) {
        const int pa = idx & 1, pb = idx >> 1;
        const int a1 = (al + 4 * pa) & 7, b1 = (be + 4 * pb) & 7, g1 = (a1 + b1) & 7;
        uint32_t w = 0;
        for (int j = 0; j < 4; ++j) {
          const int v = (j == 0 ? 1 : 0) + kUnit[a1][j] + kUnit[b1][j] - kUnit[g1][j];
          w |= (uint32_t)(uint8_t)(int8_t)v << (8 * j);
        }
        h.rows.push_back(w);
      }
      pack_row(h.rows, &h.u8[9][((size_t)g * d.td + t) * P], P, W);
      pack_row(h.rows, &h.u8[11][((size_t)g * d.td + t) * P], P, W);
    }
    rec[G_ND] = (uint32_t)nD;
    h.n_rows += nA + nB + 2 * nC + 2 * nD;
    rec[G_PHASE] = h.u8[12][g] & 7u;
    const int32_t *ff = &h.i32[2][(size_t)g * 4];
    rec[G_FFA] = (uint32_t)ff[0]; rec[G_FFB] = (uint32_t)ff[1];
    rec[G_FFC] = (uint32_t)ff[2]; rec[G_FFD] = (uint32_t)ff[3];
    rec[G_POW2] = (uint32_t)h.i32[3][g];
    memcpy(&rec[G_APRE], &h.approx_v[2 * (size_t)g], 4);
    memcpy(&rec[G_APIM], &h.approx_v[2 * (size_t)g + 1], 4);
    rec[G_FLAGS] = (ff[0] == 1 && ff[1] == 0 && ff[2] == 0 && ff[3] == 0) ? TSIMK_GFLAG_FF_IS_ONE : 0u;
  }
}

// exact element of Z[w] * 2^p on basis (1, w, i, conj w), int64 coefficients, kept canonical
struct ZW {
  long long c[4];
  int p;
};

static void zw_canon(ZW &z) {
  if (!(z.c[0] | z.c[1] | z.c[2] | z.c[3])) return;
  while (!((z.c[0] | z.c[1] | z.c[2] | z.c[3]) & 1)) {
    for (auto &v : z.c) v >>= 1;
    ++z.p;
  }
}

static void zw_mul(ZW &x, const long long y[4]) {  // exact_scalar.py:19-39, then canonicalise
  const long long a1 = x.c[0], b1 = x.c[1], c1 = x.c[2], d1 = x.c[3];
  const long long a2 = y[0], b2 = y[1], c2 = y[2], d2 = y[3];
  x.c[0] = a1 * a2 + b1 * d2 - c1 * c2 + d1 * b2;
  x.c[1] = a1 * b2 + b1 * a2 + c1 * d2 + d1 * c2;
  x.c[2] = a1 * c2 + b1 * b2 + c1 * a2 - d1 * d2;
  x.c[3] = a1 * d2 - b1 * c2 - c1 * b2 + d1 * a2;
  zw_canon(x);
}

static void unit_plus_one(int k, long long out[4]) {  // 1 + w^k
  for (int j = 0; j < 4; ++j) out[j] = kUnit[k & 7][j];
  out[0] += 1;
}

// ---- GF(2) algebra used by the fast packer -------------------------------------------------
// An affine form c ^ <m, x> over the level's P parameters: `m` points at PW = (P + 63) / 64 + 1 words (a graph's forms live
// in one arena - a vector per form was a third of the time a graph took to pack).
struct Affine {
  const uint64_t *m = nullptr;
  bool c = false;
};
struct MaskArena {
  std::vector<uint64_t> buf;
  size_t used = 0;
  int PW;
  MaskArena(int P, size_t forms) : buf(forms * (size_t)((P + 63) / 64 + 1), 0ull), PW((P + 63) / 64 + 1) {}
  uint64_t *take() {
    if (used + (size_t)PW > buf.size()) return nullptr;
    uint64_t *r = &buf[used];
    used += (size_t)PW;
    return r;
  }
};

static Affine affine_from(MaskArena &ar, const uint8_t *bits, int P, bool c) {
  Affine a;
  uint64_t *m = ar.take();
  int i = 0;
  for (; i + 8 <= P; i += 8) m[i >> 6] |= (uint64_t)bits_of_8_bytes(bits + i) << (i & 63);
  for (; i < P; ++i)
    if (bits[i] & 1) m[i >> 6] |= 1ull << (i & 63);
  a.m = m;
  a.c = c;
  return a;
}
static std::vector<uint64_t> mask_vec(const Affine &a, int PW) { return std::vector<uint64_t>(a.m, a.m + PW); }

// A quadratic form over GF(2): q(x) = sum_{i<j} B[i][j] x_i x_j  ^  <lin, x>  ^  c.
// B is kept symmetric with zero diagonal (x_i^2 = x_i goes to `lin`).
struct QForm {
  int P = 0, PW = 0;
  std::vector<uint64_t> B;  // P rows x PW words
  std::vector<uint64_t> lin;
  bool c = false;
  explicit QForm(int p) : P(p), PW((p + 63) / 64 + 1), B((size_t)p * ((p + 63) / 64 + 1), 0ull), lin((p + 63) / 64 + 1, 0ull) {}
  uint64_t *row(int i) { return &B[(size_t)i * PW]; }
  void add_linear(const Affine &a) {
    for (int w = 0; w < PW; ++w) lin[w] ^= a.m[w];
    c ^= a.c;
  }
  // q ^= (a.c ^ <a.m,x>) * (b.c ^ <b.m,x>)
  void add_product(const Affine &a, const Affine &b) {
    for (int wa = 0; wa < PW; ++wa)
      for (uint64_t rest = a.m[wa]; rest; rest &= rest - 1) {
        const int i = 64 * wa + __builtin_ctzll(rest);
        if (i >= P) break;
        uint64_t *ri = row(i);
        for (int w = 0; w < PW; ++w) ri[w] ^= b.m[w];  // row i ^= b (may set the diagonal)
      }
    // symmetrise: the loop above added the ordered pairs (i in a, j in b); fold (i,j) and (j,i) together
    // by rebuilding the symmetric part lazily in `finish()`.
    if (a.c) for (int w = 0; w < PW; ++w) lin[w] ^= b.m[w];
    if (b.c) for (int w = 0; w < PW; ++w) lin[w] ^= a.m[w];
    c ^= (a.c && b.c);
  }
  // After all add_product calls B holds an arbitrary (non-symmetric) bilinear matrix M with
  // q = x^T M x.  Convert to the canonical alternating form: B'[i][j] = M[i][j] ^ M[j][i], diagonal -> lin.
  void finish() {
    for (int i = 0; i < P; ++i)
      if ((row(i)[i >> 6] >> (i & 63)) & 1) {
        lin[i >> 6] ^= 1ull << (i & 63);
        row(i)[i >> 6] ^= 1ull << (i & 63);
      }
    // B' = M ^ M^T: every set bit (i, j) of M toggles (j, i) of a copy
    std::vector<uint64_t> S(B);
    for (int i = 0; i < P; ++i)
      for (int w = 0; w < PW; ++w)
        for (uint64_t rest = B[(size_t)i * PW + w]; rest; rest &= rest - 1) {
          const int j = 64 * w + __builtin_ctzll(rest);
          if (j < P) S[(size_t)j * PW + (i >> 6)] ^= 1ull << (i & 63);
        }
    B.swap(S);
  }
  bool get(int i, int j) { return (row(i)[j >> 6] >> (j & 63)) & 1; }
  // the first set column of row i, -1 if the row is empty
  int first(int i) {
    const uint64_t *r = row(i);
    for (int w = 0; w < PW; ++w)
      if (r[w]) {
        const int j = 64 * w + __builtin_ctzll(r[w]);
        return j < P ? j : -1;
      }
    return -1;
  }
};

// Dickson reduction: q = XOR_s <u_s,x><v_s,x> ^ <lin,x> ^ c with rank(B)/2 product pairs.
// Pivot on (i,j) with B[i][j] = 1: with alpha = B[i] \ {i,j}, beta = B[j] \ {i,j},
//   x_i x_j ^ x_i<alpha,x> ^ x_j<beta,x> = (x_i ^ <beta,x>)(x_j ^ <alpha,x>) ^ <alpha,x><beta,x>.
static void dickson_reduce(QForm &q, std::vector<std::vector<uint64_t>> &us, std::vector<std::vector<uint64_t>> &vs) {
  const int P = q.P, PW = q.PW;
  std::vector<uint64_t> alpha((size_t)PW), beta((size_t)PW);
  for (int i = 0; i < P; ++i) {
    for (;;) {
      const int j = q.first(i);
      if (j < 0) break;
      std::copy(q.row(i), q.row(i) + PW, alpha.begin());
      std::copy(q.row(j), q.row(j) + PW, beta.begin());
      alpha[j >> 6] &= ~(1ull << (j & 63));  // B[i][i] is 0 already
      beta[i >> 6] &= ~(1ull << (i & 63));
      us.push_back(beta);
      vs.push_back(alpha);
      us.back()[i >> 6] ^= 1ull << (i & 63);
      vs.back()[j >> 6] ^= 1ull << (j & 63);
      // remove variables i and j from B (B is symmetric: the rows that hold column i / j are the set bits of row i / j)
      for (int w = 0; w < PW; ++w) {
        for (uint64_t rest = q.row(i)[w]; rest; rest &= rest - 1) {
          const int k = 64 * w + __builtin_ctzll(rest);
          if (k < P) q.row(k)[i >> 6] &= ~(1ull << (i & 63));
        }
        for (uint64_t rest = q.row(j)[w]; rest; rest &= rest - 1) {
          const int k = 64 * w + __builtin_ctzll(rest);
          if (k < P) q.row(k)[j >> 6] &= ~(1ull << (j & 63));
        }
      }
      for (int w = 0; w < PW; ++w) q.row(i)[w] = q.row(j)[w] = 0ull;
      // q ^= <alpha,x><beta,x>: B[k][l] ^= alpha_k beta_l ^ alpha_l beta_k ; lin_k ^= alpha_k beta_k
      for (int w = 0; w < PW; ++w)
        for (uint64_t rest = alpha[w] | beta[w]; rest; rest &= rest - 1) {
          const int k = 64 * w + __builtin_ctzll(rest);
          if (k >= P) break;
          const bool ak = (alpha[w] >> (k & 63)) & 1, bk = (beta[w] >> (k & 63)) & 1;
          uint64_t *rk = q.row(k);
          if (ak) for (int x = 0; x < PW; ++x) rk[x] ^= beta[x];
          if (bk) for (int x = 0; x < PW; ++x) rk[x] ^= alpha[x];
          if (ak && bk) q.lin[k >> 6] ^= 1ull << (k & 63);
          rk[k >> 6] &= ~(1ull << (k & 63));  // diagonal stays zero
        }
    }
  }
}

static bool push_mask_row(std::vector<uint32_t> &dst, const uint64_t *m, int P, int W) {
  uint64_t any = 0;
  const size_t base = dst.size();
  dst.resize(base + W, 0u);
  const int nw = (P + 31) / 32;
  for (int w = 0; w < nw && w < W; ++w) {
    uint64_t v = m[w >> 1] >> (32 * (w & 1));
    if (w == nw - 1 && (P & 31)) v &= (1ull << (P & 31)) - 1ull;
    dst[base + w] = (uint32_t)v;
    any |= (uint32_t)v;
  }
  return any != 0;
}

// Can this level be evaluated by the counting formulation?  (see eval_level_fast)
bool level_fast_eligible(const HostLevel &h) {
  const tsim_level_desc &d = h.d;
  for (int g = 0; g < h.G; ++g) {
    const int nA = d.ta ? h.i32[0][g] : 0;
    if (nA > 30) {
      // Beyond 30 terms the reference's own int32 scan may wrap (each factor is at most 2 in both embeddings of Z[w]) - unless
      // the terms say otherwise.  The scan multiplies one term at a time and takes ONE factor of two out after every product
      // (exact_scalar.py:43-49,62-79): a term 1 + w^(4 par + c) with c in {0, 4} is 2 or 0, its product is even and reduced at
      // once - the running value does not grow (detectors that are parities of f and earlier outcomes: any number of them);
      // c in {2, 6}: |1 +- i| = sqrt 2; odd c: 1.848 in one embedding, 0.765 in the other.  A coefficient is at most the mean
      // of the two embeddings' magnitudes, so log2 of the running coefficients is bounded by the sum of these costs (+ 1: the
      // scan's first element enters unreduced); the product with the next term (a factor <= 2) must still fit 31 bits.
      double cost = 1.0;
      for (int t = 0; t < nA; ++t) {
        const unsigned ph = h.u8[0][(size_t)g * d.ta + t] & 7u;
        cost += (ph & 1u) ? 0.886 : ((ph & 2u) ? 0.5 : 0.0);
      }
      if (cost > 29.0) return false;
    }
    for (int t = 0; t < d.tb; ++t) {
      const unsigned c = h.u8[2][(size_t)g * d.tb + t] & 7u;
      if (c & 1u) return false;  // odd eighth-turn coefficients never come out of the compiler
    }
    if (d.td > 60000 || h.P > 60000) return false;
    for (int j = 0; j < 4; ++j)
      if (std::llabs((long long)h.i32[2][(size_t)g * 4 + j]) > (1ll << 20)) return false;
  }
  return true;
}

// Build graph records + rows + NodePhases tables for eval_level_fast.  Returns false if a table
// entry does not fit int32 (the caller then falls back to the faithful layout).
//
// Per graph the w-exponent contributed by HalfPi rows (coefficients 2,4,6), PiProducts and the
// (-i)^(m1+m2+m3) of the NodePhases is rewritten at pack time as
//     k(x) = k0 + 2 * <lam, x> + 4 * ( <lin, x> ^ XOR_s <u_s,x><v_s,x> )          (mod 8)
// using  2*(sum of bits p_t) = 2*(XOR p_t) + 4*e2(p)  (mod 8)  for the list of bits that enter with
// coefficient 2 (coefficient 6 = 2 + 4), e2 = second elementary symmetric polynomial, and the Dickson
// normal form of the resulting GF(2) quadratic form.  k0 is folded into the table as a rotation.
bool pack_level_fast(HostLevel &h, int W, std::vector<uint32_t> &tables, bool &fixed_out, int &frame_out) {
  const int G = h.G, P = h.P;
  const tsim_level_desc &d = h.d;
  h.graph_rec.assign((size_t)G * G_WORDS, 0u);
  h.rows.clear();
  h.n_rows = 0;
  tables.clear();
  fixed_out = false;
  frame_out = 0;
  std::vector<std::vector<ZW>> entries((size_t)G);   // per graph: the main table (entry 0 excluded)
  std::vector<std::vector<ZW>> dentries((size_t)G);  // per graph: the separate PhasePairs table, if any
  h.fg.assign((size_t)G, FastGraph());
  // One graph's algebra is independent of the others': the graphs of a level are packed by up to 8 threads (C4: 1008 graphs,
  // ~30 us each - 31 of the 38 ms a fresh handle took before its first launch), their rows concatenated afterwards.
  std::vector<std::vector<uint32_t>> grows((size_t)G);
  std::vector<long long> g_nrows((size_t)G, 0);
  std::vector<char> can_be_zero((size_t)G, 0);  // delta rows, PhasePairs terms or a zero floatfactor: the graph's value can be exactly 0
  // what the two bounds below need of a graph's table, gathered where the table is made (the pool): the powers of its nonzero
  // entries, the largest |coefficient| x 2^power among them, the same for its separate PhasePairs table
  struct GraphAgg { int minp = INT32_MAX, maxp = INT32_MIN; long double big = 0, dbig = 0; bool has_d = false, d_bad = false; };
  std::vector<GraphAgg> agg((size_t)G);
  const auto t_fn0 = std::chrono::steady_clock::now();
  std::atomic<long long> ns_a{0}, ns_b{0}, ns_c{0};
  auto do_graph = [&](int g) -> bool {
    const auto tg0 = std::chrono::steady_clock::now();
    std::vector<uint32_t> &rows = grows[(size_t)g];
    FastGraph &fg = h.fg[(size_t)g];
    uint32_t *rec = &h.graph_rec[(size_t)g * G_WORDS];
    const int nA = d.ta ? h.i32[0][g] : 0;
    MaskArena arena(P, (size_t)2 * std::max(0, nA) + (size_t)d.tb + 2 * (size_t)d.tc + 2 * (size_t)std::max(0, d.td) + 4);
    const int PW = arena.PW;
    std::vector<Affine> two;  // bits entering the exponent with coefficient 2
    two.reserve((size_t)std::max(0, nA) + (size_t)d.tb);
    QForm q4(P);              // bit entering with coefficient 4
    int k0 = h.u8[12][g] & 7; // static phase
    auto add_six = [&](const Affine &a) {  // 6*p = 2*p + 4*p
      two.push_back(a);
      q4.add_linear(a);
    };
    // ---- NodePhases rows counted per class 0, 1, 3 (class 2 only feeds the exponent)
    int n[4] = {0, 0, 0, 0};
    for (int cls = 0; cls < 4; ++cls)
      for (int t = 0; t < nA; ++t) {
        const unsigned ph = h.u8[0][(size_t)g * d.ta + t] & 7u;
        if ((int)(ph & 3u) != cls) continue;
        const uint8_t *bits = &h.u8[1][((size_t)g * d.ta + t) * P];
        ++n[cls];
        const Affine ab = affine_from(arena, bits, P, (ph >> 2) != 0);
        if (cls != 0) add_six(ab);  // (-i)^(par')
        if (cls == 2) continue;
        rows.push_back(ph >> 2);
        push_mask_row(rows, ab.m, P, W);
        auto &dstm = cls == 0 ? fg.c0 : (cls == 1 ? fg.c1 : fg.c3);
        auto &dstc = cls == 0 ? fg.c0c : (cls == 1 ? fg.c1c : fg.c3c);
        dstm.push_back(mask_vec(ab, PW));
        dstc.push_back((uint8_t)(ph >> 2));
      }
    rec[GF_N01] = (uint32_t)n[0] | ((uint32_t)n[1] << 16);
    can_be_zero[(size_t)g] = (char)(n[0] > 0 || (d.td && h.i32[1][g] > 0) ||
                                    !(h.i32[2][(size_t)g * 4] | h.i32[2][(size_t)g * 4 + 1] | h.i32[2][(size_t)g * 4 + 2] | h.i32[2][(size_t)g * 4 + 3]));
    fg.n1 = n[1];
    rec[GF_N1] = (uint32_t)n[1];
    // ---- PhasePairs rows: two table-index bits per term when the combined table stays small,
    //      else the faithful sequential scan (rows carry the four tabulated term values)
    const int nD = d.td ? h.i32[1][g] : 0;
    if (nD > 5) return false;  // 4^nD table entries: beyond this use the faithful layout
    const long long combos = (long long)(n[1] + n[3] + 1) << (2 * nD);
    static const long long kMaxCombined = 1024;
    const bool d_tabled = nD > 0 && combos <= kMaxCombined;   // combined table, else a separate one
    std::vector<std::array<std::array<int, 4>, 4>> dterm((size_t)nD);  // [t][pa + 2 pb] -> term value
    for (int t = 0; t < nD; ++t) {
      const int al = h.u8[8][(size_t)g * d.td + t] & 7, be = h.u8[10][(size_t)g * d.td + t] & 7;
      for (int idx = 0; idx < 4; ++idx) {
        const int pa = idx & 1, pb = idx >> 1;
        const int a1 = (al + 4 * pa) & 7, b1 = (be + 4 * pb) & 7, g1 = (a1 + b1) & 7;
        uint32_t w = 0;
        for (int j = 0; j < 4; ++j) {
          const int v = (j == 0 ? 1 : 0) + kUnit[a1][j] + kUnit[b1][j] - kUnit[g1][j];
          dterm[t][idx][j] = v;
          w |= (uint32_t)(uint8_t)(int8_t)v << (8 * j);
        }
        (void)w;
      }
      const Affine da = affine_from(arena, &h.u8[9][((size_t)g * d.td + t) * P], P, false);
      const Affine db = affine_from(arena, &h.u8[11][((size_t)g * d.td + t) * P], P, false);
      push_mask_row(rows, da.m, P, W);
      push_mask_row(rows, db.m, P, W);
      fg.dal.push_back(mask_vec(da, PW));
      fg.dbe.push_back(mask_vec(db, PW));
    }
    rec[GF_ND] = (uint32_t)nD;
    fg.nD = nD;
    fg.d_tabled = d_tabled;
    // ---- HalfPi rows
    for (int t = 0; t < d.tb; ++t) {
      const uint32_t coeff = h.u8[2][(size_t)g * d.tb + t] & 7u;
      if (!coeff) continue;
      const Affine a = affine_from(arena, &h.u8[3][((size_t)g * d.tb + t) * P], P, false);
      if (coeff == 2) two.push_back(a);
      else if (coeff == 4) q4.add_linear(a);
      else add_six(a);
    }
    // ---- PiProducts: 4 * psi * phi
    for (int t = 0; t < d.tc; ++t) {
      const Affine psi = affine_from(arena, &h.u8[5][((size_t)g * d.tc + t) * P], P, h.u8[4][(size_t)g * d.tc + t] & 1u);
      const Affine phi = affine_from(arena, &h.u8[7][((size_t)g * d.tc + t) * P], P, h.u8[6][(size_t)g * d.tc + t] & 1u);
      q4.add_product(psi, phi);
      arena.used -= 2 * (size_t)PW;  // (consumed at once: the two slots are the next term's)
      std::fill(arena.buf.begin() + (long)arena.used, arena.buf.begin() + (long)arena.used + 2 * PW, 0ull);
    }
    // ---- 2 * sum(two) = 2 * XOR(two) + 4 * e2(two)
    std::vector<uint64_t> lam_m((size_t)PW, 0ull);
    bool lam_c = false;
    // e2 = XOR_{s < t} a_s a_t = XOR_s a_s (XOR_{t > s} a_t): one product per row against the XOR of the rows behind it
    // (the pairwise form - |two|^2 / 2 products of P x P bits each - was 35 of the 38 ms C4's 1024 graphs took to pack)
    {
      std::vector<uint64_t> suffix_m((size_t)PW, 0ull);
      bool suffix_c = false;
      for (size_t s = two.size(); s-- > 0;) {
        for (int w = 0; w < PW; ++w) lam_m[(size_t)w] ^= two[s].m[w];
        lam_c ^= two[s].c;
        if (s + 1 < two.size()) q4.add_product(two[s], Affine{suffix_m.data(), suffix_c});
        for (int w = 0; w < PW; ++w) suffix_m[(size_t)w] ^= two[s].m[w];
        suffix_c ^= two[s].c;
      }
    }
    if (lam_c) {  // 2*(1 ^ y) = 2 + 6*y = 2 + 2*y + 4*y
      k0 += 2;
      q4.add_linear(Affine{lam_m.data(), false});
    }
    const auto tg1 = std::chrono::steady_clock::now();
    q4.finish();
    std::vector<std::vector<uint64_t>> us, vs;
    dickson_reduce(q4, us, vs);
    const auto tg2 = std::chrono::steady_clock::now();
    ns_a += (tg1 - tg0).count();
    ns_b += (tg2 - tg1).count();
    if (q4.c) k0 += 4;
    if (us.size() > 60000) return false;
    // ---- rows: lam, lin, then the product pairs
    uint32_t flags = d_tabled ? TSIMK_GFLAG_D_TABLED : 0u;
    {
      std::vector<uint32_t> tmp;
      if (push_mask_row(tmp, lam_m.data(), P, W)) { flags |= TSIMK_GFLAG_LAM; rows.insert(rows.end(), tmp.begin(), tmp.end()); }
      tmp.clear();
      if (push_mask_row(tmp, q4.lin.data(), P, W)) { flags |= TSIMK_GFLAG_LIN; rows.insert(rows.end(), tmp.begin(), tmp.end()); }
    }
    for (size_t s = 0; s < us.size(); ++s) {
      push_mask_row(rows, us[s].data(), P, W);
      push_mask_row(rows, vs[s].data(), P, W);
    }
    rec[GF_N3H] = (uint32_t)n[3] | ((uint32_t)us.size() << 16);
    rec[GF_FLAGS] = flags;
    fg.lam = lam_m;
    fg.lin = q4.lin;
    fg.us = us;
    fg.vs = vs;
    g_nrows[(size_t)g] = n[0] + n[1] + n[3] + ((flags & TSIMK_GFLAG_LAM) ? 1 : 0) + ((flags & TSIMK_GFLAG_LIN) ? 1 : 0) +
                2 * (long long)us.size() + 2 * nD;
    // ---- table entries (index: ((delta + n1) << 2 nD | dbits) + 1; entry 0 is the exact zero)
    //   canon( 2^n0 (1+i)^n2 (1+w)^(n1-m1) (1-w)^m1 (1+w^3)^(n3-m3) (1-w^3)^m3 * i^(m1+m3)
    //          * floatfactor * w^k0 * [product of the PhasePairs terms selected by dbits] ) * 2^power2
    //   with m3 = max(delta,0), m1 = max(-delta,0)
    const long long ff[4] = {h.i32[2][(size_t)g * 4], h.i32[2][(size_t)g * 4 + 1], h.i32[2][(size_t)g * 4 + 2],
                             h.i32[2][(size_t)g * 4 + 3]};
    const int ndb = d_tabled ? (1 << (2 * nD)) : 1;
    // (1 + w^k)^j for the four factors whose exponent moves with delta, once per graph; the rest is common to every entry.
    // (The canonical form of an element of Z[w] * 2^p is unique, so the order of the exact products does not matter; zeros get
    // their power below.)
    auto powers = [&](int k, int upto) {
      std::vector<ZW> pw((size_t)upto + 1);
      pw[0] = ZW{{1, 0, 0, 0}, 0};
      long long f[4];
      unit_plus_one(k, f);
      for (int j = 1; j <= upto; ++j) {
        pw[(size_t)j] = pw[(size_t)j - 1];
        zw_mul(pw[(size_t)j], f);
      }
      return pw;
    };
    const std::vector<ZW> pw1 = powers(1, n[1]), pw5 = powers(5, n[1]), pw3 = powers(3, n[3]), pw7 = powers(7, n[3]);
    ZW common{{1, 0, 0, 0}, 0};
    {
      long long f[4];
      unit_plus_one(0, f);
      for (int i = 0; i < n[0]; ++i) zw_mul(common, f);
      unit_plus_one(2, f);
      for (int i = 0; i < n[2]; ++i) zw_mul(common, f);
      zw_mul(common, ff);
      common.p += h.i32[3][g];  // power2
    }
    auto mul_zw = [](ZW &x, const ZW &y) {
      zw_mul(x, y.c);
      x.p += y.p;
    };
    for (int delta = -n[1]; delta <= n[3]; ++delta) {
      const int m3 = delta > 0 ? delta : 0, m1 = delta < 0 ? -delta : 0;
      ZW z = common;
      mul_zw(z, pw1[(size_t)(n[1] - m1)]);
      mul_zw(z, pw5[(size_t)m1]);
      mul_zw(z, pw3[(size_t)(n[3] - m3)]);
      mul_zw(z, pw7[(size_t)m3]);
      const int r = (2 * (m1 + m3) + k0) & 7;
      const long long rot[4] = {kUnit[r][0], kUnit[r][1], kUnit[r][2], kUnit[r][3]};
      zw_mul(z, rot);
      for (int db = 0; db < ndb; ++db) {
        ZW e = z;
        for (int t = 0; t < nD && d_tabled; ++t) {
          const int sel = (db >> (2 * (nD - 1 - t))) & 3;  // term 0 holds the most significant pair
          const long long tv[4] = {dterm[t][sel][0], dterm[t][sel][1], dterm[t][sel][2], dterm[t][sel][3]};
          zw_mul(e, tv);
        }
        for (int j = 0; j < 4; ++j)
          if (e.c[j] > INT32_MAX || e.c[j] < INT32_MIN) return false;
        if (!(e.c[0] | e.c[1] | e.c[2] | e.c[3])) e.p = h.approx ? 0 : TSIMK_ZERO_POWER;
        entries[g].push_back(e);
      }
    }
    if (nD > 0 && !d_tabled) {  // separate PhasePairs table over the 4^nD parity patterns
      for (int db = 0; db < (1 << (2 * nD)); ++db) {
        ZW e{{1, 0, 0, 0}, 0};
        for (int t = 0; t < nD; ++t) {
          const int sel = (db >> (2 * (nD - 1 - t))) & 3;
          const long long tv[4] = {dterm[t][sel][0], dterm[t][sel][1], dterm[t][sel][2], dterm[t][sel][3]};
          zw_mul(e, tv);
        }
        dentries[g].push_back(e);
      }
    }
    memcpy(&rec[GF_APRE], &h.approx_v[2 * (size_t)g], 4);
    memcpy(&rec[GF_APIM], &h.approx_v[2 * (size_t)g + 1], 4);
    {
      GraphAgg &A = agg[(size_t)g];
      for (auto &e : entries[(size_t)g]) {
        if (!(e.c[0] | e.c[1] | e.c[2] | e.c[3])) continue;
        A.minp = std::min(A.minp, e.p);
        A.maxp = std::max(A.maxp, e.p);
        // a rotation by i permutes/negates coefficients: bound by the largest one
        long double m = 0;
        for (auto v : e.c) m = std::max(m, (long double)std::llabs(v));
        A.big = std::max(A.big, ldexpl(m, e.p));
      }
      A.has_d = !dentries[(size_t)g].empty();
      for (auto &e : dentries[(size_t)g]) {
        if (e.p < 0 || e.p > 40) { A.d_bad = true; continue; }
        long double m = 0;
        for (auto v : e.c) m = std::max(m, (long double)std::llabs(v));
        A.dbig = std::max(A.dbig, ldexpl(m, e.p));
      }
    }
    ns_c += (std::chrono::steady_clock::now() - tg2).count();
    return true;
  };
  {
    // the graphs of a level on the process-wide pool (tsim_pool.cpp): big levels on up to 24 threads (C4: 256 graphs, ~50 us each)
    std::atomic<bool> failed{false};
    tsim_parallel_for((size_t)G, G >= 128 ? 24 : (G >= 32 ? 12 : (G >= 8 ? 4 : 1)), [&](size_t g) {
      if (failed.load(std::memory_order_relaxed)) return;
      if (!do_graph((int)g)) failed.store(true);
    });
    if (failed.load()) return false;
  }
  for (int g = 0; g < G; ++g) {
    h.graph_rec[(size_t)g * G_WORDS + GF_ROWS] = (uint32_t)h.rows.size();
    h.rows.insert(h.rows.end(), grows[(size_t)g].begin(), grows[(size_t)g].end());
    h.n_rows += g_nrows[(size_t)g];
  }
  // ---- can the REFERENCE's running sum of this level leave int32 (exact_scalar.py:74-84,173-189)?  Its carry takes the smallest
  //      power seen so far and is reduced by ONE factor of two per addition; a graph whose value is exactly zero still brings
  //      a power - as low as its power2, when the zero factor comes first in the scan - and aligning the carry to a lower power
  //      multiplies its coefficients (fuzz program 7048: carry 3.2e7 x 2^-30 met a zero term of power -37: x 2^7, wrapped; the
  //      exact formulation here has no power for a zero).  Ruled out when every graph's largest |coefficient|, counted in units
  //      of the lowest power any term of the level can take, sums to less than 2^31.  Not ruled out: `sum_wrap_possible` - the
  //      program still runs on the exact formulation (it differs from the reference only on inputs where the reference's own
  //      arithmetic wraps; TSIM_AMD_MODE=faithful mirrors the wrap), tsim_program_stats reports it.
  h.sum_wrap_possible = false;
  if (!h.approx && G > 0) {
    int plow = INT32_MAX;
    for (int g = 0; g < G; ++g) {
      const int32_t *ffg = &h.i32[2][(size_t)g * 4];
      const uint32_t o = (uint32_t)(ffg[0] | ffg[1] | ffg[2] | ffg[3]);
      const int v2 = o ? __builtin_ctz(o) : 0;  // (the product with the floatfactor is reduced once: its content can stay in the coefficients)
      if (can_be_zero[(size_t)g]) plow = std::min(plow, h.i32[3][g]);
      if (agg[(size_t)g].minp != INT32_MAX) plow = std::min(plow, agg[(size_t)g].minp - std::max(0, v2 - 1));
    }
    long double tot = 0;
    for (int g = 0; g < G && plow != INT32_MAX; ++g) {
      const GraphAgg &A = agg[(size_t)g];
      long double worst = ldexpl(A.big, -plow);
      if (A.has_d) worst = A.d_bad ? 1e30L : 4 * worst * A.dbig;
      tot += worst;
    }
    h.sum_wrap_possible = tot >= 2147483000.0L;
  }
  // ---- fixed frame: every term of the level comes straight from a table and, shifted to the
  //      level's smallest power, the worst-case sum of all graphs stays inside int32
  bool fixed = !h.approx && G > 0;
  int frame = INT32_MAX;
  if (fixed) {
    for (auto &A : agg) frame = std::min(frame, A.minp);
    if (frame == INT32_MAX) frame = 0;
    long double total = 0;
    for (int g = 0; g < G && fixed; ++g) {
      const GraphAgg &A = agg[(size_t)g];
      if (A.maxp != INT32_MIN && A.maxp - frame > 40) { fixed = false; break; }
      // separate PhasePairs table: entries become plain integers c * 2^p (p >= 0); the product
      // with the main entry is a sum of four coefficient products
      if (A.d_bad) { fixed = false; break; }
      long double worst = ldexpl(A.big, -frame);
      if (A.has_d) worst = 4 * worst * A.dbig;
      total += worst;
    }
    if (total >= 2147483000.0L) fixed = false;
  }
  // the tables, graph by graph through the pool into their places (the sizes are known now): one thread pushing a level's words
  // one at a time - 630 000 for C4's level of 256 graphs - was most of that level's 1.5 ms
  {
    std::vector<size_t> off_main((size_t)G + 1, 0), off_d((size_t)G + 1, 0);
    for (int g = 0; g < G; ++g) off_main[(size_t)g + 1] = off_main[(size_t)g] + (fixed ? 16 : 8) * (1 + entries[(size_t)g].size());
    for (int g = 0; g < G; ++g) off_d[(size_t)g + 1] = off_d[(size_t)g] + 8 * dentries[(size_t)g].size();
    const size_t d_base = off_main[(size_t)G];
    tables.assign(d_base + off_d[(size_t)G], 0u);
    std::atomic<bool> overflow{false};
    tsim_parallel_for((size_t)G, G >= 128 ? 16 : (G >= 32 ? 8 : (G >= 8 ? 2 : 1)), [&](size_t gs) {
      const int g = (int)gs;
      uint32_t *rec = &h.graph_rec[(size_t)g * G_WORDS];
      rec[GF_TBL] = (uint32_t)off_main[gs];
      uint32_t *t = &tables[off_main[gs]];
      if (fixed) {
        // fixed-frame levels: an entry is 16 words = the value times i^r for r = 0..3 (4 words each),
        // pre-shifted to the frame power, so the kernel adds the selected rotation without any
        // per-lane rotate/shift.  Entry 0 is the exact zero.
        t += 16;
        for (auto &e : entries[gs]) {
          const bool nz = (e.c[0] | e.c[1] | e.c[2] | e.c[3]) != 0;
          const int sh = nz ? e.p - frame : 0;
          long long v[4] = {e.c[0] * (1ll << sh), e.c[1] * (1ll << sh), e.c[2] * (1ll << sh), e.c[3] * (1ll << sh)};
          for (int r = 0; r < 4; ++r) {
            for (int j = 0; j < 4; ++j) *t++ = (uint32_t)(int32_t)v[j];
            const long long tt[4] = {-v[2], v[3], v[0], -v[1]};  // times i: (a,b,c,d) -> (-c, d, a, -b)
            for (int j = 0; j < 4; ++j) v[j] = tt[j];
          }
        }
      } else {
        t[4] = (uint32_t)(h.approx ? 0 : TSIMK_ZERO_POWER);  // entry 0: exact zero
        t += 8;
        for (auto &e : entries[gs]) {
          for (int j = 0; j < 4; ++j) t[j] = (uint32_t)(int32_t)e.c[j];
          t[4] = (uint32_t)e.p;
          t += 8;
        }
      }
      if (dentries[gs].empty()) return;
      rec[GF_TBL2] = (uint32_t)(d_base + off_d[gs]);
      uint32_t *d2 = &tables[d_base + off_d[gs]];
      for (auto &e : dentries[gs]) {
        const int sh = fixed ? e.p : 0;
        for (int j = 0; j < 4; ++j) {
          const long long v = e.c[j] * (1ll << sh);
          if (v > INT32_MAX || v < INT32_MIN) overflow.store(true);
          d2[j] = (uint32_t)(int32_t)v;
        }
        d2[4] = (uint32_t)(fixed ? 0 : e.p);
        d2 += 8;
      }
    });
    if (overflow.load()) return false;
  }
  if (tsim_debug("pack"))
    fprintf(stderr, "[tsim] pack_level_fast G=%d P=%d: %.2f ms wall; per-graph sums: terms -> forms %.2f ms, finish + Dickson %.2f ms, rows + table entries %.2f ms\n", G, P,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_fn0).count(), ns_a.load() * 1e-6, ns_b.load() * 1e-6, ns_c.load() * 1e-6);
  fixed_out = fixed;
  frame_out = fixed ? frame : 0;
  h.fixed = fixed;
  h.frame = frame_out;
  return true;
}


bool level_v4_eligible(const HostLevel &h, bool wide) {
  if (h.P > (wide ? TSIMK_LWW_MAX_F + 8 : 128)) return false;  // wide: the sparse-column kernel (tsim_kernel4w.hip.h); chunk tables: x in two or three words (65..80 parameters: the caller checks F <= 64)
  for (const FastGraph &fg : h.fg) {
    if (fg.us.size() > 32) return false;
    if (fg.c0.size() + fg.c1.size() + fg.c3.size() > 32) return false;
    if (2 * fg.nD > 28) return false;
  }
  return true;
}

static inline bool mask_bit(const std::vector<uint64_t> &m, int i) { return (m[(size_t)i >> 6] >> (i & 63)) & 1; }

// recs4: G x G4_WORDS; tabs4: ntiles x nch x 16 x GT x 4 words.  `v3recs` are the level's patched
// fast-layout graph records (for the term-table offsets and the approximate floatfactors).
// `stabs4` (optional, sparse_F >= 0): the "sparse f" table of the same tile: one entry per f COLUMN
// (bits 0..F-1), one all-zero entry, then 2 x 16 entries for the two 4-bit chunks of the output
// bits F..F+7 (row constants live in the first of them): [tile][entry][graph][4 words].
void emit_level4(const HostLevel &h, int GT, int nch, const uint32_t *v3recs, std::vector<uint32_t> &recs4,
                        std::vector<uint32_t> &tabs4, int &nch_out, int &ntiles_out, int sparse_F,
                        std::vector<uint32_t> &stabs4) {
  const int G = h.G, P = h.P;
  const int ntiles = (G + GT - 1) / GT;  // nch: chunks per tile, the same for every level (zero padded)
  nch_out = nch;
  ntiles_out = ntiles;
  recs4.assign((size_t)G * G4_WORDS, 0u);
  tabs4.assign((size_t)ntiles * nch * 16 * GT * 4, 0u);
  // entries per graph of a column-table tile: F + 33 used; tiles of several graphs pad to a constant (the kernel's per-graph
  // offsets are immediates), the one-graph tiles of the wide layout do not
  const int sent = sparse_F < 0 ? 0 : (GT > 1 ? std::max(TSIMK_SPARSE_ENTRIES, sparse_F + 33) : sparse_F + 33);
  stabs4.assign((size_t)ntiles * sent * GT * 4, 0u);
  // (one graph's tables are independent of the others': the graphs of a level go through the process-wide pool - C4's level of
  // 256 graphs took 3 ms on one thread, the longest pole of the chunk-table phase)
  auto do_graph = [&](int g) {
    const FastGraph &fg = h.fg[(size_t)g];
    std::vector<std::array<uint32_t, 4>> col((size_t)std::max(P, 1), std::array<uint32_t, 4>{0u, 0u, 0u, 0u});
    std::array<uint32_t, 4> cst = {0u, 0u, 0u, 0u};
    auto place = [&](const std::vector<uint64_t> &m, int word, int bit) {
      for (size_t w = 0; w < m.size(); ++w)
        for (uint64_t rest = m[w]; rest; rest &= rest - 1) {
          const int i = 64 * (int)w + __builtin_ctzll(rest);
          if (i < P) col[(size_t)i][(size_t)word] |= 1u << bit;
        }
    };
    const int h2 = (int)fg.us.size();
    for (int s = 0; s < h2; ++s) {
      place(fg.us[(size_t)s], 0, s);
      place(fg.vs[(size_t)s], 1, s);
    }
    uint32_t M0 = 0, M1 = 0, M3 = 0;
    int b = 0;
    for (size_t t = 0; t < fg.c0.size(); ++t, ++b) { place(fg.c0[t], 2, b); M0 |= 1u << b; if (fg.c0c[t]) cst[2] |= 1u << b; }
    for (size_t t = 0; t < fg.c1.size(); ++t, ++b) { place(fg.c1[t], 2, b); M1 |= 1u << b; if (fg.c1c[t]) cst[2] |= 1u << b; }
    for (size_t t = 0; t < fg.c3.size(); ++t, ++b) { place(fg.c3[t], 2, b); M3 |= 1u << b; if (fg.c3c[t]) cst[2] |= 1u << b; }
    for (int t = 0; t < fg.nD; ++t) {  // index bits: term 0 holds the most significant pair
      place(fg.dal[(size_t)t], 3, 2 * (fg.nD - 1 - t));
      place(fg.dbe[(size_t)t], 3, 2 * (fg.nD - 1 - t) + 1);
    }
    place(fg.lam, 3, 30);
    place(fg.lin, 3, 31);
    const int tile = g / GT, j = g % GT;
    for (int c = 0; c < nch; ++c) {
      std::array<uint32_t, 4> vals[16];
      vals[0] = (c == 0) ? cst : std::array<uint32_t, 4>{0u, 0u, 0u, 0u};
      for (int v = 1; v < 16; ++v) {  // value v = value (v without its lowest bit) ^ that bit's column
        const int i = 4 * c + __builtin_ctz((unsigned)v);
        vals[v] = vals[v & (v - 1)];
        if (i < P)
          for (int w = 0; w < 4; ++w) vals[v][(size_t)w] ^= col[(size_t)i][(size_t)w];
      }
      for (int v = 0; v < 16; ++v) {
        const std::array<uint32_t, 4> &val = vals[v];
        // [tile][chunk][graph][value]: the lanes of a wave read ONE graph's entry for THEIR chunk value - 16 bytes apart per value,
        // so the 16 values fall on 16 different groups of banks (value-major entries of GT x 16 bytes put values v and v + 4 on
        // the same banks: k_sample4's LDS pipe spent 2.4 x its busy time in bank conflicts, profiles/r04/full_kernel_pmc.txt)
        uint32_t *dst = &tabs4[((((size_t)tile * nch + c) * GT + j) * 16 + v) * 4];
        for (int w = 0; w < 4; ++w) dst[w] = val[(size_t)w];
      }
    }
    if (sent) {
      auto put = [&](int e, const std::array<uint32_t, 4> &val) {
        uint32_t *dst = &stabs4[(((size_t)tile * GT + j) * sent + e) * 4];  // [tile][graph][entry]: consecutive entries on consecutive bank groups
        for (int w = 0; w < 4; ++w) dst[w] = val[(size_t)w];
      };
      for (int i = 0; i < sparse_F && i < P; ++i) put(i, col[(size_t)i]);
      for (int c = 0; c < 2; ++c)
        for (int v = 0; v < 16; ++v) {
          std::array<uint32_t, 4> val = (c == 0) ? cst : std::array<uint32_t, 4>{0u, 0u, 0u, 0u};
          for (int bb = 0; bb < 4; ++bb) {
            const int i = sparse_F + 4 * c + bb;
            if (!((v >> bb) & 1) || i >= P) continue;
            for (int w = 0; w < 4; ++w) val[(size_t)w] ^= col[(size_t)i][(size_t)w];
          }
          put(sparse_F + 1 + 16 * c + v, val);
        }
    }
    uint32_t *r4 = &recs4[(size_t)g * G4_WORDS];
    const uint32_t *r3 = v3recs + (size_t)g * G_WORDS;
    r4[G4_M0] = M0; r4[G4_M1] = M1; r4[G4_M3] = M3;
    r4[G4_PM] = h2 >= 32 ? 0xFFFFFFFFu : ((1u << h2) - 1u);
    r4[G4_N1] = (uint32_t)fg.n1;
    r4[G4_DBITS] = (uint32_t)(2 * fg.nD);
    r4[G4_TBL] = r3[GF_TBL];
    r4[G4_TBL2] = r3[GF_TBL2];
    r4[G4_FLAGS] = fg.nD == 0 ? 0u : (fg.d_tabled ? TSIMK_G4FLAG_D_COMBINED : TSIMK_G4FLAG_D_SEPARATE);
    r4[G4_APRE] = r3[GF_APRE];
    r4[G4_APIM] = r3[GF_APIM];
  };
  tsim_parallel_for((size_t)G, G >= 128 ? 16 : (G >= 32 ? 8 : (G >= 8 ? 2 : 1)), [&](size_t g) { do_graph((int)g); });
}

// Gather program (tsim_lw.hip.h): bit moves (src f bit -> dst bit, flip) merged into runs that are
// contiguous in both the source and the destination word; 4-word runs, padded to chunks of four.
std::vector<uint32_t> emit_gather_program(std::vector<std::array<int, 3>> e) {
  std::sort(e.begin(), e.end(), [](const std::array<int, 3> &a, const std::array<int, 3> &b) {
    if ((a[0] >> 5) != (b[0] >> 5)) return (a[0] >> 5) < (b[0] >> 5);
    if ((a[1] >> 5) != (b[1] >> 5)) return (a[1] >> 5) < (b[1] >> 5);
    return a[0] < b[0];
  });
  std::vector<uint32_t> out;
  size_t i = 0;
  while (i < e.size()) {
    size_t j = i + 1;
    while (j < e.size() && e[j][0] == e[j - 1][0] + 1 && e[j][1] == e[j - 1][1] + 1 &&
           (e[j][0] >> 5) == (e[i][0] >> 5) && (e[j][1] >> 5) == (e[i][1] >> 5))
      ++j;
    const int len = (int)(j - i);
    uint32_t flip = 0;
    for (size_t k = i; k < j; ++k) flip |= (uint32_t)(e[k][2] & 1) << (k - i);
    const uint32_t mask = len >= 32 ? 0xFFFFFFFFu : ((1u << len) - 1u);
    out.push_back((uint32_t)(e[i][0] & 31) | ((uint32_t)(e[i][1] & 31) << 8) | ((uint32_t)(e[i][1] >> 5) << 16) |
                  ((uint32_t)(e[i][0] >> 5) << 24));
    out.push_back(mask);
    out.push_back(flip);
    out.push_back(0u);
    i = j;
  }
  while (out.size() % 16) out.push_back(0u);  // mask = 0 runs: no-ops
  return out;
}

// Rotate-and-mask form of a gather program, for kernels that hold the f row and the output words in REGISTERS
// (k_sample_lw_reg): f words 0..3 -> output words 0..1.  The moves are bucketed by (source word, destination
// word), so the kernel's loops over the buckets are compile-time and no run needs a select; a run is two VALU
// operations, dst |= rotr(src, rot) & mask, the flips are one XOR per output word at the end.
//   words 0..7   groups of four runs in bucket s * 2 + d          words 8..15  word offset of the bucket's runs
//   words 16..17 flip masks of output words 0 and 1               (header: 32 words)
//   runs: [rot, mask] pairs, every bucket padded to a multiple of four with mask = 0
std::vector<uint32_t> emit_rotmask_program(const std::vector<std::array<int, 3>> &e) {
  std::vector<uint32_t> out(32, 0u);
  for (int s = 0; s < 4; ++s)
    for (int d = 0; d < 2; ++d) {
      std::vector<std::array<int, 2>> m;  // (src bit, dst bit) inside the words
      for (auto &x : e)
        if ((x[0] >> 5) == s && (x[1] >> 5) == d) {
          m.push_back({x[0] & 31, x[1] & 31});
          if (x[2] & 1) out[16 + d] |= 1u << (x[1] & 31);
        }
      std::sort(m.begin(), m.end());
      const int b = s * 2 + d;
      out[8 + b] = (uint32_t)out.size();
      uint32_t runs = 0;
      size_t i = 0;
      while (i < m.size()) {
        size_t j = i + 1;
        while (j < m.size() && m[j][0] == m[j - 1][0] + 1 && m[j][1] == m[j - 1][1] + 1) ++j;
        const int len = (int)(j - i);
        const uint32_t field = len >= 32 ? 0xFFFFFFFFu : ((1u << len) - 1u);
        out.push_back((uint32_t)((m[i][0] - m[i][1]) & 31));  // rotate right: source bit p lands on p - rot
        out.push_back(field << m[i][1]);
        ++runs;
        i = j;
      }
      while (runs % 4) { out.push_back(0u); out.push_back(0u); ++runs; }
      out[b] = runs / 4;
    }
  while (out.size() % 16) out.push_back(0u);
  return out;
}

}  // namespace tsimhost
