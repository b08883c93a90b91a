// tsim_kernel4w.hip.h - sparse-column sampling kernel for WIDE components (more than 64 parameters), gfx950.
//
// The chunk-table kernel (k_sample4) forms every parity of a graph as one 128-bit word Y_g = R_g x from 4-bit
// chunks of x - ceil(P/4) LDS reads per graph, which stops paying beyond 64 parameters (C5, the d=5 surface code
// with an injected T: F = 200 selected f bits, 3 outputs).  But x is SPARSE there: a handful of error bits plus the
// outcome bits.  So Y_g = XOR over the lane's set f bits j of COLUMN_j(R_g) (one 16-byte LDS read each, from the
// same per-tile column tables k_sample4 uses for its sparse-f path) XOR the two 4-bit chunks of the outcome bits:
// K + 2 reads per graph whatever P is, instead of ~3 VALU ops per row and 32-bit word (the W = 8 row kernel spends
// ~10^4 instructions per shot on C5, this ~10^3).
//
//   * the set bits are found on the packed f row with the component's selection masks (ascending f_selection: the
//     position inside f_sel is a popcount), once per component - they do not change from level to level;
//   * a lane with more than K set bits is not evaluated here: its row goes to the hard-row lists (the same protocol
//     as the pattern-table pass) and the row kernel (k_sample<W>) evaluates it afterwards; so does the
//     normalisation-check row;
//   * everything behind Y_g - class counts, Dickson pairs, term tables, exact sum, float epilogue, Threefry draw -
//     is eval_level4 / the k_sample4 epilogue, unchanged: same values, same bits.
#pragma once
#include "tsim_kernel4.hip.h"
#include "tsim_lw.hip.h"

namespace tsimk {

// component record extension of the wide layout: C4_SELMASK = image offset of TSIMK_W_SELWORDS selection-mask
// words (f bits 0..511) followed by as many prefix counts (selected bits in the lower words)
enum { C4_SELMASK = 9 };
#define TSIMK_W_SELWORDS 16

struct Wide4Args {
  SampleArgs s;          // row_index / row_count: optional INPUT list (device-side post-selection)
  int comp4_off;
  int has_check;         // the first slot's row is the normalisation-check row: left to the row kernel
  uint32_t *hard_index;  // out: n_lists sub-lists of list_cap rows for the row kernel
  uint32_t *ctl;         // ctl[32 k] = entries of list k, ctl[32 LISTS] = check row (as LwArgs)
  uint32_t *ctl_next;    // the counter set of the NEXT launch: reset here
  int list_cap, n_lists;
  // Behind a pattern-table first pass (k_sample_lw<true>): the input is that pass's hard-row lists (s.row_lists
  // sub-lists of s.row_list_cap entries at s.row_index, counts at s.row_count[32 k]; list k is served by the blocks
  // with blockIdx % row_lists == k), the check row is the one that pass recorded, and block 0 reports the list
  // lengths to the launch planner.
  const uint32_t *check_row_in;  // nullptr: the check row is slot 0
  uint32_t *feedback;            // optional: [rows in the lists, longest list, rows of the launch]
  int resident;          // 1: the LDS table area holds ALL levels of a component at once (one copy, no per-level barriers)
  int stream_buf;        // resident = 0: bytes of each of the two LDS buffers the column tables are streamed through (groups of graphs)
};

// One level from LDS-resident column tables (graph g of the level at lds_level + g * ent_bytes): no copy, no barrier.
template <int NR, bool FIXED, bool LT = false>
__device__ __forceinline__ void eval_level4_resident(const uint32_t *gimg, cptr img, cptr lvl, const uint32_t (&e)[NR],
                                                     uint32_t lds_level, uint32_t ent_bytes, float &out_re, float &out_im,
                                                     uint32_t tt_bias = 0u) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef const __attribute__((address_space(3))) u32x4 *lds_u4p;
  const uint32_t G = lvl[L4_G];
  const bool approx = (lvl[L4_FLAGS] & TSIMK_LFLAG_APPROX) != 0;
  cptr recs = img + lvl[L4_RECS];
  Acc4 S;
  for (uint32_t g = 0; g < G; ++g) {
    const uint32_t base = lds_level + g * ent_bytes;
    uint32_t U = 0, V = 0, O1 = 0, O2 = 0;
#if defined(TSIMK_W4_SKIP) && (TSIMK_W4_SKIP & 1)
#pragma unroll
    for (int c = 0; c < NR; c += 2) { U ^= e[c] + base; V ^= e[c + 1] * 3u; O1 ^= e[c] >> 3; O2 ^= e[c + 1] >> 5; }  // diagnostic: no LDS reads
#else
#pragma unroll
    for (int c = 0; c < NR; c += 2) {
      const u32x4 v = *(lds_u4p)(uintptr_t)(base + e[c]);
      const u32x4 w = *(lds_u4p)(uintptr_t)(base + e[c + 1]);
      U = xor3(U, v.x, w.x); V = xor3(V, v.y, w.y); O1 = xor3(O1, v.z, w.z); O2 = xor3(O2, v.w, w.w);
      if (c % 6 == 4 && c + 2 < NR) __builtin_amdgcn_sched_barrier(0);
    }
#endif
#if defined(TSIMK_W4_SKIP) && (TSIMK_W4_SKIP & 2)
    S.sa ^= (int)(U ^ V); S.sb ^= (int)(O1 ^ O2);  // diagnostic: no term table, no record
#else
    acc_graph4<FIXED, LT>(S, gimg, recs + g * G4_WORDS, U, V, O1, O2, approx, tt_bias);
#endif
  }
  acc_finish4<FIXED>(S, lvl, approx, out_re, out_im);
}

// One level with the column tables STREAMED through LDS in groups of as many graphs as a buffer holds (the wide layout is
// [graph][entry]: a level's graphs are consecutive (F + 33) x 16-byte blocks, so any run of them is one linear copy).
// eval_level4<GT = 1> took a barrier per graph: 140 graphs (class F60 of scripts/shape_map.py) were 140 copy-and-barrier
// round trips per level walk, 109 us per batch for the 0.8 % of the rows the tables miss; in groups of ~20 graphs it is 9.
// All threads of the block must call this together.
template <int NR, bool FIXED>
__device__ __forceinline__ void eval_level4_groups(const uint32_t *gimg, cptr img, cptr lvl, const uint32_t (&ent0)[NR],
                                                   uint32_t ent_bytes, uint32_t table_off, uint8_t *lds_tab, uint32_t buf_bytes,
                                                   float &out_re, float &out_im) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef const __attribute__((address_space(3))) u32x4 *lds_u4p;
  const uint32_t G = lvl[L4_G];
  const bool approx = (lvl[L4_FLAGS] & TSIMK_LFLAG_APPROX) != 0;
  const uint32_t tg = buf_bytes / ent_bytes > 1u ? buf_bytes / ent_bytes : 1u;  // graphs per group
  const uint32_t ngroups = (G + tg - 1u) / tg;
  const uint32_t ent_vec = ent_bytes >> 4;
  const uint4 *gtab = reinterpret_cast<const uint4 *>(gimg + table_off);
  cptr recs = img + lvl[L4_RECS];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const uint32_t tab0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)lds_tab;
  auto copy = [&](uint32_t grp) {
    const uint32_t g0 = grp * tg, n = G - g0 < tg ? G - g0 : tg;
    tile_copy(gtab + (size_t)g0 * ent_vec, lds_tab + (grp & 1u) * buf_bytes, n * ent_vec, tid, nthr);
  };
  Acc4 S;
  __syncthreads();  // previous users of the buffers are done
  if (ngroups) copy(0u);
  __syncthreads();
  for (uint32_t grp = 0; grp < ngroups; ++grp) {
    if (grp + 1u < ngroups) copy(grp + 1u);  // into the other buffer: its last readers passed the barrier that ended group grp - 1
    const uint32_t g0 = grp * tg, n = G - g0 < tg ? G - g0 : tg;
    uint32_t base = tab0 + (grp & 1u) * buf_bytes;
    for (uint32_t j = 0; j < n; ++j, base += ent_bytes) {
      uint32_t U = 0, V = 0, O1 = 0, O2 = 0;
#pragma unroll
      for (int c = 0; c < NR; c += 2) {
        const u32x4 v = *(lds_u4p)(uintptr_t)(base + ent0[c]);
        const u32x4 w = *(lds_u4p)(uintptr_t)(base + ent0[c + 1]);
        U = xor3(U, v.x, w.x); V = xor3(V, v.y, w.y); O1 = xor3(O1, v.z, w.z); O2 = xor3(O2, v.w, w.w);
        if (c % 6 == 4 && c + 2 < NR) __builtin_amdgcn_sched_barrier(0);
      }
      acc_graph4<FIXED>(S, gimg, recs + (g0 + j) * G4_WORDS, U, V, O1, O2, approx);
    }
    __syncthreads();
  }
  acc_finish4<FIXED>(S, lvl, approx, out_re, out_im);
}

// The same level with the column tables where the packer put them - in the program image (HBM, in practice the L2):
// components whose tables do not fit the LDS (many graphs: 140 graphs x 93 entries are 208 KB).  k_sample_wide<.., GLOB>.
template <int NR, bool FIXED>
__device__ __forceinline__ void eval_level4_global(const uint32_t *gimg, cptr img, cptr lvl, const uint32_t (&e)[NR], uint32_t ent_bytes,
                                                   float &out_re, float &out_im) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef const __attribute__((address_space(1))) u32x4 *glb_u4p;
  typedef const __attribute__((address_space(1))) uint8_t *glb_u8p;
  const uint32_t G = lvl[L4_G];
  const bool approx = (lvl[L4_FLAGS] & TSIMK_LFLAG_APPROX) != 0;
  cptr recs = img + lvl[L4_RECS];
  glb_u8p tab = (glb_u8p)(uintptr_t)(gimg + lvl[L4_STAB]);
  Acc4 S;
  for (uint32_t g = 0; g < G; ++g) {
    glb_u8p base = tab + (size_t)g * ent_bytes;
    uint32_t U = 0, V = 0, O1 = 0, O2 = 0;
#pragma unroll
    for (int c = 0; c < NR; c += 2) {
      const u32x4 v = *(glb_u4p)(base + e[c]);
      const u32x4 w = *(glb_u4p)(base + e[c + 1]);
      U = xor3(U, v.x, w.x); V = xor3(V, v.y, w.y); O1 = xor3(O1, v.z, w.z); O2 = xor3(O2, v.w, w.w);
    }
    acc_graph4<FIXED, false>(S, gimg, recs + g * G4_WORDS, U, V, O1, O2, approx, 0u);
  }
  acc_finish4<FIXED>(S, lvl, approx, out_re, out_im);
}

// GT is 1 in the wide layout: one graph per tile, a level's tables are G consecutive (F + 33) x 16-byte blocks.
template <int GT, int K>
__global__ void __launch_bounds__(256) k_sample4w(Wide4Args W) {
  static_assert(K % 2 == 0, "entries are consumed in pairs");
  static_assert(GT == 1, "the wide layout has one graph per tile");
  const SampleArgs &A = W.s;
  const int nthr = blockDim.x;
  long long n_rows = A.B;
  const uint32_t *rows_in = A.row_index;
  long long first = (long long)blockIdx.x * nthr, stride = (long long)gridDim.x * nthr;
  if (A.row_index && A.row_lists > 1) {
    const uint32_t lk = blockIdx.x % (uint32_t)A.row_lists;
    n_rows = (long long)A.row_count[32u * lk];
    rows_in = A.row_index + (size_t)lk * A.row_list_cap;
    first = (long long)(blockIdx.x / (uint32_t)A.row_lists) * nthr;
    stride = (long long)(gridDim.x / (uint32_t)A.row_lists) * nthr;
    if (W.feedback && blockIdx.x == 0 && threadIdx.x < 64) {  // launch-plan feedback, as k_sample4h
      uint32_t c = 0, m = 0;
      for (int k = (int)threadIdx.x; k < A.row_lists; k += 64) {
        const uint32_t v = A.row_count[32 * k];
        c += v;
        m = max(m, v);
      }
      for (int o = 32; o > 0; o >>= 1) {
        c += (uint32_t)__shfl_xor((int)c, o, 64);
        m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
      }
      if (threadIdx.x == 0) {
        W.feedback[0] = c;
        W.feedback[1] = m;
        W.feedback[2] = (uint32_t)min(A.B, 0xFFFFFFFFll);
      }
    }
  } else if (A.row_index) {
    n_rows = (long long)*A.row_count;
  }
  const uint32_t check_row = W.check_row_in ? *W.check_row_in : 0xFFFFFFFFu;
  cptr img = (cptr)(uintptr_t)A.img;

  if (blockIdx.x == 0 && threadIdx.x <= TSIMK_LW_LISTS)
    W.ctl_next[32u * threadIdx.x] = (threadIdx.x == TSIMK_LW_LISTS) ? 0xFFFFFFFFu : 0u;

  const int WF32 = 2 * A.WF, WO32 = 2 * A.WO;
  uint32_t *lds_f = tsimk_lds + threadIdx.x;                // [WF32][nthr]
  uint32_t *lds_o = tsimk_lds + WF32 * nthr + threadIdx.x;  // [WO32][nthr]
  uint8_t *lds_tab = reinterpret_cast<uint8_t *>(tsimk_lds + (WF32 + WO32) * nthr);

  // One component with LDS-resident tables (the usual wide program): the tables are copied ONCE per block and the
  // block then strides over the rows - the launch sizes the grid to what the chip holds at once.  The staging
  // columns are private to a thread, the tables read-only: no barrier between the row chunks.
  const bool preloaded = W.resident && A.n_comp == 1;
  if (first >= n_rows) return;  // nothing for this block (short input lists): no table copy, no barrier
  if (preloaded) {
    cptr comp = img + W.comp4_off;
    cptr levels = img + comp[C4_LEVELS];
    const uint32_t tile_bytes = (comp[C_F] + 33u) * (uint32_t)GT * 16u;
    uint32_t off = 0;
    for (uint32_t li = 0; li <= comp[C_NOUT]; ++li) {
      cptr lvl = levels + li * L4_WORDS;
      const uint32_t bytes = lvl[L4_G] * tile_bytes;
      tile_copy(reinterpret_cast<const uint4 *>(A.img + lvl[L4_STAB]), lds_tab + off, bytes >> 4, threadIdx.x, nthr);
      off += bytes;
    }
    __syncthreads();
  }

  for (long long base = first; base < n_rows; base += stride) {
  const long long slot = base + threadIdx.x;
  const bool active = slot < n_rows;
  long long row = slot;
  if (rows_in) row = active ? (long long)rows_in[slot] : 0;
  if (!active) row = 0;
  const unsigned long long shot = (unsigned long long)(A.shot_offset + row);

  if (active) {
    stage_f_row(A.f + row * A.WF, A.WF, lds_f, nthr);
  } else {
    for (int w = 0; w < WF32; ++w) lds_f[w * nthr] = 0u;
  }
  for (int w = 0; w < WO32; ++w) lds_o[w * nthr] = 0u;
  bool hard = active && W.has_check && (W.check_row_in ? (uint32_t)row == check_row : slot == 0);
  if (hard) W.ctl[32 * TSIMK_LW_LISTS] = (uint32_t)row;

  // direct outputs (sampler.py:140-145)
  direct_outputs(A, img, lds_f, lds_o, nthr);

  // Pass A over the components: more than K set bits anywhere -> the whole row is the row kernel's.
  for (int ci = 0; ci < A.n_comp; ++ci) {
    cptr comp = img + W.comp4_off + ci * C4_WORDS;
    const lw_u32x16 selm = *(lw_cptr16)(img + comp[C4_SELMASK]);  // one 64-byte scalar load, static indices below
    uint32_t cnt = 0;
    const int nw = WF32 < TSIMK_W_SELWORDS ? WF32 : TSIMK_W_SELWORDS;
#pragma unroll
    for (int w = 0; w < TSIMK_W_SELWORDS; ++w)
      if (w < nw) cnt += (uint32_t)__builtin_popcount(lds_f[w * nthr] & selm[w]);
    hard = hard || (cnt > (uint32_t)K);
  }

  for (int ci = 0; ci < A.n_comp; ++ci) {
    cptr comp = img + W.comp4_off + ci * C4_WORDS;
    const uint32_t n_out = comp[C_NOUT], F = comp[C_F];
    cptr levels = img + comp[C4_LEVELS];
    cptr outpos = img + comp[C_OUTPOS];
    cptr sel = img + comp[C4_SELMASK];
    const uint32_t keybase = comp[C_KEYBASE];
    constexpr uint32_t kEntry = GT * 16;

    // the lane's set f bits as column entries (F = the all-zero column); hard lanes ride along with empty lists
    uint32_t col[K];
#pragma unroll
    for (int k = 0; k < K; ++k) col[k] = F * kEntry;
    if (!hard) {
      int n = 0;
      (void)n;
      const int nw = WF32 < TSIMK_W_SELWORDS ? WF32 : TSIMK_W_SELWORDS;
      const lw_u32x16 selm = *(lw_cptr16)sel, selp = *(lw_cptr16)(sel + TSIMK_W_SELWORDS);
#pragma unroll
      for (int w = 0; w < TSIMK_W_SELWORDS; ++w) {
        if (w >= nw) continue;
        uint32_t mw = lds_f[w * nthr] & selm[w];
        const uint32_t sw = selm[w], base = selp[w];
        while (mw) {
          const uint32_t p = (uint32_t)__builtin_ctz(mw);
          const uint32_t pos = base + (uint32_t)__builtin_popcount(sw & ((1u << p) - 1u));
          // shift-register insert (static indices only: a compare chain on `n` would be recognised as dynamic
          // indexing and park the array in LDS); the order of the columns is irrelevant to the XOR
#pragma unroll
          for (int k = K - 1; k > 0; --k) col[k] = col[k - 1];
          col[0] = pos * kEntry;
          ++n;
          mw &= mw - 1u;
        }
      }
    }
    const uint32_t tile_bytes = (F + 33u) * kEntry;
    // resident mode: every level's tables into LDS now - one burst of async copies, one barrier per component
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)lds_tab;
    if (W.resident && !preloaded) {
      __syncthreads();  // the previous component's readers are done
      uint32_t off = 0;
      for (uint32_t li = 0; li <= n_out; ++li) {
        cptr lvl = levels + li * L4_WORDS;
        const uint32_t bytes = lvl[L4_G] * tile_bytes;
        tile_copy(reinterpret_cast<const uint4 *>(A.img + lvl[L4_STAB]), lds_tab + off, bytes >> 4, threadIdx.x, nthr);
        off += bytes;
      }
      __syncthreads();
    }
    uint32_t mb = 0;  // outcome bits so far (bit i = output i), trial bit included
    uint32_t lvl_off = 0;
    float prev = 0.0f;
    for (uint32_t li = 0; li <= n_out; ++li) {
      cptr lvl = levels + li * L4_WORDS;
      if (li > 0) mb |= 1u << (li - 1u);  // trial bit 1 (sampler.py:65)
      uint32_t e[K + 2];
#pragma unroll
      for (int k = 0; k < K; ++k) e[k] = col[k];
      e[K] = (F + 1u + (mb & 15u)) * kEntry;
      e[K + 1] = (F + 17u + ((mb >> 4) & 15u)) * kEntry;
      float re, im;
      const bool fixed = (lvl[L4_FLAGS] & TSIMK_LFLAG_FIXED) != 0;
      if (W.resident) {
        if (fixed) eval_level4_resident<K + 2, true>(A.img, img, lvl, e, lds0 + lvl_off, tile_bytes, re, im);
        else eval_level4_resident<K + 2, false>(A.img, img, lvl, e, lds0 + lvl_off, tile_bytes, re, im);
        lvl_off += lvl[L4_G] * tile_bytes;
      } else {
        if (fixed) eval_level4_groups<K + 2, true>(A.img, img, lvl, e, tile_bytes, lvl[L4_STAB], lds_tab, (uint32_t)W.stream_buf, re, im);
        else eval_level4_groups<K + 2, false>(A.img, img, lvl, e, tile_bytes, lvl[L4_STAB], lds_tab, (uint32_t)W.stream_buf, re, im);
      }
      const float v1 = cabs32(re, im);
      if (li == 0) { prev = v1; continue; }
      const uint32_t i = li - 1u;
      const float u = uniform01(subkey(A, keybase + i, 0), subkey(A, keybase + i, 1), shot);
      const bool bit = u < __fdiv_rn(v1, prev);  // sampler.py:74-75
      if (!bit) mb &= ~(1u << i);
      prev = bit ? v1 : __fsub_rn(prev, v1);
      const uint32_t dst = outpos[i];
      lds_o[(dst >> 5) * nthr] |= (bit ? 1u : 0u) << (dst & 31u);
    }
  }

  if (active && !hard) {
    if (A.out) {
      uint64_t *orow = A.out + row * A.WO;
      for (int w = 0; w < A.WO; ++w)
        orow[w] = (uint64_t)lds_o[(2 * w) * nthr] | ((uint64_t)lds_o[(2 * w + 1) * nthr] << 32);
    }
    store_compact_row(A, row, lds_o, nthr);
  }

  // wave-aggregated append of the rows left to the row kernel
  const unsigned long long hm = __ballot(hard ? 1 : 0);
  if (hm != 0ull) {
    const int lane = (int)(threadIdx.x & 63u);
    const int leader = __builtin_ctzll(hm);
    uint32_t basei = 0;
    const uint32_t k = blockIdx.x % (uint32_t)W.n_lists;
    if (lane == leader) basei = atomicAdd(&W.ctl[32u * k], (uint32_t)__popcll(hm));
    basei = (uint32_t)__shfl((int)basei, leader, 64);
    if (hard)
      W.hard_index[(size_t)k * W.list_cap + basei + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull))] = (uint32_t)row;
  }
  }  // row chunks of this block
}

}  // namespace tsimk
