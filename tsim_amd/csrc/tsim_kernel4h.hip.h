// tsim_kernel4h.hip.h - the chunk-table kernel for SHORT row lists (gfx950).
//
// The second pass of a two-pass launch (tsim_lw.hip.h) sees a few thousand rows: far too few to
// hide latency with occupancy, so k_sample4 (one wave = 64 shots walking every graph of every
// level in sequence) runs at the latency of one wave.  This variant puts NW waves on the SAME 64
// rows and splits the graphs of a level between them:
//   * a group of tiles (all of a level when it fits, ~120 KB) is brought into LDS by one
//     cooperative LDS-DMA burst;
//   * wave w forms Y_g (tsim_kernel4.hip.h) for the graphs g = w, w + NW, ... of the group, issues
//     all their term-table gathers back to back, then retires them: the gather latency is paid once
//     per group instead of once per graph;
//   * fixed-frame levels: the per-wave partial sums (plain int32 adds, order-free) are combined
//     through LDS.  Levels that keep the reference's aligned-add sequence or the float32
//     approximate-floatfactor sum are order-dependent: wave 0 evaluates them alone, in graph order.
// Every wave then repeats the identical scalar epilogue (canonicalise, to_complex, |.|, Threefry
// draw), so no further exchange is needed.  Same tables, same arithmetic, same bits as k_sample4.
#pragma once
#include "tsim_kernel4.hip.h"

namespace tsimk {

#define TSIMK_H_MAX_GROUP_TILES 12

template <int GT, int NCH, int NW>
struct Hard4 {
  static constexpr int kMaxPerWave = (TSIMK_H_MAX_GROUP_TILES * GT + NW - 1) / NW;
  static constexpr int kTileBytes = NCH * Tile4<GT>::kChunkBytes;
  static constexpr int kExchWords = NW * 8 * 64;
};

// index computation shared by both paths: table entry + rotation of graph `gr` for this lane's Y
__device__ __forceinline__ void h4_index(cptr gr, uint32_t U, uint32_t V, uint32_t O1, uint32_t O2, uint32_t &idx,
                                         uint32_t &dbits, uint32_t &r) {
  const uint32_t gflags = gr[G4_FLAGS];
  const bool z = (O1 & gr[G4_M0]) != 0;
  const uint32_t m1 = (uint32_t)__builtin_popcount(O1 & gr[G4_M1]);
  const uint32_t m3 = (uint32_t)__builtin_popcount(O1 & gr[G4_M3]);
  idx = m3 - m1 + gr[G4_N1];
  const uint32_t dsh = gr[G4_DBITS];
  dbits = O2 & ((1u << dsh) - 1u);
  if (gflags & TSIMK_G4FLAG_D_COMBINED) idx = (idx << dsh) | dbits;
  idx = z ? 0u : idx + 1u;
  const uint32_t pc = (uint32_t)__builtin_popcount(U & V & gr[G4_PM]);
  r = (O2 >> 30) ^ ((pc & 1u) << 1);
}

template <int NCH>
__device__ __forceinline__ void h4_form(const uint32_t (&ent)[NCH], uint32_t off, uint32_t &U, uint32_t &V,
                                        uint32_t &O1, uint32_t &O2) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef const __attribute__((address_space(3))) u32x4 *lds_u4p;
  U = V = O1 = O2 = 0;
#pragma unroll
  for (int c = 0; c < NCH; c += 2) {
    const u32x4 v = *(lds_u4p)(uintptr_t)(ent[c] + off);
    const u32x4 w = *(lds_u4p)(uintptr_t)(ent[c + 1] + off);
    U = xor3(U, v.x, w.x); V = xor3(V, v.y, w.y); O1 = xor3(O1, v.z, w.z); O2 = xor3(O2, v.w, w.w);
  }
}

// evaluate() of one level; every thread of the block calls it, every wave returns the same value
template <int GT, int NCH, int NW>
__device__ __forceinline__ void eval_level4h(const uint32_t *gimg, cptr img, cptr lvl, const uint32_t (&ent0)[NCH],
                                             uint8_t *lds_tab, uint32_t *exch, int group_tiles, float &out_re,
                                             float &out_im) {
  typedef Hard4<GT, NCH, NW> H;
  const uint32_t G = lvl[L4_G], ntiles = lvl[L4_NTILES];
  const uint32_t lflags = lvl[L4_FLAGS];
  const bool approx = (lflags & TSIMK_LFLAG_APPROX) != 0;
  const bool fixed = (lflags & TSIMK_LFLAG_FIXED) != 0 && !approx;
  const uint32_t tile_vec = H::kTileBytes >> 4;
  const uint4 *gtab = reinterpret_cast<const uint4 *>(gimg + lvl[L4_TABLES]);
  cptr recs = img + lvl[L4_RECS];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int wave = tid >> 6, lane = tid & 63;

  const uint32_t tab0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)lds_tab;
  uint32_t ent[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) ent[c] = tab0 + ent0[c];

  int sa = 0, sb = 0, sc = 0, sd = 0, sp = TSIMK_ZERO_POWER;
  float fre = 0.0f, fim = 0.0f;

  for (uint32_t t0 = 0; t0 < ntiles; t0 += (uint32_t)group_tiles) {
    const uint32_t nt = min((uint32_t)group_tiles, ntiles - t0);
    __syncthreads();  // the previous group's readers are done
    tile_copy(gtab + (size_t)t0 * tile_vec, lds_tab, nt * tile_vec, tid, nthr);
    __syncthreads();  // LDS-DMA landed and visible
    const uint32_t gbeg = t0 * GT, gend = min(G, (t0 + nt) * GT);
    if (fixed) {
      // this wave's graphs: form all, gather all, then retire all
      uint4 tv[H::kMaxPerWave], dv[H::kMaxPerWave];
#pragma unroll
      for (int q = 0; q < H::kMaxPerWave; ++q) {
        const uint32_t g = gbeg + (uint32_t)wave + (uint32_t)q * NW;
        tv[q] = {0u, 0u, 0u, 0u};
        dv[q] = {1u, 0u, 0u, 0u};
        if (g < gend) {
          const uint32_t gl = g - gbeg;
          uint32_t U, V, O1, O2, idx, dbits, r;
          h4_form<NCH>(ent, (gl / GT) * H::kTileBytes + (gl % GT) * 256u, U, V, O1, O2);
          cptr gr = recs + g * G4_WORDS;
          h4_index(gr, U, V, O1, O2, idx, dbits, r);
          tv[q] = *reinterpret_cast<const uint4 *>(gimg + gr[G4_TBL] + 16u * idx + 4u * r);
          if (gr[G4_FLAGS] & TSIMK_G4FLAG_D_SEPARATE)
            dv[q] = *reinterpret_cast<const uint4 *>(gimg + gr[G4_TBL2] + 8u * dbits);
        }
      }
#pragma unroll
      for (int q = 0; q < H::kMaxPerWave; ++q) {
        const uint32_t g = gbeg + (uint32_t)wave + (uint32_t)q * NW;
        if (g < gend) {
          int a = (int)tv[q].x, b = (int)tv[q].y, c = (int)tv[q].z, d = (int)tv[q].w;
          if (recs[g * G4_WORDS + G4_FLAGS] & TSIMK_G4FLAG_D_SEPARATE)
            zmul(a, b, c, d, (int)dv[q].x, (int)dv[q].y, (int)dv[q].z, (int)dv[q].w);
          sa += a; sb += b; sc += c; sd += d;
        }
      }
    } else if (wave == 0) {
      // order-dependent accumulation (exact_scalar.py:74-84,173-189 / evaluate.py:56-59): graph order
      for (uint32_t g = gbeg; g < gend; ++g) {
        const uint32_t gl = g - gbeg;
        uint32_t U, V, O1, O2, idx, dbits, r;
        h4_form<NCH>(ent, (gl / GT) * H::kTileBytes + (gl % GT) * 256u, U, V, O1, O2);
        cptr gr = recs + g * G4_WORDS;
        const uint32_t gflags = gr[G4_FLAGS];
        h4_index(gr, U, V, O1, O2, idx, dbits, r);
        const uint32_t *te = gimg + gr[G4_TBL] + 8u * idx;
        const uint4 t4 = *reinterpret_cast<const uint4 *>(te);
        int a = (int)t4.x, b = (int)t4.y, c = (int)t4.z, d = (int)t4.w, p = (int)te[4];
        if (gflags & TSIMK_G4FLAG_D_SEPARATE) {
          const uint32_t *td = gimg + gr[G4_TBL2] + 8u * dbits;
          const uint4 d4 = *reinterpret_cast<const uint4 *>(td);
          zmul(a, b, c, d, (int)d4.x, (int)d4.y, (int)d4.z, (int)d4.w);
          p += (int)td[4];
        }
        {  // rotate by i^r
          const bool k2 = (r & 1u) != 0;
          const int q0 = k2 ? -c : a, q1 = k2 ? d : b, q2 = k2 ? a : c, q3 = k2 ? -b : d;
          const int nm = -(int)((r >> 1) & 1u);
          a = (q0 ^ nm) - nm; b = (q1 ^ nm) - nm; c = (q2 ^ nm) - nm; d = (q3 ^ nm) - nm;
        }
        if (!approx) {
          if ((a | b | c | d) == 0) p = TSIMK_ZERO_POWER;
          const int d1 = max(sp - p, 0), d2 = max(p - sp, 0);
          sa = (int)((unsigned)shl_sat(sa, d1) + (unsigned)shl_sat(a, d2));
          sb = (int)((unsigned)shl_sat(sb, d1) + (unsigned)shl_sat(b, d2));
          sc = (int)((unsigned)shl_sat(sc, d1) + (unsigned)shl_sat(c, d2));
          sd = (int)((unsigned)shl_sat(sd, d1) + (unsigned)shl_sat(d, d2));
          sp = min(sp, p);
          reduce1(sa, sb, sc, sd, sp);
        } else {
          float zr, zi;
          to_complex(a, b, c, d, p, zr, zi);
          const float ar = __uint_as_float(gr[G4_APRE]), ai = __uint_as_float(gr[G4_APIM]);
          const float tr = __fsub_rn(__fmul_rn(zr, ar), __fmul_rn(zi, ai));
          const float ti = __fadd_rn(__fmul_rn(zr, ai), __fmul_rn(zi, ar));
          fre = __fadd_rn(fre, tr);
          fim = __fadd_rn(fim, ti);
        }
      }
    }
  }

  // combine: exch[wave][field][lane]
  __syncthreads();  // earlier readers of exch are done (and covers levels without tiles)
  uint32_t *mine = exch + (wave * 8) * 64 + lane;
  mine[0 * 64] = (uint32_t)sa; mine[1 * 64] = (uint32_t)sb; mine[2 * 64] = (uint32_t)sc; mine[3 * 64] = (uint32_t)sd;
  mine[4 * 64] = (uint32_t)sp; mine[5 * 64] = __float_as_uint(fre); mine[6 * 64] = __float_as_uint(fim);
  __syncthreads();
  if (fixed) {
    sa = sb = sc = sd = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const uint32_t *o = exch + (w * 8) * 64 + lane;
      sa += (int)o[0 * 64]; sb += (int)o[1 * 64]; sc += (int)o[2 * 64]; sd += (int)o[3 * 64];
    }
    sp = (int)lvl[L4_FRAME];
  } else {
    const uint32_t *o = exch + lane;  // wave 0
    sa = (int)o[0 * 64]; sb = (int)o[1 * 64]; sc = (int)o[2 * 64]; sd = (int)o[3 * 64]; sp = (int)o[4 * 64];
    fre = __uint_as_float(o[5 * 64]); fim = __uint_as_float(o[6 * 64]);
  }
  if (!approx) {
    canon(sa, sb, sc, sd, sp);
    if ((sa | sb | sc | sd) == 0) sp = 0;
    to_complex(sa, sb, sc, sd, sp, out_re, out_im);
  } else {
    out_re = fre;
    out_im = fim;
  }
}

// sample_program on row lists, 64 rows per block, NW waves per row group; block `bidx` of `nblk`
template <int GT, int NCH, int NW>
__device__ __forceinline__ void sample4h_rows(const SampleArgs &A, int comp4_off, bool has_check, int group_tiles,
                                              int loop_stride, uint32_t *feedback, uint32_t bidx, uint32_t nblk) {
  typedef Hard4<GT, NCH, NW> H;
#ifndef TSIMK_HARD_PRIO
#define TSIMK_HARD_PRIO 3
#endif
  // The hard rows are a few hundred waves of latency-bound work beside a chip-full of issue-bound first-pass waves:
  // when one of these waves can issue, it should (s_setprio: issue arbitration is by priority, then age).
  if (TSIMK_HARD_PRIO) __builtin_amdgcn_s_setprio(TSIMK_HARD_PRIO);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool check_block = has_check && (bidx == nblk - 1);
  // feedback to the host (mapped pinned memory, read at later launches to choose the launch plan):
  // total and longest hard-row list of THIS launch
  if (feedback && bidx == 0 && wave == 0 && A.row_lists > 1) {
    uint32_t c = lane < A.row_lists ? A.row_count[32u * lane] : 0u, m = c;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      c += (uint32_t)__shfl_xor((int)c, o, 64);
      m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    }
    if (lane == 0) {
      feedback[0] = c;
      feedback[1] = m;
      feedback[2] = (uint32_t)min(A.B, 0xFFFFFFFFll);
    }
  }
  // loop_stride > 0: the block walks its list in steps of loop_stride rows (no overflow kernel)
  for (long long iter_base = 0;; iter_base += loop_stride) {
  long long row = (long long)bidx * 64 + lane;
  bool active = row < A.B;
  if (A.row_index) {
    const uint32_t nl = A.row_lists > 1 ? (uint32_t)A.row_lists : 1u;
    const uint32_t k = bidx % nl;
    const long long base = (long long)(bidx / nl) * 64 + iter_base;
    long long n = (long long)A.row_count[32u * k * (nl > 1 ? 1u : 0u)];
    if (A.row_slot_end > 0) n = min(n, (long long)A.row_slot_end);
    if (!check_block && base >= n) return;  // block-uniform
    row = base + lane;
    active = row < n;
    if (check_block) {  // lanes 0 and 1 both replay the check row (trial bit 1 / trial bit 0)
      active = (lane < 2) && (A.check_row ? (*A.check_row != 0xFFFFFFFFu) : (n > 0));
      row = !active ? 0 : A.check_row ? (long long)*A.check_row : (long long)A.row_index[0];
    } else {
      row = active ? (long long)A.row_index[(size_t)k * (nl > 1 ? A.row_list_cap : 0) + row] : 0;
    }
  } else if (check_block) {
    row = 0;
    active = (lane < 2);
  }
  // normalisation check (sampler.py:66-72): lane 1 of the check block evaluates every level with
  // trial bit 0 while lane 0 evaluates it with trial bit 1 - one pass gives both values
  const bool trial0 = check_block && lane == 1;
  const unsigned long long shot = (unsigned long long)(A.shot_offset + row);
  cptr img = (cptr)(uintptr_t)A.img;

  const int WF32 = 2 * A.WF, WO32 = 2 * A.WO;
  uint32_t *lds_f = tsimk_lds + lane;                  // [WF32][64]
  uint32_t *lds_o = tsimk_lds + WF32 * 64 + lane;      // [WO32][64]
  uint32_t *exch = tsimk_lds + (WF32 + WO32) * 64;     // [NW][8][64]
  uint8_t *lds_tab = reinterpret_cast<uint8_t *>(exch + H::kExchWords);

  if (wave == 0) {
    if (active) {
      stage_f_row(A.f + row * A.WF, A.WF, lds_f, 64);
    } else {
      for (int w = 0; w < WF32; ++w) lds_f[w * 64] = 0u;
    }
    for (int w = 0; w < WO32; ++w) lds_o[w * 64] = 0u;
    direct_outputs(A, img, lds_f, lds_o, 64);
  }
  __syncthreads();

  for (int ci = 0; ci < A.n_comp; ++ci) {
    cptr comp = img + comp4_off + ci * C4_WORDS;
    const uint32_t n_out = comp[C_NOUT], F = comp[C_F];
    cptr fsel = img + comp[C_FSEL];
    cptr levels = img + comp[C4_LEVELS];
    cptr outpos = img + comp[C_OUTPOS];
    const uint32_t keybase = comp[C_KEYBASE];

    constexpr int XW = NCH > 24 ? 4 : NCH > 16 ? 3 : 2;  // (words of x: tsim_kernel4.hip.h sample4_block)
    uint32_t x[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int w = 0; w < XW; ++w) {
      uint32_t v = 0;
      const int lo = w * 32;
      const int hi = min((int)F, lo + 32);
      for (int j = lo; j < hi; ++j) {
        const uint32_t src = fsel[j];
        v |= ((lds_f[(src >> 5) * 64] >> (src & 31u)) & 1u) << (j - lo);
      }
      x[w] = v;
    }
    auto set_bit = [&](uint32_t bitpos, bool v) {
      const uint32_t bm = 1u << (bitpos & 31u), bw = bitpos >> 5;
      if (bw == 0u) x[0] = v ? (x[0] | bm) : (x[0] & ~bm);
      else if (bw == 1u || XW == 2) x[1] = v ? (x[1] | bm) : (x[1] & ~bm);
      else if (bw == 2u || XW == 3) x[2] = v ? (x[2] | bm) : (x[2] & ~bm);
      else x[3] = v ? (x[3] | bm) : (x[3] & ~bm);
    };

    float prev = 0.0f, maxdev = 0.0f;
    for (uint32_t li = 0; li <= n_out; ++li) {
      cptr lvl = levels + li * L4_WORDS;
      const uint32_t bitpos = F + li - 1u;
      if (li > 0) set_bit(bitpos, !trial0);
      uint32_t en[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const uint32_t w = (c < 8) ? x[0] : (c < 16) ? x[1] : (c < 24) ? x[2] : x[3];
        en[c] = ((w >> (4 * (c & 7))) & 15u) * 16u + c * Tile4<GT>::kChunkBytes;
      }
      float re, im;
      eval_level4h<GT, NCH, NW>(A.img, img, lvl, en, lds_tab, exch, group_tiles, re, im);
      float v1 = cabs32(re, im), v0 = 0.0f;
      if (check_block) {  // block-uniform
        v0 = __shfl(v1, 1, 64);
        v1 = __shfl(v1, 0, 64);
      }
      if (li == 0) { prev = v1; continue; }
      const uint32_t i = li - 1u;
      const float p1 = v1;
      if (check_block) {
        const float norm = __fdiv_rn(__fadd_rn(v0, p1), prev);      // sampler.py:71
        maxdev = nanmax(maxdev, fabsf(__fsub_rn(norm, 1.0f)));      // sampler.py:72
      }
      const float u = uniform01(subkey(A, keybase + i, 0), subkey(A, keybase + i, 1), shot);
      const bool bit = u < __fdiv_rn(p1, prev);
      set_bit(bitpos, bit);
      prev = bit ? p1 : __fsub_rn(prev, p1);
      if (wave == 0) {
        const uint32_t dst = outpos[i];
        lds_o[(dst >> 5) * 64] |= (bit ? 1u : 0u) << (dst & 31u);
      }
    }
    if (check_block && threadIdx.x == 0 && active && A.norm_dev) A.norm_dev[ci] = maxdev;
  }

  if (wave == 0 && active && !check_block) {
    if (A.out) {
      uint64_t *orow = A.out + row * A.WO;
      for (int w = 0; w < A.WO; ++w)
        orow[w] = (uint64_t)lds_o[(2 * w) * 64] | ((uint64_t)lds_o[(2 * w + 1) * 64] << 32);
    }
    store_compact_row(A, row, lds_o, 64);
  }
  if (check_block || loop_stride <= 0 || !A.row_index) return;
  __syncthreads();  // the LDS staging of this row group is reused by the next one
  }
}

template <int GT, int NCH, int NW>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(3, 8))) k_sample4h(Sample4Args A4, int group_tiles, int loop_stride,
                                                        uint32_t *feedback) {
  sample4h_rows<GT, NCH, NW>(A4.s, A4.comp4_off, A4.has_check != 0, group_tiles, loop_stride, feedback, blockIdx.x,
                             gridDim.x);
}

// The hard rows of SEVERAL launches in one grid (deferred second pass, tsim_hip.hip): launch c owns
// the blocks [c * blocks_per_ctx, (c + 1) * blocks_per_ctx), the last of them its check block.
#define TSIMK_H_MAX_CTX 8
struct Hard4Multi {
  int n_ctx, blocks_per_ctx, group_tiles, loop_stride;
  int comp4_off, check_mask;
  uint32_t *feedback;
  uint32_t main_blocks, over_from;  // blocks of the NW-waves-per-64-rows part; the grid's further blocks are per-shot workers for the list slots from over_from on (over4_rows, tsim_kernel4.hip.h)
  SampleArgs ctx[TSIMK_H_MAX_CTX];
};

template <int GT, int NCH, int NW>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(3, 8))) k_sample4h_multi(Hard4Multi M) {
  if (blockIdx.x >= M.main_blocks) {  // block-uniform
    over4_rows<GT, NCH>(M.ctx, M.n_ctx, M.comp4_off, M.over_from, false, blockIdx.x - M.main_blocks, gridDim.x - M.main_blocks);
    return;
  }
  const uint32_t c = blockIdx.x / (uint32_t)M.blocks_per_ctx;
  const uint32_t bidx = blockIdx.x - c * (uint32_t)M.blocks_per_ctx;
  const bool has_check = ((M.check_mask >> c) & 1) != 0;
  // without a check row the launch's last block is simply idle
  if (!has_check && bidx == (uint32_t)M.blocks_per_ctx - 1u) return;
  // the grid is sized for the launch with the most lists: blocks beyond this launch's own are idle
  if (bidx + 1u < (uint32_t)M.blocks_per_ctx && bidx >= (uint32_t)(M.loop_stride / 64) * (uint32_t)M.ctx[c].row_lists) return;
  sample4h_rows<GT, NCH, NW>(M.ctx[c], M.comp4_off, has_check, M.group_tiles, M.loop_stride, c == 0 ? M.feedback : nullptr,
                             bidx, (uint32_t)M.blocks_per_ctx);
}

}  // namespace tsimk
