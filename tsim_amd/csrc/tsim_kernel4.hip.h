// tsim_kernel4.hip.h - "Four Russians" sampling kernel (k_sample4) for gfx950.
//
// Same results as k_sample<., true> (the fast exact formulation), different way of getting the
// GF(2) parities.  For one graph (stabiliser term) every row parity <r, x> the formulation needs
// is ONE BIT of a 128-bit word Y_g = R_g x.  The packer cuts x into 4-bit chunks and tabulates, per
// chunk c and chunk value v, the partial product R_g[:, 4c..4c+3] v (16 bytes per graph).  A tile
// of GT graphs shares one LDS table [chunk][v][graph] (GT*256 bytes per chunk); a lane forms
//     Y_g = XOR_c  TABLE[c][ (x >> 4c) & 15 ][g]                       (one ds_read_b128 per chunk)
// and then consumes the bits word-parallel:
//     word U, V : the Dickson product pairs  ->  e = parity(popcount(U & V & PM))
//     word O1   : counted NodePhases rows    ->  m0 = any(O1 & M0), m1 = popc(O1 & M1), m3 = popc(O1 & M3)
//     word O2   : PhasePairs index bits, lambda (bit 30), linear bit (bit 31)
// The per-graph VALU work drops from ~3 ops per ROW to ~2 XORs per CHUNK plus ~45 ops.  Row
// constants are folded into chunk 0's entries.  Tables of the next tile are prefetched from L2 into
// registers while the current tile is consumed from LDS (one barrier per tile).
//
// All lanes of a block walk the levels in lock step (they share the LDS tables), so the
// normalisation check of sampler.py:66-72 - a second evaluation per level for in-batch shot 0 - is
// done by ONE EXTRA BLOCK that replays shot 0 alone and evaluates both trial values.
#pragma once
#include "tsim_kernels.hip.h"

namespace tsimk {

// level record of the v4 layout
enum { L4_G = 0, L4_NTILES, L4_TABLES, L4_RECS, L4_NCH, L4_FLAGS, L4_FRAME, L4_STAB, L4_WORDS = 8 };
// graph record of the v4 layout
enum {
  // the first eight words are everything a fixed-frame graph needs: one s_load_dwordx8
  G4_M0 = 0, G4_M1, G4_M3, G4_PM, G4_N1, G4_DBITS /* 2*nD */, G4_TBL, G4_FLAGS, G4_TBL2,
  G4_CFIELD = 9 /* LDS copies of k_sample_wide only: the graph's field of the shared column entry (WR_CREC) */,
  G4_APRE = 11, G4_APIM = 12, G4_WORDS = 16
};
#define TSIMK_G4FLAG_D_COMBINED 1u
#define TSIMK_G4FLAG_D_SEPARATE 2u
// component record extension (words appended to the C_* record)
enum { C4_LEVELS = 8, C4_WORDS = 16 };

#ifndef TSIMK_V4_WAVES
#define TSIMK_V4_WAVES 1  // no forced register budget (forcing 6 waves/SIMD spills ~100 B/lane to scratch)
#endif

struct Sample4Args {
  SampleArgs s;
  int comp4_off;   // offset of the v4 component records (C4_WORDS each)
  int has_check;   // 1: the last block replays in-batch shot 0 for the normalisation check
  uint32_t *feedback;  // optional (two-pass launches served by this kernel alone): [hard rows, longest list, rows]
};

// a ^ b ^ c in one full-rate VALU op
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}

// Linear global -> LDS copy of n16 uint4 elements by the whole block with the async LDS-DMA path
// (global_load_lds_dwordx4: per-lane source, wave-uniform LDS base + lane*16).
__device__ __forceinline__ void tile_copy(const uint4 *src, uint8_t *lds_dst, uint32_t n16, int tid, int nthr) {
  typedef __attribute__((address_space(3))) void *lds_ptr_t;
  typedef const __attribute__((address_space(1))) void *glb_ptr_t;
  const uint32_t lane = tid & 63;
  for (uint32_t base = (uint32_t)(tid & ~63); base < n16; base += nthr)
    if (base + lane < n16)
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + base + lane), (lds_ptr_t)(lds_dst + (size_t)base * 16), 16, 0, 0);
}

template <int GT>
struct Tile4 {
  static constexpr int kChunkBytes = 16 * GT * 16;  // 16 chunk values x GT graphs x 16 B
};

// Running value of a level while its graphs are consumed one by one: the exact sum (or the float sum of the
// approximate branch) plus, in fixed-frame levels, the one table entry still in flight.
struct Acc4 {
  int sa = 0, sb = 0, sc = 0, sd = 0, sp = TSIMK_ZERO_POWER;
  float fre = 0.0f, fim = 0.0f;
  uint4 pend_tv = {0u, 0u, 0u, 0u}, pend_dv = {1u, 0u, 0u, 0u};
  bool pend_sep = false;
};

// One graph: from its 128-bit parity word Y_g = (U, V, O1, O2) to its term-table entry, added to the level's value.
//   word U, V : the Dickson product pairs  ->  e = parity(popcount(U & V & PM))
//   word O1   : counted NodePhases rows    ->  m0 = any(O1 & M0), m1 = popc(O1 & M1), m3 = popc(O1 & M3)
//   word O2   : PhasePairs index bits, lambda (bit 30), linear bit (bit 31)
// Fixed-frame levels: the table gather of graph g is consumed while graph g+1 is being formed (software pipelining
// of the VMEM latency).
// LT: the term tables of the level were copied to LDS (k_sample_wide); word offset w of the image is then the LDS byte
// address tt_bias + 4 w - no trip to L2 on the dependency chain of a level.
// GR: where the graph record comes from - `cptr` (the image, scalar loads) or anything with operator[] over the G4_* words
// (k_sample_wide keeps a copy of the records in LDS: no scalar-memory latency on the chain of a level).
template <bool FIXED, bool LT = false, class GR = cptr>
__device__ __forceinline__ void acc_graph4(Acc4 &S, const uint32_t *gimg, GR gr, uint32_t U, uint32_t V, uint32_t O1,
                                           uint32_t O2, bool approx, uint32_t tt_bias = 0u) {
  constexpr bool fixed = FIXED;
  typedef uint32_t tt_u32x4 __attribute__((ext_vector_type(4)));
  auto ld4 = [&](uint32_t w) -> uint4 {
    if constexpr (LT) {
      const tt_u32x4 v = *(const __attribute__((address_space(3))) tt_u32x4 *)(uintptr_t)(tt_bias + 4u * w);
      return make_uint4(v.x, v.y, v.z, v.w);
    } else {
      // (a buffer descriptor over the image and a 32-bit byte offset: two instructions of address arithmetic per gather instead of
      // seven 64-bit ones - profiles/r06/full_kernel_floor.txt; the image is below 4 GB: tsim_program_finalize refuses 2^30 words)
      const __amdgpu_buffer_rsrc_t r_img = __builtin_amdgcn_make_buffer_rsrc((void *)gimg, 0, 0xFFFFFFFF, 0x00020000);
      const tt_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r_img, 4u * w, 0, 0);
      return make_uint4(v.x, v.y, v.z, v.w);
    }
  };
  auto ld1 = [&](uint32_t w) -> uint32_t {
    if constexpr (LT) return *(const __attribute__((address_space(3))) uint32_t *)(uintptr_t)(tt_bias + 4u * w);
    else return gimg[w];
  };
  const uint32_t gflags = gr[G4_FLAGS];
  // ---- counted NodePhases rows and the table index ----
  const bool z = (O1 & gr[G4_M0]) != 0;
  const uint32_t m1 = (uint32_t)__builtin_popcount(O1 & gr[G4_M1]);
  const uint32_t m3 = (uint32_t)__builtin_popcount(O1 & gr[G4_M3]);
  uint32_t idx = m3 - m1 + gr[G4_N1];
  const uint32_t dsh = gr[G4_DBITS];
  const uint32_t dbits = O2 & ((1u << dsh) - 1u);
  if (gflags & TSIMK_G4FLAG_D_COMBINED) idx = (idx << dsh) | dbits;
  idx = z ? 0u : idx + 1u;
  // ---- exponent of w: k = 2 lambda + 4 (lin ^ parity(U & V & PM)); r = k / 2 ----
  const uint32_t pc = (uint32_t)__builtin_popcount(U & V & gr[G4_PM]);
  const uint32_t r = (O2 >> 30) ^ ((pc & 1u) << 1);
  // fixed-frame levels: 4 pre-rotated copies per entry (value * i^r), 16 words per entry
  const uint32_t te = gr[G4_TBL] + (fixed ? (16u * idx + 4u * r) : 8u * idx);
  const uint4 tv = ld4(te);
  if constexpr (fixed) {
    uint4 dv = {1u, 0u, 0u, 0u};
    const bool sep = (gflags & TSIMK_G4FLAG_D_SEPARATE) != 0;
    if (sep) dv = ld4(gr[G4_TBL2] + 8u * dbits);
    // retire the previous graph's entry (zero-initialised before the first graph)
    {
      int a = (int)S.pend_tv.x, b = (int)S.pend_tv.y, c = (int)S.pend_tv.z, d = (int)S.pend_tv.w;
      if (S.pend_sep) zmul(a, b, c, d, (int)S.pend_dv.x, (int)S.pend_dv.y, (int)S.pend_dv.z, (int)S.pend_dv.w);
      S.sa += a; S.sb += b; S.sc += c; S.sd += d;
    }
    S.pend_tv = tv; S.pend_dv = dv; S.pend_sep = sep;
    return;
  }
  int a = (int)tv.x, b = (int)tv.y, c = (int)tv.z, d = (int)tv.w, p = 0;
  if (!fixed) p = (int)ld1(te + 4u);
  if (gflags & TSIMK_G4FLAG_D_SEPARATE) {
    const uint32_t td = gr[G4_TBL2] + 8u * dbits;
    const uint4 dv = ld4(td);
    zmul(a, b, c, d, (int)dv.x, (int)dv.y, (int)dv.z, (int)dv.w);
    if (!fixed) p += (int)ld1(td + 4u);
  }
  if (!fixed) {  // rotate by i^r
    const bool k2 = (r & 1u) != 0;
    const int t0 = k2 ? -c : a, t1 = k2 ? d : b, t2 = k2 ? a : c, t3 = k2 ? -b : d;
    const int nm = -(int)((r >> 1) & 1u);
    a = (t0 ^ nm) - nm; b = (t1 ^ nm) - nm; c = (t2 ^ nm) - nm; d = (t3 ^ nm) - nm;
  }
  if (!approx) {
    if ((a | b | c | d) == 0) p = TSIMK_ZERO_POWER;
    const int d1 = max(S.sp - p, 0), d2 = max(p - S.sp, 0);
    S.sa = (int)((unsigned)shl_sat(S.sa, d1) + (unsigned)shl_sat(a, d2));
    S.sb = (int)((unsigned)shl_sat(S.sb, d1) + (unsigned)shl_sat(b, d2));
    S.sc = (int)((unsigned)shl_sat(S.sc, d1) + (unsigned)shl_sat(c, d2));
    S.sd = (int)((unsigned)shl_sat(S.sd, d1) + (unsigned)shl_sat(d, d2));
    S.sp = min(S.sp, p);
    reduce1(S.sa, S.sb, S.sc, S.sd, S.sp);
  } else {
    float zr, zi;
    to_complex(a, b, c, d, p, zr, zi);
    const float ar = __uint_as_float(gr[G4_APRE]), ai = __uint_as_float(gr[G4_APIM]);
    const float tr = __fsub_rn(__fmul_rn(zr, ar), __fmul_rn(zi, ai));
    const float ti = __fadd_rn(__fmul_rn(zr, ai), __fmul_rn(zi, ar));
    S.fre = __fadd_rn(S.fre, tr);
    S.fim = __fadd_rn(S.fim, ti);
  }
}

// the level's amplitude once every graph has been added
template <bool FIXED>
__device__ __forceinline__ void acc_finish4(Acc4 &S, int frame, bool approx, float &out_re, float &out_im);
template <bool FIXED>
__device__ __forceinline__ void acc_finish4(Acc4 &S, cptr lvl, bool approx, float &out_re, float &out_im) {
  acc_finish4<FIXED>(S, FIXED ? (int)lvl[L4_FRAME] : 0, approx, out_re, out_im);
}
template <bool FIXED>
__device__ __forceinline__ void acc_finish4(Acc4 &S, int frame, bool approx, float &out_re, float &out_im) {
  if constexpr (FIXED) {  // retire the last in-flight entry
    int a = (int)S.pend_tv.x, b = (int)S.pend_tv.y, c = (int)S.pend_tv.z, d = (int)S.pend_tv.w;
    if (S.pend_sep) zmul(a, b, c, d, (int)S.pend_dv.x, (int)S.pend_dv.y, (int)S.pend_dv.z, (int)S.pend_dv.w);
    S.sa += a; S.sb += b; S.sc += c; S.sd += d;
  }
  if (!approx) {
    if (FIXED) S.sp = frame;
    canon(S.sa, S.sb, S.sc, S.sd, S.sp);
    if ((S.sa | S.sb | S.sc | S.sd) == 0) S.sp = 0;
    to_complex(S.sa, S.sb, S.sc, S.sd, S.sp, out_re, out_im);
  } else {
    out_re = S.fre;
    out_im = S.fim;
  }
}

// evaluate() of one level with LDS chunk tables.  All threads of the block must call this together.
// NR = LDS reads per graph: the caller passes this lane's NR entry offsets inside a tile table
// (`ent0`, bytes from the start of the tile), the tile size and the table's offset in the image.
// `gstride`: bytes between the entries of consecutive graphs of a tile (256 in the chunk tables - [chunk][graph][value] -,
// entries x 16 in the column tables - [graph][entry]).
// (a template parameter: the per-graph offsets are then immediates of the LDS reads - no address arithmetic per read)
#define TSIMK_SPARSE_ENTRIES 128  // entries per graph of a GT > 1 column-table tile (F + 33 <= 97 used, padded: constant stride)
template <int GT, int NR, bool FIXED, int GSTRIDE = 16>
__device__ __forceinline__ void eval_level4(const uint32_t *gimg, cptr img, cptr lvl, const uint32_t (&ent0)[NR],
                                            uint32_t tile_bytes, uint32_t table_off,
                                            uint8_t *lds_tab /* 2 * tile bytes */, float &out_re, float &out_im) {
  constexpr uint32_t gstride = GSTRIDE;
  const uint32_t G = lvl[L4_G], ntiles = lvl[L4_NTILES];
  const bool approx = (lvl[L4_FLAGS] & TSIMK_LFLAG_APPROX) != 0;
  const uint32_t tile_vec = tile_bytes >> 4;  // uint4 elements per tile
  const uint4 *gtab = reinterpret_cast<const uint4 *>(gimg + table_off);
  cptr recs = img + lvl[L4_RECS];
  const int tid = threadIdx.x, nthr = blockDim.x;

  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef const __attribute__((address_space(3))) u32x4 *lds_u4p;  // 32-bit LDS addresses
  const uint32_t tab0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)lds_tab;
  uint32_t ent[NR];
#pragma unroll
  for (int c = 0; c < NR; ++c) ent[c] = tab0 + ent0[c];
  int delta = (int)tile_bytes;  // +tile_bytes / -tile_bytes: toggles ent[] between the two buffers

  Acc4 S;

  // tile 0 -> LDS buffer 0 (async global->LDS copies, 1 KiB per wave-instruction)
  __syncthreads();  // previous users of the buffers are done
  if (ntiles) tile_copy(gtab, lds_tab, tile_vec, tid, nthr);
  __syncthreads();

  for (uint32_t t = 0; t < ntiles; ++t) {
    // prefetch tile t+1 straight into the other LDS buffer: its last readers passed the barrier
    // that ended iteration t-1; the barrier ending this iteration makes it visible
    const bool more = (t + 1 < ntiles);
    if (more) tile_copy(gtab + (size_t)(t + 1) * tile_vec, lds_tab + ((t + 1) & 1u) * tile_bytes, tile_vec, tid, nthr);
#pragma unroll
    for (int j = 0; j < GT; ++j) {
      const uint32_t g = t * GT + j;
      if (g >= G) break;
      // ---- Y_g = XOR over chunks of the lane's table entries (3-input XORs: v_bitop3 0x96) ----
      uint32_t U = 0, V = 0, O1 = 0, O2 = 0;
      static_assert(NR % 2 == 0, "entries are consumed in pairs");
#pragma unroll
      for (int c = 0; c < NR; c += 2) {
        const u32x4 v = *(lds_u4p)(uintptr_t)(ent[c] + (uint32_t)j * gstride);
        const u32x4 w = *(lds_u4p)(uintptr_t)(ent[c + 1] + (uint32_t)j * gstride);
        U = xor3(U, v.x, w.x); V = xor3(V, v.y, w.y); O1 = xor3(O1, v.z, w.z); O2 = xor3(O2, v.w, w.w);
        // at most six 16-byte reads in flight: more only costs VGPRs (24 per six) without hiding
        // any more LDS latency
        if (c % 6 == 4 && c + 2 < NR) __builtin_amdgcn_sched_barrier(0);
      }
      acc_graph4<FIXED>(S, gimg, recs + g * G4_WORDS, U, V, O1, O2, approx);
    }
    if (more) {
#pragma unroll
      for (int c = 0; c < NR; ++c) ent[c] += (uint32_t)delta;
      delta = -delta;
    }
    __syncthreads();
  }
  acc_finish4<FIXED>(S, lvl, approx, out_re, out_im);
}

template <int GT, int NCH>
__device__ __forceinline__ void sample4_block(const SampleArgs &A, int comp4_off, long long row, bool active, bool check_block);

// sample_program for one batch, LDS chunk-table formulation.  Requires every sampled component to
// have at most 64 parameters (2 words of x) - checked by the packer (p->v4).
template <int GT, int NCH>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(TSIMK_V4_WAVES, 8)))
k_sample4(Sample4Args A4) {
  const SampleArgs &A = A4.s;
  const int nthr = blockDim.x;
  const bool check_block = A4.has_check && (blockIdx.x == gridDim.x - 1);
  if (A4.feedback && blockIdx.x == 0 && threadIdx.x < 64 && A.row_lists > 1) {  // launch-plan feedback, as k_sample4h
    const int lane = threadIdx.x;
    uint32_t c = lane < A.row_lists ? A.row_count[32u * lane] : 0u, m = c;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      c += (uint32_t)__shfl_xor((int)c, o, 64);
      m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    }
    if (lane == 0) {
      A4.feedback[0] = c;
      A4.feedback[1] = m;
      A4.feedback[2] = (uint32_t)min(A.B, 0xFFFFFFFFll);
    }
  }
  long long row = (long long)blockIdx.x * nthr + threadIdx.x;
  bool active = row < A.B;
  if (A.row_index) {  // row lists (device-side post-selection / hard rows of a two-pass launch)
    const uint32_t nl = A.row_lists > 1 ? (uint32_t)A.row_lists : 1u;
    const uint32_t k = blockIdx.x % nl;
    const long long base = (long long)(blockIdx.x / nl) * nthr + (nl > 1 ? A.row_slot_begin : 0);
    const long long n = (long long)A.row_count[32u * k * (nl > 1 ? 1u : 0u)];
    if (!check_block && base >= n) return;  // block-uniform: no barrier skipped
    row = base + threadIdx.x;
    active = row < n;
    if (check_block) {  // lanes 0 and 1 both replay the check row (trial bit 1 / trial bit 0)
      active = (threadIdx.x < 2) && (A.check_row ? (*A.check_row != 0xFFFFFFFFu) : (n > 0));
      row = !active ? 0 : A.check_row ? (long long)*A.check_row : (long long)A.row_index[0];
    } else {
      row = active ? (long long)A.row_index[(size_t)k * (nl > 1 ? A.row_list_cap : 0) + row] : 0;
    }
  } else if (check_block) {  // replay in-batch shot 0 (row 0 of this launch) in lanes 0 and 1
    row = 0;
    active = (threadIdx.x < 2);
  }
  sample4_block<GT, NCH>(A, A4.comp4_off, row, active, check_block);
}

// the rows of one block (thread t: `row`, or idle) through every component: stage f, direct outputs, the levels, the stores
template <int GT, int NCH>
__device__ __forceinline__ void sample4_block(const SampleArgs &A, int comp4_off, long long row, bool active, bool check_block) {
  const int nthr = blockDim.x;
  // normalisation check (sampler.py:66-72): lane 1 of the check block evaluates every level with
  // trial bit 0 while lane 0 evaluates it with trial bit 1 - one pass gives both values
  const bool trial0 = check_block && threadIdx.x == 1;
  const unsigned long long shot = (unsigned long long)(A.shot_offset + row);
  cptr img = (cptr)(uintptr_t)A.img;

  const int WF32 = 2 * A.WF, WO32 = 2 * A.WO;
  uint32_t *lds_f = tsimk_lds + threadIdx.x;                // [WF32][nthr]
  uint32_t *lds_o = tsimk_lds + WF32 * nthr + threadIdx.x;  // [WO32][nthr]
  uint8_t *lds_tab = reinterpret_cast<uint8_t *>(tsimk_lds + (WF32 + WO32) * nthr);

  if (active) {
    stage_f_row(A.f + row * A.WF, A.WF, lds_f, nthr);
  } else {
    for (int w = 0; w < WF32; ++w) lds_f[w * nthr] = 0u;
  }
  for (int w = 0; w < WO32; ++w) lds_o[w * nthr] = 0u;

  // direct outputs (sampler.py:140-145)
  direct_outputs(A, img, lds_f, lds_o, nthr);

  for (int ci = 0; ci < A.n_comp; ++ci) {
    cptr comp = img + comp4_off + ci * C4_WORDS;
    const uint32_t n_out = comp[C_NOUT], F = comp[C_F];
    cptr fsel = img + comp[C_FSEL];
    cptr levels = img + comp[C4_LEVELS];
    cptr outpos = img + comp[C_OUTPOS];
    const uint32_t keybase = comp[C_KEYBASE];

    // x = (f_sel, outcome bits): two words - or three (NCH > 16, round 5): components of at most 64 selected f bits whose
    // F + n_out parameters pass 64 (class F60 of scripts/shape_map.py).  The f bits never leave words 0 and 1.
    // Four words (NCH = 32, end of round 5): up to 128 parameters, f_sel itself may pass 64 bits (class F70).
    constexpr int XW = NCH > 24 ? 4 : NCH > 16 ? 3 : 2;
    uint32_t x[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int w = 0; w < XW; ++w) {
      uint32_t v = 0;
      const int lo = w * 32;
      const int hi = min((int)F, lo + 32);
      for (int j = lo; j < hi; ++j) {
        const uint32_t src = fsel[j];
        v |= ((lds_f[(src >> 5) * nthr] >> (src & 31u)) & 1u) << (j - lo);
      }
      x[w] = v;
    }
    // bit `bitpos` of x := v (bitpos is wave-uniform: the word is picked by scalar branches, the array stays in registers)
    auto set_bit = [&](uint32_t bitpos, bool v) {
      const uint32_t bm = 1u << (bitpos & 31u), bw = bitpos >> 5;
      if (bw == 0u) x[0] = v ? (x[0] | bm) : (x[0] & ~bm);
      else if (bw == 1u || XW == 2) x[1] = v ? (x[1] | bm) : (x[1] & ~bm);
      else if (bw == 2u || XW == 3) x[2] = v ? (x[2] | bm) : (x[2] & ~bm);
      else x[3] = v ? (x[3] | bm) : (x[3] & ~bm);
    };

    // Sparse-f decision, block-uniform: every lane has at most 4 set f bits and the component has
    // at most 8 outputs (the packer then emitted the column tables, L4_STAB != 0).
    const unsigned long long xf = ((unsigned long long)x[1] << 32) | x[0];
    const bool sparse = (n_out <= 8u) && (levels[L4_STAB] != 0u) && (__syncthreads_and(__popcll(xf) <= 4) != 0);
    float prev = 0.0f, maxdev = 0.0f;
    // level 0 is the normalisation (sampler.py:54); level li > 0 evaluates output li-1 with trial
    // bit 1 (sampler.py:65) and - in the check block only - once more with trial bit 0 (sampler.py:66)
    for (uint32_t li = 0; li <= n_out; ++li) {
      cptr lvl = levels + li * L4_WORDS;
      const uint32_t bitpos = F + li - 1u;
      const bool lvl_fixed = (lvl[L4_FLAGS] & TSIMK_LFLAG_FIXED) != 0;
      if (li > 0) set_bit(bitpos, !trial0);
      float v1, v0 = 0.0f;
      {
        float re, im;
        if (sparse && lvl_fixed) {
          // sparse-f tables: <= 4 set f bits in every lane of the block -> 4 column reads + the two
          // 4-bit chunks of the output bits, instead of one read per 4-bit chunk of all of x
          const unsigned long long xx = ((unsigned long long)x[1] << 32) | x[0];
          const uint32_t mb = (uint32_t)(xx >> F);
          // the (at most four) set f bits -> column entries; F = the all-zero column.  Recomputed per
          // level (a dozen VALU ops) rather than kept live across the component.
          uint32_t col_off[4];
          {
            unsigned long long rem = F >= 64u ? xx : (xx & ((1ull << F) - 1ull));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const uint32_t pos = rem ? (uint32_t)__builtin_ctzll(rem) : F;
              rem &= rem - 1ull;
              col_off[k] = pos * 16u;
            }
          }
          const uint32_t e6[6] = {col_off[0], col_off[1], col_off[2], col_off[3],
                                  (F + 1u + (mb & 15u)) * 16u, (F + 17u + ((mb >> 4) & 15u)) * 16u};
          eval_level4<GT, 6, true, TSIMK_SPARSE_ENTRIES * 16>(A.img, img, lvl, e6, TSIMK_SPARSE_ENTRIES * (GT * 16), lvl[L4_STAB], lds_tab, re, im);
        } else {
          uint32_t en[NCH];
#pragma unroll
          for (int c = 0; c < NCH; ++c) {
            const uint32_t w = (c < 8) ? x[0] : (c < 16) ? x[1] : (c < 24) ? x[2] : x[3];
            en[c] = ((w >> (4 * (c & 7))) & 15u) * 16u + c * Tile4<GT>::kChunkBytes;
          }
          if (lvl_fixed) eval_level4<GT, NCH, true, 256>(A.img, img, lvl, en, NCH * Tile4<GT>::kChunkBytes, lvl[L4_TABLES], lds_tab, re, im);
          else eval_level4<GT, NCH, false, 256>(A.img, img, lvl, en, NCH * Tile4<GT>::kChunkBytes, lvl[L4_TABLES], lds_tab, re, im);
        }
        v1 = cabs32(re, im);
      }
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
      const uint32_t dst = outpos[i];
      lds_o[(dst >> 5) * nthr] |= (bit ? 1u : 0u) << (dst & 31u);
    }
    if (check_block && threadIdx.x == 0 && active && A.norm_dev) A.norm_dev[ci] = maxdev;
  }

  if (active && !check_block) {
    if (A.out) {  // nullptr: the caller wants the bit_packed rows only
      uint64_t *orow = A.out + row * A.WO;
      for (int w = 0; w < A.WO; ++w)
        orow[w] = (uint64_t)lds_o[(2 * w) * nthr] | ((uint64_t)lds_o[(2 * w + 1) * nthr] << 32);
    }
    store_compact_row(A, row, lds_o, nthr);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Hard-row lists that turn out LONG (the launch plan follows the counts of EARLIER launches: the first group after a jump of
// the noise level hands lists of 10^5 rows to kernels sized for 10): the latency kernels (k_sample_hw, k_sample4h_multi) take
// the first `slot_begin` slots of every list, worker blocks everything behind them - a block of rows per step at the per-shot
// kernel's rate, a fixed number of chip-resident blocks striding over (launch, list, chunk).  The workers of k_sample_hw
// are blocks appended to ITS grid (one more kernel per group on a first-pass lane cost C2 2-4 %, an empty one too); behind
// the one-batch-per-call k_sample4h they are a grid of their own, k_sample4_over (k_sample4h_multi carries its own).  Rows are recomputed whole
// (a row is a function of its f row, the key and the shot index), so it does not matter what the first pass left in them.
// When no list is longer than slot_begin - every launch but that one - a block reads the counts and exits.
// ---------------------------------------------------------------------------------------------------------------------
#ifndef TSIMK_H_MAX_CTX
#define TSIMK_H_MAX_CTX 8
#endif
struct Over4Multi {
  int n_ctx, comp4_off;
  uint32_t slot_begin;
  int masked;  // 1: list entries carry component masks in their top four bits (partial rows of the fused first pass)
  SampleArgs ctx[TSIMK_H_MAX_CTX];
};

// the overflow share of worker `worker` of `n_workers` blocks (the whole grid of k_sample4_over, or the blocks appended to
// a k_sample_hw grid): `ctx` are the launch's contexts as they lie in the kernel arguments
template <int GT, int NCH>
__device__ __forceinline__ void over4_rows(const SampleArgs *ctx, int n_ctx, int comp4_off, uint32_t slot_begin, bool masked,
                                           uint32_t worker, uint32_t n_workers) {
  const uint32_t nthr = blockDim.x;
  // every list's length at once (one load per thread, not a chain of n_ctx x lists dependent loads per block), and
  // out when none reaches slot_begin - all launches but the first after a jump of the noise level
  __shared__ uint32_t s_cnt[TSIMK_H_MAX_CTX * 64];
  {
    bool any = false;
    for (uint32_t i = threadIdx.x; i < (uint32_t)n_ctx * 64u; i += nthr) {
      const SampleArgs &A = ctx[i >> 6];
      const uint32_t k = i & 63u;
      const uint32_t n = (int)k < A.row_lists ? A.row_count[32u * k] : 0u;
      s_cnt[i] = n;
      any = any || n > slot_begin;
    }
    if (!__syncthreads_or(any ? 1 : 0)) return;
  }
  uint32_t pair = 0;
  for (int c = 0; c < n_ctx; ++c) {
    const SampleArgs &A = ctx[c];
    const uint32_t check_row = (A.no_check || !A.check_row) ? 0xFFFFFFFFu : *A.check_row;
    for (int k = 0; k < A.row_lists; ++k, ++pair) {
      const uint32_t n = s_cnt[64 * c + k];
      if (n <= slot_begin) continue;  // block-uniform
      const uint32_t chunks = (n - slot_begin + nthr - 1u) / nthr;
      for (uint32_t q = (worker + n_workers - pair % n_workers) % n_workers; q < chunks; q += n_workers) {
        const uint32_t slot = slot_begin + q * nthr + threadIdx.x;
        const bool active = slot < n;
        uint32_t entry = active ? A.row_index[(size_t)k * A.row_list_cap + slot] : 0u;
        if (masked) entry &= 0x0FFFFFFFu;
        // the row of the normalisation check (sampler.py:66-72) landed behind the latency kernel's share: replayed here
        const bool has_check = __syncthreads_or(active && entry == check_row) != 0;
        sample4_block<GT, NCH>(A, comp4_off, (long long)entry, active, false);
        __syncthreads();  // the staging columns are reused by the next chunk
        if (has_check) {
          sample4_block<GT, NCH>(A, comp4_off, (long long)check_row, threadIdx.x < 2u, true);
          __syncthreads();
        }
      }
    }
  }
}

template <int GT, int NCH>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(TSIMK_V4_WAVES, 8))) k_sample4_over(Over4Multi M) {
  over4_rows<GT, NCH>(M.ctx, M.n_ctx, M.comp4_off, M.slot_begin, M.masked != 0, blockIdx.x, gridDim.x);
}

}  // namespace tsimk
