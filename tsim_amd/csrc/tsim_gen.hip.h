// tsim_gen.hip.h - the fused first pass for ANY program whose components are narrow (k_sample_gen): f rows of up to 2048
// bits, up to 512 outputs, any number of components of up to TSIMK_LW_MAX_NOUT outputs each, up to TSIMK_GEN_KEYS compiled
// outputs per program.
//
// Why.  The reference takes every program shape through the same code (src/tsim/sampler.py:117-167: direct bits, then
// _sample_component per component, :28-81); it has no notion of a "narrow row".  The register first passes of this
// library (k_sample_lw_fast / _fastm / _multi) keep a shot's f row in two or four VGPRs and its outputs in two - a program
// with f index 128 or a 65th output fell off them onto the one-batch-per-launch kernels of round 1-2 (LDS staging per
// thread column, a gather program per component): profiles/r05/shape_map_before.txt, 0.4-0.7 of the nearest BASELINE
// configuration.  This kernel is the fused group (up to TSIMK_GEN_MAX_STEPS batches in one grid of chip-resident blocks,
// hard rows to the batch's lists - the protocol of tsim_lw_multi.hip.h, so the hard-row kernels behind it are the same)
// with the row in LDS instead of registers:
//   * a wave works on 64 consecutive rows; they arrive by LDS-DMA (global_load_lds: 64 consecutive dwords per instruction,
//     coalesced whatever the row width; the next chunk's copy is in flight while this one is worked on when two buffers fit);
//   * direct outputs f[idx] ^ flip (sampler.py:140-145) as rotate-and-mask runs sorted by destination word;
//   * per component only the f words that hold selected bits are touched (a record of (word, mask, selected bits below)
//     triples): the lane walks ITS set bits, each adds RANK[ordinal][position inside f_sel] to the colex rank of the
//     pattern (tsim_lw.hip.h) - weight <= table depth: the threshold tree of the pattern, n_out draws, three tree levels
//     per memory access (tsim_lw_pass.hip.h);
//   * everything wave-uniform lives in LDS, copied once per block: runs, component records, rank tables, subkeys.
// Same thresholds, same draws, same hard-row lists as the other first passes: bit-identical results
// (tests/test_gpu_shape_classes.py runs every class through this kernel, the one-batch path and the oracle).
#pragma once
#include "tsim_lw_fast.hip.h"

namespace tsimk {

#define TSIMK_GEN_MAX_STEPS 8
#define TSIMK_GEN_KEYS 40
#define TSIMK_GEN_MAX_RUNS 1024
#define TSIMK_GEN_MAX_COMP 16

// gen record in the program image (uint32 words, 64-byte aligned): header, then the static LDS block
enum {
  GR_NCOMP = 0,
  GR_WO32,       // 32-bit words per output row the record was built for
  GR_NRUNS,
  GR_LDS_WORDS,  // words of the static block (copied to LDS word 0 ..)
  GR_L_RUNS,     // LDS word offsets inside the static block: runs (2 words each: src_word | rot << 8, mask) ...
  GR_L_RUNB,     // ... WO32 + 1 run boundaries ...
  GR_L_FLIPS,    // ... WO32 constant-flip words ...
  GR_L_COMP,     // ... component records (GC_WORDS each)
  GR_WF32_MIN,   // the f row must have at least this many 32-bit words
  GR_WORDS = 16
};
// component record inside the static block
enum {
  GC_NOUT = 0, GC_F, GC_KEYBASE, GC_NWORDS,
  GC_L_WORDS,   // LDS word offset of NWORDS x (f word index, selection mask, selected bits in lower words)
  GC_L_RANK,    // LDS word offset of RANK[TSIMK_LW_MAX_WEIGHT][F]: C(position, ordinal + 1)
  GC_L_OUTPOS,  // LDS word offset of the n_out output columns
  GC_RSV,
  // filled by the kernel from the live LW record (the table depth changes while a handle lives, tsim_tables.hip)
  GC_WMAX = 8, GC_TAB_LO, GC_TAB_HI, GC_TAB_BYTES,
  GC_BASES = 16,  // 8 words
  GC_WORDS = 32
};

struct GenStep {
  const uint64_t *f;      // [B, WF] packed error-mechanism rows of this batch
  uint64_t *out;          // [B, WO] padded output rows, or nullptr
  uint8_t *out_compact;   // [B, out_rb] bit_packed rows (any alignment), or nullptr
  uint32_t *hard_index;   // this batch's hard-row lists
  uint32_t *ctl;          // its counters: ctl[32 k] = entries of list k, ctl[32 LISTS] = check row
  uint32_t *ctl_next;     // the counter set of the slot's NEXT launch: reset here
  uint32_t keys[2 * TSIMK_GEN_KEYS];  // per-output subkeys of this batch (sampler.py:74,147-148), host-computed
};

struct GenArgs {
  const uint32_t *img;
  const uint32_t *tab;      // integer thresholds (bernoulli_threshold), all components
  long long B;              // rows per batch (< 2^28)
  long long shot_offset;    // in-batch index of row 0, the same for every batch of the group
  int n_steps, chunks_per_step;
  int has_check, out_rb, WF32;
  int lw_off, gr_off;
  int list_cap, n_lists;    // hard-row lists: row block (1024 rows) rb -> list rb % n_lists
  int nbuf;                 // row buffers per wave: 2 = the next chunk travels while this one is worked on
  int l_wave, wave_bytes;   // LDS byte offset of wave 0's buffers, bytes per wave
  GenStep step[TSIMK_GEN_MAX_STEPS];
};

template <int WO32>
__global__ void __launch_bounds__(1024) k_sample_gen(GenArgs A) {
  typedef const __attribute__((address_space(4))) uint8_t *cbytes;
  typedef const __attribute__((address_space(4))) GenStep *cstep;
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) void *lds_ptr_t;
  typedef const __attribute__((address_space(1))) void *glb_ptr_t;
  const int nthr = blockDim.x;
  const uint32_t lane = threadIdx.x & 63u, wpb = (uint32_t)nthr >> 6;
  const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  cptr img = (cptr)(uintptr_t)A.img;
  cptr gr = img + A.gr_off;
  const uint32_t n_comp = gr[GR_NCOMP], n_runs = gr[GR_NRUNS], lds_words = gr[GR_LDS_WORDS];
  const uint32_t l_comp = gr[GR_L_COMP];
  const uint32_t WF32 = (uint32_t)A.WF32;
  uint32_t *L = tsimk_lds;
  cstep steps = (cstep)((cbytes)__builtin_amdgcn_kernarg_segment_ptr() + __builtin_offsetof(GenArgs, step));
  uint32_t *l_keys = L + ((lds_words + 3u) & ~3u);  // per step: 2 * TSIMK_GEN_KEYS subkey words

  // ---- once per block: the static block, the live table records, the subkeys
  {
    const uint32_t *g = A.img + A.gr_off + GR_WORDS;
    for (uint32_t i = threadIdx.x; i < lds_words; i += nthr) L[i] = g[i];
    for (uint32_t i = threadIdx.x; i < 2u * TSIMK_GEN_KEYS * (uint32_t)A.n_steps; i += nthr)
      l_keys[i] = steps[i / (2u * TSIMK_GEN_KEYS)].keys[i % (2u * TSIMK_GEN_KEYS)];
    __syncthreads();
    if (threadIdx.x < n_comp) {
      const uint32_t *rec = A.img + A.lw_off + threadIdx.x * LW_WORDS;
      uint32_t *c = L + l_comp + threadIdx.x * GC_WORDS;
      const uint64_t base = (uint64_t)(uintptr_t)A.tab + (uint64_t)rec[LW_TAB] * 4ull;
      const uint64_t bytes = ((uint64_t)rec[LW_NPAT] << c[GC_NOUT]) * 4ull;
      c[GC_WMAX] = rec[LW_WMAX];
      c[GC_TAB_LO] = (uint32_t)base;
      c[GC_TAB_HI] = (uint32_t)(base >> 32);
      c[GC_TAB_BYTES] = bytes > 0xFFFFFFFCull ? 0xFFFFFFFCu : (uint32_t)bytes;
      for (int k = 0; k < 8; ++k) c[GC_BASES + k] = rec[LW_BASES_INLINE + k];
    }
    __syncthreads();
  }
  const uint32_t *l_runs = L + gr[GR_L_RUNS], *l_runb = L + gr[GR_L_RUNB], *l_flips = L + gr[GR_L_FLIPS];

  uint8_t *w8 = reinterpret_cast<uint8_t *>(L) + A.l_wave + wv * (uint32_t)A.wave_bytes;
  uint32_t *w_buf[2] = {reinterpret_cast<uint32_t *>(w8), reinterpret_cast<uint32_t *>(w8) + (A.nbuf == 2 ? 64u * WF32 : 0u)};

  const uint32_t so_lo = (uint32_t)A.shot_offset, so_hi = (uint32_t)((unsigned long long)A.shot_offset >> 32);
  const uint32_t Bu = (uint32_t)A.B;
  const uint32_t cps = (uint32_t)A.chunks_per_step;
  const uint32_t total = cps * (uint32_t)A.n_steps;
  (void)n_runs;

  // the f rows of chunk (st, ch) -> dst (LDS-DMA: 64 consecutive dwords per instruction, rows as they lie in HBM)
  auto stage_chunk = [&](uint32_t st, uint32_t ch, uint32_t *dst) {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(steps[st].f) + (size_t)ch * 64u * WF32;
    const uint32_t valid = (Bu - ch * 64u < 64u ? Bu - ch * 64u : 64u) * WF32;  // dwords of this chunk inside the batch
    if (valid == 64u * WF32) {
      for (uint32_t w = 0; w < WF32; ++w) __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + w * 64u + lane), (lds_ptr_t)(dst + w * 64u), 4, 0, 0);
    } else {
      for (uint32_t w = 0; w < WF32; ++w) {
        const uint32_t j = w * 64u + lane;
        if (j < valid) __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + j), (lds_ptr_t)(dst + w * 64u), 4, 0, 0);
      }
    }
  };

  const uint32_t tw = gridDim.x * wpb;
  uint32_t c = blockIdx.x * wpb + wv;
  if (c >= total) return;
  uint32_t st = c / cps, ch = c - st * cps;  // once; afterwards by increments
  const uint32_t tw_st = tw / cps, tw_ch = tw - tw_st * cps;
  uint32_t cur = 0;
  stage_chunk(st, ch, w_buf[0]);
  for (;;) {
    cstep S = steps + st;
    const uint32_t row = ch * 64u + lane;
    const bool active = row < Bu;
    uint32_t st_n = st + tw_st, ch_n = ch + tw_ch;
    if (ch_n >= cps) { ch_n -= cps; ++st_n; }
    const bool more = c + tw < total;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this chunk's rows are in LDS (and every earlier store of this wave has left)
    if (more && A.nbuf == 2) stage_chunk(st_n, ch_n, w_buf[cur ^ 1u]);
    const uint32_t *frow = w_buf[cur] + lane * WF32;
    if (ch == 0u) {  // the wave that owns a batch's first rows resets the slot's other counter set
      S->ctl_next[32u * lane] = 0u;
      if (lane == 0u) S->ctl_next[32u * TSIMK_LW_LISTS] = 0xFFFFFFFFu;  // "no check row"
    }
    // ---- K14: direct outputs f[idx] ^ flip (sampler.py:140-145): rotate-and-mask runs per destination word
    uint32_t o[WO32];
#pragma unroll
    for (int d = 0; d < WO32; ++d) {
      uint32_t acc = l_flips[d];
      const uint32_t r1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)l_runb[d + 1]);
      for (uint32_t r = (uint32_t)__builtin_amdgcn_readfirstlane((int)l_runb[d]); r < r1; ++r) {
        const u32x2 run = *reinterpret_cast<const u32x2 *>(&l_runs[2u * r]);
        const uint32_t ctl = (uint32_t)__builtin_amdgcn_readfirstlane((int)run.x);
        const uint32_t fw = frow[ctl & 255u];
        acc ^= __builtin_amdgcn_alignbit(fw, fw, ctl >> 8) & run.y;
      }
      o[d] = acc;
    }
    bool hard = false;
    if (A.has_check && ch == 0u && lane == 0u) {  // the normalisation-check row (sampler.py:66-72): always hard
      hard = true;
      S->ctl[32 * TSIMK_LW_LISTS] = row;
    }
    const uint32_t slo = so_lo + row;  // (the launcher keeps shot_offset + B below the next multiple of 2^32)
    const uint32_t *keys = l_keys + st * (2u * TSIMK_GEN_KEYS);
    // ---- the components, in processing order (sampler.py:147-148)
    for (uint32_t ci = 0; ci < n_comp; ++ci) {
      const uint32_t *cr = L + l_comp + ci * GC_WORDS;
      const u32x4 c0 = *reinterpret_cast<const u32x4 *>(cr);
      const u32x4 c1 = *reinterpret_cast<const u32x4 *>(cr + 4);
      const u32x4 c2 = *reinterpret_cast<const u32x4 *>(cr + 8);
      const uint32_t n_out = (uint32_t)__builtin_amdgcn_readfirstlane((int)c0.x), F = (uint32_t)__builtin_amdgcn_readfirstlane((int)c0.y);
      const uint32_t keybase = (uint32_t)__builtin_amdgcn_readfirstlane((int)c0.z), nwords = (uint32_t)__builtin_amdgcn_readfirstlane((int)c0.w);
      const uint32_t *cw = L + (uint32_t)__builtin_amdgcn_readfirstlane((int)c1.x);
      const uint32_t *rank = L + (uint32_t)__builtin_amdgcn_readfirstlane((int)c1.y);
      const uint32_t *outpos = L + (uint32_t)__builtin_amdgcn_readfirstlane((int)c1.z);
      const uint32_t wmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)c2.x);
      const uint64_t tbase = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)c2.z) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)c2.y);
      const uint32_t tbytes = (uint32_t)__builtin_amdgcn_readfirstlane((int)c2.w);
      // the descriptor ends with the component's table: a lane whose pattern index means nothing reads zeros, never beyond
      const __amdgpu_buffer_rsrc_t r_tab = __builtin_amdgcn_make_buffer_rsrc((void *)(uintptr_t)tbase, 0, tbytes, 0x00020000);
      // colex rank = sum over the set selected bits, in ascending order, of C(position inside f_sel, ordinal + 1)
      uint32_t ord = 0u, pat = 0u;
      for (uint32_t wi = 0; wi < nwords; ++wi) {
        const uint32_t widx = (uint32_t)__builtin_amdgcn_readfirstlane((int)cw[3u * wi]);
        const uint32_t sw = (uint32_t)__builtin_amdgcn_readfirstlane((int)cw[3u * wi + 1u]);
        const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)cw[3u * wi + 2u]);
        uint32_t m = active ? (frow[widx] & sw) : 0u;
        while (m) {
          const uint32_t p = (uint32_t)__builtin_ctz(m);
          const uint32_t pos = base + (uint32_t)__builtin_popcount(sw & ((1u << p) - 1u));
          const uint32_t oc = ord < (uint32_t)(TSIMK_LW_MAX_WEIGHT - 1) ? ord : (uint32_t)(TSIMK_LW_MAX_WEIGHT - 1);
          pat += rank[oc * F + pos];
          ++ord;
          m &= m - 1u;
        }
      }
      if (ord > wmax) hard = true;
      pat += cr[GC_BASES + (ord < 7u ? ord : 7u)];
      pat = hard ? 0u : pat;
      const uint32_t thr = pat << (n_out + 2u);  // byte offset of the pattern's threshold tree inside the component's table
      auto draw = [&](uint32_t i) -> uint32_t {
        const uint32_t k0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)keys[2u * (keybase + i)]);
        const uint32_t k1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)keys[2u * (keybase + i) + 1u]);
        return threefry_bits32_lo(k0, k1, k0 + so_hi, slo) >> 9;
      };
      auto emit = [&](uint32_t i, bool bit) {
        const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)outpos[i]);
        const uint32_t v = (bit ? 1u : 0u) << (dst & 31u);
#pragma unroll
        for (int d = 0; d < WO32; ++d)
          if ((dst >> 5) == (uint32_t)d) o[d] |= v;
      };
      // the threshold walk, three tree levels per memory access (tsim_lw_pass.hip.h: lw_walk_impl)
      uint32_t node = 1u, i = 0u;
      for (; i + 3u <= n_out; i += 3u) {
        const uint32_t t0 = __builtin_amdgcn_raw_buffer_load_b32(r_tab, thr + 4u * node, 0, 0);
        const u32x2 t1 = __builtin_amdgcn_raw_buffer_load_b64(r_tab, thr + 8u * node, 0, 0);
        const u32x4 t2 = __builtin_amdgcn_raw_buffer_load_b128(r_tab, thr + 16u * node, 0, 0);
        const uint32_t d0 = draw(i), d1 = draw(i + 1u), d2 = draw(i + 2u);
        const bool b0 = d0 < t0;
        const bool b1 = d1 < (b0 ? t1.y : t1.x);
        const uint32_t lo = b1 ? t2.y : t2.x, hi = b1 ? t2.w : t2.z;
        const bool b2 = d2 < (b0 ? hi : lo);
        node = 8u * node + (b0 ? 4u : 0u) + (b1 ? 2u : 0u) + (b2 ? 1u : 0u);
        emit(i, b0);
        emit(i + 1u, b1);
        emit(i + 2u, b2);
      }
      if (n_out - i == 2u) {
        const uint32_t t0 = __builtin_amdgcn_raw_buffer_load_b32(r_tab, thr + 4u * node, 0, 0);
        const u32x2 t1 = __builtin_amdgcn_raw_buffer_load_b64(r_tab, thr + 8u * node, 0, 0);
        const uint32_t d0 = draw(i), d1 = draw(i + 1u);
        const bool b0 = d0 < t0;
        const bool b1 = d1 < (b0 ? t1.y : t1.x);
        emit(i, b0);
        emit(i + 1u, b1);
      } else if (n_out - i == 1u) {
        const uint32_t t0 = __builtin_amdgcn_raw_buffer_load_b32(r_tab, thr + 4u * node, 0, 0);
        emit(i, draw(i) < t0);
      }
    }
    hard = hard && active;
    // ---- one buffer only: the next chunk's rows start to travel now (every read of this chunk has returned)
    if (more && A.nbuf != 2) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      stage_chunk(st_n, ch_n, w_buf[0]);
    }
    // ---- store the row (tabulated rows only: a hard row is written whole by the hard-row kernel)
    if (active && !hard) {
      uint64_t *out = S->out;
      uint8_t *oc = S->out_compact;
      if (out) {
        uint32_t *dst = reinterpret_cast<uint32_t *>(out) + (size_t)row * WO32;
        if constexpr (WO32 == 4) *reinterpret_cast<u32x4 *>(dst) = u32x4{o[0], o[1], o[2], o[3]};
        else {
#pragma unroll
          for (int d = 0; d < WO32; d += 2) *reinterpret_cast<u32x2 *>(dst + d) = u32x2{o[d], o[d + 1]};
        }
      }
      if (oc) {
        // out_rb bytes at row * out_rb (any alignment: the device runs in unaligned-access mode): dwords, then 2, then 1
        const __amdgpu_buffer_rsrc_t r_c = __builtin_amdgcn_make_buffer_rsrc((void *)oc, 0, 0xFFFFFFFF, 0x00020000);
        const uint32_t rb8 = (uint32_t)A.out_rb, nd = rb8 >> 2, rem = rb8 & 3u;
        const uint32_t off = row * rb8;
        uint32_t tw_ = o[0];
#pragma unroll
        for (int d = 0; d < WO32; ++d) {
          if ((uint32_t)d < nd) __builtin_amdgcn_raw_buffer_store_b32(o[d], r_c, off + 4u * (uint32_t)d, 0, 0);
          tw_ = (nd == (uint32_t)d) ? o[d] : tw_;
        }
        const uint32_t at = off + 4u * nd;
        if (rem >= 2u) __builtin_amdgcn_raw_buffer_store_b16((uint16_t)tw_, r_c, at, 0, 0);
        if (rem & 1u) __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(tw_ >> (rem == 3u ? 16 : 0)), r_c, at + (rem == 3u ? 2u : 0u), 0, 0);
      }
    }
    // ---- wave-aggregated append of the hard rows to this batch's lists
    const unsigned long long hm = __builtin_amdgcn_ballot_w64(hard);
    if (hm != 0ull) {
      const int leader = __builtin_ctzll(hm);
      uint32_t basei = 0;
      const uint32_t k = (ch >> 4) % (uint32_t)A.n_lists;  // this row block's sub-list (row blocks of 1024 rows)
      uint32_t *ctl = S->ctl;
      if ((int)lane == leader) basei = atomicAdd(&ctl[32u * k], (uint32_t)__popcll(hm));
      basei = (uint32_t)__shfl((int)basei, leader, 64);
      if (hard) S->hard_index[(size_t)k * A.list_cap + basei + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull))] = row;
    }
    if (!more) break;
    c += tw;
    st = st_n;
    ch = ch_n;
    if (A.nbuf == 2) cur ^= 1u;
  }
}

}  // namespace tsimk
