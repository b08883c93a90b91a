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
//   * everything wave-uniform comes by scalar loads from the program image / the kernel arguments (runs four at a time,
//     component and word records, subkeys); LDS holds the rows, the rank tables and the pattern bases.
// Same thresholds, same draws, same hard-row lists as the other first passes: bit-identical results
// (tests/test_gpu_shape_classes.py runs every class through this kernel, the one-batch path and the oracle).
#pragma once
#include "tsim_lw_fast.hip.h"
#include <type_traits>

namespace tsimk {

#ifndef TSIMK_GEN_MAX_STEPS
#define TSIMK_GEN_MAX_STEPS 8  // (also in tsim_sample_internal.hip.h: the steps driver sizes its groups by it)
#endif
#ifndef TSIMK_GEN_KEYS
#define TSIMK_GEN_KEYS 320     // subkey records of ONE launch: batches x compiled outputs (a group of 8 batches: 40 outputs; 64 outputs: 5 batches; 320: one)
#endif
#define TSIMK_GEN_MAX_RUNS 1024
#define TSIMK_GEN_MAX_COMP 32

// gen record in the program image (uint32 words, 64-byte aligned): header, then its tables (image offsets)
enum {
  GR_NCOMP = 0,
  GR_WO32,       // 32-bit words per output row the record was built for
  GR_LDS_WORDS,  // words of the LDS block (the components' rank tables, copied to LDS word 0 ..; behind them 8 pattern bases per component)
  GR_LDS_SRC,    // image offset of that block
  GR_DST,        // image offset of WO32 x (first run group, groups, flip word, 0)
  GR_GROUPS,     // image offset of the run groups: 8 words = 2 x (f word, rotate right by, mask, 0), padded with mask 0
  GR_COMP,       // image offset of the component records (GC_WORDS each)
  GR_WF32_MIN,   // the f row must have at least this many 32-bit words
  GR_WORDS = 16
};
// component record
enum {
  GC_NOUT = 0, GC_F, GC_KEYBASE, GC_NWORDS,
  GC_WORDREC,   // image offset of NWORDS x (f word index, selection mask, selected bits in lower words, 0)
  GC_L_RANK,    // LDS word offset of RANK[TSIMK_LW_MAX_WEIGHT][F]: C(position, ordinal + 1)
  GC_OUTPOS,    // image offset of the n_out output columns
  GC_OUTBASE,   // the output columns are GC_OUTBASE, + 1, + 2, ... (the usual case: a component's outputs are consecutive columns); 0xFFFFFFFF: see GC_OUTPOS
  GC_WORDS = 8
};

struct GenStep {
  const uint64_t *f;      // [B, WF] packed error-mechanism rows of this batch
  uint64_t *out;          // [B, WO] padded output rows, or nullptr
  uint8_t *out_compact;   // [B, out_rb] bit_packed rows (any alignment), or nullptr
  uint32_t *hard_index;   // this batch's hard-row lists
  uint32_t *ctl;          // its counters: ctl[32 k] = entries of list k, ctl[32 LISTS] = check row
  uint32_t *ctl_next;     // the counter set of the slot's NEXT launch: reset here
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
  int total_keys;           // compiled outputs of the program: batch st's subkeys (sampler.py:74,147-148, host-computed) are keys[2 st total_keys ..]
  uint32_t keys[2 * TSIMK_GEN_KEYS];
};
static_assert(sizeof(GenArgs) <= 4096, "kernel arguments");

#ifndef TSIMK_GEN_UNIFORM_WALK
#define TSIMK_GEN_UNIFORM_WALK 1
#endif
#ifndef TSIMK_GEN_SGPRS
#define TSIMK_GEN_SGPRS 96   // 8 waves per SIMD: two blocks of 16 waves per CU
#endif

template <int WO32>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(TSIMK_GEN_SGPRS))) k_sample_gen(GenArgs A) {
  typedef const __attribute__((address_space(4))) uint8_t *cbytes;
  typedef const __attribute__((address_space(4))) GenStep *cstep;
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
  typedef const __attribute__((address_space(4))) u32x4 *cptr4;
  typedef const __attribute__((address_space(4))) u32x8 *cptr8;
  typedef __attribute__((address_space(3))) void *lds_ptr_t;
  typedef const __attribute__((address_space(1))) void *glb_ptr_t;
  const int nthr = blockDim.x;
  const uint32_t lane = threadIdx.x & 63u, wpb = (uint32_t)nthr >> 6;
  const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  cptr img = (cptr)(uintptr_t)A.img;
  cptr gr = img + A.gr_off;
  const uint32_t n_comp = gr[GR_NCOMP], lds_words = gr[GR_LDS_WORDS];
  cptr g_dst = img + gr[GR_DST], g_groups = img + gr[GR_GROUPS], g_comp = img + gr[GR_COMP];
  const uint32_t WF32 = (uint32_t)A.WF32;
  uint32_t *L = tsimk_lds;
  uint32_t *l_bases = L + lds_words;  // [n_comp][8]: index of the first pattern of weight w (live: the table depth changes, tsim_tables.hip)
  cstep steps = (cstep)((cbytes)__builtin_amdgcn_kernarg_segment_ptr() + __builtin_offsetof(GenArgs, step));

  // ---- once per block: the rank tables and the pattern bases into LDS
  {
    const uint32_t *g = A.img + gr[GR_LDS_SRC];
    for (uint32_t i = threadIdx.x; i < lds_words; i += nthr) L[i] = g[i];
    if (threadIdx.x < 8u * n_comp) l_bases[threadIdx.x] = A.img[A.lw_off + (threadIdx.x >> 3) * LW_WORDS + LW_BASES_INLINE + (threadIdx.x & 7u)];
    __syncthreads();
  }

  uint32_t *w0 = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(L) + A.l_wave + wv * (uint32_t)A.wave_bytes);
  const uint32_t buf_words = A.nbuf == 2 ? 64u * WF32 : 0u;

  const uint32_t so_lo = (uint32_t)A.shot_offset, so_hi = (uint32_t)((unsigned long long)A.shot_offset >> 32);
  const uint32_t Bu = (uint32_t)A.B;
  const uint32_t cps = (uint32_t)A.chunks_per_step;
  const uint32_t total = cps * (uint32_t)A.n_steps;

  // the f rows of chunk (st, ch) -> dst (LDS-DMA: 64 consecutive dwords per instruction, rows as they lie in HBM)
  auto stage_chunk = [&](uint32_t st, uint32_t ch, uint32_t *dst) {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(steps[st].f) + (size_t)ch * 64u * WF32;
    const uint32_t valid = (Bu - ch * 64u < 64u ? Bu - ch * 64u : 64u) * WF32;  // dwords of this chunk inside the batch
    if (valid == 64u * WF32) {
      // 16 bytes per lane and instruction (1 KB of the chunk each), the last 512 bytes of an odd number of 64-bit words as dwords
      const uint32_t n16 = WF32 >> 2;
      for (uint32_t j = 0; j < n16; ++j)
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + j * 256u + 4u * lane), (lds_ptr_t)(dst + j * 256u), 16, 0, 0);
      for (uint32_t w = 4u * n16; w < WF32; ++w) __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + w * 64u + lane), (lds_ptr_t)(dst + w * 64u), 4, 0, 0);
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
  uint32_t cur = 0;  // 0 / buf_words: the buffer this chunk's rows are in
  stage_chunk(st, ch, w0);
  for (;;) {
    cstep S = steps + st;
    const uint32_t row = ch * 64u + lane;
    const bool active = row < Bu;
    uint32_t st_n = st + tw_st, ch_n = ch + tw_ch;
    if (ch_n >= cps) { ch_n -= cps; ++st_n; }
    const bool more = c + tw < total;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // this chunk's rows are in LDS; every read of the other buffer has returned
    if (more && A.nbuf == 2) stage_chunk(st_n, ch_n, w0 + (cur ^ buf_words));
    const uint32_t *frow = w0 + cur + lane * WF32;
    if (ch == 0u) {  // the wave that owns a batch's first rows resets the slot's other counter set
      S->ctl_next[32u * lane] = 0u;
      if (lane == 0u) S->ctl_next[32u * TSIMK_LW_LISTS] = 0xFFFFFFFFu;  // "no check row"
    }
    // ---- K14: direct outputs f[idx] ^ flip (sampler.py:140-145): rotate-and-mask runs per destination word, two per group
    uint32_t o[WO32];
    auto direct_word = [&](auto dc) {
      constexpr int d = decltype(dc)::value;
      if constexpr (d < WO32) {
        const u32x4 dh = *(cptr4)(g_dst + 4 * d);  // first group, groups, flips
        uint32_t acc = dh.z;
        for (uint32_t g = 0; g < dh.y; ++g) {
          const u32x8 grp = *(cptr8)(g_groups + 8u * (dh.x + g));  // two runs: (f word, rotation, mask, 0) each - nothing to decode
          const uint32_t fw0 = frow[grp[0]], fw1 = frow[grp[4]];
          acc ^= __builtin_amdgcn_alignbit(fw0, fw0, grp[1]) & grp[2];
          acc ^= __builtin_amdgcn_alignbit(fw1, fw1, grp[5]) & grp[6];
        }
        o[d] = acc;
      }
    };
    direct_word(std::integral_constant<int, 0>{});  direct_word(std::integral_constant<int, 1>{});
    direct_word(std::integral_constant<int, 2>{});  direct_word(std::integral_constant<int, 3>{});
    direct_word(std::integral_constant<int, 4>{});  direct_word(std::integral_constant<int, 5>{});
    direct_word(std::integral_constant<int, 6>{});  direct_word(std::integral_constant<int, 7>{});
    direct_word(std::integral_constant<int, 8>{});  direct_word(std::integral_constant<int, 9>{});
    direct_word(std::integral_constant<int, 10>{}); direct_word(std::integral_constant<int, 11>{});
    direct_word(std::integral_constant<int, 12>{}); direct_word(std::integral_constant<int, 13>{});
    direct_word(std::integral_constant<int, 14>{}); direct_word(std::integral_constant<int, 15>{});
    bool hard = false;
    if (A.has_check && ch == 0u && lane == 0u) {  // the normalisation-check row (sampler.py:66-72): always hard
      hard = true;
      S->ctl[32 * TSIMK_LW_LISTS] = row;
    }
    const uint32_t slo = so_lo + row;  // (the launcher keeps shot_offset + B below the next multiple of 2^32)
    cptr kp = (cptr)((cbytes)__builtin_amdgcn_kernarg_segment_ptr() + __builtin_offsetof(GenArgs, keys)) + 2u * st * (uint32_t)A.total_keys;
    // ---- the components, in processing order (sampler.py:147-148); everything wave-uniform by scalar loads from the image
    for (uint32_t ci = 0; ci < n_comp; ++ci) {
      const u32x8 cr = *(cptr8)(g_comp + ci * GC_WORDS);
      cptr lw = img + A.lw_off + ci * LW_WORDS;
      const uint32_t n_out = cr[GC_NOUT], F = cr[GC_F], keybase = cr[GC_KEYBASE], nwords = cr[GC_NWORDS];
      cptr cw = img + cr[GC_WORDREC];
      cptr outpos = img + cr[GC_OUTPOS];
      const uint32_t *rank = L + cr[GC_L_RANK];
      const uint32_t wmax = lw[LW_WMAX];
      const uint64_t tbase = (uint64_t)(uintptr_t)A.tab + (uint64_t)lw[LW_TAB] * 4ull;
      const bool trie = lw[LW_FMT] != 0u;  // the component's tables are a chunked prefix tree (tsim_trie.hip.h)
      const uint64_t tb64 = trie ? (uint64_t)lw[LW_CHUNKS] * 32ull : ((uint64_t)lw[LW_NPAT] << n_out) * 4ull;
      const uint32_t tbytes = tb64 > 0xFFFFFFFCull ? 0xFFFFFFFCu : (uint32_t)tb64;
      // the descriptor ends with the component's table: a lane whose pattern index means nothing reads zeros, never beyond
      const __amdgpu_buffer_rsrc_t r_tab = __builtin_amdgcn_make_buffer_rsrc((void *)(uintptr_t)tbase, 0, tbytes, 0x00020000);
      // colex rank = sum over the set selected bits, in ascending order, of C(position inside f_sel, ordinal + 1)
      uint32_t ord = 0u, pat = 0u;
      u32x4 wr = nwords ? *(cptr4)cw : u32x4{0u, 0u, 0u, 0u};  // the next word's record travels while this word's bits are walked
      for (uint32_t wi = 0; wi < nwords; ++wi) {
        const uint32_t widx = wr.x, sw = wr.y, base = wr.z;
        uint32_t m = active ? (frow[widx] & sw) : 0u;
        if (wi + 1u < nwords) wr = *(cptr4)(cw + 4u * (wi + 1u));
#if TSIMK_GEN_UNIFORM_WALK
        // a UNIFORM loop (trip count = the heaviest lane of the wave in this word, no exec masking): lanes without a bit left
        // run along and add nothing
        while (__builtin_amdgcn_ballot_w64(m != 0u) != 0ull) {
          const bool on = m != 0u;
          uint32_t p;
          asm("v_ffbl_b32 %0, %1" : "=v"(p) : "v"(m));  // (0xFFFFFFFF for m == 0: the shift below takes its low five bits)
          const uint32_t pos = base + (uint32_t)__builtin_popcount(sw & ((1u << (p & 31u)) - 1u));
          const uint32_t oc = ord < (uint32_t)(TSIMK_LW_MAX_WEIGHT - 1) ? ord : (uint32_t)(TSIMK_LW_MAX_WEIGHT - 1);
          const uint32_t rv = rank[on ? oc * F + pos : 0u];
          pat += on ? rv : 0u;
          ord += on ? 1u : 0u;
          m &= m - 1u;
        }
#else
        while (m) {
          const uint32_t p = (uint32_t)__builtin_ctz(m);
          const uint32_t pos = base + (uint32_t)__builtin_popcount(sw & ((1u << p) - 1u));
          const uint32_t oc = ord < (uint32_t)(TSIMK_LW_MAX_WEIGHT - 1) ? ord : (uint32_t)(TSIMK_LW_MAX_WEIGHT - 1);
          pat += rank[oc * F + pos];
          ++ord;
          m &= m - 1u;
        }
#endif
      }
      if (ord > wmax) hard = true;
      pat += l_bases[8u * ci + (ord < 7u ? ord : 7u)];
      if (pat >= lw[LW_NPAT_OK]) hard = true;  // (a prefix-tree build that the budget ended early, tsim_tables.hip)
      pat = hard ? 0u : pat;
      const uint32_t thr = pat << (n_out + 2u);  // byte offset of the pattern's threshold tree inside the component's table
      auto draw = [&](uint32_t i) -> uint32_t {
        const uint32_t k0 = kp[2u * (keybase + i)], k1 = kp[2u * (keybase + i) + 1u];
        return threefry_bits32_lo(k0, k1, k0 + so_hi, slo) >> 9;
      };
      auto emit = [&](uint32_t i, bool bit) {
        const uint32_t dst = outpos[i];
        const uint32_t v = (bit ? 1u : 0u) << (dst & 31u);
        // (selects, not conditional stores: a chain of `if (word == d) o[d] |= v` is folded into ONE dynamically indexed access
        // and the row's words move to scratch memory)
#pragma unroll
        for (int d = 0; d < WO32; ++d) o[d] |= ((dst >> 5) == (uint32_t)d) ? v : 0u;
      };
      if (trie) {
        // One 32-byte chunk per three outputs: thresholds of its seven nodes, the child-present mask, the first child.  A node
        // whose threshold is 0 or 2^23 takes no draw (u < T is decided), and when that holds for the whole wave the Threefry
        // block is skipped - detectors fixed by f and the outcomes before them cost a compare.
        uint32_t chunk = pat;
        // the sampled bits in output order (n_out <= 64); placed once behind the walk - with consecutive output columns two shifts
        // instead of a column load and WO32 selects per output (24-40 outputs: a tenth of the pass)
        uint32_t acc_lo = 0u, acc_hi = 0u;
        auto keep = [&](uint32_t i, bool bit) {
          const uint32_t v = (bit ? 1u : 0u) << (i & 31u);
          if (i < 32u) acc_lo |= v;
          else acc_hi |= v;
        };
        auto bit_of = [&](uint32_t i, uint32_t T) -> bool {
          const bool need = (T - 1u) < ((1u << 23) - 1u) && active && !hard;  // 0 < T < 2^23
          uint32_t dr = 0u;  // (T = 2^23: 0 < T; T = 0: never)
          if (__builtin_amdgcn_ballot_w64(need) != 0ull) dr = draw(i);
          return dr < T;
        };
        // (the root chunk holds the n_out mod 3 odd outputs, every other chunk three: tsim_trie.hip.h)
        uint32_t rem = n_out % 3u ? n_out % 3u : 3u;
        for (uint32_t i = 0u; i < n_out; i += rem, rem = 3u) {
          const u32x4 ca = __builtin_amdgcn_raw_buffer_load_b128(r_tab, chunk * 32u, 0, 0);
          const u32x4 cb = __builtin_amdgcn_raw_buffer_load_b128(r_tab, chunk * 32u + 16u, 0, 0);
          const uint32_t cmask = ca.y >> 24;
          const bool b0 = bit_of(i, ca.y & 0xFFFFFFu);
          keep(i, b0);
          bool b1 = false, b2 = false;
          if (rem > 1u) {
            b1 = bit_of(i + 1u, b0 ? ca.w : ca.z);
            keep(i + 1u, b1);
          }
          if (rem > 2u) {
            const uint32_t lo = b1 ? cb.y : cb.x, hi = b1 ? cb.w : cb.z;
            b2 = bit_of(i + 2u, b0 ? hi : lo);
            keep(i + 2u, b2);
          }
          if (i + rem < n_out) {
            const uint32_t leaf = rem == 3u ? (b0 ? 4u : 0u) + (b1 ? 2u : 0u) + (b2 ? 1u : 0u) : (rem == 2u ? (b0 ? 2u : 0u) + (b1 ? 1u : 0u) : (b0 ? 1u : 0u));
            if (!((cmask >> leaf) & 1u)) hard = true;  // the tree ends here: the table's budget was spent (tsim_trie.hip.h)
            chunk = ca.x + (uint32_t)__builtin_popcount(cmask & ((1u << leaf) - 1u));
          }
        }
        {
          const uint32_t ob = cr[GC_OUTBASE];
          if (ob != 0xFFFFFFFFu) {  // wave-uniform
            const uint32_t w0 = ob >> 5, sh = ob & 31u;
            const uint32_t v0 = acc_lo << sh;
            const uint32_t v1 = sh ? ((acc_lo >> (32u - sh)) | (acc_hi << sh)) : acc_hi;
            const uint32_t v2 = sh ? (acc_hi >> (32u - sh)) : 0u;
#pragma unroll
            for (int d = 0; d < WO32; ++d) o[d] |= ((uint32_t)d == w0) ? v0 : ((uint32_t)d == w0 + 1u) ? v1 : ((uint32_t)d == w0 + 2u) ? v2 : 0u;
          } else {
            for (uint32_t i = 0u; i < n_out; ++i) emit(i, (((i < 32u ? acc_lo : acc_hi) >> (i & 31u)) & 1u) != 0u);
          }
        }
        continue;
      }
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
      stage_chunk(st_n, ch_n, w0);
    }
    // ---- store the row (tabulated rows only: a hard row is written whole by the hard-row kernel)
    if (active && !hard) {
      uint64_t *out = S->out;
      uint8_t *oc = S->out_compact;
      if (out) {
        uint32_t *dst = reinterpret_cast<uint32_t *>(out) + (size_t)row * WO32;
        if constexpr (WO32 % 4 == 0) {
#pragma unroll
          for (int d = 0; d < WO32; d += 4) *reinterpret_cast<u32x4 *>(dst + d) = u32x4{o[d], o[d + 1], o[d + 2], o[d + 3]};
        } else {
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
    cur ^= buf_words;
  }
}

}  // namespace tsimk
