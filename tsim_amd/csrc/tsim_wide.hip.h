// tsim_wide.hip.h - ONE pass for programs with one WIDE component (more than 64 parameters, up to 255 selected f bits,
// at most 8 outputs: BASELINE config C5, the d = 5 surface code with an injected T), k_sample_wide.
//
// Round 2-3 served these with three kernels per batch - a pattern-table pass over every row (k_sample_lw<true>), the
// sparse-column kernel on the rows the tables miss (k_sample4w, through row lists), the row kernel on ITS overflow and
// on the normalisation-check row - each re-reading the f rows it was handed: 161 MB of traffic per 10^6 shots for
// 56 MB of algorithmic bytes, 12 waves per CU, a launch triple per batch (profiles/r03/c5_pmc.txt).  This kernel does
// everything sample_program does (reference: src/tsim/sampler.py:117-167, _sample_component :28-81) for up to 16 batches
// in one grid of chip-resident blocks, reading every f row ONCE:
//
//   phase 1, a wave on 64 consecutive rows (one lane = one shot): the rows arrive by LDS-DMA (coalesced, no staging
//     registers; the next chunk's copy is in flight while this one is worked on); direct outputs as bit-field runs; the
//     selected set bits of the row -> their positions inside f_sel (one byte each, ascending); weight <= table depth: colex
//     rank -> the pattern's threshold tree (tsim_lw.hip.h), n_out draws, done.  The draws are taken for EVERY row here
//     (they depend on (key, shot) only), so the rows the tables miss keep theirs;
//   phase 2, the same wave, whenever 64 missed rows have collected in its LDS queue (position list + draws, 28 bytes
//     a row): the sparse-column evaluation of tsim_kernel4w.hip.h - Y_g = XOR of the K <= 12 column entries of the
//     row's set bits and the two outcome-bit chunks, column tables of all levels resident in LDS - on 64 DENSE lanes,
//     then acc_graph4 / acc_finish4, |amp|, p1 / prev against the stored draw, chain rule: the same values as every other
//     kernel of the library.  The component's bits are ORed into the row phase 1 already wrote (32-bit atomics);
//   generic pass, for rows with more than K set bits (3e-4 of C5's rows) and for the normalisation-check row
//     (sampler.py:66-72: lane 0 replays shot 0 with trial bit 1, lane 1 with trial bit 0): the same evaluation with Y_g
//     formed by walking the row's set bits, whatever their number.
//
// No row lists, no second kernel, no launch-plan feedback that results or coverage depend on: the only thing the host
// learns is the share of missed / heavy rows of block 0 (a statistic: deeper tables, or the round-2 path for dense noise).
#pragma once
#include "tsim_kernel4w.hip.h"
#include "tsim_lw_fast.hip.h"

namespace tsimk {

// wide record in the program image (uint32 words, 64-byte aligned)
enum {
  WR_NRUNS = 0,   // direct-output runs, sorted by destination word
  WR_RUNS,        // image offset: n_runs x (src_word | rot << 8, mask)
  WR_RUNB,        // image offset: WO32 + 1 run boundaries (runs of destination word d: [RUNB[d], RUNB[d + 1]))
  WR_FLIPS,       // image offset: WO32 constant-flip words
  WR_LUT,         // image offset: [2^n_out][WO32] placement of the sampled bits (leaf = bits in sampling order, first output most significant)
  WR_LUTMASK,     // bit d set: destination word d holds component outputs
  WR_WO32,        // 32-bit words per output row the record was built for
  WR_COLBYTES,    // bytes of the column tables of all levels (LDS resident)
  WR_TT,          // image offset of (n_out + 1) x (image offset, words rounded up to 4) of the levels' term tables
  WR_TTBYTES,     // their total size in bytes
  WR_GTOT,        // graphs of all levels together
  WR_RUN1,        // image offset of WO32 x (ctl, mask) when every destination word has at most one run (mask 0: none), else 0
  WR_CCOL,        // image offset of the shared column table (F + 33 entries of 16 bytes: every graph's parity bits in one entry), 0: none
  WR_CREC,        // image offset of one word per graph: word of the entry | first bit << 8 | product pairs << 16 | counted rows << 24
  WR_MERGE,       // 1: a later component's pass - no direct outputs, the component's bits are ORed into the rows the first pass wrote
  WR_SELN,        // f words (32 bits) that hold selected bits of the component (<= TSIMK_WIDE_SELMAX): the only ones phase 1 looks at
  WR_SELREC,      // image offset of SELN masks, then SELN x (selected bits in the lower words | word index << 16)
  WR_BSTRIDE,     // words per row of the binomial table C(b, k + 1) at binom_off: 256, or 512 for components beyond 255 selected bits
  WR_KEYSUB,      // the launch's key records start at this compiled output (programs of more than TSIMK_LWM_KEYS outputs: every pass gets ITS component's subkeys)
  WR_WORDS = 32
};
#define TSIMK_WIDE_SELMAX 64   // (f rows of up to 2048 bits since round 5: max_f_index < 512 was the 16 mask words of the round-2 kernels)
#define TSIMK_WIDE_K 12        // set bits per row the dense pass takes
#define TSIMK_WIDE_QCAP 128    // ring capacities (power of two, >= 127)
#define TSIMK_WIDE_MAX_RUNS 256

struct WideStep {
  const uint64_t *f;      // [B, WF] packed error-mechanism rows of this batch
  uint64_t *out;          // [B, WO] padded output rows, or nullptr
  uint8_t *out_compact;   // [B, out_rb] bit_packed rows (any out_rb, any alignment), or nullptr
  float *norm_dev;        // [n_components] or nullptr (pass ci writes entry ci)
  uint32_t keys[2 * TSIMK_LWM_KEYS];  // per-output subkeys of this batch (sampler.py:74,147-148), host-computed
};

struct WideArgs {
  const uint32_t *img;
  const uint32_t *tab;      // integer thresholds (bernoulli_threshold) of THIS pass's component (its LW_TAB offset applied by the launcher)
  long long B;              // rows per batch (< 2^28)
  long long shot_offset;    // in-batch index of row 0, the same for every batch of the group
  int n_steps, chunks_per_step;
  int has_check, out_rb, WF32;
  int lw_off, comp4_off, wr_off, binom_off;
  uint32_t tab_bytes;
  uint32_t *feedback;       // optional (mapped host memory): [4] rows with more than K set bits, [5] rows the tables missed, [6] rows - estimates from block 0
  // LDS layout in bytes, computed by the launcher (tsim_sample.hip: wide_layout)
  int merge;        // 1: a later component's pass (WR_MERGE): rows are completed in place
  int dev_index;    // the component's slot in norm_dev[]
  int compact;      // 1: LDS holds the shared column table (WR_CCOL) instead of one table per graph; needs l_tt >= 0
  int l_rank, l_lut, l_runs, l_sel, l_ptrs, l_keys, l_tt, l_lvl, l_grec, l_wave, wave_bytes, w_q, w_ovf;  // l_tt < 0: the term tables stay in the image
  WideStep step[TSIMK_LWM_MAX_STEPS];
};

// threefry_bits32 with per-lane keys (the generic pass: lanes of one pass may belong to different batches)
__device__ __forceinline__ uint32_t threefry_bits32_v(uint32_t k0, uint32_t k1, uint32_t hi, uint32_t lo) {
  uint32_t x0 = hi, x1 = lo;
  threefry2x32(k0, k1, x0, x1);
  return x0 ^ x1;
}

// Optional phase timers (build with -DTSIMK_WIDE_TRACE, scripts/wide_trace.py): every wave sums the shader-clock cycles it
// spends per phase; lane 0 adds them to a device array at the end.  Compiled out by default.
#ifdef TSIMK_WIDE_TRACE
__device__ unsigned long long tsimk_wide_trace[24];
#define WT_DECL unsigned long long wt_acc[20] = {}, wt_t = __builtin_readcyclecounter(), wt_t0 = wt_t
#define WT_MARK(k) do { const unsigned long long n_ = __builtin_readcyclecounter(); wt_acc[k] += n_ - wt_t; wt_t = n_; } while (0)
#define WT_COUNT(k) do { wt_acc[k] += 1ull; } while (0)
#define WT_FLUSH do { if (lane == 0u) { wt_acc[11] = __builtin_readcyclecounter() - wt_t0; for (int k_ = 0; k_ < 20; ++k_) if (k_ != 12) atomicAdd(&tsimk_wide_trace[k_], wt_acc[k_]); atomicAdd(&tsimk_wide_trace[12], 1ull); } } while (0)
#else
#define WT_DECL do { } while (0)
#define WT_MARK(k) do { } while (0)
#define WT_COUNT(k) do { } while (0)
#define WT_FLUSH do { } while (0)
#endif

// a graph record (G4_* words) held in vector registers: read from the kernel's LDS copy with broadcast reads, the
// words that steer branches pinned as scalars
struct GrecV {
  typedef uint32_t v4 __attribute__((ext_vector_type(4)));
  v4 a, b, c, d;
  __device__ __forceinline__ uint32_t operator[](int i) const {
    const uint32_t v = i < 4 ? a[i & 3] : i < 8 ? b[i & 3] : i < 12 ? c[i & 3] : d[i & 3];
    return (i == G4_FLAGS || i == G4_DBITS || i == G4_TBL2 || i == G4_CFIELD) ? (uint32_t)__builtin_amdgcn_readfirstlane((int)v) : v;
  }
};

// diagnostic builds only (scripts/wide_skip.sh): leave parts out to see what each costs (wrong results)
#ifndef TSIMK_WIDE_SKIP
#define TSIMK_WIDE_SKIP 0
#endif
#ifndef TSIMK_WIDE_STAGE
#define TSIMK_WIDE_STAGE 0   // 1: the next chunk is staged AFTER the dense passes; 2: staged through registers instead of LDS-DMA
#endif

// GLOB: the column tables stay in the program image (the L2) - components with too many graphs for the LDS; the dense and
// generic passes read them there, everything else is the same kernel (an instantiation of its own: the register
// allocation of the resident form, 4 waves per SIMD for C5, is not touched)
// P16: positions inside f_sel as 16-bit fields (components of 256..511 selected bits: six position words per row instead of
// three; class F300 of scripts/shape_map.py) - again an instantiation of its own
template <int WO32, int K, bool GLOB = false, bool P16 = false>
__global__ void __launch_bounds__(1024) k_sample_wide(WideArgs A) {
  typedef const __attribute__((address_space(4))) uint8_t *cbytes;
  typedef const __attribute__((address_space(4))) WideStep *cstep;
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) void *lds_ptr_t;
  typedef const __attribute__((address_space(1))) void *glb_ptr_t;
  typedef __attribute__((address_space(1))) uint32_t gu32;        // output rows: global_store / global_atomic, not their flat forms
  typedef __attribute__((address_space(1))) u32x2 gu32x2;
  typedef __attribute__((address_space(1))) u32x4 gu32x4;
  static_assert(K % 2 == 0 && K <= 12, "position lists are three words");
  constexpr uint32_t QCAP = TSIMK_WIDE_QCAP;
  constexpr uint32_t NP = P16 ? 6u : 3u;   // position words of a queue entry (K positions of 8 or 16 bits)
  constexpr uint32_t QD = 1u + NP;         // queue row of the first draw: id, NP position words, n_out draws, then the output words
  const int nthr = blockDim.x;
  const uint32_t lane = threadIdx.x & 63u, wpb = (uint32_t)nthr >> 6;
  const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform: everything derived from it stays scalar
  cptr img = (cptr)(uintptr_t)A.img;
  cptr comp = img + A.comp4_off;
  cptr rec = img + A.lw_off;
  cptr wr = img + A.wr_off;
  const uint32_t n_out = comp[C_NOUT], F = comp[C_F];
  cptr levels = img + comp[C4_LEVELS];
  const uint32_t WF32 = (uint32_t)A.WF32;
  const uint32_t ent_bytes = (F + 33u) * 16u;  // one graph's column table
  uint8_t *lds8 = reinterpret_cast<uint8_t *>(tsimk_lds);
  uint32_t *l_rank = tsimk_lds + (A.l_rank >> 2);  // [4][F + 1]: C(position, ordinal + 1), 0 at position F
  uint32_t *l_lut = tsimk_lds + (A.l_lut >> 2);    // [2^n_out][WO32]
  uint32_t *l_runs = tsimk_lds + (A.l_runs >> 2);  // runs (2 words each), then RUNB[WO32 + 1], FLIPS[WO32], BASES[8]
  uint32_t *l_sel = tsimk_lds + (A.l_sel >> 2);    // SELMAX selection masks, SELMAX x (prefix count | f word << 16), two statistics counters
  uint32_t *l_ptrs = tsimk_lds + (A.l_ptrs >> 2);  // per step: out (2 words), out_compact (2 words)
  uint32_t *l_keys = tsimk_lds + (A.l_keys >> 2);  // per step: 2 * TSIMK_LWM_KEYS subkey words
  cstep steps = (cstep)((cbytes)__builtin_amdgcn_kernarg_segment_ptr() + __builtin_offsetof(WideArgs, step));

  // ---- once per block: everything loop-invariant into LDS
  const uint32_t n_runs = wr[WR_NRUNS];
  const uint32_t n_sel = wr[WR_SELN];
  {
    uint32_t off = 0;
    if (!GLOB && A.compact) {  // the shared column table: every graph's parity bits in one 16-byte entry per column
      tile_copy(reinterpret_cast<const uint4 *>(A.img + wr[WR_CCOL]), lds8, F + 33u, threadIdx.x, nthr);
    } else if (!GLOB) {
      for (uint32_t li = 0; li <= n_out; ++li) {  // the column tables of every level: one burst of LDS-DMA
        cptr lvl = levels + li * L4_WORDS;
        const uint32_t bytes = lvl[L4_G] * ent_bytes;
        tile_copy(reinterpret_cast<const uint4 *>(A.img + lvl[L4_STAB]), lds8 + off, bytes >> 4, threadIdx.x, nthr);
        off += bytes;
      }
    }
    if (!GLOB && A.l_tt >= 0) {  // the term tables of every level behind each other
      uint32_t toff = 0;
      for (uint32_t li = 0; li <= n_out; ++li) {
        const uint32_t src = img[wr[WR_TT] + 2u * li], words = img[wr[WR_TT] + 2u * li + 1u];
        tile_copy(reinterpret_cast<const uint4 *>(A.img + src), lds8 + A.l_tt + toff, words >> 2, threadIdx.x, nthr);
        toff += words * 4u;
      }
    }
    if (!GLOB && A.l_tt >= 0) {
      // ... and, with them, the level table (G, flags, frame power, first graph) and the graph records, their table offsets
      // turned into LDS word addresses: a dense pass then reads nothing but LDS until it stores its rows
      uint32_t g0 = 0, toff = 0;
      uint32_t *l_lvl = tsimk_lds + (A.l_lvl >> 2), *l_grec = tsimk_lds + (A.l_grec >> 2);
      const uint32_t tt_word0 = ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)lds8 + (uint32_t)A.l_tt) >> 2;
      for (uint32_t li = 0; li <= n_out; ++li) {
        cptr lvl = levels + li * L4_WORDS;
        const uint32_t G = lvl[L4_G], recs = lvl[L4_RECS], src = img[wr[WR_TT] + 2u * li], words = img[wr[WR_TT] + 2u * li + 1u];
        if (threadIdx.x == 0) {
          l_lvl[4u * li] = G;
          l_lvl[4u * li + 1u] = lvl[L4_FLAGS];
          l_lvl[4u * li + 2u] = lvl[L4_FRAME];
          l_lvl[4u * li + 3u] = g0;
        }
        for (uint32_t i = threadIdx.x; i < G * (uint32_t)G4_WORDS; i += nthr) {
          uint32_t v = A.img[recs + i];
          const uint32_t k = i % (uint32_t)G4_WORDS;
          if (k == (uint32_t)G4_TBL || (k == (uint32_t)G4_TBL2 && v != 0u)) v = tt_word0 + (toff >> 2) + (v - src);
          if (k == (uint32_t)G4_CFIELD && !GLOB && A.compact) v = A.img[wr[WR_CREC] + g0 + i / (uint32_t)G4_WORDS];
          l_grec[g0 * (uint32_t)G4_WORDS + i] = v;
        }
        g0 += G;
        toff += words * 4u;
      }
    }
    const uint32_t *g = A.img;
    for (uint32_t i = threadIdx.x; i < 4u * (F + 1u); i += nthr) {
      const uint32_t k = i / (F + 1u), b = i - k * (F + 1u);
      l_rank[i] = b < F ? g[A.binom_off + k * wr[WR_BSTRIDE] + b] : 0u;
    }
    for (uint32_t i = threadIdx.x; i < ((uint32_t)WO32 << n_out); i += nthr) l_lut[i] = g[wr[WR_LUT] + i];
    for (uint32_t i = threadIdx.x; i < 2u * n_runs; i += nthr) l_runs[i] = g[wr[WR_RUNS] + i];
    if (threadIdx.x <= (uint32_t)WO32) l_runs[2u * TSIMK_WIDE_MAX_RUNS + threadIdx.x] = g[wr[WR_RUNB] + threadIdx.x];
    if (threadIdx.x < (uint32_t)WO32) l_runs[2u * TSIMK_WIDE_MAX_RUNS + 16u + threadIdx.x] = g[wr[WR_FLIPS] + threadIdx.x];
    if (threadIdx.x < 8u) l_runs[2u * TSIMK_WIDE_MAX_RUNS + 32u + threadIdx.x] = g[A.lw_off + LW_BASES_INLINE + threadIdx.x];
    for (uint32_t i = threadIdx.x; i < 2u * n_sel; i += nthr) l_sel[i < n_sel ? i : TSIMK_WIDE_SELMAX + (i - n_sel)] = g[wr[WR_SELREC] + i];
    for (uint32_t i = threadIdx.x; i < 4u * (uint32_t)A.n_steps; i += nthr) {
      const uint64_t ptr = (i & 2u) ? (uint64_t)(uintptr_t)steps[i >> 2].out_compact : (uint64_t)(uintptr_t)steps[i >> 2].out;
      l_ptrs[i] = (i & 1u) ? (uint32_t)(ptr >> 32) : (uint32_t)ptr;
    }
    for (uint32_t i = threadIdx.x; i < 2u * TSIMK_LWM_KEYS * (uint32_t)A.n_steps; i += nthr)
      l_keys[i] = steps[i / (2u * TSIMK_LWM_KEYS)].keys[i % (2u * TSIMK_LWM_KEYS)];
    __syncthreads();
  }
  const uint32_t *l_runb = l_runs + 2u * TSIMK_WIDE_MAX_RUNS, *l_flips = l_runb + 16u, *l_bases = l_runb + 32u;
  const uint32_t lds_col0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)lds8;

  // ---- this wave's LDS: the staged f rows, the position lists, the queue of missed rows, the ring of heavy rows
  uint8_t *w8 = lds8 + A.l_wave + wv * (uint32_t)A.wave_bytes;
  uint32_t *w_f = reinterpret_cast<uint32_t *>(w8);                  // [64][WF32] rows as they lie in HBM
  uint32_t *w_q = reinterpret_cast<uint32_t *>(w8 + A.w_q);          // [4 + n_out + ncw][QCAP]: id, 3 position words, draws, the direct bits of the ncw output words that hold component bits
  uint32_t *w_ovf = reinterpret_cast<uint32_t *>(w8 + A.w_ovf);      // [QCAP] ids
  uint32_t qhead = 0, qtail = 0, ohead = 0, otail = 0;              // wave-uniform
  uint32_t n_missed = 0, n_heavy = 0;

  const uint32_t wmax = rec[LW_WMAX];
  const uint32_t tab_byte = 0u;  // (A.tab is the COMPONENT's table since round 5: offsets below 4 GiB per component, not per program)
  const uint32_t keybase = rec[LW_KEYBASE] - wr[WR_KEYSUB];
  const uint32_t lutmask = wr[WR_LUTMASK];
  const __amdgpu_buffer_rsrc_t r_tab = __builtin_amdgcn_make_buffer_rsrc((void *)A.tab, 0, A.tab_bytes, 0x00020000);
  const uint32_t so_lo = (uint32_t)A.shot_offset, so_hi = (uint32_t)((unsigned long long)A.shot_offset >> 32);
  const uint32_t Bu = (uint32_t)A.B;
  const uint32_t cps = (uint32_t)A.chunks_per_step;
  const uint32_t total = cps * (uint32_t)A.n_steps;
  const uint32_t zsplat = F * (P16 ? 0x00010001u : 0x01010101u);
  // direct outputs with at most one run per destination word (identity-like tables): descriptors as scalars
  const bool run1 = wr[WR_RUN1] != 0u;
  uint32_t r1_ctl[WO32], r1_mask[WO32], r1_flip[WO32];
#pragma unroll
  for (int d = 0; d < WO32; ++d) {
    r1_ctl[d] = run1 ? img[wr[WR_RUN1] + 2u * (uint32_t)d] : 0u;
    r1_mask[d] = run1 ? img[wr[WR_RUN1] + 2u * (uint32_t)d + 1u] : 0u;
    r1_flip[d] = img[wr[WR_FLIPS] + (uint32_t)d];
  }

  // the f rows of chunk c -> w_f (LDS-DMA: 64 consecutive dwords per instruction, rows as they lie in HBM)
  auto stage_chunk = [&](uint32_t st, uint32_t ch) {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(steps[st].f) + (size_t)ch * 64u * WF32;
    const uint32_t valid = (Bu - ch * 64u < 64u ? Bu - ch * 64u : 64u) * WF32;  // dwords of this chunk inside the batch
#if TSIMK_WIDE_STAGE == 2
    uint32_t v[16];
#pragma unroll
    for (int w = 0; w < 16; ++w) v[w] = ((uint32_t)w < WF32 && (uint32_t)w * 64u + lane < valid) ? src[(uint32_t)w * 64u + lane] : 0u;
#pragma unroll
    for (int w = 0; w < 16; ++w)
      if ((uint32_t)w < WF32) w_f[(uint32_t)w * 64u + lane] = v[w];
#else
    if (valid == 64u * WF32) {  // (all but a batch's last chunk)
      for (uint32_t w = 0; w < WF32; ++w) __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + w * 64u + lane), (lds_ptr_t)(w_f + w * 64u), 4, 0, 0);
    } else {
      for (uint32_t w = 0; w < WF32; ++w) {
        const uint32_t j = w * 64u + lane;
        if (j < valid) __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + j), (lds_ptr_t)(w_f + w * 64u), 4, 0, 0);
      }
    }
#endif
  };
  // K14: direct outputs f[idx] ^ flip (sampler.py:140-145) of the row staged at frow: rotate-and-mask runs per destination word
  auto direct_words = [&](const uint32_t *frow, uint32_t (&o)[WO32]) {
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
  };
  // word d of a bit_packed row of out_rb bytes (any out_rb since round 5: rows of 9 bytes start at odd addresses): a dword
  // access at the row's own byte offset - gfx950 takes unaligned global dwords - or, for the row's last 1..3 bytes, bytes.
  // Every byte touched belongs to this row.
  typedef __attribute__((address_space(1))) uint8_t gu8;
  typedef uint32_t u32_una __attribute__((aligned(1)));
  typedef __attribute__((address_space(1))) u32_una gu32_una;
  const uint32_t out_rb = (uint32_t)A.out_rb;
  auto oc_put = [&](gu8 *rowp, uint32_t d, uint32_t v, bool merge) {
    if (4u * d + 4u <= out_rb) {
      gu32_una *q = (gu32_una *)(rowp + 4u * d);
      *q = merge ? (*q | v) : v;
    } else {
      for (uint32_t b = 4u * d; b < out_rb; ++b) {
        const uint8_t x = (uint8_t)(v >> (8u * (b - 4u * d)));
        rowp[b] = merge ? (uint8_t)(rowp[b] | x) : x;
      }
    }
  };
  // a row's words -> HBM (every word of the padded row / the whole bit_packed row: dword stores)
  auto store_row = [&](uint32_t st_lo_out, uint32_t st_hi_out, uint32_t st_lo_oc, uint32_t st_hi_oc, uint32_t row, const uint32_t (&o)[WO32]) {
    gu32 *out = (gu32 *)(uintptr_t)(((uint64_t)st_hi_out << 32) | st_lo_out);
    gu32 *oc = (gu32 *)(uintptr_t)(((uint64_t)st_hi_oc << 32) | st_lo_oc);
    if (out) {
      gu32 *dst = out + (size_t)row * WO32;
      if constexpr (WO32 == 4) *(gu32x4 *)dst = u32x4{o[0], o[1], o[2], o[3]};
      else {
#pragma unroll
        for (int d = 0; d < WO32; d += 2) *(gu32x2 *)(dst + d) = u32x2{o[d], o[d + 1]};
      }
    }
    if (oc) {
      if (WO32 == 4 && out_rb == 16u && (((uintptr_t)oc) & 15u) == 0u) *(gu32x4 *)(oc + (size_t)row * 4u) = u32x4{o[0], o[1], o[2], o[WO32 > 3 ? 3 : 0]};
      else {
        gu8 *rowp = (gu8 *)oc + (size_t)row * out_rb;
#pragma unroll
        for (int d = 0; d < WO32; ++d)
          if (4u * (uint32_t)d < out_rb) oc_put(rowp, (uint32_t)d, o[d], false);
      }
    }
  };

  // a later component's pass (A.merge): OR this component's bits into the words of the row that hold them - the row is in
  // HBM since the first pass, nothing else touches it in this grid (oc_put touches the row's own bytes only)
  auto merge_row = [&](uint32_t st_lo_out, uint32_t st_hi_out, uint32_t st_lo_oc, uint32_t st_hi_oc, uint32_t row, const uint32_t (&o)[WO32]) {
    gu32 *out = (gu32 *)(uintptr_t)(((uint64_t)st_hi_out << 32) | st_lo_out);
    gu32 *oc = (gu32 *)(uintptr_t)(((uint64_t)st_hi_oc << 32) | st_lo_oc);
#pragma unroll
    for (int d = 0; d < WO32; ++d)
      if (((lutmask >> d) & 1u) && o[d] != 0u) {
        if (out) out[(size_t)row * WO32 + d] |= o[d];
        if (oc && 4u * (uint32_t)d < out_rb) oc_put((gu8 *)oc + (size_t)row * out_rb, (uint32_t)d, o[d], true);
      }
  };

  // shared column table: graph field `cf` (WR_CREC) of the entry-sized parity word yl -> the four words acc_graph4 takes
  // (U, V: product rows; O1: counted rows, the masks of the record pick theirs; O2: index bits, lambda and linear on top)
  auto field_words = [&](const u32x4 &yl, uint32_t cf, uint32_t dsh) -> u32x4 {
    const uint32_t w = cf & 3u, off = (cf >> 8) & 31u, h2 = (cf >> 16) & 63u, s2 = 2u * h2 + (cf >> 24);
    const uint32_t fld = ((w & 2u) ? ((w & 1u) ? yl.w : yl.z) : ((w & 1u) ? yl.y : yl.x)) >> off;
    u32x4 y;
    y.x = fld;
    y.y = fld >> h2;
    y.z = fld >> (2u * h2);
    y.w = ((fld >> s2) & ((1u << dsh) - 1u)) | ((fld >> (s2 + dsh)) << 30);
    return y;
  };

  WT_DECL;
  // ------------------------------------------------------------------------------------------------------------
  // phase 2: n (<= 64) queued rows, slots qhead .. qhead + n - 1: the sparse-column evaluation on dense lanes
  // ------------------------------------------------------------------------------------------------------------
  auto dense_pass = [&](uint32_t n) {
    typedef const __attribute__((address_space(3))) u32x4 *lds_u4p;
    const bool on = lane < n;
    const uint32_t slot = (qhead + lane) & (QCAP - 1u);
    const uint32_t id = on ? w_q[slot] : 0u;
    const uint32_t p0 = on ? w_q[QCAP + slot] : zsplat, p1 = on ? w_q[2u * QCAP + slot] : zsplat, p2 = on ? w_q[3u * QCAP + slot] : zsplat;
    uint32_t p3 = zsplat, p4 = zsplat, p5 = zsplat;
    if constexpr (P16) {
      p3 = on ? w_q[4u * QCAP + slot] : zsplat;
      p4 = on ? w_q[5u * QCAP + slot] : zsplat;
      p5 = on ? w_q[6u * QCAP + slot] : zsplat;
    }
    uint32_t col[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if constexpr (P16) {
        const uint32_t pw = (k < 2) ? p0 : (k < 4) ? p1 : (k < 6) ? p2 : (k < 8) ? p3 : (k < 10) ? p4 : p5;
        col[k] = (k & 1) ? ((pw >> 12) & 0xFFFF0u) : ((pw << 4) & 0xFFFF0u);
      } else {
        const uint32_t pw = (k < 4) ? p0 : (k < 8) ? p1 : p2;
        col[k] = (k & 3) == 0 ? ((pw << 4) & 0xFF0u) : ((pw >> (8 * (k & 3) - 4)) & 0xFF0u);
      }
    }
    // which pairs of column reads any lane of this pass needs (a row of weight w uses entries 0 .. w - 1; the others are the
    // zero column): bit c = some lane has more than 2 c set bits
    uint32_t need = 0u;
#pragma unroll
    for (int c = 0; c < K / 2; ++c)
      if (__builtin_amdgcn_ballot_w64(col[2 * c] != F * 16u) != 0ull) need |= 1u << c;
    uint32_t mb = 0, leaf = 0, lvl_off = 0, tt_lds = 0;
    float prev = 0.0f;
    u32x4 yf = {0u, 0u, 0u, 0u};  // shared column table: the f part of every graph's parity words, once per pass
    if (!GLOB && A.compact) {
#pragma unroll
      for (int c = 0; c < K / 2; ++c)
        if ((need >> c) & 1u) {
          const u32x4 v = *(lds_u4p)(uintptr_t)(lds_col0 + col[2 * c]);
          const u32x4 w = *(lds_u4p)(uintptr_t)(lds_col0 + col[2 * c + 1]);
          yf.x = xor3(yf.x, v.x, w.x); yf.y = xor3(yf.y, v.y, w.y); yf.z = xor3(yf.z, v.z, w.z); yf.w = xor3(yf.w, v.w, w.w);
        }
    }
#ifdef TSIMK_WIDE_TRACE
    asm volatile("" :: "v"(col[0]), "v"(col[K - 1]));
#endif
    WT_MARK(13);
    for (uint32_t li = 0; li <= ((TSIMK_WIDE_SKIP & 1) ? 0u : n_out); ++li) {
      if (li > 0) mb |= 1u << (li - 1u);  // trial bit 1 (sampler.py:65)
      const uint32_t e_lo = (F + 1u + (mb & 15u)) * 16u, e_hi = (F + 17u + ((mb >> 4) & 15u)) * 16u;
      float re, im;
      if (!GLOB && A.l_tt >= 0) {
        // everything of the level from LDS: its record, the graph records, the column entries, the term tables
        const u32x4 lv = *reinterpret_cast<const u32x4 *>(&tsimk_lds[(A.l_lvl >> 2) + 4u * li]);
        const uint32_t G = (uint32_t)__builtin_amdgcn_readfirstlane((int)lv.x), flags = (uint32_t)__builtin_amdgcn_readfirstlane((int)lv.y);
        const int frame = __builtin_amdgcn_readfirstlane((int)lv.z);
        const uint32_t g0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)lv.w);
        const bool fixed = (flags & TSIMK_LFLAG_FIXED) != 0, approx = (flags & TSIMK_LFLAG_APPROX) != 0;
        Acc4 S;
        u32x4 yl = yf;
        if (!GLOB && A.compact) {  // + the level's outcome bits and row constants: two more entries for ALL its graphs
          const u32x4 v = *(lds_u4p)(uintptr_t)(lds_col0 + e_lo);
          const u32x4 w = *(lds_u4p)(uintptr_t)(lds_col0 + e_hi);
          yl.x = xor3(yl.x, v.x, w.x); yl.y = xor3(yl.y, v.y, w.y); yl.z = xor3(yl.z, v.z, w.z); yl.w = xor3(yl.w, v.w, w.w);
        }
        for (uint32_t g = 0; g < G; ++g) {
          const uint32_t base = lds_col0 + lvl_off + g * ent_bytes;
          const uint32_t rb = lds_col0 + (uint32_t)A.l_grec + (g0 + g) * (uint32_t)(G4_WORDS * 4);
          GrecV R;
          R.a = *(lds_u4p)(uintptr_t)rb;
          R.b = *(lds_u4p)(uintptr_t)(rb + 16u);
          R.c = *(lds_u4p)(uintptr_t)(rb + 32u);
          R.d = R.c;
          if (approx) R.d = *(lds_u4p)(uintptr_t)(rb + 48u);
          u32x4 y;
          if (!GLOB && A.compact) {
            y = field_words(yl, R[G4_CFIELD], R[G4_DBITS]);
          } else {
            y = *(lds_u4p)(uintptr_t)(base + e_lo);
            {
              const u32x4 t = *(lds_u4p)(uintptr_t)(base + e_hi);
              y ^= t;
            }
#pragma unroll
            for (int c = 0; c < K / 2; ++c)
              if ((need >> c) & 1u) {
                const u32x4 v = *(lds_u4p)(uintptr_t)(base + col[2 * c]);
                const u32x4 w = *(lds_u4p)(uintptr_t)(base + col[2 * c + 1]);
                y.x = xor3(y.x, v.x, w.x); y.y = xor3(y.y, v.y, w.y); y.z = xor3(y.z, v.z, w.z); y.w = xor3(y.w, v.w, w.w);
              }
          }
          if (fixed) acc_graph4<true, true, const GrecV &>(S, A.img, R, y.x, y.y, y.z, y.w, approx, 0u);
          else acc_graph4<false, true, const GrecV &>(S, A.img, R, y.x, y.y, y.z, y.w, approx, 0u);
        }
        if (fixed) acc_finish4<true>(S, frame, approx, re, im);
        else acc_finish4<false>(S, frame, approx, re, im);
        lvl_off += G * ent_bytes;
      } else {
        cptr lvl = levels + li * L4_WORDS;
        uint32_t e[K + 2];
#pragma unroll
        for (int k = 0; k < K; ++k) e[k] = col[k];
        e[K] = e_lo;
        e[K + 1] = e_hi;
        if (GLOB) {
          if ((lvl[L4_FLAGS] & TSIMK_LFLAG_FIXED) != 0) eval_level4_global<K + 2, true>(A.img, img, lvl, e, ent_bytes, re, im);
          else eval_level4_global<K + 2, false>(A.img, img, lvl, e, ent_bytes, re, im);
        } else {
          if ((lvl[L4_FLAGS] & TSIMK_LFLAG_FIXED) != 0) eval_level4_resident<K + 2, true>(A.img, img, lvl, e, lds_col0 + lvl_off, ent_bytes, re, im);
          else eval_level4_resident<K + 2, false>(A.img, img, lvl, e, lds_col0 + lvl_off, ent_bytes, re, im);
        }
        lvl_off += lvl[L4_G] * ent_bytes;
      }
#ifdef TSIMK_WIDE_TRACE
      asm volatile("" :: "v"(re), "v"(im));
#endif
      WT_MARK(14);
      const float v1 = cabs32(re, im);
      if (li == 0) { prev = v1; continue; }
      const uint32_t i = li - 1u;
      const uint32_t m = w_q[(QD + i) * QCAP + slot];  // the draw phase 1 took for this output: bits >> 9
      const float u = __uint_as_float(m | 0x3F800000u) - 1.0f;
      const bool bit = u < __fdiv_rn(v1, prev);  // sampler.py:74-75
      if (!bit) mb &= ~(1u << i);
      prev = bit ? v1 : __fsub_rn(prev, v1);     // sampler.py:79
      leaf = 2u * leaf + (bit ? 1u : 0u);
#ifdef TSIMK_WIDE_TRACE
      asm volatile("" :: "v"(leaf), "v"(prev));
#endif
      WT_MARK(15);
    }
    (void)tt_lds;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // phase 1 stored these rows whole (direct bits only): those stores have landed
    if (on && !(TSIMK_WIDE_SKIP & 2)) {
      // the words that hold component outputs are rewritten here, complete: their direct bits waited in the queue.
      // (An atomic OR into a stored row - any scope - leaves the XCD's L2 for the fabric: 0.57e6 of them cost C5 90 of
      // 122 us per 10^6 shots, profiles/r04/wide_skip.txt.)
      const uint32_t st = id >> 28, row = id & 0x0FFFFFFFu;
      const u32x4 pt = *reinterpret_cast<const u32x4 *>(&l_ptrs[4u * st]);
      gu32 *out = (gu32 *)(uintptr_t)(((uint64_t)pt.y << 32) | pt.x);
      gu32 *oc = (gu32 *)(uintptr_t)(((uint64_t)pt.w << 32) | pt.z);
      uint32_t cwi = 0;
#pragma unroll
      for (int d = 0; d < WO32; ++d)
        if ((lutmask >> d) & 1u) {
          const uint32_t v = w_q[(QD + n_out + cwi) * QCAP + slot] | l_lut[leaf * (uint32_t)WO32 + (uint32_t)d];
          ++cwi;
          if (A.merge) {  // (the queue word is 0: the first pass wrote the direct bits)
            if (v != 0u) {
              if (out) out[(size_t)row * WO32 + d] |= v;
              if (oc && 4u * (uint32_t)d < out_rb) oc_put((gu8 *)oc + (size_t)row * out_rb, (uint32_t)d, v, true);
            }
          } else {
            if (out) out[(size_t)row * WO32 + d] = v;
            if (oc && 4u * (uint32_t)d < out_rb) oc_put((gu8 *)oc + (size_t)row * out_rb, (uint32_t)d, v, false);
          }
        }
    }
    WT_MARK(16);
    qhead += n;
  };

  // ------------------------------------------------------------------------------------------------------------
  // generic pass: rows of any weight (ids from the ring of heavy rows), or - check - the normalisation check of batch
  // `check_step`: lanes 0 and 1 replay its row 0 with trial bit 1 / 0 (sampler.py:66-72).  Uses w_f as its staging.
  // ------------------------------------------------------------------------------------------------------------
  auto generic_pass = [&](uint32_t n, bool check, uint32_t check_step) {
    const bool on = check ? (lane < 2u) : (lane < n);
    const uint32_t id = check ? (check_step << 28) : (on ? w_ovf[(ohead + lane) & (QCAP - 1u)] : 0u);
    const uint32_t st = id >> 28, row = id & 0x0FFFFFFFu;
    const bool trial0 = check && lane == 1u;
    uint32_t *frow = w_f + lane * WF32;
    {
      const uint32_t *src = reinterpret_cast<const uint32_t *>(steps[0].f);
      {  // the lane's batch: its f pointer from the kernel arguments (16 candidates, select by step)
        uint64_t fp = (uint64_t)(uintptr_t)steps[0].f;
        for (int s = 1; s < A.n_steps; ++s) fp = (st == (uint32_t)s) ? (uint64_t)(uintptr_t)steps[s].f : fp;
        src = reinterpret_cast<const uint32_t *>((uintptr_t)fp);
      }
      for (uint32_t w = 0; w < WF32; ++w) frow[w] = on ? src[(size_t)row * WF32 + w] : 0u;
    }
    uint32_t o[WO32];
    direct_words(frow, o);
    const uint32_t *keys = l_keys + st * (2u * TSIMK_LWM_KEYS) + 2u * keybase;
    const uint32_t slo = so_lo + row;
    uint32_t mb = 0, leaf = 0, lvl_off = 0, gidx = 0;
    float prev = 0.0f, maxdev = 0.0f;
    // XOR of the column entries of the row's set selected bits, from the table at `tbl`
    auto walk = [&](const uint8_t *tbl, u32x4 &y) {
      for (uint32_t w = 0; w < n_sel; ++w) {
        const uint32_t sw = l_sel[w], bw = l_sel[TSIMK_WIDE_SELMAX + w], base = bw & 0xFFFFu;
        uint32_t m = frow[bw >> 16] & sw;
        while (m) {
          const uint32_t p = (uint32_t)__builtin_ctz(m);
          const uint32_t pos = base + (uint32_t)__builtin_popcount(sw & ((1u << p) - 1u));
          const u32x4 t = *reinterpret_cast<const u32x4 *>(tbl + pos * 16u);
          y.x ^= t.x; y.y ^= t.y; y.z ^= t.z; y.w ^= t.w;
          m &= m - 1u;
        }
      }
    };
    u32x4 yf = {0u, 0u, 0u, 0u};
    if (!GLOB && A.compact) walk(lds8, yf);  // shared column table: one walk for every graph of every level
    for (uint32_t li = 0; li <= n_out; ++li) {
      cptr lvl = levels + li * L4_WORDS;
      if (li > 0) mb = trial0 ? (mb & ~(1u << (li - 1u))) : (mb | (1u << (li - 1u)));
      const uint32_t G = lvl[L4_G];
      const bool fixed = (lvl[L4_FLAGS] & TSIMK_LFLAG_FIXED) != 0, approx = (lvl[L4_FLAGS] & TSIMK_LFLAG_APPROX) != 0;
      cptr recs = img + lvl[L4_RECS];
      Acc4 S;
      u32x4 yl = yf;
      if (!GLOB && A.compact) {
        const u32x4 v = *reinterpret_cast<const u32x4 *>(lds8 + (F + 1u + (mb & 15u)) * 16u);
        const u32x4 t = *reinterpret_cast<const u32x4 *>(lds8 + (F + 17u + ((mb >> 4) & 15u)) * 16u);
        yl.x ^= v.x ^ t.x; yl.y ^= v.y ^ t.y; yl.z ^= v.z ^ t.z; yl.w ^= v.w ^ t.w;
      }
      for (uint32_t g = 0; g < G; ++g, ++gidx) {
        u32x4 y;
        if (!GLOB && A.compact) {
          y = field_words(yl, img[wr[WR_CREC] + gidx], recs[g * G4_WORDS + G4_DBITS]);
        } else {
          const uint8_t *tbl = GLOB ? reinterpret_cast<const uint8_t *>(A.img + lvl[L4_STAB]) + (size_t)g * ent_bytes : lds8 + lvl_off + g * ent_bytes;
          y = *reinterpret_cast<const u32x4 *>(tbl + (F + 1u + (mb & 15u)) * 16u);
          {
            const u32x4 t = *reinterpret_cast<const u32x4 *>(tbl + (F + 17u + ((mb >> 4) & 15u)) * 16u);
            y.x ^= t.x; y.y ^= t.y; y.z ^= t.z; y.w ^= t.w;
          }
          walk(tbl, y);
        }
        if (fixed) acc_graph4<true>(S, A.img, recs + g * G4_WORDS, y.x, y.y, y.z, y.w, approx);
        else acc_graph4<false>(S, A.img, recs + g * G4_WORDS, y.x, y.y, y.z, y.w, approx);
      }
      float re, im;
      if (fixed) acc_finish4<true>(S, lvl, approx, re, im);
      else acc_finish4<false>(S, lvl, approx, re, im);
      lvl_off += G * ent_bytes;
      float v1 = cabs32(re, im), v0 = 0.0f;
      if (check) {  // wave-uniform
        v0 = __shfl(v1, 1, 64);
        v1 = __shfl(v1, 0, 64);
      }
      if (li == 0) { prev = v1; continue; }
      const uint32_t i = li - 1u;
      if (check) {
        const float norm = __fdiv_rn(__fadd_rn(v0, v1), prev);  // sampler.py:71
        maxdev = nanmax(maxdev, fabsf(__fsub_rn(norm, 1.0f)));  // sampler.py:72
      }
      const uint32_t m = threefry_bits32_v(keys[2u * i], keys[2u * i + 1u], so_hi, slo) >> 9;
      const float u = __uint_as_float(m | 0x3F800000u) - 1.0f;
      const bool bit = u < __fdiv_rn(v1, prev);
      mb = bit ? (mb | (1u << i)) : (mb & ~(1u << i));
      prev = bit ? v1 : __fsub_rn(prev, v1);
      leaf = 2u * leaf + (bit ? 1u : 0u);
    }
    if (check) {
      float *nd = steps[check_step].norm_dev;
      if (lane == 0u && nd) nd[A.dev_index] = maxdev;  // one deviation per component (include/tsim_hip.h: max_norm_dev[n_components])
    } else {
      if (on) {
#pragma unroll
        for (int d = 0; d < WO32; ++d) o[d] |= l_lut[leaf * (uint32_t)WO32 + (uint32_t)d];
        const u32x4 pt = *reinterpret_cast<const u32x4 *>(&l_ptrs[4u * st]);
        if (A.merge) merge_row(pt.x, pt.y, pt.z, pt.w, row, o);
        else store_row(pt.x, pt.y, pt.z, pt.w, row, o);
      }
      ohead += n;
    }
  };

  // ------------------------------------------------------------------------------------------------------------
  // the chunks of this wave: c = global wave index, + waves of the grid, ...  One loop, one call site per pass:
  //   w_f free (no copy in flight): the normalisation check / heavy rows that are due, then the next chunk's copy;
  //   dense passes while 64 missed rows are queued (the copy travels meanwhile); then the chunk itself.
  // ------------------------------------------------------------------------------------------------------------
  const uint32_t tw = gridDim.x * wpb;
  uint32_t c = blockIdx.x * wpb + wv;
  uint32_t st = c / cps, ch = c - st * cps;  // once; afterwards by increments
  const uint32_t tw_st = tw / cps, tw_ch = tw - tw_st * cps;
  bool staged = false;
  uint32_t check_pending = 0u;  // step + 1 of a batch whose normalisation check is due
  for (;;) {
    const bool done = c >= total;
    if (!staged) {
      if (check_pending) {
        WT_MARK(0);
        generic_pass(0u, true, check_pending - 1u);
        check_pending = 0u;
        WT_MARK(7);
        continue;
      }
      if (otail - ohead >= 64u || (done && otail != ohead)) {
        WT_MARK(0);
        generic_pass(otail - ohead < 64u ? otail - ohead : 64u, false, 0u);
        WT_MARK(7);
        continue;
      }
      if (!done && !(TSIMK_WIDE_STAGE == 1 && qtail - qhead >= 64u)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every read of w_f has returned
        stage_chunk(st, ch);
        staged = true;
      }
    }
    if (qtail - qhead >= 64u || (done && qtail != qhead)) {
      WT_MARK(0);
      dense_pass(qtail - qhead < 64u ? qtail - qhead : 64u);
      WT_MARK(6);
      WT_COUNT(9);
      continue;
    }
    if (done) break;
    cstep S = steps + st;
    const uint32_t row = ch * 64u + lane;
    const bool active = row < Bu;
    WT_MARK(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this chunk's rows are in w_f (and every earlier store of this wave has left)
    WT_MARK(1);
    WT_COUNT(8);
    const uint32_t *frow = w_f + lane * WF32;
    // ---- direct outputs
    uint32_t o[WO32];
    if (run1) {  // at most one run per destination word: the descriptors are scalars, the WO32 row reads are independent
#pragma unroll
      for (int d = 0; d < WO32; ++d) {
        const uint32_t fw = frow[r1_ctl[d] & 255u];
        o[d] = r1_flip[d] ^ (__builtin_amdgcn_alignbit(fw, fw, r1_ctl[d] >> 8) & r1_mask[d]);
      }
    } else {
      direct_words(frow, o);
    }
#ifdef TSIMK_WIDE_TRACE
    asm volatile("" :: "v"(o[0]), "v"(o[WO32 - 1]));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    WT_MARK(2);
    // ---- the selected set bits: weight, and their positions inside f_sel as bytes of (l0, l1, l2), the newest (largest)
    // in byte 0; F = "none".  The next word's mask, prefix count and f word are requested before this word's bits are
    // walked (one LDS latency per word, under the loop); nothing in the loop touches memory.
    uint32_t l0 = zsplat, l1 = zsplat, l2 = zsplat, cnt = 0;
    uint32_t l3 = zsplat, l4 = zsplat, l5 = zsplat;  // (P16 only)
    {
      const uint32_t nw = n_sel;
      uint32_t n_sw = l_sel[0], n_base = l_sel[TSIMK_WIDE_SELMAX], n_fw = frow[n_base >> 16];
      for (uint32_t w = 0; w < ((TSIMK_WIDE_SKIP & 16) ? 0u : nw); ++w) {
        const uint32_t sw = n_sw, base = n_base & 0xFFFFu;
        uint32_t m = active ? (n_fw & sw) : 0u;
        if (w + 1u < nw) {
          n_sw = l_sel[w + 1u];
          n_base = l_sel[TSIMK_WIDE_SELMAX + 1u + w];
          n_fw = frow[n_base >> 16];
        }
        while (m) {
          const uint32_t p = (uint32_t)__builtin_ctz(m);
          const uint32_t pos = base + (uint32_t)__builtin_popcount(sw & ((1u << p) - 1u));
          if constexpr (P16) {
            l5 = __builtin_amdgcn_alignbit(l5, l4, 16);
            l4 = __builtin_amdgcn_alignbit(l4, l3, 16);
            l3 = __builtin_amdgcn_alignbit(l3, l2, 16);
            l2 = __builtin_amdgcn_alignbit(l2, l1, 16);
            l1 = __builtin_amdgcn_alignbit(l1, l0, 16);
            l0 = (l0 << 16) | pos;
          } else {
            l2 = __builtin_amdgcn_alignbit(l2, l1, 24);
            l1 = __builtin_amdgcn_alignbit(l1, l0, 24);
            l0 = (l0 << 8) | pos;
          }
          ++cnt;
          m &= m - 1u;
        }
      }
    }
#ifdef TSIMK_WIDE_TRACE
    asm volatile("" :: "v"(cnt));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    WT_MARK(3);
    const bool hit = active && cnt <= wmax;
    const bool miss = active && cnt > wmax && cnt <= (uint32_t)K;
    const bool heavy = active && cnt > (uint32_t)K;
    const unsigned long long hm = __builtin_amdgcn_ballot_w64(heavy);
    // ---- colex rank of a tabulated pattern: sum over its set bits of C(position, ordinal + 1) (tsim_lw.hip.h); byte j of
    // l0 is the bit of ordinal cnt - 1 - j (unused bytes hold F: the zero at the end of every RANK row)
    uint32_t pat = l_bases[cnt < 7u ? cnt : 7u];
    if constexpr (P16) {
#pragma unroll
      for (int j = 0; j < TSIMK_LWW_MAX_WEIGHT; ++j) pat += l_rank[((cnt - 1u - (uint32_t)j) & 3u) * (F + 1u) + (((j < 2 ? l0 : l1) >> (16 * (j & 1))) & 0xFFFFu)];
    } else {
#pragma unroll
      for (int j = 0; j < TSIMK_LWW_MAX_WEIGHT; ++j) pat += l_rank[((cnt - 1u - (uint32_t)j) & 3u) * (F + 1u) + ((l0 >> (8 * j)) & 255u)];
    }
    pat = hit ? pat : 0u;
    const uint32_t thr = (TSIMK_WIDE_SKIP & 4) ? 0u : tab_byte + (pat << (n_out + 2u));  // byte offset of the pattern's threshold tree
    // the first three levels of the tree are requested now: the draws below run while they travel
    const uint32_t ta0 = __builtin_amdgcn_raw_buffer_load_b32(r_tab, thr + 4u, 0, 0);
    const u32x2 ta1 = __builtin_amdgcn_raw_buffer_load_b64(r_tab, thr + 8u, 0, 0);
    const u32x4 ta2 = __builtin_amdgcn_raw_buffer_load_b128(r_tab, thr + 16u, 0, 0);
    // ---- w_f is free: the NEXT chunk's rows start to travel now, behind the threshold reads (loads return in order) - unless a
    // generic pass is due before it, which stages rows of its own in w_f (the loop's head then does both)
    uint32_t st_n = st + tw_st, ch_n = ch + tw_ch;
    if (ch_n >= cps) { ch_n -= cps; ++st_n; }
    bool staged_next = false;
    if (c + tw < total && !(A.has_check && ch == 0u) && otail - ohead + (uint32_t)__popcll(hm) < 64u && TSIMK_WIDE_STAGE != 1) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every read of w_f has returned
      stage_chunk(st_n, ch_n);
      staged_next = true;
    }
    // ---- the draws of every output (sampler.py:74-75): functions of (subkey, shot) only
    const uint32_t slo = so_lo + row;  // (the launcher keeps shot_offset + B below the next multiple of 2^32)
    cptr kp = (cptr)((cbytes)S + __builtin_offsetof(WideStep, keys)) + 2u * keybase;
    uint32_t dr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      dr[i] = 0u;
      if ((uint32_t)i < n_out) {
        const uint32_t k0 = kp[2 * i], k1 = kp[2 * i + 1];
        dr[i] = (TSIMK_WIDE_SKIP & 8) ? ((slo * 0x9E3779B9u + k0 + k1) >> 9) : (threefry_bits32_lo(k0, k1, k0 + so_hi, slo) >> 9);
      }
    }
    // ---- tabulated rows: walk the threshold tree, three levels per read (tsim_lw_pass.hip.h: lw_walk_impl)
    uint32_t node = 1u;
    {
      const bool b0 = dr[0] < ta0;
      const bool b1 = dr[1] < (b0 ? ta1.y : ta1.x);
      const uint32_t lo = b1 ? ta2.y : ta2.x, hi = b1 ? ta2.w : ta2.z;
      const bool b2 = dr[2] < (b0 ? hi : lo);
      node = n_out >= 3u ? (8u + (b0 ? 4u : 0u) + (b1 ? 2u : 0u) + (b2 ? 1u : 0u)) : n_out == 2u ? (4u + (b0 ? 2u : 0u) + (b1 ? 1u : 0u)) : (2u + (b0 ? 1u : 0u));
    }
    if (n_out > 3u) {
      if (n_out >= 6u) {
        const uint32_t t0 = __builtin_amdgcn_raw_buffer_load_b32(r_tab, thr + 4u * node, 0, 0);
        const u32x2 t1 = __builtin_amdgcn_raw_buffer_load_b64(r_tab, thr + 8u * node, 0, 0);
        const u32x4 t2 = __builtin_amdgcn_raw_buffer_load_b128(r_tab, thr + 16u * node, 0, 0);
        const bool b0 = dr[3] < t0;
        const bool b1 = dr[4] < (b0 ? t1.y : t1.x);
        const uint32_t lo = b1 ? t2.y : t2.x, hi = b1 ? t2.w : t2.z;
        const bool b2 = dr[5] < (b0 ? hi : lo);
        node = 8u * node + (b0 ? 4u : 0u) + (b1 ? 2u : 0u) + (b2 ? 1u : 0u);
      }
#pragma unroll
      for (int i = 3; i <= 6; i += 3)
        if (n_out / 3u * 3u == (uint32_t)i && n_out % 3u != 0u) {
          const uint32_t t0 = __builtin_amdgcn_raw_buffer_load_b32(r_tab, thr + 4u * node, 0, 0);
          const bool b0 = dr[i] < t0;
          if (n_out % 3u == 2u) {
            const u32x2 t1 = __builtin_amdgcn_raw_buffer_load_b64(r_tab, thr + 8u * node, 0, 0);
            const bool b1 = dr[i + 1] < (b0 ? t1.y : t1.x);
            node = 4u * node + (b0 ? 2u : 0u) + (b1 ? 1u : 0u);
          } else {
            node = 2u * node + (b0 ? 1u : 0u);
          }
        }
    }
#ifdef TSIMK_WIDE_TRACE
    asm volatile("" :: "v"(node));
#endif
    WT_MARK(4);
    const uint32_t leaf = hit ? (node & ((1u << n_out) - 1u)) : 0u;
    if (hit) {
#pragma unroll
      for (int d = 0; d < WO32; ++d) o[d] |= l_lut[leaf * (uint32_t)WO32 + (uint32_t)d];
    }
    // ---- tabulated rows leave now; a missed row leaves with its direct bits only (the dense pass completes and rewrites the
    // words that hold component outputs), a heavy row is written by the generic pass
    // (whole rows, one 16-byte store each instead of one store per word the dense pass does not own: 0.57 -> 0.20e6 store
    // instructions per 8 x 10^6 shots, 46.6 -> 46.0 us.  The HBM-side write traffic stays 27 B per shot for 16: the word the
    // dense pass rewrites later leaves the L2 a second time as a 32-byte request; writing a missed row ONCE needs its direct
    // words in the queue - 24 KB of LDS the block does not have)
    if (A.merge) {  // a later component's pass: a tabulated row with sampled ones is completed in place; nothing else is touched here
      if (hit && leaf != 0u) merge_row((uint32_t)(uintptr_t)S->out, (uint32_t)((uint64_t)(uintptr_t)S->out >> 32), (uint32_t)(uintptr_t)S->out_compact,
                                       (uint32_t)((uint64_t)(uintptr_t)S->out_compact >> 32), row, o);
    } else if ((hit || miss) && !(TSIMK_WIDE_SKIP & 32)) store_row((uint32_t)(uintptr_t)S->out, (uint32_t)((uint64_t)(uintptr_t)S->out >> 32), (uint32_t)(uintptr_t)S->out_compact,
                                                            (uint32_t)((uint64_t)(uintptr_t)S->out_compact >> 32), row, o);
    // ---- missed rows -> the queue (position list + draws); heavy rows -> their ring
    {
      const unsigned long long mm = (TSIMK_WIDE_SKIP & 64) ? 0ull : __builtin_amdgcn_ballot_w64(miss);
      if (mm != 0ull) {
        if (miss) {
          const uint32_t slot = (qtail + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull))) & (QCAP - 1u);
          w_q[slot] = (st << 28) | row;
          w_q[QCAP + slot] = l0;
          w_q[2u * QCAP + slot] = l1;
          w_q[3u * QCAP + slot] = l2;
          if constexpr (P16) {
            w_q[4u * QCAP + slot] = l3;
            w_q[5u * QCAP + slot] = l4;
            w_q[6u * QCAP + slot] = l5;
          }
#pragma unroll
          for (int i = 0; i < 8; ++i)
            if ((uint32_t)i < n_out) w_q[(QD + (uint32_t)i) * QCAP + slot] = dr[i];
          uint32_t cwi = 0;
#pragma unroll
          for (int d = 0; d < WO32; ++d)
            if ((lutmask >> d) & 1u) {
              w_q[(QD + n_out + cwi) * QCAP + slot] = o[d];
              ++cwi;
            }
        }
        qtail += (uint32_t)__popcll(mm);
        n_missed += (uint32_t)__popcll(mm);
      }
      if (hm != 0ull) {
        if (heavy) w_ovf[(otail + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull))) & (QCAP - 1u)] = (st << 28) | row;
        otail += (uint32_t)__popcll(hm);
        n_heavy += (uint32_t)__popcll(hm);
      }
    }
    if (A.has_check && ch == 0u) check_pending = st + 1u;  // this batch's row 0 lives in this chunk (sampler.py:66-72)
    WT_MARK(5);
    staged = staged_next;
    c += tw;
    st = st_n;
    ch = ch_n;
  }
  WT_MARK(0);
  WT_FLUSH;
  // ---- launch-plan statistics: block 0's share, scaled to the grid (a sample: nothing but heuristics reads it)
  if (A.feedback && blockIdx.x == 0) {
    // (no static __shared__ here: it would sit in front of the dynamic segment and push the 16-byte table entries off their
    // alignment - 8 bytes of counters cost every ds_read_b128 of the kernel a split access, 120 -> 50 us per 10^6 shots)
    uint32_t *s_cnt = l_sel + 2 * TSIMK_WIDE_SELMAX;
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    if (lane == 0u) {
      atomicAdd(&s_cnt[0], n_heavy);
      atomicAdd(&s_cnt[1], n_missed);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned long long rows = (unsigned long long)Bu * (unsigned long long)A.n_steps;
      const unsigned long long scale = gridDim.x;
      // (words 4.. of the feedback block: the round-2 kernels keep words 0..2 for their list counts)
      A.feedback[4] = (uint32_t)min((unsigned long long)s_cnt[0] * scale, 0xFFFFFFFEull);
      A.feedback[5] = (uint32_t)min((unsigned long long)s_cnt[1] * scale, 0xFFFFFFFEull);
      A.feedback[6] = (uint32_t)min(rows, 0xFFFFFFFEull);
    }
  }
}

}  // namespace tsimk
