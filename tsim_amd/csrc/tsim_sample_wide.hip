// tsim_sample_wide.hip - the launchers of k_sample_wide (tsim_wide.hip.h): LDS layout, eligibility, one pass per component,
// the fused groups of tsim_sample_steps_device.  Split from tsim_sample.hip in round 5 (tsim_sample_internal.hip.h).
#include "tsim_sample_internal.hip.h"
#include "tsim_wide.hip.h"

using namespace tsimk;
using namespace tsimhost;


// ---------------------------------------------------------------------------
// One wide component (k_sample_wide, tsim_wide.hip.h): up to TSIMK_LWM_MAX_STEPS batches as ONE grid of chip-resident
// blocks - tables, sparse-column evaluation of the rows they miss, heavy rows and the normalisation check all inside.
// ---------------------------------------------------------------------------
WideLayout wide_layout(const tsim_program *p, int WF32, size_t ci) {
  WideLayout L;
  const HostComponent &c = p->comps[ci];
  const uint32_t *wr = &p->img[p->wr_offs[ci]];
  const int wo32 = (int)wr[WR_WO32];
  auto up = [](size_t v, size_t a) { return (v + a - 1) / a * a; };
  // the shared column table (one entry holds every graph's parity bits) when the packer made one and the term tables fit
  // beside it; else one table per graph
  // (third choice, compact = -1: no column table in LDS at all - components with many graphs, round 5)
  // - up to kWideGlobGraphs graphs: a dense pass walks the graphs one after the other, ~2.5 us each from the L2 (F60 class, 140
  // graphs: 300-400 us per pass, 77 us per 10^6 shots against 59 on the round-2 kernels - profiles/r05/wide_glob.txt)
  const int min_mode = (int)wr[WR_GTOT] <= kWideGlobGraphs ? -1 : 0;
  for (int compact = wr[WR_CCOL] != 0u ? 1 : 0; compact >= min_mode && !L.block; --compact) {
    size_t off = compact > 0 ? (size_t)(c.F + 33) * 16 : compact == 0 ? up(wr[WR_COLBYTES], 16) : 0;
    L.l_rank = (int)off;  off += (size_t)4 * (c.F + 1) * 4;
    off = up(off, 16);
    L.l_lut = (int)off;   off += ((size_t)wo32 << c.n_out) * 4;
    off = up(off, 16);
    L.l_runs = (int)off;  off += (size_t)(2 * TSIMK_WIDE_MAX_RUNS + 48) * 4;
    L.l_sel = (int)off;   off += (2 * TSIMK_WIDE_SELMAX + 4) * 4;  // + the two statistics counters
    L.l_ptrs = (int)off;  off += (size_t)TSIMK_LWM_MAX_STEPS * 4 * 4;
    L.l_keys = (int)off;  off += (size_t)TSIMK_LWM_MAX_STEPS * 2 * TSIMK_LWM_KEYS * 4;
    off = up(off, 16);
    const size_t fixed_end = off;
    size_t w = (size_t)64 * WF32 * 4;
    w = up(w, 16);
    const size_t np = c.F > 255 ? 6 : 3;  // position words of a queue entry (k_sample_wide<.., P16>)
    L.w_q = (int)w;       w += (size_t)(1 + np + c.n_out + __builtin_popcount(wr[WR_LUTMASK])) * TSIMK_WIDE_QCAP * 4;
    L.w_ovf = (int)w;     w += (size_t)TSIMK_WIDE_QCAP * 4;
    L.wave_bytes = (int)up(w, 16);
    // the levels' term tables join the column tables in LDS when 16 waves still fit next to them (C5: 3 KB)
    // (+ the level table and the graph records: 16 bytes per level, 64 per graph)
    const size_t tt_only = up(wr[WR_TTBYTES], 16);
    const size_t tt_bytes = tt_only + 16 * (size_t)(c.n_out + 1) + 64 * (size_t)wr[WR_GTOT];
    for (int with_tt = compact < 0 ? 0 : 1; with_tt >= (compact > 0 ? 1 : 0) && !L.block; --with_tt) {
      if (with_tt && tt_only == 0) continue;
      off = fixed_end + (with_tt ? tt_bytes : 0);
      for (int blk : {1024, 512, 256}) {  // 16 waves per CU when everything fits beside the tables, fewer otherwise
        if (with_tt && blk != 1024) break;
        if (compact == 0 && blk < 512) break;  // (rather the tables in the L2 and 16 waves than 4 waves beside them)
        const size_t tot = off + (size_t)(blk / 64) * L.wave_bytes + 64;  // + the kernel's static words
        if (tot <= 160 * 1024) {
          L.block = blk;
          L.compact = compact > 0 ? 1 : 0;
          L.glob = compact < 0 ? 1 : 0;
          L.l_tt = with_tt ? (int)fixed_end : -1;
          L.l_lvl = (int)(fixed_end + tt_only);
          L.l_grec = L.l_lvl + 16 * (c.n_out + 1);
          L.l_wave = (int)off;
          L.lds = off + (size_t)(blk / 64) * L.wave_bytes;
          break;
        }
      }
    }
  }
  return L;
}

// can this launch go to k_sample_wide?  (32-bit offsets: batches below 2^28 rows, tables below 4 GB, a shot range that does
// not cross a multiple of 2^32; bit_packed rows are written and merged as dwords)
bool wide_applies(const tsim_program *p, int64_t B, int32_t num_f, int64_t shot_offset) {
  if (!(p->lw && p->lw_wide && p->wr_off != 0 && p->knobs.wide_fused)) return false;
  // (up to TSIMK_INLINE_KEYS compiled outputs: each pass carries its own component's subkeys - WR_KEYSUB - when they are more than TSIMK_LWM_KEYS)
  if (p->total_keys <= 0 || p->total_keys > TSIMK_INLINE_KEYS || B <= 0 || B >= (1ll << 28)) return false;
  for (size_t ci = 0; ci < p->comps.size() && ci < p->lw_npat.size(); ++ci)  // 32-bit byte offsets inside a component's table
    if (((p->lw_npat[ci] << p->comps[ci].n_out) * 4) >= (1ll << 32)) return false;
  if (((unsigned long long)shot_offset >> 32) != ((unsigned long long)(shot_offset + B - 1) >> 32)) return false;
  const int WF = std::max(1, (num_f + 63) / 64);
  if (WF > 32 || p->wr_offs.size() != p->comps.size()) return false;  // (f rows of up to 2048 bits; wide_layout says whether they fit)
  for (size_t ci = 0; ci < p->comps.size(); ++ci) {
    const WideLayout L = wide_layout(p, 2 * WF, ci);
    if (L.block == 0 || (L.glob && p->comps[ci].F > 255)) return false;
  }
  return true;
}
bool wide_buffers_ok(const tsim_program *p, const SampleArgs &a) {
  if (!a.out_compact) return true;
  return (a.out_rb + 3) / 4 <= 2 * ((p->num_outputs + 63) / 64);  // (any row size, any alignment: tsim_wide.hip.h oc_put)
}

// `args[j]`: the SampleArgs of batch j as fill_sample_args made them (buffers, inline keys)
int launch_wide(tsim_program *p, int n, const SampleArgs *const *args, int64_t B, int32_t num_f, int64_t shot_offset, hipStream_t s) {
  const int WF32 = 2 * std::max(1, (num_f + 63) / 64);
  if (p->wr_offs.size() != p->comps.size()) return tsim_fail(TSIM_ESTATE, "wide records missing");
  // the statistics the launch plan follows (missed / heavy rows) come from the pass of the component with most parameters
  size_t fb_ci = 0;
  for (size_t ci = 1; ci < p->comps.size(); ++ci)
    if (p->comps[ci].F > p->comps[fb_ci].F) fb_ci = ci;
  for (size_t ci = 0; ci < p->comps.size(); ++ci) {  // one pass per component, in stream order (tsim_wide.hip.h: WR_MERGE)
  const WideLayout L = wide_layout(p, WF32, ci);
  if (!L.block) return tsim_fail(TSIM_ESTATE, "wide kernel does not fit");
  const long long comp_tab_bytes = (p->lw_npat[ci] << p->comps[ci].n_out) * 4;
  if (comp_tab_bytes >= (1ll << 32)) return tsim_fail(TSIM_ESTATE, "pattern table of %lld bytes: k_sample_wide addresses a component's table with 32-bit offsets", comp_tab_bytes);
  if (tsim_debug("host")) {
    static bool said = false;
    if (!said) fprintf(stderr, "[tsim] k_sample_wide: block %d, LDS %zu bytes (per wave %d), shared column table %d, term tables in LDS %d, column tables in the image %d\n", L.block, L.lds, L.wave_bytes, L.compact, L.l_tt >= 0 ? 1 : 0, L.glob);
    said = true;
  }
  WideArgs W{};
  W.img = p->d_img;
  W.tab = p->d_lw_tab + p->img[(size_t)p->lw_off + ci * LW_WORDS + LW_TAB];
  W.B = B;
  W.shot_offset = shot_offset;
  W.n_steps = n;
  W.chunks_per_step = (int)((B + 63) / 64);
  W.has_check = shot_offset == 0 ? 1 : 0;
  W.out_rb = (p->num_outputs + 7) / 8;
  W.WF32 = WF32;
  W.lw_off = p->lw_off + (int)ci * LW_WORDS;
  W.comp4_off = p->comp4_off + (int)ci * C4_WORDS;
  W.wr_off = p->wr_offs[ci];
  W.binom_off = p->lw_binom_off;
  const bool p16 = p->comps[ci].F > 255;
  if (p16 && L.glob) return tsim_fail(TSIM_ESTATE, "no instantiation of k_sample_wide for 16-bit positions with the column tables in the L2");
  W.tab_bytes = (uint32_t)comp_tab_bytes;
  W.feedback = ci == fb_ci ? p->d_feedback : nullptr;
  W.merge = ci > 0 ? 1 : 0;
  W.dev_index = (int)ci;
  W.compact = L.compact;
  W.l_rank = L.l_rank; W.l_lut = L.l_lut; W.l_runs = L.l_runs; W.l_sel = L.l_sel; W.l_ptrs = L.l_ptrs; W.l_keys = L.l_keys;
  W.l_tt = L.l_tt; W.l_lvl = L.l_lvl; W.l_grec = L.l_grec; W.l_wave = L.l_wave; W.wave_bytes = L.wave_bytes; W.w_q = L.w_q; W.w_ovf = L.w_ovf;
  for (int j = 0; j < n; ++j) {
    const SampleArgs &a = *args[j];
    WideStep &st = W.step[j];
    st.f = a.f;
    st.out = a.out;
    st.out_compact = a.out_compact;
    st.norm_dev = a.norm_dev;
    const int ksub = (int)p->img[(size_t)p->wr_offs[ci] + WR_KEYSUB];
    memcpy(st.keys, a.inline_keys + 2 * ksub, sizeof(uint32_t) * 2 * (size_t)std::min(TSIMK_LWM_KEYS, p->total_keys - ksub));
  }
  const long long chunks = (long long)W.chunks_per_step * n;
  const int wpb = L.block / 64;
  // chip-resident blocks: as many as the LDS lets a CU hold (one of 1024 threads when the program is C5-sized)
  const int per_cu = std::max(1, (int)((160 * 1024) / (L.lds + 64)));
  const long long grid = std::max(1ll, std::min((long long)p->n_cu * per_cu, (chunks + wpb - 1) / wpb));
  const int wo32 = (int)p->img[p->wr_offs[ci] + WR_WO32];
#define TSIM_LWIDE(N)                                                                                                   \
  case N: {                                                                                                             \
    auto kfn = p16 ? k_sample_wide<N, TSIMK_WIDE_K, false, true>                                                        \
                   : L.glob ? k_sample_wide<N, TSIMK_WIDE_K, true> : k_sample_wide<N, TSIMK_WIDE_K, false>;             \
    const unsigned abit = 1u << (N + (L.glob ? 16 : 0) + (p16 ? 8 : 0));                                               \
    if (!(p->wide_attr_set & abit)) { /* per handle: the attribute is per device */                        \
      HIP_TRY(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));     \
      p->wide_attr_set |= abit;                                                                                         \
    }                                                                                                                   \
    ++p->path_count[TP_WIDE];                                                                                           \
    hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(L.block), L.lds, s, W);                                          \
  } break;
  switch (wo32) {
    TSIM_LWIDE(2) TSIM_LWIDE(4) TSIM_LWIDE(6) TSIM_LWIDE(8)
    default: return tsim_fail(TSIM_ESTATE, "wide record with %d output words", wo32);
  }
#undef TSIM_LWIDE
  }
  HIP_TRY(hipGetLastError());
  return 0;
}



// One wide component: up to `n` batches as one k_sample_wide grid on a first-pass lane, lanes alternating between groups.
// Nothing is left behind a group - no hard-row lists, no second kernel: the group's slots are done when the grid is.
int steps_group_wide(tsim_program *p, int n, const uint64_t *const *d_f, int64_t B, int32_t num_f, uint32_t key[2],
                            int64_t shot_offset, void *const *d_out, float *const *d_dev, uint32_t flags) {
  const bool packed = (flags & TSIM_PIPE_OUT_BIT_PACKED) != 0;
  if (!p->deferred.empty())
    if (int r = tsim_flush_hard(p)) return r;
  hipStream_t s = p->slots[1 + (int)(p->steps_groups++ & 1ull)].side;
  if (!(flags & TSIM_PIPE_INPUTS_READY) && p->stream != s) {
    if (!p->sync_ev) HIP_TRY(hipEventCreateWithFlags(&p->sync_ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(p->sync_ev, p->stream));
    HIP_TRY(hipStreamWaitEvent(s, p->sync_ev, 0));
  }
  const SampleArgs *args[TSIMK_LWM_MAX_STEPS];
  int first = 0;
  for (int j = 0; j < n; ++j) {
    const int sidx = 1 + (int)(p->steps_slot++ % (unsigned long long)TSIM_PIPELINE_SLOTS);
    if (j == 0) first = sidx;
    if (int r = slot_prepare(p, sidx, 0)) return r;  // (k_sample_wide leaves no row lists: counters, keys and events only)
    tsim_program::Slot &sl = p->slots[sidx];
    if (sl.deferred) return tsim_fail(TSIM_ESTATE, "pipeline slot %d still holds a parked launch", sidx - 1);
    if (int r = slot_order_after_previous(p, sl, s)) return r;
    uint32_t o[4];
    tsim_key_split(key[0], key[1], o);  // key, subkey = split(key)  (sampler.py:399)
    key[0] = o[0];
    key[1] = o[1];
    SampleArgs &a = sl.ctx;
    a = SampleArgs{};
    if (int r = fill_sample_args(p, sl, a, d_f[j], B, num_f, o[2], o[3], shot_offset, (uint64_t *)d_out[j], d_dev ? d_dev[j] : nullptr, s, sidx, packed))
      return r;
    if (!wide_buffers_ok(p, a)) return tsim_fail(TSIM_ESTATE, "bit_packed rows wider than the wide record's output words");
    args[j] = &a;
  }
  TSIM_MARK("args");
  const bool prof = p->profiling && (p->prof_counter++ % p->prof_every == 0);
  if (prof) { if (int r = prof_event(p, s, PROF_BEGIN)) return r; }
  if (int r = tsim_tables_slice(p, s)) return r;  // (a table build in the background: its next slice goes first)
  if (int r = launch_wide(p, n, args, B, num_f, shot_offset, s)) return r;
  TSIM_MARK("launch");
  if (prof) {
    if (int r = prof_event(p, s, PROF_PASS1)) return r;
    p->prof_steps += n;
  }
  hipEvent_t ev = p->slots[first].ev2;
  HIP_TRY(hipEventRecord(ev, s));
  for (int j = 0; j < n; ++j) {
    tsim_program::Slot &sl = p->slots[1 + (int)((p->steps_slot - (unsigned long long)n + (unsigned long long)j) % (unsigned long long)TSIM_PIPELINE_SLOTS)];
    sl.pending = true;
    sl.last_done = s;
    sl.done_ev = ev;
    sl.batch_seq = 0;
  }
  p->stat_begins += (unsigned long long)n;
  ++p->stat_fused;
  return 0;
}

