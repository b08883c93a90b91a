// tsim_sample_gen.hip - the launchers of k_sample_gen (tsim_gen.hip.h): LDS layout, eligibility, the fused groups of
// tsim_sample_steps_device.  Split from tsim_sample.hip in round 5 (tsim_sample_internal.hip.h).
#include "tsim_sample_internal.hip.h"
#include "tsim_gen.hip.h"

using namespace tsimk;
using namespace tsimhost;


// Any narrow program (k_sample_gen, tsim_gen.hip.h): up to TSIMK_GEN_MAX_STEPS batches as one grid of chip-resident blocks, the
// rows staged in LDS wave by wave; hard rows to each batch's lists, the group's hard-row grid behind it - the protocol of
// steps_group_fused, whose bookkeeping this shares.
struct GenLayout {
  int block = 0, nbuf = 1;
  size_t lds = 0;
  int l_wave = 0, wave_bytes = 0;
};
GenLayout gen_layout(const tsim_program *p, int WF32, int n_steps) {
  GenLayout L;
  if (!p->gr_off) return L;
  const uint32_t *h = &p->img[p->gr_off];
  auto up = [](size_t v, size_t a) { return (v + a - 1) / a * a; };
  (void)n_steps;
  size_t off = ((size_t)h[GR_LDS_WORDS] + 8 * (size_t)h[GR_NCOMP]) * 4;  // rank tables, pattern bases
  off = up(off, 16);
  const size_t buf = (size_t)64 * WF32 * 4;
  // two resident blocks of 16 waves per CU (the register budget of the kernel allows it) when the row buffers fit 80 KB
  // each, double-buffered if that still fits; wider rows: fewer waves per block
  for (int blk : {1024, 512, 256}) {
    for (int nbuf : {2, 1}) {
      const size_t tot = off + (size_t)(blk / 64) * nbuf * buf;
      if (tot <= (blk == 1024 ? 80u : 64u) * 1024) {
        L.block = blk;
        L.nbuf = nbuf;
        L.l_wave = (int)off;
        L.wave_bytes = (int)(nbuf * buf);
        L.lds = tot;
        return L;
      }
    }
  }
  return L;
}
// can a fused group of this program go to k_sample_gen?  (32-bit row offsets: batches below 2^28 rows, a shot range that does
// not cross a multiple of 2^32)
bool gen_applies(const tsim_program *p, int64_t B, int32_t num_f, int64_t shot_offset) {
  if (!(p->lw && !p->lw_wide && p->gr_off != 0 && p->knobs.gen > 0)) return false;
  if (B <= 0 || B >= (1ll << 28)) return false;
  if (((unsigned long long)shot_offset >> 32) != ((unsigned long long)(shot_offset + B - 1) >> 32)) return false;
  const int WF32 = 2 * std::max(1, (num_f + 63) / 64);
  if (WF32 < (int)p->img[p->gr_off + GR_WF32_MIN] || WF32 > 64) return false;
  return gen_layout(p, WF32, 1).block != 0;
}

static int gen_launch(tsim_program *p, const GenArgs &G, const GenLayout &L, long long grid, hipStream_t s) {
  const int wo32 = (int)p->img[p->gr_off + GR_WO32];
#define TSIM_LGEN(N)                                                                                                    \
  case N: {                                                                                                             \
    auto kfn = k_sample_gen<N>;                                                                                         \
    if (!(p->gen_attr_set & (1u << N))) {                                                                               \
      HIP_TRY(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));           \
      p->gen_attr_set |= 1u << N;                                                                                       \
    }                                                                                                                   \
    hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(L.block), L.lds, s, G);                                          \
  } break;
  switch (wo32) {
    TSIM_LGEN(2) TSIM_LGEN(4) TSIM_LGEN(6) TSIM_LGEN(8) TSIM_LGEN(10) TSIM_LGEN(12) TSIM_LGEN(14) TSIM_LGEN(16)
    default: return tsim_fail(TSIM_ESTATE, "gen record with %d output words", wo32);
  }
#undef TSIM_LGEN
  return 0;
}
// the subkeys of one batch as k_sample_gen wants them: the inline copy, or the chain of splits (sampler.py:399's key schedule)
static void gen_step_keys(const tsim_program *p, const tsim_program::Slot &sl, const SampleArgs &a, uint32_t k0, uint32_t k1, uint32_t *stkeys) {
  if (p->total_keys <= TSIMK_INLINE_KEYS) {
    memcpy(stkeys, a.inline_keys, sizeof(uint32_t) * 2 * (size_t)p->total_keys);
    return;
  }
  if (sl.host_keys.size() == 2 * (size_t)p->total_keys) {  // (fill_sample_args has just computed them)
    memcpy(stkeys, sl.host_keys.data(), sizeof(uint32_t) * 2 * (size_t)p->total_keys);
    return;
  }
  for (int i = 0; i < p->total_keys; ++i) {  // (the hard-row kernels read the k_keygen buffer; this pass wants the subkeys in its arguments)
    uint32_t a0 = 0u, a1 = 0u, b0 = 0u, b1 = 1u;
    threefry2x32(k0, k1, a0, a1);
    threefry2x32(k0, k1, b0, b1);
    stkeys[2 * i] = b0;
    stkeys[2 * i + 1] = b1;
    k0 = a0;
    k1 = a1;
  }
}
static void gen_common_args(const tsim_program *p, GenArgs &G, const GenLayout &L, int64_t B, int64_t shot_offset, int n, int WF, bool has_check,
                            long long list_cap, int n_lists) {
  G.img = p->d_img;
  G.tab = p->d_lw_tab;
  G.B = B;
  G.shot_offset = shot_offset;
  G.n_steps = n;
  G.total_keys = p->total_keys;
  G.chunks_per_step = (int)((B + 63) / 64);
  G.has_check = has_check ? 1 : 0;
  G.out_rb = (p->num_outputs + 7) / 8;
  G.WF32 = 2 * WF;
  G.lw_off = p->lw_off;
  G.gr_off = p->gr_off;
  G.list_cap = (int)list_cap;
  G.n_lists = n_lists;
  G.nbuf = L.nbuf;
  G.l_wave = L.l_wave;
  G.wave_bytes = L.wave_bytes;
}
static long long gen_grid(const tsim_program *p, const GenLayout &L, long long chunks) {
  const int wpb = L.block / 64;
  const int per_cu = std::max(1, std::min(2048 / L.block, (int)((160 * 1024) / (L.lds + 64))));
  return std::max(1ll, std::min((long long)p->n_cu * per_cu, (chunks + wpb - 1) / wpb));
}
// The serial API (tsim_sample_batch*, the seam of backend.sample_program) on a program whose tables only k_sample_gen reads
// (prefix trees, narrow_big): the same first pass as a group of ONE batch on the caller's stream, with the subkey as given.
// launch_sample goes on with the hard-row lists this leaves (list geometry as in steps_group_gen).
int gen_one(tsim_program *p, const tsim_program::Slot &sl, const SampleArgs &a, int64_t B, int32_t num_f, uint32_t key_hi, uint32_t key_lo, int64_t shot_offset,
            uint32_t *hard_index, uint32_t *ctl, uint32_t *ctl_next, int n_lists, bool has_check, long long *list_cap_out, hipStream_t s) {
  const int WF = std::max(1, (num_f + 63) / 64);
  const GenLayout L = gen_layout(p, 2 * WF, 1);
  if (!L.block) return tsim_fail(TSIM_ESTATE, "k_sample_gen does not fit");
  if (p->total_keys > TSIMK_GEN_KEYS) return tsim_fail(TSIM_ESTATE, "k_sample_gen: %d compiled outputs", p->total_keys);
  const long long bps = (B + 1023) / 1024;
  const long long list_cap = (bps + n_lists - 1) / n_lists * 1024;
  if (list_cap > 0x7FFFFFFFll) return tsim_fail(TSIM_ENOTSUP, "batch too large for the row lists");
  GenArgs G{};
  gen_common_args(p, G, L, B, shot_offset, 1, WF, has_check, list_cap, n_lists);
  GenStep &st = G.step[0];
  st.f = a.f;
  st.out = a.out;
  st.out_compact = a.out_compact;
  st.hard_index = hard_index;
  st.ctl = ctl;
  st.ctl_next = ctl_next;
  gen_step_keys(p, sl, a, key_hi, key_lo, G.keys);
  ++p->path_count[TP_GEN];
  if (int r = gen_launch(p, G, L, gen_grid(p, L, G.chunks_per_step), s)) return r;
  HIP_TRY(hipGetLastError());
  *list_cap_out = list_cap;
  return 0;
}

int steps_group_gen(tsim_program *p, int n, const uint64_t *const *d_f, int64_t B, int32_t num_f, uint32_t key[2],
                           int64_t shot_offset, void *const *d_out, float *const *d_dev, uint32_t flags, const LaunchPlan &plan) {
  const bool packed = (flags & TSIM_PIPE_OUT_BIT_PACKED) != 0;
  if (!p->deferred.empty())
    if (int r = flush_batch(p)) return r;
  const int lanes = p->knobs.fused_lanes > 0 ? p->knobs.fused_lanes : ((long long)n * B <= (1ll << 21) ? 3 : 2);
  hipStream_t s = p->slots[1 + (int)(p->steps_groups++ % (unsigned long long)lanes)].side;
  if (!(flags & TSIM_PIPE_INPUTS_READY) && p->stream != s) {
    if (!p->sync_ev) HIP_TRY(hipEventCreateWithFlags(&p->sync_ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(p->sync_ev, p->stream));
    HIP_TRY(hipStreamWaitEvent(s, p->sync_ev, 0));
  }
  const int WF = std::max(1, (num_f + 63) / 64);
  const GenLayout L = gen_layout(p, 2 * WF, n);
  if (!L.block) return tsim_fail(TSIM_ESTATE, "k_sample_gen does not fit");
  const long long bps = (B + 1023) / 1024;  // hard-row lists by row blocks of 1024 rows, whatever the kernel's block
  const int n_lists = plan.lists;
  const long long list_cap = (bps + n_lists - 1) / n_lists * 1024;
  if (list_cap > 0x7FFFFFFFll) return tsim_fail(TSIM_ENOTSUP, "batch too large for the row lists");
  const bool has_check = shot_offset == 0;
  GenArgs G{};
  if ((long long)n * p->total_keys > TSIMK_GEN_KEYS) return tsim_fail(TSIM_ESTATE, "k_sample_gen: %d batches of %d compiled outputs in one launch", n, p->total_keys);
  gen_common_args(p, G, L, B, shot_offset, n, WF, has_check, list_cap, n_lists);
  p->last_lists = n_lists;
  int slots[TSIMK_GEN_MAX_STEPS];
  for (int j = 0; j < n; ++j)
    if (p->slots[1 + (int)((p->steps_slot + (unsigned long long)j) % (unsigned long long)TSIM_PIPELINE_SLOTS)].deferred) {
      if (int r = tsim_flush_hard(p)) return r;
      break;
    }
  for (int j = 0; j < n; ++j) {
    const int sidx = 1 + (int)(p->steps_slot++ % (unsigned long long)TSIM_PIPELINE_SLOTS);
    slots[j] = sidx;
    tsim_program::Slot &sl = p->slots[sidx];
    if (sl.deferred) return tsim_fail(TSIM_ESTATE, "pipeline slot %d still holds a parked launch", sidx - 1);
    if ((size_t)list_cap * n_lists * 4 > sl.hard_sz) return tsim_fail(TSIM_ESTATE, "hard-row list too small");
    if (int r = slot_order_after_previous(p, sl, s)) return r;
    uint32_t o[4];
    tsim_key_split(key[0], key[1], o);  // key, subkey = split(key)  (sampler.py:399)
    key[0] = o[0];
    key[1] = o[1];
    SampleArgs &a = sl.ctx;
    a = SampleArgs{};
    if (int r = fill_sample_args(p, sl, a, d_f[j], B, num_f, o[2], o[3], shot_offset, (uint64_t *)d_out[j], d_dev ? d_dev[j] : nullptr, s,
                                 sidx, packed))
      return r;
    GenStep &st = G.step[j];
    st.f = d_f[j];
    st.out = a.out;
    st.out_compact = a.out_compact;
    st.hard_index = (uint32_t *)sl.hard;
    uint32_t *ctl = sl.ctl + sl.parity * (TSIMK_LW_LISTS + 1) * 32;
    st.ctl = ctl;
    st.ctl_next = sl.ctl + (sl.parity ^ 1) * (TSIMK_LW_LISTS + 1) * 32;
    sl.parity ^= 1;
    gen_step_keys(p, sl, a, o[2], o[3], G.keys + 2 * (size_t)j * (size_t)p->total_keys);
    a.row_index = st.hard_index;
    a.row_count = ctl;
    a.row_lists = n_lists;
    a.row_list_cap = (int)list_cap;
    a.check_row = has_check ? ctl + 32 * TSIMK_LW_LISTS : nullptr;
    a.no_check = has_check ? 0 : 1;
    a.row_slot_begin = 0;
    a.row_slot_end = 0;
  }
  const long long chunks = (long long)G.chunks_per_step * n;
  const long long grid = gen_grid(p, L, chunks);
  TSIM_MARK("args");
  if (int r = tsim_tables_slice(p, s)) return r;  // (a table build in the background: its next slice goes first)
  const bool prof = p->profiling && (p->prof_counter++ % p->prof_every == 0);
  if (prof) { if (int r = prof_event(p, s, PROF_BEGIN)) return r; }
  ++p->path_count[TP_GEN];
  if (int r = gen_launch(p, G, L, grid, s)) return r;
  HIP_TRY(hipGetLastError());
  TSIM_MARK("launch");
  if (prof) {
    if (int r = prof_event(p, s, PROF_PASS1)) return r;
    p->prof_steps += n;
  }
  hard_geometry(p, WF, (p->num_outputs + 63) / 64);
  for (int j = 0; j < n; ++j) {
    tsim_program::Slot &sl = p->slots[slots[j]];
    sl.ctx_check = has_check;
    sl.deferred = true;
    sl.pending = true;
    sl.p1_stream = s;
    sl.partial = false;
    p->deferred.push_back(slots[j]);
  }
  p->stat_begins += (unsigned long long)n;
  p->stat_deferred += (unsigned long long)n;
  ++p->stat_fused;
  if ((long long)n * B <= p->knobs.hard_inline_rows) p->flush_inline = s;
  const int rf = flush_chunks(p);
  TSIM_MARK("hard");
  return rf;
}
