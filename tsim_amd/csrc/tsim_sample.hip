// tsim_sample.hip - the hot path of the C ABI: launch planner, pipeline scheduler and the sampling
// kernels that need the chunk-table layout (k_sample_lw, k_sample4, k_sample4h, k_sample4h_multi),
// device-side post-selection and the HIP-event profiling of the launches.
#include "tsim_internal.hip.h"

#include <chrono>
#include "tsim_kernel4h.hip.h"
#include "tsim_filter.hip.h"
#include "tsim_lw_pass.hip.h"
#include "tsim_lw_multi.hip.h"
#include "tsim_lw_fast.hip.h"
#include "tsim_kernel4w.hip.h"
#include "tsim_kernel_hw.hip.h"
#include "tsim_direct.hip.h"
#include "tsim_lw_fastm.hip.h"
#include "tsim_sample_internal.hip.h"
#include "tsim_noise_fused.hip.h"

using namespace tsimk;
using namespace tsimhost;

thread_local TsimNoiseRequest *g_noise_req = nullptr;
HostMarks *g_marks = nullptr;


int prof_event(tsim_program *p, hipStream_t s, int tag) {
  if (p->ev_used == p->ev_pool.size()) {
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    p->ev_pool.push_back(e);
    p->ev_tag.push_back(0);
  }
  p->ev_tag[p->ev_used] = tag;
  HIP_TRY(hipEventRecord(p->ev_pool[p->ev_used++], s));
  return 0;
}

static int prof_drain(tsim_program *p) {
  size_t begin = 0;
  for (size_t i = 0; i < p->ev_used; ++i) {
    if (p->ev_tag[i] == PROF_BEGIN) { begin = i; continue; }
    HIP_TRY(hipEventSynchronize(p->ev_pool[i]));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, p->ev_pool[i - 1], p->ev_pool[i]));
    p->prof_stage_ms[p->ev_tag[i]] += ms;
    if (i + 1 == p->ev_used || p->ev_tag[i + 1] == PROF_BEGIN) {  // last event of this launch
      HIP_TRY(hipEventElapsedTime(&ms, p->ev_pool[begin], p->ev_pool[i]));
      p->prof_ms += ms;
      p->prof_launches += 1;
    }
  }
  p->ev_used = 0;
  return 0;
}
// per-slot resources, created on first use
// `need_stream`: the slot's own stream too (the lanes - slots 1..4 - always; the others only when a launch runs on the slot's
// own stream, tsim_sample_batch_device_begin outside the deferred plan: a stream costs ~3 ms to create, 32 of them 100 ms of
// the first pipelined call of every handle - scripts/microbench/hip_setup_cost.hip)
int slot_prepare(tsim_program *p, int slot, size_t hard_bytes, bool need_stream) {
  tsim_program::Slot &sl = p->slots[slot];
  if ((p->lw || p->v4w) && !sl.ctl) {
    // two counter sets per slot, used alternately: pass 1 of a launch resets the set of the slot's next one.  The sets of ALL
    // slots (and, for wide programs, the sparse-column pass's own) are one allocation, initialised by one copy - 32 slots x
    // (2 hipMalloc + 6 hipMemset on the null stream) were 1-4 ms of a handle's first pipelined call
    const size_t set_words = (TSIMK_LW_LISTS + 1) * 32, slot_words = 2 * set_words, kinds = p->lw_wide ? 2 : 1;
    if (!p->ctl_block) {
      const size_t words = (size_t)(1 + TSIM_PIPELINE_SLOTS) * kinds * slot_words;
      std::vector<uint32_t> init(words, 0u);
      for (size_t q = 0; q < words / set_words; ++q) init[q * set_words + TSIMK_LW_LISTS * 32] = 0xFFFFFFFFu;  // "no check row"
      HIP_TRY(hipMalloc((void **)&p->ctl_block, words * 4));
      HIP_TRY(hipMemcpy(p->ctl_block, init.data(), words * 4, hipMemcpyHostToDevice));
    }
    sl.ctl = p->ctl_block + (size_t)slot * kinds * slot_words;
    if (p->lw_wide) sl.ctl2 = sl.ctl + slot_words;
  }
  if (p->total_keys > TSIMK_INLINE_KEYS && !sl.keys) HIP_TRY(hipMalloc((void **)&sl.keys, (size_t)p->total_keys * 8));
  if (slot > 0 && !sl.ev1) {
    HIP_TRY(hipEventCreateWithFlags(&sl.ev1, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&sl.ev2, hipEventDisableTiming));
  }
  if (slot > 0 && !sl.side_ready && (slot <= 4 || need_stream)) {
    sl.side_ready = true;
    // default priority on purpose: a low- (or high-) priority lane lands on a different class of
    // hardware queue and tripled the step time (134 us vs 43 us, measured)
    if (slot == 1) {
      // The handle's own stream doubles as the first lane: HIP gave the lanes it created only two distinct
      // hardware queues (kernel trace: three created streams -> queues 3, 4, 4), the handle's stream sits on
      // a third one.  Three truly concurrent lanes: 36 us per step instead of 42.
      sl.side = p->stream;
      sl.side_borrowed = true;
    } else {
      if (int r = tsim_stream_acquire(p->device, p->stream_idx, &sl.side)) return r;  // (pooled: hardware queues the handle does not use yet first)
    }
  }
  if (sl.hard_sz < hard_bytes) {
    if (sl.hard) {
      HIP_TRY(hipDeviceSynchronize());
      HIP_TRY(hipFree(sl.hard));
      sl.hard = nullptr;
      sl.hard_sz = 0;
    }
    hipError_t e = hipMalloc(&sl.hard, hard_bytes);
    if (e != hipSuccess) return tsim_fail(TSIM_ENOMEM, "hipMalloc(%zu) failed: %s", hard_bytes, hipGetErrorString(e));
    sl.hard_sz = hard_bytes;
  }
  if (p->lw_wide && sl.hard2_sz < hard_bytes) {
    if (sl.hard2) {
      HIP_TRY(hipDeviceSynchronize());
      HIP_TRY(hipFree(sl.hard2));
      sl.hard2 = nullptr;
      sl.hard2_sz = 0;
    }
    hipError_t e = hipMalloc(&sl.hard2, hard_bytes);
    if (e != hipSuccess) return tsim_fail(TSIM_ENOMEM, "hipMalloc(%zu) failed: %s", hard_bytes, hipGetErrorString(e));
    sl.hard2_sz = hard_bytes;
  }
  return 0;
}

// Launch plan from the feedback of earlier launches (results do not depend on it):
//  * most rows hard (dense error patterns): the pattern pass is wasted work - run the full kernel
//    on every row for the next 15 launches, then probe again with one two-pass launch;
//  * hard-row lists short: k_sample4h walks them alone, no overflow launch of k_sample4 - and a
//    pipelined launch may leave its hard rows to a later batch (flush_hard) instead of making its
//    lane wait for them.

// `rows`: the rows the launches made under this plan will carry (the deepening rule counts them)
static LaunchPlan make_plan(tsim_program *p, bool has_row_index, bool pipelined, unsigned long long rows) {
  LaunchPlan pl;
  pl.use_tables = p->lw;
  // deeper tables built in the background (tsim_tables_extend_begin) are done: swap them in - the lanes drain, ~0.1 ms
  if (p->ext_pending && tsim_tables_extend_poll(p, false) == 1) {
    for (int i = 0; i < 8 && p->h_feedback; ++i) p->h_feedback[i] = 0xFFFFFFFFu;  // the old counts describe the old tables
    p->lw_direct_left = 0;
    p->deep_rows = 0;  // (rows counted against the old depth - the shallow start's - say nothing about the new one)
    p->lw_dense_launches = 0;
  }
  if (p->lw && p->lw_wide && p->wr_off != 0 && p->knobs.wide_fused && !has_row_index) {
    // One wide component (k_sample_wide): the kernel serves every row itself - nothing here decides coverage.  Two things
    // follow its statistics (block 0's share of the last launch): a deeper table when many rows miss the current one
    // (C5 at depth 3: 57 % -> depth 4, 2.1 GB built once: 37 %), and the round-2 path (sparse-column kernel with K = 10
    // on every row + row kernel) for 15 launches when a quarter of the rows carry more set bits than a dense pass takes.
    if (p->h_feedback && p->knobs.adaptive) {
      const uint32_t heavy = p->h_feedback[4], missed = p->h_feedback[5], fb_rows = p->h_feedback[6];
      const bool known = fb_rows != 0xFFFFFFFFu && fb_rows >= 4096u;
      if (p->lw_direct_left > 0) {
        --p->lw_direct_left;
        pl.use_tables = false;
      } else if (known && (double)heavy > 0.25 * (double)fb_rows) {
        p->lw_direct_left = 15;
        for (int i = 4; i < 8; ++i) p->h_feedback[i] = 0xFFFFFFFFu;
        pl.use_tables = false;
      } else if (known && p->lw_cap_now < p->lw_cap_max && p->knobs.deep_tables >= 0) {
        // The weight-4 table of a 200-bit component is 2.1 GB and 24 ms of build; it moves C5 from 46.8 to 42.1 us per 10^6
        // shots (profiles/r04/shapes.txt) - 5 ms back per 10^9 shots.  So, as for the narrow programs: at once on request
        // (TSIM_AMD_DEEP_TABLES=1), otherwise only for a handle that has launched deep_after rows in this state.
        if ((double)missed > 0.2 * (double)fb_rows && p->knobs.deep_tables == 0) {
          p->deep_rows += rows;
          if (p->deep_rows < tsim_tables_deep_after(p)) return pl;
        }
        p->lw_dense_launches = (double)missed > 0.2 * (double)fb_rows ? p->lw_dense_launches + 1 : 0;
        if (p->lw_dense_launches >= 3) {
          p->lw_dense_launches = 0;
          if (p->knobs.deep_tables == 1) {  // asked for by name: now, with the one stall that costs
            if (tsim_synchronize(p) == TSIM_OK && tsim_tables_extend(p) == TSIM_OK)
              for (int i = 4; i < 8; ++i) p->h_feedback[i] = 0xFFFFFFFFu;
          } else {
            (void)tsim_tables_extend_begin(p, p->lw_cap_max);  // in the background, slice by slice; make_plan's poll puts them in place
          }
        }
      }
    }
    return pl;
  }
  if (p->lw && p->h_feedback && p->knobs.adaptive) {
    uint32_t fb_sum = p->h_feedback[0], fb_max = p->h_feedback[1], fb_rows = p->h_feedback[2];
    // (counts of a tiny launch - the one-row reference sample, sampler.py:263-276 - say nothing about the batches to come:
    // one hard row of one was read as "dense" and cost the next fifteen launches their tables)
    bool known = fb_rows != 0xFFFFFFFFu && fb_rows >= 32u && fb_sum != 0xFFFFFFFFu;
    // More than 1 % of the rows hard, three launches in a row, and deeper tables are allowed: build them now
    // (once; every lane is drained first because the records in the device image are rewritten).
    if (known && p->lw_cap_now < p->lw_cap_max && !has_row_index) {
      // ... or so many that a batch of launches cannot use the block-per-row kernel (hw_eligible: more than
      // hard_wave_rows rows per eight launches - C3 at weight 5: 340 per 10^6 shots): one level deeper usually brings
      // them under it (C3: 45), and the per-shot hard-row grid (95 us per group for C3) leaves the lanes
      // Building a 1-2 GB table takes 20-200 ms (scripts/table_build_time.py) and gains 3-11 us per 10^6 shots: it pays
      // after some 10^10 shots, so by default only a handle that has seen knobs.deep_after rows in this state deepens
      // (TSIM_AMD_DEEP_TABLES=1: at once, =-1: never)
      bool too_many_for_hw = p->knobs.deep_tables >= 0 && p->knobs.hard_wave &&
                             (unsigned long long)fb_sum * 8ull > (unsigned long long)p->knobs.hard_wave_rows && fb_rows >= 65536u;
      if (too_many_for_hw && p->knobs.deep_tables == 0) {
        p->deep_rows += rows;
        too_many_for_hw = p->deep_rows >= tsim_tables_deep_after(p);
      }
      // (not when MOST rows are hard: no table depth helps a dense phase, the full kernel takes it - below)
      p->lw_dense_launches = (((double)fb_sum > 0.01 * (double)fb_rows && (double)fb_sum <= 0.5 * (double)fb_rows) || too_many_for_hw) ? p->lw_dense_launches + 1 : 0;
      if (p->lw_dense_launches >= 3) {
        p->lw_dense_launches = 0;
        if (p->knobs.deep_tables == 1) {  // asked for by name: now, with the one stall that costs
          if (tsim_synchronize(p) == TSIM_OK && tsim_tables_extend(p) == TSIM_OK) {
            for (int i = 0; i < 3; ++i) p->h_feedback[i] = 0xFFFFFFFFu;  // the old counts describe the old tables
            p->lw_direct_left = 0;
            known = false;
          }
        } else {
          (void)tsim_tables_extend_begin(p, p->lw_cap_max);  // in the background, slice by slice; the poll at the top puts them in place
        }
      }
    }
    // Many hard rows: they are throughput work for the full kernel, not latency work for the 8-waves-per-group one
    pl.hard_kernel = !(known && fb_sum > 16384u);
    bool dense = false;
    if (p->lw_direct_left > 0 && known && (double)fb_sum <= 0.5 * (double)fb_rows) {
      p->lw_direct_left = 0;  // the probe's counts have arrived and say "sparse again": back to the tables now
    } else if (p->lw_direct_left > 0) {
      --p->lw_direct_left;
      pl.use_tables = false;
    } else if (known && (double)fb_sum > 0.5 * (double)fb_rows && !has_row_index) {
      p->lw_direct_left = 15;  // this launch is the probe; the full-kernel launches behind it write no counts, so
      dense = true;            // "unknown" until the probe has run, then its verdict (above)
      for (int i = 0; i < 3; ++i) p->h_feedback[i] = 0xFFFFFFFFu;
    }
    if (known && fb_max <= 192u) pl.need_overflow = false;
    pl.fb_max = known ? fb_max : 0xFFFFFFFFu;
    if (known) {
      const uint32_t per = (uint32_t)std::max(8, kListRows);
      const uint32_t want = (fb_sum + per - 1u) / per;
      pl.lists = std::max(2, std::min(TSIMK_LW_LISTS, kMinLists));
      while ((uint32_t)pl.lists < want && pl.lists < TSIMK_LW_LISTS) pl.lists <<= 1;
      // the longest list of the last launch was measured with ITS list count: rescale the overflow test
      const uint32_t last = p->last_lists > 0 ? (uint32_t)p->last_lists : (uint32_t)TSIMK_LW_LISTS;
      const uint32_t est_max = (uint32_t)std::min<unsigned long long>(0xFFFFFFFFull, (unsigned long long)fb_max * last / (uint32_t)pl.lists + 16u);
      pl.need_overflow = !(pl.lists >= (int)last ? fb_max <= 192u : est_max <= 192u);
    }
    pl.defer = pipelined && pl.use_tables && !dense && !pl.need_overflow && pl.hard_kernel && p->knobs.defer &&
               p->v4 && !(p->profiling && !p->prof_light);
  }
  return pl;
}

// Every pipeline slot has its stream.  (Streams are created HERE, in one go, and nowhere earlier: HIP deals streams to its
// few hardware queues in creation order, and one more stream in front of the lanes - created at finalize for the table
// build of an earlier version - moved two lanes onto one queue: C4 at 10^5 shots per step 1.73 -> 1.07e10.)
static int slots_now_ready(tsim_program *p) {
  p->slots_ready = true;
  return 0;
}
static thread_local const LaunchPlan *g_carry_plan = nullptr;  // a plan drawn by the caller of tsim_sample_batch_device_begin

// k_sample4h geometry (LDS budget -> tiles per group), 0 tiles = the kernel cannot run this program
void hard_geometry(tsim_program *p, int WF, int WO) {
  constexpr int NW = TSIM_HARD_NW;
  const size_t tile_b = (size_t)p->v4_max_nch * 16 * p->v4_gt * 16;
  const size_t fixed_b = (size_t)(2 * WF + 2 * WO) * 64 * 4 + (size_t)NW * 8 * 64 * 4;
  const size_t budget = (size_t)kHardLdsKb * 1024;
  p->h_group_tiles = fixed_b + tile_b <= budget ? (int)std::min<size_t>(TSIMK_H_MAX_GROUP_TILES, (budget - fixed_b) / tile_b) : 0;
  p->h_lds = fixed_b + (size_t)std::max(1, p->h_group_tiles) * tile_b;
}

// One BLOCK per hard row (tsim_kernel_hw.hip.h) where the fast row layout exists and the rows are narrow: a batch then
// takes the time of one row, and the approximate branch is no slower than the exact one ...
// ... while the rows are FEW: a row costs this kernel ~10^4 wave instructions (one wave's worth of every level),
// the per-shot kernel ~1.5 * 10^3 - beyond ~10^3 rows per batch of launches the block-per-row kernel would take the
// vector ALUs from the first passes (C3: 340 rows per launch, p_bit 0.05: 5000), so those go the per-shot way
static bool hw_eligible(tsim_program *p, const SampleArgs &a, int n_ctx) {
  int wmax = 1;
  for (int w : p->comp_w) wmax = std::max(wmax, w);
  const uint32_t fb_rows = p->h_feedback ? p->h_feedback[0] : 0u;
  return p->fast && p->knobs.hard_wave && wmax <= 4 && a.WF <= 32 && a.WO <= 2 && p->hw_max_rows < 60000 &&
         (unsigned long long)fb_rows * (unsigned)n_ctx <= (unsigned long long)p->knobs.hard_wave_rows;
}
// Behind the latency kernels of a hard-row batch: the list slots they left (k_sample4_over, tsim_kernel4.hip.h).  A fixed
// grid of chip-resident blocks; when every list ends before `slot_begin` - all launches but the first after a jump of
// the noise level - they read the counts and exit (~3 us on the batch's stream).
static int launch_over(tsim_program *p, const SampleArgs *ctx, int n_ctx, uint32_t slot_begin, bool masked, hipStream_t hs, bool fill_chip = false) {
  Over4Multi M{};
  M.n_ctx = n_ctx;
  M.comp4_off = p->comp4_off;
  M.slot_begin = slot_begin;
  M.masked = masked ? 1 : 0;
  for (int i = 0; i < n_ctx; ++i) {
    M.ctx[i] = ctx[i];
    M.ctx[i].kernarg_off = (int)(offsetof(Over4Multi, ctx) + (size_t)i * sizeof(SampleArgs));
    M.ctx[i].row_slot_begin = 0;
    M.ctx[i].row_slot_end = 0;
  }
  // (a small footprint: in all launches but one the blocks only read the counts - 256 threads and ~25 KB of LDS start beside
  // a first pass that holds the chip, 512 threads with 60 KB wait for it and then delay the next one: 6-12 % of C2 / C3)
  const int blk = 256;
  const size_t tile_bytes = std::max((size_t)p->v4_max_nch * 16, (size_t)p->v4_max_sent) * p->v4_gt * 16;
  const size_t lds4 = (size_t)(2 * ctx[0].WF + 2 * ctx[0].WO) * blk * 4 + 2 * tile_bytes;
  if (lds4 > (p->v4_max_nch == 32 ? 160 : 64) * 1024) return tsim_fail(TSIM_ENOTSUP, "v4 kernel needs %zu B of LDS", lds4);
  // fill_chip: the lists are known to be long (throughput work) - as many blocks as the chip holds at once
  const unsigned grid = (unsigned)p->n_cu * (fill_chip ? (unsigned)std::max<size_t>(1, std::min<size_t>(2048 / blk, (160 * 1024) / (lds4 + 256))) : 1u);
  ++p->path_count[TP_OVER];
  switch (p->v4_max_nch) {
#define TSIM_LO(N) case N: hipLaunchKernelGGL((k_sample4_over<4, N>), dim3(grid), dim3(blk), lds4, hs, M); break;
    TSIM_LO(2) TSIM_LO(4) TSIM_LO(6) TSIM_LO(8) TSIM_LO(10) TSIM_LO(12) TSIM_LO(14) TSIM_LO(20)
#undef TSIM_LO
    case 32: {  // 81..128 parameters: two 32-KB tiles - more dynamic LDS than a kernel gets without asking
      auto kfn = k_sample4_over<4, 32>;
      if (!(p->x4_attr_set & 1u)) { HIP_TRY(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096)); p->x4_attr_set |= 1u; }
      hipLaunchKernelGGL(kfn, dim3(grid), dim3(blk), lds4, hs, M);
    } break;
    default: hipLaunchKernelGGL((k_sample4_over<4, 16>), dim3(grid), dim3(blk), lds4, hs, M); break;
  }
  HIP_TRY(hipGetLastError());
  return 0;
}
// the per-shot overflow grid exists for this program (chunk tables) and is not switched off
static bool over_available(const tsim_program *p) { return p->v4 && p->knobs.hard_overflow; }

static int launch_hw(tsim_program *p, const SampleArgs *ctx, int n_ctx, int max_lists, hipStream_t hs, bool partial, bool with_feedback = true) {
  int wmax = 1;
  for (int w : p->comp_w) wmax = std::max(wmax, w);
  const uint32_t fb_max = p->h_feedback ? p->h_feedback[1] : 192u;
  HwMulti H{};
  H.n_ctx = n_ctx;
  H.max_lists = max_lists;
  // as many blocks per list as the longest list of the last launch had rows (plus a margin): each block then takes one
  // row; longer lists are walked in turns (blocks of four waves: one row at a time each, see k_sample_hw)
  H.waves_per_list = (int)std::max(8u, std::min(128u, std::min(fb_max, 4096u) * 2u + 8u));
  H.feedback = with_feedback ? p->d_feedback : nullptr;
  for (int i = 0; i < n_ctx; ++i) {
    H.ctx[i] = ctx[i];
    H.ctx[i].kernarg_off = (int)(offsetof(HwMulti, ctx) + (size_t)i * sizeof(SampleArgs));
  }
  H.comp_par = partial ? (int)p->comps.size() : 1;  // the lists carry component masks: one block per (row, component)
  // each block takes at most four turns; per-shot worker blocks appended to the grid serve what lies behind them (lists
  // sized by STALE counts: the first group after a jump of the noise level) and exit at once otherwise
  const bool workers = over_available(p);
  H.slot_cap = workers ? (uint32_t)H.waves_per_list * 4u : 0u;
  H.comp4_off = p->comp4_off;
  const unsigned gridw = (unsigned)((long long)H.n_ctx * H.max_lists * H.waves_per_list * H.comp_par);
  H.hw_blocks = gridw;
  H.par_words = (int)(2 * ((p->hw_max_rows + 63) / 64) + 2);
  size_t ldsw = (size_t)H.par_words * 8 * 4 + 16;  // two buffers of four bit arrays (previous bit x trial bit) + the sampled bit.s word
  unsigned grid = gridw;
  if (workers) {
    const size_t tile_bytes = std::max((size_t)p->v4_max_nch * 16, (size_t)p->v4_max_sent) * p->v4_gt * 16;
    ldsw = std::max(ldsw, (size_t)(2 * ctx[0].WF + 2 * ctx[0].WO) * 256 * 4 + 2 * tile_bytes);
    if (ldsw > (p->v4_max_nch == 32 ? 160 : 64) * 1024) return tsim_fail(TSIM_ENOTSUP, "v4 kernel needs %zu B of LDS", ldsw);
    grid += (unsigned)(2 * p->n_cu);
  }
  // (blocks of 512 / 1024 threads - more helper waves per row - were tried: no faster alone, the row pass is not the chain;
  // next to a first pass slower, 8.8 -> 8.3 / 5.9e10 at --steps 200)
  const int nch = workers ? p->v4_max_nch : 0;
  ++p->path_count[TP_HW];
#define TSIM_LHW(WV, N) case N: hipLaunchKernelGGL((k_sample_hw<WV, N>), dim3(grid), dim3(256), ldsw, hs, H); break;
  if (wmax == 1) switch (nch) {
    TSIM_LHW(1, 0) TSIM_LHW(1, 2) TSIM_LHW(1, 4) TSIM_LHW(1, 6) TSIM_LHW(1, 8) TSIM_LHW(1, 10) TSIM_LHW(1, 12) TSIM_LHW(1, 14) TSIM_LHW(1, 16)
    default: return tsim_fail(TSIM_ESTATE, "bad chunk count %d", nch);
  } else if (wmax > 4) {  // parameter rows of 129..256 bits: the check / overflow rows of wide components with many graphs (class F140:
    // one lane of the row kernel took 413 us per batch for the check row of 140 graphs)
    if (nch != 0 || wmax > 8) return tsim_fail(TSIM_ESTATE, "block-per-row kernel: %d parameter words, %d chunks", wmax, nch);
    hipLaunchKernelGGL((k_sample_hw<8, 0>), dim3(grid), dim3(256), ldsw, hs, H);
  } else if (wmax > 2) {  // parameter rows of 65..128 bits (chunk tables, and so workers, only for 65..80 parameters: NCH = 20)
    switch (nch) {
      TSIM_LHW(4, 0) TSIM_LHW(4, 20)
      case 32: {
        auto kfn = k_sample_hw<4, 32>;
        if (!(p->x4_attr_set & 2u)) { HIP_TRY(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096)); p->x4_attr_set |= 2u; }
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), ldsw, hs, H);
      } break;
      default: return tsim_fail(TSIM_ESTATE, "hard-row workers need chunk tables (%d chunks)", nch);
    }
  } else switch (nch) {
    TSIM_LHW(2, 0) TSIM_LHW(2, 2) TSIM_LHW(2, 4) TSIM_LHW(2, 6) TSIM_LHW(2, 8) TSIM_LHW(2, 10) TSIM_LHW(2, 12) TSIM_LHW(2, 14) TSIM_LHW(2, 16)
    default: return tsim_fail(TSIM_ESTATE, "bad chunk count %d", nch);
  }
#undef TSIM_LHW
  HIP_TRY(hipGetLastError());
  return 0;
}

// The deferred second pass: ONE k_sample4h_multi grid serves the hard rows of every launch whose
// first pass is enqueued, on the third lane's stream, after those first passes.
int flush_batch(tsim_program *p) {
  if (p->deferred.empty()) {
    p->flush_inline = nullptr;
    return 0;
  }
  constexpr int NW = TSIM_HARD_NW;
  // (alternating the batches over two streams was tried for C4: both landed on ONE hardware queue, 29.2 -> 27.6 us per step
  // only with GPU_MAX_HW_QUEUES=8 - gone)
  const unsigned long long seq = p->batch_next++;
  // Inline (steps_group_fused, small groups): the batch runs on the group's own first-pass lane, behind its first pass.
  // A hard-row batch is latency-bound - ~30 us for the distillation shapes, ~80 us for the cultivation one, however
  // few the rows - so on ONE batch stream the batches of small groups (8 x 10^5 shots: a 36-us first pass) are the
  // pipeline's period; on the two lanes they overlap each other and the other lane's first pass.
  int bl = 0;
  hipStream_t hs = p->slots[3].side;
  if (p->flush_inline) {
    hs = p->flush_inline;
    bl = 2;
    for (int k = 0; k < 4; ++k)
      if (hs == p->slots[1 + k].side) bl = 2 + k;
    p->flush_inline = nullptr;
    p->inline_seen = true;
  }
  Hard4Multi M{};
  M.n_ctx = (int)p->deferred.size();
  const uint32_t fb_max = p->h_feedback ? p->h_feedback[1] : 192u;
  // (so many hard rows that they are throughput work - below: the 8-waves-per-64-rows blocks take ONE block's worth of every list,
  // the per-shot workers everything behind it)
  const uint32_t fb_sum0 = p->h_feedback ? p->h_feedback[0] : 0xFFFFFFFFu;
  const bool many0 = over_available(p) && fb_sum0 != 0xFFFFFFFFu && (unsigned long long)fb_sum0 * (unsigned)p->deferred.size() > 16384ull;
  const int hb = many0 ? 1 : (int)std::max(1u, std::min(4u, (std::min(fb_max, 192u) + 32u + 63u) / 64u));
  int max_lists = 1;
  for (int sidx : p->deferred) max_lists = std::max(max_lists, p->slots[sidx].ctx.row_lists);
  M.blocks_per_ctx = hb * max_lists + 1;
  M.group_tiles = p->h_group_tiles;
  M.loop_stride = hb * 64;
  M.comp4_off = p->comp4_off;
  M.feedback = p->d_feedback;
  // the batch starts after the first passes: streams are in order, so one event per lane covers them all
  bool lane_used[2] = {false, false};
  for (int i = 0; i < M.n_ctx; ++i) {
    tsim_program::Slot &d = p->slots[p->deferred[i]];
    for (int k = 0; k < 2; ++k)
      if (d.p1_stream == p->slots[1 + k].side) lane_used[k] = true;
    M.ctx[i] = d.ctx;
    M.ctx[i].kernarg_off = (int)(offsetof(Hard4Multi, ctx) + (size_t)i * sizeof(SampleArgs));
    if (d.ctx_check) M.check_mask |= 1 << i;
  }
  for (int k = 0; k < 2; ++k)
    if (lane_used[k] && p->slots[1 + k].side != hs) {
      if (!p->lane_ev[k]) HIP_TRY(hipEventCreateWithFlags(&p->lane_ev[k], hipEventDisableTiming));
      HIP_TRY(hipEventRecord(p->lane_ev[k], p->slots[1 + k].side));
      HIP_TRY(hipStreamWaitEvent(hs, p->lane_ev[k], 0));
    }
  // (a group whose first pass stored partial rows and component masks - steps_group_fused - was promised this kernel)
  const bool partial = p->slots[p->deferred[0]].partial;
  // (no chunk tables - a program of prefix-tree tables whose graphs exceed them: the block-per-row kernel is all there is)
  if (partial || hw_eligible(p, M.ctx[0], M.n_ctx) || !p->v4) {
    if (int r = launch_hw(p, M.ctx, M.n_ctx, max_lists, hs, partial)) return r;  // (its overflow workers ride in the same grid)
  } else {
  // (each block walks its list in strides of hb * 64 slots: at most 16 of them, the rest is the worker blocks')
  // ... or ONE stride when the last launches left so many hard rows that they are throughput work (more than 16384 per group -
  // programs whose tables stop at weight 4, class F70: 1.4 % of the rows, 112 000 per group, took 450 us on the 8-waves-per-64-rows
  // form; the per-shot workers take them at the full kernel's rate)
  const uint32_t fb_sum = p->h_feedback ? p->h_feedback[0] : 0xFFFFFFFFu;
  const bool many = fb_sum != 0xFFFFFFFFu && (unsigned long long)fb_sum * (unsigned)M.n_ctx > 16384ull;
  const uint32_t cap4h = over_available(p) ? (uint32_t)M.loop_stride * (many ? 1u : 16u) : 0u;
  for (int i = 0; i < M.n_ctx; ++i) M.ctx[i].row_slot_end = (int)cap4h;
  unsigned grid = (unsigned)(M.n_ctx * M.blocks_per_ctx);
  M.main_blocks = grid;
  M.over_from = cap4h;
  size_t lds_m = p->h_lds;
  // many: the workers are a grid of their own behind this one - inside it they would run with ITS dynamic LDS (up to 128 KB of
  // resident tiles: one block of eight waves per CU)
  const bool over_apart = many && cap4h != 0;
  if (cap4h && !over_apart) {  // the workers ride in the same grid (a kernel of their own behind this one cost C3 4 %)
    const size_t tile_bytes = std::max((size_t)p->v4_max_nch * 16, (size_t)p->v4_max_sent) * p->v4_gt * 16;
    lds_m = std::max(lds_m, (size_t)(2 * M.ctx[0].WF + 2 * M.ctx[0].WO) * (NW * 64) * 4 + 2 * tile_bytes);
    grid += (unsigned)p->n_cu * (many ? 2u : 1u);
  }
  if (over_apart) {
    for (hipEvent_t &e : p->over_ev)
      if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(p->over_ev[0], hs));
  }
  switch (p->v4_max_nch) {
#define TSIM_LHM(N)                                                                                          \
  case N: {                                                                                                  \
    auto kfn = k_sample4h_multi<4, N, NW>;                                                                   \
    if (!p->hm_attr_set) /* (the workers' list counts are 2 KB of static LDS) */                            \
      HIP_TRY(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024)); \
    ++p->path_count[TP_SAMPLE4H_MULTI];                                                                      \
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(NW * 64), lds_m, hs, M);                                        \
  } break;
    TSIM_LHM(2) TSIM_LHM(4) TSIM_LHM(6) TSIM_LHM(8) TSIM_LHM(10) TSIM_LHM(12) TSIM_LHM(14) TSIM_LHM(16) TSIM_LHM(20) TSIM_LHM(32)
#undef TSIM_LHM
    default: return tsim_fail(TSIM_ESTATE, "bad chunk count %d", p->v4_max_nch);
  }
  HIP_TRY(hipGetLastError());
  p->hm_attr_set = true;
  if (over_apart) {
    // ... and on a lane of its own: the latency kernel above is ONE pass's latency on a few CUs, the workers are throughput work
    // that need not wait for it (n16: 363 + 707 us per group one behind the other; 10^6 shots: n13 77 -> 66 us, n16 152 -> 124)
    hipStream_t os = p->slots[4].side != hs ? p->slots[4].side : p->slots[3].side;
    HIP_TRY(hipStreamWaitEvent(os, p->over_ev[0], 0));  // (recorded on hs in front of the latency kernel: behind the group's first passes)
    if (int r = launch_over(p, M.ctx, M.n_ctx, cap4h, false, os, true)) return r;
    HIP_TRY(hipEventRecord(p->over_ev[1], os));
    HIP_TRY(hipStreamWaitEvent(hs, p->over_ev[1], 0));
  }
  }
  ++p->stat_flushes;
  hipEvent_t &be = p->batch_ev[seq % 16u];
  if (!be) HIP_TRY(hipEventCreateWithFlags(&be, hipEventDisableTiming));
  if (seq > 16u) {  // the ring slot's previous batch: 16 batches ago, long done
    const int pl = p->batch_ev_lane[seq % 16u];
    if (p->batch_confirmed[pl] < seq - 16u) {
      HIP_TRY(hipEventSynchronize(be));
      p->batch_confirmed[pl] = seq - 16u;
    }
  }
  HIP_TRY(hipEventRecord(be, hs));
  p->batch_ev_lane[seq % 16u] = bl;
  for (int i = 0; i < M.n_ctx; ++i) {
    tsim_program::Slot &d = p->slots[p->deferred[i]];
    d.deferred = false;
    d.last_done = hs;
    d.done_ev = be;
    d.batch_seq = seq;
    d.batch_lane = bl;
  }
  p->deferred.clear();
  return 0;
}

// p->deferred in batches of at most TSIMK_H_MAX_CTX launches (a hard-row grid carries that many contexts in its kernel
// arguments; fused groups may be larger), all on the stream the caller chose
int flush_chunks(tsim_program *p) {
  if ((int)p->deferred.size() <= TSIMK_H_MAX_CTX) return flush_batch(p);
  std::vector<int> all;
  all.swap(p->deferred);
  hipStream_t inl = p->flush_inline;
  for (size_t i = 0; i < all.size(); i += TSIMK_H_MAX_CTX) {
    p->deferred.assign(all.begin() + i, all.begin() + std::min(all.size(), i + (size_t)TSIMK_H_MAX_CTX));
    p->flush_inline = inl;
    if (int r = flush_batch(p)) return r;
  }
  return 0;
}

// Everything that is waiting: the launches parked by the per-step API.
int tsim_flush_hard(tsim_program *p) { return flush_chunks(p); }

// The arguments every sampling kernel of one launch shares (SampleArgs): per-output subkeys - key, subkey =
// split(key) once per output, threaded through the components in processing order (sampler.py:74,147-148) - inline
// for programs with at most TSIMK_INLINE_KEYS compiled outputs, else by k_keygen on `s`; buffers; output layout.
int fill_sample_args(tsim_program *p, tsim_program::Slot &sl, SampleArgs &a, const uint64_t *d_f, int64_t B, int32_t num_f,
                            uint32_t key_hi, uint32_t key_lo, int64_t shot_offset, uint64_t *d_out, float *d_dev, hipStream_t s,
                            int slot, bool out_bit_packed) {
  if (p->total_keys > 0 && p->total_keys <= TSIMK_INLINE_KEYS) {
    uint32_t k0 = key_hi, k1 = key_lo;
    for (int i = 0; i < p->total_keys; ++i) {
      uint32_t a0 = 0u, a1 = 0u, b0 = 0u, b1 = 1u;
      threefry2x32(k0, k1, a0, a1);  // split(key)[0] -> next key
      threefry2x32(k0, k1, b0, b1);  // split(key)[1] -> this output's subkey
      a.inline_keys[2 * i] = b0;
      a.inline_keys[2 * i + 1] = b1;
      k0 = a0;
      k1 = a1;
    }
    a.n_inline_keys = p->total_keys;
  } else if (p->total_keys > 0 && p->total_keys <= TSIMK_KEYPUT_MAX) {
    sl.host_keys.resize(2 * (size_t)p->total_keys);  // (k_sample_gen takes them from here: gen_step_keys)
    KeyPutArgs K;
    K.n = p->total_keys;
    uint32_t k0 = key_hi, k1 = key_lo;
    for (int i = 0; i < p->total_keys; ++i) {
      uint32_t a0 = 0u, a1 = 0u, b0 = 0u, b1 = 1u;
      threefry2x32(k0, k1, a0, a1);
      threefry2x32(k0, k1, b0, b1);
      K.keys[2 * i] = sl.host_keys[2 * (size_t)i] = b0;
      K.keys[2 * i + 1] = sl.host_keys[2 * (size_t)i + 1] = b1;
      k0 = a0;
      k1 = a1;
    }
    hipLaunchKernelGGL(k_keyput, dim3(1), dim3(64), 0, s, K, sl.keys);
    HIP_TRY(hipGetLastError());
  } else if (p->total_keys > 0) {
    sl.host_keys.clear();
    hipLaunchKernelGGL(k_keygen, dim3(1), dim3(1), 0, s, key_hi, key_lo, p->total_keys, sl.keys);
    HIP_TRY(hipGetLastError());
  }
  a.img = p->d_img;
  a.f = d_f;
  a.out = d_out;
  a.subkeys = sl.keys;
  a.norm_dev = d_dev;
  a.B = B;
  a.shot_offset = shot_offset;
  a.WF = std::max(1, (num_f + 63) / 64);
  a.WO = (p->num_outputs + 63) / 64;
  a.n_direct = p->n_direct;
  a.direct_off = p->direct_off;
  a.direct_prog = p->lw_direct_prog;
  a.direct_chunks = p->lw_direct_chunks;
  a.n_comp = (int)p->comps.size();
  a.comp_off = p->comp_off;
  a.row_index = nullptr;  // (launch_sample sets an input row list)
  a.row_count = nullptr;
  a.row_lists = 0;
  a.row_list_cap = 0;
  a.row_slot_begin = 0;
  a.row_slot_end = 0;
  if (out_bit_packed) {  // TSIM_PIPE_OUT_BIT_PACKED: d_out IS the bit_packed buffer, no padded rows at all
    a.out = nullptr;
    a.out_compact = (uint8_t *)d_out;
    a.out_rb = (p->num_outputs + 7) / 8;
  } else if (sl.compact_out) {  // tsim_pipeline_set_compact_output: consumed by this launch
    a.out_compact = sl.compact_out;
    a.out_rb = (p->num_outputs + 7) / 8;
    sl.compact_out = nullptr;
  } else if (slot > 0 && p->series_left > 0) {  // tsim_pipeline_set_compact_series: next buffer of the series
    a.out_compact = p->series_ptr;
    a.out_rb = (p->num_outputs + 7) / 8;
    p->series_ptr += p->series_stride;
    --p->series_left;
  }
  a.check_row = nullptr;
  a.no_check = 0;
  if (num_f == 0) a.WF = 0;
  return 0;
}
static int launch_sample(tsim_program *p, const uint64_t *d_f, int64_t B, int32_t num_f, uint32_t key_hi,
                         uint32_t key_lo, int64_t shot_offset, uint64_t *d_out, float *d_dev, hipStream_t s,
                         const uint32_t *d_row_index = nullptr, const uint32_t *d_row_count = nullptr,
                         int slot = 0, const LaunchPlan *plan_in = nullptr, bool out_bit_packed = false) {
  if (!p->sampleable) return tsim_fail(TSIM_ESTATE, "program has joint-mode components (evaluate-only)");
  if (B < 0 || num_f < 0 || shot_offset < 0) return tsim_fail(TSIM_EINVAL, "negative B/num_f/shot_offset");
  // a stream that is not the handle's own: the caller's - a table swap has to wait for what it carries too (tsim_tables_extend_poll):
  // an event of the handle's is recorded behind this launch's last kernel (finish)
  bool foreign = !(s == p->stream || s == p->ext_stream);
  for (int k = 1; k <= TSIM_PIPELINE_SLOTS && foreign; ++k) foreign = !(p->slots[k].side_ready && p->slots[k].side == s);
  if (p->max_f_index >= num_f)
    return tsim_fail(TSIM_EINVAL, "program references f index %d but num_f=%d", p->max_f_index, num_f);
  if (B == 0 || p->num_outputs == 0) return 0;
  if (!d_f && num_f > 0) return tsim_fail(TSIM_EINVAL, "f buffer is NULL");
  if (!d_out) return tsim_fail(TSIM_EINVAL, "out buffer is NULL");
  // per-output subkeys: key, subkey = split(key) once per output, threaded through the
  // components in processing order (sampler.py:74,147-148)
  tsim_program::Slot &sl = p->slots[slot];
  {
    size_t hard_bytes = 0;
    if (p->lw || p->v4w) {
      const long long g1 = (B + 255) / 256;  // the pattern pass uses 256-thread blocks unless overridden
      hard_bytes = (size_t)((g1 + TSIMK_LW_LISTS - 1) / TSIMK_LW_LISTS * 1024) * TSIMK_LW_LISTS * 4;
    }
    if (int r = slot_prepare(p, slot, hard_bytes)) return r;
  }
  SampleArgs a{};
  if (int r = fill_sample_args(p, sl, a, d_f, B, num_f, key_hi, key_lo, shot_offset, d_out, d_dev, s, slot, out_bit_packed)) return r;
  a.row_index = d_row_index;
  a.row_count = d_row_index ? d_row_count : nullptr;
  if (B > 0x7FFFFFFFll * 64) return tsim_fail(TSIM_ENOTSUP, "batch too large");
  const bool prof = p->profiling && (p->prof_counter++ % p->prof_every == 0);
  if (prof) { int r = prof_event(p, s, PROF_BEGIN); if (r) return r; }
  // the normalisation check applies to in-batch shot 0 (sampler.py:66-72) or the first listed row
  bool has_check = (shot_offset == 0 || d_row_index);
  long long B2 = B;  // slots per row list of the full kernel's launch
  const bool pipelined = slot > 0;  // lane launch: the caller passed the slot's own stream as `s`
  auto finish = [&]() -> int {
    if (!pipelined) p->first_call_out.store(true, std::memory_order_release);
    if (foreign) {  // (looked up HERE: the launch plan drawn above may have swapped tables and dropped the events noted so far)
      hipEvent_t ev = nullptr;
      for (auto &cs : p->caller_streams)
        if (cs.s == s) ev = cs.ev;
      if (!ev && p->caller_streams.size() < 16) {
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        p->caller_streams.push_back({s, ev});
      }
      if (ev) HIP_TRY(hipEventRecord(ev, s));
      else p->caller_streams_overflow = true;
    }
    if (pipelined) {
      HIP_TRY(hipEventRecord(sl.ev2, s));
      sl.pending = true;
      sl.last_done = s;
      sl.done_ev = sl.ev2;
      sl.batch_seq = 0;
    }
    return 0;
  };
  const LaunchPlan plan = plan_in ? *plan_in : make_plan(p, d_row_index != nullptr, false, (unsigned long long)B);
  bool use_tables = plan.use_tables;
  const bool need_overflow = plan.need_overflow;
  // f rows wider than the round-2 wide kernels read (p->wide_big): k_sample_wide, or every row on the row kernel
  const bool big_out = p->wide_big && !(use_tables && !d_row_index && wide_applies(p, B, num_f, shot_offset) && wide_buffers_ok(p, a));
  if (big_out) use_tables = false;
  // a narrow program with more than 64 selected bits in a component: the one-batch first passes (k_sample_lw / _lw_reg / _lw_fast as a
  // group of one) hold f_sel in 64 bits - every row on the chunk-table kernel here; the fused groups ride k_sample_gen
  // ... and prefix-tree tables (components of more than 12 outputs, tsim_trie.hip.h) are walked by k_sample_gen only
  // - unless k_sample_gen takes the batch as a group of one (gen_one)
  const bool gen1 = (p->narrow_big || p->lw_trie) && use_tables && !d_row_index && gen_applies(p, B, num_f, shot_offset) &&
                    p->total_keys <= TSIMK_GEN_KEYS;
  if ((p->narrow_big || p->lw_trie) && !gen1) use_tables = false;
  if (int r = tsim_tables_slice(p, s)) return r;  // (a table build in the background: its next slice goes first)
  // The sparse-column pass (k_sample4w) over every row, or - behind a pattern-table first pass - over that pass's
  // hard-row lists (from_lists; `a` then describes them).  Its own overflow (more than K set bits, the check row)
  // goes to row lists of its own, which `a` describes afterwards: the row kernel below serves them.
  const bool wide_fits = p->v4w && !p->wide_big &&
                         (size_t)(2 * a.WF + 2 * a.WO) * 256 * 4 + 2 * (size_t)p->v4_max_sent * p->v4_gt * 16 <= 64 * 1024;
  auto wide_pass = [&](bool from_lists, int par) -> int {
    if (B > 0xFFFFFFFFll) return tsim_fail(TSIM_ENOTSUP, "batch too large for the row list");
    constexpr int kWideBlock = 256, kWideK = 10, kWideLists = 16;
    // (streamed tables: two buffers of up to 24 KB - groups of graphs, eval_level4_groups - never less than one graph's table)
    const size_t stage_b = (size_t)(2 * a.WF + 2 * a.WO) * kWideBlock * 4, one_tile = (size_t)p->v4_max_sent * p->v4_gt * 16;
    const size_t stream_buf = std::max(one_tile, std::min<size_t>(24 * 1024, (64 * 1024 - stage_b) / 2) / 16 * 16), stream_b = 2 * stream_buf;
    // all levels of a component resident in LDS when that still leaves room for two blocks per CU
    const bool resident = stage_b + p->v4w_resident_bytes <= 64 * 1024;
    const size_t ldsw = stage_b + (resident ? std::max(p->v4w_resident_bytes, (size_t)16) : stream_b);
    if (p->v4w_occ_lds != ldsw) {
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_sample4w<1, kWideK>, kWideBlock, ldsw) != hipSuccess || nb < 1) {
        (void)hipGetLastError();
        nb = 1;
      }
      p->v4w_occ_blocks = nb;
      p->v4w_occ_lds = ldsw;
    }
    const long long chip = (long long)p->n_cu * p->v4w_occ_blocks * 1;
    long long grid1, list_cap;
    if (from_lists) {
      // every input list gets the same number of blocks, which stride over it
      const long long chip_l = chip;
      const long long per_list = std::max(1ll, std::min(chip_l / a.row_lists, ((long long)a.row_list_cap + kWideBlock - 1) / kWideBlock));
      grid1 = per_list * a.row_lists;
      const long long rows_per_block = ((long long)a.row_list_cap + per_list * kWideBlock - 1) / (per_list * kWideBlock) * kWideBlock;
      // an output list takes the overflow of the blocks with its residue - never more than the launch has rows
      list_cap = std::min((grid1 + kWideLists - 1) / kWideLists * rows_per_block, (long long)(B + kWideBlock - 1) / kWideBlock * kWideBlock);
    } else {
      const long long blocks = (B + kWideBlock - 1) / kWideBlock;
      // One component with resident tables: the kernel copies the tables once per block and strides over the rows, so
      // the grid is what the chip holds at once (occupancy of this kernel with this much LDS), not one block per 256 rows
      grid1 = blocks;
      if (resident && p->comps.size() == 1) grid1 = std::min(blocks, chip);
      const long long iters = (blocks + grid1 - 1) / grid1;
      list_cap = (grid1 + kWideLists - 1) / kWideLists * iters * kWideBlock;
    }
    void *lists = from_lists ? sl.hard2 : sl.hard;
    uint32_t *ctl_base = from_lists ? sl.ctl2 : sl.ctl;
    const size_t lists_sz = from_lists ? sl.hard2_sz : sl.hard_sz;
    if (list_cap > 0x7FFFFFFFll) return tsim_fail(TSIM_ENOTSUP, "batch too large for the row lists");
    if (!lists || !ctl_base || (size_t)list_cap * kWideLists * 4 > lists_sz) return tsim_fail(TSIM_ESTATE, "row list too small");
    Wide4Args w;
    w.s = a;
    w.comp4_off = p->comp4_off;
    w.has_check = has_check ? 1 : 0;
    w.check_row_in = (from_lists && has_check) ? a.check_row : nullptr;
    w.feedback = from_lists ? p->d_feedback : nullptr;
    w.hard_index = (uint32_t *)lists;
    uint32_t *ctl = ctl_base + par * (TSIMK_LW_LISTS + 1) * 32;
    w.ctl = ctl;
    w.ctl_next = ctl_base + (par ^ 1) * (TSIMK_LW_LISTS + 1) * 32;
    w.list_cap = (int)list_cap;
    w.n_lists = kWideLists;
    w.resident = resident ? 1 : 0;
    w.stream_buf = (int)stream_buf;
    ++p->path_count[TP_SAMPLE4W];
    hipLaunchKernelGGL((k_sample4w<1, kWideK>), dim3((unsigned)grid1), dim3(kWideBlock), ldsw, s, w);
    HIP_TRY(hipGetLastError());
    if (prof && !(from_lists && p->prof_light)) { int r = prof_event(p, s, from_lists ? PROF_HARD : PROF_PASS1); if (r) return r; }
    a.row_index = w.hard_index;
    a.row_count = ctl;
    a.row_lists = kWideLists;
    a.row_list_cap = (int)list_cap;
    a.check_row = has_check ? ctl + 32 * TSIMK_LW_LISTS : nullptr;
    a.no_check = has_check ? 0 : 1;
    B2 = list_cap;
    return 0;
  };
  if (use_tables && !d_row_index && wide_applies(p, B, num_f, shot_offset) && wide_buffers_ok(p, a)) {
    // one wide component: everything in one kernel (tsim_wide.hip.h), here as a group of one batch
    const SampleArgs *one = &a;
    if (int r = launch_wide(p, 1, &one, B, num_f, shot_offset, s)) return r;
    if (prof) { int r = prof_event(p, s, PROF_PASS1); if (r) return r; }
    return finish();
  }
  if (use_tables) {
    // pass 1: shots whose f_sel patterns are tabulated finish here, the others go to the hard list
    if (B > 0xFFFFFFFFll) return tsim_fail(TSIM_ENOTSUP, "batch too large for the row list");
    // few large blocks: 1024 threads finish a batch of 10^6 rows in 977 blocks - measurably better than
    // 3906 blocks of 256 when the blocks of several launches and of the hard-row kernel share the CUs
    const int blk1 = ((size_t)(2 * a.WF + 2 * a.WO) * 1024 * 4 <= 32 * 1024 ? 1024 : 256);
    const bool reg_form = p->lw_reg && (a.WF == 1 || a.WF == 2) && a.WO == 1;
    const long long blocks = (B + blk1 - 1) / blk1;
    // the register form strides over the rows: no more blocks than the chip holds at once (2048 threads per CU)
    long long grid1 = blocks;
    if (reg_form)
      grid1 = std::min(blocks, (long long)p->n_cu * (2048 / blk1));
    const long long iters = (blocks + grid1 - 1) / grid1;
    // n_lists sub-lists share the buffer sized for TSIMK_LW_LISTS of them: a list can hold every row of
    // the blocks that feed it
    const int n_lists = plan.lists;
    const long long list_cap = (grid1 + n_lists - 1) / n_lists * iters * blk1;
    if (list_cap > 0x7FFFFFFFll) return tsim_fail(TSIM_ENOTSUP, "batch too large for the row lists");
    if ((size_t)list_cap * n_lists * 4 > sl.hard_sz) return tsim_fail(TSIM_ESTATE, "hard-row list too small");
    p->last_lists = n_lists;
    LwArgs l;
    l.s = a;
    l.tab = p->d_lw_tab;
    l.lw_off = p->lw_off;
    l.direct_prog = p->lw_direct_prog;
    l.direct_chunks = p->lw_direct_chunks;
    l.direct_rot = p->lw_direct_rot;
    l.has_check = has_check ? 1 : 0;
    l.hard_index = (uint32_t *)sl.hard;
    uint32_t *ctl = sl.ctl + sl.parity * (TSIMK_LW_LISTS + 1) * 32;
    l.ctl = ctl;
    l.ctl_next = sl.ctl + (sl.parity ^ 1) * (TSIMK_LW_LISTS + 1) * 32;
    sl.parity ^= 1;
    l.list_cap = (int)list_cap;
    l.n_lists = n_lists;
    l.binom_off = p->lw_binom_off;
    const size_t lds1 = (size_t)(2 * a.WF + 2 * a.WO) * blk1 * 4 + (p->lw_wide ? 4096 : 0);
    // one component of at most 8 outputs: the specialised pass of the fused groups (tsim_lw_fast.hip.h), as a group of ONE
    // batch - 14.5 instead of 19.6 us per 10^6 shots for the serial API too (same conditions as in steps_group_fused)
    const bool fast1 = reg_form && p->lwf_off != 0 && p->knobs.lw_fast && blk1 == 1024 && a.n_inline_keys > 0 && p->total_keys <= TSIMK_LWM_KEYS &&
                       B < (1ll << 28) && p->lw_bytes < (1ll << 32) && (n_lists & (n_lists - 1)) == 0 && a.WO == 1 &&
                       ((unsigned long long)shot_offset >> 32) == ((unsigned long long)(shot_offset + B - 1) >> 32);
    if (gen1) {
      long long cap1 = 0;
      if (int r = gen_one(p, sl, a, B, num_f, key_hi, key_lo, shot_offset, l.hard_index, l.ctl, l.ctl_next, n_lists, has_check, &cap1, s)) return r;
      if ((size_t)cap1 * n_lists * 4 > sl.hard_sz) return tsim_fail(TSIM_ESTATE, "hard-row list too small");
      l.list_cap = (int)cap1;
    } else if (fast1) {
      const long long cap1 = (blocks + n_lists - 1) / n_lists * blk1;  // the fused kernels' list geometry: row block rb -> list rb % n_lists
      if ((size_t)cap1 * n_lists * 4 > sl.hard_sz) return tsim_fail(TSIM_ESTATE, "hard-row list too small");
      LwMultiArgs M{};
      M.img = p->d_img;
      M.tab = p->d_lw_tab;
      M.B = B;
      M.shot_offset = shot_offset;
      M.n_steps = 1;
      M.blocks_per_step = (int)blocks;
      M.n_comp = (int)p->comps.size();
      M.lw_off = p->lw_off;
      M.direct_rot = p->lw_direct_rot;
      M.binom_off = p->lw_binom_off;
      M.has_check = has_check ? 1 : 0;
      M.list_cap = (int)cap1;
      M.n_lists = n_lists;
      M.out_rb = (p->num_outputs + 7) / 8;
      M.lwf_off = p->lwf_off;
      M.tab_bytes = (uint32_t)p->lw_bytes;
      LwStep &st = M.step[0];
      st.f = a.f;
      st.out = a.out;
      st.out_compact = a.out_compact;
      st.hard_index = l.hard_index;
      st.ctl = l.ctl;
      st.ctl_next = l.ctl_next;
      memcpy(st.keys, a.inline_keys, sizeof(uint32_t) * 2 * (size_t)p->total_keys);
      const long long chip = (long long)p->n_cu * (2048 / blk1);
      const long long it1 = (blocks + chip - 1) / chip;
      const long long gridf = (blocks + it1 - 1) / it1;
      const int n_out = p->comps[0].n_out;
#define TSIM_LF1(N)                                                                                        \
  case N:                                                                                                  \
    if (a.WF == 1) hipLaunchKernelGGL((k_sample_lw_fast<2, N>), dim3((unsigned)gridf), dim3(blk1), 0, s, M); \
    else hipLaunchKernelGGL((k_sample_lw_fast<4, N>), dim3((unsigned)gridf), dim3(blk1), 0, s, M);          \
    break;
      ++p->path_count[TP_LW_FAST1];
      switch (n_out) {
        TSIM_LF1(1) TSIM_LF1(2) TSIM_LF1(3) TSIM_LF1(4) TSIM_LF1(5) TSIM_LF1(6) TSIM_LF1(7) TSIM_LF1(8)
        default: return tsim_fail(TSIM_ESTATE, "fast record with %d outputs", n_out);
      }
#undef TSIM_LF1
      l.list_cap = (int)cap1;
    } else if (reg_form) {
      // narrow rows: everything in registers, no LDS
      ++p->path_count[TP_LW_REG];
      if (a.WF == 1) hipLaunchKernelGGL(k_sample_lw_reg<2>, dim3((unsigned)grid1), dim3(blk1), 0, s, l);
      else hipLaunchKernelGGL(k_sample_lw_reg<4>, dim3((unsigned)grid1), dim3(blk1), 0, s, l);
    } else {
      if (lds1 > 64 * 1024) return tsim_fail(TSIM_ENOTSUP, "num_f + num_outputs too large for LDS staging (%zu B)", lds1);
      ++p->path_count[p->lw_wide ? TP_LW_LDS_WIDE : TP_LW_LDS];
      if (p->lw_wide) hipLaunchKernelGGL(k_sample_lw<true>, dim3((unsigned)grid1), dim3(blk1), lds1, s, l);
      else hipLaunchKernelGGL(k_sample_lw<false>, dim3((unsigned)grid1), dim3(blk1), lds1, s, l);
    }
    HIP_TRY(hipGetLastError());
    if (prof) { int r = prof_event(p, s, PROF_PASS1); if (r) return r; }
    // pass 2 below runs on the hard lists; the check row was forced into one of them
    a.row_index = l.hard_index;
    a.row_count = ctl;
    a.row_lists = n_lists;
    a.row_list_cap = l.list_cap;
    a.check_row = has_check ? ctl + 32 * TSIMK_LW_LISTS : nullptr;
    a.no_check = has_check ? 0 : 1;
    B2 = l.list_cap;
    // wide components: the listed rows go through the sparse-column pass first, its overflow to the row kernel
    // (ctl2 alternates on its own: launches that skip the tables - dense batches - do not touch it, and its reset
    // is done by the launch that used it last)
    if (p->lw_wide && wide_fits) {
      if (int r = wide_pass(true, sl.parity2)) return r;
      sl.parity2 ^= 1;
    }
  } else if (wide_fits) {
    // wide components: sparse-column pass on every row (k_sample4w); rows with more than K set f bits and the
    // normalisation-check row go to the row lists, which the row kernel below serves
    if (int r = wide_pass(false, sl.parity)) return r;
    sl.parity ^= 1;
  } else if (!has_check) {
    a.no_check = 1;
  }
  int block = 256;
  size_t lds = (size_t)(2 * a.WF + 2 * a.WO) * block * 4;
  if (lds > 60 * 1024) { block = 64; lds = (size_t)(2 * a.WF + 2 * a.WO) * block * 4; }
  if (lds > 60 * 1024) return tsim_fail(TSIM_ENOTSUP, "num_f + num_outputs too large for LDS staging (%zu B)", lds);
  const long long nlists = a.row_lists > 1 ? a.row_lists : 1;
  const long long grid = (B2 + block - 1) / block * nlists;
  if (grid > 0x7FFFFFFFll) return tsim_fail(TSIM_ENOTSUP, "batch too large");
  if (p->v4) {
    // chunk-table kernel: LDS = f/out staging + two tile buffers; one extra block replays shot 0
    Sample4Args a4;
    a4.s = a;
    a4.comp4_off = p->comp4_off;
    a4.has_check = has_check ? 1 : 0;
    a4.feedback = (a.row_lists > 1 && !plan.hard_kernel) ? p->d_feedback : nullptr;
    const int blk = kV4Block;
    const size_t tile_bytes = std::max((size_t)p->v4_max_nch * 16, (size_t)p->v4_max_sent) * p->v4_gt * 16;
    const size_t lds4 = (size_t)(2 * a.WF + 2 * a.WO) * blk * 4 + 2 * tile_bytes;
    if (lds4 > (p->v4_max_nch == 32 ? 160 : 64) * 1024) return tsim_fail(TSIM_ENOTSUP, "v4 kernel needs %zu B of LDS", lds4);
    if (a.row_lists > 1 && plan.hard_kernel) {
      // short row lists (second pass of a two-pass launch): NW waves per 64 rows, tsim_kernel4h.hip.h
      constexpr int NW = TSIM_HARD_NW;
      hard_geometry(p, a.WF, a.WO);
      const int group_tiles = p->h_group_tiles;
      if (group_tiles >= 1 && plan.defer && pipelined) {
        // leave the hard rows to the next batch: this lane goes on with the next launch's first pass
        if (!p->deferred.empty() && (p->slots[p->deferred[0]].ctx.WF != a.WF || p->slots[p->deferred[0]].ctx.WO != a.WO))
          if (int r = tsim_flush_hard(p)) return r;  // one LDS layout per batch
        sl.ctx = a;
        sl.ctx.row_slot_begin = 0;
        sl.ctx.row_slot_end = 0;
        sl.ctx_check = has_check;
        sl.deferred = true;
        sl.pending = true;
        sl.p1_stream = s;
        sl.partial = false;
        p->deferred.push_back(slot);
        if ((int)p->deferred.size() >= p->knobs.defer_group) return tsim_flush_hard(p);
        return 0;
      }
      if (group_tiles >= 1 && hw_eligible(p, a, 1)) {
        // few hard rows, fast row layout: one block per row (tsim_kernel_hw.hip.h), whole lists - nothing left for k_sample4
        if (int r = launch_hw(p, &a, 1, a.row_lists, s, false)) return r;
        if (prof && !p->prof_light) { int r = prof_event(p, s, PROF_HARD); if (r) return r; }
        if (prof && !(p->prof_light && use_tables)) { int r = prof_event(p, s, PROF_FULL); if (r) return r; }
        return finish();
      }
      if (group_tiles >= 1) {
        // the first kHardBlocks * 64 slots of every list go to the NW-wave kernel; k_sample4 below
        // serves the rest (its blocks exit at once when the lists are short - the usual case)
        constexpr int kHardBlocks = 4;
        const size_t ldsh = p->h_lds;
        const long long gridh = (long long)kHardBlocks * nlists + a4.has_check;
        Sample4Args ah = a4;
        const int cap1 = over_available(p) ? kHardBlocks * 64 * 16 : 0;  // without an overflow launch: 16 strides, then k_sample4_over
        ah.s.row_slot_end = need_overflow ? kHardBlocks * 64 : cap1;
        const int loop_stride = need_overflow ? 0 : kHardBlocks * 64;
        switch (p->v4_max_nch) {
#define TSIM_LH(N)                                                                                          \
  case N: {                                                                                                 \
    auto kfn = k_sample4h<4, N, NW>;                                                                        \
    if (!p->h_attr_set)                                                                                     \
      HIP_TRY(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
    ++p->path_count[TP_SAMPLE4H];                                                                           \
    hipLaunchKernelGGL(kfn, dim3((unsigned)gridh), dim3(NW * 64), ldsh, s, ah, group_tiles, loop_stride,    \
                       p->d_feedback);                                                                      \
  } break;
          TSIM_LH(2) TSIM_LH(4) TSIM_LH(6) TSIM_LH(8) TSIM_LH(10) TSIM_LH(12) TSIM_LH(14) TSIM_LH(16) TSIM_LH(20) TSIM_LH(32)
#undef TSIM_LH
          default: return tsim_fail(TSIM_ESTATE, "bad chunk count %d", p->v4_max_nch);
        }
        HIP_TRY(hipGetLastError());
        if (prof && !p->prof_light) { int r = prof_event(p, s, PROF_HARD); if (r) return r; }
        p->h_attr_set = true;
        a4.has_check = 0;  // done by the kernel above
        a4.s.no_check = 1;
        a4.s.row_slot_begin = kHardBlocks * 64;
        B2 = need_overflow ? std::max<long long>(0, B2 - kHardBlocks * 64) : 0;
        if (!need_overflow && cap1)
          if (int r = launch_over(p, &a, 1, (uint32_t)cap1, false, s)) return r;
      }
    }
    const long long grid4 = (B2 + blk - 1) / blk * nlists + a4.has_check;
    if (grid4 > 0x7FFFFFFFll) return tsim_fail(TSIM_ENOTSUP, "batch too large");
    if (grid4 > 0) ++p->path_count[TP_SAMPLE4];
    if (grid4 > 0) switch (p->v4_max_nch) {
#define TSIM_L4(N) case N: hipLaunchKernelGGL((k_sample4<4, N>), dim3((unsigned)grid4), dim3(blk), lds4, s, a4); break;
      TSIM_L4(2) TSIM_L4(4) TSIM_L4(6) TSIM_L4(8) TSIM_L4(10) TSIM_L4(12) TSIM_L4(14) TSIM_L4(20)
#undef TSIM_L4
      case 32: {
        auto kfn = k_sample4<4, 32>;
        if (!(p->x4_attr_set & 4u)) { HIP_TRY(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096)); p->x4_attr_set |= 4u; }
        hipLaunchKernelGGL(kfn, dim3((unsigned)grid4), dim3(blk), lds4, s, a4);
      } break;
      default: hipLaunchKernelGGL((k_sample4<4, 16>), dim3((unsigned)grid4), dim3(blk), lds4, s, a4); break;
    }
    HIP_TRY(hipGetLastError());
    if (prof && !(p->prof_light && use_tables)) { int r = prof_event(p, s, PROF_FULL); if (r) return r; }
    return finish();
  }
  int wmax = 1;
  for (int w : p->comp_w) wmax = std::max(wmax, w);
  // The rows left in lists behind the sparse-column pass (more than K set bits, the normalisation-check row): ONE lane of the row
  // kernel walks all levels and graphs of such a row by itself - 280 us per batch for the check row of a 148-term program with
  // 65 parameters (class F60, profiles/r05/shape_map.txt) - the block-per-row kernel takes ~25.  (Programs of a few graphs - the C5 family - are
  // quicker on the one lane: 2wide 111 -> 127 us per step with the block-per-row kernel, measured.)
  // (no feedback from this grid: the plan's counts are the sparse-column pass's - the rows that miss the tables - not its overflow)
  if (a.row_lists > 1 && !p->v4 && use_tables &&
      ((gen1 && p->lw_trie) || (p->fast && p->knobs.hard_wave && wmax <= 8 && p->total_graphs >= 64 && a.WF <= 32 && a.WO <= 2 && p->hw_max_rows < 60000))) {  // (gen1: as flush_batch does)
    if (int r = launch_hw(p, &a, 1, a.row_lists, s, false, false)) return r;
    if (prof && !(p->prof_light && use_tables)) { int r = prof_event(p, s, PROF_FULL); if (r) return r; }
    return finish();
  }
  ++p->path_count[TP_ROWS];
  if (int r = tsim_launch_rows(p, wmax, a, grid, block, lds, s)) return r;
  if (prof && !(p->prof_light && use_tables)) { int r = prof_event(p, s, PROF_FULL); if (r) return r; }
  return finish();
}
// A slot's previous launch (its lists, counters and output rows are reused) may have finished on another stream than
// `s`: order this launch after it.  Batches complete in order (one stream) and a lane is in order too: once a lane
// waits for batch b it is behind every batch <= b.  The slots of a batch alternate over the two first-pass lanes, so
// this is one stream wait per lane and batch - no event query (the host usually runs several batches ahead of the
// GPU, the query would fail and cost as much as the wait).
int slot_order_after_previous(tsim_program *p, tsim_program::Slot &sl, hipStream_t s) {
  if (!(sl.last_done && sl.last_done != s && sl.done_ev)) return 0;
  bool done = false;
  if (sl.batch_seq) {
    int lane = -1;
    for (int k = 0; k < 4; ++k)
      if (s == p->slots[1 + k].side) lane = k;
    const int bl = sl.batch_lane;  // the stream that batch ran on (batches of one stream complete in order)
    if (sl.batch_seq <= p->batch_confirmed[bl] || (lane >= 0 && sl.batch_seq <= p->lane_waited[lane][bl])) done = true;
    else if (lane >= 0) {
      p->lane_waited[lane][bl] = sl.batch_seq;
      // how far back the slots of this caller's rotation reach, in batches (for the pre-wait of _begin)
      if (p->deferred.size() < 2 && p->batch_next > sl.batch_seq) p->lane_reach[lane] = (int)(p->batch_next - sl.batch_seq);
    }
  } else {
    ++p->stat_queries;
    done = hipEventQuery(sl.done_ev) == hipSuccess;
    if (!done) (void)hipGetLastError();
  }
  if (!done) { ++p->stat_waits; HIP_TRY(hipStreamWaitEvent(s, sl.done_ev, 0)); }
  return 0;
}

extern "C" int tsim_sample_batch_device_begin(tsim_program *p, int32_t slot, const uint64_t *d_f, int64_t B,
                                              int32_t num_f, uint32_t key_hi, uint32_t key_lo, int64_t shot_offset,
                                              uint64_t *d_out, float *d_max_norm_dev, void *stream, uint32_t flags) {
  const bool carried = g_carry_plan != nullptr;
  const LaunchPlan carried_plan = carried ? *g_carry_plan : LaunchPlan{};
  g_carry_plan = nullptr;
  if (int r = tsim_need_final(p)) return r;
  if (int r = tsim_set_device(p)) return r;
  if (slot < 0 || slot >= TSIM_PIPELINE_SLOTS) return tsim_fail(TSIM_EINVAL, "slot %d out of range", slot);
  hipStream_t s_user = stream ? (hipStream_t)stream : p->stream;
  tsim_program::Slot &sl = p->slots[1 + slot];
  if (!p->slots_ready) {  // first pipelined launch: create every slot's stream/buffers now, not mid-run
    size_t hard_bytes = 0;
    if (p->lw || p->v4w) hard_bytes = (size_t)(((B + 255) / 256 + TSIMK_LW_LISTS - 1) / TSIMK_LW_LISTS * 1024) * TSIMK_LW_LISTS * 4;
    (void)hard_bytes;  // (each launch sizes its own slot's lists: launch_sample)
    for (int k = 1; k <= 4; ++k)  // the lanes, in one go (hardware queues follow the creation order); other slots' buffers on first use
      if (int r = slot_prepare(p, k, 0)) return r;
    if (int r = slots_now_ready(p)) return r;
  }
  if (int r = slot_prepare(p, 1 + slot, 0)) return r;
  // The whole launch runs on the slot's own stream (a "lane"): launches of one slot are ordered by the
  // stream itself, launches of different slots overlap.  Unless the caller vouches for its inputs the
  // lane first waits for what is already queued on the caller's stream.
  // Deferred plan (short hard-row lists): the first passes alternate between the first two lanes and
  // the hard rows of several launches go to the third lane in one batch (flush_hard), so no lane
  // waits for a second pass before it starts the next first pass.
  if (sl.deferred)  // begin twice without end: finish the earlier launch's hard rows first
    if (int r = tsim_flush_hard(p)) return r;
  // (tsim_sample_steps_device has already drawn the plan of this launch: drawing it twice would count the launch twice
  // and swallow the probe that leads a dense phase back to the tables)
  const LaunchPlan plan = carried ? carried_plan : make_plan(p, false, true, (unsigned long long)B);
  if (!plan.defer && !sl.side_ready)
    if (int r = slot_prepare(p, 1 + slot, 0, true)) return r;
  hipStream_t s = plan.defer ? p->slots[1 + (slot & 1)].side : sl.side;
  if (!plan.defer) {
    // the slot's own stream: tsim_pipeline_wait_stream orders only streams that carried work before - a first launch here
    // (a plan that leaves the deferred path: dense batches, unknown feedback) must still see what the caller ordered
    if (sl.needs_sync && p->sync_ev) HIP_TRY(hipStreamWaitEvent(s, p->sync_ev, 0));
    sl.needs_sync = false;
    sl.used = true;
  }
  if (int r = slot_order_after_previous(p, sl, s)) return r;
  // Pre-wait, mid-batch.  When the caller rotates through a slot count that is a multiple of the batch size, the
  // wait a lane needs for its next batch of slots falls on the lane's FIRST launch of that batch - right behind the
  // event record of the flush that closed the previous batch: two non-kernel packets in a row, on both lanes at
  // once, and the kernel trace shows both lanes idle for 25-40 us after every flush (16 slots: 16.3 us per step
  // where 14 slots, whose waits fall mid-batch, reach 15.0).  So the lane's SECOND launch of a batch (which needs
  // no wait of its own) already waits for the batch the first launch of the next batch will ask for - known from
  // how far back the last such wait reached (lane_reach) and at least two flushes old, i.e. complete; that launch
  // then finds the lane already behind it.  Ordering only ever gets stricter.
  if (plan.defer && !p->inline_seen && p->deferred.size() >= 2) {
    const int lane = (s == p->slots[1].side) ? 0 : (s == p->slots[2].side) ? 1 : -1;
    if (lane >= 0 && p->lane_reach[lane] >= 3 && p->lane_reach[lane] <= 15 && p->batch_next > (unsigned long long)p->lane_reach[lane]) {
      const unsigned long long want = p->batch_next - (unsigned long long)(p->lane_reach[lane] - 1);
      hipEvent_t ev = p->batch_ev[want % 16u];
      if (ev && want > p->lane_waited[lane][0] && want > p->batch_confirmed[0]) {
        p->lane_waited[lane][0] = want;
        ++p->stat_waits;
        HIP_TRY(hipStreamWaitEvent(s, ev, 0));
      }
    }
  }
  ++p->stat_begins;
  if (plan.defer) ++p->stat_deferred;
  if (!(flags & TSIM_PIPE_INPUTS_READY) && s_user != s) {
    HIP_TRY(hipEventRecord(sl.ev1, s_user));
    HIP_TRY(hipStreamWaitEvent(s, sl.ev1, 0));
  }
  return launch_sample(p, d_f, B, num_f, key_hi, key_lo, shot_offset, d_out, d_max_norm_dev, s, nullptr, nullptr,
                       1 + slot, &plan, (flags & TSIM_PIPE_OUT_BIT_PACKED) != 0);
}

// ---------------------------------------------------------------------------
// Several consecutive batches in one call (the reference's batch loop, sampler.py:340-420: per batch
// `key, subkey = split(key)` and one sample_program).  Batches whose first pass is the register form go out in
// groups of up to TSIMK_LWM_MAX_STEPS as ONE grid (k_sample_lw_multi, tsim_lw_multi.hip.h) on a first-pass lane,
// lanes alternating between groups; the group's hard rows are one k_sample4h_multi batch on the batch lane behind it
// (tsim_flush_hard) while the next group's first pass runs.  Anything else - dense plans, wide programs, row kernels -
// goes through tsim_sample_batch_device_begin batch by batch: same results either way.
// ---------------------------------------------------------------------------

static int steps_group_fused(tsim_program *p, int n, const uint64_t *const *d_f, int64_t B, int32_t num_f, uint32_t key[2],
                             int64_t shot_offset, void *const *d_out, float *const *d_dev, uint32_t flags, const LaunchPlan &plan) {
  const bool packed = (flags & TSIM_PIPE_OUT_BIT_PACKED) != 0;
  if (!p->deferred.empty())
    if (int r = flush_batch(p)) return r;  // rows parked by batch-by-batch launches: their own batch first
  // two first-pass lanes; three for small groups, whose hard-row grids (latency-bound, on the group's lane) take longer
  // than their first passes (C4 at 8 x 10^5 shots per group: 1.16 -> 1.53e10 shots/s; the large groups of C2: no change)
  const int lanes = p->knobs.fused_lanes > 0 ? p->knobs.fused_lanes : ((long long)n * B <= (1ll << 21) ? 3 : 2);
  hipStream_t s = p->slots[1 + (int)(p->steps_groups++ % (unsigned long long)lanes)].side;
  if (!(flags & TSIM_PIPE_INPUTS_READY) && p->stream != s) {
    if (!p->sync_ev) HIP_TRY(hipEventCreateWithFlags(&p->sync_ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(p->sync_ev, p->stream));
    HIP_TRY(hipStreamWaitEvent(s, p->sync_ev, 0));
  }
  const int WF = std::max(1, (num_f + 63) / 64);
  const int blk1 = 1024;
  const long long bps = (B + blk1 - 1) / blk1;
  const int n_lists = plan.lists;
  if (p->stat_fused == 0 && tsim_debug("pipeline")) {
    int nb = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_sample_lw_multi<2>, blk1, 0);
    fprintf(stderr, "[tsim] k_sample_lw_multi<2>: %d blocks of %d threads per CU (occupancy API)\n", nb, blk1);
  }
  const long long list_cap = (bps + n_lists - 1) / n_lists * blk1;
  if (list_cap > 0x7FFFFFFFll || bps * n > 0x7FFFFFFFll) return tsim_fail(TSIM_ENOTSUP, "batch too large for the row lists");
  const bool has_check = shot_offset == 0;
  LwMultiArgs M{};
  M.img = p->d_img;
  M.tab = p->d_lw_tab;
  M.B = B;
  M.shot_offset = shot_offset;
  M.n_steps = n;
  M.blocks_per_step = (int)bps;
  M.n_comp = (int)p->comps.size();
  M.lw_off = p->lw_off;
  M.direct_rot = p->lw_direct_rot;
  M.binom_off = p->lw_binom_off;
  M.has_check = has_check ? 1 : 0;
  M.list_cap = (int)list_cap;
  M.n_lists = n_lists;
  M.out_rb = (p->num_outputs + 7) / 8;
  p->last_lists = n_lists;
  int slots[TSIMK_LWM_MAX_STEPS];
  for (int j = 0; j < n; ++j)  // a slot this group takes still waits for its hard rows (many lanes, few slots): those first
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
    LwStep &st = M.step[j];
    st.f = d_f[j];
    st.out = a.out;
    st.out_compact = a.out_compact;
    st.hard_index = (uint32_t *)sl.hard;
    uint32_t *ctl = sl.ctl + sl.parity * (TSIMK_LW_LISTS + 1) * 32;
    st.ctl = ctl;
    st.ctl_next = sl.ctl + (sl.parity ^ 1) * (TSIMK_LW_LISTS + 1) * 32;
    sl.parity ^= 1;
    memcpy(st.keys, a.inline_keys, sizeof(uint32_t) * 2 * (size_t)p->total_keys);
    // what the hard-row batch needs of this launch (launch_sample fills the same fields behind its first pass)
    a.row_index = st.hard_index;
    a.row_count = ctl;
    a.row_lists = n_lists;
    a.row_list_cap = (int)list_cap;
    a.check_row = has_check ? ctl + 32 * TSIMK_LW_LISTS : nullptr;
    a.no_check = has_check ? 0 : 1;
    a.row_slot_begin = 0;
    a.row_slot_end = 0;
  }
  // one chip-full of blocks (2048 threads per CU), every block the same number of (batch, row block) pairs
  const long long total = bps * n;
  const long long chip = (long long)p->n_cu * (2048 / blk1);
  const long long iters = (total + chip - 1) / chip;
  const long long grid = (total + iters - 1) / iters;
  TSIM_MARK("args");
  if (int r = tsim_tables_slice(p, s)) return r;  // (a table build in the background: its next slice goes first)
  const bool prof = p->profiling && (p->prof_counter++ % p->prof_every == 0);
  if (prof) { if (int r = prof_event(p, s, PROF_BEGIN)) return r; }
  // one component of at most 8 outputs: the specialised pass (tsim_lw_fast.hip.h) - 32-bit byte offsets everywhere, so
  // batches below 2^28 rows, tables below 4 GB, and a shot range that does not cross a multiple of 2^32
  bool part = false;
  const bool fast = p->lwf_off != 0 && p->knobs.lw_fast && B < (1ll << 28) && p->lw_bytes < (1ll << 32) && (n_lists & (n_lists - 1)) == 0 &&
                    ((unsigned long long)shot_offset >> 32) == ((unsigned long long)(shot_offset + B - 1) >> 32);
  M.lwf_off = fast ? p->lwf_off : 0;
  M.tab_bytes = fast ? (uint32_t)p->lw_bytes : 0u;
  // device noise in front of this group (tsim_sample_steps_noise_device): inside the first pass when that is the one-component
  // register pass and the sampler's tile is a whole number of row blocks, else k_noise_wave for every batch on this lane
  TsimNoiseRequest *nq = g_noise_req;
  const bool fuse_noise = nq && fast && nq->fusable && nq->N.WF == WF && (nq->N.tile % blk1) == 0 && p->knobs.noise_fused;
  if (nq && !fuse_noise)
    for (int j = 0; j < n; ++j)
      if (int r = nq->launch(nq->noise, B, nq->keys[2 * (nq->base + j)], nq->keys[2 * (nq->base + j) + 1], const_cast<uint64_t *>(d_f[j]), s)) return r;
  if (fuse_noise) {
    NoiseFusedArgs FA{};
    FA.M = M;
    FA.N = nq->N;
    for (int j = 0; j < n; ++j) {
      FA.nkeys[2 * j] = nq->keys[2 * (nq->base + j)];
      FA.nkeys[2 * j + 1] = nq->keys[2 * (nq->base + j) + 1];
    }
    const long long tiles = ((B + nq->N.tile - 1) / nq->N.tile) * n;
    const long long gridn = std::max<long long>(1, std::min<long long>((long long)p->n_cu * 2, tiles));
    const size_t ldsn = (size_t)nq->N.tile * nq->N.WF * 8 + (size_t)nq->N.n_ch * 24;
    const int n_out = p->comps[0].n_out;
#define TSIM_LNF(N)                                                                                              \
  case N:                                                                                                        \
    if (WF == 1) hipLaunchKernelGGL((k_noise_sample_fast<2, N>), dim3((unsigned)gridn), dim3(blk1), ldsn, s, FA); \
    else hipLaunchKernelGGL((k_noise_sample_fast<4, N>), dim3((unsigned)gridn), dim3(blk1), ldsn, s, FA);        \
    break;
    ++p->path_count[TP_NOISE_FAST];
    switch (n_out) {
      TSIM_LNF(1) TSIM_LNF(2) TSIM_LNF(3) TSIM_LNF(4) TSIM_LNF(5) TSIM_LNF(6) TSIM_LNF(7) TSIM_LNF(8)
      default: return tsim_fail(TSIM_ESTATE, "fast record with %d outputs", n_out);
    }
#undef TSIM_LNF
    ++p->stat_fast;
  } else if (fast) {
    const int n_out = p->comps[0].n_out;
#define TSIM_LF(N)                                                                                     \
  case N:                                                                                              \
    if (WF == 1) hipLaunchKernelGGL((k_sample_lw_fast<2, N>), dim3((unsigned)grid), dim3(blk1), 0, s, M); \
    else hipLaunchKernelGGL((k_sample_lw_fast<4, N>), dim3((unsigned)grid), dim3(blk1), 0, s, M);        \
    break;
    ++p->path_count[TP_LW_FAST];
    switch (n_out) {
      TSIM_LF(1) TSIM_LF(2) TSIM_LF(3) TSIM_LF(4) TSIM_LF(5) TSIM_LF(6) TSIM_LF(7) TSIM_LF(8)
      default: return tsim_fail(TSIM_ESTATE, "fast record with %d outputs", n_out);
    }
#undef TSIM_LF
    ++p->stat_fast;
  } else if (p->lwfm_off != 0 && p->knobs.lw_fast && B < (1ll << 28) && p->lw_bytes < (1ll << 32) &&
             (n_lists & (n_lists - 1)) == 0 && ((unsigned long long)shot_offset >> 32) == ((unsigned long long)(shot_offset + B - 1) >> 32)) {
    // 2..4 components of at most 8 outputs each: the specialised pass with per-component tables (tsim_lw_fastm.hip.h)
    M.lwf_off = p->lwfm_off;
    M.tab_bytes = (uint32_t)p->lw_bytes;
    // Component-parallel hard rows: the pass stores the hard rows too (direct outputs + the tabulated components' bits) and
    // marks in the list entry which components are left; k_sample_hw runs one block per (row, component) and ORs the bits in.
    // C4's batches then wait for the slowest component (~25 us) instead of the three in turn (~65 us).  Decided HERE, once:
    // the hard-row launch below must be the block-per-row kernel, whatever the feedback says by then.  (Compact rows are
    // merged with 32-bit atomics: the batch's byte range must not share a word with another batch's.)
    {
      bool aligned = true;
      for (int j = 0; j < n; ++j) {
        const SampleArgs &aj = p->slots[slots[j]].ctx;
        if (aj.out_compact && ((((uintptr_t)aj.out_compact) & 3u) || (((unsigned long long)B * (unsigned long long)aj.out_rb) & 3ull))) aligned = false;
      }
      part = p->knobs.hard_comp_par && p->comps.size() >= 2 && aligned && hw_eligible(p, p->slots[slots[0]].ctx, std::min(n, TSIMK_HW_MAX_CTX));
      M.partial = part ? 1 : 0;
      if (part) ++p->stat_partial;
    }
    const unsigned rstr = 32u * (WF == 1 ? 2u : 4u) + 1u;
    unsigned l = 4u * TSIMK_LWF_MAX_RUNS;  // the kernel's LDS layout (same running sum there)
    for (auto &c : p->comps) {
      const unsigned l_bases = l + 8u * rstr, l_lut = (l_bases + 72u + 1u) & ~1u;
      l = l_lut + (2u << c.n_out);
    }
    const size_t ldsb = (size_t)l * 4;
    ++p->path_count[TP_LW_FASTM];
    if (WF == 1) hipLaunchKernelGGL(k_sample_lw_fastm<2>, dim3((unsigned)grid), dim3(blk1), ldsb, s, M);
    else hipLaunchKernelGGL(k_sample_lw_fastm<4>, dim3((unsigned)grid), dim3(blk1), ldsb, s, M);
    ++p->stat_fast;
  } else {
    ++p->path_count[TP_LW_MULTI];
    if (WF == 1) hipLaunchKernelGGL(k_sample_lw_multi<2>, dim3((unsigned)grid), dim3(blk1), 0, s, M);
    else hipLaunchKernelGGL(k_sample_lw_multi<4>, dim3((unsigned)grid), dim3(blk1), 0, s, M);
  }
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
    sl.partial = part;
    p->deferred.push_back(slots[j]);
  }
  p->stat_begins += (unsigned long long)n;
  p->stat_deferred += (unsigned long long)n;
  ++p->stat_fused;
  if ((long long)n * B <= p->knobs.hard_inline_rows) {
    // (letting the hard rows wait for the lane's NEXT first pass was tried - profiles/r03/hard_lag_experiment.txt: slower)
    p->flush_inline = s;
  }
  const int rf = flush_chunks(p);
  TSIM_MARK("hard");
  return rf;
}

// Programs without components (Clifford-only circuits): up to TSIMK_DIRECT_MAX_STEPS batches as one streaming grid
// (tsim_direct.hip.h), lanes alternating between groups.  The key is split once per batch all the same - the reference
// does (sampler.py:399), and the caller's key state must not depend on what the program contains.
static int steps_group_direct(tsim_program *p, int n, const uint64_t *const *d_f, int64_t B, int32_t num_f, uint32_t key[2],
                              void *const *d_out, uint32_t flags) {
  const bool packed = (flags & TSIM_PIPE_OUT_BIT_PACKED) != 0;
  if (!p->deferred.empty())
    if (int r = tsim_flush_hard(p)) return r;
  hipStream_t s = p->slots[1 + (int)(p->steps_groups++ & 1ull)].side;
  if (!(flags & TSIM_PIPE_INPUTS_READY) && p->stream != s) {
    if (!p->sync_ev) HIP_TRY(hipEventCreateWithFlags(&p->sync_ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(p->sync_ev, p->stream));
    HIP_TRY(hipStreamWaitEvent(s, p->sync_ev, 0));
  }
  const int WF = std::max(1, (num_f + 63) / 64);
  const long long bps = (B + 256 * TSIMK_DIRECT_RPT - 1) / (256 * TSIMK_DIRECT_RPT);
  if (bps * n > 0x7FFFFFFFll) return tsim_fail(TSIM_ENOTSUP, "batch too large");
  DirectMultiArgs M{};
  M.img = p->d_img;
  M.B = B;
  M.n_steps = n;
  M.blocks_per_step = (int)bps;
  M.prog = p->lw_direct_prog;
  M.chunks = p->lw_direct_chunks;
  M.WO = (p->num_outputs + 63) / 64;
  M.out_rb = (p->num_outputs + 7) / 8;
  int first = 0;
  for (int j = 0; j < n; ++j) {
    const int sidx = 1 + (int)(p->steps_slot++ % (unsigned long long)TSIM_PIPELINE_SLOTS);
    if (j == 0) first = sidx;
    if (int r = slot_prepare(p, sidx, 0)) return r;
    tsim_program::Slot &sl = p->slots[sidx];
    if (sl.deferred) return tsim_fail(TSIM_ESTATE, "pipeline slot %d still holds a parked launch", sidx - 1);
    if (int r = slot_order_after_previous(p, sl, s)) return r;
    uint32_t o[4];
    tsim_key_split(key[0], key[1], o);
    key[0] = o[0];
    key[1] = o[1];
    M.step[j].f = reinterpret_cast<const uint32_t *>(d_f[j]);
    uint8_t *series = sl.compact_out;  // (tsim_pipeline_set_compact_output: the slot's next launch also writes bit_packed rows there)
    M.step[j].out = packed ? nullptr : (uint64_t *)d_out[j];
    M.step[j].out_compact = packed ? (uint8_t *)d_out[j] : series;
    sl.compact_out = nullptr;
  }
  ++p->path_count[TP_DIRECT_MULTI];
  if (WF == 1) hipLaunchKernelGGL(k_direct_multi<2>, dim3((unsigned)(bps * n)), dim3(256), 0, s, M);
  else hipLaunchKernelGGL(k_direct_multi<4>, dim3((unsigned)(bps * n)), dim3(256), 0, s, M);
  HIP_TRY(hipGetLastError());
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
extern "C" int tsim_sample_steps_device(tsim_program *p, int32_t n_steps, const uint64_t *const *d_f, int64_t B, int32_t num_f,
                                        uint32_t key[2], int64_t shot_offset, void *const *d_out, float *const *d_max_norm_dev,
                                        uint32_t flags) {
  if (int r = tsim_need_final(p)) return r;
  if (int r = tsim_set_device(p)) return r;
  if (n_steps < 0 || B < 0 || num_f < 0 || shot_offset < 0) return tsim_fail(TSIM_EINVAL, "negative n_steps/B/num_f/shot_offset");
  if (!key || (n_steps > 0 && (!d_f || !d_out))) return tsim_fail(TSIM_EINVAL, "NULL argument");
  if (!p->sampleable) return tsim_fail(TSIM_ESTATE, "program has joint-mode components (evaluate-only)");
  if (p->max_f_index >= num_f) return tsim_fail(TSIM_EINVAL, "program references f index %d but num_f=%d", p->max_f_index, num_f);
  for (int j = 0; j < n_steps; ++j)
    if ((!d_f[j] && num_f > 0) || !d_out[j]) return tsim_fail(TSIM_EINVAL, "buffer %d is NULL", j);
  const int WF = std::max(1, (num_f + 63) / 64), WO = (p->num_outputs + 63) / 64;
  HostMarks marks;
  g_marks = marks.on ? &marks : nullptr;
  TSIM_MARK("entry");
  int done = 0;
  while (done < n_steps) {
    if (g_noise_req) g_noise_req->base = done;
    // (every path but the register first passes' fused groups: the batches' noise on the handle's stream, the lanes wait for it)
    auto noise_first = [&](int n) -> int {
      TsimNoiseRequest *nq = g_noise_req;
      if (!nq) return 0;
      // (the f buffer of a batch is its pipeline slot's as far as this call can tell: the handle's stream first gets behind the
      // launches those slots carried last - their hard rows may still read the rows the noise is about to overwrite)
      for (int j = 0; j < n; ++j) {
        tsim_program::Slot &sl = p->slots[1 + (int)((p->steps_slot + (unsigned long long)j) % (unsigned long long)TSIM_PIPELINE_SLOTS)];
        if (sl.deferred)
          if (int r = tsim_flush_hard(p)) return r;
        if (int r = slot_order_after_previous(p, sl, p->stream)) return r;
      }
      for (int j = 0; j < n; ++j)
        if (int r = nq->launch(nq->noise, B, nq->keys[2 * (done + j)], nq->keys[2 * (done + j) + 1], const_cast<uint64_t *>(d_f[done + j]), p->stream)) return r;
      flags &= ~(uint32_t)TSIM_PIPE_INPUTS_READY;
      return 0;
    };
    // the fused first pass applies when the register form does, the subkeys fit its records, and the launch plan
    // says "tables, short lists": decided per group - the plan follows the feedback of earlier launches
    const bool reg_fused = p->lw && !p->lw_wide && p->lw_reg && (WF == 1 || WF == 2) && WO == 1 && p->total_keys > 0 &&
                           p->total_keys <= TSIMK_LWM_KEYS;
    // ... and every other narrow program through k_sample_gen (rows in LDS: any row width, up to 256 outputs)
    const bool gen_fused = gen_applies(p, B, num_f, shot_offset) && (p->knobs.gen == 2 || !reg_fused);
    bool fused = (reg_fused || gen_fused) && p->knobs.fused_steps && p->num_outputs > 0 && B > 0 && B <= 0x7FFFFFFFll && p->knobs.defer_group >= 1;
    // no components at all: the streaming kernel for direct outputs (rows of at most 128 f bits and 128 outputs)
    if (p->comps.empty() && p->knobs.fused_steps && p->num_outputs > 0 && p->num_outputs <= 128 && WF <= 2 && p->lw_direct_chunks > 0 &&
        B > 0 && B <= 0x7FFFFFFFll) {
      if (!p->slots_ready) {
        for (int k = 1; k <= 4; ++k)
          if (int r = slot_prepare(p, k, 0)) return r;
        if (int r = slots_now_ready(p)) return r;
      }
      const int left = n_steps - done;
      const int groups = (left + TSIMK_DIRECT_MAX_STEPS - 1) / TSIMK_DIRECT_MAX_STEPS;
      const int n = (left + groups - 1) / groups;
      if (int r = noise_first(n)) return r;
      if (int r = steps_group_direct(p, n, d_f + done, B, num_f, key, d_out + done, flags)) return r;
      done += n;
      continue;
    }
    // wide components: groups of batches through k_sample_wide, one pass per component
    LaunchPlan wide_plan;
    bool have_wide_plan = false;
    if (p->knobs.fused_steps && wide_applies(p, B, num_f, shot_offset) && p->series_left == 0) {
      const int left = n_steps - done;
      const int gmax = std::min(TSIMK_LWM_MAX_STEPS, p->knobs.fused_max);
      const int groups = (left + gmax - 1) / gmax;
      const int n = (left + groups - 1) / groups;
      bool okb = true;
      if (flags & TSIM_PIPE_OUT_BIT_PACKED) {
        okb = true;  // (rows of any size and alignment since round 5: tsim_wide.hip.h oc_put)
      } else {
        for (int j = 0; j < n; ++j) okb = okb && !p->slots[1 + (int)((p->steps_slot + (unsigned long long)j) % (unsigned long long)TSIM_PIPELINE_SLOTS)].compact_out;
      }
      if (okb) {
        if (!p->slots_ready) {
          for (int k = 1; k <= 4; ++k)
            if (int r = slot_prepare(p, k, 0)) return r;
          if (int r = slots_now_ready(p)) return r;
        }
        const LaunchPlan wplan = make_plan(p, false, true, (unsigned long long)n * (unsigned long long)B);
        TSIM_MARK("plan");
        // (make_plan may have swapped deeper tables in: the 32-bit offsets of k_sample_wide are checked against THOSE - ADVICE r04)
        if (wplan.use_tables && wide_applies(p, B, num_f, shot_offset)) {
          if (int r = noise_first(n)) return r;
          if (int r = steps_group_wide(p, n, d_f + done, B, num_f, key, shot_offset, d_out + done, d_max_norm_dev ? d_max_norm_dev + done : nullptr, flags))
            return r;
          done += n;
          continue;
        }
        wide_plan = wplan;
        have_wide_plan = true;
      }
    }
    LaunchPlan plan;
    bool have_plan = false;
    if (have_wide_plan) {
      plan = wide_plan;
      have_plan = true;
    }
    if (fused) {
      if (!p->slots_ready) {  // as in _begin: every slot's stream / buffers now, not mid-run
        for (int k = 1; k <= 4; ++k)
          if (int r = slot_prepare(p, k, 0)) return r;
        if (int r = slots_now_ready(p)) return r;
      }
      plan = make_plan(p, false, true, (unsigned long long)std::min(n_steps - done, std::min(TSIMK_LWM_MAX_STEPS, p->knobs.fused_max)) * (unsigned long long)B);
      hard_geometry(p, WF, WO);
      TSIM_MARK("plan");
      have_plan = true;
      // (narrow_big: the one-batch path runs no table pass - f_sel beyond 64 bits - so it never reports the list lengths `defer`
      // waits for: fused whenever the plan says "tables"; long lists are the hard-row grid's worker blocks' business)
      // (prefix-tree tables: likewise - and without chunk tables, p->v4, the plan never defers)
      fused = (plan.defer || p->narrow_big || p->lw_trie) && plan.use_tables && p->h_group_tiles >= 1;
    }
    if (fused) {
      // even groups of at most TSIMK_LWM_MAX_STEPS batches
      const int left = n_steps - done;
      // (k_sample_gen carries batches x compiled outputs subkey records per launch: 8 batches up to 40 outputs, fewer beyond)
      const int gmax = std::min(gen_fused ? std::min(TSIMK_GEN_MAX_STEPS, std::max(1, TSIMK_GEN_KEYS / std::max(1, p->total_keys))) : TSIMK_LWM_MAX_STEPS, p->knobs.fused_max);
      // (20 batches as 7+7+6; 5+5+5+5 - both lanes ending together - measured slower: a launch more)
      const int groups = (left + gmax - 1) / gmax;
      const int n = (left + groups - 1) / groups;
      // (group sizes that fill whole chip-fulls of first-pass blocks - 15 batches of 10^5 shots instead of 8 - make the first
      // pass cheaper per batch and the step slower: two hard-row grids per group, C4 at 10^5 shots 1.88 -> 1.53e10, C2 3.6 -> 2.9e10)
      {
        const size_t hard_bytes = (size_t)(((B + 255) / 256 + TSIMK_LW_LISTS - 1) / TSIMK_LW_LISTS * 1024) * TSIMK_LW_LISTS * 4;
        for (int j = 0; j < n; ++j)
          if (int r = slot_prepare(p, 1 + (int)((p->steps_slot + (unsigned long long)j) % TSIM_PIPELINE_SLOTS), hard_bytes)) return r;
      }
      if (gen_fused) {
        if (int r = noise_first(n)) return r;
        if (int r = steps_group_gen(p, n, d_f + done, B, num_f, key, shot_offset, d_out + done, d_max_norm_dev ? d_max_norm_dev + done : nullptr,
                                    flags, plan))
          return r;
      } else if (int r = steps_group_fused(p, n, d_f + done, B, num_f, key, shot_offset, d_out + done,
                                           d_max_norm_dev ? d_max_norm_dev + done : nullptr, flags, plan))
        return r;
      done += n;
    } else {
      const int slot = (int)(p->steps_slot++ % (unsigned long long)TSIM_PIPELINE_SLOTS);
      uint32_t o[4];
      tsim_key_split(key[0], key[1], o);
      key[0] = o[0];
      key[1] = o[1];
      g_carry_plan = have_plan ? &plan : nullptr;
      if (int r = noise_first(1)) return r;
      if (int r = tsim_sample_batch_device_begin(p, slot, d_f[done], B, num_f, o[2], o[3], shot_offset, (uint64_t *)d_out[done],
                                                 d_max_norm_dev ? d_max_norm_dev[done] : nullptr, nullptr, flags))
        return r;
      ++done;
    }
  }
  // the groups whose hard rows were waiting for their lane's next first pass: there is none in this call
  const int rfl = tsim_flush_hard(p);
  TSIM_MARK("end");
  marks.print();
  g_marks = nullptr;
  p->first_call_out.store(true, std::memory_order_release);
  return rfl;
}

extern "C" int tsim_pipeline_lane_stream(tsim_program *p, int32_t lane, void **stream) {
  if (int r = tsim_need_final(p)) return r;
  if (int r = tsim_set_device(p)) return r;
  if (lane < 0 || lane >= TSIM_PIPELINE_SLOTS || !stream) return tsim_fail(TSIM_EINVAL, "bad lane %d", lane);
  if (int r = slot_prepare(p, 1 + lane, 0, true)) return r;
  *stream = (void *)p->slots[1 + lane].side;
  return TSIM_OK;
}

extern "C" int tsim_pipeline_wait_stream(tsim_program *p, void *stream) {
  if (int r = tsim_need_final(p)) return r;
  if (int r = tsim_set_device(p)) return r;
  hipStream_t s_user = stream ? (hipStream_t)stream : p->stream;
  if (!p->slots_ready) {
    // No lane exists yet.  Rounds 2-4 returned here ("the first launches order themselves") - they do not when the caller then
    // launches with TSIM_PIPE_INPUTS_READY, as sampler.py does: the first group of a fresh handle raced with the noise kernel that
    // fills its f buffers (hidden while creating the 32 slot streams took 80 ms; seen once the streams came from the pool:
    // tests/test_gpu_sampler.py::test_device_postselection[rows] behind test_gpu_noise.py).  Create the lanes and order them.
    for (int k = 1; k <= 4; ++k)
      if (int r = slot_prepare(p, k, 0)) return r;
    if (int r = slots_now_ready(p)) return r;
  }
  if (!p->sync_ev) HIP_TRY(hipEventCreateWithFlags(&p->sync_ev, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(p->sync_ev, s_user));
  // every lane a launch may run on: the slots' own streams that were used so far and the first three
  // (first passes / hard-row batches of the deferred plan)
  std::vector<hipStream_t> seen;
  for (int k = 1; k <= TSIM_PIPELINE_SLOTS; ++k) {
    tsim_program::Slot &sl = p->slots[k];
    if (!sl.side_ready) {  // (a slot whose own stream does not exist yet: created by its first launch there, which then waits)
      if (k > 4) sl.needs_sync = true;
      continue;
    }
    if (sl.side == s_user) continue;
    if (k > 4 && !sl.used) {  // never carried work: not worth a wait packet now - its first launch takes the wait (launch on a slot's own stream)
      sl.needs_sync = true;
      continue;
    }
    if (std::find(seen.begin(), seen.end(), sl.side) != seen.end()) continue;
    seen.push_back(sl.side);
    HIP_TRY(hipStreamWaitEvent(sl.side, p->sync_ev, 0));
  }
  return TSIM_OK;
}

extern "C" int tsim_pipeline_set_compact_series(tsim_program *p, uint8_t *d_base, int64_t stride_bytes, int32_t count) {
  if (int r = tsim_need_final(p)) return r;
  if (count < 0 || stride_bytes < 0 || (count > 0 && !d_base)) return tsim_fail(TSIM_EINVAL, "bad compact series");
  p->series_ptr = d_base;
  p->series_stride = stride_bytes;
  p->series_left = count;
  return TSIM_OK;
}

extern "C" int tsim_pipeline_set_compact_output(tsim_program *p, int32_t slot, uint8_t *d_compact) {
  if (int r = tsim_need_final(p)) return r;
  if (slot < 0 || slot >= TSIM_PIPELINE_SLOTS) return tsim_fail(TSIM_EINVAL, "slot %d out of range", slot);
  p->slots[1 + slot].compact_out = d_compact;
  return TSIM_OK;
}

extern "C" int tsim_sample_batch_device_compact(tsim_program *p, int32_t slot, const uint64_t *d_rows, int64_t B,
                                                int32_t nbits, uint8_t *d_out, void *stream) {
  if (int r = tsim_need_final(p)) return r;
  if (int r = tsim_set_device(p)) return r;
  if (slot < 0 || slot >= TSIM_PIPELINE_SLOTS) return tsim_fail(TSIM_EINVAL, "slot %d out of range", slot);
  if (B < 0 || nbits < 0) return tsim_fail(TSIM_EINVAL, "negative size");
  if (B == 0 || nbits == 0) return TSIM_OK;
  if (!d_rows || !d_out) return tsim_fail(TSIM_EINVAL, "NULL buffer");
  tsim_program::Slot &sl = p->slots[1 + slot];
  if (sl.deferred)
    if (int r = tsim_flush_hard(p)) return r;
  // after the slot's second pass when there is one in flight, else simply on the caller's stream
  hipStream_t s = sl.pending ? sl.last_done : (stream ? (hipStream_t)stream : p->stream);
  if (int r = tsim_launch_compact(d_rows, B, (nbits + 63) / 64, nbits, d_out, s)) return r;
  if (sl.pending) {
    HIP_TRY(hipEventRecord(sl.ev2, s));
    sl.done_ev = sl.ev2;
    sl.batch_seq = 0;
  }
  return TSIM_OK;
}

extern "C" int tsim_sample_batch_device_end(tsim_program *p, int32_t slot, void *stream) {
  if (int r = tsim_need_final(p)) return r;
  if (int r = tsim_set_device(p)) return r;
  if (slot < 0 || slot >= TSIM_PIPELINE_SLOTS) return tsim_fail(TSIM_EINVAL, "slot %d out of range", slot);
  hipStream_t s = stream ? (hipStream_t)stream : p->stream;
  tsim_program::Slot &sl = p->slots[1 + slot];
  if (sl.deferred)  // its batch is not full yet: run what is waiting now
    if (int r = tsim_flush_hard(p)) return r;
  if (sl.pending) {
    if (sl.last_done != s) {
      // Batches complete in order on their lane: a stream that already waits for batch b is behind every batch <= b.
      // "End every slot" - the usual way to join a pipeline - is then ONE stream wait, not one per slot (each is a
      // barrier packet on `s`; when `s` is the handle's stream, i.e. first-pass lane 0, fourteen of them behind its
      // last kernel were ~60 us of a 20-launch burst).
      // (only for streams the handle owns - a caller's stream handle may be destroyed and its address reused)
      bool own = s == p->stream;
      for (int k = 1; k <= 4 && !own; ++k) own = p->slots[k].side_ready && s == p->slots[k].side;
      // (per batch stream: batches of ONE stream complete in order, inline batches run on the first-pass lanes)
      const bool in_order = own && sl.batch_seq != 0;
      const int bl = sl.batch_lane;
      if (in_order && s == p->joined_stream && sl.batch_seq <= p->joined_seq[bl]) {
        // nothing to add
      } else {
        HIP_TRY(hipStreamWaitEvent(s, sl.done_ev, 0));
        if (in_order) {
          if (s != p->joined_stream) {
            p->joined_stream = s;
            for (auto &q : p->joined_seq) q = 0;
          }
          p->joined_seq[bl] = std::max(p->joined_seq[bl], sl.batch_seq);
        }
      }
    }
    sl.pending = false;
  }
  return TSIM_OK;
}

// `stream` waits for the launch the slot carried last - also when another stream already joined it (_end cleared
// `pending`): the guard of a producer that refills the slot's f buffer while a consumer stream owns its output.
extern "C" int tsim_pipeline_wait_slot(tsim_program *p, int32_t slot, void *stream) {
  if (int r = tsim_need_final(p)) return r;
  if (int r = tsim_set_device(p)) return r;
  if (slot < 0 || slot >= TSIM_PIPELINE_SLOTS) return tsim_fail(TSIM_EINVAL, "slot %d out of range", slot);
  hipStream_t s = stream ? (hipStream_t)stream : p->stream;
  tsim_program::Slot &sl = p->slots[1 + slot];
  if (sl.deferred)  // its hard rows are still parked: run what is waiting now
    if (int r = tsim_flush_hard(p)) return r;
  if (sl.done_ev && sl.last_done && sl.last_done != s) HIP_TRY(hipStreamWaitEvent(s, sl.done_ev, 0));
  return TSIM_OK;
}

// tsim_sample_batch_device_end for EVERY slot in one call (joining a pipeline on `stream`: one library call instead of
// TSIM_PIPELINE_SLOTS - 32 ctypes round trips per gather group were ~60 us of host time in bench.py's N > 1 path)
extern "C" int tsim_pipeline_join(tsim_program *p, void *stream) {
  if (int r = tsim_need_final(p)) return r;
  for (int slot = 0; slot < TSIM_PIPELINE_SLOTS; ++slot)
    if (p->slots[1 + slot].pending || p->slots[1 + slot].deferred)
      if (int r = tsim_sample_batch_device_end(p, slot, stream)) return r;
  return TSIM_OK;
}

extern "C" int tsim_sample_batch_device(tsim_program *p, const uint64_t *d_f, int64_t B, int32_t num_f,
                                        uint32_t key_hi, uint32_t key_lo, int64_t shot_offset,
                                        uint64_t *d_out, float *d_max_norm_dev, void *stream) {
  if (int r = tsim_need_final(p)) return r;
  if (int r = tsim_set_device(p)) return r;
  hipStream_t s = stream ? (hipStream_t)stream : p->stream;
  return launch_sample(p, d_f, B, num_f, key_hi, key_lo, shot_offset, d_out, d_max_norm_dev, s);
}
extern "C" int tsim_sample_rows_device(tsim_program *p, const uint64_t *d_f, int64_t B, int32_t num_f,
                                       uint32_t key_hi, uint32_t key_lo, int64_t shot_offset, uint64_t *d_out,
                                       float *d_max_norm_dev, const uint32_t *d_row_index,
                                       const uint32_t *d_row_count, void *stream) {
  if (int r = tsim_need_final(p)) return r;
  if (int r = tsim_set_device(p)) return r;
  if (!d_row_index || !d_row_count) return tsim_fail(TSIM_EINVAL, "row list is NULL");
  if (B >= (1ll << 32)) return tsim_fail(TSIM_ENOTSUP, "row indices are 32-bit");
  hipStream_t s = stream ? (hipStream_t)stream : p->stream;
  return launch_sample(p, d_f, B, num_f, key_hi, key_lo, shot_offset, d_out, d_max_norm_dev, s, d_row_index,
                       d_row_count);
}

extern "C" int tsim_postselect_device(tsim_program *p, const uint64_t *d_f, int64_t B, int32_t num_f,
                                      const uint64_t *d_mask, const uint64_t *d_ref, uint64_t *d_out,
                                      uint32_t *d_row_index, uint32_t *d_row_count, uint8_t *d_discarded,
                                      void *stream) {
  if (int r = tsim_need_final(p)) return r;
  if (int r = tsim_set_device(p)) return r;
  if (B < 0 || num_f < 0) return tsim_fail(TSIM_EINVAL, "negative size");
  if (B >= (1ll << 32)) return tsim_fail(TSIM_ENOTSUP, "row indices are 32-bit");
  if (p->max_f_index >= num_f) return tsim_fail(TSIM_EINVAL, "program references f index %d but num_f=%d", p->max_f_index, num_f);
  if (!d_mask || !d_out || !d_row_index || !d_row_count) return tsim_fail(TSIM_EINVAL, "NULL buffer");
  hipStream_t s = stream ? (hipStream_t)stream : p->stream;
  HIP_TRY(hipMemsetAsync(d_row_count, 0, 4, s));
  if (B == 0 || p->num_outputs == 0) return TSIM_OK;
  FilterArgs a;
  a.img = p->d_img;
  a.f = d_f;
  a.out = d_out;
  a.mask = d_mask;
  a.ref = d_ref;
  a.row_index = d_row_index;
  a.row_count = d_row_count;
  a.discarded = d_discarded;
  a.B = B;
  a.WF = num_f == 0 ? 0 : std::max(1, (num_f + 63) / 64);
  a.WO = (p->num_outputs + 63) / 64;
  a.n_direct = p->n_direct;
  a.direct_off = p->direct_off;
  int block = 256;
  size_t lds = (size_t)(2 * a.WF + 2 * a.WO) * block * 4;
  if (lds > 60 * 1024) { block = 64; lds = (size_t)(2 * a.WF + 2 * a.WO) * block * 4; }
  if (lds > 60 * 1024) return tsim_fail(TSIM_ENOTSUP, "num_f + num_outputs too large for LDS staging (%zu B)", lds);
  hipLaunchKernelGGL(k_direct_filter, dim3((unsigned)((B + block - 1) / block)), dim3(block), lds, s, a);
  HIP_TRY(hipGetLastError());
  return TSIM_OK;
}
extern "C" int tsim_sample_batch(tsim_program *p, const uint8_t *f, int64_t B, int32_t num_f, uint32_t key_hi,
                                 uint32_t key_lo, int64_t shot_offset, uint8_t *out, int32_t out_packed,
                                 float *max_norm_dev) {
  if (int r = tsim_need_final(p)) return r;
  if (int r = tsim_set_device(p)) return r;
  if (B < 0 || num_f < 0) return tsim_fail(TSIM_EINVAL, "negative B/num_f");
  if (B == 0 || p->num_outputs == 0) return 0;
  if (!out) return tsim_fail(TSIM_EINVAL, "out is NULL");
  if (!f && num_f > 0) return tsim_fail(TSIM_EINVAL, "f is NULL");
  const int WF = std::max(1, (num_f + 63) / 64), WO = (p->num_outputs + 63) / 64;
  hipStream_t s = p->stream;
  if (int r = tsim_ensure_scratch(p, 0, (size_t)B * std::max(1, num_f))) return r;
  if (int r = tsim_ensure_scratch(p, 1, (size_t)B * WF * 8)) return r;
  if (int r = tsim_ensure_scratch(p, 2, (size_t)B * WO * 8)) return r;
  if (num_f > 0) {
    HIP_TRY(hipMemcpyAsync(p->scratch[0], f, (size_t)B * num_f, hipMemcpyHostToDevice, s));
    if (int r = tsim_launch_pack(p, (const uint8_t *)p->scratch[0], B, num_f, (uint64_t *)p->scratch[1], s)) return r;
  }
  if (int r = launch_sample(p, (const uint64_t *)p->scratch[1], B, num_f, key_hi, key_lo, shot_offset,
                            (uint64_t *)p->scratch[2], p->d_dev, s))
    return r;
  if (out_packed) {
    HIP_TRY(hipMemcpyAsync(out, p->scratch[2], (size_t)B * WO * 8, hipMemcpyDeviceToHost, s));
  } else {
    if (int r = tsim_ensure_scratch(p, 3, (size_t)B * p->num_outputs)) return r;
    if (int r = tsim_launch_unpack(p, (const uint64_t *)p->scratch[2], B, p->num_outputs, (uint8_t *)p->scratch[3], s)) return r;
    HIP_TRY(hipMemcpyAsync(out, p->scratch[3], (size_t)B * p->num_outputs, hipMemcpyDeviceToHost, s));
  }
  if (max_norm_dev && shot_offset == 0 && !p->comps.empty())
    HIP_TRY(hipMemcpyAsync(max_norm_dev, p->d_dev, p->comps.size() * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return TSIM_OK;
}
#ifdef TSIMK_LW_TRACE
// diagnostic builds only (scripts/lw_trace.py): the s_memtime stamps of the first pass's first blocks
extern "C" int tsim_debug_lw_trace(unsigned long long *out) {
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(tsimk::tsimk_lw_trace), 16 * 32 * 8));
  return 0;
}
#endif

#ifdef TSIMK_WIDE_TRACE
// diagnostic builds only (scripts/wide_trace.py): the phase timers of k_sample_wide, read and reset
extern "C" int tsim_debug_wide_trace(unsigned long long *out) {
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(tsimk::tsimk_wide_trace), 24 * 8));
  unsigned long long z[24] = {};
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(tsimk::tsimk_wide_trace), z, 24 * 8));
  return 0;
}
#endif

extern "C" int tsim_profile_enable(tsim_program *p, int32_t on) {
  if (int r = tsim_need_final(p)) return r;
  if (int r = tsim_set_device(p)) return r;
  if (!on && p->ev_used) { if (int r = prof_drain(p)) return r; }
  p->profiling = on != 0;
  p->prof_light = on == 2;
  p->prof_counter = 0;
  return TSIM_OK;
}

extern "C" int tsim_profile_set_sampling(tsim_program *p, int32_t every) {
  if (!p || every < 1) return tsim_fail(TSIM_EINVAL, "bad argument");
  p->prof_every = every;
  p->prof_counter = 0;
  return TSIM_OK;
}

extern "C" int tsim_profile_read(tsim_program *p, double *kernel_ms, int64_t *launches, int32_t reset) {
  if (int r = tsim_need_final(p)) return r;
  if (int r = tsim_set_device(p)) return r;
  if (int r = prof_drain(p)) return r;
  if (kernel_ms) *kernel_ms = p->prof_ms;
  if (launches) *launches = p->prof_launches;
  if (reset) {
    p->prof_ms = 0.0;
    p->prof_launches = 0;
    for (double &v : p->prof_stage_ms) v = 0.0;
  }
  return TSIM_OK;
}

extern "C" int tsim_profile_read_steps(tsim_program *p, int64_t *steps, int32_t reset) {
  if (int r = tsim_need_final(p)) return r;
  if (!steps) return tsim_fail(TSIM_EINVAL, "NULL argument");
  *steps = p->prof_steps;
  if (reset) p->prof_steps = 0;
  return TSIM_OK;
}

extern "C" int tsim_profile_read_stages(tsim_program *p, double stage_ms[3]) {
  if (int r = tsim_need_final(p)) return r;
  if (int r = tsim_set_device(p)) return r;
  if (!stage_ms) return tsim_fail(TSIM_EINVAL, "NULL argument");
  if (int r = prof_drain(p)) return r;
  stage_ms[0] = p->prof_stage_ms[PROF_PASS1];
  stage_ms[1] = p->prof_stage_ms[PROF_HARD];
  stage_ms[2] = p->prof_stage_ms[PROF_FULL];
  return TSIM_OK;
}
