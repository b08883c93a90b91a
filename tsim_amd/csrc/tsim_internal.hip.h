// tsim_internal.hip.h - host-side state shared by the translation units of libtsim_hip.so.
//
//   tsim_program.hip   handle life cycle: description -> packed image -> upload; memory/stream plumbing
//   tsim_pack.hip      the packer: reference-layout rows -> bit-packed rows, pack-time GF(2) algebra,
//                      chunk tables (host only)
//   tsim_sample.hip    launch planner + pipeline scheduler + the sampling kernels (k_sample_lw, k_sample4,
//                      k_sample4h), device-side post-selection, HIP-event profiling
//   tsim_rows*.hip     the row-formulation kernels (k_sample<W>, k_evaluate<W>, k_lw_nodes<W>), one TU per
//                      formulation so that they compile in parallel
//   tsim_format.hip    byte-per-bit <-> packed rows, bit_packed compaction, row gather/scatter
//   tsim_noise.hip     device-side channel sampler
//   tsim_pcg.cpp       numpy-stream-exact host channel sampler (PCG64 + ziggurat), no HIP
//   tsim_dist.hip      RCCL communicator (shot sharding over the GPUs of a node)
#pragma once
#include "../../include/tsim_hip.h"
#include "tsim_kernels.hip.h"
#include "tsim_kernel4.hip.h"
#ifndef TSIM_HARD_NW
#define TSIM_HARD_NW 8   // waves per 64-row group in k_sample4h
#endif
#include "tsim_lw.hip.h"

#include <algorithm>
#include <array>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <atomic>
#include <mutex>
#include <chrono>
#include <cstring>
#include <functional>
#include <new>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

#define TSIMK_H_MAX_CTX 8   // launches served by one deferred hard-row batch (k_sample4h_multi)
// Launch constants that were A/B switches until round 4 / 5 (the experiments are closed: HISTORY.md, profiles/r03 .. r05)
constexpr int kListRows = 40;        // expected rows per hard-row sub-list
constexpr int kMinLists = 4;
constexpr int kHardLdsKb = 128;      // k_sample4h: a first-pass block still fits next to a hard-row block
constexpr int kV4Block = 256;        // k_sample4 block
constexpr int kWideGlobGraphs = 48;  // components of up to this many graphs whose column tables do not fit the LDS run k_sample_wide with the tables in the L2

// ---------------------------------------------------------------------------
// errors: every entry point returns 0 or a negative TSIM_E* code; the message is thread-local
// ---------------------------------------------------------------------------
int tsim_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define HIP_TRY(expr)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) return tsim_fail(TSIM_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

// fn(0) .. fn(n - 1) on the process-wide worker pool (tsim_pool.cpp): at most max_threads threads work on the call, the caller
// among them; bounded however many components a program has, and safe to nest (a thread per level, created
// before any join, could throw EAGAIN inside an extern "C" function).
void tsim_parallel_for_impl(size_t n, int max_threads, const std::function<void(size_t)> &fn);
template <class F>
inline void tsim_parallel_for(size_t n, int max_threads, F &&fn) {
  tsim_parallel_for_impl(n, max_threads, std::function<void(size_t)>(std::forward<F>(fn)));
}

// ---------------------------------------------------------------------------
// host-side program
// ---------------------------------------------------------------------------
namespace tsimhost {

// per-graph result of the fast packer's algebra, kept for the v4 (chunk table) emitter
struct FastGraph {
  std::vector<std::vector<uint64_t>> c0, c1, c3;  // counted NodePhases rows per class (masks)
  std::vector<uint8_t> c0c, c1c, c3c;             // their constants
  std::vector<std::vector<uint64_t>> dal, dbe;    // PhasePairs alpha / beta masks
  std::vector<uint64_t> lam, lin;
  std::vector<std::vector<uint64_t>> us, vs;      // Dickson product pairs
  int n1 = 0, nD = 0;
  bool d_tabled = false;
};

struct HostLevel {
  std::vector<FastGraph> fg;
  bool fixed = false;
  bool sum_wrap_possible = false;  // the reference's int32 running sum of this level cannot be ruled out to wrap (pack_level_fast)
  int frame = 0;
  int G = 0, P = 0;
  bool approx = false;
  // per graph
  std::vector<uint32_t> graph_rec;        // G * G_WORDS (row offsets relative to `rows`)
  std::vector<uint32_t> rows;             // packed rows, built for word count W
  long long n_rows = 0;
  uint32_t tt_off = 0, tt_words = 0;      // fast layout: the level's term tables in the image (one contiguous block)
  // raw copy of the description (packing happens at finalize when W is known)
  tsim_level_desc d{};
  std::vector<uint8_t> u8[14];
  std::vector<int32_t> i32[4];
  std::vector<float> approx_v;
};

struct HostComponent {
  int n_out = 0, F = 0, n_levels = 0;
  std::vector<int32_t> output_indices, f_selection;
  std::vector<HostLevel> levels;
  bool trie = false;  // pattern tables as a chunked prefix tree (tsim_trie.hip.h): components of more than TSIMK_LW_MAX_NOUT outputs
};

static const int kWVariants[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64};

inline int round_w(int w) {
  for (int v : kWVariants)
    if (w <= v) return v;
  return -1;
}

}  // namespace tsimhost

using tsimk::SampleArgs;
using tsimhost::FastGraph;
using tsimhost::HostComponent;
using tsimhost::HostLevel;

struct TsimTablePlan {  // pattern tables of one depth: per component the deepest weight and the patterns, bytes of all
  std::vector<int> wmax;
  std::vector<long long> npat;
  std::vector<long long> chunks;  // per component: chunks of its prefix tree (0: dense format)
  long long bytes = 0;
};

struct TsimBuildJob {  // the table build of one component, cut into slices of patterns
  tsimk::LwBuildArgs a;
  int W, n_out;
  long long next_pat;
};

// kernel families (tsim_program_path_counts, include/tsim_hip.h)
enum TsimPath { TP_LW_FAST = 0, TP_LW_FASTM, TP_LW_MULTI, TP_DIRECT_MULTI, TP_WIDE, TP_LW_FAST1, TP_LW_REG, TP_LW_LDS, TP_LW_LDS_WIDE,
                TP_SAMPLE4W, TP_SAMPLE4, TP_SAMPLE4H, TP_HW, TP_OVER, TP_ROWS, TP_SAMPLE4H_MULTI, TP_GEN, TP_NOISE_FAST };

struct tsim_program {
  long long path_count[TSIM_PATH_COUNT] = {};
  // description
  int num_outputs = 0, num_detectors = 0, n_direct = 0;
  std::vector<int32_t> direct_f, output_order;
  std::vector<uint8_t> direct_flips;
  std::vector<HostComponent> comps;
  bool finalized = false;
  // packed image
  std::vector<uint32_t> img;
  int direct_off = 0, comp_off = 0;
  int total_keys = 0;  // total compiled outputs (sequential components)
  bool sampleable = true;
  int mode = TSIM_MODE_AUTO;  // requested
  bool fast = false;          // chosen at finalize: counting formulation (eval_level_fast)
  bool v4 = false;            // chunk-table layout present (k_sample4)
  size_t v4w_resident_bytes = 0;  // LDS needed to hold all levels' column tables of the largest component
  bool v4w = false;           // wide components: column tables only (k_sample4w + row kernel for the overflow)
  int v4_gt = 4;              // graphs per LDS tile
  int comp4_off = 0;
  int v4_max_nch = 1;
  int v4_max_sent = 0;        // entries per tile of the sparse-f tables (0: none)
  long long total_graphs = 0, total_rows = 0;
  long long hw_max_rows = 0;  // longest per-level row stream (LDS of the wave-per-row kernel: one parity bit per row)
  long long stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // see tsim_program_stats
  int max_f_index = -1;
  std::vector<int> level_off;  // flattened [component][level] -> offset of level record
  std::vector<int> level_base; // per component index into level_off
  std::vector<int> comp_w;
  // device
  int device = -1;
  uint32_t *d_img = nullptr;
  float *d_dev = nullptr;
  hipStream_t stream = nullptr;
  std::vector<int> stream_idx;             // creation indices of the pooled streams this handle holds (tsim_stream_acquire)
  hipStream_t aux[TSIM_AUX_STREAMS] = {};  // tsim_aux_stream
  // low-weight pattern tables (tsim_lw.hip.h)
  int lw_request = -1;        // -1 default (on in TSIM_MODE_AUTO), 0 off, 1 on
  int lw_weight_cap = -1;     // -1 = TSIMK_LW_MAX_WEIGHT
  bool lw = false;            // tables built, pass 1 active
  bool lw_wide = false;       // ... for wide components: k_sample_lw<true> -> k_sample4w on its lists -> row kernel
  int lw_binom_off = 0;       // image offset of the binomial table of the register first pass
  int lw_binom_stride = 256;    // words per row of that table in its wide layout ([4][256], [8][256] saturated, or [4][512] for components beyond 255 selected bits)
  int n_cu = 256;             // compute units of the device (grid sizing)
  int v4w_occ_blocks = 1;     // blocks of the sparse-column kernel a CU holds at once with v4w_occ_lds bytes of LDS
  size_t v4w_occ_lds = 0;
  bool lw_reg = false;        // the register form of pass 1 applies (narrow f rows, ascending f_selection)
  int lwfm_off = 0;           // fast record of programs with 2..4 components of <= 8 outputs each (k_sample_lw_fastm), 0 = none
  int wr_off = 0;             // image offset of the first wide record (k_sample_wide, tsim_wide.hip.h), 0 = none
  bool lw_trie = false;       // some component's tables are a chunked prefix tree: first pass k_sample_gen only
  std::vector<long long> lw_chunks;  // per component: chunks of its prefix tree (0: dense format)
  bool narrow_big = false;    // narrow program with a component of more than 64 selected bits: first pass k_sample_gen only, tables to weight 4
  bool wide_big = false;      // wide program with f indices >= 512: the round-2 wide kernels (16 selection-mask words) must not see it
  std::vector<int> wr_offs;   // one wide record per component (the passes of k_sample_wide, in component order)
  // streams of the CALLER that carried sampling launches of this handle (the device entry points take one): what a table swap
  // must wait for besides the handle's own lanes - through an EVENT of the handle's, recorded behind every such launch: the stream
  // handle itself is never used again (the caller may have destroyed it, its address may be another stream's by then: ADVICE r05).
  // More than 16 distinct ones: caller_streams_overflow, the swap drains the device.
  struct CallerStream { hipStream_t s; hipEvent_t ev; };
  std::vector<CallerStream> caller_streams;
  bool caller_streams_overflow = false;
  int gr_off = 0;             // image offset of the gen record (any narrow program: k_sample_gen, tsim_gen.hip.h), 0 = none
  int lwf_off = 0;            // image offset of the fast record (one component of <= 8 outputs: k_sample_lw_fast), 0 = none
  // Launch slots: slot 0 serves the serial API (everything on the caller's stream); slots 1..4 serve
  // tsim_sample_batch_device_begin/_end: a slot's launches run on the slot's own stream (`side`) so
  // that it overlaps the first pass of the following launches.
  struct Slot {
    int parity = 0;               // counter set of the next launch
    uint32_t *ctl = nullptr;      // 2 counter sets (hard-row counters + check row)
    void *hard = nullptr;         // hard-row lists
    size_t hard_sz = 0;
    int parity2 = 0;              // counter set of ctl2 for its next use (ctl2 is not used by every launch)
    uint32_t *ctl2 = nullptr;     // wide programs with tables: the sparse-column pass's own lists (its overflow rows)
    void *hard2 = nullptr;
    size_t hard2_sz = 0;
    uint32_t *keys = nullptr;     // k_keygen output (programs with > TSIMK_INLINE_KEYS outputs)
    std::vector<uint32_t> host_keys;  // ... the same subkeys on the host, when fill_sample_args computed them there (k_keyput)
    hipStream_t side = nullptr;
    hipEvent_t ev1 = nullptr, ev2 = nullptr;  // input dependency, launch done
    bool pending = false;         // second pass enqueued on `side`, not yet joined
    bool side_borrowed = false;   // `side` is not owned by the slot (the handle's main stream / the null stream)
    bool side_ready = false;
    uint8_t *compact_out = nullptr;  // next launch of the slot also writes bit_packed rows here
    // deferred second pass (flush_hard): pass 1 is enqueued, the hard rows wait for the next batch
    bool deferred = false;
    bool ctx_check = false;
    bool used = false;                // a launch ran on the slot's own stream
    bool needs_sync = false;          // tsim_pipeline_wait_stream skipped this (unused) stream: its first launch waits for sync_ev
    hipStream_t p1_stream = nullptr;  // lane of that first pass
    hipStream_t last_done = nullptr;  // stream on which done_ev of the slot's last launch was recorded
    hipEvent_t done_ev = nullptr;     // ev2 (own second pass) or the event of the batch that served the slot
    unsigned long long batch_seq = 0; // sequence number of that batch (0: own second pass)
    int batch_lane = 0;               // the stream that batch ran on: 0/1 = the batch lanes, 2.. = first-pass lane 0.. itself (inline)
    bool partial = false;             // this launch's hard-row lists carry component masks and its hard rows are stored partially (LwMultiArgs.partial)
    SampleArgs ctx;                   // the hard-row kernel's arguments for that launch
  };
  Slot slots[1 + TSIM_PIPELINE_SLOTS];
  uint32_t *ctl_block = nullptr;  // the counter sets of every slot (Slot::ctl / ctl2 point into it)
  std::atomic<bool> slots_ready{false};  // (read by the table-build helper thread: it takes its stream once the lanes exist)
  std::vector<int> deferred;  // slots whose hard rows are waiting, in launch order
  hipEvent_t lane_ev[2] = {nullptr, nullptr};  // "first passes enqueued so far on lane k are done"
  hipEvent_t over_ev[2] = {nullptr, nullptr};  // long hard-row lists: the per-shot workers run on a lane of their own beside the latency kernel (flush_batch)
  hipEvent_t batch_ev[16] = {};
  int batch_ev_lane[16] = {};                  // ... and the stream (batch_lane) it was recorded on
  hipStream_t flush_inline = nullptr;          // set by a caller of tsim_flush_hard: run this batch on that first-pass lane itself
  bool inline_seen = false;                    // some batch ran inline: batches no longer complete in one global order                // ring: one event per hard-row batch
  hipEvent_t sync_ev = nullptr;                // tsim_pipeline_wait_stream
  uint8_t *series_ptr = nullptr;               // tsim_pipeline_set_compact_series
  int64_t series_stride = 0;
  int series_left = 0;
  unsigned long long batch_next = 1;       // sequence number of the next batch (event = batch_ev[seq % 16])
  unsigned long long batch_confirmed[6] = {};  // per batch lane: every batch up to this one is known to be complete
  hipStream_t joined_stream = nullptr;  // tsim_sample_batch_device_end: the stream that last joined a batch ...
  unsigned long long joined_seq[6] = {};    // ... and that batch: it is behind every batch up to this one
  int lane_reach[4] = {0, 0, 0, 0};  // batches between a lane's last start-of-batch wait and the batch it waited for (pre-wait)
  unsigned long long lane_waited[4][6] = {};  // [first-pass lane][batch lane]: newest batch already waited for
  unsigned long long stat_queries = 0, stat_waits = 0, stat_begins = 0, stat_flushes = 0, stat_deferred = 0, stat_fused = 0, stat_fast = 0, stat_partial = 0;
  unsigned long long steps_slot = 0;    // tsim_sample_steps_device: next pipeline slot of its rotation ...
  unsigned long long steps_groups = 0;  // ... and the fused groups launched so far (first-pass lanes alternate)
  int last_lists = 0;         // list count of the most recent two-pass launch (what the feedback refers to)
  int h_group_tiles = 0;      // k_sample4h geometry, fixed at the first two-pass launch
  size_t h_lds = 0;
  // launch-plan feedback (mapped pinned host memory written by k_sample4h): [0] hard rows,
  // [1] longest hard-row list, [2] rows of that launch; 0xFFFFFFFF = nothing seen yet
  volatile uint32_t *h_feedback = nullptr;
  uint32_t *d_feedback = nullptr;
  int lw_direct_left = 0;     // launches still to run on the full kernel before the next probe
  // launch-time tuning knobs, read from the environment once at finalize (experiments only)
  struct Knobs {
    // public switches
    bool adaptive = true;     // TSIM_AMD_ADAPTIVE=0 pins the default launch plan
    bool fused_steps = true;  // TSIM_AMD_FUSED_STEPS=0: tsim_sample_steps_device launches batch by batch
    int deep_tables = 0;      // TSIM_AMD_DEEP_TABLES: deeper pattern tables when the hard rows are merely too many for k_sample_hw -
                              // 0: after deep_after rows in that state, 1: at once, -1: never
    // TSIM_AMD_TUNE keys (A/B parameters of tests/ and scripts/)
    bool defer = true;        // defer_hard=0: every pipelined launch runs its own second pass
    int defer_group = 4;      // defer_group: launches per deferred hard-row batch (<= TSIMK_H_MAX_CTX)
    bool lw_fast = true;      // lw_fast=0: the generic fused pass (k_sample_lw_multi) also for one-component programs
    bool wide_fused = true;   // wide_fused=0: wide programs on the three-kernel path of round 2 (tables, k_sample4w, row kernel)
    bool hard_wave = true;    // hard_wave=0: hard-row batches on k_sample4h_multi (64 rows per block) instead of one block per row
    int hard_wave_rows = 1024;  // hard_wave_rows: ... while a batch of launches has at most this many hard rows (last feedback)
    long long hard_inline_rows = 1ll << 40;  // hard_inline_rows: fused groups of at most this many shots run their hard rows on their own lane
    bool hard_comp_par = true;  // hard_comp_par=0: the hard rows of multi-component programs one block per row (all components in turn)
    unsigned long long deep_after = 0;        // deep_after: rows in that state before the next depth is built; 0 = by the estimated build time
    int fused_lanes = 0;      // fused_lanes: first-pass lanes the fused groups rotate over (1-4; 0 = 2, 3 for small groups)
    int fused_max = 8;        // fused_max: batches per fused first pass (<= TSIMK_LWM_MAX_STEPS = 16)
    bool wide_tables = true;  // wide_tables=0: no pattern tables in front of the wide kernels
    bool hard_overflow = true; // hard_overflow=0: the latency kernels of a hard-row batch walk whole lists (no per-shot workers behind them)
    bool shallow = true;      // shallow=0: finalize builds the default table depth at once (round 4) instead of starting shallow
    bool noise_fused = true;  // noise_fused=0: tsim_sample_steps_noise_device runs k_noise_wave in front of every first pass (no k_noise_sample_fast)
    int trie = 1;             // trie=0: no prefix-tree tables - components of more than 12 outputs run without tables (round 5); 2: prefix trees for every narrow component
    int gen = 1;              // gen: k_sample_gen for fused groups - 0 never, 1 where no register first pass applies, 2 wherever it applies
    int x4 = 32;              // x4=N: components of 81..128 parameters or more than 64 selected bits AND at least N graphs take the narrow family (four words of x); 0: never
    bool x3 = true;           // x3=0: components of 65..80 parameters (F <= 64) stay on the wide path (round-4 behaviour)
    int wide_depth = 4;       // wide_depth=N: default table depth of wide components (finalize builds at most weight 3; the rest in the background)
    int wide_passes = 8;      // wide_passes=N: programs of up to N wide-path components run as N k_sample_wide passes (1: round-4 behaviour)
    bool wide_compact = true; // wide_compact=0: k_sample_wide keeps one 16-byte column table per graph even when all graphs fit one entry
  } knobs;
  bool h_attr_set = false;    // k_sample4h: large dynamic LDS enabled
  bool hm_attr_set = false;   // k_sample4h_multi: the same
  unsigned gen_attr_set = 0;  // k_sample_gen<WO32>: bit WO32
  unsigned x4_attr_set = 0;   // NCH = 32 instantiations (81..128 parameters) whose dynamic-LDS limit was raised: 1 over, 2 hw, 4 k_sample4
  unsigned wide_attr_set = 0; // k_sample_wide<WO32, K>: bit WO32
  int lw_off = 0;             // image offset of the LW component records
  int lw_direct_prog = 0;     // image offset of the direct-output gather program
  int lw_direct_chunks = 0;
  int lw_direct_rot = 0;      // the same moves in rotate-and-mask form (register first pass), 0 = none
  std::vector<int> lw_wmax;   // per component
  std::vector<long long> lw_npat;  // per component: tabulated patterns
  int lw_cap_default = 0;     // the depth a handle reaches on its own soon after finalize (5 narrow, 3 wide, or the caller's)
  int lw_cap_now = 0, lw_cap_max = 0;  // table depth built / allowed (tsim_tables_extend deepens on demand)
  long long lw_budget = 0;    // bytes per component
  long long lw_trie_budget = 512ll << 20;  // ... of a prefix-tree table (tsim_trie.hip.h)
  int lw_dense_launches = 0;  // consecutive launches whose hard-row share says "deeper tables would pay"
  unsigned long long deep_after_auto = 0;  // tsim_tables_deep_after's estimate (0: not made yet)
  long long lw_build_bytes = 0;        // ... and their size
  double lw_build_ms = 0.0;            // the finalize build of the tables, timed (entries per ms -> the estimate)
  unsigned long long deep_rows = 0;  // rows launched while the hard rows were too many for k_sample_hw (knobs.deep_after)
  long long lw_bytes = 0;
  uint32_t *d_lw_tab = nullptr;  // integer Bernoulli thresholds (tsimk::bernoulli_threshold)
  // deeper tables being built in the background (tsim_tables_extend_begin / _poll)
  int lw_shadow_off = 0;         // image offset of the shadow copy of the LW records the build works from
  bool ext_pending = false;
  hipEvent_t ext_ev = nullptr;
  uint32_t *ext_tab = nullptr;
  std::vector<void *> ext_scratch;
  std::vector<TsimBuildJob> ext_jobs;  // one per component; launched slice by slice
  size_t ext_job = 0;
  int ext_slices = 0;
  bool ext_uploaded = false, ext_slice_due = false, ext_recorded = false;
  long long ext_entries = 1ll << 21;    // table entries per slice
  std::thread ext_thread;              // allocates the new table and the build scratch (hipMalloc of GBs: up to 30 ms)
  std::atomic<int> ext_alloc{0};       // 0: running, 1: buffers allocated (the launch plans drive the slices), 2: built by the thread, -1: failed
  std::atomic<bool> first_call_out{false};  // a sampling call of this handle has been enqueued (the shallow start's helper thread waits for it, briefly)
  bool ext_self = false;               // this build is driven by the helper thread (the shallow start's default depth)
  std::atomic<bool> ext_abort{false};  // destroy: stop between slices
  hipStream_t ext_stream = nullptr;    // the thread's stream (pooled)
  TsimTablePlan ext_plan;
  std::chrono::steady_clock::time_point ext_t0;
  // device allocations handed out by tsim_malloc_device and not yet freed: the handle owns them
  std::unordered_set<void *> owned;
  // scratch (host-buffer API)
  void *scratch[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t scratch_sz[4] = {0, 0, 0, 0};
  // profiling
  bool profiling = false;
  bool prof_light = false;    // only the events of the first kernel of a launch
  int prof_every = 1;         // bracket one launch in prof_every
  long long prof_counter = 0;
  std::vector<hipEvent_t> ev_pool;
  std::vector<int> ev_tag;
  double prof_stage_ms[4] = {0.0, 0.0, 0.0, 0.0};
  size_t ev_used = 0;
  double prof_ms = 0.0;
  long long prof_launches = 0;
  long long prof_steps = 0;   // batches covered by the bracketed fused first passes (tsim_profile_read_steps)
};

// ---- tsim_program.hip
// Streams come from a process-wide pool per device: hipStreamCreate costs 2.5-3.3 ms on this box and hipStreamDestroy as much
// (scripts/microbench/hip_setup_cost.hip) - a handle with its lanes paid ~100 ms to be born and ~75 ms to die.  A released
// stream is drained first; a handle asks for streams on DIFFERENT hardware queues (HIP deals streams to its four queues in
// creation order: `held` = the creation indices the handle already holds, profiles/r04/hw_queues.txt).
int tsim_stream_acquire(int device, std::vector<int> &held, hipStream_t *out);
void tsim_stream_release(int device, hipStream_t s);
bool tsim_debug(const char *what);  // TSIM_AMD_DEBUG=tables,host,pipeline,pcg contains `what`
int tsim_set_device(const tsim_program *p);
int tsim_need_final(const tsim_program *p);
int tsim_ensure_scratch(tsim_program *p, int slot, size_t bytes);

// ---- tsim_pack.hip (host only)
namespace tsimhost {
void pack_level(HostLevel &h, int W);
bool level_fast_eligible(const HostLevel &h);
bool pack_level_fast(HostLevel &h, int W, std::vector<uint32_t> &tables, bool &fixed_out, int &frame_out);
bool level_v4_eligible(const HostLevel &h, bool wide = false);
void emit_level4(const HostLevel &h, int GT, int nch, const uint32_t *v3recs, std::vector<uint32_t> &recs4,
                 std::vector<uint32_t> &tabs4, int &nch_out, int &ntiles_out, int sparse_F,
                 std::vector<uint32_t> &stabs4);
std::vector<uint32_t> emit_gather_program(std::vector<std::array<int, 3>> e);
std::vector<uint32_t> emit_rotmask_program(const std::vector<std::array<int, 3>> &e);
}  // namespace tsimhost

// ---- tsim_tables.hip
bool tsim_tables_plan(tsim_program *p, int cap, long long budget);
int tsim_tables_build(tsim_program *p, uint32_t **old);
int tsim_tables_extend(tsim_program *p);
int tsim_tables_extend_begin(tsim_program *p, int target_cap);
bool tsim_tables_plan_at(tsim_program *p, int cap, long long budget, int rec_off, TsimTablePlan &out);
int tsim_tables_extend_poll(tsim_program *p, bool wait);
int tsim_tables_slice(tsim_program *p, hipStream_t s);
unsigned long long tsim_tables_deep_after(tsim_program *p);

// ---- tsim_sample.hip
int tsim_flush_hard(tsim_program *p);

// ---- tsim_rows*.hip: launchers of the row-formulation kernels (W = 32-bit words per parameter row)
int tsim_launch_rows(tsim_program *p, int wmax, const tsimk::SampleArgs &a, long long grid, int block, size_t lds,
                     hipStream_t s);
int tsim_launch_lw_build(int W, bool fast, const tsimk::LwBuildArgs &a, int n_out, hipStream_t s);
int tsim_launch_trie_nodes4(const tsimk::LwBuildArgs &a, unsigned grid, hipStream_t s);  // tsim_build4.hip: k_trie_nodes on the chunk tables (a.comp4 != 0)
int tsim_launch_lw_build4(tsim_program *p, int ci, const tsimk::LwBuildArgs &a, int n_out, hipStream_t s);  // tsim_build4.hip: on the chunk tables
namespace tsimrows {
int sample_fast(int wmax, const tsimk::SampleArgs &a, long long grid, int block, size_t lds, hipStream_t s);
int sample_faithful(int wmax, const tsimk::SampleArgs &a, long long grid, int block, size_t lds, hipStream_t s);
int eval_fast(int W, const tsimk::EvalArgs &a, hipStream_t s);
int eval_faithful(int W, const tsimk::EvalArgs &a, hipStream_t s);
int lw_build_fast(int W, const tsimk::LwBuildArgs &a, int n_out, hipStream_t s);
int lw_build_faithful(int W, const tsimk::LwBuildArgs &a, int n_out, hipStream_t s);
}  // namespace tsimrows

// ---- tsim_format.hip
int tsim_launch_pack(tsim_program *p, const uint8_t *d_in, int64_t B, int32_t nbits, uint64_t *d_out, hipStream_t s);
int tsim_launch_unpack(tsim_program *p, const uint64_t *d_in, int64_t B, int32_t nbits, uint8_t *d_out, hipStream_t s);
// padded uint64[B, WO] rows -> uint8[B, ceil(nbits/8)] rows (the reference's bit_packed layout)
int tsim_launch_compact(const uint64_t *d_in, int64_t B, int32_t WO, int32_t nbits, uint8_t *d_out, hipStream_t s);
