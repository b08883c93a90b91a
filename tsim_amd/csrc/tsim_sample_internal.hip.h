// tsim_sample_internal.hip.h - what the translation units of the sampling launchers share (tsim_sample.hip
// had grown to 2200 lines of three launch families):
//   tsim_sample.hip       the launch plan, the pipeline slots, the hard-row stages, the one-batch launcher, the register first
//                         passes' fused groups, the C entry points
//   tsim_sample_wide.hip  k_sample_wide: layout, eligibility, one pass per component, its fused groups
//   tsim_sample_gen.hip   k_sample_gen: layout, eligibility, its fused groups
#pragma once
#include "tsim_internal.hip.h"
#include <chrono>
#include "tsim_kernels.hip.h"
#include "tsim_lw.hip.h"
#include "tsim_noise_wave.hip.h"

// stage tags of the profiling events: 0 opens a launch, the others close a stage
enum { PROF_BEGIN = 0, PROF_PASS1 = 1, PROF_HARD = 2, PROF_FULL = 3 };

// Launch plan from the feedback of earlier launches (results do not depend on it; make_plan, tsim_sample.hip)
struct LaunchPlan {
  bool use_tables = false, need_overflow = true, defer = false;
  bool hard_kernel = true;  // the NW-waves-per-64-rows kernel for the head of every list (few hard rows: latency)
  uint32_t fb_max = 0xFFFFFFFFu;
  int lists = TSIMK_LW_LISTS;  // hard-row sub-lists of this launch: about 40 expected rows each
};

// TSIM_HOST_TIMING=1: where the host time of a several-batches call goes (stderr, one line per call)
struct HostMarks {
  bool on;
  std::chrono::steady_clock::time_point t[24];
  const char *name[24];
  int n = 0;
  HostMarks() {
    static const bool e = tsim_debug("host");
    on = e;
  }
  void mark(const char *what) {
    if (on && n < 24) { name[n] = what; t[n++] = std::chrono::steady_clock::now(); }
  }
  void print() {
    if (!on || n < 2) return;
    fprintf(stderr, "[tsim] host:");
    for (int i = 1; i < n; ++i) fprintf(stderr, " %s %.1f", name[i], std::chrono::duration<double, std::micro>(t[i] - t[i - 1]).count());
    fprintf(stderr, " us\n");
  }
};
extern HostMarks *g_marks;  // (tsim_sample.hip; one sampling thread per handle)
#define TSIM_MARK(w) do { if (g_marks) g_marks->mark(w); } while (0)

int prof_event(tsim_program *p, hipStream_t s, int tag);
int slot_prepare(tsim_program *p, int slot, size_t hard_bytes, bool need_stream = false);
int slot_order_after_previous(tsim_program *p, tsim_program::Slot &sl, hipStream_t s);
int fill_sample_args(tsim_program *p, tsim_program::Slot &sl, tsimk::SampleArgs &a, const uint64_t *d_f, int64_t B, int32_t num_f,
                     uint32_t key_hi, uint32_t key_lo, int64_t shot_offset, uint64_t *d_out, float *d_dev, hipStream_t s,
                     int slot, bool out_bit_packed);
int flush_batch(tsim_program *p);
int flush_chunks(tsim_program *p);
void hard_geometry(tsim_program *p, int WF, int WO);

// ---- device noise in front of the steps (tsim_sample_steps_noise_device, tsim_noise.hip): the request travels thread-locally
// through tsim_sample_steps_device.  A fused group of the one-component register pass runs k_noise_sample_fast (noise + first
// pass in one kernel, tsim_noise_fused.hip.h); every other path gets k_noise_wave on its stream in front of its first pass.
struct TsimNoiseRequest {
  tsimk::NoiseWaveArgs N;        // the sampler's tables and tile geometry (f, B, k0, k1 unset)
  void *noise;                   // tsim_noise *
  int (*launch)(void *noise, int64_t B, uint32_t k0, uint32_t k1, uint64_t *d_f, hipStream_t s);  // k_noise_wave for one batch
  const uint32_t *keys;          // per step: the batch's noise key (2 words)
  int base;                      // step index of the first batch of the group being dispatched
  bool fusable;                  // the tile is a whole number of 1024-row blocks, the channel records fit beside the first pass's tables
};
extern thread_local TsimNoiseRequest *g_noise_req;

// ---- tsim_sample_wide.hip
struct WideLayout {
  int block = 0;       // threads per block (0: the program does not fit)
  int compact = 0;     // 1: the shared column table (WR_CCOL)
  int glob = 0;        // 1: the column tables stay in the image (k_sample_wide<.., GLOB>): they do not fit the LDS
  size_t lds = 0;
  int l_rank, l_lut, l_runs, l_sel, l_ptrs, l_keys, l_tt, l_lvl, l_grec, l_wave, wave_bytes, w_q, w_ovf;
};
WideLayout wide_layout(const tsim_program *p, int WF32, size_t ci = 0);
bool wide_applies(const tsim_program *p, int64_t B, int32_t num_f, int64_t shot_offset);
bool wide_buffers_ok(const tsim_program *p, const tsimk::SampleArgs &a);
int launch_wide(tsim_program *p, int n, const tsimk::SampleArgs *const *args, int64_t B, int32_t num_f, int64_t shot_offset, hipStream_t s);
int steps_group_wide(tsim_program *p, int n, const uint64_t *const *d_f, int64_t B, int32_t num_f, uint32_t key[2],
                     int64_t shot_offset, void *const *d_out, float *const *d_dev, uint32_t flags);

// ---- tsim_sample_gen.hip
#ifndef TSIMK_GEN_MAX_STEPS
#define TSIMK_GEN_MAX_STEPS 8  // batches per fused group of k_sample_gen (tsim_gen.hip.h)
#endif
#ifndef TSIMK_GEN_KEYS
#define TSIMK_GEN_KEYS 320     // subkey records per launch of k_sample_gen: batches x compiled outputs (tsim_gen.hip.h)
#endif
bool gen_applies(const tsim_program *p, int64_t B, int32_t num_f, int64_t shot_offset);
int gen_one(tsim_program *p, const tsim_program::Slot &sl, const SampleArgs &a, int64_t B, int32_t num_f, uint32_t key_hi, uint32_t key_lo, int64_t shot_offset,
            uint32_t *hard_index, uint32_t *ctl, uint32_t *ctl_next, int n_lists, bool has_check, long long *list_cap_out, hipStream_t s);
int steps_group_gen(tsim_program *p, int n, const uint64_t *const *d_f, int64_t B, int32_t num_f, uint32_t key[2],
                    int64_t shot_offset, void *const *d_out, float *const *d_dev, uint32_t flags, const LaunchPlan &plan);
