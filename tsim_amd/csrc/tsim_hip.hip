// tsim_hip.hip - C ABI (include/tsim_hip.h) + host-side packer for the gfx950 engine.
//
// Host responsibilities (all native, no Python in here):
//   * bit-pack the reference-layout program description into one uint32 image
//     that the kernels read through the scalar cache;
//   * own device memory, the stream and HIP-event timing.
#include "../../include/tsim_hip.h"
#include "tsim_kernels.hip.h"
#include "tsim_kernel4.hip.h"
#ifndef TSIM_HARD_NW
#define TSIM_HARD_NW 8   // waves per 64-row group in k_sample4h
#endif
#include "tsim_kernel4h.hip.h"
#include "tsim_lw.hip.h"
#include "tsim_noise.hip.h"
#include "tsim_format.hip.h"

#include <algorithm>
#include <array>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace tsimk;

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) return fail(TSIM_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

extern "C" const char *tsim_last_error(void) { return g_err; }
extern "C" const char *tsim_version(void) { return "tsim_amd-hip 0.1 (gfx950)"; }

// ---------------------------------------------------------------------------
// host-side program
// ---------------------------------------------------------------------------
namespace {

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
  int frame = 0;
  int G = 0, P = 0;
  bool approx = false;
  // per graph
  std::vector<uint32_t> graph_rec;        // G * G_WORDS (row offsets relative to `rows`)
  std::vector<uint32_t> rows;             // packed rows, built for word count W
  long long n_rows = 0;
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
};

static const int kWVariants[] = {1, 2, 3, 4, 6, 8, 12, 16};

static int round_w(int w) {
  for (int v : kWVariants)
    if (w <= v) return v;
  return -1;
}

}  // namespace

struct tsim_program {
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
  int v4_gt = 4;              // graphs per LDS tile
  int comp4_off = 0;
  int v4_max_nch = 1;
  int v4_max_sent = 0;        // entries per tile of the sparse-f tables (0: none)
  long long total_graphs = 0, total_rows = 0;
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
  // low-weight pattern tables (tsim_lw.hip.h)
  int lw_request = -1;        // -1 default (on in TSIM_MODE_AUTO), 0 off, 1 on
  int lw_weight_cap = -1;     // -1 = TSIMK_LW_MAX_WEIGHT
  bool lw = false;            // tables built, pass 1 active
  // Launch slots: slot 0 serves the serial API (everything on the caller's stream); slots 1..4 serve
  // tsim_sample_batch_device_begin/_end: a slot's launches run on the slot's own stream (`side`) so
  // that it overlaps the first pass of the following launches.
  struct Slot {
    int parity = 0;               // counter set of the next launch
    uint32_t *ctl = nullptr;      // 2 counter sets (hard-row counters + check row)
    void *hard = nullptr;         // hard-row lists
    size_t hard_sz = 0;
    uint32_t *keys = nullptr;     // k_keygen output (programs with > TSIMK_INLINE_KEYS outputs)
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
    hipStream_t p1_stream = nullptr;  // lane of that first pass
    hipStream_t last_done = nullptr;  // stream on which done_ev of the slot's last launch was recorded
    hipEvent_t done_ev = nullptr;     // ev2 (own second pass) or the event of the batch that served the slot
    unsigned long long batch_seq = 0; // sequence number of that batch (0: own second pass)
    SampleArgs ctx;                   // the hard-row kernel's arguments for that launch
  };
  Slot slots[1 + TSIM_PIPELINE_SLOTS];
  bool slots_ready = false;
  std::vector<int> deferred;  // slots whose hard rows are waiting, in launch order
  hipEvent_t lane_ev[2] = {nullptr, nullptr};  // "first passes enqueued so far on lane k are done"
  hipEvent_t batch_ev[16] = {};                // ring: one event per hard-row batch
  hipEvent_t sync_ev = nullptr;                // tsim_pipeline_wait_stream
  uint8_t *series_ptr = nullptr;               // tsim_pipeline_set_compact_series
  int64_t series_stride = 0;
  int series_left = 0;
  unsigned long long batch_next = 1;       // sequence number of the next batch (event = batch_ev[seq % 16])
  unsigned long long batch_confirmed = 0;  // every batch up to this one is known to be complete
  unsigned long long lane_waited[2] = {0, 0};  // newest batch each first-pass lane already waits for
  unsigned long long stat_queries = 0, stat_waits = 0, stat_begins = 0, stat_flushes = 0, stat_deferred = 0;
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
    bool adaptive = true;     // TSIM_AMD_ADAPTIVE=0 pins the default launch plan
    bool hard_kernel = true;  // TSIM_AMD_HARD_KERNEL=0: hard rows on k_sample4 instead of k_sample4h
    bool lane0_main = true;   // TSIM_AMD_LANE0_MAIN=0: pipeline slot 0 gets a stream of its own
    int lw_block = 0;         // TSIM_AMD_LW_BLOCK (0: 1024 threads when the f/out staging fits 32 KB, else 256)
    int v4_block = 256;       // TSIM_AMD_V4_BLOCK
    int hard_lds_kb = 128;    // TSIM_AMD_HARD_LDS_KB (128: a first-pass block still fits next to a hard-row block)
    bool merge_lists = true;  // TSIM_AMD_MERGE_LISTS=0: always TSIMK_LW_LISTS hard-row sub-lists
    int list_rows = 40;       // TSIM_AMD_LIST_ROWS: expected hard rows per list the list count aims at
    int min_lists = 4;        // TSIM_AMD_MIN_LISTS (power of two >= 2)
    bool defer = true;        // TSIM_AMD_DEFER_HARD=0: every pipelined launch runs its own second pass
    int defer_group = 4;      // TSIM_AMD_DEFER_GROUP: launches per deferred hard-row batch (<= TSIMK_H_MAX_CTX)
  } knobs;
  bool h_attr_set = false;    // k_sample4h: large dynamic LDS enabled
  bool hm_attr_set = false;   // k_sample4h_multi: the same
  int lw_off = 0;             // image offset of the LW component records
  int lw_direct_prog = 0;     // image offset of the direct-output gather program
  int lw_direct_chunks = 0;
  std::vector<int> lw_wmax;   // per component
  long long lw_bytes = 0;
  float *d_lw_tab = nullptr;
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
};

static int set_device(const tsim_program *p) {
  HIP_TRY(hipSetDevice(p->device));
  return 0;
}

// ---------------------------------------------------------------------------
// construction
// ---------------------------------------------------------------------------
extern "C" int tsim_program_create(int32_t num_outputs, int32_t num_detectors, int32_t n_direct,
                                   const int32_t *direct_f_indices, const uint8_t *direct_flips,
                                   const int32_t *output_order, tsim_program **out) {
  if (!out) return fail(TSIM_EINVAL, "out is NULL");
  if (num_outputs < 0 || n_direct < 0 || n_direct > num_outputs || num_detectors < 0)
    return fail(TSIM_EINVAL, "bad counts: num_outputs=%d n_direct=%d num_detectors=%d", num_outputs,
                n_direct, num_detectors);
  if (n_direct > 0 && (!direct_f_indices || !direct_flips)) return fail(TSIM_EINVAL, "direct arrays NULL");
  if (num_outputs > 0 && !output_order) return fail(TSIM_EINVAL, "output_order is NULL");
  tsim_program *p = new (std::nothrow) tsim_program();
  if (!p) return fail(TSIM_ENOMEM, "out of host memory");
  p->num_outputs = num_outputs;
  p->num_detectors = num_detectors;
  p->n_direct = n_direct;
  p->direct_f.assign(direct_f_indices, direct_f_indices + n_direct);
  p->direct_flips.assign(direct_flips, direct_flips + n_direct);
  p->output_order.assign(output_order, output_order + num_outputs);
  std::vector<char> seen(num_outputs, 0);
  for (int i = 0; i < num_outputs; ++i) {
    int o = p->output_order[i];
    if (o < 0 || o >= num_outputs || seen[o]) {
      delete p;
      return fail(TSIM_EINVAL, "output_order is not a permutation of 0..%d", num_outputs - 1);
    }
    seen[o] = 1;
  }
  for (int i = 0; i < n_direct; ++i) {
    if (p->direct_f[i] < 0) {
      delete p;
      return fail(TSIM_EINVAL, "negative direct_f_indices[%d]", i);
    }
    p->max_f_index = std::max(p->max_f_index, p->direct_f[i]);
  }
  *out = p;
  return TSIM_OK;
}

extern "C" int tsim_program_add_component(tsim_program *p, int32_t n_out, const int32_t *output_indices,
                                          int32_t F, const int32_t *f_selection, int32_t n_levels) {
  if (!p) return fail(TSIM_EINVAL, "program is NULL");
  if (p->finalized) return fail(TSIM_ESTATE, "program already finalized");
  if (n_out < 0 || F < 0) return fail(TSIM_EINVAL, "negative n_out/F");
  if (n_levels != n_out + 1 && n_levels != 2)
    return fail(TSIM_EINVAL, "component with %d outputs needs %d (sequential) or 2 (joint) levels, got %d",
                n_out, n_out + 1, n_levels);
  if ((n_out > 0 && !output_indices) || (F > 0 && !f_selection)) return fail(TSIM_EINVAL, "NULL index array");
  HostComponent c;
  c.n_out = n_out;
  c.F = F;
  c.n_levels = n_levels;
  c.output_indices.assign(output_indices, output_indices + n_out);
  c.f_selection.assign(f_selection, f_selection + F);
  for (int i = 0; i < F; ++i) {
    if (c.f_selection[i] < 0) return fail(TSIM_EINVAL, "negative f_selection[%d]", i);
    p->max_f_index = std::max(p->max_f_index, c.f_selection[i]);
  }
  p->comps.push_back(std::move(c));
  return (int)p->comps.size() - 1;
}

template <class T>
static void copy_arr(std::vector<T> &dst, const T *src, size_t n) {
  if (n && src) dst.assign(src, src + n);
  else dst.assign(n, T());
}

extern "C" int tsim_program_add_level(tsim_program *p, int32_t component, const tsim_level_desc *L) {
  if (!p || !L) return fail(TSIM_EINVAL, "NULL argument");
  if (p->finalized) return fail(TSIM_ESTATE, "program already finalized");
  if (component < 0 || component >= (int)p->comps.size()) return fail(TSIM_EINVAL, "bad component index %d", component);
  HostComponent &c = p->comps[component];
  if ((int)c.levels.size() >= c.n_levels) return fail(TSIM_EINVAL, "component %d already has all %d levels", component, c.n_levels);
  const int k = (int)c.levels.size();
  const bool sequential = (c.n_levels == c.n_out + 1);
  const int want_P = c.F + (sequential ? k : (k == 0 ? 0 : c.n_out));
  if (L->n_params != want_P)
    return fail(TSIM_EINVAL, "component %d level %d: n_params=%d, expected %d", component, k, L->n_params, want_P);
  if (L->num_graphs < 0 || L->ta < 0 || L->tb < 0 || L->tc < 0 || L->td < 0) return fail(TSIM_EINVAL, "negative sizes");
  if (L->n_params > TSIM_MAX_PARAMS)
    return fail(TSIM_ENOTSUP, "n_params=%d exceeds TSIM_MAX_PARAMS=%d", L->n_params, TSIM_MAX_PARAMS);
  const size_t G = L->num_graphs, P = L->n_params;
  if (G > 0) {
    const void *req[] = {L->phase_indices, L->floatfactor, L->power2};
    for (const void *q : req)
      if (!q) return fail(TSIM_EINVAL, "prefactor array is NULL");
    if ((L->ta && (!L->a_phases || !L->a_counts || (P && !L->a_params))) ||
        (L->tb && (!L->b_coeffs || (P && !L->b_params))) ||
        (L->tc && (!L->c_psi_const || !L->c_phi_const || (P && (!L->c_psi_params || !L->c_phi_params)))) ||
        (L->td && (!L->d_alpha || !L->d_beta || !L->d_counts || (P && (!L->d_alpha_params || !L->d_beta_params)))))
      return fail(TSIM_EINVAL, "term array is NULL");
    if (L->has_approx && !L->approx) return fail(TSIM_EINVAL, "has_approx set but approx is NULL");
  }
  HostLevel h;
  h.G = (int)G;
  h.P = (int)P;
  h.approx = L->has_approx != 0;
  h.d = *L;
  copy_arr(h.u8[0], L->a_phases, G * L->ta);
  copy_arr(h.u8[1], L->a_params, G * L->ta * P);
  copy_arr(h.u8[2], L->b_coeffs, G * L->tb);
  copy_arr(h.u8[3], L->b_params, G * L->tb * P);
  copy_arr(h.u8[4], L->c_psi_const, G * L->tc);
  copy_arr(h.u8[5], L->c_psi_params, G * L->tc * P);
  copy_arr(h.u8[6], L->c_phi_const, G * L->tc);
  copy_arr(h.u8[7], L->c_phi_params, G * L->tc * P);
  copy_arr(h.u8[8], L->d_alpha, G * L->td);
  copy_arr(h.u8[9], L->d_alpha_params, G * L->td * P);
  copy_arr(h.u8[10], L->d_beta, G * L->td);
  copy_arr(h.u8[11], L->d_beta_params, G * L->td * P);
  copy_arr(h.u8[12], L->phase_indices, G);
  copy_arr(h.i32[0], L->a_counts, L->ta ? G : 0);
  copy_arr(h.i32[1], L->d_counts, L->td ? G : 0);
  copy_arr(h.i32[2], L->floatfactor, G * 4);
  copy_arr(h.i32[3], L->power2, G);
  if (L->approx) copy_arr(h.approx_v, L->approx, G * 2);
  else { h.approx_v.assign(G * 2, 0.0f); for (size_t g = 0; g < G; ++g) h.approx_v[2 * g] = 1.0f; }
  for (size_t g = 0; g < G; ++g) {
    if (L->ta && (h.i32[0][g] < 0 || h.i32[0][g] > L->ta)) return fail(TSIM_EINVAL, "a_counts[%zu] out of range", g);
    if (L->td && (h.i32[1][g] < 0 || h.i32[1][g] > L->td)) return fail(TSIM_EINVAL, "d_counts[%zu] out of range", g);
  }
  c.levels.push_back(std::move(h));
  return TSIM_OK;
}

// pack one byte-per-bit row into W 32-bit words appended to `dst`; returns true if any bit set
static bool pack_row(std::vector<uint32_t> &dst, const uint8_t *bits, int P, int W) {
  bool any = false;
  size_t base = dst.size();
  dst.resize(base + W, 0u);
  for (int i = 0; i < P; ++i)
    if (bits[i] & 1) {  // reference bit-matrices hold 0/1 (compile.py:40-238)
      dst[base + (i >> 5)] |= 1u << (i & 31);
      any = true;
    }
  return any;
}

static const int8_t kUnit[8][4] = {{1, 0, 0, 0}, {0, 1, 0, 0},  {0, 0, 1, 0},  {0, 0, 0, -1},
                                   {-1, 0, 0, 0}, {0, -1, 0, 0}, {0, 0, -1, 0}, {0, 0, 0, 1}};

// Build graph records + rows for a level with W words per row.
static void pack_level(HostLevel &h, int W) {
  const int G = h.G, P = h.P;
  const tsim_level_desc &d = h.d;
  h.graph_rec.assign((size_t)G * G_WORDS, 0u);
  h.rows.clear();
  h.n_rows = 0;
  std::vector<uint32_t> tmp;
  for (int g = 0; g < G; ++g) {
    uint32_t *rec = &h.graph_rec[(size_t)g * G_WORDS];
    rec[G_ROWS] = (uint32_t)h.rows.size();
    // A: the first counts[g] slots are real (terms.py:70-71)
    const int nA = d.ta ? h.i32[0][g] : 0;
    for (int t = 0; t < nA; ++t) {
      h.rows.push_back((uint32_t)(h.u8[0][(size_t)g * d.ta + t] & 7));
      pack_row(h.rows, &h.u8[1][((size_t)g * d.ta + t) * P], P, W);
    }
    rec[G_NA] = (uint32_t)nA;
    // B: zero coefficient or empty parity contributes exponent 0 (terms.py:104-106) -> dropped
    int nB = 0;
    for (int t = 0; t < d.tb; ++t) {
      const uint32_t coeff = h.u8[2][(size_t)g * d.tb + t] & 7u;  // (rowsum*coeff) % 8
      if (!coeff) continue;
      tmp.clear();
      if (!pack_row(tmp, &h.u8[3][((size_t)g * d.tb + t) * P], P, W)) continue;
      h.rows.push_back(coeff);
      h.rows.insert(h.rows.end(), tmp.begin(), tmp.end());
      ++nB;
    }
    rec[G_NB] = (uint32_t)nB;
    // C: a slot whose psi or phi is identically 0 contributes (-1)^0 (terms.py:136-141) -> dropped
    int nC = 0;
    for (int t = 0; t < d.tc; ++t) {
      const uint32_t pc = h.u8[4][(size_t)g * d.tc + t] & 1u, qc = h.u8[6][(size_t)g * d.tc + t] & 1u;
      std::vector<uint32_t> r1, r2;
      const bool any1 = pack_row(r1, &h.u8[5][((size_t)g * d.tc + t) * P], P, W);
      const bool any2 = pack_row(r2, &h.u8[7][((size_t)g * d.tc + t) * P], P, W);
      if ((!any1 && !pc) || (!any2 && !qc)) continue;
      h.rows.push_back(pc | (qc << 1));
      h.rows.insert(h.rows.end(), r1.begin(), r1.end());
      h.rows.insert(h.rows.end(), r2.begin(), r2.end());
      ++nC;
    }
    rec[G_NC] = (uint32_t)nC;
    // D: first counts[g] slots are real; the four possible term values are tabulated
    const int nD = d.td ? h.i32[1][g] : 0;
    for (int t = 0; t < nD; ++t) {
      const int al = h.u8[8][(size_t)g * d.td + t] & 7, be = h.u8[10][(size_t)g * d.td + t] & 7;
      for (int idx = 0; idx < 4; ++idx) {
        const int pa = idx & 1, pb = idx >> 1;
        const int a1 = (al + 4 * pa) & 7, b1 = (be + 4 * pb) & 7, g1 = (a1 + b1) & 7;
        uint32_t w = 0;
        for (int j = 0; j < 4; ++j) {
          const int v = (j == 0 ? 1 : 0) + kUnit[a1][j] + kUnit[b1][j] - kUnit[g1][j];
          w |= (uint32_t)(uint8_t)(int8_t)v << (8 * j);
        }
        h.rows.push_back(w);
      }
      pack_row(h.rows, &h.u8[9][((size_t)g * d.td + t) * P], P, W);
      pack_row(h.rows, &h.u8[11][((size_t)g * d.td + t) * P], P, W);
    }
    rec[G_ND] = (uint32_t)nD;
    h.n_rows += nA + nB + 2 * nC + 2 * nD;
    rec[G_PHASE] = h.u8[12][g] & 7u;
    const int32_t *ff = &h.i32[2][(size_t)g * 4];
    rec[G_FFA] = (uint32_t)ff[0]; rec[G_FFB] = (uint32_t)ff[1];
    rec[G_FFC] = (uint32_t)ff[2]; rec[G_FFD] = (uint32_t)ff[3];
    rec[G_POW2] = (uint32_t)h.i32[3][g];
    memcpy(&rec[G_APRE], &h.approx_v[2 * (size_t)g], 4);
    memcpy(&rec[G_APIM], &h.approx_v[2 * (size_t)g + 1], 4);
    rec[G_FLAGS] = (ff[0] == 1 && ff[1] == 0 && ff[2] == 0 && ff[3] == 0) ? TSIMK_GFLAG_FF_IS_ONE : 0u;
  }
}

// ---------------------------------------------------------------------------
// "fast exact" packing (kernel: eval_level_fast)
// ---------------------------------------------------------------------------
namespace {

// exact element of Z[w] * 2^p on basis (1, w, i, conj w), int64 coefficients, kept canonical
struct ZW {
  long long c[4];
  int p;
};

static void zw_canon(ZW &z) {
  if (!(z.c[0] | z.c[1] | z.c[2] | z.c[3])) return;
  while (!((z.c[0] | z.c[1] | z.c[2] | z.c[3]) & 1)) {
    for (auto &v : z.c) v >>= 1;
    ++z.p;
  }
}

static void zw_mul(ZW &x, const long long y[4]) {  // exact_scalar.py:19-39, then canonicalise
  const long long a1 = x.c[0], b1 = x.c[1], c1 = x.c[2], d1 = x.c[3];
  const long long a2 = y[0], b2 = y[1], c2 = y[2], d2 = y[3];
  x.c[0] = a1 * a2 + b1 * d2 - c1 * c2 + d1 * b2;
  x.c[1] = a1 * b2 + b1 * a2 + c1 * d2 + d1 * c2;
  x.c[2] = a1 * c2 + b1 * b2 + c1 * a2 - d1 * d2;
  x.c[3] = a1 * d2 - b1 * c2 - c1 * b2 + d1 * a2;
  zw_canon(x);
}

static void unit_plus_one(int k, long long out[4]) {  // 1 + w^k
  for (int j = 0; j < 4; ++j) out[j] = kUnit[k & 7][j];
  out[0] += 1;
}

// ---- GF(2) algebra used by the fast packer -------------------------------------------------
// An affine form c ^ <m, x> over the level's P parameters.
struct Affine {
  std::vector<uint64_t> m;
  bool c = false;
};

static Affine affine_from(const uint8_t *bits, int P, bool c) {
  Affine a;
  a.m.assign((size_t)(P + 63) / 64 + 1, 0ull);
  for (int i = 0; i < P; ++i)
    if (bits[i] & 1) a.m[i >> 6] ^= 1ull << (i & 63);
  a.c = c;
  return a;
}

// A quadratic form over GF(2): q(x) = sum_{i<j} B[i][j] x_i x_j  ^  <lin, x>  ^  c.
// B is kept symmetric with zero diagonal (x_i^2 = x_i goes to `lin`).
struct QForm {
  int P = 0, PW = 0;
  std::vector<uint64_t> B;  // P rows x PW words
  std::vector<uint64_t> lin;
  bool c = false;
  explicit QForm(int p) : P(p), PW((p + 63) / 64 + 1), B((size_t)p * ((p + 63) / 64 + 1), 0ull), lin((p + 63) / 64 + 1, 0ull) {}
  uint64_t *row(int i) { return &B[(size_t)i * PW]; }
  void add_linear(const Affine &a) {
    for (int w = 0; w < PW && w < (int)a.m.size(); ++w) lin[w] ^= a.m[w];
    c ^= a.c;
  }
  // q ^= (a.c ^ <a.m,x>) * (b.c ^ <b.m,x>)
  void add_product(const Affine &a, const Affine &b) {
    for (int i = 0; i < P; ++i) {
      if (!((a.m[i >> 6] >> (i & 63)) & 1)) continue;
      uint64_t *ri = row(i);
      for (int w = 0; w < PW && w < (int)b.m.size(); ++w) ri[w] ^= b.m[w];  // row i ^= b (may set the diagonal)
    }
    // symmetrise: the loop above added the ordered pairs (i in a, j in b); fold (i,j) and (j,i) together
    // by rebuilding the symmetric part lazily in `finish()`.
    if (a.c) for (int w = 0; w < PW && w < (int)b.m.size(); ++w) lin[w] ^= b.m[w];
    if (b.c) for (int w = 0; w < PW && w < (int)a.m.size(); ++w) lin[w] ^= a.m[w];
    c ^= (a.c && b.c);
  }
  // After all add_product calls B holds an arbitrary (non-symmetric) bilinear matrix M with
  // q = x^T M x.  Convert to the canonical alternating form: B'[i][j] = M[i][j] ^ M[j][i], diagonal -> lin.
  void finish() {
    for (int i = 0; i < P; ++i)
      if ((row(i)[i >> 6] >> (i & 63)) & 1) {
        lin[i >> 6] ^= 1ull << (i & 63);
        row(i)[i >> 6] ^= 1ull << (i & 63);
      }
    for (int i = 0; i < P; ++i)
      for (int j = i + 1; j < P; ++j) {
        const bool mij = (row(i)[j >> 6] >> (j & 63)) & 1, mji = (row(j)[i >> 6] >> (i & 63)) & 1;
        const bool s = mij ^ mji;
        if (mij != s) row(i)[j >> 6] ^= 1ull << (j & 63);
        if (mji != s) row(j)[i >> 6] ^= 1ull << (i & 63);
      }
  }
  bool get(int i, int j) { return (row(i)[j >> 6] >> (j & 63)) & 1; }
};

// Dickson reduction: q = XOR_s <u_s,x><v_s,x> ^ <lin,x> ^ c with rank(B)/2 product pairs.
// Pivot on (i,j) with B[i][j] = 1: with alpha = B[i] \ {i,j}, beta = B[j] \ {i,j},
//   x_i x_j ^ x_i<alpha,x> ^ x_j<beta,x> = (x_i ^ <beta,x>)(x_j ^ <alpha,x>) ^ <alpha,x><beta,x>.
static void dickson_reduce(QForm &q, std::vector<std::vector<uint64_t>> &us, std::vector<std::vector<uint64_t>> &vs) {
  const int P = q.P, PW = q.PW;
  for (int i = 0; i < P; ++i) {
    for (;;) {
      int j = -1;
      for (int t = 0; t < P; ++t)
        if (q.get(i, t)) { j = t; break; }
      if (j < 0) break;
      std::vector<uint64_t> alpha(q.row(i), q.row(i) + PW), beta(q.row(j), q.row(j) + PW);
      alpha[j >> 6] &= ~(1ull << (j & 63));  // B[i][i] is 0 already
      beta[i >> 6] &= ~(1ull << (i & 63));
      std::vector<uint64_t> u = beta, v = alpha;
      u[i >> 6] ^= 1ull << (i & 63);
      v[j >> 6] ^= 1ull << (j & 63);
      us.push_back(u);
      vs.push_back(v);
      // remove variables i and j from B
      for (int w = 0; w < PW; ++w) q.row(i)[w] = q.row(j)[w] = 0ull;
      for (int k = 0; k < P; ++k) {
        q.row(k)[i >> 6] &= ~(1ull << (i & 63));
        q.row(k)[j >> 6] &= ~(1ull << (j & 63));
      }
      // q ^= <alpha,x><beta,x>: B[k][l] ^= alpha_k beta_l ^ alpha_l beta_k ; lin_k ^= alpha_k beta_k
      for (int k = 0; k < P; ++k) {
        const bool ak = (alpha[k >> 6] >> (k & 63)) & 1, bk = (beta[k >> 6] >> (k & 63)) & 1;
        if (ak) for (int w = 0; w < PW; ++w) q.row(k)[w] ^= beta[w];
        if (bk) for (int w = 0; w < PW; ++w) q.row(k)[w] ^= alpha[w];
        if (ak && bk) q.lin[k >> 6] ^= 1ull << (k & 63);
        q.row(k)[k >> 6] &= ~(1ull << (k & 63));  // diagonal stays zero
      }
    }
  }
}

static bool push_mask_row(std::vector<uint32_t> &dst, const std::vector<uint64_t> &m, int P, int W) {
  bool any = false;
  const size_t base = dst.size();
  dst.resize(base + W, 0u);
  for (int i = 0; i < P; ++i)
    if ((m[i >> 6] >> (i & 63)) & 1) {
      dst[base + (i >> 5)] |= 1u << (i & 31);
      any = true;
    }
  return any;
}

// Can this level be evaluated by the counting formulation?  (see eval_level_fast)
static bool level_fast_eligible(const HostLevel &h) {
  const tsim_level_desc &d = h.d;
  for (int g = 0; g < h.G; ++g) {
    const int nA = d.ta ? h.i32[0][g] : 0;
    if (nA > 30) return false;  // beyond this the reference's own int32 scan may wrap
    for (int t = 0; t < d.tb; ++t) {
      const unsigned c = h.u8[2][(size_t)g * d.tb + t] & 7u;
      if (c & 1u) return false;  // odd eighth-turn coefficients never come out of the compiler
    }
    if (d.td > 60000 || h.P > 60000) return false;
    for (int j = 0; j < 4; ++j)
      if (std::llabs((long long)h.i32[2][(size_t)g * 4 + j]) > (1ll << 20)) return false;
  }
  return true;
}

// Build graph records + rows + NodePhases tables for eval_level_fast.  Returns false if a table
// entry does not fit int32 (the caller then falls back to the faithful layout).
//
// Per graph the w-exponent contributed by HalfPi rows (coefficients 2,4,6), PiProducts and the
// (-i)^(m1+m2+m3) of the NodePhases is rewritten at pack time as
//     k(x) = k0 + 2 * <lam, x> + 4 * ( <lin, x> ^ XOR_s <u_s,x><v_s,x> )          (mod 8)
// using  2*(sum of bits p_t) = 2*(XOR p_t) + 4*e2(p)  (mod 8)  for the list of bits that enter with
// coefficient 2 (coefficient 6 = 2 + 4), e2 = second elementary symmetric polynomial, and the Dickson
// normal form of the resulting GF(2) quadratic form.  k0 is folded into the table as a rotation.
static bool pack_level_fast(HostLevel &h, int W, std::vector<uint32_t> &tables, bool &fixed_out, int &frame_out) {
  const int G = h.G, P = h.P;
  const tsim_level_desc &d = h.d;
  h.graph_rec.assign((size_t)G * G_WORDS, 0u);
  h.rows.clear();
  h.n_rows = 0;
  tables.clear();
  fixed_out = false;
  frame_out = 0;
  std::vector<std::vector<ZW>> entries((size_t)G);   // per graph: the main table (entry 0 excluded)
  std::vector<std::vector<ZW>> dentries((size_t)G);  // per graph: the separate PhasePairs table, if any
  h.fg.assign((size_t)G, FastGraph());
  for (int g = 0; g < G; ++g) {
    FastGraph &fg = h.fg[(size_t)g];
    uint32_t *rec = &h.graph_rec[(size_t)g * G_WORDS];
    rec[GF_ROWS] = (uint32_t)h.rows.size();
    const int nA = d.ta ? h.i32[0][g] : 0;
    std::vector<Affine> two;  // bits entering the exponent with coefficient 2
    QForm q4(P);              // bit entering with coefficient 4
    int k0 = h.u8[12][g] & 7; // static phase
    auto add_six = [&](const Affine &a) {  // 6*p = 2*p + 4*p
      two.push_back(a);
      q4.add_linear(a);
    };
    // ---- NodePhases rows counted per class 0, 1, 3 (class 2 only feeds the exponent)
    int n[4] = {0, 0, 0, 0};
    for (int cls = 0; cls < 4; ++cls)
      for (int t = 0; t < nA; ++t) {
        const unsigned ph = h.u8[0][(size_t)g * d.ta + t] & 7u;
        if ((int)(ph & 3u) != cls) continue;
        const uint8_t *bits = &h.u8[1][((size_t)g * d.ta + t) * P];
        ++n[cls];
        if (cls != 0) add_six(affine_from(bits, P, (ph >> 2) != 0));  // (-i)^(par')
        if (cls == 2) continue;
        h.rows.push_back(ph >> 2);
        pack_row(h.rows, bits, P, W);
        auto &dstm = cls == 0 ? fg.c0 : (cls == 1 ? fg.c1 : fg.c3);
        auto &dstc = cls == 0 ? fg.c0c : (cls == 1 ? fg.c1c : fg.c3c);
        dstm.push_back(affine_from(bits, P, false).m);
        dstc.push_back((uint8_t)(ph >> 2));
      }
    rec[GF_N01] = (uint32_t)n[0] | ((uint32_t)n[1] << 16);
    fg.n1 = n[1];
    rec[GF_N1] = (uint32_t)n[1];
    // ---- PhasePairs rows: two table-index bits per term when the combined table stays small,
    //      else the faithful sequential scan (rows carry the four tabulated term values)
    const int nD = d.td ? h.i32[1][g] : 0;
    if (nD > 5) return false;  // 4^nD table entries: beyond this use the faithful layout
    const long long combos = (long long)(n[1] + n[3] + 1) << (2 * nD);
    static const long long kMaxCombined = getenv("TSIM_AMD_COMBINED") ? atoll(getenv("TSIM_AMD_COMBINED")) : 1024;
    const bool d_tabled = nD > 0 && combos <= kMaxCombined;   // combined table, else a separate one
    std::vector<std::array<std::array<int, 4>, 4>> dterm((size_t)nD);  // [t][pa + 2 pb] -> term value
    for (int t = 0; t < nD; ++t) {
      const int al = h.u8[8][(size_t)g * d.td + t] & 7, be = h.u8[10][(size_t)g * d.td + t] & 7;
      for (int idx = 0; idx < 4; ++idx) {
        const int pa = idx & 1, pb = idx >> 1;
        const int a1 = (al + 4 * pa) & 7, b1 = (be + 4 * pb) & 7, g1 = (a1 + b1) & 7;
        uint32_t w = 0;
        for (int j = 0; j < 4; ++j) {
          const int v = (j == 0 ? 1 : 0) + kUnit[a1][j] + kUnit[b1][j] - kUnit[g1][j];
          dterm[t][idx][j] = v;
          w |= (uint32_t)(uint8_t)(int8_t)v << (8 * j);
        }
        (void)w;
      }
      pack_row(h.rows, &h.u8[9][((size_t)g * d.td + t) * P], P, W);
      pack_row(h.rows, &h.u8[11][((size_t)g * d.td + t) * P], P, W);
      fg.dal.push_back(affine_from(&h.u8[9][((size_t)g * d.td + t) * P], P, false).m);
      fg.dbe.push_back(affine_from(&h.u8[11][((size_t)g * d.td + t) * P], P, false).m);
    }
    rec[GF_ND] = (uint32_t)nD;
    fg.nD = nD;
    fg.d_tabled = d_tabled;
    // ---- HalfPi rows
    for (int t = 0; t < d.tb; ++t) {
      const uint32_t coeff = h.u8[2][(size_t)g * d.tb + t] & 7u;
      if (!coeff) continue;
      const Affine a = affine_from(&h.u8[3][((size_t)g * d.tb + t) * P], P, false);
      if (coeff == 2) two.push_back(a);
      else if (coeff == 4) q4.add_linear(a);
      else add_six(a);
    }
    // ---- PiProducts: 4 * psi * phi
    for (int t = 0; t < d.tc; ++t) {
      const Affine psi = affine_from(&h.u8[5][((size_t)g * d.tc + t) * P], P, h.u8[4][(size_t)g * d.tc + t] & 1u);
      const Affine phi = affine_from(&h.u8[7][((size_t)g * d.tc + t) * P], P, h.u8[6][(size_t)g * d.tc + t] & 1u);
      q4.add_product(psi, phi);
    }
    // ---- 2 * sum(two) = 2 * XOR(two) + 4 * e2(two)
    Affine lam;
    lam.m.assign((size_t)(P + 63) / 64 + 1, 0ull);
    for (size_t s = 0; s < two.size(); ++s) {
      for (size_t w = 0; w < lam.m.size(); ++w) lam.m[w] ^= two[s].m[w];
      lam.c ^= two[s].c;
      for (size_t t2 = s + 1; t2 < two.size(); ++t2) q4.add_product(two[s], two[t2]);
    }
    if (lam.c) {  // 2*(1 ^ y) = 2 + 6*y = 2 + 2*y + 4*y
      k0 += 2;
      Affine y = lam;
      y.c = false;
      q4.add_linear(y);
    }
    q4.finish();
    std::vector<std::vector<uint64_t>> us, vs;
    dickson_reduce(q4, us, vs);
    if (q4.c) k0 += 4;
    if (us.size() > 60000) return false;
    // ---- rows: lam, lin, then the product pairs
    uint32_t flags = d_tabled ? TSIMK_GFLAG_D_TABLED : 0u;
    {
      std::vector<uint32_t> tmp;
      if (push_mask_row(tmp, lam.m, P, W)) { flags |= TSIMK_GFLAG_LAM; h.rows.insert(h.rows.end(), tmp.begin(), tmp.end()); }
      tmp.clear();
      if (push_mask_row(tmp, q4.lin, P, W)) { flags |= TSIMK_GFLAG_LIN; h.rows.insert(h.rows.end(), tmp.begin(), tmp.end()); }
    }
    for (size_t s = 0; s < us.size(); ++s) {
      push_mask_row(h.rows, us[s], P, W);
      push_mask_row(h.rows, vs[s], P, W);
    }
    rec[GF_N3H] = (uint32_t)n[3] | ((uint32_t)us.size() << 16);
    rec[GF_FLAGS] = flags;
    fg.lam = lam.m;
    fg.lin = q4.lin;
    fg.us = us;
    fg.vs = vs;
    h.n_rows += n[0] + n[1] + n[3] + ((flags & TSIMK_GFLAG_LAM) ? 1 : 0) + ((flags & TSIMK_GFLAG_LIN) ? 1 : 0) +
                2 * (long long)us.size() + 2 * nD;
    // ---- table entries (index: ((delta + n1) << 2 nD | dbits) + 1; entry 0 is the exact zero)
    //   canon( 2^n0 (1+i)^n2 (1+w)^(n1-m1) (1-w)^m1 (1+w^3)^(n3-m3) (1-w^3)^m3 * i^(m1+m3)
    //          * floatfactor * w^k0 * [product of the PhasePairs terms selected by dbits] ) * 2^power2
    //   with m3 = max(delta,0), m1 = max(-delta,0)
    const long long ff[4] = {h.i32[2][(size_t)g * 4], h.i32[2][(size_t)g * 4 + 1], h.i32[2][(size_t)g * 4 + 2],
                             h.i32[2][(size_t)g * 4 + 3]};
    const int ndb = d_tabled ? (1 << (2 * nD)) : 1;
    for (int delta = -n[1]; delta <= n[3]; ++delta) {
      const int m3 = delta > 0 ? delta : 0, m1 = delta < 0 ? -delta : 0;
      ZW z{{1, 0, 0, 0}, 0};
      long long f[4];
      auto mul_k = [&](int k, int times) {
        unit_plus_one(k, f);
        for (int i = 0; i < times; ++i) zw_mul(z, f);
      };
      mul_k(0, n[0]);
      mul_k(2, n[2]);
      mul_k(1, n[1] - m1);
      mul_k(5, m1);
      mul_k(3, n[3] - m3);
      mul_k(7, m3);
      const int r = (2 * (m1 + m3) + k0) & 7;
      const long long rot[4] = {kUnit[r][0], kUnit[r][1], kUnit[r][2], kUnit[r][3]};
      zw_mul(z, rot);
      zw_mul(z, ff);
      z.p += h.i32[3][g];  // power2
      for (int db = 0; db < ndb; ++db) {
        ZW e = z;
        for (int t = 0; t < nD && d_tabled; ++t) {
          const int sel = (db >> (2 * (nD - 1 - t))) & 3;  // term 0 holds the most significant pair
          const long long tv[4] = {dterm[t][sel][0], dterm[t][sel][1], dterm[t][sel][2], dterm[t][sel][3]};
          zw_mul(e, tv);
        }
        for (int j = 0; j < 4; ++j)
          if (e.c[j] > INT32_MAX || e.c[j] < INT32_MIN) return false;
        if (!(e.c[0] | e.c[1] | e.c[2] | e.c[3])) e.p = h.approx ? 0 : TSIMK_ZERO_POWER;
        entries[g].push_back(e);
      }
    }
    if (nD > 0 && !d_tabled) {  // separate PhasePairs table over the 4^nD parity patterns
      for (int db = 0; db < (1 << (2 * nD)); ++db) {
        ZW e{{1, 0, 0, 0}, 0};
        for (int t = 0; t < nD; ++t) {
          const int sel = (db >> (2 * (nD - 1 - t))) & 3;
          const long long tv[4] = {dterm[t][sel][0], dterm[t][sel][1], dterm[t][sel][2], dterm[t][sel][3]};
          zw_mul(e, tv);
        }
        dentries[g].push_back(e);
      }
    }
    memcpy(&rec[GF_APRE], &h.approx_v[2 * (size_t)g], 4);
    memcpy(&rec[GF_APIM], &h.approx_v[2 * (size_t)g + 1], 4);
  }
  // ---- fixed frame: every term of the level comes straight from a table and, shifted to the
  //      level's smallest power, the worst-case sum of all graphs stays inside int32
  bool fixed = !h.approx && G > 0;
  int frame = INT32_MAX;
  if (fixed) {
    for (auto &ge : entries)
      for (auto &e : ge)
        if (e.c[0] | e.c[1] | e.c[2] | e.c[3]) frame = std::min(frame, e.p);
    if (frame == INT32_MAX) frame = 0;
    long double total = 0;
    for (int g = 0; g < G && fixed; ++g) {
      long double worst = 0;
      for (auto &e : entries[g]) {
        if (!(e.c[0] | e.c[1] | e.c[2] | e.c[3])) continue;
        const int sh = e.p - frame;
        if (sh > 40) { fixed = false; break; }
        // a rotation by i permutes/negates coefficients: bound by the largest one
        long double m = 0;
        for (auto v : e.c) m = std::max(m, (long double)std::llabs(v));
        worst = std::max(worst, m * (long double)(1ll << sh));
      }
      // separate PhasePairs table: entries become plain integers c * 2^p (p >= 0); the product
      // with the main entry is a sum of four coefficient products
      long double dworst = 0;
      for (auto &e : dentries[g]) {
        if (e.p < 0 || e.p > 40) { fixed = false; break; }
        long double m = 0;
        for (auto v : e.c) m = std::max(m, (long double)std::llabs(v));
        dworst = std::max(dworst, m * (long double)(1ll << e.p));
      }
      if (!dentries[g].empty()) worst = 4 * worst * dworst;
      total += worst;
    }
    if (total >= 2147483000.0L) fixed = false;
  }
  for (int g = 0; g < G; ++g) {
    uint32_t *rec = &h.graph_rec[(size_t)g * G_WORDS];
    rec[GF_TBL] = (uint32_t)tables.size();
    if (fixed) {
      // fixed-frame levels: an entry is 16 words = the value times i^r for r = 0..3 (4 words each),
      // pre-shifted to the frame power, so the kernel adds the selected rotation without any
      // per-lane rotate/shift.  Entry 0 is the exact zero.
      for (int j = 0; j < 16; ++j) tables.push_back(0u);
      for (auto &e : entries[g]) {
        const bool nz = (e.c[0] | e.c[1] | e.c[2] | e.c[3]) != 0;
        const int sh = nz ? e.p - frame : 0;
        long long v[4] = {e.c[0] * (1ll << sh), e.c[1] * (1ll << sh), e.c[2] * (1ll << sh), e.c[3] * (1ll << sh)};
        for (int r = 0; r < 4; ++r) {
          for (int j = 0; j < 4; ++j) tables.push_back((uint32_t)(int32_t)v[j]);
          const long long t[4] = {-v[2], v[3], v[0], -v[1]};  // times i: (a,b,c,d) -> (-c, d, a, -b)
          for (int j = 0; j < 4; ++j) v[j] = t[j];
        }
      }
    } else {
      for (int j = 0; j < 4; ++j) tables.push_back(0u);  // entry 0: exact zero
      tables.push_back((uint32_t)(h.approx ? 0 : TSIMK_ZERO_POWER));
      tables.push_back(0u); tables.push_back(0u); tables.push_back(0u);
      for (auto &e : entries[g]) {
        for (int j = 0; j < 4; ++j) tables.push_back((uint32_t)(int32_t)e.c[j]);
        tables.push_back((uint32_t)e.p);
        tables.push_back(0u); tables.push_back(0u); tables.push_back(0u);
      }
    }
  }
  for (int g = 0; g < G; ++g) {
    if (dentries[g].empty()) continue;
    uint32_t *rec = &h.graph_rec[(size_t)g * G_WORDS];
    rec[GF_TBL2] = (uint32_t)tables.size();
    for (auto &e : dentries[g]) {
      const int sh = fixed ? e.p : 0;
      for (int j = 0; j < 4; ++j) {
        const long long v = e.c[j] * (1ll << sh);
        if (v > INT32_MAX || v < INT32_MIN) return false;
        tables.push_back((uint32_t)(int32_t)v);
      }
      tables.push_back((uint32_t)(fixed ? 0 : e.p));
      tables.push_back(0u); tables.push_back(0u); tables.push_back(0u);
    }
  }
  fixed_out = fixed;
  frame_out = fixed ? frame : 0;
  h.fixed = fixed;
  h.frame = frame_out;
  return true;
}

}  // namespace

// ---------------------------------------------------------------------------
// v4 layout: per-tile 4-bit chunk tables (kernel: k_sample4, tsim_kernel4.hip.h)
// ---------------------------------------------------------------------------
namespace {

static bool level_v4_eligible(const HostLevel &h) {
  if (h.P > 64) return false;
  for (const FastGraph &fg : h.fg) {
    if (fg.us.size() > 32) return false;
    if (fg.c0.size() + fg.c1.size() + fg.c3.size() > 32) return false;
    if (2 * fg.nD > 28) return false;
  }
  return true;
}

static inline bool mask_bit(const std::vector<uint64_t> &m, int i) { return (m[(size_t)i >> 6] >> (i & 63)) & 1; }

// recs4: G x G4_WORDS; tabs4: ntiles x nch x 16 x GT x 4 words.  `v3recs` are the level's patched
// fast-layout graph records (for the term-table offsets and the approximate floatfactors).
// `stabs4` (optional, sparse_F >= 0): the "sparse f" table of the same tile: one entry per f COLUMN
// (bits 0..F-1), one all-zero entry, then 2 x 16 entries for the two 4-bit chunks of the output
// bits F..F+7 (row constants live in the first of them): [tile][entry][graph][4 words].
static void emit_level4(const HostLevel &h, int GT, int nch, const uint32_t *v3recs, std::vector<uint32_t> &recs4,
                        std::vector<uint32_t> &tabs4, int &nch_out, int &ntiles_out, int sparse_F,
                        std::vector<uint32_t> &stabs4) {
  const int G = h.G, P = h.P;
  const int ntiles = (G + GT - 1) / GT;  // nch: chunks per tile, the same for every level (zero padded)
  nch_out = nch;
  ntiles_out = ntiles;
  recs4.assign((size_t)G * G4_WORDS, 0u);
  tabs4.assign((size_t)ntiles * nch * 16 * GT * 4, 0u);
  const int sent = sparse_F >= 0 ? sparse_F + 1 + 32 : 0;
  stabs4.assign((size_t)ntiles * sent * GT * 4, 0u);
  std::vector<std::array<uint32_t, 4>> col((size_t)std::max(P, 1));
  for (int g = 0; g < G; ++g) {
    const FastGraph &fg = h.fg[(size_t)g];
    for (auto &c : col) c = {0u, 0u, 0u, 0u};
    std::array<uint32_t, 4> cst = {0u, 0u, 0u, 0u};
    auto place = [&](const std::vector<uint64_t> &m, int word, int bit) {
      for (int i = 0; i < P; ++i)
        if (mask_bit(m, i)) col[(size_t)i][(size_t)word] |= 1u << bit;
    };
    const int h2 = (int)fg.us.size();
    for (int s = 0; s < h2; ++s) {
      place(fg.us[(size_t)s], 0, s);
      place(fg.vs[(size_t)s], 1, s);
    }
    uint32_t M0 = 0, M1 = 0, M3 = 0;
    int b = 0;
    for (size_t t = 0; t < fg.c0.size(); ++t, ++b) { place(fg.c0[t], 2, b); M0 |= 1u << b; if (fg.c0c[t]) cst[2] |= 1u << b; }
    for (size_t t = 0; t < fg.c1.size(); ++t, ++b) { place(fg.c1[t], 2, b); M1 |= 1u << b; if (fg.c1c[t]) cst[2] |= 1u << b; }
    for (size_t t = 0; t < fg.c3.size(); ++t, ++b) { place(fg.c3[t], 2, b); M3 |= 1u << b; if (fg.c3c[t]) cst[2] |= 1u << b; }
    for (int t = 0; t < fg.nD; ++t) {  // index bits: term 0 holds the most significant pair
      place(fg.dal[(size_t)t], 3, 2 * (fg.nD - 1 - t));
      place(fg.dbe[(size_t)t], 3, 2 * (fg.nD - 1 - t) + 1);
    }
    place(fg.lam, 3, 30);
    place(fg.lin, 3, 31);
    const int tile = g / GT, j = g % GT;
    for (int c = 0; c < nch; ++c)
      for (int v = 0; v < 16; ++v) {
        std::array<uint32_t, 4> val = (c == 0) ? cst : std::array<uint32_t, 4>{0u, 0u, 0u, 0u};
        for (int bb = 0; bb < 4; ++bb) {
          const int i = 4 * c + bb;
          if (!((v >> bb) & 1) || i >= P) continue;
          for (int w = 0; w < 4; ++w) val[(size_t)w] ^= col[(size_t)i][(size_t)w];
        }
        uint32_t *dst = &tabs4[((((size_t)tile * nch + c) * 16 + v) * GT + j) * 4];
        for (int w = 0; w < 4; ++w) dst[w] = val[(size_t)w];
      }
    if (sent) {
      auto put = [&](int e, const std::array<uint32_t, 4> &val) {
        uint32_t *dst = &stabs4[(((size_t)tile * sent + e) * GT + j) * 4];
        for (int w = 0; w < 4; ++w) dst[w] = val[(size_t)w];
      };
      for (int i = 0; i < sparse_F && i < P; ++i) put(i, col[(size_t)i]);
      for (int c = 0; c < 2; ++c)
        for (int v = 0; v < 16; ++v) {
          std::array<uint32_t, 4> val = (c == 0) ? cst : std::array<uint32_t, 4>{0u, 0u, 0u, 0u};
          for (int bb = 0; bb < 4; ++bb) {
            const int i = sparse_F + 4 * c + bb;
            if (!((v >> bb) & 1) || i >= P) continue;
            for (int w = 0; w < 4; ++w) val[(size_t)w] ^= col[(size_t)i][(size_t)w];
          }
          put(sparse_F + 1 + 16 * c + v, val);
        }
    }
    uint32_t *r4 = &recs4[(size_t)g * G4_WORDS];
    const uint32_t *r3 = v3recs + (size_t)g * G_WORDS;
    r4[G4_M0] = M0; r4[G4_M1] = M1; r4[G4_M3] = M3;
    r4[G4_PM] = h2 >= 32 ? 0xFFFFFFFFu : ((1u << h2) - 1u);
    r4[G4_N1] = (uint32_t)fg.n1;
    r4[G4_DBITS] = (uint32_t)(2 * fg.nD);
    r4[G4_TBL] = r3[GF_TBL];
    r4[G4_TBL2] = r3[GF_TBL2];
    r4[G4_FLAGS] = fg.nD == 0 ? 0u : (fg.d_tabled ? TSIMK_G4FLAG_D_COMBINED : TSIMK_G4FLAG_D_SEPARATE);
    r4[G4_APRE] = r3[GF_APRE];
    r4[G4_APIM] = r3[GF_APIM];
  }
}

}  // namespace

extern "C" int tsim_program_set_mode(tsim_program *p, int32_t mode) {
  if (!p) return fail(TSIM_EINVAL, "program is NULL");
  if (p->finalized) return fail(TSIM_ESTATE, "program already finalized");
  if (mode != TSIM_MODE_AUTO && mode != TSIM_MODE_FAITHFUL && mode != TSIM_MODE_ROW_KERNEL)
    return fail(TSIM_EINVAL, "bad mode %d", mode);
  p->mode = mode;
  return TSIM_OK;
}

// Gather program (tsim_lw.hip.h): bit moves (src f bit -> dst bit, flip) merged into runs that are
// contiguous in both the source and the destination word; 4-word runs, padded to chunks of four.
static std::vector<uint32_t> emit_gather_program(std::vector<std::array<int, 3>> e) {
  std::sort(e.begin(), e.end(), [](const std::array<int, 3> &a, const std::array<int, 3> &b) {
    if ((a[0] >> 5) != (b[0] >> 5)) return (a[0] >> 5) < (b[0] >> 5);
    if ((a[1] >> 5) != (b[1] >> 5)) return (a[1] >> 5) < (b[1] >> 5);
    return a[0] < b[0];
  });
  std::vector<uint32_t> out;
  size_t i = 0;
  while (i < e.size()) {
    size_t j = i + 1;
    while (j < e.size() && e[j][0] == e[j - 1][0] + 1 && e[j][1] == e[j - 1][1] + 1 &&
           (e[j][0] >> 5) == (e[i][0] >> 5) && (e[j][1] >> 5) == (e[i][1] >> 5))
      ++j;
    const int len = (int)(j - i);
    uint32_t flip = 0;
    for (size_t k = i; k < j; ++k) flip |= (uint32_t)(e[k][2] & 1) << (k - i);
    const uint32_t mask = len >= 32 ? 0xFFFFFFFFu : ((1u << len) - 1u);
    out.push_back((uint32_t)(e[i][0] & 31) | ((uint32_t)(e[i][1] & 31) << 8) | ((uint32_t)(e[i][1] >> 5) << 16) |
                  ((uint32_t)(e[i][0] >> 5) << 24));
    out.push_back(mask);
    out.push_back(flip);
    out.push_back(0u);
    i = j;
  }
  while (out.size() % 16) out.push_back(0u);  // mask = 0 runs: no-ops
  return out;
}

extern "C" int tsim_program_set_pattern_tables(tsim_program *p, int32_t enable, int32_t max_weight) {
  if (!p) return fail(TSIM_EINVAL, "program is NULL");
  if (p->finalized) return fail(TSIM_ESTATE, "program already finalized");
  if (enable < -1 || enable > 1 || max_weight < -1 || max_weight > TSIMK_LW_MAX_WEIGHT)
    return fail(TSIM_EINVAL, "bad pattern-table setting (%d, %d)", enable, max_weight);
  p->lw_request = enable;
  p->lw_weight_cap = max_weight;
  return TSIM_OK;
}

extern "C" int tsim_program_pattern_table_info(const tsim_program *p, int32_t *enabled, int64_t *table_bytes,
                                               int32_t *max_weight) {
  if (!p) return fail(TSIM_EINVAL, "program is NULL");
  if (!p->finalized) return fail(TSIM_ESTATE, "program not finalized");
  if (enabled) *enabled = p->lw ? 1 : 0;
  if (table_bytes) *table_bytes = p->lw ? p->lw_bytes : 0;
  if (max_weight)
    for (size_t i = 0; i < p->comps.size(); ++i) max_weight[i] = p->lw ? p->lw_wmax[i] : -1;
  return TSIM_OK;
}

// ---------------------------------------------------------------------------
// low-weight pattern tables (tsim_lw.hip.h): enumerate the patterns of every component in table
// order and let k_lw_build fill the thresholds with the sampling kernels' own arithmetic
// ---------------------------------------------------------------------------
template <int W>
static void launch_lw_build(const LwBuildArgs &a, long long lanes, hipStream_t s, bool fast) {
  const dim3 grid((unsigned)((lanes + 255) / 256));
  if (fast) hipLaunchKernelGGL((k_lw_build<W, true>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((k_lw_build<W, false>), grid, dim3(256), 0, s, a);
}

static int build_pattern_tables(tsim_program *p, const std::vector<long long> &npat) {
  HIP_TRY(hipMalloc((void **)&p->d_lw_tab, std::max<size_t>(16, (size_t)p->lw_bytes)));
  long long tab_off = 0;
  for (size_t ci = 0; ci < p->comps.size(); ++ci) {
    const HostComponent &c = p->comps[ci];
    const int F = c.F, wmax = p->lw_wmax[ci];
    std::vector<unsigned long long> pats;
    pats.reserve((size_t)npat[ci]);
    pats.push_back(0ull);
    if (wmax >= 1)
      for (int b0 = 0; b0 < F; ++b0) pats.push_back(1ull << b0);
    if (wmax >= 2)
      for (int b1 = 1; b1 < F; ++b1)
        for (int b0 = 0; b0 < b1; ++b0) pats.push_back((1ull << b1) | (1ull << b0));
    if (wmax >= 3)
      for (int b2 = 2; b2 < F; ++b2)
        for (int b1 = 1; b1 < b2; ++b1)
          for (int b0 = 0; b0 < b1; ++b0) pats.push_back((1ull << b2) | (1ull << b1) | (1ull << b0));
    if (wmax >= 4)
      for (int b3 = 3; b3 < F; ++b3)
        for (int b2 = 2; b2 < b3; ++b2)
          for (int b1 = 1; b1 < b2; ++b1)
            for (int b0 = 0; b0 < b1; ++b0)
              pats.push_back((1ull << b3) | (1ull << b2) | (1ull << b1) | (1ull << b0));
    if (wmax >= 5)
      for (int b4 = 4; b4 < F; ++b4)
        for (int b3 = 3; b3 < b4; ++b3)
          for (int b2 = 2; b2 < b3; ++b2)
            for (int b1 = 1; b1 < b2; ++b1)
              for (int b0 = 0; b0 < b1; ++b0)
                pats.push_back((1ull << b4) | (1ull << b3) | (1ull << b2) | (1ull << b1) | (1ull << b0));
    if ((long long)pats.size() != npat[ci]) return fail(TSIM_ESTATE, "pattern enumeration mismatch");
    unsigned long long *d_pats = nullptr;
    HIP_TRY(hipMalloc((void **)&d_pats, pats.size() * 8));
    HIP_TRY(hipMemcpy(d_pats, pats.data(), pats.size() * 8, hipMemcpyHostToDevice));
    LwBuildArgs a;
    a.img = p->d_img;
    a.patbits = d_pats;
    a.tab = p->d_lw_tab + tab_off;
    a.comp_off = p->comp_off + (int)ci * C_WORDS;
    a.npat = (int)npat[ci];
    const long long lanes = npat[ci] << c.n_out;
    switch (p->comp_w[ci]) {
      case 1: launch_lw_build<1>(a, lanes, p->stream, p->fast); break;
      case 2: launch_lw_build<2>(a, lanes, p->stream, p->fast); break;
      default: (void)hipFree(d_pats); return fail(TSIM_ESTATE, "pattern tables need <= 64 parameters");
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
    (void)hipFree(d_pats);
    if (e != hipSuccess) return fail(TSIM_EHIP, "pattern table build failed: %s", hipGetErrorString(e));
    tab_off += lanes;
  }
  if (p->v4) {
    void *h = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocMapped) == hipSuccess) {
      void *d = nullptr;
      if (hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
        p->h_feedback = (volatile uint32_t *)h;
        p->d_feedback = (uint32_t *)d;
        for (int i = 0; i < 16; ++i) p->h_feedback[i] = 0xFFFFFFFFu;
      } else {
        (void)hipHostFree(h);
      }
    }
    (void)hipGetLastError();  // feedback is optional: without it every launch takes the default plan
  }
  return 0;
}

extern "C" int tsim_program_finalize(tsim_program *p, int32_t device) {
  if (!p) return fail(TSIM_EINVAL, "program is NULL");
  if (p->finalized) return fail(TSIM_ESTATE, "program already finalized");
  // ---- choose the evaluation formulation ----
  {
    const char *env = getenv("TSIM_AMD_MODE");
    bool fast = (p->mode != TSIM_MODE_FAITHFUL) && !(env && strcmp(env, "faithful") == 0);
    for (auto &c : p->comps)
      for (auto &lv : c.levels) fast = fast && level_fast_eligible(lv);
    p->fast = fast;
  }
retry_pack:
  // ---- validate the output bookkeeping (pipeline.py:83-102) ----
  int pos = p->n_direct;
  for (size_t ci = 0; ci < p->comps.size(); ++ci) {
    HostComponent &c = p->comps[ci];
    if ((int)c.levels.size() != c.n_levels)
      return fail(TSIM_EINVAL, "component %zu has %zu of %d levels", ci, c.levels.size(), c.n_levels);
    for (int j = 0; j < c.n_out; ++j) {
      if (pos >= p->num_outputs || p->output_order[pos] != c.output_indices[j])
        return fail(TSIM_EINVAL, "output_order[%d] does not match component %zu output %d", pos, ci, j);
      ++pos;
    }
  }
  if (pos != p->num_outputs)
    return fail(TSIM_EINVAL, "direct entries + component outputs cover %d of %d outputs", pos, p->num_outputs);

  // ---- build the image ----
  std::vector<uint32_t> &img = p->img;
  img.clear();
  img.resize(16, 0u);  // word 0..15 reserved (keeps every offset non-zero)
  p->direct_off = (int)img.size();
  for (int j = 0; j < p->n_direct; ++j) {
    img.push_back((uint32_t)p->direct_f[j] | ((p->direct_flips[j] ? 1u : 0u) << 31));
    img.push_back((uint32_t)p->output_order[j]);
  }
  p->comp_off = (int)img.size();
  img.resize(img.size() + p->comps.size() * C_WORDS, 0u);
  p->total_keys = 0;
  p->sampleable = true;
  p->total_graphs = p->total_rows = 0;
  for (auto &v : p->stats) v = 0;
  p->level_off.clear();
  p->level_base.clear();
  p->comp_w.clear();
  for (size_t ci = 0; ci < p->comps.size(); ++ci) {
    HostComponent &c = p->comps[ci];
    int maxP = 1;
    for (auto &lv : c.levels) maxP = std::max(maxP, lv.P);
    // sampling appends the trial bit at position F+i (< F+n_out)
    const bool sequential = (c.n_levels == c.n_out + 1);
    if (sequential) maxP = std::max(maxP, c.F + c.n_out);
    const int W = round_w((maxP + 31) / 32);
    if (W < 0) return fail(TSIM_ENOTSUP, "component %zu needs %d parameter bits (max %d)", ci, maxP, TSIM_MAX_PARAMS);
    p->comp_w.push_back(W);
    if (!sequential && c.n_out != 1) p->sampleable = false;
    uint32_t rec[C_WORDS] = {0};
    rec[C_NOUT] = (uint32_t)c.n_out;
    rec[C_F] = (uint32_t)c.F;
    rec[C_W] = (uint32_t)W;
    rec[C_NLEVELS] = (uint32_t)c.n_levels;
    rec[C_KEYBASE] = (uint32_t)p->total_keys;
    rec[C_FSEL] = (uint32_t)img.size();
    for (int v : c.f_selection) img.push_back((uint32_t)v);
    rec[C_OUTPOS] = (uint32_t)img.size();
    for (int v : c.output_indices) img.push_back((uint32_t)v);
    // level records, then graph records + rows of every level
    rec[C_LEVELS] = (uint32_t)img.size();
    const size_t lrec = img.size();
    img.resize(img.size() + (size_t)c.n_levels * L_WORDS, 0u);
    p->level_base.push_back((int)p->level_off.size());
    for (int k = 0; k < c.n_levels; ++k) {
      HostLevel &h = c.levels[k];
      std::vector<uint32_t> tables;
      bool fixed = false;
      int frame = 0;
      if (p->fast) {
        if (!pack_level_fast(h, W, tables, fixed, frame)) {  // a table entry exceeds int32: use the faithful layout
          p->fast = false;
          goto retry_pack;
        }
      } else {
        pack_level(h, W);
      }
      // align graph records to 16 words (one s_load_dwordx16 each)
      while (img.size() % 16) img.push_back(0u);
      const uint32_t goff = (uint32_t)img.size();
      img.insert(img.end(), h.graph_rec.begin(), h.graph_rec.end());
      const uint32_t roff = (uint32_t)img.size();
      img.insert(img.end(), h.rows.begin(), h.rows.end());
      static_assert((int)G_ROWS == (int)GF_ROWS, "row offset slot is shared by both layouts");
      for (int g = 0; g < h.G; ++g) img[goff + (size_t)g * G_WORDS + G_ROWS] += roff;
      if (p->fast) {
        while (img.size() % 16) img.push_back(0u);  // 64-byte aligned table entries (uint4 loads)
        const uint32_t toff = (uint32_t)img.size();
        img.insert(img.end(), tables.begin(), tables.end());
        for (int g = 0; g < h.G; ++g) {
          img[goff + (size_t)g * G_WORDS + GF_TBL] += toff;
          if (img[goff + (size_t)g * G_WORDS + GF_TBL2]) img[goff + (size_t)g * G_WORDS + GF_TBL2] += toff;
        }
      }
      uint32_t *lr = &img[lrec + (size_t)k * L_WORDS];
      lr[L_G] = (uint32_t)h.G;
      lr[L_GRAPHS] = goff;
      lr[L_FLAGS] = (h.approx ? TSIMK_LFLAG_APPROX : 0u) | (fixed ? TSIMK_LFLAG_FIXED : 0u);
      lr[L_FRAME] = (uint32_t)frame;
      p->stats[1] += 1;
      p->stats[2] += fixed ? 1 : 0;
      p->stats[5] += (long long)tables.size() * 4;  // table bytes
      if (p->fast)
        for (int g = 0; g < h.G; ++g) {
          const uint32_t *r = &h.graph_rec[(size_t)g * G_WORDS];
          p->stats[3] += r[GF_N3H] >> 16;
          p->stats[4] += (r[GF_N01] & 0xFFFF) + (r[GF_N01] >> 16) + (r[GF_N3H] & 0xFFFF);
          p->stats[6] += (r[GF_FLAGS] & TSIMK_GFLAG_D_TABLED) ? 1 : 0;
        }
      lr[L_NPARAMS] = (uint32_t)h.P;
      p->level_off.push_back((int)(lrec + (size_t)k * L_WORDS));
      p->total_graphs += h.G;
      p->total_rows += h.n_rows;
    }
    if (sequential) p->total_keys += c.n_out;
    memcpy(&img[p->comp_off + ci * C_WORDS], rec, sizeof rec);
  }
  // ---- v4 (chunk table) layout, when every sampled component qualifies ----
  p->v4 = false;
  if (p->fast && p->sampleable) {
    bool ok = !p->comps.empty();
    for (auto &c : p->comps)
      for (auto &lv : c.levels) ok = ok && level_v4_eligible(lv);
    const char *kenv = getenv("TSIM_AMD_KERNEL");
    if (kenv && strcmp(kenv, "v3") == 0) ok = false;
    if (p->mode == TSIM_MODE_ROW_KERNEL) ok = false;
    p->v4_gt = 4;
    if (ok) {
      while (img.size() % 16) img.push_back(0u);
      p->comp4_off = (int)img.size();
      img.resize(img.size() + p->comps.size() * C4_WORDS, 0u);
      p->v4_max_sent = 0;
      int maxp = 1;
      for (auto &c : p->comps)
        for (auto &lv : c.levels) maxp = std::max(maxp, lv.P);
      static const int kNch[] = {2, 4, 6, 8, 10, 12, 14, 16};
      p->v4_max_nch = 16;
      for (int v : kNch)
        if (4 * v >= maxp) { p->v4_max_nch = v; break; }
      for (size_t ci = 0; ci < p->comps.size(); ++ci) {
        HostComponent &c = p->comps[ci];
        for (int w = 0; w < 8; ++w) img[p->comp4_off + ci * C4_WORDS + w] = img[p->comp_off + ci * C_WORDS + w];
        while (img.size() % 16) img.push_back(0u);
        const size_t l4 = img.size();
        img[p->comp4_off + ci * C4_WORDS + C4_LEVELS] = (uint32_t)l4;
        img.resize(img.size() + (size_t)c.n_levels * L4_WORDS, 0u);
        for (int k = 0; k < c.n_levels; ++k) {
          HostLevel &h = c.levels[k];
          const uint32_t v3lvl = (uint32_t)p->level_off[p->level_base[ci] + k];
          const uint32_t v3recs = img[v3lvl + L_GRAPHS];
          std::vector<uint32_t> recs4, tabs4, stabs4;
          int nch = 1, ntiles = 0;
          const bool sequential = (c.n_levels == c.n_out + 1);
          const int sparse_F = (sequential && c.n_out <= 8 && c.F + c.n_out <= 64) ? c.F : -1;
          std::vector<uint32_t> v3copy(img.begin() + v3recs, img.begin() + v3recs + (size_t)h.G * G_WORDS);
          emit_level4(h, p->v4_gt, p->v4_max_nch, v3copy.data(), recs4, tabs4, nch, ntiles, sparse_F, stabs4);
          while (img.size() % 16) img.push_back(0u);
          const uint32_t roff = (uint32_t)img.size();
          img.insert(img.end(), recs4.begin(), recs4.end());
          while (img.size() % 16) img.push_back(0u);
          const uint32_t toff = (uint32_t)img.size();
          img.insert(img.end(), tabs4.begin(), tabs4.end());
          while (img.size() % 16) img.push_back(0u);
          const uint32_t stoff = stabs4.empty() ? 0u : (uint32_t)img.size();
          img.insert(img.end(), stabs4.begin(), stabs4.end());
          p->v4_max_sent = std::max(p->v4_max_sent, sparse_F >= 0 ? sparse_F + 33 : 0);
          uint32_t *lr = &img[l4 + (size_t)k * L4_WORDS];
          lr[L4_STAB] = stoff;
          lr[L4_G] = (uint32_t)h.G;
          lr[L4_NTILES] = (uint32_t)ntiles;
          lr[L4_TABLES] = toff;
          lr[L4_RECS] = roff;
          lr[L4_NCH] = (uint32_t)nch;
          lr[L4_FLAGS] = (h.approx ? TSIMK_LFLAG_APPROX : 0u) | (h.fixed ? TSIMK_LFLAG_FIXED : 0u);
          lr[L4_FRAME] = (uint32_t)h.frame;
        }
      }
      p->v4 = true;
    }
  }
  p->stats[7] = p->v4 ? 1 : 0;

  // ---- low-weight pattern tables: plan (records + sizes); built on the device after upload ----
  p->lw = false;
  p->lw_wmax.clear();
  p->lw_bytes = 0;
  std::vector<long long> lw_npat;
  std::vector<std::vector<std::array<int, 3>>> lw_fsel_progs;
  {
    bool want = p->lw_request < 0 ? (p->mode == TSIM_MODE_AUTO) : (p->lw_request != 0);
    if (const char *e = getenv("TSIM_AMD_PATTERN_TABLES")) want = atoi(e) != 0;
    bool ok = want && p->sampleable && !p->comps.empty();
    for (auto &c : p->comps)
      ok = ok && (c.n_levels == c.n_out + 1) && c.n_out <= TSIMK_LW_MAX_NOUT && c.F + c.n_out <= 64;
    if (ok) {
      const int cap = p->lw_weight_cap < 0 ? TSIMK_LW_MAX_WEIGHT : std::min(p->lw_weight_cap, TSIMK_LW_MAX_WEIGHT);
      // bytes per component.  Patterns are stored weight by weight, so the rows most shots read (weight
      // 0..2) are a small cache-resident prefix whatever the total; the heavier tail is read rarely
      long long budget = 256ll << 20;
      if (const char *e = getenv("TSIM_AMD_PATTERN_TABLE_MB")) budget = std::max(1ll, atoll(e)) << 20;
      while (img.size() % 16) img.push_back(0u);
      p->lw_off = (int)img.size();
      img.resize(img.size() + p->comps.size() * LW_WORDS, 0u);
      long long tab_off = 0;
      for (size_t ci = 0; ci < p->comps.size(); ++ci) {
        const HostComponent &c = p->comps[ci];
        const long long F = c.F;
        const long long cnt[6] = {1, F, F * (F - 1) / 2, F * (F - 1) * (F - 2) / 6,
                                  F * (F - 1) * (F - 2) * (F - 3) / 24,
                                  F * (F - 1) * (F - 2) * (F - 3) * (F - 4) / 120};
        long long npat = 0;
        int wmax = -1;
        for (int w = 0; w <= cap; ++w) {
          const long long bytes = ((npat + cnt[w]) << c.n_out) * 4;
          if (bytes > budget || (w > 1 && tab_off * 4 + bytes > 4 * budget)) break;  // per component / all together
          npat += cnt[w];
          wmax = w;
        }
        if (wmax < 0) { ok = false; break; }
        const uint32_t *crec = &img[p->comp_off + ci * C_WORDS];
        uint32_t *r = &img[p->lw_off + ci * LW_WORDS];
        r[LW_NOUT] = (uint32_t)c.n_out;
        r[LW_F] = (uint32_t)c.F;
        lw_fsel_progs.push_back({});
        for (int j = 0; j < c.F; ++j) lw_fsel_progs.back().push_back({c.f_selection[j], j, 0});
        r[LW_OUTPOS] = crec[C_OUTPOS];
        r[LW_KEYBASE] = crec[C_KEYBASE];
        r[LW_WMAX] = (uint32_t)wmax;
        r[LW_TAB] = (uint32_t)tab_off;
        r[LW_OFF2] = (uint32_t)(1 + F);
        r[LW_OFF3] = (uint32_t)(1 + F + cnt[2]);
        r[LW_OFF4] = (uint32_t)(1 + F + cnt[2] + cnt[3]);
        r[LW_OFF5] = (uint32_t)(1 + F + cnt[2] + cnt[3] + cnt[4]);
        r[LW_NPAT] = (uint32_t)npat;
        p->lw_wmax.push_back(wmax);
        lw_npat.push_back(npat);
        tab_off += npat << c.n_out;
      }
      if (ok) {
        p->lw = true;
        p->lw_bytes = tab_off * 4;
        // gather programs: direct outputs, then every component's f_sel
        std::vector<std::array<int, 3>> de;
        for (int j = 0; j < p->n_direct; ++j) de.push_back({p->direct_f[j], p->output_order[j], p->direct_flips[j] ? 1 : 0});
        std::vector<uint32_t> prog = emit_gather_program(de);
        while (img.size() % 16) img.push_back(0u);
        p->lw_direct_prog = (int)img.size();
        p->lw_direct_chunks = (int)(prog.size() / 16);
        img.insert(img.end(), prog.begin(), prog.end());
        for (size_t ci = 0; ci < p->comps.size(); ++ci) {
          prog = emit_gather_program(lw_fsel_progs[ci]);
          img[p->lw_off + ci * LW_WORDS + LW_FSELP] = (uint32_t)img.size();
          img[p->lw_off + ci * LW_WORDS + LW_FSELN] = (uint32_t)(prog.size() / 16);
          img.insert(img.end(), prog.begin(), prog.end());
        }
      } else {
        p->lw_wmax.clear();
      }
    }
  }
  img.resize(img.size() + 256, 0u);  // tail padding: wide scalar loads may over-read
  if (img.size() >= (1ull << 31)) return fail(TSIM_ENOTSUP, "program image too large");

  {
    auto env_int = [](const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; };
    p->knobs.adaptive = env_int("TSIM_AMD_ADAPTIVE", 1) != 0;
    p->knobs.hard_kernel = env_int("TSIM_AMD_HARD_KERNEL", 1) != 0;
    p->knobs.lane0_main = env_int("TSIM_AMD_LANE0_MAIN", 1) != 0;
    p->knobs.lw_block = env_int("TSIM_AMD_LW_BLOCK", 0);
    if (p->knobs.lw_block != 0) p->knobs.lw_block = std::max(64, std::min(1024, p->knobs.lw_block & ~63));
    const int vb = env_int("TSIM_AMD_V4_BLOCK", 256);
    p->knobs.v4_block = (vb == 512 || vb == 128) ? vb : 256;
    p->knobs.hard_lds_kb = std::max(24, std::min(156, env_int("TSIM_AMD_HARD_LDS_KB", 128)));
    p->knobs.defer = env_int("TSIM_AMD_DEFER_HARD", 1) != 0;
    p->knobs.merge_lists = env_int("TSIM_AMD_MERGE_LISTS", 1) != 0;
    p->knobs.list_rows = env_int("TSIM_AMD_LIST_ROWS", 40);
    p->knobs.min_lists = env_int("TSIM_AMD_MIN_LISTS", 4);
    if (p->knobs.min_lists & (p->knobs.min_lists - 1)) p->knobs.min_lists = 4;
    // launches per deferred batch: a batch lasts about as long as ONE hard-row pass (its blocks run side by
    // side), and batches are serial on their lane - so the batch must cover at least (pass time / step time)
    // launches.  The pass time grows with the chunk tables a 64-row block streams through LDS (C2: 1.7 MB,
    // 35-50 us; C4: 10.8 MB, 450 us): 4 launches for small programs, 8 (the kernel's limit) beyond 4 MB.
    p->knobs.defer_group = p->stats[5] > (4ll << 20) ? TSIMK_H_MAX_CTX : 4;
    p->knobs.defer_group = std::max(1, std::min(TSIMK_H_MAX_CTX, env_int("TSIM_AMD_DEFER_GROUP", p->knobs.defer_group)));
  }

  // ---- upload ----
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(TSIM_EINVAL, "device %d out of range (%d visible)", device, ndev);
  p->device = device;
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
  HIP_TRY(hipMalloc((void **)&p->d_img, img.size() * 4));
  HIP_TRY(hipMemcpy(p->d_img, img.data(), img.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc((void **)&p->d_dev, std::max<size_t>(1, p->comps.size()) * 4));
  HIP_TRY(hipMemset(p->d_dev, 0, std::max<size_t>(1, p->comps.size()) * 4));
  if (p->lw) {
    if (int r = build_pattern_tables(p, lw_npat)) return r;
  }
  p->finalized = true;
  return TSIM_OK;
}

static int flush_hard(tsim_program *p);

extern "C" void tsim_program_destroy(tsim_program *p) {
  if (!p) return;
  if (p->finalized && p->device >= 0) {
    (void)hipSetDevice(p->device);
    (void)flush_hard(p);  // parked hard rows of launches that were never joined: finish them, then drain every lane
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    for (auto &sl : p->slots)
      if (sl.side_ready) (void)hipStreamSynchronize(sl.side);
    for (hipEvent_t e : p->ev_pool) (void)hipEventDestroy(e);
    if (getenv("TSIM_AMD_PIPELINE_STATS"))
      fprintf(stderr, "tsim pipeline: begins %llu deferred %llu flushes %llu queries %llu waits %llu\n", p->stat_begins,
              p->stat_deferred, p->stat_flushes, p->stat_queries, p->stat_waits);
    if (p->sync_ev) (void)hipEventDestroy(p->sync_ev);
    for (hipEvent_t e : p->lane_ev) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : p->batch_ev) if (e) (void)hipEventDestroy(e);
    for (void *s : p->scratch)
      if (s) (void)hipFree(s);
    if (p->d_img) (void)hipFree(p->d_img);
    if (p->d_dev) (void)hipFree(p->d_dev);
    if (p->d_lw_tab) (void)hipFree(p->d_lw_tab);
    if (p->h_feedback) (void)hipHostFree((void *)p->h_feedback);
    for (auto &sl : p->slots) {
      if (sl.ctl) (void)hipFree(sl.ctl);
      if (sl.hard) (void)hipFree(sl.hard);
      if (sl.keys) (void)hipFree(sl.keys);
      if (sl.ev1) (void)hipEventDestroy(sl.ev1);
      if (sl.ev2) (void)hipEventDestroy(sl.ev2);

      if (sl.side && !sl.side_borrowed) (void)hipStreamDestroy(sl.side);
    }
    if (p->stream) (void)hipStreamDestroy(p->stream);
  }
  delete p;
}

extern "C" int tsim_program_stats(const tsim_program *p, int64_t out[8]) {
  if (!p || !out) return fail(TSIM_EINVAL, "NULL argument");
  if (!p->finalized) return fail(TSIM_ESTATE, "program not finalized");
  for (int i = 0; i < 8; ++i) out[i] = p->stats[i];
  out[0] = p->fast ? 1 : 0;
  return TSIM_OK;
}

extern "C" int tsim_program_get_mode(const tsim_program *p, int32_t *fast) {
  if (!p || !fast) return fail(TSIM_EINVAL, "NULL argument");
  if (!p->finalized) return fail(TSIM_ESTATE, "program not finalized");
  *fast = p->fast ? 1 : 0;
  return TSIM_OK;
}

extern "C" int tsim_program_info(const tsim_program *p, int32_t *n_components, int32_t *num_outputs,
                                 int64_t *image_bytes, int64_t *total_graphs, int64_t *total_rows) {
  if (!p) return fail(TSIM_EINVAL, "program is NULL");
  if (n_components) *n_components = (int32_t)p->comps.size();
  if (num_outputs) *num_outputs = p->num_outputs;
  if (image_bytes) *image_bytes = (int64_t)p->img.size() * 4;
  if (total_graphs) *total_graphs = p->total_graphs;
  if (total_rows) *total_rows = p->total_rows;
  return TSIM_OK;
}

// ---------------------------------------------------------------------------
// launches
// ---------------------------------------------------------------------------
static int need_final(const tsim_program *p) {
  if (!p) return fail(TSIM_EINVAL, "program is NULL");
  if (!p->finalized) return fail(TSIM_ESTATE, "program not finalized");
  return 0;
}

static int ensure_scratch(tsim_program *p, int slot, size_t bytes) {
  if (p->scratch_sz[slot] >= bytes) return 0;
  if (p->scratch[slot]) {
    HIP_TRY(hipStreamSynchronize(p->stream));
    HIP_TRY(hipFree(p->scratch[slot]));
    p->scratch[slot] = nullptr;
    p->scratch_sz[slot] = 0;
  }
  size_t cap = std::max<size_t>(bytes, 256);
  hipError_t e = hipMalloc(&p->scratch[slot], cap);
  if (e != hipSuccess) return fail(TSIM_ENOMEM, "hipMalloc(%zu) failed: %s", cap, hipGetErrorString(e));
  p->scratch_sz[slot] = cap;
  return 0;
}

// stage tags of the profiling events: 0 opens a launch, the others close a stage
enum { PROF_BEGIN = 0, PROF_PASS1 = 1, PROF_HARD = 2, PROF_FULL = 3 };

static int prof_event(tsim_program *p, hipStream_t s, int tag) {
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
static int slot_prepare(tsim_program *p, int slot, size_t hard_bytes) {
  tsim_program::Slot &sl = p->slots[slot];
  if (p->lw && !sl.ctl) {
    // two counter sets used alternately: pass 1 of a launch resets the set of the slot's next one
    const size_t set_bytes = (TSIMK_LW_LISTS + 1) * 128;
    HIP_TRY(hipMalloc((void **)&sl.ctl, 2 * set_bytes));
    HIP_TRY(hipMemset(sl.ctl, 0, 2 * set_bytes));
    for (int st = 0; st < 2; ++st)
      HIP_TRY(hipMemset(sl.ctl + st * (TSIMK_LW_LISTS + 1) * 32 + TSIMK_LW_LISTS * 32, 0xFF, 4));
  }
  if (p->total_keys > TSIMK_INLINE_KEYS && !sl.keys) HIP_TRY(hipMalloc((void **)&sl.keys, (size_t)p->total_keys * 8));
  if (slot > 0 && !sl.side_ready) {
    sl.side_ready = true;
    // default priority on purpose: a low- (or high-) priority lane lands on a different class of
    // hardware queue and tripled the step time (134 us vs 43 us, measured)
    if (slot == 1 && p->knobs.lane0_main) {
      // The handle's own stream doubles as the first lane: HIP gave the lanes it created only two distinct
      // hardware queues (kernel trace: three created streams -> queues 3, 4, 4), the handle's stream sits on
      // a third one.  Three truly concurrent lanes: 36 us per step instead of 42.
      sl.side = p->stream;
      sl.side_borrowed = true;
    } else {
      HIP_TRY(hipStreamCreateWithFlags(&sl.side, hipStreamNonBlocking));
    }
    HIP_TRY(hipEventCreateWithFlags(&sl.ev1, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&sl.ev2, hipEventDisableTiming));
  }
  if (sl.hard_sz < hard_bytes) {
    if (sl.hard) {
      HIP_TRY(hipDeviceSynchronize());
      HIP_TRY(hipFree(sl.hard));
      sl.hard = nullptr;
      sl.hard_sz = 0;
    }
    hipError_t e = hipMalloc(&sl.hard, hard_bytes);
    if (e != hipSuccess) return fail(TSIM_ENOMEM, "hipMalloc(%zu) failed: %s", hard_bytes, hipGetErrorString(e));
    sl.hard_sz = hard_bytes;
  }
  return 0;
}

// Launch plan from the feedback of earlier launches (results do not depend on it):
//  * most rows hard (dense error patterns): the pattern pass is wasted work - run the full kernel
//    on every row for the next 15 launches, then probe again with one two-pass launch;
//  * hard-row lists short: k_sample4h walks them alone, no overflow launch of k_sample4 - and a
//    pipelined launch may leave its hard rows to a later batch (flush_hard) instead of making its
//    lane wait for them.
struct LaunchPlan {
  bool use_tables = false, need_overflow = true, defer = false;
  uint32_t fb_max = 0xFFFFFFFFu;
  int lists = TSIMK_LW_LISTS;  // hard-row sub-lists of this launch: about 40 expected rows each
};

static LaunchPlan make_plan(tsim_program *p, bool has_row_index, bool pipelined) {
  LaunchPlan pl;
  pl.use_tables = p->lw;
  if (p->lw && p->h_feedback && p->knobs.adaptive) {
    const uint32_t fb_sum = p->h_feedback[0], fb_max = p->h_feedback[1], fb_rows = p->h_feedback[2];
    const bool known = fb_rows != 0xFFFFFFFFu && fb_rows > 0u && fb_sum != 0xFFFFFFFFu;
    bool dense = false;
    if (p->lw_direct_left > 0) {
      --p->lw_direct_left;
      pl.use_tables = false;
    } else if (known && (double)fb_sum > 0.5 * (double)fb_rows && !has_row_index) {
      p->lw_direct_left = 15;  // this launch is the probe
      dense = true;
    }
    if (known && fb_max <= 192u) pl.need_overflow = false;
    pl.fb_max = known ? fb_max : 0xFFFFFFFFu;
    if (known && p->knobs.merge_lists) {
      const uint32_t per = (uint32_t)std::max(8, p->knobs.list_rows);
      const uint32_t want = (fb_sum + per - 1u) / per;
      pl.lists = std::max(2, std::min(TSIMK_LW_LISTS, p->knobs.min_lists));
      while ((uint32_t)pl.lists < want && pl.lists < TSIMK_LW_LISTS) pl.lists <<= 1;
      // the longest list of the last launch was measured with ITS list count: rescale the overflow test
      const uint32_t last = p->last_lists > 0 ? (uint32_t)p->last_lists : (uint32_t)TSIMK_LW_LISTS;
      const uint32_t est_max = (uint32_t)std::min<unsigned long long>(0xFFFFFFFFull, (unsigned long long)fb_max * last / (uint32_t)pl.lists + 16u);
      pl.need_overflow = !(pl.lists >= (int)last ? fb_max <= 192u : est_max <= 192u);
    }
    pl.defer = pipelined && pl.use_tables && !dense && !pl.need_overflow && p->knobs.defer && p->knobs.hard_kernel &&
               p->v4 && !(p->profiling && !p->prof_light);
  }
  return pl;
}

// k_sample4h geometry (LDS budget -> tiles per group), 0 tiles = the kernel cannot run this program
static void hard_geometry(tsim_program *p, int WF, int WO) {
  constexpr int NW = TSIM_HARD_NW;
  const size_t tile_b = (size_t)p->v4_max_nch * 16 * p->v4_gt * 16;
  const size_t fixed_b = (size_t)(2 * WF + 2 * WO) * 64 * 4 + (size_t)NW * 8 * 64 * 4;
  const size_t budget = (size_t)p->knobs.hard_lds_kb * 1024;
  p->h_group_tiles = fixed_b + tile_b <= budget ? (int)std::min<size_t>(TSIMK_H_MAX_GROUP_TILES, (budget - fixed_b) / tile_b) : 0;
  p->h_lds = fixed_b + (size_t)std::max(1, p->h_group_tiles) * tile_b;
}

// The deferred second pass: ONE k_sample4h_multi grid serves the hard rows of every launch whose
// first pass is enqueued, on the third lane's stream, after those first passes.
static int flush_hard(tsim_program *p) {
  if (p->deferred.empty()) return 0;
  constexpr int NW = TSIM_HARD_NW;
  hipStream_t hs = p->slots[3].side;
  Hard4Multi M{};
  M.n_ctx = (int)p->deferred.size();
  const uint32_t fb_max = p->h_feedback ? p->h_feedback[1] : 192u;
  const int hb = (int)std::max(1u, std::min(4u, (std::min(fb_max, 192u) + 32u + 63u) / 64u));
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
    if (lane_used[k]) {
      if (!p->lane_ev[k]) HIP_TRY(hipEventCreateWithFlags(&p->lane_ev[k], hipEventDisableTiming));
      HIP_TRY(hipEventRecord(p->lane_ev[k], p->slots[1 + k].side));
      HIP_TRY(hipStreamWaitEvent(hs, p->lane_ev[k], 0));
    }
  const unsigned grid = (unsigned)(M.n_ctx * M.blocks_per_ctx);
  switch (p->v4_max_nch) {
#define TSIM_LHM(N)                                                                                          \
  case N: {                                                                                                  \
    auto kfn = k_sample4h_multi<4, N, NW>;                                                                   \
    if (!p->hm_attr_set)                                                                                     \
      HIP_TRY(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(NW * 64), p->h_lds, hs, M);                                     \
  } break;
    TSIM_LHM(2) TSIM_LHM(4) TSIM_LHM(6) TSIM_LHM(8) TSIM_LHM(10) TSIM_LHM(12) TSIM_LHM(14) TSIM_LHM(16)
#undef TSIM_LHM
    default: return fail(TSIM_ESTATE, "bad chunk count %d", p->v4_max_nch);
  }
  HIP_TRY(hipGetLastError());
  p->hm_attr_set = true;
  ++p->stat_flushes;
  const unsigned long long seq = p->batch_next++;
  hipEvent_t &be = p->batch_ev[seq % 16u];
  if (!be) HIP_TRY(hipEventCreateWithFlags(&be, hipEventDisableTiming));
  if (seq > 16u && p->batch_confirmed < seq - 16u) {  // the ring slot's previous batch: 16 batches ago, long done
    HIP_TRY(hipEventSynchronize(be));
    p->batch_confirmed = seq - 16u;
  }
  HIP_TRY(hipEventRecord(be, hs));
  for (int i = 0; i < M.n_ctx; ++i) {
    tsim_program::Slot &d = p->slots[p->deferred[i]];
    d.deferred = false;
    d.last_done = hs;
    d.done_ev = be;
    d.batch_seq = seq;
  }
  p->deferred.clear();
  return 0;
}

static int launch_sample(tsim_program *p, const uint64_t *d_f, int64_t B, int32_t num_f, uint32_t key_hi,
                         uint32_t key_lo, int64_t shot_offset, uint64_t *d_out, float *d_dev, hipStream_t s,
                         const uint32_t *d_row_index = nullptr, const uint32_t *d_row_count = nullptr,
                         int slot = 0, const LaunchPlan *plan_in = nullptr) {
  if (!p->sampleable) return fail(TSIM_ESTATE, "program has joint-mode components (evaluate-only)");
  if (B < 0 || num_f < 0 || shot_offset < 0) return fail(TSIM_EINVAL, "negative B/num_f/shot_offset");
  if (p->max_f_index >= num_f)
    return fail(TSIM_EINVAL, "program references f index %d but num_f=%d", p->max_f_index, num_f);
  if (B == 0 || p->num_outputs == 0) return 0;
  if (!d_f && num_f > 0) return fail(TSIM_EINVAL, "f buffer is NULL");
  if (!d_out) return fail(TSIM_EINVAL, "out buffer is NULL");
  // per-output subkeys: key, subkey = split(key) once per output, threaded through the
  // components in processing order (sampler.py:74,147-148)
  tsim_program::Slot &sl = p->slots[slot];
  {
    size_t hard_bytes = 0;
    if (p->lw) {
      const long long g1 = (B + 255) / 256;  // the pattern pass uses 256-thread blocks unless overridden
      hard_bytes = (size_t)((g1 + TSIMK_LW_LISTS - 1) / TSIMK_LW_LISTS * 1024) * TSIMK_LW_LISTS * 4;
    }
    if (int r = slot_prepare(p, slot, hard_bytes)) return r;
  }
  SampleArgs a{};
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
  } else if (p->total_keys > 0) {
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
  a.n_comp = (int)p->comps.size();
  a.comp_off = p->comp_off;
  a.row_index = d_row_index;
  a.row_count = d_row_index ? d_row_count : nullptr;
  a.row_lists = 0;
  a.row_list_cap = 0;
  a.row_slot_begin = 0;
  a.row_slot_end = 0;
  if (sl.compact_out) {  // tsim_pipeline_set_compact_output: consumed by this launch
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
  if (B > 0x7FFFFFFFll * 64) return fail(TSIM_ENOTSUP, "batch too large");
  const bool prof = p->profiling && (p->prof_counter++ % p->prof_every == 0);
  if (prof) { int r = prof_event(p, s, PROF_BEGIN); if (r) return r; }
  // the normalisation check applies to in-batch shot 0 (sampler.py:66-72) or the first listed row
  bool has_check = (shot_offset == 0 || d_row_index);
  long long B2 = B;  // slots per row list of the full kernel's launch
  const bool pipelined = slot > 0;  // lane launch: the caller passed the slot's own stream as `s`
  auto finish = [&]() -> int {
    if (pipelined) {
      HIP_TRY(hipEventRecord(sl.ev2, s));
      sl.pending = true;
      sl.last_done = s;
      sl.done_ev = sl.ev2;
      sl.batch_seq = 0;
    }
    return 0;
  };
  const LaunchPlan plan = plan_in ? *plan_in : make_plan(p, d_row_index != nullptr, false);
  const bool use_tables = plan.use_tables, need_overflow = plan.need_overflow;
  if (use_tables) {
    // pass 1: shots whose f_sel patterns are tabulated finish here, the others go to the hard list
    if (B > 0xFFFFFFFFll) return fail(TSIM_ENOTSUP, "batch too large for the row list");
    // few large blocks: 1024 threads finish a batch of 10^6 rows in 977 blocks - measurably better than
    // 3906 blocks of 256 when the blocks of several launches and of the hard-row kernel share the CUs
    const int blk1 = p->knobs.lw_block ? p->knobs.lw_block : ((size_t)(2 * a.WF + 2 * a.WO) * 1024 * 4 <= 32 * 1024 ? 1024 : 256);
    const long long grid1 = (B + blk1 - 1) / blk1;
    // n_lists sub-lists share the buffer sized for TSIMK_LW_LISTS of them: a list can hold every row of
    // the blocks that feed it
    const int n_lists = plan.lists;
    const long long list_cap = (grid1 + TSIMK_LW_LISTS - 1) / TSIMK_LW_LISTS * blk1 * (TSIMK_LW_LISTS / n_lists);
    if (list_cap > 0x7FFFFFFFll) return fail(TSIM_ENOTSUP, "batch too large for the row lists");
    if ((size_t)list_cap * n_lists * 4 > sl.hard_sz) return fail(TSIM_ESTATE, "hard-row list too small");
    p->last_lists = n_lists;
    LwArgs l;
    l.s = a;
    l.tab = p->d_lw_tab;
    l.lw_off = p->lw_off;
    l.direct_prog = p->lw_direct_prog;
    l.direct_chunks = p->lw_direct_chunks;
    l.has_check = has_check ? 1 : 0;
    l.hard_index = (uint32_t *)sl.hard;
    uint32_t *ctl = sl.ctl + sl.parity * (TSIMK_LW_LISTS + 1) * 32;
    l.ctl = ctl;
    l.ctl_next = sl.ctl + (sl.parity ^ 1) * (TSIMK_LW_LISTS + 1) * 32;
    sl.parity ^= 1;
    l.list_cap = (int)list_cap;
    l.n_lists = n_lists;
    const size_t lds1 = (size_t)(2 * a.WF + 2 * a.WO) * blk1 * 4;
    if (lds1 > 64 * 1024) return fail(TSIM_ENOTSUP, "num_f + num_outputs too large for LDS staging (%zu B)", lds1);
    hipLaunchKernelGGL(k_sample_lw, dim3((unsigned)grid1), dim3(blk1), lds1, s, l);
    HIP_TRY(hipGetLastError());
    if (prof) { int r = prof_event(p, s, PROF_PASS1); if (r) return r; }
    // pass 2 below runs on the hard lists; the check row was forced into one of them
    a.row_index = l.hard_index;
    a.row_count = ctl;
    a.row_lists = n_lists;
    a.row_list_cap = (int)list_cap;
    a.check_row = has_check ? ctl + 32 * TSIMK_LW_LISTS : nullptr;
    a.no_check = has_check ? 0 : 1;
    B2 = list_cap;
  } else if (!has_check) {
    a.no_check = 1;
  }
  int block = 256;
  size_t lds = (size_t)(2 * a.WF + 2 * a.WO) * block * 4;
  if (lds > 60 * 1024) { block = 64; lds = (size_t)(2 * a.WF + 2 * a.WO) * block * 4; }
  if (lds > 60 * 1024) return fail(TSIM_ENOTSUP, "num_f + num_outputs too large for LDS staging (%zu B)", lds);
  const long long nlists = a.row_lists > 1 ? a.row_lists : 1;
  const long long grid = (B2 + block - 1) / block * nlists;
  if (grid > 0x7FFFFFFFll) return fail(TSIM_ENOTSUP, "batch too large");
  if (p->v4) {
    // chunk-table kernel: LDS = f/out staging + two tile buffers; one extra block replays shot 0
    Sample4Args a4;
    a4.s = a;
    a4.comp4_off = p->comp4_off;
    a4.has_check = has_check ? 1 : 0;
    const int blk = p->knobs.v4_block;
    const size_t tile_bytes = std::max((size_t)p->v4_max_nch * 16, (size_t)p->v4_max_sent) * p->v4_gt * 16;
    const size_t lds4 = (size_t)(2 * a.WF + 2 * a.WO) * blk * 4 + 2 * tile_bytes;
    if (lds4 > 64 * 1024) return fail(TSIM_ENOTSUP, "v4 kernel needs %zu B of LDS", lds4);
    if (a.row_lists > 1 && p->knobs.hard_kernel) {
      // short row lists (second pass of a two-pass launch): NW waves per 64 rows, tsim_kernel4h.hip.h
      constexpr int NW = TSIM_HARD_NW;
      hard_geometry(p, a.WF, a.WO);
      const int group_tiles = p->h_group_tiles;
      if (group_tiles >= 1 && plan.defer && pipelined) {
        // leave the hard rows to the next batch: this lane goes on with the next launch's first pass
        if (!p->deferred.empty() && (p->slots[p->deferred[0]].ctx.WF != a.WF || p->slots[p->deferred[0]].ctx.WO != a.WO))
          if (int r = flush_hard(p)) return r;  // one LDS layout per batch
        sl.ctx = a;
        sl.ctx.row_slot_begin = 0;
        sl.ctx.row_slot_end = 0;
        sl.ctx_check = has_check;
        sl.deferred = true;
        sl.pending = true;
        sl.p1_stream = s;
        p->deferred.push_back(slot);
        if ((int)p->deferred.size() >= p->knobs.defer_group) return flush_hard(p);
        return 0;
      }
      if (group_tiles >= 1) {
        // the first kHardBlocks * 64 slots of every list go to the NW-wave kernel; k_sample4 below
        // serves the rest (its blocks exit at once when the lists are short - the usual case)
        constexpr int kHardBlocks = 4;
        const size_t ldsh = p->h_lds;
        const long long gridh = (long long)kHardBlocks * nlists + a4.has_check;
        Sample4Args ah = a4;
        ah.s.row_slot_end = need_overflow ? kHardBlocks * 64 : 0;
        const int loop_stride = need_overflow ? 0 : kHardBlocks * 64;
        switch (p->v4_max_nch) {
#define TSIM_LH(N)                                                                                          \
  case N: {                                                                                                 \
    auto kfn = k_sample4h<4, N, NW>;                                                                        \
    if (!p->h_attr_set)                                                                                     \
      HIP_TRY(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
    hipLaunchKernelGGL(kfn, dim3((unsigned)gridh), dim3(NW * 64), ldsh, s, ah, group_tiles, loop_stride,    \
                       p->d_feedback);                                                                      \
  } break;
          TSIM_LH(2) TSIM_LH(4) TSIM_LH(6) TSIM_LH(8) TSIM_LH(10) TSIM_LH(12) TSIM_LH(14) TSIM_LH(16)
#undef TSIM_LH
          default: return fail(TSIM_ESTATE, "bad chunk count %d", p->v4_max_nch);
        }
        HIP_TRY(hipGetLastError());
        if (prof && !p->prof_light) { int r = prof_event(p, s, PROF_HARD); if (r) return r; }
        p->h_attr_set = true;
        a4.has_check = 0;  // done by the kernel above
        a4.s.no_check = 1;
        a4.s.row_slot_begin = kHardBlocks * 64;
        B2 = need_overflow ? std::max<long long>(0, B2 - kHardBlocks * 64) : 0;
      }
    }
    const long long grid4 = (B2 + blk - 1) / blk * nlists + a4.has_check;
    if (grid4 > 0x7FFFFFFFll) return fail(TSIM_ENOTSUP, "batch too large");
    if (grid4 > 0) switch (p->v4_max_nch) {
#define TSIM_L4(N) case N: hipLaunchKernelGGL((k_sample4<4, N>), dim3((unsigned)grid4), dim3(blk), lds4, s, a4); break;
      TSIM_L4(2) TSIM_L4(4) TSIM_L4(6) TSIM_L4(8) TSIM_L4(10) TSIM_L4(12) TSIM_L4(14)
#undef TSIM_L4
      default: hipLaunchKernelGGL((k_sample4<4, 16>), dim3((unsigned)grid4), dim3(blk), lds4, s, a4); break;
    }
    HIP_TRY(hipGetLastError());
    if (prof && !(p->prof_light && use_tables)) { int r = prof_event(p, s, PROF_FULL); if (r) return r; }
    return finish();
  }
  int wmax = 1;
  for (int w : p->comp_w) wmax = std::max(wmax, w);
  switch (wmax) {
#define TSIM_LAUNCH(WV)                                                                             \
  case WV:                                                                                          \
    if (p->fast) hipLaunchKernelGGL((k_sample<WV, true>), dim3((unsigned)grid), dim3(block), lds, s, a);  \
    else hipLaunchKernelGGL((k_sample<WV, false>), dim3((unsigned)grid), dim3(block), lds, s, a);   \
    break;
    TSIM_LAUNCH(1) TSIM_LAUNCH(2) TSIM_LAUNCH(3) TSIM_LAUNCH(4) TSIM_LAUNCH(6) TSIM_LAUNCH(8)
    TSIM_LAUNCH(12) TSIM_LAUNCH(16)
#undef TSIM_LAUNCH
    default: return fail(TSIM_ENOTSUP, "unsupported word count %d", wmax);
  }
  HIP_TRY(hipGetLastError());
  if (prof && !(p->prof_light && use_tables)) { int r = prof_event(p, s, PROF_FULL); if (r) return r; }
  return finish();
}

extern "C" int tsim_sample_batch_device_begin(tsim_program *p, int32_t slot, const uint64_t *d_f, int64_t B,
                                              int32_t num_f, uint32_t key_hi, uint32_t key_lo, int64_t shot_offset,
                                              uint64_t *d_out, float *d_max_norm_dev, void *stream, uint32_t flags) {
  if (int r = need_final(p)) return r;
  if (int r = set_device(p)) return r;
  if (slot < 0 || slot >= TSIM_PIPELINE_SLOTS) return fail(TSIM_EINVAL, "slot %d out of range", slot);
  hipStream_t s_user = stream ? (hipStream_t)stream : p->stream;
  tsim_program::Slot &sl = p->slots[1 + slot];
  if (!p->slots_ready) {  // first pipelined launch: create every slot's stream/buffers now, not mid-run
    size_t hard_bytes = 0;
    if (p->lw) hard_bytes = (size_t)(((B + 255) / 256 + TSIMK_LW_LISTS - 1) / TSIMK_LW_LISTS * 1024) * TSIMK_LW_LISTS * 4;
    for (int k = 1; k <= TSIM_PIPELINE_SLOTS; ++k)
      if (int r = slot_prepare(p, k, hard_bytes)) return r;
    p->slots_ready = true;
  }
  if (int r = slot_prepare(p, 1 + slot, 0)) return r;
  // The whole launch runs on the slot's own stream (a "lane"): launches of one slot are ordered by the
  // stream itself, launches of different slots overlap.  Unless the caller vouches for its inputs the
  // lane first waits for what is already queued on the caller's stream.
  // Deferred plan (short hard-row lists): the first passes alternate between the first two lanes and
  // the hard rows of several launches go to the third lane in one batch (flush_hard), so no lane
  // waits for a second pass before it starts the next first pass.
  if (sl.deferred)  // begin twice without end: finish the earlier launch's hard rows first
    if (int r = flush_hard(p)) return r;
  const LaunchPlan plan = make_plan(p, false, true);
  hipStream_t s = plan.defer ? p->slots[1 + (slot & 1)].side : sl.side;
  if (!plan.defer) sl.used = true;
  // the slot's previous launch (its lists, counters and output rows are reused) finished on another stream:
  // this launch must be ordered after it
  if (sl.last_done && sl.last_done != s && sl.done_ev) {
    bool done = false;
    if (sl.batch_seq) {
      // Batches complete in order (one stream) and a lane is in order too: once a lane waits for batch b it
      // is behind every batch <= b.  The four slots of a batch alternate over the two lanes, so this is
      // one stream wait per lane and batch - no event query (the host usually runs several batches ahead
      // of the GPU, the query would fail and cost as much as the wait).
      const int lane = (s == p->slots[1].side) ? 0 : (s == p->slots[2].side) ? 1 : -1;
      if (sl.batch_seq <= p->batch_confirmed || (lane >= 0 && sl.batch_seq <= p->lane_waited[lane])) done = true;
      else if (lane >= 0) p->lane_waited[lane] = sl.batch_seq;
    } else {
      ++p->stat_queries;
      done = hipEventQuery(sl.done_ev) == hipSuccess;
      if (!done) (void)hipGetLastError();
    }
    if (!done) { ++p->stat_waits; HIP_TRY(hipStreamWaitEvent(s, sl.done_ev, 0)); }
  }
  ++p->stat_begins;
  if (plan.defer) ++p->stat_deferred;
  if (!(flags & TSIM_PIPE_INPUTS_READY) && s_user != s) {
    HIP_TRY(hipEventRecord(sl.ev1, s_user));
    HIP_TRY(hipStreamWaitEvent(s, sl.ev1, 0));
  }
  return launch_sample(p, d_f, B, num_f, key_hi, key_lo, shot_offset, d_out, d_max_norm_dev, s, nullptr, nullptr,
                       1 + slot, &plan);
}

extern "C" int tsim_pipeline_lane_stream(tsim_program *p, int32_t lane, void **stream) {
  if (int r = need_final(p)) return r;
  if (int r = set_device(p)) return r;
  if (lane < 0 || lane >= TSIM_PIPELINE_SLOTS || !stream) return fail(TSIM_EINVAL, "bad lane %d", lane);
  if (int r = slot_prepare(p, 1 + lane, 0)) return r;
  *stream = (void *)p->slots[1 + lane].side;
  return TSIM_OK;
}

extern "C" int tsim_pipeline_wait_stream(tsim_program *p, void *stream) {
  if (int r = need_final(p)) return r;
  if (int r = set_device(p)) return r;
  hipStream_t s_user = stream ? (hipStream_t)stream : p->stream;
  if (!p->slots_ready) return TSIM_OK;  // no lane exists yet: the first launches order themselves
  if (!p->sync_ev) HIP_TRY(hipEventCreateWithFlags(&p->sync_ev, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(p->sync_ev, s_user));
  // every lane a launch may run on: the slots' own streams that were used so far and the first three
  // (first passes / hard-row batches of the deferred plan)
  std::vector<hipStream_t> seen;
  for (int k = 1; k <= TSIM_PIPELINE_SLOTS; ++k) {
    tsim_program::Slot &sl = p->slots[k];
    if (!sl.side_ready || (k > 3 && !sl.used) || sl.side == s_user) continue;
    if (std::find(seen.begin(), seen.end(), sl.side) != seen.end()) continue;
    seen.push_back(sl.side);
    HIP_TRY(hipStreamWaitEvent(sl.side, p->sync_ev, 0));
  }
  return TSIM_OK;
}

extern "C" int tsim_pipeline_set_compact_series(tsim_program *p, uint8_t *d_base, int64_t stride_bytes, int32_t count) {
  if (int r = need_final(p)) return r;
  if (count < 0 || stride_bytes < 0 || (count > 0 && !d_base)) return fail(TSIM_EINVAL, "bad compact series");
  p->series_ptr = d_base;
  p->series_stride = stride_bytes;
  p->series_left = count;
  return TSIM_OK;
}

extern "C" int tsim_pipeline_set_compact_output(tsim_program *p, int32_t slot, uint8_t *d_compact) {
  if (int r = need_final(p)) return r;
  if (slot < 0 || slot >= TSIM_PIPELINE_SLOTS) return fail(TSIM_EINVAL, "slot %d out of range", slot);
  p->slots[1 + slot].compact_out = d_compact;
  return TSIM_OK;
}

extern "C" int tsim_sample_batch_device_compact(tsim_program *p, int32_t slot, const uint64_t *d_rows, int64_t B,
                                                int32_t nbits, uint8_t *d_out, void *stream) {
  if (int r = need_final(p)) return r;
  if (int r = set_device(p)) return r;
  if (slot < 0 || slot >= TSIM_PIPELINE_SLOTS) return fail(TSIM_EINVAL, "slot %d out of range", slot);
  if (B < 0 || nbits < 0) return fail(TSIM_EINVAL, "negative size");
  if (B == 0 || nbits == 0) return TSIM_OK;
  if (!d_rows || !d_out) return fail(TSIM_EINVAL, "NULL buffer");
  tsim_program::Slot &sl = p->slots[1 + slot];
  if (sl.deferred)
    if (int r = flush_hard(p)) return r;
  // after the slot's second pass when there is one in flight, else simply on the caller's stream
  hipStream_t s = sl.pending ? sl.last_done : (stream ? (hipStream_t)stream : p->stream);
  const int WO = (nbits + 63) / 64, rb = (nbits + 7) / 8;
  const long long nthreads = (B + 3) / 4;
  const uint32_t tail_mask = (nbits & 7) ? ((1u << (nbits & 7)) - 1u) : 255u;
  hipLaunchKernelGGL(k_compact_rows, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, s, d_rows, d_out,
                     (long long)B, WO, rb, tail_mask);
  HIP_TRY(hipGetLastError());
  if (sl.pending) {
    HIP_TRY(hipEventRecord(sl.ev2, s));
    sl.done_ev = sl.ev2;
    sl.batch_seq = 0;
  }
  return TSIM_OK;
}

extern "C" int tsim_sample_batch_device_end(tsim_program *p, int32_t slot, void *stream) {
  if (int r = need_final(p)) return r;
  if (int r = set_device(p)) return r;
  if (slot < 0 || slot >= TSIM_PIPELINE_SLOTS) return fail(TSIM_EINVAL, "slot %d out of range", slot);
  hipStream_t s = stream ? (hipStream_t)stream : p->stream;
  tsim_program::Slot &sl = p->slots[1 + slot];
  if (sl.deferred)  // its batch is not full yet: run what is waiting now
    if (int r = flush_hard(p)) return r;
  if (sl.pending) {
    if (sl.last_done != s) HIP_TRY(hipStreamWaitEvent(s, sl.done_ev, 0));
    sl.pending = false;
  }
  return TSIM_OK;
}

extern "C" int tsim_sample_batch_device(tsim_program *p, const uint64_t *d_f, int64_t B, int32_t num_f,
                                        uint32_t key_hi, uint32_t key_lo, int64_t shot_offset,
                                        uint64_t *d_out, float *d_max_norm_dev, void *stream) {
  if (int r = need_final(p)) return r;
  if (int r = set_device(p)) return r;
  hipStream_t s = stream ? (hipStream_t)stream : p->stream;
  return launch_sample(p, d_f, B, num_f, key_hi, key_lo, shot_offset, d_out, d_max_norm_dev, s);
}

static int launch_pack(tsim_program *p, const uint8_t *d_in, int64_t B, int32_t nbits, uint64_t *d_out, hipStream_t s) {
  const int n32 = 2 * ((nbits + 63) / 64);
  const long long n = B * n32;
  if (n == 0) return 0;
  if ((nbits & 15) == 0 && ((uintptr_t)d_in & 15) == 0) {
    const long long nw = B * (n32 / 2);
    hipLaunchKernelGGL(k_pack_bits_a16, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, s, (const uint4 *)d_in, d_out,
                       (long long)B, nbits, n32 / 2);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  const long long in_dwords = (B * (long long)nbits + 3) / 4;  // the last dword may be partial
  hipLaunchKernelGGL(k_pack_bits, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_in, (uint32_t *)d_out,
                     (long long)B, nbits, n32, in_dwords);
  HIP_TRY(hipGetLastError());
  return 0;
}

static int launch_unpack(tsim_program *p, const uint64_t *d_in, int64_t B, int32_t nbits, uint8_t *d_out, hipStream_t s) {
  const int n32 = 2 * ((nbits + 63) / 64);
  const long long total = B * (long long)nbits;
  if (total == 0) return 0;
  if (total >= (1ll << 32)) return fail(TSIM_ENOTSUP, "unpack of %lld bytes in one call (limit 2^32)", total);
  if ((nbits & 15) == 0 && ((uintptr_t)d_out & 15) == 0) {
    const long long total16 = total >> 4;
    const unsigned long long per = (unsigned long long)(nbits >> 4);
    const unsigned long long magic16 = ((1ull << 40) + per - 1ull) / per;
    hipLaunchKernelGGL(k_unpack_bits_a16, dim3((unsigned)((total16 + 255) / 256)), dim3(256), 0, s, (const uint32_t *)d_in,
                       (uint4 *)d_out, total16, nbits, n32, magic16);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  const long long nthreads = (total + 3) / 4;
  const unsigned long long magic = ((1ull << 40) + (unsigned long long)nbits - 1ull) / (unsigned long long)nbits;
  hipLaunchKernelGGL(k_unpack_bits, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, s, (const uint32_t *)d_in,
                     (uint32_t *)d_out, d_out, total, nbits, n32, magic, (long long)B * n32);
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int tsim_sample_rows_device(tsim_program *p, const uint64_t *d_f, int64_t B, int32_t num_f,
                                       uint32_t key_hi, uint32_t key_lo, int64_t shot_offset, uint64_t *d_out,
                                       float *d_max_norm_dev, const uint32_t *d_row_index,
                                       const uint32_t *d_row_count, void *stream) {
  if (int r = need_final(p)) return r;
  if (int r = set_device(p)) return r;
  if (!d_row_index || !d_row_count) return fail(TSIM_EINVAL, "row list is NULL");
  if (B >= (1ll << 32)) return fail(TSIM_ENOTSUP, "row indices are 32-bit");
  hipStream_t s = stream ? (hipStream_t)stream : p->stream;
  return launch_sample(p, d_f, B, num_f, key_hi, key_lo, shot_offset, d_out, d_max_norm_dev, s, d_row_index,
                       d_row_count);
}

extern "C" int tsim_postselect_device(tsim_program *p, const uint64_t *d_f, int64_t B, int32_t num_f,
                                      const uint64_t *d_mask, const uint64_t *d_ref, uint64_t *d_out,
                                      uint32_t *d_row_index, uint32_t *d_row_count, uint8_t *d_discarded,
                                      void *stream) {
  if (int r = need_final(p)) return r;
  if (int r = set_device(p)) return r;
  if (B < 0 || num_f < 0) return fail(TSIM_EINVAL, "negative size");
  if (B >= (1ll << 32)) return fail(TSIM_ENOTSUP, "row indices are 32-bit");
  if (p->max_f_index >= num_f) return fail(TSIM_EINVAL, "program references f index %d but num_f=%d", p->max_f_index, num_f);
  if (!d_mask || !d_out || !d_row_index || !d_row_count) return fail(TSIM_EINVAL, "NULL buffer");
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
  if (lds > 60 * 1024) return fail(TSIM_ENOTSUP, "num_f + num_outputs too large for LDS staging (%zu B)", lds);
  hipLaunchKernelGGL(k_direct_filter, dim3((unsigned)((B + block - 1) / block)), dim3(block), lds, s, a);
  HIP_TRY(hipGetLastError());
  return TSIM_OK;
}

extern "C" int tsim_pack_bits_device(tsim_program *p, const uint8_t *d_in, int64_t B, int32_t nbits,
                                     uint64_t *d_out, void *stream) {
  if (int r = need_final(p)) return r;
  if (int r = set_device(p)) return r;
  if (B < 0 || nbits < 0) return fail(TSIM_EINVAL, "negative size");
  return launch_pack(p, d_in, B, nbits, d_out, stream ? (hipStream_t)stream : p->stream);
}

extern "C" int tsim_unpack_bits_device(tsim_program *p, const uint64_t *d_in, int64_t B, int32_t nbits,
                                       uint8_t *d_out, void *stream) {
  if (int r = need_final(p)) return r;
  if (int r = set_device(p)) return r;
  if (B < 0 || nbits < 0) return fail(TSIM_EINVAL, "negative size");
  return launch_unpack(p, d_in, B, nbits, d_out, stream ? (hipStream_t)stream : p->stream);
}

extern "C" int tsim_compact_rows_device(tsim_program *p, const uint64_t *d_in, int64_t B, int32_t in_words,
                                        int32_t nbits, uint8_t *d_out, void *stream) {
  if (int r = need_final(p)) return r;
  if (int r = set_device(p)) return r;
  if (B < 0 || nbits < 0 || in_words < 0) return fail(TSIM_EINVAL, "negative size");
  if (B == 0 || nbits == 0) return TSIM_OK;
  if (!d_in || !d_out) return fail(TSIM_EINVAL, "NULL buffer");
  const int WO = in_words ? in_words : (nbits + 63) / 64, rb = (nbits + 7) / 8;
  if ((long long)WO * 64 < nbits) return fail(TSIM_EINVAL, "rows of %d words hold fewer than %d bits", WO, nbits);
  const long long nthreads = (B + 3) / 4;
  const uint32_t tail_mask = (nbits & 7) ? ((1u << (nbits & 7)) - 1u) : 255u;
  hipLaunchKernelGGL(k_compact_rows, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0,
                     stream ? (hipStream_t)stream : p->stream, d_in, d_out, (long long)B, WO, rb, tail_mask);
  HIP_TRY(hipGetLastError());
  return TSIM_OK;
}

extern "C" int tsim_sample_batch(tsim_program *p, const uint8_t *f, int64_t B, int32_t num_f, uint32_t key_hi,
                                 uint32_t key_lo, int64_t shot_offset, uint8_t *out, int32_t out_packed,
                                 float *max_norm_dev) {
  if (int r = need_final(p)) return r;
  if (int r = set_device(p)) return r;
  if (B < 0 || num_f < 0) return fail(TSIM_EINVAL, "negative B/num_f");
  if (B == 0 || p->num_outputs == 0) return 0;
  if (!out) return fail(TSIM_EINVAL, "out is NULL");
  if (!f && num_f > 0) return fail(TSIM_EINVAL, "f is NULL");
  const int WF = std::max(1, (num_f + 63) / 64), WO = (p->num_outputs + 63) / 64;
  hipStream_t s = p->stream;
  if (int r = ensure_scratch(p, 0, (size_t)B * std::max(1, num_f))) return r;
  if (int r = ensure_scratch(p, 1, (size_t)B * WF * 8)) return r;
  if (int r = ensure_scratch(p, 2, (size_t)B * WO * 8)) return r;
  if (num_f > 0) {
    HIP_TRY(hipMemcpyAsync(p->scratch[0], f, (size_t)B * num_f, hipMemcpyHostToDevice, s));
    if (int r = launch_pack(p, (const uint8_t *)p->scratch[0], B, num_f, (uint64_t *)p->scratch[1], s)) return r;
  }
  if (int r = launch_sample(p, (const uint64_t *)p->scratch[1], B, num_f, key_hi, key_lo, shot_offset,
                            (uint64_t *)p->scratch[2], p->d_dev, s))
    return r;
  if (out_packed) {
    HIP_TRY(hipMemcpyAsync(out, p->scratch[2], (size_t)B * WO * 8, hipMemcpyDeviceToHost, s));
  } else {
    if (int r = ensure_scratch(p, 3, (size_t)B * p->num_outputs)) return r;
    if (int r = launch_unpack(p, (const uint64_t *)p->scratch[2], B, p->num_outputs, (uint8_t *)p->scratch[3], s)) return r;
    HIP_TRY(hipMemcpyAsync(out, p->scratch[3], (size_t)B * p->num_outputs, hipMemcpyDeviceToHost, s));
  }
  if (max_norm_dev && shot_offset == 0 && !p->comps.empty())
    HIP_TRY(hipMemcpyAsync(max_norm_dev, p->d_dev, p->comps.size() * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return TSIM_OK;
}

template <int W>
static void launch_eval_w(const EvalArgs &a, hipStream_t s, bool fast) {
  const dim3 grid((unsigned)((a.B + 255) / 256));
  if (fast) hipLaunchKernelGGL((k_evaluate<W, true>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((k_evaluate<W, false>), grid, dim3(256), 0, s, a);
}

extern "C" int tsim_evaluate(tsim_program *p, int32_t component, int32_t level, const uint8_t *params,
                             int64_t B, float *re, float *im, float *abs_out, int32_t *coeffs_power) {
  if (int r = need_final(p)) return r;
  if (int r = set_device(p)) return r;
  if (component < 0 || component >= (int)p->comps.size()) return fail(TSIM_EINVAL, "bad component %d", component);
  const HostComponent &c = p->comps[component];
  if (level < 0 || level >= c.n_levels) return fail(TSIM_EINVAL, "bad level %d", level);
  if (B < 0) return fail(TSIM_EINVAL, "negative B");
  if (B == 0) return 0;
  if (!re || !im) return fail(TSIM_EINVAL, "output is NULL");
  const int P = c.levels[level].P, W = p->comp_w[component];
  if (P > 0 && !params) return fail(TSIM_EINVAL, "params is NULL");
  // host-side pack to W 32-bit words per row (astype(bool): nonzero == 1)
  std::vector<uint32_t> x((size_t)B * W, 0u);
  for (int64_t r = 0; r < B; ++r) {
    const uint8_t *src = params + (size_t)r * P;
    uint32_t *dst = &x[(size_t)r * W];
    for (int i = 0; i < P; ++i)
      if (src[i]) dst[i >> 5] |= 1u << (i & 31);
  }
  hipStream_t s = p->stream;
  if (int r = ensure_scratch(p, 0, x.size() * 4)) return r;
  if (int r = ensure_scratch(p, 1, (size_t)B * 12)) return r;
  if (int r = ensure_scratch(p, 2, (size_t)B * 20)) return r;
  HIP_TRY(hipMemcpyAsync(p->scratch[0], x.data(), x.size() * 4, hipMemcpyHostToDevice, s));
  EvalArgs a;
  a.img = p->d_img;
  a.x = (const uint32_t *)p->scratch[0];
  a.re = (float *)p->scratch[1];
  a.im = a.re + B;
  a.abs = abs_out ? a.re + 2 * B : nullptr;
  a.exact = coeffs_power ? (int *)p->scratch[2] : nullptr;
  a.B = B;
  a.level_off = p->level_off[p->level_base[component] + level];
  a.W = W;
  switch (W) {
    case 1: launch_eval_w<1>(a, s, p->fast); break;
    case 2: launch_eval_w<2>(a, s, p->fast); break;
    case 3: launch_eval_w<3>(a, s, p->fast); break;
    case 4: launch_eval_w<4>(a, s, p->fast); break;
    case 6: launch_eval_w<6>(a, s, p->fast); break;
    case 8: launch_eval_w<8>(a, s, p->fast); break;
    case 12: launch_eval_w<12>(a, s, p->fast); break;
    case 16: launch_eval_w<16>(a, s, p->fast); break;
    default: return fail(TSIM_ENOTSUP, "unsupported word count %d", W);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(re, a.re, (size_t)B * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(im, a.im, (size_t)B * 4, hipMemcpyDeviceToHost, s));
  if (abs_out) HIP_TRY(hipMemcpyAsync(abs_out, a.abs, (size_t)B * 4, hipMemcpyDeviceToHost, s));
  if (coeffs_power) HIP_TRY(hipMemcpyAsync(coeffs_power, a.exact, (size_t)B * 20, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return TSIM_OK;
}

// ---------------------------------------------------------------------------
// device-side noise sampler (statistical twin of ChannelSampler.sample)
// ---------------------------------------------------------------------------
struct tsim_noise {
  tsim_program *prog = nullptr;
  int num_f = 0, n_ch = 0, WF = 0, seg = 4096;
  double *d_l1p = nullptr;
  uint32_t *d_off = nullptr;
  float *d_cdf = nullptr;
  uint64_t *d_pat = nullptr;
};

extern "C" int tsim_noise_create(tsim_program *p, int32_t num_f, int32_t n_channels, const double *p_fire,
                                 const int32_t *n_outcomes, const double *cond_cdf, const uint8_t *xor_patterns,
                                 tsim_noise **out) {
  if (int r = need_final(p)) return r;
  if (int r = set_device(p)) return r;
  if (!out || num_f < 0 || n_channels < 0) return fail(TSIM_EINVAL, "bad argument");
  if (n_channels > 0 && (!p_fire || !n_outcomes || !cond_cdf || (num_f > 0 && !xor_patterns)))
    return fail(TSIM_EINVAL, "NULL channel table");
  tsim_noise *n = new (std::nothrow) tsim_noise();
  if (!n) return fail(TSIM_ENOMEM, "out of host memory");
  n->prog = p;
  n->num_f = num_f;
  n->n_ch = n_channels;
  n->WF = std::max(1, (num_f + 63) / 64);
  std::vector<double> l1p((size_t)std::max(1, n_channels));
  std::vector<uint32_t> off((size_t)n_channels + 1, 0u);
  double pmax = 1e-9;
  for (int c = 0; c < n_channels; ++c) {
    if (!(p_fire[c] > 0.0) || p_fire[c] > 1.0 || n_outcomes[c] < 1) {
      delete n;
      return fail(TSIM_EINVAL, "channel %d: p_fire=%g, outcomes=%d", c, p_fire[c], n_outcomes[c]);
    }
    l1p[c] = p_fire[c] >= 1.0 ? 0.0 : log1p(-p_fire[c]);
    off[c + 1] = off[c] + (uint32_t)n_outcomes[c];
    pmax = std::max(pmax, p_fire[c]);
  }
  const size_t tot = off[n_channels];
  std::vector<float> cdf(std::max<size_t>(1, tot));
  for (size_t i = 0; i < tot; ++i) cdf[i] = (float)cond_cdf[i];
  std::vector<uint64_t> pat(std::max<size_t>(1, tot * n->WF), 0ull);
  for (size_t o = 0; o < tot; ++o)
    for (int i = 0; i < num_f; ++i)
      if (xor_patterns[o * (size_t)num_f + i]) pat[o * n->WF + (i >> 6)] |= 1ull << (i & 63);
  // segment length: about 8 expected fires of the most active channel per thread
  int seg = 64;
  while (seg < 65536 && seg * pmax < 8.0) seg *= 2;
  n->seg = seg;
  hipError_t e = hipSuccess;
  if ((e = hipMalloc((void **)&n->d_l1p, l1p.size() * 8)) != hipSuccess ||
      (e = hipMalloc((void **)&n->d_off, off.size() * 4)) != hipSuccess ||
      (e = hipMalloc((void **)&n->d_cdf, cdf.size() * 4)) != hipSuccess ||
      (e = hipMalloc((void **)&n->d_pat, pat.size() * 8)) != hipSuccess) {
    delete n;
    return fail(TSIM_ENOMEM, "hipMalloc failed: %s", hipGetErrorString(e));
  }
  HIP_TRY(hipMemcpy(n->d_l1p, l1p.data(), l1p.size() * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(n->d_off, off.data(), off.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(n->d_cdf, cdf.data(), cdf.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(n->d_pat, pat.data(), pat.size() * 8, hipMemcpyHostToDevice));
  *out = n;
  return TSIM_OK;
}

extern "C" int tsim_noise_sample_device(tsim_noise *n, int64_t B, uint32_t key_hi, uint32_t key_lo, uint64_t *d_f,
                                        void *stream) {
  if (!n || !n->prog) return fail(TSIM_EINVAL, "noise sampler is NULL");
  if (int r = set_device(n->prog)) return r;
  if (B < 0) return fail(TSIM_EINVAL, "negative B");
  if (B == 0) return TSIM_OK;
  if (!d_f) return fail(TSIM_EINVAL, "f buffer is NULL");
  hipStream_t s = stream ? (hipStream_t)stream : n->prog->stream;
  HIP_TRY(hipMemsetAsync(d_f, 0, (size_t)B * n->WF * 8, s));
  if (n->n_ch == 0) return TSIM_OK;
  NoiseArgs a;
  a.log1m_p = n->d_l1p;
  a.cdf_off = n->d_off;
  a.cdf = n->d_cdf;
  a.patterns = n->d_pat;
  a.f = (unsigned long long *)d_f;
  a.B = B;
  a.n_ch = n->n_ch;
  a.WF = n->WF;
  a.seg = n->seg;
  a.n_seg = (B + n->seg - 1) / n->seg;
  a.k0 = key_hi;
  a.k1 = key_lo;
  const long long total = (long long)a.n_ch * a.n_seg;
  const long long grid = (total + 255) / 256;
  if (grid > 0x7FFFFFFFll) return fail(TSIM_ENOTSUP, "noise launch too large");
  hipLaunchKernelGGL(k_noise, dim3((unsigned)grid), dim3(256), 0, s, a);
  HIP_TRY(hipGetLastError());
  return TSIM_OK;
}

extern "C" void tsim_noise_destroy(tsim_noise *n) {
  if (!n) return;
  if (n->prog && n->prog->device >= 0) (void)hipSetDevice(n->prog->device);
  if (n->d_l1p) (void)hipFree(n->d_l1p);
  if (n->d_off) (void)hipFree(n->d_off);
  if (n->d_cdf) (void)hipFree(n->d_cdf);
  if (n->d_pat) (void)hipFree(n->d_pat);
  delete n;
}

// ---------------------------------------------------------------------------
// plumbing
// ---------------------------------------------------------------------------
extern "C" int tsim_device_count(int32_t *count) {
  if (!count) return fail(TSIM_EINVAL, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(TSIM_EHIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
  }
  *count = n;
  return TSIM_OK;
}

extern "C" int tsim_malloc_device(tsim_program *p, int64_t nbytes, void **d_ptr) {
  if (int r = need_final(p)) return r;
  if (int r = set_device(p)) return r;
  if (!d_ptr || nbytes < 0) return fail(TSIM_EINVAL, "bad argument");
  hipError_t e = hipMalloc(d_ptr, (size_t)std::max<int64_t>(nbytes, 1));
  if (e != hipSuccess) return fail(TSIM_ENOMEM, "hipMalloc(%lld) failed: %s", (long long)nbytes, hipGetErrorString(e));
  return TSIM_OK;
}

extern "C" int tsim_free_device(tsim_program *p, void *d_ptr) {
  if (int r = need_final(p)) return r;
  if (int r = set_device(p)) return r;
  HIP_TRY(hipFree(d_ptr));
  return TSIM_OK;
}

extern "C" int tsim_malloc_pinned(int64_t nbytes, void **h_ptr) {
  if (!h_ptr || nbytes < 0) return fail(TSIM_EINVAL, "bad argument");
  hipError_t e = hipHostMalloc(h_ptr, (size_t)std::max<int64_t>(nbytes, 1), hipHostMallocDefault);
  if (e != hipSuccess) return fail(TSIM_ENOMEM, "hipHostMalloc(%lld) failed: %s", (long long)nbytes, hipGetErrorString(e));
  return TSIM_OK;
}

extern "C" int tsim_free_pinned(void *h_ptr) {
  HIP_TRY(hipHostFree(h_ptr));
  return TSIM_OK;
}

extern "C" int tsim_memcpy_h2d(tsim_program *p, void *d_dst, const void *h_src, int64_t nbytes) {
  if (int r = need_final(p)) return r;
  if (int r = set_device(p)) return r;
  if (nbytes < 0) return fail(TSIM_EINVAL, "negative size");
  if (nbytes == 0) return TSIM_OK;
  HIP_TRY(hipMemcpyAsync(d_dst, h_src, (size_t)nbytes, hipMemcpyHostToDevice, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  return TSIM_OK;
}

extern "C" int tsim_memcpy_d2h(tsim_program *p, void *h_dst, const void *d_src, int64_t nbytes) {
  if (int r = need_final(p)) return r;
  if (int r = set_device(p)) return r;
  if (nbytes < 0) return fail(TSIM_EINVAL, "negative size");
  if (nbytes == 0) return TSIM_OK;
  HIP_TRY(hipMemcpyAsync(h_dst, d_src, (size_t)nbytes, hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  return TSIM_OK;
}

extern "C" int tsim_get_stream(tsim_program *p, void **stream) {
  if (int r = need_final(p)) return r;
  if (!stream) return fail(TSIM_EINVAL, "stream is NULL");
  *stream = (void *)p->stream;
  return TSIM_OK;
}

extern "C" void tsim_key_split(uint32_t key_hi, uint32_t key_lo, uint32_t out[4]) {
  // new_key, subkey = jax.random.split(key) (threefry_partitionable): counters (0,0) and (0,1)
  uint32_t a0 = 0u, a1 = 0u, b0 = 0u, b1 = 1u;
  threefry2x32(key_hi, key_lo, a0, a1);
  threefry2x32(key_hi, key_lo, b0, b1);
  out[0] = a0; out[1] = a1; out[2] = b0; out[3] = b1;
}

extern "C" int tsim_sample_batch_device_begin_split(tsim_program *p, int32_t slot, const uint64_t *d_f, int64_t B,
                                                    int32_t num_f, uint32_t key[2], int64_t shot_offset, uint64_t *d_out,
                                                    float *d_max_norm_dev, void *stream, uint32_t flags) {
  if (!key) return fail(TSIM_EINVAL, "key is NULL");
  uint32_t o[4];
  tsim_key_split(key[0], key[1], o);  // key, subkey = split(key)  (sampler.py:399)
  key[0] = o[0];
  key[1] = o[1];
  return tsim_sample_batch_device_begin(p, slot, d_f, B, num_f, o[2], o[3], shot_offset, d_out, d_max_norm_dev, stream,
                                        flags);
}

extern "C" int tsim_synchronize(tsim_program *p) {
  if (int r = need_final(p)) return r;
  if (int r = set_device(p)) return r;
  if (int r = flush_hard(p)) return r;
  HIP_TRY(hipStreamSynchronize(p->stream));
  // the three lanes of the deferred plan and every slot stream a launch actually ran on (a stream that never
  // carried work has nothing to wait for - and each hipStreamSynchronize costs a few microseconds)
  for (int k = 1; k <= TSIM_PIPELINE_SLOTS; ++k) {
    tsim_program::Slot &sl = p->slots[k];
    if (!sl.side_ready) continue;
    if ((k <= 3 || sl.used) && sl.side != p->stream) HIP_TRY(hipStreamSynchronize(sl.side));
    sl.pending = false;
  }
  return TSIM_OK;
}

extern "C" int tsim_profile_enable(tsim_program *p, int32_t on) {
  if (int r = need_final(p)) return r;
  if (int r = set_device(p)) return r;
  if (!on && p->ev_used) { if (int r = prof_drain(p)) return r; }
  p->profiling = on != 0;
  p->prof_light = on == 2;
  p->prof_counter = 0;
  return TSIM_OK;
}

extern "C" int tsim_profile_set_sampling(tsim_program *p, int32_t every) {
  if (!p || every < 1) return fail(TSIM_EINVAL, "bad argument");
  p->prof_every = every;
  p->prof_counter = 0;
  return TSIM_OK;
}

extern "C" int tsim_profile_read(tsim_program *p, double *kernel_ms, int64_t *launches, int32_t reset) {
  if (int r = need_final(p)) return r;
  if (int r = set_device(p)) return r;
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

extern "C" int tsim_profile_read_stages(tsim_program *p, double stage_ms[3]) {
  if (int r = need_final(p)) return r;
  if (int r = set_device(p)) return r;
  if (!stage_ms) return fail(TSIM_EINVAL, "NULL argument");
  if (int r = prof_drain(p)) return r;
  stage_ms[0] = p->prof_stage_ms[PROF_PASS1];
  stage_ms[1] = p->prof_stage_ms[PROF_HARD];
  stage_ms[2] = p->prof_stage_ms[PROF_FULL];
  return TSIM_OK;
}
