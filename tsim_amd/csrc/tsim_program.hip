// tsim_program.hip - handle life cycle of the C ABI (include/tsim_hip.h): program description ->
// packed image -> upload, plus the memory / stream plumbing.  No kernels are defined here.
#include "tsim_internal.hip.h"
#include <map>
#include "tsim_sample_internal.hip.h"  // slot_prepare: the lanes are created before the table-build helper takes its stream
#include "tsim_lw_fastm.hip.h"
#include "tsim_kernel4w.hip.h"  // C4_SELMASK
#include "tsim_wide.hip.h"      // WR_*: the wide record
#include "tsim_gen.hip.h"       // GR_*, GC_*: the gen record

using namespace tsimk;
using namespace tsimhost;

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";

bool tsim_debug(const char *what) {
  const char *e = getenv("TSIM_AMD_DEBUG");
  return e && strstr(e, what) != nullptr;
}

int tsim_fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

// ---------------------------------------------------------------------------
// the stream pool (tsim_internal.hip.h)
// ---------------------------------------------------------------------------
namespace {
struct PooledStream { hipStream_t s; int index; bool free; };
struct StreamPool { std::mutex m; std::map<int, std::vector<PooledStream>> all; };  // keyed by the real device id (`device & 15` aliased devices 16+)
StreamPool &stream_pool() { static StreamPool *sp = new StreamPool; return *sp; }  // (never destroyed: HIP may be gone at exit)
}  // namespace

int tsim_stream_acquire(int device, std::vector<int> &held, hipStream_t *out) {
  StreamPool &sp = stream_pool();
  std::lock_guard<std::mutex> lk(sp.m);
  auto &v = sp.all[device];
  unsigned busy_q = 0;
  for (int i : held) busy_q |= 1u << (i & 3);
  PooledStream *pick = nullptr;
  for (auto &e : v)  // a free stream on a hardware queue this handle does not use yet, else any free one
    if (e.free && !((busy_q >> (e.index & 3)) & 1u)) { pick = &e; break; }
  if (!pick && busy_q == 0xFu)
    for (auto &e : v)
      if (e.free) { pick = &e; break; }
  if (!pick) {
    hipStream_t s = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    v.push_back(PooledStream{s, (int)v.size(), true});
    pick = &v.back();
  }
  pick->free = false;
  held.push_back(pick->index);
  *out = pick->s;
  return 0;
}

void tsim_stream_release(int device, hipStream_t s) {
  if (!s) return;
  (void)hipStreamSynchronize(s);
  StreamPool &sp = stream_pool();
  std::lock_guard<std::mutex> lk(sp.m);
  for (auto &e : sp.all[device])
    if (e.s == s) { e.free = true; return; }
  (void)hipStreamDestroy(s);  // not one of ours
}

extern "C" const char *tsim_last_error(void) { return g_err; }
extern "C" const char *tsim_tune_keys(void) {
  return "defer_hard,defer_group,lw_fast,wide_fused,wide_compact,wide_tables,wide_depth,wide_passes,hard_wave,hard_wave_rows,hard_inline_rows,"
         "hard_comp_par,hard_overflow,deep_after,fused_lanes,fused_max,gen,trie,shallow,x3,x4,noise_wave,noise_fused";
}
extern "C" const char *tsim_version(void) { return "tsim_amd-hip 0.2 (gfx950)"; }

int tsim_set_device(const tsim_program *p) {
  HIP_TRY(hipSetDevice(p->device));
  return 0;
}

// ---------------------------------------------------------------------------
// construction
// ---------------------------------------------------------------------------
extern "C" int tsim_program_create(int32_t num_outputs, int32_t num_detectors, int32_t n_direct,
                                   const int32_t *direct_f_indices, const uint8_t *direct_flips,
                                   const int32_t *output_order, tsim_program **out) {
  if (!out) return tsim_fail(TSIM_EINVAL, "out is NULL");
  if (num_outputs < 0 || n_direct < 0 || n_direct > num_outputs || num_detectors < 0)
    return tsim_fail(TSIM_EINVAL, "bad counts: num_outputs=%d n_direct=%d num_detectors=%d", num_outputs,
                n_direct, num_detectors);
  if (n_direct > 0 && (!direct_f_indices || !direct_flips)) return tsim_fail(TSIM_EINVAL, "direct arrays NULL");
  if (num_outputs > 0 && !output_order) return tsim_fail(TSIM_EINVAL, "output_order is NULL");
  tsim_program *p = new (std::nothrow) tsim_program();
  if (!p) return tsim_fail(TSIM_ENOMEM, "out of host memory");
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
      return tsim_fail(TSIM_EINVAL, "output_order is not a permutation of 0..%d", num_outputs - 1);
    }
    seen[o] = 1;
  }
  for (int i = 0; i < n_direct; ++i) {
    if (p->direct_f[i] < 0) {
      delete p;
      return tsim_fail(TSIM_EINVAL, "negative direct_f_indices[%d]", i);
    }
    p->max_f_index = std::max(p->max_f_index, p->direct_f[i]);
  }
  *out = p;
  return TSIM_OK;
}

extern "C" int tsim_program_add_component(tsim_program *p, int32_t n_out, const int32_t *output_indices,
                                          int32_t F, const int32_t *f_selection, int32_t n_levels) {
  if (!p) return tsim_fail(TSIM_EINVAL, "program is NULL");
  if (p->finalized) return tsim_fail(TSIM_ESTATE, "program already finalized");
  if (n_out < 0 || F < 0) return tsim_fail(TSIM_EINVAL, "negative n_out/F");
  if (n_levels != n_out + 1 && n_levels != 2)
    return tsim_fail(TSIM_EINVAL, "component with %d outputs needs %d (sequential) or 2 (joint) levels, got %d",
                n_out, n_out + 1, n_levels);
  if ((n_out > 0 && !output_indices) || (F > 0 && !f_selection)) return tsim_fail(TSIM_EINVAL, "NULL index array");
  HostComponent c;
  c.n_out = n_out;
  c.F = F;
  c.n_levels = n_levels;
  c.output_indices.assign(output_indices, output_indices + n_out);
  c.f_selection.assign(f_selection, f_selection + F);
  for (int i = 0; i < F; ++i) {
    if (c.f_selection[i] < 0) return tsim_fail(TSIM_EINVAL, "negative f_selection[%d]", i);
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
  if (!p || !L) return tsim_fail(TSIM_EINVAL, "NULL argument");
  if (p->finalized) return tsim_fail(TSIM_ESTATE, "program already finalized");
  if (component < 0 || component >= (int)p->comps.size()) return tsim_fail(TSIM_EINVAL, "bad component index %d", component);
  HostComponent &c = p->comps[component];
  if ((int)c.levels.size() >= c.n_levels) return tsim_fail(TSIM_EINVAL, "component %d already has all %d levels", component, c.n_levels);
  const int k = (int)c.levels.size();
  const bool sequential = (c.n_levels == c.n_out + 1);
  const int want_P = c.F + (sequential ? k : (k == 0 ? 0 : c.n_out));
  if (L->n_params != want_P)
    return tsim_fail(TSIM_EINVAL, "component %d level %d: n_params=%d, expected %d", component, k, L->n_params, want_P);
  if (L->num_graphs < 0 || L->ta < 0 || L->tb < 0 || L->tc < 0 || L->td < 0) return tsim_fail(TSIM_EINVAL, "negative sizes");
  if (L->n_params > TSIM_MAX_PARAMS)
    return tsim_fail(TSIM_ENOTSUP, "n_params=%d exceeds TSIM_MAX_PARAMS=%d", L->n_params, TSIM_MAX_PARAMS);
  const size_t G = L->num_graphs, P = L->n_params;
  if (G > 0) {
    const void *req[] = {L->phase_indices, L->floatfactor, L->power2};
    for (const void *q : req)
      if (!q) return tsim_fail(TSIM_EINVAL, "prefactor array is NULL");
    if ((L->ta && (!L->a_phases || !L->a_counts || (P && !L->a_params))) ||
        (L->tb && (!L->b_coeffs || (P && !L->b_params))) ||
        (L->tc && (!L->c_psi_const || !L->c_phi_const || (P && (!L->c_psi_params || !L->c_phi_params)))) ||
        (L->td && (!L->d_alpha || !L->d_beta || !L->d_counts || (P && (!L->d_alpha_params || !L->d_beta_params)))))
      return tsim_fail(TSIM_EINVAL, "term array is NULL");
    if (L->has_approx && !L->approx) return tsim_fail(TSIM_EINVAL, "has_approx set but approx is NULL");
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
    if (L->ta && (h.i32[0][g] < 0 || h.i32[0][g] > L->ta)) return tsim_fail(TSIM_EINVAL, "a_counts[%zu] out of range", g);
    if (L->td && (h.i32[1][g] < 0 || h.i32[1][g] > L->td)) return tsim_fail(TSIM_EINVAL, "d_counts[%zu] out of range", g);
  }
  c.levels.push_back(std::move(h));
  return TSIM_OK;
}
extern "C" int tsim_program_set_mode(tsim_program *p, int32_t mode) {
  if (!p) return tsim_fail(TSIM_EINVAL, "program is NULL");
  if (p->finalized) return tsim_fail(TSIM_ESTATE, "program already finalized");
  if (mode != TSIM_MODE_AUTO && mode != TSIM_MODE_FAITHFUL && mode != TSIM_MODE_ROW_KERNEL)
    return tsim_fail(TSIM_EINVAL, "bad mode %d", mode);
  p->mode = mode;
  return TSIM_OK;
}
extern "C" int tsim_program_set_pattern_tables(tsim_program *p, int32_t enable, int32_t max_weight) {
  if (!p) return tsim_fail(TSIM_EINVAL, "program is NULL");
  if (p->finalized) return tsim_fail(TSIM_ESTATE, "program already finalized");
  if (enable < -1 || enable > 1 || max_weight < -1 || max_weight > TSIMK_LW_MAX_WEIGHT)
    return tsim_fail(TSIM_EINVAL, "bad pattern-table setting (%d, %d)", enable, max_weight);
  p->lw_request = enable;
  p->lw_weight_cap = max_weight;
  return TSIM_OK;
}

extern "C" int tsim_program_pattern_table_info(const tsim_program *p, int32_t *enabled, int64_t *table_bytes,
                                               int32_t *max_weight) {
  if (!p) return tsim_fail(TSIM_EINVAL, "program is NULL");
  if (!p->finalized) return tsim_fail(TSIM_ESTATE, "program not finalized");
  if (enabled) *enabled = p->lw ? 1 : 0;
  if (table_bytes) *table_bytes = p->lw ? p->lw_bytes : 0;
  if (max_weight)
    for (size_t i = 0; i < p->comps.size(); ++i) max_weight[i] = p->lw ? p->lw_wmax[i] : -1;
  return TSIM_OK;
}
extern "C" int tsim_program_tables_pending(const tsim_program *p, int32_t *pending) {
  if (!p) return tsim_fail(TSIM_EINVAL, "program is NULL");
  if (!p->finalized) return tsim_fail(TSIM_ESTATE, "program not finalized");
  if (pending) *pending = p->ext_pending ? 1 : 0;
  return TSIM_OK;
}
// launch-plan feedback buffer (mapped pinned host memory the hard-row kernels write)
static int alloc_feedback(tsim_program *p) {
  if (p->v4 || p->lw_wide) {
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
  if (!p) return tsim_fail(TSIM_EINVAL, "program is NULL");
  if (p->finalized) return tsim_fail(TSIM_ESTATE, "program already finalized");
  {
    // Switches.  Results never depend on any of them.  Public: TSIM_AMD_ADAPTIVE, TSIM_AMD_FUSED_STEPS,
    // TSIM_AMD_DEEP_TABLES (and, read where they act, TSIM_AMD_MODE / _KERNEL / _PATTERN_TABLES / _PATTERN_TABLE_MB / _DEBUG).
    // Everything else the launch planner can be told - the A/B parameters tests/ and scripts/ use to run two paths against
    // each other - travels in ONE variable, TSIM_AMD_TUNE="key=value,key=value".
    auto env_int = [](const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; };
    const std::string tune = getenv("TSIM_AMD_TUNE") ? std::string(",") + getenv("TSIM_AMD_TUNE") + "," : std::string();
    auto tune_ll = [&](const char *key, long long dflt) -> long long {
      const std::string k = std::string(",") + key + "=";
      const size_t at = tune.find(k);
      return at == std::string::npos ? dflt : atoll(tune.c_str() + at + k.size());
    };
    p->knobs.adaptive = env_int("TSIM_AMD_ADAPTIVE", 1) != 0;
    p->knobs.fused_steps = env_int("TSIM_AMD_FUSED_STEPS", 1) != 0;
    p->knobs.deep_tables = env_int("TSIM_AMD_DEEP_TABLES", 0);
    p->knobs.defer = tune_ll("defer_hard", 1) != 0;
    p->knobs.defer_group = (int)std::max(0ll, std::min((long long)TSIMK_H_MAX_CTX, tune_ll("defer_group", 0)));  // 0: by table size, below
    p->knobs.lw_fast = tune_ll("lw_fast", 1) != 0;
    p->knobs.wide_fused = tune_ll("wide_fused", 1) != 0;
    p->knobs.hard_wave = tune_ll("hard_wave", 1) != 0;
    p->knobs.hard_wave_rows = (int)std::max(0ll, tune_ll("hard_wave_rows", 1024));
    p->knobs.hard_inline_rows = std::max(0ll, tune_ll("hard_inline_rows", 1ll << 40));
    p->knobs.hard_comp_par = tune_ll("hard_comp_par", 1) != 0;
    p->knobs.deep_after = (unsigned long long)std::max(0ll, tune_ll("deep_after", 0));  // 0: by the estimated build time (tsim_tables_deep_after)
    p->knobs.fused_lanes = (int)std::max(0ll, std::min(4ll, tune_ll("fused_lanes", 0)));
    p->knobs.fused_max = (int)std::max(1ll, std::min((long long)TSIMK_LWM_MAX_STEPS, tune_ll("fused_max", 8)));
    p->knobs.wide_tables = tune_ll("wide_tables", 1) != 0;
    p->knobs.wide_compact = tune_ll("wide_compact", 1) != 0;
    p->knobs.x3 = tune_ll("x3", 1) != 0;
    p->knobs.x4 = (int)std::max(0ll, tune_ll("x4", 32));
    p->knobs.wide_depth = (int)std::max(0ll, std::min(4ll, tune_ll("wide_depth", 4)));
    p->knobs.wide_passes = (int)std::max(1ll, std::min(16ll, tune_ll("wide_passes", 8)));
    p->knobs.gen = (int)std::max(0ll, std::min(2ll, tune_ll("gen", 1)));
    p->knobs.trie = (int)std::max(0ll, std::min(2ll, tune_ll("trie", 1)));
    p->knobs.noise_fused = tune_ll("noise_fused", 1) != 0;
    p->knobs.shallow = tune_ll("shallow", 1) != 0;
    p->knobs.hard_overflow = tune_ll("hard_overflow", 1) != 0;
  }

  const bool fin_dbg = tsim_debug("finalize");
  auto fin_t0 = std::chrono::steady_clock::now();
  auto fin_mark = [&](const char *what) {
    if (!fin_dbg) return;
    const auto n = std::chrono::steady_clock::now();
    unsigned long long hsh = 1469598103934665603ull;  // (FNV-1a of the image so far: a packer change that must not move a word shows here)
    if (tsim_debug("imghash"))
      for (uint32_t w : p->img) hsh = (hsh ^ w) * 1099511628211ull;
    fprintf(stderr, "[tsim] finalize: %s %.2f ms (image %zu words, hash %016llx)\n", what, std::chrono::duration<double, std::milli>(n - fin_t0).count(),
            p->img.size(), hsh);
    fin_t0 = std::chrono::steady_clock::now();
  };
  // ---- choose the evaluation formulation ----
  {
    const char *env = getenv("TSIM_AMD_MODE");
    bool fast = (p->mode != TSIM_MODE_FAITHFUL) && !(env && strcmp(env, "faithful") == 0);
    for (auto &c : p->comps)
      for (auto &lv : c.levels) fast = fast && level_fast_eligible(lv);
    p->fast = fast;
  }
retry_pack:
  p->hw_max_rows = 0;
  // ---- validate the output bookkeeping (pipeline.py:83-102) ----
  int pos = p->n_direct;
  for (size_t ci = 0; ci < p->comps.size(); ++ci) {
    HostComponent &c = p->comps[ci];
    if ((int)c.levels.size() != c.n_levels)
      return tsim_fail(TSIM_EINVAL, "component %zu has %zu of %d levels", ci, c.levels.size(), c.n_levels);
    for (int j = 0; j < c.n_out; ++j) {
      if (pos >= p->num_outputs || p->output_order[pos] != c.output_indices[j])
        return tsim_fail(TSIM_EINVAL, "output_order[%d] does not match component %zu output %d", pos, ci, j);
      ++pos;
    }
  }
  if (pos != p->num_outputs)
    return tsim_fail(TSIM_EINVAL, "direct entries + component outputs cover %d of %d outputs", pos, p->num_outputs);

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
  // The fast formulation's pack-time algebra of every level, the levels side by side (C4: seven levels of 16-256 graphs, 1.1-2 ms
  // each even with their graphs on eight threads - one after the other they were 12 of the 21 ms of a fresh handle)
  struct PrePacked { std::vector<uint32_t> tables; bool fixed = false, ok = true; int frame = 0; };
  std::vector<PrePacked> prepacked;
  if (p->fast) {
    std::vector<std::pair<HostLevel *, int>> jobs;
    for (auto &c : p->comps) {
      int maxP = 1;
      for (auto &lv : c.levels) maxP = std::max(maxP, lv.P);
      if (c.n_levels == c.n_out + 1) maxP = std::max(maxP, c.F + c.n_out);
      const int W = round_w((maxP + 31) / 32);
      for (auto &lv : c.levels) jobs.push_back({&lv, W});
    }
    prepacked.resize(jobs.size());
    // (up to 16 levels at a time on the process-wide pool, each level's graphs through the same pool: tsim_pool.cpp)
    tsim_parallel_for(jobs.size(), 16, [&](size_t i) {
      if (jobs[i].second < 0) { prepacked[i].ok = true; return; }  // (reported below: too many parameters)
      prepacked[i].ok = pack_level_fast(*jobs[i].first, jobs[i].second, prepacked[i].tables, prepacked[i].fixed, prepacked[i].frame);
    });
  }
  fin_mark("  levels packed (pool)");
  // The bulk of the image - graph records, rows, term tables, the block-per-row kernel's row streams, the chunk tables below - is
  // laid out here (space reserved, zero-filled) and copied in by the pool afterwards: one thread appending 12 MB piece by piece
  // was 5 of the 12 ms of a fresh C4 handle.
  bool sum_wrap = false;  // some level's reference sum cannot be ruled out to wrap int32 (pack_level_fast)
  std::vector<std::function<void()>> copy_jobs;
  auto img_grow = [&](size_t n) {
    const size_t o = img.size();
    img.resize(o + n, 0u);
    return o;
  };
  auto img_align16 = [&]() { img.resize((img.size() + 15) / 16 * 16, 0u); };
  auto copy_later = [&](const std::vector<uint32_t> &v) {
    const size_t o = img_grow(v.size());
    if (!v.empty()) copy_jobs.push_back([&img, &v, o]() { memcpy(&img[o], v.data(), v.size() * sizeof(uint32_t)); });
    return o;
  };
  auto run_copy_jobs = [&]() {
    tsim_parallel_for(copy_jobs.size(), 16, [&](size_t i) { copy_jobs[i](); });
    copy_jobs.clear();
  };
  if (p->fast) {  // (an upper bound: address space only - no reallocation while the pieces are laid out)
    size_t need = img.size() + 4096, li = 0;
    for (auto &c : p->comps) {
      need += (size_t)c.F + (size_t)c.n_out + (size_t)c.n_levels * (L_WORDS + L4_WORDS) + 256;
      for (auto &lv : c.levels) {
        need += lv.graph_rec.size() + 3 * lv.rows.size() + prepacked[li++].tables.size() + 256;
        need += (size_t)lv.G * G4_WORDS + (size_t)((lv.G + 3) / 4) * (32 * 16 * 4 * 4 + (size_t)std::max(TSIMK_SPARSE_ENTRIES, c.F + 33) * 16) + 64;
      }
    }
    img.reserve(need);
  }
  for (size_t ci = 0; ci < p->comps.size(); ++ci) {
    HostComponent &c = p->comps[ci];
    int maxP = 1;
    for (auto &lv : c.levels) maxP = std::max(maxP, lv.P);
    // sampling appends the trial bit at position F+i (< F+n_out)
    const bool sequential = (c.n_levels == c.n_out + 1);
    if (sequential) maxP = std::max(maxP, c.F + c.n_out);
    const int W = round_w((maxP + 31) / 32);
    if (W < 0) return tsim_fail(TSIM_ENOTSUP, "component %zu needs %d parameter bits (max %d)", ci, maxP, TSIM_MAX_PARAMS);
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
      static const std::vector<uint32_t> no_tables;
      const std::vector<uint32_t> *tables_p = &no_tables;
      bool fixed = false;
      int frame = 0;
      if (p->fast) {
        // (packed by the pre-pass above, all levels of the program side by side)
        PrePacked &pp = prepacked[(size_t)p->level_off.size()];
        if (!pp.ok) {  // a table entry exceeds int32: use the faithful layout
          p->fast = false;
          goto retry_pack;
        }
        tables_p = &pp.tables;
        fixed = pp.fixed;
        frame = pp.frame;
      } else {
        pack_level(h, W);
      }
      const std::vector<uint32_t> &tables = *tables_p;
      // align graph records to 16 words (one s_load_dwordx16 each)
      img_align16();
      const uint32_t goff = (uint32_t)img_grow(h.graph_rec.size());  // (copied by the job below, which also patches the offsets in)
      const uint32_t roff = (uint32_t)copy_later(h.rows);
      static_assert((int)G_ROWS == (int)GF_ROWS, "row offset slot is shared by both layouts");
      uint32_t toff = 0;
      if (p->fast) {
        img_align16();  // 64-byte aligned table entries (uint4 loads)
        toff = (uint32_t)copy_later(tables);
        h.tt_off = toff;
        h.tt_words = (uint32_t)tables.size();
      }
      // The same rows once more as ONE stream of uniform stride per level - [const, w_0 .. w_(W-1)] per row, graph after
      // graph in the order eval_graph_fast reads them - for the wave-per-row kernel (tsim_kernel_hw.hip.h), whose lanes
      // take one ROW each (coalesced) and whose graphs then read their parities by position.
      uint32_t hw_off = 0, hw_n = 0;
      std::vector<uint32_t> hw_start;
      if (p->fast) {
        img_align16();
        hw_start.resize((size_t)h.G);
        for (int g = 0; g < h.G; ++g) {
          const uint32_t *r = &h.graph_rec[(size_t)g * G_WORDS];
          const uint32_t n_meta = (r[GF_N01] & 0xFFFFu) + (r[GF_N01] >> 16) + (r[GF_N3H] & 0xFFFFu);
          const uint32_t n_plain = 2u * r[GF_ND] + ((r[GF_FLAGS] & TSIMK_GFLAG_LAM) ? 1u : 0u) + ((r[GF_FLAGS] & TSIMK_GFLAG_LIN) ? 1u : 0u) +
                                   2u * (r[GF_N3H] >> 16);
          hw_start[(size_t)g] = hw_n;
          hw_n += n_meta + n_plain;
        }
        hw_off = (uint32_t)img_grow((size_t)hw_n * (size_t)(1 + W));
        p->hw_max_rows = std::max(p->hw_max_rows, (long long)hw_n);
      }
      {  // (behind the copies of this level's records: the offsets of the image, the row streams)
        const HostLevel *hp = &h;
        const bool fast = p->fast;
        copy_jobs.push_back([&img, hp, goff, roff, toff, hw_off, W, fast, hs = std::move(hw_start)]() {
          const HostLevel &h = *hp;
          if (!h.graph_rec.empty()) memcpy(&img[goff], h.graph_rec.data(), h.graph_rec.size() * sizeof(uint32_t));
          for (int g = 0; g < h.G; ++g) {
            uint32_t *gr = &img[goff + (size_t)g * G_WORDS];
            const uint32_t *r = &h.graph_rec[(size_t)g * G_WORDS];
            gr[G_ROWS] = r[G_ROWS] + roff;
            if (!fast) continue;
            gr[GF_TBL] = r[GF_TBL] + toff;
            if (r[GF_TBL2]) gr[GF_TBL2] = r[GF_TBL2] + toff;
            gr[GF_HWROW] = hs[(size_t)g];
            const uint32_t n_meta = (r[GF_N01] & 0xFFFFu) + (r[GF_N01] >> 16) + (r[GF_N3H] & 0xFFFFu);
            const uint32_t n_plain = 2u * r[GF_ND] + ((r[GF_FLAGS] & TSIMK_GFLAG_LAM) ? 1u : 0u) + ((r[GF_FLAGS] & TSIMK_GFLAG_LIN) ? 1u : 0u) +
                                     2u * (r[GF_N3H] >> 16);
            const uint32_t *src = &h.rows[r[GF_ROWS]];
            uint32_t *dst = &img[hw_off + (size_t)hs[(size_t)g] * (size_t)(1 + W)];
            memcpy(dst, src, (size_t)n_meta * (size_t)(1 + W) * sizeof(uint32_t));
            src += (size_t)n_meta * (size_t)(1 + W);
            dst += (size_t)n_meta * (size_t)(1 + W);
            for (uint32_t t = 0; t < n_plain; ++t, src += W, dst += 1 + W) {
              dst[0] = 0u;
              memcpy(dst + 1, src, (size_t)W * sizeof(uint32_t));
            }
          }
        });
      }
      uint32_t *lr = &img[lrec + (size_t)k * L_WORDS];
      lr[L_G] = (uint32_t)h.G;
      lr[L_GRAPHS] = goff;
      lr[L_HWROWS] = hw_off;
      lr[L_HWN] = hw_n;
      lr[L_FLAGS] = (h.approx ? TSIMK_LFLAG_APPROX : 0u) | (fixed ? TSIMK_LFLAG_FIXED : 0u);
      lr[L_FRAME] = (uint32_t)frame;
      p->stats[1] += 1;
      p->stats[2] += fixed ? 1 : 0;
      if (p->fast && h.sum_wrap_possible) sum_wrap = true;
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
  {  // TSIM_AMD_MODE=strict: the exact formulation only where NO int32 operation of the reference can wrap - products (above) and
    // the levels' running sums (pack_level_fast: sum_wrap_possible); everything else on the faithful formulation
    const char *env = getenv("TSIM_AMD_MODE");
    if (p->fast && sum_wrap && env && strcmp(env, "strict") == 0) {
      p->fast = false;
      goto retry_pack;
    }
  }
  fin_mark("  records laid out");
  run_copy_jobs();
  fin_mark("rows / fast formulation packed");
  // ---- v4 (chunk table) layout, when every sampled component qualifies ----
  p->v4 = false;
  p->v4w = false;
  if (p->fast && p->sampleable) {
    bool ok = !p->comps.empty();
    for (auto &c : p->comps) {
      for (auto &lv : c.levels) ok = ok && level_v4_eligible(lv);
      // more than 64 parameters (round 5, knobs.x3): only as f_sel (<= 64 bits, words 0 and 1 of x) + outcome bits of a sequential component
      int maxp = 1;
      for (auto &lv : c.levels) maxp = std::max(maxp, lv.P);
      // (81..128, or more than 64 SELECTED bits - knobs.x4, end of round 5: four words, and the first pass must be k_sample_gen)
      // Four words only for components of MANY graphs (knobs.x4 = the least number, 32): those are what k_sample_wide cannot hold
      // (column tables beyond the LDS, a chain of graphs per dense pass); a 100-bit component of 8 graphs is better off there.
      int gtot = 0;
      for (auto &lv : c.levels) gtot += lv.G;
      if (maxp > 64) ok = ok && p->knobs.x3 && c.n_levels == c.n_out + 1 && ((c.F <= 64 && maxp <= 80) || (p->knobs.x4 > 0 && gtot >= p->knobs.x4 && maxp <= 128));
    }
    // wide components (more than 64 parameters): column tables only, for the sparse-column kernel k_sample4w.
    // Needs sequential components of at most 8 outputs over at most 256 ascending f indices below 512.
    // (f indices up to 2047 since round 5: the round-2 kernels k_sample4w / k_sample_lw<true> read 16 mask words - f rows of at most
    // 512 bits, p->wide_big below keeps them away from wider rows - k_sample_wide reads a list of the words that hold selected bits)
    bool wide = !ok && !p->comps.empty() && p->max_f_index < 2048;
    for (auto &c : p->comps) {
      wide = wide && (c.n_levels == c.n_out + 1) && c.n_out <= 8 && c.F <= TSIMK_LWW_MAX_F;
      for (int j = 1; j < c.F; ++j) wide = wide && c.f_selection[j] > c.f_selection[j - 1];
      for (auto &lv : c.levels) wide = wide && level_v4_eligible(lv, true);
    }
    const char *kenv = getenv("TSIM_AMD_KERNEL");
    if (kenv && strcmp(kenv, "v3") == 0) ok = wide = false;
    if (p->mode == TSIM_MODE_ROW_KERNEL) ok = wide = false;
    p->v4_gt = 4;
    if (ok || wide) {
      while (img.size() % 16) img.push_back(0u);
      p->comp4_off = (int)img.size();
      img.resize(img.size() + p->comps.size() * C4_WORDS, 0u);
      p->v4_max_sent = 0;
      int maxp = 1;
      for (auto &c : p->comps)
        for (auto &lv : c.levels) maxp = std::max(maxp, lv.P);
      static const int kNch[] = {2, 4, 6, 8, 10, 12, 14, 16, 20, 32};
      p->v4_max_nch = 32;
      for (int v : kNch)
        if (4 * v >= maxp) { p->v4_max_nch = v; break; }
      if (wide) {  // no chunk tables at all, one graph per tile (column tables only)
        p->v4_max_nch = 0;
        p->v4_gt = 1;
      }
      // the chunk / column tables of every level, emitted side by side (C4: 3.5 ms one after the other), assembled below
      struct Emit4 { std::vector<uint32_t> recs4, tabs4, stabs4; int nch = 1, ntiles = 0, sparse_F = -1; };
      std::vector<std::vector<Emit4>> emitted(p->comps.size());
      {
        std::vector<std::pair<size_t, int>> ejobs;
        for (size_t ci = 0; ci < p->comps.size(); ++ci) {
          HostComponent &c = p->comps[ci];
          emitted[ci].resize((size_t)c.n_levels);
          for (int k = 0; k < c.n_levels; ++k) {
            const bool sequential = (c.n_levels == c.n_out + 1);
            emitted[ci][(size_t)k].sparse_F = (wide || (sequential && c.n_out <= 8 && c.F + c.n_out <= 64)) ? c.F : -1;
            ejobs.push_back({ci, k});
          }
        }
        fin_mark("  chunk-table setup");
        tsim_parallel_for(ejobs.size(), 16, [&](size_t j) {
          const size_t ci = ejobs[j].first;
          const int k = ejobs[j].second;
          HostLevel &h = p->comps[ci].levels[(size_t)k];
          Emit4 &e = emitted[ci][(size_t)k];
          const uint32_t v3recs = img[(size_t)p->level_off[p->level_base[ci] + k] + L_GRAPHS];
          std::vector<uint32_t> v3copy(img.begin() + v3recs, img.begin() + v3recs + (size_t)h.G * G_WORDS);
          emit_level4(h, p->v4_gt, p->v4_max_nch, v3copy.data(), e.recs4, e.tabs4, e.nch, e.ntiles, e.sparse_F, e.stabs4);
        });
      }
      fin_mark("  chunk tables emitted (pool)");
      for (size_t ci = 0; ci < p->comps.size(); ++ci) {
        HostComponent &c = p->comps[ci];
        for (int w = 0; w < 8; ++w) img[p->comp4_off + ci * C4_WORDS + w] = img[p->comp_off + ci * C_WORDS + w];
        img_align16();
        const size_t l4 = img.size();
        img[p->comp4_off + ci * C4_WORDS + C4_LEVELS] = (uint32_t)l4;
        img.resize(img.size() + (size_t)c.n_levels * L4_WORDS, 0u);
        for (int k = 0; k < c.n_levels; ++k) {
          HostLevel &h = c.levels[k];
          const uint32_t v3lvl = (uint32_t)p->level_off[p->level_base[ci] + k];
          const uint32_t v3recs = img[v3lvl + L_GRAPHS];
          Emit4 &em = emitted[ci][(size_t)k];
          std::vector<uint32_t> &recs4 = em.recs4, &tabs4 = em.tabs4, &stabs4 = em.stabs4;
          const int nch = em.nch, ntiles = em.ntiles, sparse_F = em.sparse_F;
          img_align16();
          const uint32_t roff = (uint32_t)copy_later(recs4);
          img_align16();
          const uint32_t toff = (uint32_t)copy_later(tabs4);
          img_align16();
          const uint32_t stoff = stabs4.empty() ? 0u : (uint32_t)img.size();
          copy_later(stabs4);
          p->v4_max_sent = std::max(p->v4_max_sent, sparse_F < 0 ? 0 : (p->v4_gt > 1 ? std::max(TSIMK_SPARSE_ENTRIES, sparse_F + 33) : sparse_F + 33));
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
        if (wide) {  // selection masks over f bits 0..255 and, per word, the selected bits below it
          uint32_t sel[2 * TSIMK_W_SELWORDS] = {0};
          for (int v : c.f_selection)
            if (v < 32 * TSIMK_W_SELWORDS) sel[v >> 5] |= 1u << (v & 31);
          for (int w = 1; w < TSIMK_W_SELWORDS; ++w)
            sel[TSIMK_W_SELWORDS + w] = sel[TSIMK_W_SELWORDS + w - 1] + (uint32_t)__builtin_popcount(sel[w - 1]);
          while (img.size() % 16) img.push_back(0u);
          img[p->comp4_off + ci * C4_WORDS + C4_SELMASK] = (uint32_t)img.size();
          img.insert(img.end(), sel, sel + 2 * TSIMK_W_SELWORDS);
        }
      }
      run_copy_jobs();
      p->v4 = ok;
      p->v4w = wide;
      // f rows of more than 512 bits: k_sample_wide or the row kernel, never the round-2 kernels (16 mask words)
      p->wide_big = wide && p->max_f_index >= 32 * TSIMK_W_SELWORDS;
      for (auto &c : p->comps) p->wide_big = p->wide_big || (wide && c.F > 255);  // (positions are bytes in the round-2 kernels too)
      p->v4w_resident_bytes = 0;
      if (wide)
        for (auto &c : p->comps) {
          size_t tot = 0;
          for (auto &lv : c.levels) tot += (size_t)lv.G * (size_t)(c.F + 33) * 16;
          p->v4w_resident_bytes = std::max(p->v4w_resident_bytes, tot);
        }
    }
  }
  p->stats[7] = (p->v4 ? 1 : (p->v4w ? 2 : 0)) | (sum_wrap ? 64 : 0);
  if (fin_dbg) fprintf(stderr, "[tsim] finalize: the reference's int32 sums %s\n", sum_wrap ? "CAN wrap on some level (sum_wrap_possible)" : "cannot wrap");

  fin_mark("chunk / column tables");
  // ---- direct outputs as a gather program (bit-field runs), for every sampling kernel ----
  {
    std::vector<std::array<int, 3>> de;
    for (int j = 0; j < p->n_direct; ++j) de.push_back({p->direct_f[j], p->output_order[j], p->direct_flips[j] ? 1 : 0});
    std::vector<uint32_t> prog = emit_gather_program(de);
    while (img.size() % 16) img.push_back(0u);
    p->lw_direct_prog = (int)img.size();
    p->lw_direct_chunks = (int)(prog.size() / 16);
    img.insert(img.end(), prog.begin(), prog.end());
    p->lw_direct_rot = 0;
    if (p->max_f_index < 128 && p->num_outputs <= 64) {
      prog = emit_rotmask_program(de);
      p->lw_direct_rot = (int)img.size();
      img.insert(img.end(), prog.begin(), prog.end());
    }
  }
  // ---- low-weight pattern tables: plan (records + sizes); built on the device after upload ----
  p->lw = false;
  p->lw_wmax.clear();
  p->lw_bytes = 0;
  std::vector<std::vector<std::array<int, 3>>> lw_fsel_progs;
  {
    bool want = p->lw_request < 0 ? (p->mode == TSIM_MODE_AUTO) : (p->lw_request != 0);
    if (const char *e = getenv("TSIM_AMD_PATTERN_TABLES")) want = atoi(e) != 0;
    bool ok = want && p->sampleable && !p->comps.empty();
    bool narrow = true;
    // (F + n_out up to 80 with F <= 64 since round 5: the chunk-table kernels hold x in three words then - p->v4 says whether they took the program)
    // Components of more than TSIMK_LW_MAX_NOUT outputs (round 6): a pattern's 2^n_out thresholds no longer fit a row - their
    // tables are pruned prefix trees in chunks of three levels (tsim_trie.hip.h), walked by k_sample_gen alone.  Without chunk
    // tables (p->v4: more than 32 counted rows in some graph, say) their hard rows are the block-per-row kernel's, whose x
    // holds up to 128 parameters (k_sample_hw<4, 0>: launch_hw)
    int cw_max = 1;
    for (int w : p->comp_w) cw_max = std::max(cw_max, w);
    const bool hw_rows_ok = p->fast && cw_max <= 4 && p->num_outputs <= 128 && p->max_f_index < 2048 && p->hw_max_rows < 60000;
    p->lw_trie = false;
    for (auto &c : p->comps) {
      c.trie = (c.n_out > TSIMK_LW_MAX_NOUT && p->knobs.trie) || (p->knobs.trie == 2 && c.n_out >= 1);  // (trie=2: every component - tests)
      narrow = narrow && (c.n_levels == c.n_out + 1) && c.n_out <= (p->knobs.trie ? 64 : TSIMK_LW_MAX_NOUT) &&
               (c.F + c.n_out <= 64 || (p->v4 && c.F <= 64 && c.F + c.n_out <= 80) || (p->v4 && p->knobs.x4 > 0 && c.F + c.n_out <= 128) ||
                (c.trie && !p->v4 && hw_rows_ok && c.F + c.n_out <= 128));
      p->lw_trie = p->lw_trie || c.trie;
    }
    if (p->lw_trie && !p->v4 && !hw_rows_ok) narrow = false;  // (nothing to serve the hard rows of a fused group)
    if (tsim_debug("tables"))
      fprintf(stderr, "[tsim] pattern tables: narrow %d trie %d (chunk tables %d, block-per-row hard rows possible %d: fast %d, x words %d, outputs %d, f index %d, rows per level %lld)\n",
              narrow ? 1 : 0, p->lw_trie ? 1 : 0, p->v4 ? 1 : 0, hw_rows_ok ? 1 : 0, p->fast ? 1 : 0, cw_max, p->num_outputs, p->max_f_index, p->hw_max_rows);
    // more than 64 selected bits somewhere in a narrow program: every first pass but k_sample_gen holds f_sel in 64 bits, and the
    // closed-form binomials of the table builder end at b = 64 (tsim_lw.hip.h) - tables to weight 4 through the [4][256]
    // table of the wide path, gen or the full kernel (launch_sample)
    p->narrow_big = false;
    if (narrow)
      for (auto &c : p->comps) p->narrow_big = p->narrow_big || c.F > 64;
    // wide components (the sparse-column kernel's programs): tables to weight TSIMK_LWW_MAX_WEIGHT in front of it
    p->lw_wide = false;
    if (ok && !narrow && p->v4w && p->max_f_index < 2048) {
      bool wide_ok = true;
      wide_ok = p->knobs.wide_tables;
      for (auto &c : p->comps) {
        wide_ok = wide_ok && (c.n_levels == c.n_out + 1) && c.n_out <= 8 && c.F <= TSIMK_LWW_MAX_F;
        for (int j = 1; j < c.F; ++j) wide_ok = wide_ok && c.f_selection[j] > c.f_selection[j - 1];
      }
      p->lw_wide = wide_ok;
    }
    ok = ok && (narrow || p->lw_wide);
    if (ok) {
      // depth: the caller's, or 5 now and up to TSIMK_LW_MAX_WEIGHT on demand (tsim_tables.hip).  Patterns are
      // stored weight by weight, so the rows most shots read (weight 0..2) are a small cache-resident prefix
      // whatever the total; the heavier tail is read rarely.
      const bool pinned = p->lw_weight_cap >= 0;
      const int hw_cap = p->lw_wide ? TSIMK_LWW_MAX_WEIGHT : TSIMK_LW_MAX_WEIGHT;
      p->lw_cap_max = pinned ? std::min(p->lw_weight_cap, hw_cap) : hw_cap;
      // (wide components: weight 4 since round 5 - C5's 2.1 GB lift the tabulated share of its rows from 43 % to 63 %, 46 -> 41 us
      // per 10^6 shots; finalize still builds weight 3 only, the rest follows in the background like every default depth.
      // `wide_depth=3` keeps the round-4 default: the deeper table then waits for deep_after rows as before)
      p->lw_cap_default = pinned ? p->lw_cap_max : std::min(p->lw_wide ? p->knobs.wide_depth : 5, p->lw_cap_max);
      p->lw_cap_now = p->lw_cap_default;
      p->lw_budget = 4096ll << 20;  // per component: HBM is 288 GB, and only the prefix is hot (C3's weight-6 table is 1.8 GB)
      // wide components: C(200, 4) patterns of 8 thresholds are 2.1 GB - what lifts the tabulated share of C5's
      // shots from 43 % to 63 %; still under 1 % of the HBM
      if (p->lw_wide) p->lw_budget = 4096ll << 20;
      p->lw_trie_budget = 512ll << 20;
      if (const char *e = getenv("TSIM_AMD_PATTERN_TABLE_MB")) p->lw_budget = p->lw_trie_budget = std::max(1ll, atoll(e)) << 20;
      while (img.size() % 32) img.push_back(0u);
      p->lw_off = (int)img.size();
      img.resize(img.size() + p->comps.size() * LW_WORDS, 0u);  // records (selection masks and bases inline)
      for (size_t ci = 0; ci < p->comps.size(); ++ci) {
        const HostComponent &c = p->comps[ci];
        const uint32_t *crec = &img[p->comp_off + ci * C_WORDS];
        uint32_t *r = &img[p->lw_off + ci * LW_WORDS];
        r[LW_NOUT] = (uint32_t)c.n_out;
        r[LW_F] = (uint32_t)c.F;
        r[LW_OUTPOS] = crec[C_OUTPOS];
        r[LW_KEYBASE] = crec[C_KEYBASE];
        r[LW_BASES] = (uint32_t)(p->lw_off + ci * LW_WORDS + LW_BASES_INLINE);
        lw_fsel_progs.push_back({});
        for (int j = 0; j < c.F; ++j) lw_fsel_progs.back().push_back({c.f_selection[j], j, 0});
      }
      // Start shallow: the BASELINE jobs are 10^5-10^6 shots, and a depth-5 table (C4: 195 MB, 35 ms of
      // build) pays only from ~10^8 rows on.  Finalize builds the deepest tables that cost a millisecond or two (table
      // entries x graphs per level <= 1.2e8, ~2.4e-8 ms each: C2 depth 4 in 0.7 ms, C3 depth 3, C4 depth 3 in 2.4 ms - a third of
      // its fresh handle; at depth 2 it starts 2 ms sooner, and a job enqueued in one go, whose launch plans are all drawn before
      // any deeper table lands, pays for it: 10^8 shots in 46-75 instead of 32 ms, 4.6 % of the rows hard - also with the next
      // depth alone as a first stage of the background build, measured); the default depth follows in the background, slice by slice
      // next to the first launches (tsim_tables_extend_begin below, tsim_tables.hip).  A caller who named a depth gets it here.
      if (!pinned && p->knobs.shallow && p->knobs.deep_tables <= 0) {
        while (p->lw_cap_now > (p->lw_wide ? 3 : 2)) {
          TsimTablePlan t;
          std::vector<uint32_t> keep(img.begin() + p->lw_off, img.begin() + p->lw_off + (long)(p->comps.size() * LW_WORDS));
          double cost = 0;
          if (tsim_tables_plan_at(p, p->lw_cap_now, p->lw_budget, p->lw_off, t))
            for (size_t ci = 0; ci < p->comps.size(); ++ci) {
              double g = 0;
              for (auto &lv : p->comps[ci].levels) g += lv.G;
              cost += (double)(p->comps[ci].trie ? std::min(t.chunks[ci], t.npat[ci] * 64) * 8 : (t.npat[ci] << p->comps[ci].n_out)) * g / (double)p->comps[ci].levels.size();
            }
          std::copy(keep.begin(), keep.end(), img.begin() + p->lw_off);
          if (cost <= 1.2e8) break;
          --p->lw_cap_now;
        }
      }
      ok = tsim_tables_plan(p, p->lw_cap_now, p->lw_budget);
      if (ok && p->lw_wide) {
        p->lw = true;
        p->lw_reg = false;
        // selection masks: the comp4 record's (16 mask words + 16 prefix counts); binomials C(b, k + 1), k < 4, b < 256
        for (size_t ci = 0; ci < p->comps.size(); ++ci)
          img[p->lw_off + ci * LW_WORDS + LW_SELMASK] = img[p->comp4_off + ci * C4_WORDS + C4_SELMASK];
        while (img.size() % 16) img.push_back(0u);
        p->lw_binom_off = (int)img.size();
        p->lw_binom_stride = 256;
        for (auto &c : p->comps)
          if (c.F > 255) p->lw_binom_stride = 512;
        for (int k = 0; k < 4; ++k)
          for (int b = 0; b < p->lw_binom_stride; ++b) {
            unsigned long long c = 1;
            for (int i = 1; i <= k + 1; ++i) c = c * (unsigned long long)(b - (k + 1) + i > 0 ? b - (k + 1) + i : 0) / (unsigned long long)i;
            img.push_back(b >= k + 1 ? (uint32_t)c : 0u);
          }
        // The wide record (k_sample_wide, tsim_wide.hip.h): ONE component.  Direct outputs as rotate-and-mask runs sorted by
        // destination word, constant flips per word, the placement table LUT[leaf][word] of the sampled bits (leaf = the bits
        // in sampling order, first output most significant) and the mask of the words that hold component outputs.
        // Several components (round 5): one record per component, one k_sample_wide pass each, in stream order.  The first
        // pass writes whole rows (direct outputs + its component); the records of the later ones carry no runs and no flips
        // and WR_MERGE: their passes OR the component's bits into the rows in place (words are private to a row).
        p->wr_off = 0;
        p->wr_offs.clear();
        const int wo32 = 2 * ((p->num_outputs + 63) / 64);
        bool wide_rec_ok = !p->comps.empty() && p->comps.size() <= (size_t)p->knobs.wide_passes && wo32 >= 2 && wo32 <= 8;
        for (auto &c : p->comps) wide_rec_ok = wide_rec_ok && c.n_out >= 1 && c.n_out <= 8;
        for (size_t wci = 0; wide_rec_ok && wci < p->comps.size(); ++wci) {
          const HostComponent &c = p->comps[wci];
          const bool merge = wci > 0;
          std::vector<uint32_t> runs, runb(1, 0u), flips((size_t)wo32, 0u);
          for (int d = 0; d < wo32; ++d) {
            for (int sw = 0; !merge && sw <= p->max_f_index / 32 && sw < 256; ++sw) {
              std::vector<std::array<int, 2>> m;  // (src bit, dst bit) inside the words
              for (int j = 0; j < p->n_direct; ++j) {
                const int src = p->direct_f[j], dst = p->output_order[j];
                if ((src >> 5) != sw || (dst >> 5) != d) continue;
                m.push_back({src & 31, dst & 31});
                if (p->direct_flips[j]) flips[(size_t)d] |= 1u << (dst & 31);
              }
              std::sort(m.begin(), m.end());
              for (size_t i = 0; i < m.size();) {
                size_t j = i + 1;
                while (j < m.size() && m[j][0] == m[j - 1][0] + 1 && m[j][1] == m[j - 1][1] + 1) ++j;
                const int len = (int)(j - i);
                const uint32_t field = (len >= 32 ? 0xFFFFFFFFu : ((1u << len) - 1u)) << m[i][1];
                runs.push_back((uint32_t)sw | ((uint32_t)((m[i][0] - m[i][1]) & 31) << 8));
                runs.push_back(field);
                i = j;
              }
            }
            runb.push_back((uint32_t)(runs.size() / 2));
          }
          if (runs.size() / 2 > TSIMK_WIDE_MAX_RUNS) {
            wide_rec_ok = false;
            p->wr_offs.clear();
            break;
          }
          {
            while (img.size() % 16) img.push_back(0u);
            const int wr_this = (int)img.size();
            p->wr_offs.push_back(wr_this);
            img.resize(img.size() + WR_WORDS, 0u);
            img[(size_t)wr_this + WR_MERGE] = merge ? 1u : 0u;
            img[(size_t)wr_this + WR_KEYSUB] = p->total_keys > TSIMK_LWM_KEYS ? img[p->comp_off + wci * C_WORDS + C_KEYBASE] : 0u;
            img[(size_t)wr_this + WR_BSTRIDE] = (uint32_t)p->lw_binom_stride;
            {  // the f words that hold selected bits: masks, then (selected bits in the lower words | word index << 16)
              std::vector<uint32_t> words, masks;
              for (int v : c.f_selection) {
                if (words.empty() || words.back() != (uint32_t)(v >> 5)) { words.push_back((uint32_t)(v >> 5)); masks.push_back(0u); }
                masks.back() |= 1u << (v & 31);
              }
              if (words.empty()) { words.push_back(0u); masks.push_back(0u); }
              if (words.size() > TSIMK_WIDE_SELMAX) wide_rec_ok = false;
              const uint32_t sel_off = (uint32_t)img.size();
              img.insert(img.end(), masks.begin(), masks.end());
              uint32_t below = 0;
              for (size_t k = 0; k < words.size(); ++k) {
                img.push_back(below | (words[k] << 16));
                below += (uint32_t)__builtin_popcount(masks[k]);
              }
              img[(size_t)wr_this + WR_SELN] = (uint32_t)words.size();
              img[(size_t)wr_this + WR_SELREC] = sel_off;
            }
            const uint32_t runs_off = (uint32_t)img.size();
            img.insert(img.end(), runs.begin(), runs.end());
            const uint32_t runb_off = (uint32_t)img.size();
            img.insert(img.end(), runb.begin(), runb.end());
            const uint32_t flips_off = (uint32_t)img.size();
            img.insert(img.end(), flips.begin(), flips.end());
            const uint32_t lut_off = (uint32_t)img.size();
            uint32_t lutmask = 0u;
            for (uint32_t leaf = 0; leaf < (1u << c.n_out); ++leaf) {
              std::vector<uint32_t> w((size_t)wo32, 0u);
              for (int i = 0; i < c.n_out; ++i)
                if ((leaf >> (c.n_out - 1 - i)) & 1u) {
                  const int dst = c.output_indices[i];
                  w[(size_t)(dst >> 5)] |= 1u << (dst & 31);
                  lutmask |= 1u << (dst >> 5);
                }
              img.insert(img.end(), w.begin(), w.end());
            }
            size_t colbytes = 0;
            for (auto &lv : c.levels) colbytes += (size_t)lv.G * (size_t)(c.F + 33) * 16;
            // Shared column table: when the parity words of ALL graphs of all levels fit one 16-byte entry (graph g owns a bit
            // field of one 32-bit word: product rows U, V, counted rows, PhasePairs index bits, lambda, linear), a row's
            // Y words for every graph are ONE walk over its set bits instead of one walk per graph.  Field layout from
            // bit 0: U[h2] V[h2] O1[nc] D[2 nD] lambda lin.
            uint32_t ccol_off = 0, crec_off = 0;
            {
              struct Fld { int w, off, h2, nc; };
              std::vector<Fld> fl;
              int fill[4] = {0, 0, 0, 0};
              bool okc = p->knobs.wide_compact;
              for (auto &lv : c.levels)
                for (auto &fg : lv.fg) {
                  const int h2 = (int)fg.us.size(), nc = (int)(fg.c0.size() + fg.c1.size() + fg.c3.size());
                  const int width = 2 * h2 + nc + 2 * fg.nD + 2;
                  int w = -1;
                  for (int k = 0; k < 4 && w < 0 && width <= 32; ++k)
                    if (fill[k] + width <= 32) w = k;
                  if (w < 0) { okc = false; break; }
                  fl.push_back({w, fill[w], h2, nc});
                  fill[w] += width;
                }
              if (tsim_debug("pack")) {
                size_t gi = 0;
                for (size_t k = 0; k < c.levels.size(); ++k)
                  for (size_t g = 0; g < c.levels[k].fg.size(); ++g, ++gi) {
                    const FastGraph &fg = c.levels[k].fg[g];
                    fprintf(stderr, "[tsim] wide level %zu graph %zu: product pairs %zu, counted rows %zu + %zu + %zu, PhasePairs terms %d (tabled %d), term table %u words", k, g,
                            fg.us.size(), fg.c0.size(), fg.c1.size(), fg.c3.size(), fg.nD, fg.d_tabled ? 1 : 0, c.levels[k].tt_words);
                    if (okc) fprintf(stderr, ", shared entry word %d bit %d", fl[gi].w, fl[gi].off);
                    fprintf(stderr, "\n");
                  }
              }
              if (okc) {
                const int F = c.F;
                std::vector<std::array<uint32_t, 4>> cc((size_t)F + 33, std::array<uint32_t, 4>{0u, 0u, 0u, 0u});
                size_t gi = 0;
                for (auto &lv : c.levels)
                  for (auto &fg : lv.fg) {
                    const Fld f = fl[gi++];
                    const int P = lv.P;
                    auto place = [&](const std::vector<uint64_t> &m, int bit, bool cst) {
                      const uint32_t b = 1u << (f.off + bit);
                      for (int i = 0; i < F && i < P; ++i)
                        if ((m[(size_t)i >> 6] >> (i & 63)) & 1) cc[(size_t)i][(size_t)f.w] ^= b;
                      for (int ch = 0; ch < 2; ++ch)
                        for (int v = 0; v < 16; ++v) {
                          bool par = ch == 0 && cst;  // row constants ride on the first chunk's entries
                          for (int bb = 0; bb < 4; ++bb) {
                            const int i = F + 4 * ch + bb;
                            if (((v >> bb) & 1) && i < P && ((m[(size_t)i >> 6] >> (i & 63)) & 1)) par = !par;
                          }
                          if (par) cc[(size_t)(F + 1 + 16 * ch + v)][(size_t)f.w] ^= b;
                        }
                    };
                    int bit = 0;
                    for (int s = 0; s < f.h2; ++s) place(fg.us[(size_t)s], bit++, false);
                    for (int s = 0; s < f.h2; ++s) place(fg.vs[(size_t)s], bit++, false);
                    for (size_t t = 0; t < fg.c0.size(); ++t) place(fg.c0[t], bit++, fg.c0c[t] != 0);
                    for (size_t t = 0; t < fg.c1.size(); ++t) place(fg.c1[t], bit++, fg.c1c[t] != 0);
                    for (size_t t = 0; t < fg.c3.size(); ++t) place(fg.c3[t], bit++, fg.c3c[t] != 0);
                    for (int t = 0; t < fg.nD; ++t) {  // index bits: term 0 holds the most significant pair (as emit_level4)
                      place(fg.dal[(size_t)t], bit + 2 * (fg.nD - 1 - t), false);
                      place(fg.dbe[(size_t)t], bit + 2 * (fg.nD - 1 - t) + 1, false);
                    }
                    bit += 2 * fg.nD;
                    place(fg.lam, bit++, false);
                    place(fg.lin, bit++, false);
                  }
                while (img.size() % 4) img.push_back(0u);
                ccol_off = (uint32_t)img.size();
                for (auto &e : cc) img.insert(img.end(), e.begin(), e.end());
                crec_off = (uint32_t)img.size();
                for (auto &f : fl) img.push_back((uint32_t)f.w | ((uint32_t)f.off << 8) | ((uint32_t)f.h2 << 16) | ((uint32_t)f.nc << 24));
              }
            }
            uint32_t *h = &img[(size_t)wr_this];
            h[WR_NRUNS] = (uint32_t)(runs.size() / 2);
            h[WR_RUNS] = runs_off;
            h[WR_RUNB] = runb_off;
            h[WR_FLIPS] = flips_off;
            h[WR_LUT] = lut_off;
            h[WR_LUTMASK] = lutmask;
            h[WR_WO32] = (uint32_t)wo32;
            h[WR_COLBYTES] = (uint32_t)colbytes;
            h[WR_CCOL] = ccol_off;
            h[WR_CREC] = crec_off;
            p->stats[7] |= 32 | (ccol_off ? 16 : 0);
            {  // the term tables of every level, for a copy in LDS: (image offset, words) per level
              const uint32_t tt_rec = (uint32_t)img.size();
              uint32_t tt_total = 0;
              for (auto &lv : c.levels) {
                img.push_back(lv.tt_off);
                img.push_back((lv.tt_words + 3u) & ~3u);
                tt_total += (lv.tt_words + 3u) & ~3u;
              }
              uint32_t gtot = 0;
              for (auto &lv : c.levels) gtot += (uint32_t)lv.G;
              img[(size_t)wr_this + WR_GTOT] = gtot;
              img[(size_t)wr_this + WR_TT] = tt_rec;
              img[(size_t)wr_this + WR_TTBYTES] = tt_total * 4u;
            }
            bool one = true;
            for (int d = 0; d < wo32; ++d) one = one && runb[(size_t)d + 1] - runb[(size_t)d] <= 1u;
            if (one) {
              const uint32_t r1_off = (uint32_t)img.size();
              for (int d = 0; d < wo32; ++d) {
                const bool has = runb[(size_t)d + 1] > runb[(size_t)d];
                img.push_back(has ? runs[2 * (size_t)runb[(size_t)d]] : 0u);
                img.push_back(has ? runs[2 * (size_t)runb[(size_t)d] + 1] : 0u);
              }
              img[(size_t)wr_this + WR_RUN1] = r1_off;
            }
          }
        }
        if (!wide_rec_ok) p->wr_offs.clear();
        p->wr_off = p->wr_offs.empty() ? 0 : p->wr_offs[0];

      } else if (ok) {
        p->lw = true;
        // gather programs of every component's f_sel (the LDS-staged first pass)
        std::vector<uint32_t> prog;
        for (size_t ci = 0; ci < p->comps.size(); ++ci) {
          prog = emit_gather_program(lw_fsel_progs[ci]);
          img[p->lw_off + ci * LW_WORDS + LW_FSELP] = (uint32_t)img.size();
          img[p->lw_off + ci * LW_WORDS + LW_FSELN] = (uint32_t)(prog.size() / 16);
          img.insert(img.end(), prog.begin(), prog.end());
        }
        // register form of the first pass (k_sample_lw_reg): selection masks over the f row and, per word, the
        // number of selected bits in the lower words.  Needs ascending f_selection (then the position of an f bit
        // inside f_sel is a popcount), f indices below 128 and at most 64 outputs.
        p->lw_reg = p->max_f_index < 128 && p->num_outputs <= 64 && !p->narrow_big && !p->lw_trie;
        for (auto &c : p->comps)
          for (int j = 1; j < c.F; ++j) p->lw_reg = p->lw_reg && c.f_selection[j] > c.f_selection[j - 1];
        if (p->narrow_big) {  // the table builder's binomials C(b, k + 1), k < 8, b < 256 (the wide path's [k][256] layout, saturated:
          // a value beyond 32 bits belongs to no tabulated pattern - the plan keeps a component's patterns below 2^32 - and must
          // only compare greater than every rank)
          while (img.size() % 16) img.push_back(0u);
          p->lw_binom_off = (int)img.size();
          for (int k = 0; k < 8; ++k)
            for (int b = 0; b < 256; ++b) {
              unsigned __int128 c = 1;
              for (int i = 1; i <= k + 1; ++i) c = c * (unsigned __int128)(b - (k + 1) + i > 0 ? b - (k + 1) + i : 0) / (unsigned __int128)i;
              img.push_back(b >= k + 1 ? (c > (unsigned __int128)0xFFFFFFFFu ? 0xFFFFFFFFu : (uint32_t)c) : 0u);
            }
        }
        if (p->lw_reg) {  // binomial table of the rank computation: C(b, k + 1), k < 8, b < 64
          while (img.size() % 16) img.push_back(0u);
          p->lw_binom_off = (int)img.size();
          for (int k = 0; k < 8; ++k)
            for (int b = 0; b < 64; ++b) {
              unsigned long long c = 1;
              for (int i = 1; i <= k + 1; ++i) c = c * (unsigned long long)(b - (k + 1) + i > 0 ? b - (k + 1) + i : 0) / (unsigned long long)i;
              img.push_back(b >= k + 1 ? (uint32_t)c : 0u);
            }
        }
        if (p->lw_reg)
          for (size_t ci = 0; ci < p->comps.size(); ++ci) {
            uint32_t sel[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int v : p->comps[ci].f_selection) sel[v >> 5] |= 1u << (v & 31);
            for (int w = 1; w < 4; ++w) sel[4 + w] = sel[4 + w - 1] + (uint32_t)__builtin_popcount(sel[w - 1]);
            uint32_t *r = &img[p->lw_off + ci * LW_WORDS];
            r[LW_SELMASK] = (uint32_t)(p->lw_off + ci * LW_WORDS + LW_SEL_INLINE);
            memcpy(r + LW_SEL_INLINE, sel, sizeof sel);
          }
        // The fast record (k_sample_lw_fast, tsim_lw_fast.hip.h): ONE component of at most 8 outputs.  Direct outputs
        // as a flat list of runs (rotate right, destination masks), the rank table RANK[ordinal][f-row bit position] =
        // C(position inside f_sel, ordinal + 1) and the placement table LUT[leaf] = the n_out sampled bits at their
        // final columns (leaf = the bits in sampling order, first output most significant).
        p->lwf_off = 0;
        if (p->lw_reg && p->comps.size() == 1 && p->comps[0].n_out >= 1 && p->comps[0].n_out <= TSIMK_LWF_MAX_NOUT) {
          const HostComponent &c = p->comps[0];
          std::vector<uint32_t> runs;
          uint32_t flip[2] = {0u, 0u};
          for (int s = 0; s < 4; ++s)
            for (int d = 0; d < 2; ++d) {
              std::vector<std::array<int, 2>> m;  // (src bit, dst bit) inside the words
              for (int j = 0; j < p->n_direct; ++j) {
                const int src = p->direct_f[j], dst = p->output_order[j];
                if ((src >> 5) != s || (dst >> 5) != d) continue;
                m.push_back({src & 31, dst & 31});
                if (p->direct_flips[j]) flip[d] |= 1u << (dst & 31);
              }
              std::sort(m.begin(), m.end());
              for (size_t i = 0; i < m.size();) {
                size_t j = i + 1;
                while (j < m.size() && m[j][0] == m[j - 1][0] + 1 && m[j][1] == m[j - 1][1] + 1) ++j;
                const int len = (int)(j - i);
                const uint32_t field = (len >= 32 ? 0xFFFFFFFFu : ((1u << len) - 1u)) << m[i][1];
                runs.push_back((uint32_t)((m[i][0] - m[i][1]) & 31) | ((uint32_t)s << 8));
                runs.push_back(d == 0 ? field : 0u);
                runs.push_back(d == 1 ? field : 0u);
                runs.push_back(0u);
                i = j;
              }
            }
          if (runs.size() / 4 <= TSIMK_LWF_MAX_RUNS) {
            while (img.size() % 16) img.push_back(0u);
            p->lwf_off = (int)img.size();
            img.resize(img.size() + LWF_WORDS, 0u);
            const uint32_t runs_off = (uint32_t)img.size();
            img.insert(img.end(), runs.begin(), runs.end());
            while (img.size() % 16) img.push_back(0u);
            const uint32_t rank_off = (uint32_t)img.size();
            std::vector<int> pos_in(128, -1);
            for (int j = 0; j < c.F; ++j) pos_in[c.f_selection[j]] = j;
            for (int k = 0; k < 8; ++k)
              for (int q = 0; q < 128; ++q) {
                const int b = pos_in[q];
                unsigned long long v = 0;
                if (b >= k + 1) {
                  v = 1;
                  for (int i = 1; i <= k + 1; ++i) v = v * (unsigned long long)(b - (k + 1) + i) / (unsigned long long)i;
                }
                img.push_back((uint32_t)v);
              }
            const uint32_t lut_off = (uint32_t)img.size();
            const uint32_t outpos_off = img[p->lw_off + LW_OUTPOS];
            for (uint32_t leaf = 0; leaf < (1u << c.n_out); ++leaf) {
              uint32_t w[2] = {0u, 0u};
              for (int i = 0; i < c.n_out; ++i)
                if ((leaf >> (c.n_out - 1 - i)) & 1u) {
                  const uint32_t dst = img[outpos_off + i];
                  w[dst >> 5] |= 1u << (dst & 31u);
                }
              img.push_back(w[0]);
              img.push_back(w[1]);
            }
            uint32_t *h = &img[p->lwf_off];
            h[LWF_NRUNS] = (uint32_t)(runs.size() / 4);
            h[LWF_FLIP0] = flip[0];
            h[LWF_FLIP1] = flip[1];
            h[LWF_RUNS] = runs_off;
            h[LWF_RANK] = rank_off;
            h[LWF_LUT] = lut_off;
            h[LWF_NOUT] = (uint32_t)c.n_out;
          }
        }
        // The same for programs of 2..4 components of at most 8 outputs each (k_sample_lw_fastm): one rank table, base
        // table and placement table per component, their LDS offsets fixed here.  Header (16 words): n_runs, flip0,
        // flip1, runs offset, n_comp, LDS words; then 8 words per component: rank table, LUT (image offsets), n_out,
        // (their LDS offsets are a running sum the kernel and the launcher both form).
        p->lwfm_off = 0;
        if (p->lw_reg && p->comps.size() >= 2 && p->comps.size() <= TSIMK_LWFM_MAX_COMP) {
          bool ok = true;
          for (auto &c : p->comps) ok = ok && c.n_out >= 1 && c.n_out <= TSIMK_LWF_MAX_NOUT;
          std::vector<uint32_t> runs;
          uint32_t flip[2] = {0u, 0u};
          for (int s2 = 0; s2 < 4 && ok; ++s2)
            for (int d = 0; d < 2; ++d) {
              std::vector<std::array<int, 2>> m;
              for (int j2 = 0; j2 < p->n_direct; ++j2) {
                const int src = p->direct_f[j2], dst = p->output_order[j2];
                if ((src >> 5) != s2 || (dst >> 5) != d) continue;
                m.push_back({src & 31, dst & 31});
                if (p->direct_flips[j2]) flip[d] |= 1u << (dst & 31);
              }
              std::sort(m.begin(), m.end());
              for (size_t a = 0; a < m.size();) {
                size_t b = a + 1;
                while (b < m.size() && m[b][0] == m[b - 1][0] + 1 && m[b][1] == m[b - 1][1] + 1) ++b;
                const int len = (int)(b - a);
                const uint32_t field = (len >= 32 ? 0xFFFFFFFFu : ((1u << len) - 1u)) << m[a][1];
                runs.push_back((uint32_t)((m[a][0] - m[a][1]) & 31) | ((uint32_t)s2 << 8));
                runs.push_back(d == 0 ? field : 0u);
                runs.push_back(d == 1 ? field : 0u);
                runs.push_back(0u);
                a = b;
              }
            }
          if (ok && runs.size() / 4 <= TSIMK_LWF_MAX_RUNS) {
            while (img.size() % 16) img.push_back(0u);
            p->lwfm_off = (int)img.size();
            img.resize(img.size() + 16 + 8 * TSIMK_LWFM_MAX_COMP, 0u);
            const uint32_t runs_off = (uint32_t)img.size();
            img.insert(img.end(), runs.begin(), runs.end());
            for (size_t ci = 0; ci < p->comps.size(); ++ci) {
              const HostComponent &c = p->comps[ci];
              while (img.size() % 16) img.push_back(0u);
              const uint32_t rank_off = (uint32_t)img.size();
              std::vector<int> pos_in(128, -1);
              for (int j2 = 0; j2 < c.F; ++j2) pos_in[c.f_selection[j2]] = j2;
              for (int k = 0; k < 8; ++k)
                for (int q = 0; q < 128; ++q) {
                  const int b = pos_in[q];
                  unsigned long long v = 0;
                  if (b >= k + 1) {
                    v = 1;
                    for (int t = 1; t <= k + 1; ++t) v = v * (unsigned long long)(b - (k + 1) + t) / (unsigned long long)t;
                  }
                  img.push_back((uint32_t)v);
                }
              const uint32_t lut_off = (uint32_t)img.size();
              const uint32_t outpos_off = img[p->lw_off + ci * LW_WORDS + LW_OUTPOS];
              for (uint32_t leaf = 0; leaf < (1u << c.n_out); ++leaf) {
                uint32_t w[2] = {0u, 0u};
                for (int t = 0; t < c.n_out; ++t)
                  if ((leaf >> (c.n_out - 1 - t)) & 1u) {
                    const uint32_t dst = img[outpos_off + t];
                    w[dst >> 5] |= 1u << (dst & 31u);
                  }
                img.push_back(w[0]);
                img.push_back(w[1]);
              }
              uint32_t *hc = &img[p->lwfm_off + 16 + 8 * ci];
              hc[0] = rank_off;
              hc[1] = lut_off;
              hc[2] = (uint32_t)c.n_out;
            }
            uint32_t *h = &img[p->lwfm_off];
            h[0] = (uint32_t)(runs.size() / 4);
            h[1] = flip[0];
            h[2] = flip[1];
            h[3] = runs_off;
            h[4] = (uint32_t)p->comps.size();
          }
        }
        // The gen record (k_sample_gen, tsim_gen.hip.h): ANY narrow program - f rows of up to 2048 bits, up to 512 outputs,
        // up to TSIMK_GEN_MAX_COMP components, up to TSIMK_GEN_KEYS compiled outputs.  Direct-output runs sorted by destination
        // word in groups of four (one scalar load each), the constant flips, a record per component, per component the
        // (f word, selection mask, selected bits below) records of the words that hold selected bits and the output columns;
        // RANK[ordinal][position inside f_sel] of every component as one block the kernel copies to LDS.
        p->gr_off = 0;
        {
          const int wo32 = 2 * ((p->num_outputs + 63) / 64);
          const int wf32 = (p->max_f_index >> 5) + 1;
          bool okg = p->num_outputs >= 1 && wo32 <= 16 && p->max_f_index < 2048 && (int)p->comps.size() <= TSIMK_GEN_MAX_COMP &&
                     p->total_keys >= 1 && p->total_keys <= TSIMK_GEN_KEYS;
          for (auto &c : p->comps)
            for (int j = 1; j < c.F; ++j) okg = okg && c.f_selection[j] > c.f_selection[j - 1];
          std::vector<uint32_t> runs, runb(1, 0u), flips((size_t)wo32, 0u);
          for (int d = 0; d < wo32 && okg; ++d) {
            for (int sw = 0; sw < wf32; ++sw) {
              std::vector<std::array<int, 2>> m;  // (src bit, dst bit) inside the words
              for (int j = 0; j < p->n_direct; ++j) {
                const int src = p->direct_f[j], dst = p->output_order[j];
                if ((src >> 5) != sw || (dst >> 5) != d) continue;
                m.push_back({src & 31, dst & 31});
                if (p->direct_flips[j]) flips[(size_t)d] |= 1u << (dst & 31);
              }
              std::sort(m.begin(), m.end());
              for (size_t i = 0; i < m.size();) {
                size_t j = i + 1;
                while (j < m.size() && m[j][0] == m[j - 1][0] + 1 && m[j][1] == m[j - 1][1] + 1) ++j;
                const int len = (int)(j - i);
                const uint32_t field = (len >= 32 ? 0xFFFFFFFFu : ((1u << len) - 1u)) << m[i][1];
                runs.push_back((uint32_t)sw | ((uint32_t)((m[i][0] - m[i][1]) & 31) << 8));
                runs.push_back(field);
                i = j;
              }
            }
            runb.push_back((uint32_t)(runs.size() / 2));
          }
          okg = okg && runs.size() / 2 <= TSIMK_GEN_MAX_RUNS;
          if (okg) {
            // LDS block: the rank tables of the components, one after the other
            std::vector<uint32_t> blk;
            std::vector<uint32_t> l_rank;
            for (auto &c : p->comps) {
              while (blk.size() % 4) blk.push_back(0u);
              l_rank.push_back((uint32_t)blk.size());
              for (int k = 0; k < TSIMK_LW_MAX_WEIGHT; ++k)
                for (int b = 0; b < c.F; ++b) {
                  unsigned long long v = 0;
                  if (b >= k + 1) {
                    v = 1;
                    for (int i = 1; i <= k + 1; ++i) v = v * (unsigned long long)(b - (k + 1) + i) / (unsigned long long)i;
                  }
                  blk.push_back((uint32_t)v);
                }
            }
            while (blk.size() % 4) blk.push_back(0u);
            if (blk.size() * 4 <= 40 * 1024) {  // (the launcher adds the pattern bases and the waves' row buffers)
              auto align16 = [&]() { while (img.size() % 16) img.push_back(0u); };
              align16();
              p->gr_off = (int)img.size();
              img.resize(img.size() + GR_WORDS, 0u);
              const uint32_t lds_src = (uint32_t)img.size();
              img.insert(img.end(), blk.begin(), blk.end());
              align16();
              // run groups: two runs per 32-byte group (one scalar load), every destination word's runs padded with mask 0
              const uint32_t dst_off = (uint32_t)img.size();
              img.resize(img.size() + 4 * (size_t)wo32, 0u);
              align16();
              const uint32_t groups_off = (uint32_t)img.size();
              uint32_t ngroups = 0;
              for (int d = 0; d < wo32; ++d) {
                const uint32_t r0 = runb[(size_t)d], r1 = runb[(size_t)d + 1];
                const uint32_t ng = (r1 - r0 + 1u) / 2u;
                img[dst_off + 4 * (size_t)d] = ngroups;
                img[dst_off + 4 * (size_t)d + 1] = ng;
                img[dst_off + 4 * (size_t)d + 2] = flips[(size_t)d];
                for (uint32_t r = r0; r < r0 + 2u * ng; ++r) {
                  img.push_back(r < r1 ? (runs[2 * (size_t)r] & 255u) : 0u);
                  img.push_back(r < r1 ? (runs[2 * (size_t)r] >> 8) : 0u);
                  img.push_back(r < r1 ? runs[2 * (size_t)r + 1] : 0u);
                  img.push_back(0u);
                }
                ngroups += ng;
              }
              align16();
              const uint32_t comp_off = (uint32_t)img.size();
              img.resize(img.size() + p->comps.size() * GC_WORDS, 0u);
              for (size_t ci = 0; ci < p->comps.size(); ++ci) {
                const HostComponent &c = p->comps[ci];
                std::vector<uint32_t> sel((size_t)wf32, 0u);
                for (int v : c.f_selection) sel[(size_t)(v >> 5)] |= 1u << (v & 31);
                align16();
                const uint32_t words_off = (uint32_t)img.size();
                uint32_t below = 0, nwords = 0;
                for (int w = 0; w < wf32; ++w)
                  if (sel[(size_t)w]) {
                    img.push_back((uint32_t)w);
                    img.push_back(sel[(size_t)w]);
                    img.push_back(below);
                    img.push_back(0u);
                    below += (uint32_t)__builtin_popcount(sel[(size_t)w]);
                    ++nwords;
                  }
                align16();
                const uint32_t outpos_new = (uint32_t)img.size();
                const uint32_t outpos_off = img[p->lw_off + ci * LW_WORDS + LW_OUTPOS];
                for (int i = 0; i < c.n_out; ++i) img.push_back(img[outpos_off + (uint32_t)i]);
                uint32_t *cr = &img[comp_off + ci * GC_WORDS];
                cr[GC_NOUT] = (uint32_t)c.n_out;
                cr[GC_F] = (uint32_t)c.F;
                cr[GC_KEYBASE] = img[p->lw_off + ci * LW_WORDS + LW_KEYBASE];
                cr[GC_NWORDS] = nwords;
                cr[GC_WORDREC] = words_off;
                cr[GC_L_RANK] = l_rank[ci];
                cr[GC_OUTPOS] = outpos_new;
                {
                  bool affine = c.n_out > 0;
                  for (int i = 1; i < c.n_out; ++i) affine = affine && img[outpos_off + (uint32_t)i] == img[outpos_off] + (uint32_t)i;
                  cr[GC_OUTBASE] = affine ? img[outpos_off] : 0xFFFFFFFFu;
                }
              }
              uint32_t *h = &img[p->gr_off];
              h[GR_NCOMP] = (uint32_t)p->comps.size();
              h[GR_WO32] = (uint32_t)wo32;
              h[GR_LDS_WORDS] = (uint32_t)blk.size();
              h[GR_LDS_SRC] = lds_src;
              h[GR_DST] = dst_off;
              h[GR_GROUPS] = groups_off;
              h[GR_COMP] = comp_off;
              h[GR_WF32_MIN] = (uint32_t)wf32;
            }
          }
        }
        if (p->lw_trie && !p->gr_off) {  // (only k_sample_gen walks the prefix trees)
          p->lw = false;
          p->lw_trie = false;
          p->lw_wmax.clear();
        }
      } else {
        p->lw_wmax.clear();
      }
    }
  }
  if (p->lw) {  // shadow copy of the LW records: what a table build in the background unranks from (tsim_tables.hip)
    while (img.size() % 32) img.push_back(0u);
    p->lw_shadow_off = (int)img.size();
    img.resize(img.size() + p->comps.size() * LW_WORDS, 0u);
  }
  img.resize(img.size() + 256, 0u);  // tail padding: wide scalar loads may over-read
  if (img.size() >= (1ull << 30)) return tsim_fail(TSIM_ENOTSUP, "program image too large");  // (4 GB: the term-table gathers use 32-bit byte offsets)

  if (p->knobs.defer_group <= 0)  // (not told by TSIM_AMD_TUNE) launches per deferred batch: a batch lasts about as long as ONE hard-row
    p->knobs.defer_group = p->stats[5] > (4ll << 20) ? TSIMK_H_MAX_CTX : 4;  // pass; 4 for small programs, 8 beyond 4 MB of chunk tables (C4)

  fin_mark("pattern-table plan, first-pass records");
  // ---- upload ----
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return tsim_fail(TSIM_EINVAL, "device %d out of range (%d visible)", device, ndev);
  p->device = device;
  HIP_TRY(hipSetDevice(device));
  {
    int cu = 0;
    if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cu > 0) p->n_cu = cu;
  }
  if (int r = tsim_stream_acquire(device, p->stream_idx, &p->stream)) return r;
  HIP_TRY(hipMalloc((void **)&p->d_img, img.size() * 4));
  HIP_TRY(hipMemcpy(p->d_img, img.data(), img.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc((void **)&p->d_dev, std::max<size_t>(1, p->comps.size()) * 4));
  HIP_TRY(hipMemset(p->d_dev, 0, std::max<size_t>(1, p->comps.size()) * 4));
  fin_mark("stream, image upload");
  if (p->lw) {
    if (int r = tsim_tables_build(p, nullptr)) return r;
    if (int r = alloc_feedback(p)) return r;
  }
  fin_mark("pattern tables built");
  if (p->lw && p->lw_cap_now < p->lw_cap_default) {
    // the default depth, in the background - behind the handle's lanes in stream order of creation (tsim_tables.hip: ext_alloc_thread;
    // pooled streams: only the first handles of a process pay for hipStreamCreate)
    for (int k = 1; k <= 4; ++k)
      if (int r = slot_prepare(p, k, 0)) return r;
    (void)tsim_tables_extend_begin(p, p->lw_cap_default);
  }
  p->finalized = true;
  return TSIM_OK;
}
extern "C" void tsim_program_destroy(tsim_program *p) {
  if (!p) return;
  if (p->device >= 0) (void)hipSetDevice(p->device);
  for (hipStream_t &a : p->aux)
    if (a) { tsim_stream_release(p->device, a); a = nullptr; }
  if (p->finalized && p->device >= 0) {
    (void)hipSetDevice(p->device);
    (void)tsim_flush_hard(p);  // parked hard rows of launches that were never joined: finish them, then drain every lane
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    for (auto &sl : p->slots)
      if (sl.side_ready) (void)hipStreamSynchronize(sl.side);
    for (hipEvent_t e : p->ev_pool) (void)hipEventDestroy(e);
    if (tsim_debug("pipeline"))
      fprintf(stderr, "tsim pipeline: begins %llu deferred %llu flushes %llu queries %llu waits %llu fused groups %llu (specialised %llu, component-parallel hard rows %llu)\n",
              p->stat_begins, p->stat_deferred, p->stat_flushes, p->stat_queries, p->stat_waits, p->stat_fused, p->stat_fast, p->stat_partial);
    if (p->sync_ev) (void)hipEventDestroy(p->sync_ev);
    for (hipEvent_t e : p->lane_ev) if (e) (void)hipEventDestroy(e);
    for (auto &cs : p->caller_streams) if (cs.ev) (void)hipEventDestroy(cs.ev);
    for (hipEvent_t e : p->over_ev) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : p->batch_ev) if (e) (void)hipEventDestroy(e);
    for (void *s : p->scratch)
      if (s) (void)hipFree(s);
    for (void *q : p->owned) (void)hipFree(q);  // buffers the caller never returned
    p->owned.clear();
    p->ext_abort.store(true, std::memory_order_release);
    if (p->ext_thread.joinable()) p->ext_thread.join();
    if (p->ext_stream) tsim_stream_release(p->device, p->ext_stream);
    // a table build in the background: its slices ran on whatever stream the next launch used - the lanes drained above, but
    // tsim_sample_batch_device puts them on CALLER streams too, which the handle cannot drain one by one
    if (p->ext_pending) (void)hipDeviceSynchronize();
    if (p->d_img) (void)hipFree(p->d_img);
    if (p->d_dev) (void)hipFree(p->d_dev);
    for (void *q : p->ext_scratch) (void)hipFree(q);
    if (p->ext_tab) (void)hipFree(p->ext_tab);
    if (p->ext_ev) (void)hipEventDestroy(p->ext_ev);
    if (p->d_lw_tab) (void)hipFree(p->d_lw_tab);
    if (p->ctl_block) (void)hipFree(p->ctl_block);  // (the slots' counter sets: one allocation)
    if (p->h_feedback) (void)hipHostFree((void *)p->h_feedback);
    for (auto &sl : p->slots) {
      if (sl.hard) (void)hipFree(sl.hard);
      if (sl.hard2) (void)hipFree(sl.hard2);
      if (sl.keys) (void)hipFree(sl.keys);
      if (sl.ev1) (void)hipEventDestroy(sl.ev1);
      if (sl.ev2) (void)hipEventDestroy(sl.ev2);

      if (sl.side && !sl.side_borrowed) tsim_stream_release(p->device, sl.side);
    }
    if (p->stream) tsim_stream_release(p->device, p->stream);
  } else if (p->device >= 0) {
    // a finalize that failed after it had taken streams / device memory (the pooled streams stayed marked busy for good)
    if (p->ext_thread.joinable()) { p->ext_abort.store(true, std::memory_order_release); p->ext_thread.join(); }
    if (p->ext_stream) tsim_stream_release(p->device, p->ext_stream);
    for (auto &sl : p->slots)
      if (sl.side && !sl.side_borrowed) tsim_stream_release(p->device, sl.side);
    if (p->stream) tsim_stream_release(p->device, p->stream);
    if (p->d_img) (void)hipFree(p->d_img);
    if (p->d_dev) (void)hipFree(p->d_dev);
    if (p->d_lw_tab) (void)hipFree(p->d_lw_tab);
    if (p->h_feedback) (void)hipHostFree((void *)p->h_feedback);
    (void)hipGetLastError();
  }
  delete p;
}

extern "C" int tsim_program_stats(const tsim_program *p, int64_t out[8]) {
  if (!p || !out) return tsim_fail(TSIM_EINVAL, "NULL argument");
  if (!p->finalized) return tsim_fail(TSIM_ESTATE, "program not finalized");
  for (int i = 0; i < 8; ++i) out[i] = p->stats[i];
  out[0] = p->fast ? 1 : 0;
  return TSIM_OK;
}

extern "C" int tsim_program_path_counts(tsim_program *p, int64_t out[TSIM_PATH_COUNT], int32_t reset) {
  if (!p || !out) return tsim_fail(TSIM_EINVAL, "NULL argument");
  for (int i = 0; i < TSIM_PATH_COUNT; ++i) {
    out[i] = p->path_count[i];
    if (reset) p->path_count[i] = 0;
  }
  return TSIM_OK;
}

extern "C" int tsim_program_get_mode(const tsim_program *p, int32_t *fast) {
  if (!p || !fast) return tsim_fail(TSIM_EINVAL, "NULL argument");
  if (!p->finalized) return tsim_fail(TSIM_ESTATE, "program not finalized");
  *fast = p->fast ? 1 : 0;
  return TSIM_OK;
}

extern "C" int tsim_program_info(const tsim_program *p, int32_t *n_components, int32_t *num_outputs,
                                 int64_t *image_bytes, int64_t *total_graphs, int64_t *total_rows) {
  if (!p) return tsim_fail(TSIM_EINVAL, "program is NULL");
  if (n_components) *n_components = (int32_t)p->comps.size();
  if (num_outputs) *num_outputs = p->num_outputs;
  if (image_bytes) *image_bytes = (int64_t)p->img.size() * 4;
  if (total_graphs) *total_graphs = p->total_graphs;
  if (total_rows) *total_rows = p->total_rows;
  return TSIM_OK;
}
int tsim_need_final(const tsim_program *p) {
  if (!p) return tsim_fail(TSIM_EINVAL, "program is NULL");
  if (!p->finalized) return tsim_fail(TSIM_ESTATE, "program not finalized");
  return 0;
}

int tsim_ensure_scratch(tsim_program *p, int slot, size_t bytes) {
  if (p->scratch_sz[slot] >= bytes) return 0;
  if (p->scratch[slot]) {
    HIP_TRY(hipStreamSynchronize(p->stream));
    HIP_TRY(hipFree(p->scratch[slot]));
    p->scratch[slot] = nullptr;
    p->scratch_sz[slot] = 0;
  }
  size_t cap = std::max<size_t>(bytes, 256);
  hipError_t e = hipMalloc(&p->scratch[slot], cap);
  if (e != hipSuccess) return tsim_fail(TSIM_ENOMEM, "hipMalloc(%zu) failed: %s", cap, hipGetErrorString(e));
  p->scratch_sz[slot] = cap;
  return 0;
}
// ---------------------------------------------------------------------------
// plumbing
// ---------------------------------------------------------------------------
extern "C" int tsim_device_count(int32_t *count) {
  if (!count) return tsim_fail(TSIM_EINVAL, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return tsim_fail(TSIM_EHIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
  }
  *count = n;
  return TSIM_OK;
}

extern "C" int tsim_malloc_device(tsim_program *p, int64_t nbytes, void **d_ptr) {
  if (int r = tsim_need_final(p)) return r;
  if (int r = tsim_set_device(p)) return r;
  if (!d_ptr || nbytes < 0) return tsim_fail(TSIM_EINVAL, "bad argument");
  hipError_t e = hipMalloc(d_ptr, (size_t)std::max<int64_t>(nbytes, 1));
  if (e != hipSuccess) return tsim_fail(TSIM_ENOMEM, "hipMalloc(%lld) failed: %s", (long long)nbytes, hipGetErrorString(e));
  p->owned.insert(*d_ptr);
  return TSIM_OK;
}

extern "C" int tsim_free_device(tsim_program *p, void *d_ptr) {
  if (int r = tsim_need_final(p)) return r;
  if (int r = tsim_set_device(p)) return r;
  if (!d_ptr) return TSIM_OK;
  if (!p->owned.erase(d_ptr)) return tsim_fail(TSIM_EINVAL, "pointer was not allocated by tsim_malloc_device on this handle");
  HIP_TRY(hipFree(d_ptr));
  return TSIM_OK;
}

extern "C" int tsim_mem_info(tsim_program *p, int64_t *free_bytes, int64_t *total_bytes) {
  if (int r = tsim_need_final(p)) return r;
  if (int r = tsim_set_device(p)) return r;
  size_t f = 0, t = 0;
  HIP_TRY(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = (int64_t)f;
  if (total_bytes) *total_bytes = (int64_t)t;
  return TSIM_OK;
}

extern "C" int tsim_malloc_pinned(int64_t nbytes, void **h_ptr) {
  if (!h_ptr || nbytes < 0) return tsim_fail(TSIM_EINVAL, "bad argument");
  hipError_t e = hipHostMalloc(h_ptr, (size_t)std::max<int64_t>(nbytes, 1), hipHostMallocDefault);
  if (e != hipSuccess) return tsim_fail(TSIM_ENOMEM, "hipHostMalloc(%lld) failed: %s", (long long)nbytes, hipGetErrorString(e));
  return TSIM_OK;
}

extern "C" int tsim_free_pinned(void *h_ptr) {
  HIP_TRY(hipHostFree(h_ptr));
  return TSIM_OK;
}

extern "C" int tsim_memcpy_h2d(tsim_program *p, void *d_dst, const void *h_src, int64_t nbytes) {
  if (int r = tsim_need_final(p)) return r;
  if (int r = tsim_set_device(p)) return r;
  if (nbytes < 0) return tsim_fail(TSIM_EINVAL, "negative size");
  if (nbytes == 0) return TSIM_OK;
  HIP_TRY(hipMemcpyAsync(d_dst, h_src, (size_t)nbytes, hipMemcpyHostToDevice, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  return TSIM_OK;
}

extern "C" int tsim_memcpy_d2h(tsim_program *p, void *h_dst, const void *d_src, int64_t nbytes) {
  if (int r = tsim_need_final(p)) return r;
  if (int r = tsim_set_device(p)) return r;
  if (nbytes < 0) return tsim_fail(TSIM_EINVAL, "negative size");
  if (nbytes == 0) return TSIM_OK;
  HIP_TRY(hipMemcpyAsync(h_dst, d_src, (size_t)nbytes, hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  return TSIM_OK;
}

// Asynchronous copies on a caller-chosen stream (e.g. an auxiliary stream of the handle): the end-to-end sampler moves
// finished batches to the host while later ones are still being sampled (utils/cuda_helpers.py:105-141 copies once, at
// the end).  Pageable host memory is allowed (the runtime stages it; the call may then block until the copy is done).
extern "C" int tsim_memcpy_d2h_async(tsim_program *p, void *h_dst, const void *d_src, int64_t nbytes, void *stream) {
  if (int r = tsim_need_final(p)) return r;
  if (int r = tsim_set_device(p)) return r;
  if (nbytes < 0) return tsim_fail(TSIM_EINVAL, "negative size");
  if (nbytes == 0) return TSIM_OK;
  HIP_TRY(hipMemcpyAsync(h_dst, d_src, (size_t)nbytes, hipMemcpyDeviceToHost, stream ? (hipStream_t)stream : p->stream));
  return TSIM_OK;
}

extern "C" int tsim_memcpy_h2d_async(tsim_program *p, void *d_dst, const void *h_src, int64_t nbytes, void *stream) {
  if (int r = tsim_need_final(p)) return r;
  if (int r = tsim_set_device(p)) return r;
  if (nbytes < 0) return tsim_fail(TSIM_EINVAL, "negative size");
  if (nbytes == 0) return TSIM_OK;
  HIP_TRY(hipMemcpyAsync(d_dst, h_src, (size_t)nbytes, hipMemcpyHostToDevice, stream ? (hipStream_t)stream : p->stream));
  return TSIM_OK;
}

extern "C" int tsim_stream_synchronize(tsim_program *p, void *stream) {
  if (int r = tsim_need_final(p)) return r;
  if (int r = tsim_set_device(p)) return r;
  HIP_TRY(hipStreamSynchronize(stream ? (hipStream_t)stream : p->stream));
  return TSIM_OK;
}

// Auxiliary streams owned by the handle (index 0 .. TSIM_AUX_STREAMS-1, created on first use, destroyed with it): for
// work beside the sampling lanes - the device-side channel sampler, host transfers.
extern "C" int tsim_aux_stream(tsim_program *p, int32_t index, void **stream) {
  if (int r = tsim_need_final(p)) return r;
  if (int r = tsim_set_device(p)) return r;
  if (index < 0 || index >= TSIM_AUX_STREAMS || !stream) return tsim_fail(TSIM_EINVAL, "bad auxiliary stream %d", index);
  if (!p->aux[index])
    if (int r = tsim_stream_acquire(p->device, p->stream_idx, &p->aux[index])) return r;
  *stream = (void *)p->aux[index];
  return TSIM_OK;
}

// The pipeline slot the next batch of tsim_sample_steps_device will use (its rotation over all TSIM_PIPELINE_SLOTS):
// what a caller needs to join exactly that batch later (tsim_sample_batch_device_end on the slot).
extern "C" int tsim_pipeline_next_slot(tsim_program *p, int32_t *slot) {
  if (int r = tsim_need_final(p)) return r;
  if (!slot) return tsim_fail(TSIM_EINVAL, "NULL argument");
  *slot = (int32_t)(p->steps_slot % (unsigned long long)TSIM_PIPELINE_SLOTS);
  return TSIM_OK;
}

extern "C" int tsim_get_stream(tsim_program *p, void **stream) {
  if (int r = tsim_need_final(p)) return r;
  if (!stream) return tsim_fail(TSIM_EINVAL, "stream is NULL");
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
  if (!key) return tsim_fail(TSIM_EINVAL, "key is NULL");
  uint32_t o[4];
  tsim_key_split(key[0], key[1], o);  // key, subkey = split(key)  (sampler.py:399)
  key[0] = o[0];
  key[1] = o[1];
  return tsim_sample_batch_device_begin(p, slot, d_f, B, num_f, o[2], o[3], shot_offset, d_out, d_max_norm_dev, stream,
                                        flags);
}

extern "C" int tsim_synchronize(tsim_program *p) {
  if (int r = tsim_need_final(p)) return r;
  if (int r = tsim_set_device(p)) return r;
  if (int r = tsim_flush_hard(p)) return r;
  HIP_TRY(hipStreamSynchronize(p->stream));
  // the four lanes of the deferred plan (two of first passes, two of hard-row batches) and every slot stream a launch actually ran on (a stream that never
  // carried work has nothing to wait for - and each hipStreamSynchronize costs a few microseconds)
  for (int k = 1; k <= TSIM_PIPELINE_SLOTS; ++k) {
    tsim_program::Slot &sl = p->slots[k];
    if (!sl.side_ready) continue;
    if ((k <= 4 || sl.used) && sl.side != p->stream) HIP_TRY(hipStreamSynchronize(sl.side));
    sl.pending = false;
  }
  return TSIM_OK;
}
