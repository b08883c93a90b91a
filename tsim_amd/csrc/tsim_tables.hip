// tsim_tables.hip - planning, building and (on demand) deepening the low-weight error-pattern tables of the
// first pass (tsim_lw.hip.h): thresholds of every prefix-tree node for the f_sel patterns of weight <= w.
//
// Depth policy.  Tables to weight 5 are cheap (C2: 243 k patterns, 31 MB, 6 ms) and catch all but 6e-5 of the
// shots at the per-bit fire probabilities the sampler is built for; at denser noise (C2 at p_bit = 0.1: mean
// weight 3.2) weights 6 and 7 matter (91 % -> 99 % of the shots tabulated) but cost 578 MB and ~0.1 s to build.
// So finalize builds weight <= min(cap, 5) and the launch planner calls tsim_tables_extend() when the hard-row
// feedback says that a deeper table would pay - unless the caller pinned the depth
// (tsim_program_set_pattern_tables).  Results never depend on the depth.
#include "tsim_internal.hip.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>

using namespace tsimk;
using namespace tsimhost;

static long long binom(long long n, int k) {
  if (k < 0 || n < k) return 0;
  long long r = 1;
  for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i;
  return r;
}

// Fill p->lw_wmax / lw_npat / lw_bytes and the LW records' depth-dependent words for tables up to weight `cap`
// within `budget` bytes per component (4 x that per program).  Returns false if some component gets no table.
bool tsim_tables_plan(tsim_program *p, int cap, long long budget) {
  std::vector<uint32_t> &img = p->img;
  p->lw_wmax.clear();
  p->lw_npat.clear();
  long long tab_off = 0;
  for (size_t ci = 0; ci < p->comps.size(); ++ci) {
    const HostComponent &c = p->comps[ci];
    long long npat = 0;
    int wmax = -1;
    uint32_t bases[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int w = 0; w <= cap && w <= TSIMK_LW_MAX_WEIGHT; ++w) {
      const long long cnt = binom(c.F, w);
      const long long bytes = ((npat + cnt) << c.n_out) * 4;  // (the component's table starts on a 128-byte line)
      if (bytes > budget || (w > 1 && tab_off * 4 + bytes > 4 * budget)) break;  // per component / all together
      if (((npat + cnt) << c.n_out) + tab_off >= (1ll << 32)) break;              // float offsets are 32-bit
      bases[w] = (uint32_t)npat;
      npat += cnt;
      wmax = w;
    }
    if (wmax < 0) return false;
    uint32_t *r = &img[p->lw_off + ci * LW_WORDS];
    r[LW_WMAX] = (uint32_t)wmax;
    r[LW_TAB] = (uint32_t)tab_off;
    r[LW_NPAT] = (uint32_t)npat;
    memcpy(&img[r[LW_BASES]], bases, sizeof bases);
    p->lw_wmax.push_back(wmax);
    p->lw_npat.push_back(npat);
    tab_off += npat << c.n_out;
    tab_off = (tab_off + 31) & ~31ll;  // 128-byte lines: the first pass reads a node's subtree as 8- and 16-byte words
  }
  p->lw_bytes = tab_off * 4;
  return true;
}

// All f_sel patterns of weight <= wmax over F bits in table order: weight by weight, colex within a weight
// (rank = sum_i C(b_i, i + 1) for set bits b_0 < b_1 < ...: what the first pass computes from a shot's bits).
static void enumerate_patterns(int F, int wmax, std::vector<unsigned long long> &pats) {
  pats.push_back(0ull);
  for (int w = 1; w <= wmax && w <= F; ++w) {
    int c[TSIMK_LW_MAX_WEIGHT + 1];
    for (int i = 0; i < w; ++i) c[i] = i;
    c[w] = F;  // sentinel
    for (;;) {
      unsigned long long m = 0;
      for (int i = 0; i < w; ++i) m |= 1ull << c[i];
      pats.push_back(m);
      int i = 0;
      while (i < w && c[i] + 1 == c[i + 1]) ++i;  // first element that can move up
      if (i == w) break;
      ++c[i];
      for (int j = 0; j < i; ++j) c[j] = j;
    }
  }
}

// Allocate and fill the tables for the current plan (p->lw_wmax); the previous buffer, if any, is returned in
// *old (the caller frees it once nothing in flight reads it).
int tsim_tables_build(tsim_program *p, uint32_t **old) {
  uint32_t *tab = nullptr;
  const auto t_start = std::chrono::steady_clock::now();
  hipError_t me = hipMalloc((void **)&tab, std::max<size_t>(16, (size_t)p->lw_bytes));
  if (me != hipSuccess) return tsim_fail(TSIM_ENOMEM, "hipMalloc(%lld) for the pattern tables failed: %s", p->lw_bytes, hipGetErrorString(me));
  for (size_t ci = 0; ci < p->comps.size(); ++ci) {
    const HostComponent &c = p->comps[ci];
    const long long tab_off = (long long)p->img[p->lw_off + ci * LW_WORDS + LW_TAB];
    // (no pattern list: the build kernels unrank the row index, narrow components on lw_rank_term, wide ones in
    // the binomial table - the host enumeration + copy of 14 million patterns cost as much as the kernels)
    const long long lanes = p->lw_npat[ci] << c.n_out;
    float *p1 = nullptr;
    hipError_t e = hipMalloc((void **)&p1, std::max<size_t>(16, (size_t)lanes * 4));  // node values, freed below
    LwBuildArgs a;
    a.img = p->d_img;
    a.patbits = nullptr;
    a.wide_binom_off = p->lw_wide ? p->lw_binom_off : 0;
    a.bases_off = p->lw_off + (int)ci * LW_WORDS + LW_BASES_INLINE;
    a.wmax = p->lw_wmax[ci];
    a.tab = tab + tab_off;
    a.comp_off = p->comp_off + (int)ci * C_WORDS;
    a.npat = (int)p->lw_npat[ci];
    a.p1 = p1;
    a.depth = -1;
    int r = 0;
    if (e == hipSuccess) r = tsim_launch_lw_build(p->comp_w[ci], p->fast, a, c.n_out, p->stream);
    if (e == hipSuccess && r == 0) e = hipStreamSynchronize(p->stream);
    if (p1) (void)hipFree(p1);
    if (r || e != hipSuccess) {
      (void)hipFree(tab);
      return r ? r : tsim_fail(TSIM_EHIP, "pattern table build failed: %s", hipGetErrorString(e));
    }
  }
  if (old) *old = p->d_lw_tab;
  p->d_lw_tab = tab;
  static const bool timing = tsim_debug("tables");
  if (timing)
    fprintf(stderr, "[tsim] pattern tables: %.1f MB built in %.1f ms\n", (double)p->lw_bytes / 1e6,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
  return 0;
}

// Deepen the tables to the largest weight the budget allows (called by the launch planner between launches,
// with every lane idle: the records in the device image are rewritten).
int tsim_tables_extend(tsim_program *p) {
  if (!p->lw || p->lw_cap_now >= p->lw_cap_max) return 0;
  // The component records in the device image are rewritten below and the old table is freed: nothing may be in
  // flight - not on the handle's lanes (the planner drained them) and not on a stream the CALLER passed to the
  // device entry points either (include/tsim_hip.h lets it).  Once per handle.
  HIP_TRY(hipDeviceSynchronize());
  const std::vector<int> before = p->lw_wmax;
  const std::vector<long long> npat_before = p->lw_npat;
  const long long bytes_before = p->lw_bytes;
  std::vector<uint32_t> saved(p->img.begin() + p->lw_off, p->img.begin() + p->lw_off + p->comps.size() * LW_WORDS);
  p->lw_cap_now = p->lw_cap_max;
  if (!tsim_tables_plan(p, p->lw_cap_max, p->lw_budget) || p->lw_wmax == before) {
    // nothing to gain (budget): restore the plan, never ask again
    std::copy(saved.begin(), saved.end(), p->img.begin() + p->lw_off);
    p->lw_wmax = before;
    p->lw_npat = npat_before;
    p->lw_bytes = bytes_before;
    return 0;
  }
  uint32_t *old = nullptr;
  const size_t rec_bytes = p->comps.size() * LW_WORDS * 4;
  // the build kernels unrank with the NEW bases, read from the device image (every lane is idle)
  HIP_TRY(hipMemcpy(p->d_img + p->lw_off, p->img.data() + p->lw_off, rec_bytes, hipMemcpyHostToDevice));
  if (int r = tsim_tables_build(p, &old)) {  // e.g. out of memory: keep what we have
    (void)r;
    std::copy(saved.begin(), saved.end(), p->img.begin() + p->lw_off);
    (void)hipMemcpy(p->d_img + p->lw_off, p->img.data() + p->lw_off, rec_bytes, hipMemcpyHostToDevice);
    p->lw_wmax = before;
    p->lw_npat = npat_before;
    p->lw_bytes = bytes_before;
    (void)hipGetLastError();
    return 0;
  }
  HIP_TRY(hipMemcpy(p->d_img + p->lw_off, p->img.data() + p->lw_off, p->comps.size() * LW_WORDS * 4, hipMemcpyHostToDevice));
  if (old) HIP_TRY(hipFree(old));
  return 0;
}
