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
#include "tsim_trie.hip.h"

#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <cstdlib>

using namespace tsimk;
using namespace tsimhost;

static long long binom(long long n, int k) {
  if (k < 0 || n < k) return 0;
  long long r = 1;
  for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i;
  return r;
}

// chunks of a pattern's FULL outcome tree (no node pruned), saturated
static long long trie_full_chunks(int n_out) {
  long long tot = 0, lvl = 1;
  for (int k = 0; k < tsimk::trie_levels(n_out); ++k) {
    tot += lvl;
    if (tot > (1ll << 40)) return 1ll << 40;
    lvl *= 1ll << tsimk::trie_outputs(n_out, k);
  }
  return tot;
}
// scratch of the build of component `ci` under plan `t` (bytes): node values (dense), header + one TrieMeta per chunk (trie)
static size_t tables_scratch_bytes(const tsim_program *p, size_t ci, const TsimTablePlan &t) {
  if (p->comps[ci].trie) return (size_t)tsimk::TH_WORDS * 4 + (size_t)t.chunks[ci] * sizeof(tsimk::TrieMeta);
  return std::max<size_t>(16, (size_t)(t.npat[ci] << p->comps[ci].n_out) * 4);
}
// The depth-dependent words of the LW records at image offset `rec_off` (the live records at p->lw_off, or their
// shadow copy at p->lw_shadow_off that a build in the background works from) for tables up to weight `cap` within
// `budget` bytes per component (4 x that per program).  Returns false if some component gets no table.
bool tsim_tables_plan_at(tsim_program *p, int cap, long long budget, int rec_off, TsimTablePlan &out) {
  std::vector<uint32_t> &img = p->img;
  out.wmax.clear();
  out.npat.clear();
  out.chunks.clear();
  long long tab_off = 0;
  for (size_t ci = 0; ci < p->comps.size(); ++ci) {
    const HostComponent &c = p->comps[ci];
    long long npat = 0;
    int wmax = -1;
    uint32_t bases[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // Prefix-tree format: the roots of every tabulated pattern may take a quarter of the component's budget (512 MB unless the
    // caller named one); the rest is handed out while the trees are built, lightest patterns first (tsim_trie.hip.h)
    const long long tbudget = std::min(budget, p->lw_trie_budget);
    for (int w = 0; w <= cap && w <= TSIMK_LW_MAX_WEIGHT; ++w) {
      const long long cnt = binom(c.F, w);
      if (c.trie) {
        if ((npat + cnt) * 32 > tbudget / 4 && w > 0) break;
        if ((npat + cnt) * 32 > tbudget) break;
      } else {
        const long long bytes = ((npat + cnt) << c.n_out) * 4;  // (the component's table starts on a 128-byte line)
        // (10..12 outputs: a pattern is 4-16 KB of thresholds - the prefix-tree budget applies, n11: 1.9 GB at weight 5 -> 324 MB at weight 4
        // for 15 % of its rate; fewer outputs keep the large budget: F59's weight-5 class is 640 MB and worth 4x, profiles/r06/budget_512.txt)
        const long long dbudget = c.n_out >= 10 ? std::min(budget, p->lw_trie_budget) : budget;
        if (bytes > dbudget || (w > 1 && tab_off * 4 + bytes > 4 * budget)) break;  // per component / all together
        if (((npat + cnt) << c.n_out) + tab_off >= (1ll << 32)) break;              // float offsets are 32-bit
        // one wide component (k_sample_wide): byte offsets into the table are 32-bit there - a deeper table of 4 GiB or more would
        // push the program off that kernel for good
        // (per component since round 5: every pass of k_sample_wide gets its component's table as its base)
        if (p->lw_wide && bytes >= (1ll << 32)) break;
      }
      bases[w] = (uint32_t)npat;
      npat += cnt;
      wmax = w;
    }
    if (wmax < 0) return false;
    long long chunks = 0;
    if (c.trie) {
      const long long full = trie_full_chunks(c.n_out);
      chunks = std::max(npat, std::min(tbudget / 32, npat >= (1ll << 40) / full ? (1ll << 40) : npat * full));
      if (chunks * 8 + tab_off >= (1ll << 32)) chunks = ((1ll << 32) - 64 - tab_off) / 8;
      if (chunks < npat) return false;
    }
    uint32_t *r = &img[(size_t)rec_off + ci * LW_WORDS];
    r[LW_WMAX] = (uint32_t)wmax;
    r[LW_TAB] = (uint32_t)tab_off;
    r[LW_NPAT] = (uint32_t)npat;
    r[LW_FMT] = c.trie ? 1u : 0u;
    r[LW_CHUNKS] = (uint32_t)chunks;
    r[LW_NPAT_OK] = (uint32_t)npat;  // (a prefix-tree build lowers it when the budget ends it early: trie_build_sync)
    memcpy(r + LW_BASES_INLINE, bases, sizeof bases);  // (LW_BASES of the live record points at these words)
    out.wmax.push_back(wmax);
    out.npat.push_back(npat);
    out.chunks.push_back(chunks);
    tab_off += c.trie ? chunks * 8 : (npat << c.n_out);
    tab_off = (tab_off + 31) & ~31ll;  // 128-byte lines: the first pass reads a node's subtree as 8- and 16-byte words
  }
  out.bytes = tab_off * 4;
  return true;
}

// the live records (finalize): p->lw_wmax / lw_npat / lw_bytes follow
bool tsim_tables_plan(tsim_program *p, int cap, long long budget) {
  TsimTablePlan t;
  p->lw_wmax.clear();
  p->lw_npat.clear();
  if (!tsim_tables_plan_at(p, cap, budget, p->lw_off, t)) return false;
  p->lw_wmax = t.wmax;
  p->lw_npat = t.npat;
  p->lw_chunks = t.chunks;
  p->lw_bytes = t.bytes;
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

// (the prefix-tree builder's node kernel on the chunk tables too: tsim_build4.hip, k_trie_nodes4)
static bool trie_on_chunk_tables(const tsim_program *p) {
  const char *e = getenv("TSIM_AMD_TABLE_BUILD");
  return p->v4 && p->fast && p->v4_max_nch > 0 && !(e && strcmp(e, "rows") == 0);
}

// One slice of a component's dense tables: on the LDS chunk tables when the program has them (tsim_build4.hip: a 16-byte LDS read
// per 4-bit chunk of x and graph instead of the row kernel's walk over every parameter row), else the row formulation.
// TSIM_AMD_TABLE_BUILD=rows: always the latter (tests compare the two builders' tables word for word).
static int launch_table_build(tsim_program *p, int W, const tsimk::LwBuildArgs &a, int n_out, hipStream_t s) {
  const char *e = getenv("TSIM_AMD_TABLE_BUILD");  // (read per build: the tests switch it inside one process)
  const bool rows_only = e && strcmp(e, "rows") == 0;
  if (p->v4 && p->fast && !a.trie && !rows_only && p->v4_max_nch > 0)
    return tsim_launch_lw_build4(p, (a.comp_off - p->comp_off) / C_WORDS, a, n_out, s);
  return tsim_launch_lw_build(W, p->fast, a, n_out, s);
}

// The prefix trees of one component, slice after slice of patterns in table order (weight by weight), each slice built to
// its full depth and waited for before the next goes out: when the budget runs out inside a slice, the patterns before
// that slice are complete and the first pass is told so (LW_NPAT_OK: a row whose pattern lies beyond is a hard row before
// it reads anything).  Slice sizes follow what the patterns so far have cost: at most a quarter of the chunks left.
static int trie_build_sync(tsim_program *p, const tsimk::LwBuildArgs &a0, int W, int n_out, hipStream_t s, std::atomic<bool> *abort,
                           long long *npat_ok, int *slices) {
  long long next = 0, per = 64;
  *npat_ok = 0;
  uint32_t h[tsimk::TH_WORDS] = {0};
  while (next < (long long)a0.npat) {
    if (abort && abort->load(std::memory_order_acquire)) return -1;
    tsimk::LwBuildArgs a = a0;
    a.pat_begin = (int)next;
    a.pat_count = (int)std::min<long long>(per, (long long)a0.npat - next);
    if (int r = tsim_launch_lw_build(W, p->fast, a, n_out, s)) return r;
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipMemcpy(h, a0.p1, sizeof h, hipMemcpyDeviceToHost));
    if (slices) ++*slices;
    if (h[tsimk::TH_LOST] != 0u) break;  // (some tree of this slice is cut short)
    next += a.pat_count;
    *npat_ok = next;
    const long long used = (long long)h[tsimk::TH_NEXT] - (long long)a0.npat, left = (long long)a0.trie_cap - (long long)h[tsimk::TH_NEXT];
    const long long avg = std::max<long long>(1, used / next);
    per = std::max<long long>(64, std::min<long long>(1ll << 16, left / (4 * avg)));
  }
  if (tsim_debug("tables"))
    fprintf(stderr, "[tsim] prefix-tree tables: %lld of %d patterns complete, %u of %u chunks (%.1f MB), %u children refused\n", *npat_ok, a0.npat,
            std::min(h[tsimk::TH_NEXT], h[tsimk::TH_VALID_END]), a0.trie_cap, (double)std::min(h[tsimk::TH_NEXT], h[tsimk::TH_VALID_END]) * 32e-6, h[tsimk::TH_LOST]);
  return 0;
}

// Allocate and fill the tables of plan `t`, whose records lie at image offset `rec_off` (host image; the device image must
// hold them too), on stream `s`.  wait: return when they are built (scratch freed); otherwise the kernels are queued and
// the scratch buffers handed back in `scratch` (the caller frees them once `s` has passed them).
static int tables_build_at(tsim_program *p, int rec_off, const TsimTablePlan &t, hipStream_t s, bool wait, uint32_t **tab_out,
                           std::vector<void *> &scratch) {
  // ONE allocation for the table and every component's scratch (node values / tree bookkeeping), one wait at the end: a fresh
  // handle of the cultivation shape spent 2.5-3 ms here on three hipMalloc / synchronize / hipFree rounds (round 6, cold start)
  std::vector<size_t> soff(p->comps.size() + 1, 0);
  for (size_t ci = 0; ci < p->comps.size(); ++ci) soff[ci + 1] = soff[ci] + (tables_scratch_bytes(p, ci, t) + 255) / 256 * 256;
  uint32_t *tab = nullptr;
  hipError_t me = hipMalloc((void **)&tab, std::max<size_t>(16, (size_t)t.bytes));
  if (me != hipSuccess) return tsim_fail(TSIM_ENOMEM, "hipMalloc(%lld) for the pattern tables failed: %s", t.bytes, hipGetErrorString(me));
  char *sbase = nullptr;
  me = hipMalloc((void **)&sbase, std::max<size_t>(256, soff.back()));
  if (me != hipSuccess) {
    (void)hipFree(tab);
    return tsim_fail(TSIM_ENOMEM, "hipMalloc(%zu) for the pattern-table scratch failed: %s", soff.back(), hipGetErrorString(me));
  }
  int r = 0;
  hipError_t e = hipSuccess;
  for (size_t ci = 0; ci < p->comps.size() && r == 0 && e == hipSuccess; ++ci) {
    const HostComponent &c = p->comps[ci];
    const long long tab_off = (long long)p->img[(size_t)rec_off + ci * LW_WORDS + LW_TAB];
    // (no pattern list: the build kernels unrank the row index, narrow components on lw_rank_term, wide ones in
    // the binomial table - the host enumeration + copy of 14 million patterns cost as much as the kernels)
    LwBuildArgs a;
    a.img = p->d_img;
    a.patbits = nullptr;
    a.wide_binom_off = (p->lw_wide || p->narrow_big) ? p->lw_binom_off : 0;
    a.binom_stride = p->lw_binom_stride;
    a.bases_off = rec_off + (int)ci * LW_WORDS + LW_BASES_INLINE;
    a.wmax = t.wmax[ci];
    a.tab = tab + tab_off;
    a.comp_off = p->comp_off + (int)ci * C_WORDS;
    a.npat = (int)t.npat[ci];
    a.p1 = reinterpret_cast<float *>(sbase + soff[ci]);
    a.depth = wait ? -2 : -1;  // (-2: all nodes of the trees in one launch - nothing of this handle runs beside a build that is waited for)
    a.pat_begin = 0;
    a.pat_count = 0;
    a.trie = c.trie ? 1 : 0;
    a.trie_level = 0;
    a.trie_cap = (uint32_t)t.chunks[ci];
    a.comp4 = trie_on_chunk_tables(p) ? p->comp4_off + (int)ci * C4_WORDS : 0;
    a.nch = p->v4_max_nch;
    if (c.trie) {
      long long okp = 0;
      r = trie_build_sync(p, a, p->comp_w[ci], c.n_out, s, nullptr, &okp, nullptr);
      const uint32_t v = (uint32_t)okp;
      p->img[(size_t)rec_off + ci * LW_WORDS + LW_NPAT_OK] = v;
      if (r == 0) e = hipMemcpy(p->d_img + rec_off + ci * LW_WORDS + LW_NPAT_OK, &v, 4, hipMemcpyHostToDevice);
    } else {
      r = launch_table_build(p, p->comp_w[ci], a, c.n_out, s);
    }
  }
  if (r == 0 && e == hipSuccess && wait) e = hipStreamSynchronize(s);
  if (r || e != hipSuccess) {
    (void)hipStreamSynchronize(s);  // kernels of earlier components may still write
    (void)hipFree(sbase);
    (void)hipFree(tab);
    return r ? r : tsim_fail(TSIM_EHIP, "pattern table build failed: %s", hipGetErrorString(e));
  }
  if (wait) (void)hipFree(sbase);
  else scratch.push_back(sbase);
  *tab_out = tab;
  return 0;
}

// the tables of the current plan (finalize): built when this returns
int tsim_tables_build(tsim_program *p, uint32_t **old) {
  const auto t_start = std::chrono::steady_clock::now();
  TsimTablePlan t;
  t.wmax = p->lw_wmax;
  t.npat = p->lw_npat;
  t.chunks = p->lw_chunks;
  t.bytes = p->lw_bytes;
  uint32_t *tab = nullptr;
  std::vector<void *> scratch;
  if (int r = tables_build_at(p, p->lw_off, t, p->stream, true, &tab, scratch)) return r;
  if (old) *old = p->d_lw_tab;
  p->d_lw_tab = tab;
  p->lw_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
  p->lw_build_bytes = p->lw_bytes;
  static const bool timing = tsim_debug("tables");
  if (timing)
    fprintf(stderr, "[tsim] pattern tables: %.1f MB built in %.1f ms\n", (double)p->lw_bytes / 1e6,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
  return 0;
}

// Deepen the tables to the largest weight the budget allows, in the background: the launch planner calls _begin when the
// hard-row feedback says deeper tables would pay and polls at every later plan.
//  * The new depth is planned into a SHADOW copy of the component records: the build kernels unrank with the new bases,
//    the sampling kernels in flight keep the live records and the current tables.
//  * The buffers (0.5-2 GB each) are allocated by a helper thread: hipMalloc of that size takes 0.1-30 ms.
//  * The build goes out in slices of 2^21 table entries (0.3-1.5 ms next to a first pass), ONE per launch plan, each enqueued on the
//    stream of the launch that plan is for, in front of it (tsim_tables_slice): build and sampling alternate in GPU time
//    however far the host runs ahead, and no sampling kernel ever faces more than one slice.  (All slices at once on a
//    stream of their own: a first pass waited 8-18 ms for wave slots - its 512-thread blocks with their LDS do not fit
//    into the slots that 256-thread build blocks free one by one, at lower stream priority too; on 64 masked CUs the
//    build slowed the chip-resident first passes tenfold for its whole, longer, duration; one slice per plan on its own
//    stream, the next when the last was done: a host running ahead polls in a burst and the build never advances.)
//  * When the last slice is done the poll swaps: every stream drained (what is in flight reads the old records), the live
//    records rewritten, the table pointer exchanged, the old table freed - a fraction of a millisecond.
static void ext_alloc_thread(tsim_program *p) {
  int state = 1;
  if (hipSetDevice(p->device) != hipSuccess) state = -1;
  // The shallow start's build: let the handle's FIRST sampling call go out first (at most 3 ms).  The allocations below - twice
  // the table, hundreds of MB - hold the runtime's lock for milliseconds, and a first call that met them took 10-20 ms instead
  // of 0.03 to enqueue (C3 fresh handle + 10^6 shots: 1.5 or 19 ms by the luck of the timing).
  if (p->ext_self)
    for (int i = 0; i < 30 && !p->first_call_out.load(std::memory_order_acquire) && !p->ext_abort.load(std::memory_order_acquire); ++i)
      std::this_thread::sleep_for(std::chrono::microseconds(100));
  uint32_t *tab = nullptr;
  {
    // the new table and its scratch (node values: as many floats again) live next to the old table until the swap; a caller
    // that sized its own buffers by tsim_mem_info must not find the memory gone - leave a quarter of what is free, and at
    // least 1 GiB, alone
    size_t fr = 0, tot = 0;
    const size_t need = 2 * (size_t)p->ext_plan.bytes;
    if (state > 0 && hipMemGetInfo(&fr, &tot) == hipSuccess && need + std::max<size_t>(fr / 4, (size_t)1 << 30) > fr) {
      if (tsim_debug("tables")) fprintf(stderr, "[tsim] deeper tables (%zu MB with scratch) not built: %zu MB free\n", need >> 20, fr >> 20);
      state = -1;
    }
  }
  if (state > 0 && hipMalloc((void **)&tab, std::max<size_t>(16, (size_t)p->ext_plan.bytes)) != hipSuccess) state = -1;
  for (size_t ci = 0; state > 0 && ci < p->comps.size(); ++ci) {
    const HostComponent &c = p->comps[ci];
    const long long tab_off = (long long)p->img[(size_t)p->lw_shadow_off + ci * LW_WORDS + LW_TAB];
    float *p1 = nullptr;
    if (hipMalloc((void **)&p1, tables_scratch_bytes(p, ci, p->ext_plan)) != hipSuccess) { state = -1; break; }
    p->ext_scratch.push_back(p1);
    LwBuildArgs a;
    a.img = p->d_img;
    a.patbits = nullptr;
    a.wide_binom_off = (p->lw_wide || p->narrow_big) ? p->lw_binom_off : 0;
    a.binom_stride = p->lw_binom_stride;
    a.bases_off = p->lw_shadow_off + (int)ci * LW_WORDS + LW_BASES_INLINE;
    a.wmax = p->ext_plan.wmax[ci];
    a.tab = tab + tab_off;
    a.comp_off = p->comp_off + (int)ci * C_WORDS;
    a.npat = (int)p->ext_plan.npat[ci];
    a.p1 = p1;
    a.depth = -1;
    a.pat_begin = 0;
    a.pat_count = 0;
    a.trie = c.trie ? 1 : 0;
    a.trie_level = 0;
    a.trie_cap = (uint32_t)p->ext_plan.chunks[ci];
    a.comp4 = trie_on_chunk_tables(p) ? p->comp4_off + (int)ci * C4_WORDS : 0;
    a.nch = p->v4_max_nch;
    p->ext_jobs.push_back(TsimBuildJob{a, p->comp_w[ci], c.n_out, 0});
  }
  if (state < 0) {
    (void)hipGetLastError();
    if (tsim_debug("tables")) fprintf(stderr, "[tsim] deeper tables: allocation failed or skipped - the handle stays at its depth\n");
    for (void *q : p->ext_scratch) (void)hipFree(q);
    p->ext_scratch.clear();
    p->ext_jobs.clear();
    if (tab) (void)hipFree(tab);
    tab = nullptr;
  }
  p->ext_tab = tab;
  if (state > 0 && p->ext_self) {
    // The shallow start's build of the default depth is driven from HERE, slice after slice on a stream of its own, each
    // waited for before the next goes out (first passes get in between; nothing is ever queued behind a whole build).  The
    // plan-driven form - one slice per launch plan - needs a host that plans no faster than the GPU samples: a caller that
    // enqueues a 10^8-shot job in one go (scripts/time_to_n.py: 13 plans within 0.3 ms) had drawn its last plan before the
    // buffers were even allocated, and sampled the whole job with the shallow tables (C4 at weight 2: 1.9e9 shots/s).
    // ... a pooled stream of its own, taken AFTER the handle's lanes: tsim_program_finalize creates the lanes before it starts
    // this thread.  In front of the lanes it moved them on the hardware queues - the cultivation shape at 10^5 shots per
    // step ran at 1.05 instead of 1.73e10 for the handle's whole life (profiles/r05/hw_queues.txt: the round-4 finding, "one
    // more stream in front of the lanes", through the back door).  On the handle's own stream instead: 1.71e10, but every
    // synchronize then waits for the slice in flight (fresh C4 handle + 10^6 shots 15.5 -> 23 ms).
    hipStream_t bs = nullptr;
    std::vector<int> held;
    if (tsim_stream_acquire(p->device, held, &bs) != 0) state = -1;
    p->ext_stream = bs;
    const size_t rec_words = p->comps.size() * LW_WORDS;
    if (state > 0 && hipMemcpyAsync(p->d_img + p->lw_shadow_off, p->img.data() + p->lw_shadow_off, rec_words * 4, hipMemcpyHostToDevice, bs) != hipSuccess) state = -1;
    for (size_t ji = 0; state > 0 && ji < p->ext_jobs.size(); ++ji) {
      TsimBuildJob &j = p->ext_jobs[ji];
      if (j.a.trie) {  // (slices of its own: trie_build_sync)
        long long okp = 0;
        if (trie_build_sync(p, j.a, j.W, j.n_out, bs, &p->ext_abort, &okp, &p->ext_slices) != 0) { state = -1; break; }
        const uint32_t v = (uint32_t)okp;
        p->img[(size_t)p->lw_shadow_off + ji * LW_WORDS + LW_NPAT_OK] = v;
        if (hipMemcpy(p->d_img + p->lw_shadow_off + ji * LW_WORDS + LW_NPAT_OK, &v, 4, hipMemcpyHostToDevice) != hipSuccess) { state = -1; break; }
        j.next_pat = (long long)j.a.npat;
        continue;
      }
      const long long per = std::max<long long>(1, j.a.trie ? p->ext_entries >> 9 : p->ext_entries >> j.n_out);
      while (state > 0 && j.next_pat < (long long)j.a.npat) {
        if (p->ext_abort.load(std::memory_order_acquire)) { state = -1; break; }
        tsimk::LwBuildArgs a = j.a;
        a.pat_begin = (int)j.next_pat;
        a.pat_count = (int)std::min<long long>(per, (long long)a.npat - j.next_pat);
        if (launch_table_build(p, j.W, a, j.n_out, bs) != 0 || hipStreamSynchronize(bs) != hipSuccess) { state = -1; break; }
        j.next_pat += a.pat_count;
        ++p->ext_slices;
      }
    }
    if (state > 0) {
      // what a table build costs on THIS program, for tsim_tables_deep_after's estimate: the finalize build is half a millisecond
      // of launch latency (it underestimated the build rate severalfold) - this one is tens of milliseconds of kernels
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - p->ext_t0).count();
      if (ms > 1.0 && p->ext_plan.bytes > p->lw_build_bytes) {
        p->lw_build_ms = ms;
        p->lw_build_bytes = p->ext_plan.bytes;
      }
      state = 2;  // built
    }
    else {
      (void)hipGetLastError();
      if (bs) (void)hipStreamSynchronize(bs);
      for (void *q : p->ext_scratch) (void)hipFree(q);
      p->ext_scratch.clear();
      p->ext_jobs.clear();
      if (tab) (void)hipFree(tab);
      p->ext_tab = nullptr;
    }
  }
  p->ext_alloc.store(state, std::memory_order_release);
}

int tsim_tables_extend_begin(tsim_program *p, int target_cap) {
  target_cap = std::min(target_cap, p->lw_cap_max);
  if (!p->lw || p->lw_cap_now >= target_cap || p->ext_pending) return 0;
  // (prefix-tree tables are built by the helper thread alone, slice by slice with the budget in view: the shallow start's
  // default depth - nothing deeper, and never by the launch plans' slices)
  if (p->lw_trie && !(p->lw_cap_now < p->lw_cap_default && target_cap <= p->lw_cap_default)) return 0;
  // (the shallow start's build of the default depth goes out in slices four times as large: until it is in place most rows of
  // an expensive program are hard rows - C4 at weight 2: 1.8e9 instead of 3e10 shots/s - so the build IS the work that matters)
  const bool stage_a = p->lw_cap_now < p->lw_cap_default && target_cap <= p->lw_cap_default;
  p->lw_cap_now = target_cap;  // asked once
  const size_t rec_words = p->comps.size() * LW_WORDS;
  std::copy(p->img.begin() + p->lw_off, p->img.begin() + p->lw_off + (long)rec_words, p->img.begin() + p->lw_shadow_off);
  TsimTablePlan t;
  if (!tsim_tables_plan_at(p, target_cap, p->lw_budget, p->lw_shadow_off, t) || t.wmax == p->lw_wmax) return 0;  // nothing to gain (budget)
  if (!p->ext_ev) HIP_TRY(hipEventCreateWithFlags(&p->ext_ev, hipEventDisableTiming));
  p->ext_t0 = std::chrono::steady_clock::now();
  p->ext_jobs.clear();
  p->ext_scratch.clear();
  p->ext_job = 0;
  p->ext_slices = 0;
  p->ext_entries = stage_a ? (1ll << 23) : (1ll << 21);
  p->ext_plan = t;
  p->ext_uploaded = false;
  p->ext_slice_due = false;
  p->ext_recorded = false;
  p->ext_alloc.store(0, std::memory_order_release);
  if (p->ext_thread.joinable()) p->ext_thread.join();
  p->ext_self = stage_a;  // (the helper thread drives the shallow start's build itself; later depths go out with the launch plans)
  p->ext_abort.store(false, std::memory_order_release);
  p->ext_thread = std::thread(ext_alloc_thread, p);
  p->ext_pending = true;
  return 0;
}

// the next slice of the build on stream `s`, in front of the launch the caller is about to enqueue there
int tsim_tables_slice(tsim_program *p, hipStream_t s) {
  if (!p->ext_pending || p->ext_self || !p->ext_slice_due || p->ext_job >= p->ext_jobs.size()) return 0;
  p->ext_slice_due = false;
  TsimBuildJob &j = p->ext_jobs[p->ext_job];
  tsimk::LwBuildArgs a = j.a;
  // Slice size: 2^21 table entries whatever the program.  A slice's kernels (one per tree depth, then the thresholds) start in
  // the gaps the other lane's first pass leaves, so a slice lasts about one first pass however small it is: smaller slices
  // for expensive programs (cultivation shape, 2^18: 1238 slices, 1.3 s instead of 143 slices, 0.27 s) and slices sized by
  // their measured time (the events span the other lane's pass: the size collapsed) were both worse.
  const long long per = std::max<long long>(1, j.a.trie ? p->ext_entries >> 9 : p->ext_entries >> j.n_out);
  a.pat_begin = (int)j.next_pat;
  a.pat_count = (int)std::min<long long>(per, (long long)a.npat - j.next_pat);
  if (int r = launch_table_build(p, j.W, a, j.n_out, s)) return r;
  ++p->ext_slices;
  j.next_pat += a.pat_count;
  if (j.next_pat >= (long long)a.npat) ++p->ext_job;
  if (p->ext_job >= p->ext_jobs.size()) {  // that was the last one
    HIP_TRY(hipEventRecord(p->ext_ev, s));
    p->ext_recorded = true;
  }
  return 0;
}

// 1: the deeper tables are in place (the caller forgets the feedback of the old ones); 0: nothing changed
int tsim_tables_extend_poll(tsim_program *p, bool wait) {
  if (!p->ext_pending) return 0;
  int st = p->ext_alloc.load(std::memory_order_acquire);
  if (st == 0 && wait) {
    p->ext_thread.join();
    st = p->ext_alloc.load(std::memory_order_acquire);
  }
  if (st == 0) return 0;  // the buffers are still being allocated
  if (p->ext_thread.joinable()) p->ext_thread.join();
  if (st < 0) {  // e.g. out of memory: keep what we have
    p->ext_pending = false;
    return 0;
  }
  const size_t rec_words = p->comps.size() * LW_WORDS;
  const bool self_built = p->ext_self;
  if (self_built && st != 2) return 0;  // (cannot happen: the thread leaves 2 or a negative state behind)
  if (self_built) {
    p->ext_uploaded = true;
    p->ext_job = p->ext_jobs.size();
    p->ext_recorded = true;
  }
  if (!p->ext_uploaded) {  // (128 bytes per component; nothing reads the shadow records yet)
    HIP_TRY(hipMemcpy(p->d_img + p->lw_shadow_off, p->img.data() + p->lw_shadow_off, rec_words * 4, hipMemcpyHostToDevice));
    p->ext_uploaded = true;
    static const bool timing = tsim_debug("tables");
    if (timing)
      fprintf(stderr, "[tsim] pattern tables: buffers for %.1f MB allocated %.2f ms after the plan asked\n", (double)p->ext_plan.bytes / 1e6,
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - p->ext_t0).count());
  }
  if (p->ext_job < p->ext_jobs.size()) {
    if (!wait) {
      p->ext_slice_due = true;  // the launch this plan is for takes one slice along (tsim_tables_slice)
      return 0;
    }
    while (p->ext_job < p->ext_jobs.size()) {
      p->ext_slice_due = true;
      if (int r = tsim_tables_slice(p, p->stream)) return r;
    }
  }
  if (!p->ext_recorded) return 0;
  if (self_built) {
    // (the thread waited for its last slice)
  } else if (wait) {
    HIP_TRY(hipEventSynchronize(p->ext_ev));
  } else if (hipEventQuery(p->ext_ev) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  p->ext_pending = false;
  // The component records in the device image are rewritten below and the old table is freed: nothing of THIS handle may
  // be in flight - not on its lanes and not on a stream the CALLER passed to the device entry points either
  // (include/tsim_hip.h lets it; launch_sample notes them).  Round 4 drained the whole device here - other handles' and the
  // host application's streams included; now only what can hold a reader of the old records.
  if (p->caller_streams_overflow) {
    HIP_TRY(hipDeviceSynchronize());
  } else {
    if (p->stream) HIP_TRY(hipStreamSynchronize(p->stream));
    for (int k = 1; k <= TSIM_PIPELINE_SLOTS; ++k) {
      tsim_program::Slot &sl = p->slots[k];
      if (sl.side_ready && sl.side && sl.side != p->stream) HIP_TRY(hipStreamSynchronize(sl.side));
    }
    bool gone = false;  // (the handle's own event behind the last launch on each caller stream - never the stream itself)
    for (auto &cs : p->caller_streams)
      if (hipEventSynchronize(cs.ev) != hipSuccess) { (void)hipGetLastError(); gone = true; }
    if (p->ext_stream) HIP_TRY(hipStreamSynchronize(p->ext_stream));
    if (gone) HIP_TRY(hipDeviceSynchronize());
  }
  for (auto &cs : p->caller_streams) (void)hipEventDestroy(cs.ev);
  p->caller_streams.clear();  // (noted again by the launches that use them)
  p->caller_streams_overflow = false;
  for (size_t ci = 0; ci < p->comps.size(); ++ci) {
    const uint32_t *sh = &p->img[(size_t)p->lw_shadow_off + ci * LW_WORDS];
    uint32_t *r = &p->img[(size_t)p->lw_off + ci * LW_WORDS];
    r[LW_WMAX] = sh[LW_WMAX];
    r[LW_TAB] = sh[LW_TAB];
    r[LW_NPAT] = sh[LW_NPAT];
    r[LW_CHUNKS] = sh[LW_CHUNKS];
    r[LW_NPAT_OK] = sh[LW_NPAT_OK];
    memcpy(r + LW_BASES_INLINE, sh + LW_BASES_INLINE, 8 * sizeof(uint32_t));
  }
  HIP_TRY(hipMemcpy(p->d_img + p->lw_off, p->img.data() + p->lw_off, rec_words * 4, hipMemcpyHostToDevice));
  uint32_t *old = p->d_lw_tab;
  p->d_lw_tab = p->ext_tab;
  p->ext_tab = nullptr;
  p->lw_wmax = p->ext_plan.wmax;
  p->lw_npat = p->ext_plan.npat;
  p->lw_chunks = p->ext_plan.chunks;
  p->lw_bytes = p->ext_plan.bytes;
  for (void *q : p->ext_scratch) (void)hipFree(q);
  p->ext_scratch.clear();
  if (old) HIP_TRY(hipFree(old));
  static const bool timing = tsim_debug("tables");
  if (timing)
    fprintf(stderr, "[tsim] pattern tables: %.1f MB in place %.1f ms after the plan asked (%d slices)\n", (double)p->lw_bytes / 1e6,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - p->ext_t0).count(), p->ext_slices);
  return 1;
}

// Rows a handle samples with too many hard / missed rows before it builds the next depth: about twice the time the build
// will take (table entries of the next depth at the rate the finalize build ran at; 4e10 rows/s assumed) - a job that
// ends early then loses at most a third of its time to a build that never paid.  Measured break-evens (build time /
// gain per row, profiles/r04/long_runs.txt): C5 4e9, C4 2e10, C3 3e10 rows; this gives 1.0e10, 1.8e10, 0.6e10.
unsigned long long tsim_tables_deep_after(tsim_program *p) {
  if (p->knobs.deep_after) return p->knobs.deep_after;
  if (p->ext_pending) return ~0ull >> 1;  // (a build is under way - the shallow start's: its shadow records must not be replanned)
  if (p->deep_after_auto) return p->deep_after_auto;
  double rows = 4e9;
  if (p->lw && p->lw_build_ms > 0.0 && p->lw_build_bytes > 0) {
    const size_t rec_words = p->comps.size() * LW_WORDS;
    std::copy(p->img.begin() + p->lw_off, p->img.begin() + p->lw_off + (long)rec_words, p->img.begin() + p->lw_shadow_off);
    TsimTablePlan t;
    if (tsim_tables_plan_at(p, p->lw_cap_max, p->lw_budget, p->lw_shadow_off, t)) {
      const double entries_per_ms = (double)p->lw_build_bytes / 4.0 / p->lw_build_ms;  // (what finalize built - not what is in place now)
      const double build_s = (double)t.bytes / 4.0 / entries_per_ms * 1e-3;
      rows = 2.0 * build_s * 4e10;
    }
  }
  p->deep_after_auto = (unsigned long long)std::min(1e11, std::max(1e9, rows));
  static const bool timing = tsim_debug("tables");
  if (timing) fprintf(stderr, "[tsim] pattern tables: the next depth after %.2e rows that want it\n", (double)p->deep_after_auto);
  return p->deep_after_auto;
}

// (blocking form: TSIM_AMD_DEEP_TABLES=1 - the deeper tables NOW)
int tsim_tables_extend(tsim_program *p) {
  if (p->ext_pending) {  // (the shallow start's build of the default depth: finish it first)
    const int r0 = tsim_tables_extend_poll(p, true);
    if (r0 < 0) return r0;
  }
  if (int r = tsim_tables_extend_begin(p, p->lw_cap_max)) return r;
  const int r = tsim_tables_extend_poll(p, true);
  return r < 0 ? r : 0;
}
