// tsim_rows_impl.hip.h - instantiations of the row-formulation kernels for ONE evaluation formulation
// (TSIM_ROWS_FAST = true: pack-time algebra / false: operation-by-operation mirror of the reference).
// Included by tsim_rows_fast.hip and tsim_rows_faithful.hip so that the two sets compile in parallel.
#include "tsim_internal.hip.h"
#include "tsim_trie.hip.h"

#ifndef TSIM_ROWS_FAST
#error "define TSIM_ROWS_FAST (true/false) and TSIM_ROWS_NAME(sym) before including this file"
#endif

using namespace tsimk;

// every W the packer can choose (tsimhost::kWVariants)
#define TSIM_FOR_EACH_W(X) X(1) X(2) X(3) X(4) X(6) X(8) X(12) X(16) X(24) X(32) X(48) X(64)

namespace tsimrows {

int TSIM_ROWS_NAME(sample)(int wmax, const SampleArgs &a, long long grid, int block, size_t lds, hipStream_t s) {
  switch (wmax) {
#define TSIM_X(WV)                                                                                              \
  case WV:                                                                                                      \
    hipLaunchKernelGGL((k_sample<WV, TSIM_ROWS_FAST>), dim3((unsigned)grid), dim3(block), lds, s, a);           \
    break;
    TSIM_FOR_EACH_W(TSIM_X)
#undef TSIM_X
    default: return tsim_fail(TSIM_ENOTSUP, "unsupported word count %d", wmax);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

int TSIM_ROWS_NAME(eval)(int W, const EvalArgs &a, hipStream_t s) {
  const dim3 grid((unsigned)((a.B + 255) / 256));
  switch (W) {
#define TSIM_X(WV)                                                                                 \
  case WV:                                                                                         \
    hipLaunchKernelGGL((k_evaluate<WV, TSIM_ROWS_FAST>), grid, dim3(256), 0, s, a);                \
    break;
    TSIM_FOR_EACH_W(TSIM_X)
#undef TSIM_X
    default: return tsim_fail(TSIM_ENOTSUP, "unsupported word count %d", W);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// pattern-table build (tsim_lw.hip.h): components with <= 64 parameters use W = 1, 2; wide components
// (multi-word patterns) the W of their rows
// ... as a chunked prefix tree (tsim_trie.hip.h): breadth first, one chunk level (three outputs) after the other; per level
// one launch per local depth (every lane of a launch evaluates the same level), then the thresholds and the children.  The
// ranges live on the device (the scratch header): nothing here waits for the GPU.
static int TSIM_ROWS_NAME(trie_build)(int W, const LwBuildArgs &a0, int n_out, hipStream_t s) {
  const int nlev = trie_levels(n_out);
  const long long roots = a0.pat_count ? a0.pat_count : a0.npat - a0.pat_begin;
  hipLaunchKernelGGL((k_trie_begin<TSIM_ROWS_FAST>), dim3(1), dim3(1), 0, s, a0);
  for (int L = 0; L < nlev; ++L) {
    LwBuildArgs a = a0;
    a.trie_level = L;
    const int rem = trie_outputs(n_out, L);
    // level 0 has exactly `roots` chunks; deeper levels hold what the device allocated: a chip-full of blocks strides over it
    for (int d = (L == 0 ? -1 : 0); d < rem; ++d) {
      a.depth = d;
      const long long lanes = roots << (d < 0 ? 0 : d);
      const unsigned grid = L == 0 ? (unsigned)std::min<long long>((lanes + 255) / 256, 8192) : 2048u;
      if (a.comp4 != 0 && TSIM_ROWS_FAST) {  // (the program has chunk tables: the same nodes through eval_level4)
        if (int r = tsim_launch_trie_nodes4(a, grid, s)) return r;
        continue;
      }
      switch (W) {
#define TSIM_X(WV)                                                                                 \
  case WV:                                                                                         \
    hipLaunchKernelGGL((k_trie_nodes<WV, TSIM_ROWS_FAST>), dim3(grid), dim3(256), 0, s, a);        \
    break;
        TSIM_FOR_EACH_W(TSIM_X)
#undef TSIM_X
        default: return tsim_fail(TSIM_ENOTSUP, "pattern tables: unsupported word count %d", W);
      }
      HIP_TRY(hipGetLastError());
    }
    const unsigned gridf = L == 0 ? (unsigned)std::min<long long>((roots + 255) / 256, 8192) : 2048u;
    hipLaunchKernelGGL((k_trie_finish<TSIM_ROWS_FAST>), dim3(gridf), dim3(256), 0, s, a);
    if (L + 1 < nlev) hipLaunchKernelGGL((k_trie_advance<TSIM_ROWS_FAST>), dim3(1), dim3(1), 0, s, a);
    HIP_TRY(hipGetLastError());
  }
  return 0;
}

int TSIM_ROWS_NAME(lw_build)(int W, const LwBuildArgs &a0, int n_out, hipStream_t s) {
  if (a0.trie) return TSIM_ROWS_NAME(trie_build)(W, a0, n_out, s);
  // a0.depth == -2 (the build of tsim_program_finalize: the handle samples nothing yet): every node in one launch, then the
  // thresholds.  Otherwise (a build beside running launches: the background depths) one launch per depth - a single grid of the
  // whole slice, deepest nodes first, kept the first passes of a 10^8-shot job off the chip: C2 fresh handle + 10^8 shots 5 -> 14-17 ms
  if (a0.depth != -2) {
    for (int d = -1; d < n_out; ++d) {
      LwBuildArgs a = a0;
      a.depth = d;
      const long long lanes = (long long)(a.pat_count ? a.pat_count : a.npat - a.pat_begin) << (d < 0 ? 0 : d);
      const dim3 grid((unsigned)((lanes + 255) / 256));
      switch (W) {
#define TSIM_X(WV)                                                                                 \
  case WV:                                                                                         \
    hipLaunchKernelGGL((k_lw_nodes<WV, TSIM_ROWS_FAST>), grid, dim3(256), 0, s, a);                \
    break;
        TSIM_FOR_EACH_W(TSIM_X)
#undef TSIM_X
        default: return tsim_fail(TSIM_ENOTSUP, "pattern tables: unsupported word count %d", W);
      }
      HIP_TRY(hipGetLastError());
    }
  } else {
    LwBuildArgs a = a0;
    a.depth = -2;
    const long long lanes = (long long)(a.pat_count ? a.pat_count : a.npat - a.pat_begin) << n_out;
    if (lanes > 0x7FFFFFFFll * 256) return tsim_fail(TSIM_ENOTSUP, "pattern tables: slice too large");
    const dim3 grid((unsigned)((lanes + 255) / 256));
    switch (W) {
#define TSIM_X(WV)                                                                                 \
  case WV:                                                                                         \
    hipLaunchKernelGGL((k_lw_nodes<WV, TSIM_ROWS_FAST>), grid, dim3(256), 0, s, a);                \
    break;
      TSIM_FOR_EACH_W(TSIM_X)
#undef TSIM_X
      default: return tsim_fail(TSIM_ENOTSUP, "pattern tables: unsupported word count %d", W);
    }
    HIP_TRY(hipGetLastError());
  }
  const long long lanes = (long long)(a0.pat_count ? a0.pat_count : a0.npat - a0.pat_begin) << n_out;
  hipLaunchKernelGGL((k_lw_finish<TSIM_ROWS_FAST>), dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, s, a0);
  HIP_TRY(hipGetLastError());
  return 0;
}

}  // namespace tsimrows
