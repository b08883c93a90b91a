// tsim_rows.hip - evaluate() seam (compile/evaluate.py:15-59) and the dispatch to the row-formulation
// kernels, which are instantiated in tsim_rows_fast.hip / tsim_rows_faithful.hip.
#include "tsim_internal.hip.h"

using namespace tsimk;
using namespace tsimhost;

int tsim_launch_rows(tsim_program *p, int wmax, const SampleArgs &a, long long grid, int block, size_t lds, hipStream_t s) {
  return p->fast ? tsimrows::sample_fast(wmax, a, grid, block, lds, s) : tsimrows::sample_faithful(wmax, a, grid, block, lds, s);
}

int tsim_launch_lw_build(int W, bool fast, const LwBuildArgs &a, int n_out, hipStream_t s) {
  return fast ? tsimrows::lw_build_fast(W, a, n_out, s) : tsimrows::lw_build_faithful(W, a, n_out, s);
}

extern "C" int tsim_evaluate(tsim_program *p, int32_t component, int32_t level, const uint8_t *params,
                             int64_t B, float *re, float *im, float *abs_out, int32_t *coeffs_power) {
  if (int r = tsim_need_final(p)) return r;
  if (int r = tsim_set_device(p)) return r;
  if (component < 0 || component >= (int)p->comps.size()) return tsim_fail(TSIM_EINVAL, "bad component %d", component);
  const HostComponent &c = p->comps[component];
  if (level < 0 || level >= c.n_levels) return tsim_fail(TSIM_EINVAL, "bad level %d", level);
  if (B < 0) return tsim_fail(TSIM_EINVAL, "negative B");
  if (B == 0) return 0;
  if (!re || !im) return tsim_fail(TSIM_EINVAL, "output is NULL");
  const int P = c.levels[level].P, W = p->comp_w[component];
  if (P > 0 && !params) return tsim_fail(TSIM_EINVAL, "params is NULL");
  // host-side pack to W 32-bit words per row (astype(bool): nonzero == 1)
  std::vector<uint32_t> x((size_t)B * W, 0u);
  for (int64_t r = 0; r < B; ++r) {
    const uint8_t *src = params + (size_t)r * P;
    uint32_t *dst = &x[(size_t)r * W];
    for (int i = 0; i < P; ++i)
      if (src[i]) dst[i >> 5] |= 1u << (i & 31);
  }
  hipStream_t s = p->stream;
  if (int r = tsim_ensure_scratch(p, 0, x.size() * 4)) return r;
  if (int r = tsim_ensure_scratch(p, 1, (size_t)B * 12)) return r;
  if (int r = tsim_ensure_scratch(p, 2, (size_t)B * 20)) return r;
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
  if (int r = p->fast ? tsimrows::eval_fast(W, a, s) : tsimrows::eval_faithful(W, a, s)) return r;
  HIP_TRY(hipMemcpyAsync(re, a.re, (size_t)B * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(im, a.im, (size_t)B * 4, hipMemcpyDeviceToHost, s));
  if (abs_out) HIP_TRY(hipMemcpyAsync(abs_out, a.abs, (size_t)B * 4, hipMemcpyDeviceToHost, s));
  if (coeffs_power) HIP_TRY(hipMemcpyAsync(coeffs_power, a.exact, (size_t)B * 20, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return TSIM_OK;
}
