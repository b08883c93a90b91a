// tsim_noise.hip - device-side channel sampler (statistical twin of ChannelSampler.sample).
#include "tsim_internal.hip.h"
#include "tsim_noise.hip.h"
#include "tsim_sample_internal.hip.h"

using namespace tsimk;

// ---------------------------------------------------------------------------
// device-side noise sampler (statistical twin of ChannelSampler.sample)
// ---------------------------------------------------------------------------
struct tsim_noise {
  tsim_program *prog = nullptr;
  int device = -1;  // copy of prog->device: destroy must not touch the program handle
  int num_f = 0, n_ch = 0, WF = 0, seg = 4096;
  double *d_l1p = nullptr;
  float *d_inv = nullptr;  // 1 / log2(1 - p) per channel (k_noise_tile)
  int tile = 0, tseg = 0;  // k_noise_tile geometry; 0 = the tile form does not apply (rows too wide for LDS)
  uint32_t *d_off = nullptr;
  uint32_t *d_cdf = nullptr;
  uint64_t *d_pat = nullptr;
  // k_noise_wave (round 6): groups of g lanes per (channel, tile); the outcomes' non-zero pattern words as records
  int wave_g = 0;              // 0: that kernel does not apply
  int wave_tile = 0;
  uint32_t *d_pw_off = nullptr, *d_pw = nullptr;
};

extern "C" int tsim_noise_create(tsim_program *p, int32_t num_f, int32_t n_channels, const double *p_fire,
                                 const int32_t *n_outcomes, const double *cond_cdf, const uint8_t *xor_patterns,
                                 tsim_noise **out) {
  if (int r = tsim_need_final(p)) return r;
  if (int r = tsim_set_device(p)) return r;
  if (!out || num_f < 0 || n_channels < 0) return tsim_fail(TSIM_EINVAL, "bad argument");
  if (n_channels > 0 && (!p_fire || !n_outcomes || !cond_cdf || (num_f > 0 && !xor_patterns)))
    return tsim_fail(TSIM_EINVAL, "NULL channel table");
  tsim_noise *n = new (std::nothrow) tsim_noise();
  if (!n) return tsim_fail(TSIM_ENOMEM, "out of host memory");
  n->prog = p;
  n->device = p->device;
  n->num_f = num_f;
  n->n_ch = n_channels;
  n->WF = std::max(1, (num_f + 63) / 64);
  std::vector<double> l1p((size_t)std::max(1, n_channels));
  std::vector<uint32_t> off((size_t)n_channels + 1, 0u);
  double pmax = 1e-9;
  for (int c = 0; c < n_channels; ++c) {
    if (!(p_fire[c] > 0.0) || p_fire[c] > 1.0 || n_outcomes[c] < 1) {
      delete n;
      return tsim_fail(TSIM_EINVAL, "channel %d: p_fire=%g, outcomes=%d", c, p_fire[c], n_outcomes[c]);
    }
    l1p[c] = p_fire[c] >= 1.0 ? 0.0 : log1p(-p_fire[c]);
    off[c + 1] = off[c] + (uint32_t)n_outcomes[c];
    pmax = std::max(pmax, p_fire[c]);
  }
  const size_t tot = off[n_channels];
  // outcome thresholds: u = w / 2^32 with a 32-bit draw w, outcome = #{i : cdf[i] <= u} = #{i : ceil(cdf[i] 2^32) <= w}
  // (round 3 compared a 24-bit uniform with a float32 CDF: a conditional outcome below 6e-8 could never be drawn)
  std::vector<uint32_t> cdf(std::max<size_t>(1, tot));
  for (size_t i = 0; i < tot; ++i) {
    const double t = std::ceil(std::min(std::max(cond_cdf[i], 0.0), 1.0) * 4294967296.0);
    cdf[i] = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
  }
  std::vector<uint64_t> pat(std::max<size_t>(1, tot * n->WF), 0ull);
  for (size_t o = 0; o < tot; ++o)
    for (int i = 0; i < num_f; ++i)
      if (xor_patterns[o * (size_t)num_f + i]) pat[o * n->WF + (i >> 6)] |= 1ull << (i & 63);
  // segment length: about 8 expected fires of the most active channel per thread
  int seg = 64;
  while (seg < 65536 && seg * pmax < 8.0) seg *= 2;
  n->seg = seg;
  // tile form: 32 KB of rows per block; sub-segments of about 4 expected fires of the most active channel
  std::vector<float> inv((size_t)std::max(1, n_channels));
  for (int c = 0; c < n_channels; ++c) inv[c] = p_fire[c] >= 1.0 ? 0.0f : (float)(1.0 / log2(1.0 - p_fire[c]));
  n->tile = 0;
  if (n->WF * 8 * 64 <= 32 * 1024) {
    n->tile = 64;
    while (n->tile * 2 * n->WF * 8 <= 32 * 1024) n->tile *= 2;
    int ts = 64;
    while (ts < n->tile && ts * pmax < 4.0) ts *= 2;
    n->tseg = ts;
  }
  // k_noise_wave: the tile as above (at most 4096 shots: the fires a channel expects in it decide the group size); lanes per
  // group: about 1.5 x the expected fires of the most active channel / 2 gaps per lane.  TSIM_AMD_TUNE=noise_wave=0: k_noise_tile
  std::vector<uint32_t> pw_off(tot + 1, 0u), pw;
  for (size_t o = 0; o < tot; ++o) {
    for (int w = 0; w < n->WF; ++w)
      if (pat[o * n->WF + w]) {
        pw.push_back((uint32_t)w);
        pw.push_back((uint32_t)pat[o * n->WF + w]);
        pw.push_back((uint32_t)(pat[o * n->WF + w] >> 32));
        pw.push_back(0u);
      }
    pw_off[o + 1] = (uint32_t)(pw.size() / 4);
  }
  if (pw.empty()) pw.assign(4, 0u);
  n->wave_g = 0;
  {
    const char *tn = getenv("TSIM_AMD_TUNE");
    const bool off = tn && strstr(tn, "noise_wave=0");
    if (n->WF * 8 * 64 <= 48 * 1024 && !off && (size_t)n_channels * 24 <= 24 * 1024) {
      // rows of the tile in up to 48 KB of LDS, at most 4096 shots (C2: 4096 x 8 B, C3: 2048 x 16 B, C5: 1024 x 40 B)
      n->wave_tile = 64;
      while (n->wave_tile < 4096 && (size_t)n->wave_tile * 2 * n->WF * 8 <= 48 * 1024) n->wave_tile *= 2;
      // gaps per round: what covers the tile in ONE round but for ~2 % of the groups (mean + two sigma of the fires a channel expects)
      const double fires = pmax * n->wave_tile;
      int g = 8;
      while (g < 64 && 2.0 * g < fires + 2.0 * std::sqrt(fires) + 2.0) g *= 2;
      n->wave_g = g;
    }
  }
  hipError_t e = hipSuccess;
  if ((e = hipMalloc((void **)&n->d_pw_off, pw_off.size() * 4)) != hipSuccess || (e = hipMalloc((void **)&n->d_pw, pw.size() * 4)) != hipSuccess ||
      (e = hipMemcpy(n->d_pw_off, pw_off.data(), pw_off.size() * 4, hipMemcpyHostToDevice)) != hipSuccess ||
      (e = hipMemcpy(n->d_pw, pw.data(), pw.size() * 4, hipMemcpyHostToDevice)) != hipSuccess) {
    tsim_noise_destroy(n);
    return tsim_fail(TSIM_ENOMEM, "noise sampler tables: %s", hipGetErrorString(e));
  }
  if ((e = hipMalloc((void **)&n->d_l1p, l1p.size() * 8)) != hipSuccess ||
      (e = hipMalloc((void **)&n->d_inv, inv.size() * 4)) != hipSuccess ||
      (e = hipMalloc((void **)&n->d_off, off.size() * 4)) != hipSuccess ||
      (e = hipMalloc((void **)&n->d_cdf, cdf.size() * 4)) != hipSuccess ||
      (e = hipMalloc((void **)&n->d_pat, pat.size() * 8)) != hipSuccess) {
    tsim_noise_destroy(n);
    return tsim_fail(TSIM_ENOMEM, "hipMalloc failed: %s", hipGetErrorString(e));
  }
  if ((e = hipMemcpy(n->d_l1p, l1p.data(), l1p.size() * 8, hipMemcpyHostToDevice)) != hipSuccess ||
      (e = hipMemcpy(n->d_inv, inv.data(), inv.size() * 4, hipMemcpyHostToDevice)) != hipSuccess ||
      (e = hipMemcpy(n->d_off, off.data(), off.size() * 4, hipMemcpyHostToDevice)) != hipSuccess ||
      (e = hipMemcpy(n->d_cdf, cdf.data(), cdf.size() * 4, hipMemcpyHostToDevice)) != hipSuccess ||
      (e = hipMemcpy(n->d_pat, pat.data(), pat.size() * 8, hipMemcpyHostToDevice)) != hipSuccess) {
    tsim_noise_destroy(n);
    return tsim_fail(TSIM_EHIP, "hipMemcpy failed: %s", hipGetErrorString(e));
  }
  *out = n;
  return TSIM_OK;
}

extern "C" int tsim_noise_sample_device(tsim_noise *n, int64_t B, uint32_t key_hi, uint32_t key_lo, uint64_t *d_f,
                                        void *stream) {
  if (!n || !n->prog) return tsim_fail(TSIM_EINVAL, "noise sampler is NULL");
  if (int r = tsim_set_device(n->prog)) return r;
  if (B < 0) return tsim_fail(TSIM_EINVAL, "negative B");
  if (B == 0) return TSIM_OK;
  if (!d_f) return tsim_fail(TSIM_EINVAL, "f buffer is NULL");
  hipStream_t s = stream ? (hipStream_t)stream : n->prog->stream;
  if (n->wave_g > 0 && n->n_ch > 0) {  // ... with groups of lanes per (channel, tile): k_noise_wave
    NoiseWaveArgs a{};
    a.inv_log2_1mp = n->d_inv;
    a.cdf_off = n->d_off;
    a.cdf = n->d_cdf;
    a.pw_off = n->d_pw_off;
    a.pw = n->d_pw;
    a.f = (unsigned long long *)d_f;
    a.B = B;
    a.n_ch = n->n_ch;
    a.WF = n->WF;
    a.tile = n->wave_tile;
    a.g = n->wave_g;
    a.k0 = key_hi;
    a.k1 = key_lo;
    const long long tiles = (B + n->wave_tile - 1) / n->wave_tile;
    if (tiles > 0x7FFFFFFFll) return tsim_fail(TSIM_ENOTSUP, "noise launch too large");
    // threads: one group of g lanes per channel, up to 1024 (a grid of 245 tiles is one block per CU: the block's own chain -
    // channels per group - is the kernel's time)
    int threads = 256;
    while (threads < 1024 && threads < n->n_ch * n->wave_g) threads *= 2;
    threads = std::max(threads, n->wave_g);
    hipLaunchKernelGGL(k_noise_wave, dim3((unsigned)tiles), dim3(threads), (size_t)n->wave_tile * n->WF * 8 + (size_t)n->n_ch * 24, s, a);
    HIP_TRY(hipGetLastError());
    return TSIM_OK;
  }
  if (n->tile > 0 && n->n_ch > 0) {  // one block per tile of shots: rows built in LDS, written once (tsim_noise.hip.h)
    NoiseArgs a{};
    a.inv_log2_1mp = n->d_inv;
    a.cdf_off = n->d_off;
    a.cdf = n->d_cdf;
    a.patterns = n->d_pat;
    a.f = (unsigned long long *)d_f;
    a.B = B;
    a.n_ch = n->n_ch;
    a.WF = n->WF;
    a.seg = n->tseg;
    a.tile = n->tile;
    a.n_tiles = (int)((B + n->tile - 1) / n->tile);
    a.k0 = key_hi;
    a.k1 = key_lo;
    hipLaunchKernelGGL(k_noise_tile, dim3((unsigned)a.n_tiles), dim3(256), (size_t)n->tile * n->WF * 8, s, a);
    HIP_TRY(hipGetLastError());
    return TSIM_OK;
  }
  HIP_TRY(hipMemsetAsync(d_f, 0, (size_t)B * n->WF * 8, s));
  if (n->n_ch == 0) return TSIM_OK;
  NoiseArgs a{};
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
  if (grid > 0x7FFFFFFFll) return tsim_fail(TSIM_ENOTSUP, "noise launch too large");
  hipLaunchKernelGGL(k_noise, dim3((unsigned)grid), dim3(256), 0, s, a);
  HIP_TRY(hipGetLastError());
  return TSIM_OK;
}

// the request's batch launcher (tsim_sample.hip calls it for the paths that do not fuse the noise into their first pass)
static int noise_launch_one(void *noise, int64_t B, uint32_t k0, uint32_t k1, uint64_t *d_f, hipStream_t s) {
  return tsim_noise_sample_device((tsim_noise *)noise, B, k0, k1, d_f, (void *)s);
}

// sample_program over n_steps batches whose f rows are drawn HERE (reference: src/tsim/sampler.py:393-400 - channel sampler, then
// sample_program, per batch): batch j's noise key is split(noise_key) threaded like the sampling key (noise_key advances by one
// split per batch); its f rows land in d_f[j] (the caller's buffers: the hard-row kernels read them, and so may the caller).
extern "C" int tsim_sample_steps_noise_device(tsim_program *p, tsim_noise *n, int32_t n_steps, uint64_t *const *d_f, int64_t B, int32_t num_f,
                                              uint32_t key[2], uint32_t noise_key[2], int64_t shot_offset, void *const *d_out,
                                              float *const *d_max_norm_dev, uint32_t flags) {
  if (!n || n->prog != p) return tsim_fail(TSIM_EINVAL, "noise sampler is NULL or belongs to another program");
  if (!noise_key) return tsim_fail(TSIM_EINVAL, "NULL argument");
  if (num_f != n->num_f) return tsim_fail(TSIM_EINVAL, "the noise sampler draws %d f bits, the call says %d", n->num_f, num_f);
  if (n_steps <= 0) return tsim_sample_steps_device(p, n_steps, (const uint64_t *const *)d_f, B, num_f, key, shot_offset, d_out, d_max_norm_dev, flags);
  std::vector<uint32_t> keys(2 * (size_t)n_steps);
  for (int j = 0; j < n_steps; ++j) {
    uint32_t o[4];
    tsim_key_split(noise_key[0], noise_key[1], o);
    noise_key[0] = o[0];
    noise_key[1] = o[1];
    keys[2 * (size_t)j] = o[2];
    keys[2 * (size_t)j + 1] = o[3];
  }
  TsimNoiseRequest rq{};
  rq.noise = n;
  rq.launch = noise_launch_one;
  rq.keys = keys.data();
  rq.base = 0;
  rq.fusable = false;
  if (n->wave_g > 0 && n->n_ch > 0) {
    rq.N.inv_log2_1mp = n->d_inv;
    rq.N.cdf_off = n->d_off;
    rq.N.cdf = n->d_cdf;
    rq.N.pw_off = n->d_pw_off;
    rq.N.pw = n->d_pw;
    rq.N.f = nullptr;
    rq.N.B = B;
    rq.N.n_ch = n->n_ch;
    rq.N.WF = n->WF;
    rq.N.tile = n->wave_tile;
    rq.N.g = n->wave_g;
    // (the fused kernel's blocks are 1024 threads: groups of g lanes for up to 1024 / g channels at a time; LDS: the tile, the
    // channel records and the first pass's ~5 KB of tables - two blocks per CU)
    rq.fusable = (n->wave_tile % 1024) == 0 && (size_t)n->wave_tile * n->WF * 8 + (size_t)n->n_ch * 24 <= 56 * 1024;
  }
  g_noise_req = &rq;
  // (the f rows do not exist before this call: whatever the caller says about its inputs, every path orders its reads behind the noise)
  const int r = tsim_sample_steps_device(p, n_steps, (const uint64_t *const *)d_f, B, num_f, key, shot_offset, d_out, d_max_norm_dev, flags);
  g_noise_req = nullptr;
  return r;
}

extern "C" void tsim_noise_destroy(tsim_noise *n) {
  if (!n) return;
  if (n->device >= 0) (void)hipSetDevice(n->device);
  if (n->d_l1p) (void)hipFree(n->d_l1p);
  if (n->d_inv) (void)hipFree(n->d_inv);
  if (n->d_off) (void)hipFree(n->d_off);
  if (n->d_cdf) (void)hipFree(n->d_cdf);
  if (n->d_pat) (void)hipFree(n->d_pat);
  if (n->d_pw_off) (void)hipFree(n->d_pw_off);
  if (n->d_pw) (void)hipFree(n->d_pw);
  delete n;
}
