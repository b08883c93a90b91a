// tsim_direct.hip.h - programs WITHOUT compiled components: every output is a direct one, out = f[idx] ^ flip
// (sample_program, src/tsim/sampler.py:140-145,164-166 - the whole of it for Clifford-only circuits: BASELINE config C1,
// everything tsim_amd.clifford compiles).  No random numbers, no tables: 8 * WF bytes of packed f in, ceil(n_out / 8)
// bytes out per shot - the one sampling path that IS HBM-bound.  The general row kernel served it until round 3 (one
// launch per batch, LDS staging, 256-thread blocks: 22 us per 10^6 shots of C1 = 0.5 TB/s); this is the streaming form:
// a lane per shot, the f row in registers (one 8- or 16-byte load), the packer's bit-field runs (emit_gather_program:
// ((f_word >> s) & mask ^ flip) << d, wave-uniform operands read through the scalar cache), the row out in as few stores
// as its size allows - and up to eight batches per launch (tsim_sample_steps_device), so that launches are not the bound.
#pragma once
#include "tsim_kernels.hip.h"

namespace tsimk {

#define TSIMK_DIRECT_MAX_STEPS 8
struct DirectStep {
  const uint32_t *f;     // [B, WF32] packed f rows
  uint64_t *out;         // [B, WO] or nullptr
  uint8_t *out_compact;  // [B, out_rb] or nullptr
};
struct DirectMultiArgs {
  const uint32_t *img;
  long long B;
  int n_steps, blocks_per_step;
  int prog, chunks;  // gather program: image offset, 16-word chunks of four runs
  int WO, out_rb;    // 64-bit words per padded row, bytes per compact row
  DirectStep step[TSIMK_DIRECT_MAX_STEPS];
};

#define TSIMK_DIRECT_RPT 4  // rows per thread: their loads are all in flight before the first row is processed
template <int WF32>
__global__ void __launch_bounds__(256) k_direct_multi(DirectMultiArgs M) {
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const uint32_t st = blockIdx.x / (uint32_t)M.blocks_per_step, rb = blockIdx.x - st * (uint32_t)M.blocks_per_step;
  typedef const __attribute__((address_space(4))) uint8_t *cbytes;
  typedef const __attribute__((address_space(4))) DirectStep *cstep;
  cstep S = (cstep)((cbytes)__builtin_amdgcn_kernarg_segment_ptr() + __builtin_offsetof(DirectMultiArgs, step)) + st;
  const long long row0 = (long long)rb * (256 * TSIMK_DIRECT_RPT) + threadIdx.x;
  uint32_t x[TSIMK_DIRECT_RPT][WF32];
  const uint32_t *f = S->f;
#pragma unroll
  for (int t = 0; t < TSIMK_DIRECT_RPT; ++t) {
    const long long row = row0 + 256 * t;
#pragma unroll
    for (int w = 0; w < WF32; ++w) x[t][w] = 0u;
    if (row < M.B) {
      const uint32_t *fr = f + row * WF32;
      if constexpr (WF32 == 2) {
        const u32x2 v = *reinterpret_cast<const u32x2 *>(fr);
        x[t][0] = v.x; x[t][1] = v.y;
      } else {
#pragma unroll
        for (int w = 0; w < WF32; w += 4) {
          const u32x4 v = *reinterpret_cast<const u32x4 *>(fr + w);
          x[t][w] = v.x; x[t][w + 1] = v.y; x[t][w + 2] = v.z; x[t][w + 3] = v.w;
        }
      }
    }
  }
  uint32_t o[TSIMK_DIRECT_RPT][4];
#pragma unroll
  for (int t = 0; t < TSIMK_DIRECT_RPT; ++t)
#pragma unroll
    for (int w = 0; w < 4; ++w) o[t][w] = 0u;
  cptr prog = (cptr)(uintptr_t)M.img + M.prog;
  for (int c = 0; c < M.chunks; ++c) {
    const lw_u32x16 q = *(lw_cptr16)(prog + 16 * c);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t ctl = q[4 * k];
      const uint32_t sw = ctl >> 24;  // wave-uniform: the selects below are scalar compares
      const uint32_t dw = (ctl >> 16) & 255u;
#pragma unroll
      for (int t = 0; t < TSIMK_DIRECT_RPT; ++t) {
        uint32_t fw = x[t][0];
#pragma unroll
        for (int w = 1; w < WF32; ++w) fw = sw == (uint32_t)w ? x[t][w] : fw;
        const uint32_t v = (((fw >> (ctl & 31u)) & q[4 * k + 1]) ^ q[4 * k + 2]) << ((ctl >> 8) & 31u);
#pragma unroll
        for (int w = 0; w < 4; ++w) o[t][w] |= dw == (uint32_t)w ? v : 0u;
      }
    }
  }
  uint64_t *out = S->out;
  uint8_t *oc = S->out_compact;
  const int rb8 = M.out_rb;
#pragma unroll
  for (int t = 0; t < TSIMK_DIRECT_RPT; ++t) {
    const long long row = row0 + 256 * t;
    if (row >= M.B) break;
    if (out) {
      uint32_t *dst = reinterpret_cast<uint32_t *>(out + row * M.WO);
      u32x2 v;
      v.x = o[t][0]; v.y = o[t][1];
      *reinterpret_cast<u32x2 *>(dst) = v;
      if (M.WO > 1) {
        v.x = o[t][2]; v.y = o[t][3];
        *reinterpret_cast<u32x2 *>(dst + 2) = v;
      }
    }
    if (oc) {
      // out_rb bytes at row * out_rb, any alignment (the device runs in unaligned-access mode): dwords, then a short, a byte
      uint8_t *dst = oc + row * rb8;
      int k = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w)
        if (k + 4 <= rb8) {
          typedef uint32_t __attribute__((aligned(1))) u32u;
          *reinterpret_cast<u32u *>(dst + k) = o[t][w];
          k += 4;
        }
      const int rem = rb8 - k;
      if (rem > 0) {
        uint32_t wv = o[t][0];
#pragma unroll
        for (int w = 1; w < 4; ++w) wv = (k >> 2) == w ? o[t][w] : wv;
        typedef uint16_t __attribute__((aligned(1))) u16u;
        if (rem >= 2) *reinterpret_cast<u16u *>(dst + k) = (uint16_t)wv;
        if (rem & 1) dst[k + (rem == 3 ? 2 : 0)] = (uint8_t)(wv >> (rem == 3 ? 16 : 0));
      }
    }
  }
}

}  // namespace tsimk
