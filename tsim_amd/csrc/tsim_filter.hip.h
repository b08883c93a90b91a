// tsim_filter.hip.h - the two small non-template kernels of the sampling TU: the direct-detector
// post-selection filter and the per-output subkey chain.
#pragma once
#include "tsim_kernels.hip.h"

namespace tsimk {

// ---------------------------------------------------------------------------
// device-side post-selection (reference: src/tsim/sampler.py:422-545, the part that decides
// which shots reach the sampling kernel): every row gets its DIRECT output bits written; rows in
// which a masked direct detector fires (after the optional XOR with the reference sample) are
// discarded, the others are appended to the survivor list.
// ---------------------------------------------------------------------------
struct FilterArgs {
  const uint32_t *img;
  const uint64_t *f;       // [B, WF]
  uint64_t *out;           // [B, WO] direct bits, zero elsewhere
  const uint64_t *mask;    // [WO] masked direct detector columns
  const uint64_t *ref;     // [WO] reference bits XORed before the test (or nullptr)
  uint32_t *row_index;     // [B] survivors (unordered)
  uint32_t *row_count;     // zeroed by the caller
  uint8_t *discarded;      // [B] 0/1 (or nullptr)
  long long B;
  int WF, WO, n_direct, direct_off;
};

__global__ void __launch_bounds__(256) k_direct_filter(FilterArgs A) {
  const int nthr = blockDim.x;
  const long long row = (long long)blockIdx.x * nthr + threadIdx.x;
  if (row >= A.B) return;
  cptr img = (cptr)(uintptr_t)A.img;
  const int WF32 = 2 * A.WF, WO32 = 2 * A.WO;
  uint32_t *lds_f = tsimk_lds + threadIdx.x;
  uint32_t *lds_o = tsimk_lds + WF32 * nthr + threadIdx.x;
  const uint64_t *frow = A.f + row * A.WF;
  for (int w = 0; w < A.WF; ++w) {
    const uint64_t v = frow[w];
    lds_f[(2 * w) * nthr] = (uint32_t)v;
    lds_f[(2 * w + 1) * nthr] = (uint32_t)(v >> 32);
  }
  for (int w = 0; w < WO32; ++w) lds_o[w * nthr] = 0u;
  cptr dt = img + A.direct_off;
  for (int j = 0; j < A.n_direct; ++j) {
    const uint32_t s = dt[2 * j], dst = dt[2 * j + 1];
    const uint32_t src = s & 0x7FFFFFFFu;
    const uint32_t bit = ((lds_f[(src >> 5) * nthr] >> (src & 31u)) ^ (s >> 31)) & 1u;
    lds_o[(dst >> 5) * nthr] |= bit << (dst & 31u);
  }
  bool discard = false;
  uint64_t *orow = A.out + row * A.WO;
  for (int w = 0; w < A.WO; ++w) {
    const uint64_t v = (uint64_t)lds_o[(2 * w) * nthr] | ((uint64_t)lds_o[(2 * w + 1) * nthr] << 32);
    orow[w] = v;
    const uint64_t r = A.ref ? A.ref[w] : 0ull;
    discard = discard || (((v ^ r) & A.mask[w]) != 0ull);
  }
  if (A.discarded) A.discarded[row] = discard ? 1 : 0;
  if (!discard) A.row_index[atomicAdd(A.row_count, 1u)] = (uint32_t)row;
}

// ---------------------------------------------------------------------------
// per-output subkeys: `key, subkey = jax.random.split(key)` once per compiled
// output, threaded through the components in processing order
// (sampler.py:74,147-148).  A sequential chain, one thread, stream-ordered.
// ---------------------------------------------------------------------------
__global__ void k_keygen(uint32_t k0, uint32_t k1, int n, uint32_t *__restrict__ subkeys) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  for (int i = 0; i < n; ++i) {
    uint32_t a0 = 0u, a1 = 0u, b0 = 0u, b1 = 1u;
    threefry2x32(k0, k1, a0, a1);  // split(key)[0] -> next key
    threefry2x32(k0, k1, b0, b1);  // split(key)[1] -> this output's subkey
    subkeys[2 * i] = b0;
    subkeys[2 * i + 1] = b1;
    k0 = a0;
    k1 = a1;
  }
}

// ... computed on the host when they fit the kernel arguments (the chain is 2 n dependent Threefry blocks: 44 us on one lane for
// 39 outputs, per batch, in front of the first pass - class 20narrow; ~3 us on a host core): this kernel only stores them.
#define TSIMK_KEYPUT_MAX 320
struct KeyPutArgs {
  int n;
  uint32_t keys[2 * TSIMK_KEYPUT_MAX];
};
__global__ void k_keyput(KeyPutArgs K, uint32_t *__restrict__ subkeys) {
  for (int i = (int)threadIdx.x; i < 2 * K.n; i += (int)blockDim.x) subkeys[i] = K.keys[i];
}

}  // namespace tsimk
