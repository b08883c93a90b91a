// tsim_format.hip - data-format kernels either side of the path: byte-per-bit <-> packed rows
// (the reference's H2D / D2H formats, sampler.py:398,415) and the bit_packed compaction (sampler.py:665-669).
#include "tsim_internal.hip.h"
#include "tsim_format.hip.h"

using namespace tsimk;

int tsim_launch_pack(tsim_program *p, const uint8_t *d_in, int64_t B, int32_t nbits, uint64_t *d_out, hipStream_t s) {
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

// one launch covers fewer than 2^32 output bytes (32-bit indexing in the kernels): larger requests are cut
// into row chunks (a multiple of 4 rows, so that every chunk starts 4-byte - and, for 16-column-aligned
// rows, 16-byte - aligned)
static int unpack_chunk(const uint64_t *d_in, int64_t B, int32_t nbits, uint8_t *d_out, hipStream_t s) {
  const int n32 = 2 * ((nbits + 63) / 64);
  const long long total = B * (long long)nbits;
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

int tsim_launch_unpack(tsim_program *p, const uint64_t *d_in, int64_t B, int32_t nbits, uint8_t *d_out, hipStream_t s) {
  (void)p;
  if (B == 0 || nbits == 0) return 0;
  const int WQ = (nbits + 63) / 64;
  const long long max_rows = std::max<long long>(4, (((1ll << 32) - 1) / nbits) & ~3ll);
  for (long long r0 = 0; r0 < B; r0 += max_rows) {
    const long long rows = std::min<long long>(max_rows, B - r0);
    if (int r = unpack_chunk(d_in + r0 * WQ, rows, nbits, d_out + r0 * (long long)nbits, s)) return r;
  }
  return 0;
}

extern "C" int tsim_pack_bits_device(tsim_program *p, const uint8_t *d_in, int64_t B, int32_t nbits,
                                     uint64_t *d_out, void *stream) {
  if (int r = tsim_need_final(p)) return r;
  if (int r = tsim_set_device(p)) return r;
  if (B < 0 || nbits < 0) return tsim_fail(TSIM_EINVAL, "negative size");
  return tsim_launch_pack(p, d_in, B, nbits, d_out, stream ? (hipStream_t)stream : p->stream);
}

extern "C" int tsim_unpack_bits_device(tsim_program *p, const uint64_t *d_in, int64_t B, int32_t nbits,
                                       uint8_t *d_out, void *stream) {
  if (int r = tsim_need_final(p)) return r;
  if (int r = tsim_set_device(p)) return r;
  if (B < 0 || nbits < 0) return tsim_fail(TSIM_EINVAL, "negative size");
  return tsim_launch_unpack(p, d_in, B, nbits, d_out, stream ? (hipStream_t)stream : p->stream);
}

int tsim_launch_compact(const uint64_t *d_in, int64_t B, int32_t WO, int32_t nbits, uint8_t *d_out, hipStream_t s) {
  const int rb = (nbits + 7) / 8;
  const long long nthreads = (B + 3) / 4;
  const uint32_t tail_mask = (nbits & 7) ? ((1u << (nbits & 7)) - 1u) : 255u;
  hipLaunchKernelGGL(k_compact_rows, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, s, d_in, d_out, (long long)B, WO,
                     rb, tail_mask);
  HIP_TRY(hipGetLastError());
  return TSIM_OK;
}

extern "C" int tsim_compact_rows_device(tsim_program *p, const uint64_t *d_in, int64_t B, int32_t in_words,
                                        int32_t nbits, uint8_t *d_out, void *stream) {
  if (int r = tsim_need_final(p)) return r;
  if (int r = tsim_set_device(p)) return r;
  if (B < 0 || nbits < 0 || in_words < 0) return tsim_fail(TSIM_EINVAL, "negative size");
  if (B == 0 || nbits == 0) return TSIM_OK;
  if (!d_in || !d_out) return tsim_fail(TSIM_EINVAL, "NULL buffer");
  const int WO = in_words ? in_words : (nbits + 63) / 64;
  if ((long long)WO * 64 < nbits) return tsim_fail(TSIM_EINVAL, "rows of %d words hold fewer than %d bits", WO, nbits);
  return tsim_launch_compact(d_in, B, WO, nbits, d_out, stream ? (hipStream_t)stream : p->stream);
}

// ---------------------------------------------------------------------------
// Post-selection AFTER sampling (device-side noise: no host stream prescribes which shots reach sample_program, so every
// row is sampled by the fast path and the discarded ones are blanked here - what _sample_batches_with_postselection,
// src/tsim/sampler.py:422-545, returns for them).  Rows are byte strings (`row_bytes` each: the padded 8-byte words or
// the bit_packed layout - same bit order).  masks = five byte strings of row_bytes:
//   [0] test mask (masked, directly readable detectors)   [1] reference bits XORed before the test
//   [2] columns a discarded row keeps (its direct detector columns, sampler.py:532-540)
//   [3] XOR applied to surviving rows                      [4] XOR applied to discarded rows
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_postselect_blank(uint8_t *rows, long long B, int row_bytes, const uint8_t *masks, uint8_t *gone) {
  const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
  if (row >= B) return;
  uint8_t *r = rows + row * row_bytes;
  unsigned fired = 0u;
  for (int k = 0; k < row_bytes; ++k) fired |= (unsigned)((r[k] ^ masks[row_bytes + k]) & masks[k]);
  const bool g = fired != 0u;
  const uint8_t *keep = masks + 2 * row_bytes, *xs = masks + 3 * row_bytes, *xg = masks + 4 * row_bytes;
  for (int k = 0; k < row_bytes; ++k) r[k] = g ? (uint8_t)((r[k] & keep[k]) ^ xg[k]) : (uint8_t)(r[k] ^ xs[k]);
  if (gone) gone[row] = g ? 1 : 0;
}

extern "C" int tsim_postselect_rows_device(tsim_program *p, uint8_t *d_rows, int64_t B, int32_t row_bytes, const uint8_t *d_masks,
                                           uint8_t *d_gone, void *stream) {
  if (int r = tsim_need_final(p)) return r;
  if (int r = tsim_set_device(p)) return r;
  if (B < 0 || row_bytes < 0) return tsim_fail(TSIM_EINVAL, "negative size");
  if (B == 0 || row_bytes == 0) return TSIM_OK;
  if (!d_rows || !d_masks) return tsim_fail(TSIM_EINVAL, "NULL buffer");
  hipLaunchKernelGGL(k_postselect_blank, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, stream ? (hipStream_t)stream : p->stream, d_rows,
                     (long long)B, (int)row_bytes, d_masks, d_gone);
  HIP_TRY(hipGetLastError());
  return TSIM_OK;
}

// ---------------------------------------------------------------------------
// Survivors of a chunk, IN SHOT ORDER, appended to a device-resident queue (the reference's compacted batches,
// sampler.py:466-508: a survivor's Threefry counter is its position in the batch it leaves in, so the order is part of
// the contract).  gone[i] = 1 for discarded rows (tsim_postselect_device); queue[*tail ...] receives base + i for the
// others; *tail advances by their number.  Two small kernels: survivors per 1024-row block, then every block finds
// its offset (a sum over the blocks before it) and writes its ids by ballot prefix.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_survivor_count(const uint8_t *gone, long long n, uint32_t *block_counts) {
  const long long i = (long long)blockIdx.x * 1024 + threadIdx.x;
  const bool keep = i < n && gone[i] == 0;
  const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
  __shared__ uint32_t wave_n[16];
  if ((threadIdx.x & 63u) == 0u) wave_n[threadIdx.x >> 6] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0u) {
    uint32_t t = 0;
    for (int w = 0; w < 16; ++w) t += wave_n[w];
    block_counts[blockIdx.x] = t;
  }
}
__global__ void __launch_bounds__(1024) k_survivor_write(const uint8_t *gone, long long n, uint32_t base, const uint32_t *block_counts,
                                                         uint32_t *queue, uint32_t *tail) {
  __shared__ uint32_t wave_n[16];
  __shared__ uint32_t block_off;
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  const uint32_t start = *tail;  // (advanced by the LAST block only, after every block has read it: see below)
  // offset of this block: survivors of the blocks before it
  uint32_t part = 0;
  for (uint32_t b = threadIdx.x; b < blockIdx.x; b += 1024u) part += block_counts[b];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += (uint32_t)__shfl_xor((int)part, o, 64);
  if (lane == 0u) wave_n[wv] = part;
  __syncthreads();
  if (threadIdx.x == 0u) {
    uint32_t t = 0;
    for (int w = 0; w < 16; ++w) t += wave_n[w];
    block_off = t;
  }
  __syncthreads();
  const long long i = (long long)blockIdx.x * 1024 + threadIdx.x;
  const bool keep = i < n && gone[i] == 0;
  const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
  __syncthreads();  // (wave_n is reused)
  if (lane == 0u) wave_n[wv] = (uint32_t)__popcll(m);
  __syncthreads();
  uint32_t before = 0;
  for (uint32_t w = 0; w < wv; ++w) before += wave_n[w];
  if (keep) queue[start + block_off + before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = base + (uint32_t)i;
}
__global__ void k_survivor_advance(const uint32_t *block_counts, uint32_t n_blocks, uint32_t *tail) {
  uint32_t part = 0;
  for (uint32_t b = threadIdx.x; b < n_blocks; b += 64u) part += block_counts[b];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += (uint32_t)__shfl_xor((int)part, o, 64);
  if (threadIdx.x == 0u) *tail += part;
}

extern "C" int tsim_survivors_append_device(tsim_program *p, const uint8_t *d_gone, int64_t n, uint32_t base, uint32_t *d_scratch,
                                            uint32_t *d_queue, uint32_t *d_tail, void *stream) {
  if (int r = tsim_need_final(p)) return r;
  if (int r = tsim_set_device(p)) return r;
  if (n < 0 || n >= (1ll << 32)) return tsim_fail(TSIM_EINVAL, "bad row count");
  if (n == 0) return TSIM_OK;
  if (!d_gone || !d_scratch || !d_queue || !d_tail) return tsim_fail(TSIM_EINVAL, "NULL buffer");
  hipStream_t s = stream ? (hipStream_t)stream : p->stream;
  const unsigned nb = (unsigned)((n + 1023) / 1024);
  hipLaunchKernelGGL(k_survivor_count, dim3(nb), dim3(1024), 0, s, d_gone, (long long)n, d_scratch);
  hipLaunchKernelGGL(k_survivor_write, dim3(nb), dim3(1024), 0, s, d_gone, (long long)n, base, (const uint32_t *)d_scratch, d_queue, d_tail);
  hipLaunchKernelGGL(k_survivor_advance, dim3(1), dim3(64), 0, s, (const uint32_t *)d_scratch, nb, d_tail);
  HIP_TRY(hipGetLastError());
  return TSIM_OK;
}

// ---------------------------------------------------------------------------
// Output arrangement on the device (CompiledDetectorSampler.sample's epilogue, sampler.py:850-868, and _maybe_bit_pack,
// :665-669): out column c = in column (cols[c] & 0x7FFFFFFF) XOR (cols[c] >> 31), as one byte per column
// (bool arrays) or bit-packed, np.packbits(..., axis=1, bitorder="little").  Covers detectors only / appended /
// prepended / separate observables and the reference-sample flips without a host pass over the rows.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_arrange_rows(const uint64_t *__restrict__ in, long long B, int WO, const uint32_t *__restrict__ cols,
                                                      int n_cols, int packed, uint8_t *__restrict__ out) {
  const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
  if (row >= B) return;
  const uint32_t *w = reinterpret_cast<const uint32_t *>(in + row * WO);
  if (!packed) {
    uint8_t *dst = out + row * n_cols;
    for (int c = 0; c < n_cols; ++c) {
      const uint32_t s = cols[c], src = s & 0x7FFFFFFFu;
      dst[c] = (uint8_t)(((w[src >> 5] >> (src & 31u)) ^ (s >> 31)) & 1u);
    }
    return;
  }
  const int rb = (n_cols + 7) / 8;
  uint8_t *dst = out + row * rb;
  uint32_t acc = 0u;
  for (int c = 0; c < n_cols; ++c) {
    const uint32_t s = cols[c], src = s & 0x7FFFFFFFu;
    acc |= (((w[src >> 5] >> (src & 31u)) ^ (s >> 31)) & 1u) << (c & 7);
    if ((c & 7) == 7 || c == n_cols - 1) {
      dst[c >> 3] = (uint8_t)acc;
      acc = 0u;
    }
  }
}

extern "C" int tsim_arrange_rows_device(tsim_program *p, const uint64_t *d_in, int64_t B, int32_t in_words, const uint32_t *d_cols,
                                        int32_t n_cols, int32_t packed, uint8_t *d_out, void *stream) {
  if (int r = tsim_need_final(p)) return r;
  if (int r = tsim_set_device(p)) return r;
  if (B < 0 || n_cols < 0 || in_words < 1) return tsim_fail(TSIM_EINVAL, "bad size");
  if (B == 0 || n_cols == 0) return TSIM_OK;
  if (!d_in || !d_cols || !d_out) return tsim_fail(TSIM_EINVAL, "NULL buffer");
  hipLaunchKernelGGL(k_arrange_rows, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, stream ? (hipStream_t)stream : p->stream, d_in,
                     (long long)B, (int)in_words, d_cols, (int)n_cols, (int)packed, d_out);
  HIP_TRY(hipGetLastError());
  return TSIM_OK;
}

extern "C" int tsim_gather_rows_device(tsim_program *p, const uint64_t *d_src, int32_t words, const uint32_t *d_index,
                                       int64_t n_valid, int64_t n_total, uint64_t *d_dst, void *stream) {
  if (int r = tsim_need_final(p)) return r;
  if (int r = tsim_set_device(p)) return r;
  if (words < 1 || n_valid < 0 || n_total < n_valid) return tsim_fail(TSIM_EINVAL, "bad gather sizes");
  if (n_total == 0) return TSIM_OK;
  if (n_valid == 0) return tsim_fail(TSIM_EINVAL, "gather with padding needs at least one valid row");
  if (!d_src || !d_index || !d_dst) return tsim_fail(TSIM_EINVAL, "NULL buffer");
  const long long n = (long long)n_total * words;
  hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream ? (hipStream_t)stream : p->stream,
                     d_src, d_index, (long long)n_valid, (long long)n_total, words, d_dst);
  HIP_TRY(hipGetLastError());
  return TSIM_OK;
}

extern "C" int tsim_scatter_rows_device(tsim_program *p, const uint64_t *d_src, int32_t words, const uint32_t *d_index,
                                        int64_t n, uint64_t *d_dst, void *stream) {
  if (int r = tsim_need_final(p)) return r;
  if (int r = tsim_set_device(p)) return r;
  if (words < 1 || n < 0) return tsim_fail(TSIM_EINVAL, "bad scatter sizes");
  if (n == 0) return TSIM_OK;
  if (!d_src || !d_index || !d_dst) return tsim_fail(TSIM_EINVAL, "NULL buffer");
  const long long t = (long long)n * words;
  hipLaunchKernelGGL(k_scatter_rows, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, stream ? (hipStream_t)stream : p->stream,
                     d_src, d_index, (long long)n, words, d_dst);
  HIP_TRY(hipGetLastError());
  return TSIM_OK;
}
