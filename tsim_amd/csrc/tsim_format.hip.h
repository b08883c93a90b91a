// tsim_format.hip.h - data-format kernels either side of the path (gfx950, HBM-bound).
//
// The reference hands the sampling path one BYTE per bit (uint8 [B, num_f] in, bool [B, n_out] out:
// src/tsim/sampler.py:398,415); the engine works on packed rows.  These two kernels convert on the
// device.  They are the rows of the path where the HBM roofline is the right yardstick
// (SURVEY.md section 8(d)): per shot they move nbits + 8*ceil(nbits/64) bytes and do ~1-2 VALU ops
// per byte.
//   * pack:   one thread per 32-bit OUTPUT word = 32 input bytes, fetched as aligned dwords and
//             re-aligned with v_alignbyte (rows are not dword aligned in general); "byte != 0"
//             (astype(bool)) by the carry trick, four flags -> one nibble with one multiply.
//   * unpack: one thread per OUTPUT dword (4 result bytes, flat indexing so that stores are always
//             aligned and coalesced); the row index comes from a multiply-shift division, the four
//             bits from a funnel shift, nibble -> 4 bytes with one multiply.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tsimk {

// 4 bytes -> 4 flag bits (bit k = byte k != 0)
__device__ __forceinline__ uint32_t nz_nibble(uint32_t v) {
  const uint32_t m = ((((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) & 0x80808080u) >> 7;  // 0/1 per byte
  return ((m * 0x00204081u) >> 21) & 0xFu;
}

// uint8 [B, nbits] -> uint32 [B, 2*WQ] (== uint64 [B, WQ], little endian)
__global__ void __launch_bounds__(256) k_pack_bits(const uint8_t *__restrict__ in, uint32_t *__restrict__ out32,
                                                    long long B, int nbits, int n32, long long in_dwords) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * n32) return;
  const long long row = idx / n32;
  const int j = (int)(idx - row * n32);
  const int count = nbits - 32 * j;  // input bytes feeding this word
  if (count <= 0) { out32[idx] = 0u; return; }
  const long long base = row * (long long)nbits + 32 * j;
  const long long a = base >> 2;           // first aligned dword
  const uint32_t sh = (uint32_t)(base & 3);  // byte offset inside it
  const uint32_t *src = reinterpret_cast<const uint32_t *>(in);
  const int ndw = (min(count, 32) + (int)sh + 3) >> 2;  // aligned dwords covering the bytes
  uint32_t d[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) d[k] = (k < ndw && a + k < in_dwords) ? src[a + k] : 0u;
  uint32_t w = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const uint32_t v = __builtin_amdgcn_alignbyte(d[k + 1], d[k], sh);
    w |= nz_nibble(v) << (4 * k);
  }
  if (count < 32) w &= (1u << count) - 1u;
  out32[idx] = w;
}

// uint32 [B, n32] packed rows -> uint8 [B, nbits], written as aligned dwords.
// 16-byte-aligned rows (nbits % 16 == 0): one thread per 64-bit output word, four 16-byte loads.
__global__ void __launch_bounds__(256) k_pack_bits_a16(const uint4 *__restrict__ in, uint64_t *__restrict__ out,
                                                        long long B, int nbits, int WQ) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * WQ) return;
  const long long row = idx / WQ;
  const int j = (int)(idx - row * WQ);
  const int nv = min(4, (nbits - 64 * j) >> 4);  // 16-byte vectors feeding this word
  const uint4 *src = in + ((row * (long long)nbits + 64 * j) >> 4);
  uint64_t w = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (k < nv) {
      const uint4 v = src[k];
      const uint32_t h = nz_nibble(v.x) | (nz_nibble(v.y) << 4) | (nz_nibble(v.z) << 8) | (nz_nibble(v.w) << 12);
      w |= (uint64_t)h << (16 * k);
    }
  }
  out[idx] = w;
}

// nbits % 16 == 0: one thread per 16 output bytes (one 16-byte store).
__global__ void __launch_bounds__(256) k_unpack_bits_a16(const uint32_t *__restrict__ in32, uint4 *__restrict__ out,
                                                          long long total16, int nbits, int n32,
                                                          unsigned long long magic16) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total16) return;
  // row = (16 t) / nbits = t / (nbits / 16)
  unsigned long long row = (unsigned long long)(((unsigned __int128)(unsigned long long)t * magic16) >> 40);
  const unsigned long long per = (unsigned long long)(nbits >> 4);
  if (row * per > (unsigned long long)t) --row;
  const uint32_t bit = (uint32_t)(t - (long long)(row * per)) << 4;  // multiple of 16: inside one 32-bit word
  const uint32_t x = (in32[(long long)row * n32 + (bit >> 5)] >> (bit & 31u)) & 0xFFFFu;
  uint4 v;
  v.x = ((x & 0xFu) * 0x00204081u) & 0x01010101u;
  v.y = (((x >> 4) & 0xFu) * 0x00204081u) & 0x01010101u;
  v.z = (((x >> 8) & 0xFu) * 0x00204081u) & 0x01010101u;
  v.w = (((x >> 12) & 0xFu) * 0x00204081u) & 0x01010101u;
  out[t] = v;
}

// magic = ceil(2^40 / nbits): row = (i * magic) >> 40 with one correction step (i < 2^32).
__global__ void __launch_bounds__(256) k_unpack_bits(const uint32_t *__restrict__ in32, uint32_t *__restrict__ out32,
                                                      uint8_t *__restrict__ out8, long long total_bytes, int nbits,
                                                      int n32, unsigned long long magic, long long in_words) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long i0 = 4 * t;
  if (i0 >= total_bytes) return;
  unsigned long long row = (unsigned long long)(((unsigned __int128)(unsigned long long)i0 * magic) >> 40);
  if (row * (unsigned long long)nbits > (unsigned long long)i0) --row;
  const uint32_t bit = (uint32_t)(i0 - (long long)row * nbits);
  if (bit + 3 < (uint32_t)nbits && i0 + 3 < total_bytes) {
    // the four bits live in one row: funnel-shift them out of two consecutive 32-bit words
    const long long wi = (long long)row * n32 + (bit >> 5);
    const uint32_t lo = in32[wi];
    const uint32_t hi = (wi + 1 < in_words) ? in32[wi + 1] : 0u;
    const uint32_t x = __builtin_amdgcn_alignbit(hi, lo, bit & 31u) & 0xFu;
    out32[t] = (x * 0x00204081u) & 0x01010101u;
    return;
  }
  // row boundary (or the tail of the array) inside these four bytes: byte by byte
  long long r = (long long)row;
  uint32_t b = bit;
  for (int k = 0; k < 4 && i0 + k < total_bytes; ++k) {
    out8[i0 + k] = (uint8_t)((in32[r * n32 + (b >> 5)] >> (b & 31u)) & 1u);
    if (++b == (uint32_t)nbits) { b = 0; ++r; }
  }
}


// uint64[B, WO] padded rows -> ceil(nbits/8)-byte rows (== np.packbits(bits, axis=1,
// bitorder="little"), the reference's bit_packed=True format, sampler.py:665-669): what the RCCL
// gather and a packed D2H actually have to move.  One thread = 4 rows = rb whole output words.
__global__ void __launch_bounds__(256) k_compact_rows(const uint64_t *__restrict__ in, uint8_t *__restrict__ out,
                                                      long long B, int WO, int rb, uint32_t tail_mask) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long r0 = 4 * t;
  if (r0 >= B) return;
  const uint64_t *src = in + r0 * WO;
  if (r0 + 4 <= B) {
    uint32_t *dst = reinterpret_cast<uint32_t *>(out + r0 * rb);  // 4 * rb bytes: word aligned
    int row = 0, off = 0;
    for (int j = 0; j < rb; ++j) {
      uint32_t w = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        uint32_t byte = (uint32_t)(src[row * WO + (off >> 3)] >> (8 * (off & 7))) & 255u;
        if (off == rb - 1) byte &= tail_mask;  // columns beyond nbits do not belong to the row
        w |= byte << (8 * k);
        if (++off == rb) { off = 0; ++row; }
      }
      dst[j] = w;
    }
  } else {
    for (long long r = r0; r < B; ++r)
      for (int o = 0; o < rb; ++o)
        out[r * rb + o] = (uint8_t)((uint32_t)(in[r * WO + (o >> 3)] >> (8 * (o & 7))) & (o == rb - 1 ? tail_mask : 255u));
  }
}

}  // namespace tsimk

namespace tsimk {

// Row gather / scatter by index (W 64-bit words per row), the data movement of host-noise post-selection
// (reference src/tsim/sampler.py:466-508: survivors are compacted into dense batches, padded with the first
// one, and their result rows written back to the shots they came from):
//   gather : dst[i] = src[index[i < n_valid ? i : 0]]   for i < n_total
//   scatter: dst[index[i]] = src[i]                      for i < n
__global__ void __launch_bounds__(256) k_gather_rows(const uint64_t *__restrict__ src, const uint32_t *__restrict__ index,
                                                      long long n_valid, long long n_total, int W,
                                                      uint64_t *__restrict__ dst) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_total * W) return;
  const long long i = t / W;
  const int w = (int)(t - i * W);
  const uint32_t r = index[i < n_valid ? i : 0];
  dst[t] = src[(long long)r * W + w];
}

__global__ void __launch_bounds__(256) k_scatter_rows(const uint64_t *__restrict__ src, const uint32_t *__restrict__ index,
                                                       long long n, int W, uint64_t *__restrict__ dst) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * W) return;
  const long long i = t / W;
  const int w = (int)(t - i * W);
  dst[(long long)index[i] * W + w] = src[t];
}

}  // namespace tsimk
