// Swizzled LDS images of 64-element bf16 rows and the per-lane addressing of their two MFMA read patterns (shared by the
// encoder self-attention fast path, attn_enc.hip, and the encoder-space cross-attention, xattn.hip).
// An image row is 128 bytes (64 bf16), unpadded; its 32-byte slots are XOR-swizzled by (row >> 1) & 3, which is conflict-free
// for BOTH ds_read_b128 row fragments (operands contracted along the row) and ds_read_b64_tr_b16 transposed fragments (operands
// contracted along the image rows), with per-lane offsets that are loop constants.
#pragma once
#include "common.h"

namespace {

constexpr int EROW = 128;                 // bytes per image row: 64 bf16, unpadded

// 16-byte chunk c (0..7) of image row r lives at chunk position ((c >> 1) ^ ((r >> 1) & 3)) << 1 | (c & 1)
__device__ __forceinline__ int eswz(int c, int r) { return ((((c >> 1) ^ ((r >> 1) & 3)) << 1) | (c & 1)); }

// Stage rows [0, nrows) of a strided [L, 64] bf16 tensor into a swizzled image; rows >= L are zero-filled.
template <int NT = 256>
__device__ __forceinline__ void estage(char* img, const char* base, int64_t stride_b, int nrows, int L) {
  const int total = nrows * 8;
  for (int idx0 = threadIdx.x; idx0 < total; idx0 += NT * 8) {
    u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = idx0 + u * NT;
      const int row = idx >> 3, c = idx & 7;
      v[u] = u32x4{0u, 0u, 0u, 0u};
      if (idx < total && row < L) v[u] = *reinterpret_cast<const u32x4*>(base + (int64_t)row * stride_b + c * 16);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = idx0 + u * NT;
      const int row = idx >> 3, c = idx & 7;
      if (idx < total) *reinterpret_cast<u32x4*>(img + row * EROW + eswz(c, row) * 16) = v[u];
    }
  }
}

__device__ __forceinline__ f32x4 emma(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}


struct ELane {          // loop-invariant per-lane addressing
  uint32_t roff[2];     // row-fragment reads (ds_read_b128)
  uint32_t toff[4];     // transposed reads (ds_read_b64_tr_b16)
};
__device__ __forceinline__ ELane elane(int lane) {
  ELane e;
  const int l15 = lane & 15, g = lane >> 4, qq = l15 >> 2, pp = l15 & 3;
#pragma unroll
  for (int i = 0; i < 2; ++i) e.roff[i] = (uint32_t)(l15 * EROW + eswz(i * 4 + g, l15) * 16);
  const int tr = 4 * g + qq;                      // row inside the 16-row block
#pragma unroll
  for (int db = 0; db < 4; ++db) e.toff[db] = (uint32_t)(tr * EROW + ((db ^ ((tr >> 1) & 3)) << 5) + pp * 8);
  return e;
}

}  // namespace
