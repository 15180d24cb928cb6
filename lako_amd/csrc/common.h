// Shared device/host helpers for the lako_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/lako_hip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// LDS-DMA (buffer_load_dwordx4 … lds: 64 lanes × 16 B straight into LDS at lds_dst + 16·lane) issued as inline asm.
// hipcc's waitcnt pass puts `s_waitcnt vmcnt(0)` in front of every LDS read it cannot prove disjoint from the
// destination of a *builtin* LDS-DMA (seen before each ds_read_b64_tr_b16 of the TN kernel), which makes the prefetch
// synchronous.  The asm form is outside that bookkeeping: the caller owns the wait (s_waitcnt vmcnt) and the barrier.
// lds_dst and the descriptor must be wave-uniform.  M0 is compiler-reserved: saved and restored in the statement.
typedef uint32_t lako_u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ lako_u32x4_t lds_dma_rsrc(const void* base, uint32_t num_bytes) {
  const uint64_t b = reinterpret_cast<uint64_t>(base);
  lako_u32x4_t r;
  r[0] = __builtin_amdgcn_readfirstlane((uint32_t)b);
  r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32) & 0xFFFFu);
  r[2] = __builtin_amdgcn_readfirstlane(num_bytes);
  r[3] = 0x00020000u;
  return r;
}
__device__ __forceinline__ void lds_dma16(const void* lds_dst, lako_u32x4_t rsrc, uint32_t voff) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)LDS_PTR(lds_dst));
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(dst), "s"(rsrc)
      : "memory");
}

// ---------------------------------------------------------------------------------------------
// error plumbing (never throws, never exits; see include/lako_hip.h)
// ---------------------------------------------------------------------------------------------
void lako_set_error(const char* fmt, ...);

#define LAKO_CHECK_ARG(cond, ...)            \
  do {                                       \
    if (!(cond)) {                           \
      lako_set_error(__VA_ARGS__);           \
      return LAKO_E_BADARG;                  \
    }                                        \
  } while (0)

#define LAKO_CHECK_ALIGN(ptr, a)                                        \
  do {                                                                  \
    if ((reinterpret_cast<uintptr_t>(ptr) % (a)) != 0) {                \
      lako_set_error("%s: pointer %s not %d-byte aligned", __func__, #ptr, (int)(a)); \
      return LAKO_E_ALIGN;                                              \
    }                                                                   \
  } while (0)

#define LAKO_LAUNCH_CHECK()                                             \
  do {                                                                  \
    hipError_t e_ = hipGetLastError();                                  \
    if (e_ != hipSuccess) {                                             \
      lako_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e_)); \
      return LAKO_E_LAUNCH;                                             \
    }                                                                   \
  } while (0)

// ---------------------------------------------------------------------------------------------
// element helpers
// ---------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }

// 4 consecutive elements <-> 4 floats
__device__ __forceinline__ f32x4 load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 load4(const bf16_t* p) {
  bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ void store4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void store4(bf16_t* p, f32x4 v) {
  bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
  *reinterpret_cast<bf16x4*>(p) = o;
}

// ---------------------------------------------------------------------------------------------
// stateless dropout RNG: keep(seed, site, idx).  One 32-bit hash per element (lowbias32 finalizer);
// forward and backward regenerate the same mask from (seed, site, element index), nothing is stored.
// tests/ref_ops.py carries the identical integer recipe in numpy.
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t lako_hash32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
__host__ __device__ __forceinline__ uint32_t lako_drop_key(uint32_t seed, uint32_t site) {
  return lako_hash32(seed * 0x9E3779B9U + site * 0x85EBCA6BU + 0x1234567U);
}
// returns true when the element is KEPT; thresh = round(p * 2^32)
__host__ __device__ __forceinline__ bool lako_keep(uint32_t key, uint64_t idx, uint32_t thresh) {
  uint32_t lo = (uint32_t)idx, hi = (uint32_t)(idx >> 32);
  uint32_t h = lako_hash32(lo ^ key ^ (hi * 0x27d4eb2fU));
  return h >= thresh;
}
static inline uint32_t lako_drop_thresh(float p) {
  double t = (double)p * 4294967296.0;
  if (t < 0) t = 0;
  if (t > 4294967295.0) t = 4294967295.0;
  return (uint32_t)t;
}

// ---------------------------------------------------------------------------------------------
// wave helpers (wave = 64 lanes)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
