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

// the same transfer as `global_load_lds_dwordx4` (saddr form: wave-uniform 64-bit base + one 32-bit byte offset per lane): no descriptor, no
// range check — every lane's address must be valid (clamp it), nothing is zero-filled.  Measured cheaper to issue than the buffer form
// inside the GEMM main loops (round 4: 4–14 % per launch, profiles/r04g_gemm_eight_phase.txt).
__device__ __forceinline__ void lds_dma16_g(const void* lds_dst, const void* base, uint32_t voff) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)LDS_PTR(lds_dst));
  const uint64_t b = reinterpret_cast<uint64_t>(base);
  const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)b), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
  const uint64_t sb = ((uint64_t)bhi << 32) | blo;
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(dst), "s"(sb)
      : "memory");
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: set it once per (kernel instantiation, device) —
// a process-wide "done" flag would leave the kernel without its LDS on the second GPU of a process (launch failure).  One call site
// per instantiation (the static array belongs to the call site); 64 devices; relaxed atomics: setting it twice is harmless.
#ifdef __cplusplus
#include <atomic>
#define LAKO_SET_MAX_LDS(kernel_ptr, bytes)                                                                              \
  do {                                                                                                                    \
    static std::atomic<bool> lako_lds_done_[64];                                                                          \
    int lako_dev_ = 0;                                                                                                    \
    (void)hipGetDevice(&lako_dev_);                                                                                       \
    if (lako_dev_ < 0 || lako_dev_ >= 64 || !lako_lds_done_[lako_dev_].load(std::memory_order_relaxed)) {                 \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel_ptr), hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                (int)(bytes));                                                                            \
      if (lako_dev_ >= 0 && lako_dev_ < 64) lako_lds_done_[lako_dev_].store(true, std::memory_order_relaxed);             \
    }                                                                                                                     \
  } while (0)
// the same for kernels whose LDS need varies per launch: `cur` holds, PER DEVICE, the largest size set so far (a static array of the call site)
struct lako_lds_cur_t { std::atomic<int> v[64]; };
inline void lako_raise_max_lds(const void* kern, int bytes, lako_lds_cur_t& cur) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const bool ok = dev >= 0 && dev < 64;
  if (ok && bytes <= cur.v[dev].load(std::memory_order_relaxed)) return;
  (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (ok) cur.v[dev].store(bytes, std::memory_order_relaxed);
}
#endif

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
// stateless dropout RNG: keep(seed, site, idx); forward and backward regenerate the same mask from (seed, site, element index),
// nothing is stored.  lako_hash32 (lowbias32 finalizer) derives the per-site key on the host; the per-element draws are lako_keep4's.
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
// Element-wise dropout draws (round 6: multiply-free, the attention kernels' mix): the four consecutive elements 4q … 4q+3 share
//   x = lo(q) ^ key ^ hi(q)·0x27d4eb2f,   h = mix(x, 0x5BD1E9, 13),   w0 = mix(h, 0x6C8E95, 6),   w1 = mix(h, 0x1B873B, 11),
//   mix(a, c, s) = t ^ (t >> 16),  t = ((a mod 2^24)·c + (a >> s)) mod 2^32        (v_mad_u32_u24 + a shift + an xor),
// with 16-bit draws w0>>16, w0&0xffff, w1>>16, w1&0xffff; an element is KEPT iff its draw >= round(p·65536).
// `thresh` below is that 16-bit threshold shifted left by 16, so that no field has to be extracted:
//   x>>16 >= t  ⟺  x >= t<<16,   x&0xffff >= t  ⟺  x<<16 >= t<<16.
// (Rounds 1-5 used two rounds of a 32-bit multiplicative hash + a third multiply: three quarter-rate multiplies per quad, which the
//  compiler made four.  In the GEMM epilogues of the four-wave kernels — one wave per SIMD, nothing to hide behind — that was a third of
//  a dropout epilogue's 16 000 cycles per tile: profiles/r06h_nt4_epilogue_stamps.txt.  Keep rate and neighbour correlations of this
//  recipe: tests/test_engine_cpu.py::test_elementwise_dropout_recipe_statistics.)  tests/ref_ops.py carries the same integers.
__host__ __device__ __forceinline__ uint32_t lako_mix24(uint32_t a, uint32_t c, uint32_t s) {
  const uint32_t t = (a & 0xFFFFFFu) * c + (a >> s);      // c < 2^24: a 24-bit multiply-add on the device
  return t ^ (t >> 16);
}
__host__ __device__ __forceinline__ void lako_keep4(uint32_t key, uint64_t quad, uint32_t thresh, bool (&k)[4]) {
  const uint32_t x = (uint32_t)quad ^ key ^ ((uint32_t)(quad >> 32) * 0x27d4eb2fU);
  const uint32_t h = lako_mix24(x, 0x5BD1E9u, 13u);
  const uint32_t w0 = lako_mix24(h, 0x6C8E95u, 6u), w1 = lako_mix24(h, 0x1B873Bu, 11u);
  k[0] = w0 >= thresh;
  k[1] = (w0 << 16) >= thresh;
  k[2] = w1 >= thresh;
  k[3] = (w1 << 16) >= thresh;
}
// single element idx (callers that hold one element per lane)
__host__ __device__ __forceinline__ bool lako_keep(uint32_t key, uint64_t idx, uint32_t thresh) {
  bool k[4];
  lako_keep4(key, idx >> 2, thresh, k);
  return k[idx & 3];
}
static inline uint32_t lako_drop_thresh(float p) {   // 0 = dropout off
  double t = (double)p * 65536.0 + 0.5;
  if (t < 1) t = 1;
  if (t > 65535.0) t = 65535.0;
  return (uint32_t)t << 16;
}

// ---------------------------------------------------------------------------------------------
// MX (OCP microscaling) quantiser of 8 consecutive k held by one lane: the 4 lanes l, l^1, l^2, l^3 of a 32-element block agree on amax
// by two shuffles; scale exponent e = exponent(amax) − 8 clamped to [0, 254] (8 = the exponent of e4m3's largest normal, 448 = 1.75·2^8),
// elements sat_e4m3(x · 2^(127 − e)).  ±Inf saturates to ±448·scale; a NaN anywhere in the block makes the block's scale the E8M0 NaN
// (0xFF) and the element 0x7F.  Returns the scale byte; w[0..1] = the 8 e4m3 bytes.  (mx_quantize_kernel and the fused RMSNorm forward.)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint8_t mx_quant8(const float (&f)[8], uint32_t (&w)[2]) {
  float amax = 0.f;
  int has_nan = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    amax = fmaxf(amax, fabsf(f[e]));            // fmaxf drops a NaN: tracked separately
    has_nan |= f[e] != f[e];
  }
  amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
  amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
  has_nan |= __shfl_xor(has_nan, 1, 64);
  has_nan |= __shfl_xor(has_nan, 2, 64);
  int ex = (int)((__builtin_bit_cast(uint32_t, amax) >> 23) & 0xff) - 8;
  ex = min(max(ex, 0), 254);
  const float inv = __builtin_bit_cast(float, (uint32_t)(254 - ex) << 23);        // 2^(127 − ex): exact
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float t[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = fminf(fmaxf(f[4 * h + e] * inv, -448.f), 448.f);
    int p = __builtin_amdgcn_cvt_pk_fp8_f32(t[0], t[1], 0, false);
    p = __builtin_amdgcn_cvt_pk_fp8_f32(t[2], t[3], p, true);
    uint32_t pw = (uint32_t)p;
#pragma unroll
    for (int e = 0; e < 4; ++e)                           // fminf / fmaxf turned a NaN into ∓448: put the e4m3 NaN back
      if (f[4 * h + e] != f[4 * h + e]) pw = (pw & ~(0xffu << (8 * e))) | (0x7fu << (8 * e));
    w[h] = pw;
  }
  return has_nan ? (uint8_t)0xff : (uint8_t)ex;
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
