// Declarations shared by the attention kernels: attn.hip (generic: fp32 / bf16, d_head 32 / 64, causal, key-split cross-attention,
// score capture, any key count) and attn_enc.hip (fast path of the encoder self-attention: bf16, d_head 64, <= 256 keys).
#pragma once
#include <float.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"

#ifdef LAKO_EXPERIMENTS
#define ATTN_DBG(a, bits) ((a).dbg_flags & (bits))
#else
#define ATTN_DBG(a, bits) 0
#endif

constexpr int CH_MAX = 256;  // LDS-side rows per chunk
constexpr int OFFS_MAX = 64; // batch rows per workgroup (+1) whose ragged offsets are staged in LDS

template <typename T, int DK> struct AC {
  static constexpr int ES = sizeof(T);
  static constexpr int ROWB = DK * ES + 16;   // padded LDS row
  static constexpr int CPR = DK * ES / 16;    // 16-B chunks per row
  static constexpr int NF = DK * ES / 64;     // fragment steps along d (4 lane groups × 16 B)
  static constexpr int NDB = DK / 16;         // 16-wide output blocks along d
};

struct AttnArgs {
  const char *q, *k, *v, *o, *dout;
  char *out, *dq, *dk, *dv;
  float* stats;        // [Bn, H, Lq, 4] = (row max, 1 / row sum, delta = rowsum(dO∘O) [written by the dQ pass], -)
  const float* rel_bias;
  float* drel;
  const uint8_t* key_mask;
  float* scores_out;
  // ragged (unpadded) sequences: rows of sequence b are rows [off[b], off[b+1]) of ONE packed [rows, H·dk] buffer (the batch
  // stride is ignored); Lq / Lk are then the maxima (grid sizing, statistics and dropout indexing stay in padded coordinates)
  const int32_t *q_off, *k_off;
  const int32_t* order;   // optional [Bn]: the sequence the i-th workgroup / item slot processes (longest first: see lako_attn_fwd_t.order); attn_enc.hip only
  int64_t qsb, qst, ksb, kst, vsb, vst, osb, ost;  // strides in elements
  int R, rel_off, causal, causal_off;
  int Bn, H, Lq, Lk;
  int chunk_rows;     // LDS-side rows per chunk (multiple of 32, <= CH_MAX)
  int blocks_per_wg;  // register-side 16-row blocks per workgroup
  int bn_per_wg;      // dQ pass: batch rows walked by one workgroup (bias-gradient register accumulation)
  int dma_stage;      // attn_enc.hip per-item kernels: 1 = the LDS images arrive by LDS-DMA, everything in flight together (round 4); 0 = register staging
  int dbg_flags;      // timing experiments (LAKO_ATTN_DEBUG): bit 2 = skip the global bias-gradient flush, bit 3 = skip the LDS flush too
  void* dbg_buf;      // (experiments build) stamp buffer of the forward kernel, see FWD_STAMP in attn_enc.hip
  int rq, rk;         // fused backward of attn_enc.hip, padded launches: rows per sequence of the q / k side (batch stride ÷ token stride)
  int grid_x, grid_z; // dQ pass, grid_x > 0: 1-D XCD-grouped grid standing for (grid_x, H, grid_z) — see attn_bwd_kernel
  uint32_t drop_t16, drop_key;   // attention dropout: keep iff 16-bit half >= drop_t16 (0 = off)
  float drop_scale;
  int det;            // LAKO_DETERMINISTIC=1 (det.h): the bias gradient is summed in fixed point and converted once by the last workgroup
};

// i-th item in processing order → sequence id
__device__ __forceinline__ int attn_seq(const AttnArgs& a, int i) {
  return a.order ? __builtin_amdgcn_readfirstlane(a.order[i]) : i;
}

// f(integral_constant<int, 0>) … f(integral_constant<int, N-1>): a compile-time-indexed unrolled loop
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

template <int N> using FragArr = u32x4[N];
template <int N> using AccArr = f32x4[N];

template <typename T> struct Mma16;
template <> struct Mma16<bf16_t> {
  static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0,
                                                   0, 0);
  }
};
template <> struct Mma16<float> {
  static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
    f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
    for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e], bf[e], c, 0, 0, 0);
    return c;
  }
};

// Attention-probability dropout (stateless: forward and backward regenerate the same mask from the indices).
// The 4×4 block (queries 4a … 4a+3) × (keys 4c … 4c+3) of head-row bh = b·H + h shares ONE base hash
//     x = (((bh·QB + a)·KB + c) ^ key) mod 2^32,      h = mix(x, C0, 13)
// and element (q, k) of the block draws 16 bits of the word  W = mix(h, M[q&3][(k&3)>>1], S[q&3][(k&3)>>1]):
// the HIGH half for even k, the LOW half for odd k;  keep iff draw >= round(p·65536).   QB = ⌈Lq/4⌉, KB = ⌈Lk/4⌉ in PADDED
// coordinates (ragged launches index like their padded equivalent).
//     mix(a, c, s) = t ^ (t >> 16),  t = ((a mod 2^24)·c + (a >> s)) mod 2^32
// — v_mad_u32_u24 + a shift + an xor: full-rate instructions only (a 32-bit multiply is quarter rate; the hashes were the
// largest single item of the attention kernels' vector work).  A lane that holds 4 consecutive keys of one query needs the
// base hash and 2 words, a lane that holds 4 consecutive queries of one key the base hash and 4 words.  Measured at p = 0.1
// over 1.9 M draws per key: keep rate 0.8998–0.9001, |correlation| < 2·10⁻³ between neighbours along keys, queries, diagonals
// and head-rows (tests/ref_ops.py carries the same integers; tests/test_kernels_gpu.py::test_attention_dropout_statistics).
constexpr uint32_t DROP_C0 = 0x5BD1E9u;
__host__ __device__ __forceinline__ uint32_t drop_mul(int i, int j) {   // 24-bit odd multipliers of the 8 words of a block, [i·2 + j]
  const uint32_t a = i == 0 ? 0x6C8E95u : i == 1 ? 0x4F1BBDu : i == 2 ? 0x35D2A7u : 0x2545F5u;    // selects, not a table: per-lane i
  const uint32_t b = i == 0 ? 0x1B873Bu : i == 1 ? 0x7A3C6Fu : i == 2 ? 0x59E4C1u : 0x63D9ABu;
  return j ? b : a;
}
__host__ __device__ __forceinline__ uint32_t drop_shift(int i, int j) { return 6u + 2u * (uint32_t)i + 5u * (uint32_t)j; }
__device__ __forceinline__ uint32_t drop_mix(uint32_t a, uint32_t c, uint32_t s) {
  const uint32_t t = __umul24(a, c) + (a >> s);
  return t ^ (t >> 16);
}
__device__ __forceinline__ uint32_t drop_base(uint32_t blk, uint32_t key) { return drop_mix(blk ^ key, DROP_C0, 13u); }
// per-lane constants of a lane whose 4 elements are 4 consecutive KEYS (k = 4c … 4c+3) of query q: words (q&3, 0) and (q&3, 1)
struct DropRow { uint32_t m0, m1, s0, s1; };
__device__ __forceinline__ DropRow drop_row_consts(int q) {
  const int i = q & 3;
  return DropRow{drop_mul(i, 0), drop_mul(i, 1), drop_shift(i, 0), drop_shift(i, 1)};
}
__device__ __forceinline__ void drop_keep_row(uint32_t h, const DropRow& c, uint32_t t_hi, bool (&k)[4]) {
  const uint32_t w0 = drop_mix(h, c.m0, c.s0), w1 = drop_mix(h, c.m1, c.s1);
  k[0] = w0 >= t_hi;
  k[1] = (w0 << 16) >= t_hi;
  k[2] = w1 >= t_hi;
  k[3] = (w1 << 16) >= t_hi;
}
// a lane whose 4 elements are 4 consecutive QUERIES (q = 4a … 4a+3) of key k: words (r, (k&3)>>1), r = 0..3, half k&1
struct DropCol { uint32_t m[4], s[4], sh; };
__device__ __forceinline__ DropCol drop_col_consts(int k) {
  DropCol c;
  const int j = (k & 3) >> 1;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    c.m[r] = drop_mul(r, j);
    c.s[r] = drop_shift(r, j);
  }
  c.sh = (uint32_t)(k & 1) * 16u;
  return c;
}
__device__ __forceinline__ void drop_keep_col(uint32_t h, const DropCol& c, uint32_t t_hi, bool (&k)[4]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) k[r] = (drop_mix(h, c.m[r], c.s[r]) << c.sh) >= t_hi;
}

// fast path of attn_enc.hip (bf16, d_head 64, <= 256 keys, no causal mask, no score capture); attn.hip dispatches
bool lako_attn_enc_supported(const AttnArgs& a, int dtype, int d_head);
int lako_attn_enc_fwd(AttnArgs& a, hipStream_t s);
int lako_attn_enc_bwd(AttnArgs& a, hipStream_t s);
