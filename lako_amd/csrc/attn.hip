// T5 attention for the FiD reader (HF5:144-173,281-369; same math as the reference's own
// cross_attention_forward, src/model.py:286-349):  P = softmax_fp32(Q·Kᵀ + rel_bias + mask),
// UNSCALED scores, dropout on P, O = P·V.  Forward + flash-style backward (P recomputed from the
// saved row max / 1/rowsum, nothing of size Lq×Lk is ever stored).
//
// One workgroup (4 waves) per (batch row, head[, row-block group]).  One operand side lives in LDS
// (16-B padded rows, staged once per workgroup), the other side is held per wave in registers as MFMA
// fragments loaded straight from HBM; each wave owns 16-row blocks of the register side.
//   forward, dQ pass : keys (K, V) in LDS, a wave owns 16 queries;   tile = Sᵀ[key][query]
//   dK/dV pass       : queries (Q, dO) in LDS, a wave owns 16 keys;  tile = S[query][key]
// In both cases the score tile comes out of the MFMA with the LDS-side index in the accumulator
// registers and the register-side index on the lane, so (a) softmax row statistics of the forward are
// 2 cross-lane shuffles, and (b) the tile is directly the B operand of the following product that
// contracts over the LDS-side index (P·V, dSᵀ·K, Pᵀ·dO, dSᵀ·Q) — its A operand is a transposed read of the
// LDS image (ds_read_b64_tr_b16 for bf16, plain dwords for fp32).  No P/dS round trip through LDS.
//
// The kernels are VALU-bound, not MFMA-bound (≈50 score elements per lane per 16-query block), so the
// element-wise path is kept branch-free and short: the relative-position bias and the additive key
// mask (0 / −FLT_MAX / −inf, exactly HF's "scores + finfo.min") are folded into the MFMA accumulator
// INIT, the causal mask is one select, and dropout draws two 16-bit keep decisions from one 32-bit hash.
// bf16: v_mfma_f32_16x16x32_bf16; fp32 (parity mode): v_mfma_f32_16x16x4_f32.
#include "attn_shared.h"
#include "det.h"

namespace {

LAKO_DET_STORAGE      // LAKO_DETERMINISTIC=1: see det.h

// Stage rows [r0, r0 + nrows) of a strided [L, ·] tensor into a padded LDS image; rows >= L are
// zero-filled (so padded keys/queries contribute exact zeros).  8 independent loads in flight per thread.
template <typename T, int DK>
__device__ __forceinline__ void stage_image(char* img, const char* base, int64_t stride, int r0, int nrows, int L) {
  using C = AC<T, DK>;
  const int total = nrows * C::CPR;
  for (int idx0 = threadIdx.x; idx0 < total; idx0 += 256 * 8) {
    u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = idx0 + u * 256;
      const int row = idx / C::CPR, c = idx % C::CPR;
      v[u] = u32x4{0u, 0u, 0u, 0u};
      if (idx < total && r0 + row < L)
        v[u] = *reinterpret_cast<const u32x4*>(base + (int64_t)(r0 + row) * stride * C::ES + c * 16);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = idx0 + u * 256;
      const int row = idx / C::CPR, c = idx % C::CPR;
      if (idx < total) *reinterpret_cast<u32x4*>(img + row * C::ROWB + c * 16) = v[u];
    }
  }
}

template <typename T, int DK>
__device__ __forceinline__ void load_reg_frags(FragArr<AC<T, DK>::NF>& f, const char* base, int64_t stride, int row,
                                               int L, int lane) {
  using C = AC<T, DK>;
  const int g = lane >> 4;
#pragma unroll
  for (int i = 0; i < C::NF; ++i) {
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row < L) v = *reinterpret_cast<const u32x4*>(base + (int64_t)row * stride * C::ES + (i * 4 + g) * 16);
    f[i] = v;
  }
}

// tile[LDS row = row0 + 4g + r][reg-side = lane & 15] = init + Σ_d X[row0 + ·][d] · Y[·][d]
template <typename T, int DK>
__device__ __forceinline__ f32x4 score_tile(const char* img, int row0, const FragArr<AC<T, DK>::NF>& yf, int lane,
                                            f32x4 acc) {
  using C = AC<T, DK>;
  const int g = lane >> 4;
  const char* rp = img + (row0 + (lane & 15)) * C::ROWB + g * 16;
#pragma unroll
  for (int i = 0; i < C::NF; ++i) {
    u32x4 xf = *reinterpret_cast<const u32x4*>(rp + i * 64);
    acc = Mma16<T>::run(xf, yf[i], acc);
  }
  return acc;
}

// acc[db][r] (d = db*16 + 4g + r, reg-side = lane & 15) += Σ_{rows of the tile pair} X[row][d] · w[row][lane&15]
// where w0 / w1 are the two 16-row tiles starting at LDS rows row0 and row0 + 16.
template <typename T, int DK>
__device__ __forceinline__ void pv_accumulate(AccArr<AC<T, DK>::NDB>& acc, f32x4 w0, f32x4 w1, const char* img,
                                              int row0, int lane) {
  using C = AC<T, DK>;
  const int g = lane >> 4;
  if constexpr (sizeof(T) == 2) {
    bf16x8 bfrag;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      bfrag[e] = (bf16_t)w0[e];
      bfrag[4 + e] = (bf16_t)w1[e];
    }
    const int q = (lane & 15) >> 2, p = lane & 3;
    const char* a0 = img + (row0 + 4 * g + q) * C::ROWB + p * 8;
#pragma unroll
    for (int db = 0; db < C::NDB; ++db) {
      s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + db * 32));
      s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4*)(a0 + db * 32 + 16 * C::ROWB));
      u32x2 u0 = __builtin_bit_cast(u32x2, t0), u1 = __builtin_bit_cast(u32x2, t1);
      u32x4 afrag = {u0[0], u0[1], u1[0], u1[1]};
      acc[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, afrag), bfrag, acc[db], 0, 0, 0);
    }
  } else {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const f32x4 w = t == 0 ? w0 : w1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const char* ap = img + (row0 + t * 16 + 4 * g + r) * C::ROWB + (lane & 15) * 4;
#pragma unroll
        for (int db = 0; db < C::NDB; ++db) {
          float av = *reinterpret_cast<const float*>(ap + db * 64);
          acc[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, w[r], acc[db], 0, 0, 0);
        }
      }
    }
  }
}

__device__ __forceinline__ float group_max(float v) {  // across the 4 lane groups (same lane & 15)
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}
__device__ __forceinline__ int clampi(int x, int lo, int hi) { return min(max(x, lo), hi); }

// additive key mask of LDS-side chunk rows: 0 attend, −FLT_MAX padding key (HF: scores + finfo.min), −inf beyond Lk
// pair_any[tp] (tp < 8) = 1 when tile pair tp (keys kc0 + 32·tp … + 31) holds at least one attendable key.  A masked
// key's probability is EXACTLY 0 whenever its row has one attendable key (exp(−FLT_MAX − m) = 0), so tile pairs
// without any can be skipped; a sequence with no attendable key at all keeps every tile (HF: uniform attention).
__device__ __forceinline__ void stage_key_add(float* kadd, const uint8_t* key_mask, int b, int kc0, int nk, int Lk,
                                              int mask_ld, int* pair_any) {
  static_assert(CH_MAX <= 256, "one key per thread");
  const int i = threadIdx.x, j = kc0 + i;
  float f = 0.f;
  if (j >= Lk) f = -INFINITY;
  else if (key_mask && !key_mask[(int64_t)b * mask_ld + j]) f = -FLT_MAX;
  if (i < nk) kadd[i] = f;
  const uint64_t bal = __ballot(i < nk && f == 0.f);
  if ((threadIdx.x & 63) == 0) {
    pair_any[2 * (threadIdx.x >> 6)] = (uint32_t)bal != 0u;
    pair_any[2 * (threadIdx.x >> 6) + 1] = (uint32_t)(bal >> 32) != 0u;
  }
}
__device__ __forceinline__ int pair_bits(const int* pair_any) {
  int bits = 0;
#pragma unroll
  for (int tp = 0; tp < 8; ++tp) bits |= (pair_any[tp] != 0) << tp;
  return __builtin_amdgcn_readfirstlane(bits);
}

template <typename T, int DK> __host__ __device__ constexpr int img_bytes(int ch) { return ch * AC<T, DK>::ROWB; }

// One staged key chunk for one wave's 16 queries: scores → online softmax → P·V for NP tile pairs, with NO runtime
// branch inside (the caller dispatches on the pair count): per-tile `if`s cut the unrolled code into basic blocks
// of one LDS-read → wait → MFMA chain each, nothing overlapped.  Pair p is tile pair tp = tp0 + p·tstride of the
// chunk (key-split mode: tp0 = wave, tstride = 4).
struct FwdRow {          // per-lane state of one query row across chunks
  float m, lsum;
};
template <typename T, int DK, bool CAPTURE, int NP>
__device__ __forceinline__ void fwd_chunk(const AttnArgs& a, const char* Kimg, const char* Vimg, const float* kadd,
                                          const float* bias_l, const FragArr<AC<T, DK>::NF>& qf, FwdRow& st,
                                          AccArr<AC<T, DK>::NDB>& oacc, int b, int h, int qi, int kc0, int tp0, int tstride,
                                          uint32_t prow, uint32_t t_hi, int jmax, int lane) {
  using C = AC<T, DK>;
  const int g = lane >> 4;
  const bool has_bias = bias_l != nullptr;
  const DropRow drc = drop_row_consts(qi);
  f32x4 s[2 * NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int row0 = (tp0 + p * tstride) * 32 + u * 16;     // first key of the tile inside the chunk
      f32x4 init = *reinterpret_cast<const f32x4*>(kadd + row0 + 4 * g);
      if (has_bias) {
        const int bi0 = kc0 + row0 + 4 * g - qi + a.rel_off;
#pragma unroll
        for (int r = 0; r < 4; ++r) init[r] += bias_l[clampi(bi0 + r, 0, a.R - 1)];
      }
      s[2 * p + u] = score_tile<T, DK>(Kimg, row0, qf, lane, init);
    }
  }
  if (a.causal) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int j0 = kc0 + (tp0 + p * tstride) * 32 + u * 16 + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) s[2 * p + u][r] = (j0 + r > jmax) ? -FLT_MAX : s[2 * p + u][r];
      }
  }
  if constexpr (CAPTURE) {   // score capture at decode step 0: raw scores, masked keys → 0
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int j0 = kc0 + (tp0 + p * tstride) * 32 + u * 16 + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (qi < a.Lq && j0 + r < a.Lk)
            a.scores_out[(((int64_t)b * a.H + h) * a.Lq + qi) * a.Lk + j0 + r] =
                s[2 * p + u][r] <= -FLT_MAX ? 0.f : s[2 * p + u][r];
      }
  }
  float cmax = -INFINITY;
#pragma unroll
  for (int t = 0; t < 2 * NP; ++t) cmax = fmaxf(fmaxf(cmax, fmaxf(s[t][0], s[t][1])), fmaxf(s[t][2], s[t][3]));
  cmax = group_max(cmax);
  if (cmax == -INFINITY) return;           // every key of these tiles lies beyond Lk (wave-uniform)
  const float m_new = fmaxf(st.m, cmax);
  const float alpha = __expf(st.m - m_new);  // m = -inf on the first chunk → 0
  float psum = 0.f;
#pragma unroll
  for (int t = 0; t < 2 * NP; ++t) {
    f32x4 pr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      pr[r] = __expf(s[t][r] - m_new);
      psum += pr[r];
    }
    if (a.drop_t16) {
      const int row0 = (tp0 + (t >> 1) * tstride) * 32 + (t & 1) * 16;
      bool kp[4];
      drop_keep_row(drop_base(prow + (uint32_t)((kc0 + row0 + 4 * g) >> 2), a.drop_key), drc, t_hi, kp);
#pragma unroll
      for (int r = 0; r < 4; ++r) pr[r] = kp[r] ? pr[r] * a.drop_scale : 0.f;
    }
    s[t] = pr;
  }
  psum = group_sum(psum);
  st.lsum = st.lsum * alpha + psum;
  st.m = m_new;
#pragma unroll
  for (int db = 0; db < C::NDB; ++db) oacc[db] *= alpha;
#pragma unroll
  for (int p = 0; p < NP; ++p) pv_accumulate<T, DK>(oacc, s[2 * p], s[2 * p + 1], Vimg, (tp0 + p * tstride) * 32, lane);
}

// =============================================================================================
// forward
// =============================================================================================
template <typename T, int DK, bool CAPTURE>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a) {
  using C = AC<T, DK>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int CH = a.chunk_rows;
  char* Kimg = smem;
  char* Vimg = smem + img_bytes<T, DK>(CH);
  float* kadd = reinterpret_cast<float*>(smem + 2 * img_bytes<T, DK>(CH));  // [CH]
  float* bias_l = kadd + CH;                                                // [R]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  const int b = blockIdx.z, h = blockIdx.y;
  const int nqb = (a.Lq + 15) >> 4;                        // padded maximum (grid, key-split decision)
  // this sequence's own lengths and row offsets (ragged mode) — bounds use Lq / Lk, index hashing a.Lq / a.Lk
  // (readfirstlane: values loaded from memory are per-lane to the compiler; as loop bounds they would turn every loop
  //  below into divergent control flow)
  const int q0 = a.q_off ? __builtin_amdgcn_readfirstlane(a.q_off[b]) : 0;
  const int k0 = a.k_off ? __builtin_amdgcn_readfirstlane(a.k_off[b]) : 0;
  const int Lq = a.q_off ? __builtin_amdgcn_readfirstlane(a.q_off[b + 1]) - q0 : a.Lq;
  const int Lk = a.k_off ? __builtin_amdgcn_readfirstlane(a.k_off[b + 1]) - k0 : a.Lk;
  const int qb_begin = blockIdx.x * a.blocks_per_wg;
  const int qb_end = min((Lq + 15) >> 4, qb_begin + a.blocks_per_wg);
  const int nchunks = (Lk + CH - 1) / CH;
  const int64_t hoff = (int64_t)h * DK;
  const char* qbase = a.q + ((a.q_off ? (int64_t)q0 * a.qst : (int64_t)b * a.qsb) + hoff) * C::ES;
  const char* kbase = a.k + ((a.k_off ? (int64_t)k0 * a.kst : (int64_t)b * a.ksb) + hoff) * C::ES;
  const char* vbase = a.v + ((a.k_off ? (int64_t)k0 * a.vst : (int64_t)b * a.vsb) + hoff) * C::ES;
  char* obase = a.out + ((a.q_off ? (int64_t)q0 * a.ost : (int64_t)b * a.osb) + hoff) * C::ES;
  const bool has_bias = a.rel_bias != nullptr;
  if (has_bias)
    for (int i = threadIdx.x; i < a.R; i += 256) bias_l[i] = a.rel_bias[(int64_t)h * a.R + i];
  const int hq = (a.Lk + 3) >> 2, hqq = (a.Lq + 3) >> 2;   // dropout blocks per score row / per score column (padded coordinates)
  const uint32_t t_hi = a.drop_t16 << 16;

  // Key-split mode: a workgroup that owns a single 16-query block (cross-attention, decode steps) lets its
  // 4 waves share that block and split every key chunk between them (tile pair tp → wave tp & 3); the
  // partial (max, sum, O) of the 4 waves are merged through LDS at the end.
  const bool ksplit = nqb == 1 && a.Lk > 64;   // host reserves the merge scratch only when Lq <= 16
  int* pair_any = reinterpret_cast<int*>(bias_l + ((a.R + 3) & ~3));   // [8]
  float* mrg = reinterpret_cast<float*>(pair_any + 8);                 // [4 waves][16 queries][DK + 2] (ksplit only)
  int okbits = 0xff;   // tile pairs of the staged chunk worth computing (single-chunk sequences only)

  // The Q fragments of round r+1 are requested before round r is computed: a round is only ≈7 tile pairs long, a
  // load → wait → compute chain per round would expose one global round trip (≈2 µs) per ≈2 µs of work.
  u32x4 qf_next[C::NF];
  auto request_q = [&](int qb0n) {
    const int qbn = ksplit ? qb0n : qb0n + wave;
    load_reg_frags<T, DK>(qf_next, qbase, a.qst, (qb0n < qb_end && qbn < qb_end) ? qbn * 16 + l15 : Lq, Lq, lane);
  };
  request_q(qb_begin);
  for (int qb0 = qb_begin; qb0 < qb_end; qb0 += 4) {
    const int qb = ksplit ? qb0 : qb0 + wave;
    const bool active = qb < qb_end;
    const int qi = qb * 16 + l15;  // this lane's query
    u32x4 qf[C::NF];
#pragma unroll
    for (int i = 0; i < C::NF; ++i) qf[i] = qf_next[i];
    request_q(qb0 + 4);
    float m = -INFINITY, lsum = 0.f;
    f32x4 oacc[C::NDB];
#pragma unroll
    for (int db = 0; db < C::NDB; ++db) oacc[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int jmax = a.causal ? qi + a.causal_off : 0x7fffffff;       // keys j > jmax are causally masked
    const uint32_t prow = (uint32_t)((b * a.H + h) * hqq + (qi >> 2)) * (uint32_t)hq;  // dropout block index of (qi, key 0)

    for (int ch = 0; ch < nchunks; ++ch) {
      const int kc0 = ch * CH;
      const int nk = min(CH, ((Lk - kc0 + 31) >> 5) << 5);  // rows used in this chunk (multiple of 32)
      if (nchunks > 1 || qb0 == qb_begin) {
        __syncthreads();
        stage_image<T, DK>(Kimg, kbase, a.kst, kc0, nk, Lk);
        stage_image<T, DK>(Vimg, vbase, a.vst, kc0, nk, Lk);
        stage_key_add(kadd, a.key_mask, b, kc0, nk, Lk, a.Lk, pair_any);
        __syncthreads();
        if (!CAPTURE && nchunks == 1) {
          okbits = pair_bits(pair_any);
          if (okbits == 0) okbits = 0xff;   // fully padded sequence: uniform attention over every key, like the reference
        }
      }
      if (!active) continue;
      // pairs of this wave in this chunk: all up to the last one that holds an attendable key (single-chunk
      // sequences; fully masked pairs inside that prefix are computed, their probabilities come out as exact zeros),
      // or — key-split mode — pairs wave, wave + 4
      const int npair = nk >> 5;
      int np, tp0 = 0, tstride = 1;
      if (ksplit) {
        tp0 = wave;
        tstride = 4;
        np = npair > wave ? (npair - wave + 3) >> 2 : 0;
      } else {
        np = min(npair, 32 - __builtin_clz(okbits & 0xff));
      }
      FwdRow st{m, lsum};
      const float* bl = has_bias ? bias_l : nullptr;
#define CHUNK(N) case N: fwd_chunk<T, DK, CAPTURE, N>(a, Kimg, Vimg, kadd, bl, qf, st, oacc, b, h, qi, kc0, tp0, tstride, prow, t_hi, jmax, lane); break;
      switch (np) {
        CHUNK(1) CHUNK(2) CHUNK(3) CHUNK(4) CHUNK(5) CHUNK(6) CHUNK(7) CHUNK(8)
        default: break;
      }
#undef CHUNK
      m = st.m;
      lsum = st.lsum;
    }
    if (ksplit) {
      // merge the 4 waves' partial softmax states (all waves hold the SAME 16 queries)
      float* mine = mrg + (wave * 16 + l15) * (DK + 2);
#pragma unroll
      for (int db = 0; db < C::NDB; ++db)
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[db * 16 + 4 * g + r] = oacc[db][r];
      if (g == 0) {
        mine[DK] = m;
        mine[DK + 1] = lsum;
      }
      __syncthreads();
      if (wave == 0) {
        float mw[4], M = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          mw[w] = mrg[(w * 16 + l15) * (DK + 2) + DK];
          M = fmaxf(M, mw[w]);
        }
        float sc[4];
        lsum = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          sc[w] = mw[w] == -INFINITY ? 0.f : __expf(mw[w] - M);
          lsum += sc[w] * mrg[(w * 16 + l15) * (DK + 2) + DK + 1];
        }
#pragma unroll
        for (int db = 0; db < C::NDB; ++db)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float acc = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) acc += sc[w] * mrg[(w * 16 + l15) * (DK + 2) + db * 16 + 4 * g + r];
            oacc[db][r] = acc;
          }
        m = M;
      }
    }
    if (active && qi < Lq && (!ksplit || wave == 0)) {
      const float inv = 1.0f / lsum;
      T* op = reinterpret_cast<T*>(obase + (int64_t)qi * a.ost * C::ES);
#pragma unroll
      for (int db = 0; db < C::NDB; ++db) store4(op + db * 16 + 4 * g, oacc[db] * inv);
      if (g == 0 && a.stats) {
        float* st = a.stats + (((int64_t)b * a.H + h) * a.Lq + qi) * 4;
        st[0] = m;
        st[1] = inv;
      }
    }
  }
}

// =============================================================================================
// backward.  MODE 0: dQ pass (keys in LDS, wave owns queries)   MODE 1: dK/dV pass (queries in LDS)
// (The branch-free per-pair-count dispatch of fwd_chunk was tried here too: 8 copies of this much larger pair body
// ran 35 % SLOWER — 256 VGPRs and an instruction footprint beyond the I-cache — so the pair loop keeps its `continue`s.)
// =============================================================================================
template <typename T, int DK, int MODE>
__global__ __launch_bounds__(256) void attn_bwd_kernel(AttnArgs a) {
  using C = AC<T, DK>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int CH = a.chunk_rows;
  char* X1 = smem;                               // MODE 0: K   MODE 1: Q
  char* X2 = smem + img_bytes<T, DK>(CH);        // MODE 0: V   MODE 1: dO
  float* aux0 = reinterpret_cast<float*>(smem + 2 * img_bytes<T, DK>(CH));  // MODE 0: key add   MODE 1: row max
  float* aux1 = aux0 + CH;                       // MODE 1: 1/rowsum (0 for padded query rows)
  float* aux2 = aux1 + CH;                       // MODE 1: delta
  float* bias_l = aux2 + CH;                     // [R]
  float* drel_l = bias_l + a.R;                  // [R] (MODE 0)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  // grid (register-side block groups, heads, batch-row groups) — or, dQ pass with several block groups per sequence
  // (bias gradient), a 1-D grid dealt out so that the groups sharing the staged K / V of the same (head, batch rows) run on
  // ONE XCD (workgroup id mod 8), i.e. behind one L2: id → (xcd, slot), slot → (block group, j), (j, xcd) → (head, rows)
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (MODE == 0 && a.grid_x > 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    bx = slot % a.grid_x;
    const int grp = (slot / a.grid_x) * 8 + xcd;
    if (grp >= a.H * a.grid_z) return;
    by = grp % a.H;
    bz = grp / a.H;
  }
  const int h = by;
  const int bn_per = MODE == 0 ? a.bn_per_wg : 1;
  const int b_begin = bz * bn_per, b_end = min(a.Bn, b_begin + bn_per);
  const int LXm = MODE == 0 ? a.Lk : a.Lq;  // LDS side length      } padded maxima: grid-uniform decisions
  const int LYm = MODE == 0 ? a.Lq : a.Lk;  // register side length }
  const int nyb = (LYm + 15) >> 4;
  const int yb_begin = bx * a.blocks_per_wg;
  const int yb_end = min(nyb, yb_begin + a.blocks_per_wg);
  const int nchunks_m = (LXm + CH - 1) / CH;
  // ragged mode: row offset and length of sequence bb on the query / key side.  The offsets of the workgroup's batch rows
  // are staged in LDS once: read from global they would put a dependent load in front of every round's operand requests.
  int* offs_l = reinterpret_cast<int*>(drel_l + ((a.R + 3) & ~3)) + 8;   // [2][OFFS_MAX], behind pair_any[8]
  if (threadIdx.x <= b_end - b_begin) {
    if (a.q_off) offs_l[threadIdx.x] = a.q_off[b_begin + threadIdx.x];
    if (a.k_off) offs_l[OFFS_MAX + threadIdx.x] = a.k_off[b_begin + threadIdx.x];
  }
  __syncthreads();
  // (readfirstlane: LDS values are per-lane to the compiler; as loop bounds they would make the loops divergent)
  auto seq_q = [&](int bb, int& o0, int& len) {
    o0 = a.q_off ? __builtin_amdgcn_readfirstlane(offs_l[bb - b_begin]) : 0;
    len = a.q_off ? __builtin_amdgcn_readfirstlane(offs_l[bb - b_begin + 1]) - o0 : a.Lq;
  };
  auto seq_k = [&](int bb, int& o0, int& len) {
    o0 = a.k_off ? __builtin_amdgcn_readfirstlane(offs_l[OFFS_MAX + bb - b_begin]) : 0;
    len = a.k_off ? __builtin_amdgcn_readfirstlane(offs_l[OFFS_MAX + bb - b_begin + 1]) - o0 : a.Lk;
  };
  auto base_q = [&](const char* ptr, int bb, int o0, int64_t sb, int64_t st) {
    return ptr + ((a.q_off ? (int64_t)o0 * st : (int64_t)bb * sb) + (int64_t)h * DK) * C::ES;
  };
  auto base_k = [&](const char* ptr, int bb, int o0, int64_t sb, int64_t st) {
    return ptr + ((a.k_off ? (int64_t)o0 * st : (int64_t)bb * sb) + (int64_t)h * DK) * C::ES;
  };
  const int64_t hoff = (int64_t)h * DK;
  const bool has_bias = a.rel_bias != nullptr;
  const bool want_drel = MODE == 0 && a.drel != nullptr;
  if (has_bias)
    for (int i = threadIdx.x; i < a.R; i += 256) bias_l[i] = a.rel_bias[(int64_t)h * a.R + i];
  if (want_drel)
    for (int i = threadIdx.x; i < a.R; i += 256) drel_l[i] = 0.f;
  // one contribution to the bias gradient of offset idx: an LDS float atomic into the workgroup's table — or, LAKO_DETERMINISTIC (det.h), a
  // fixed-point integer atomic straight into the launch's table (slower: global instead of LDS; in that mode this kernel serves the
  // decoder's eight-token self-attention and whatever the encoder's one-pass backward does not take)
  auto drel_add = [&](int idx, float v) {
    if (a.det) lako_det::add(&lako_det_slots[h * a.R + idx], v);
    else atomicAdd(&drel_l[idx], v);
  };
  const int hq = (a.Lk + 3) >> 2, hqq = (a.Lq + 3) >> 2;
  const uint32_t t_hi = a.drop_t16 << 16;

  // key-split (dQ pass only): a workgroup owning one query block shares it between its 4 waves, which split the
  // keys (tile pair tp → wave tp & 3); the partial dQ tiles are summed through LDS at the end.
  const bool ksplit = MODE == 0 && nyb == 1 && a.Lk > 64;   // merge scratch reserved only when Lq <= 16
  int* pair_any = reinterpret_cast<int*>(drel_l + ((a.R + 3) & ~3));   // [8] (MODE 0), [0] = sequence has a key (MODE 1)
  float* mrg = reinterpret_cast<float*>(pair_any + 8 + 2 * OFFS_MAX);  // [4][16][DK] (ksplit only; after offs_l)
  int okbits = 0xff;          // MODE 0: tile pairs of the staged keys worth computing (see stage_key_add)

  // Bias gradient (dQ pass): d rel_bias[key − query] = Σ_batch Σ_(q,k) dS.  LDS float atomics per score
  // element are ~40 cycles per wave-instruction, so when one round covers the workgroup's query blocks
  // (each wave keeps ONE query block) the dS tiles are summed in registers over the workgroup's batch rows
  // (tile position → fixed diagonal offsets) and reduced over diagonals once at the end.
  const bool fast_drel = want_drel && nchunks_m == 1 && (yb_end - yb_begin) <= 4;
  f32x4 dsacc[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) dsacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // register-side operands of the NEXT round (possibly the next batch row's first) are requested before this round is
  // computed — see attn_fwd_kernel
  u32x4 n1[C::NF], n2[C::NF], n3[C::NF];
  float ns0 = 0.f, ns1 = 0.f, nkadd = 0.f;
  auto request = [&](int bb, int ybb0) {
    int o0 = 0, len = 0;
    if (bb < b_end) {
      if (MODE == 0) seq_q(bb, o0, len);
      else seq_k(bb, o0, len);
    }
    const int ybn = ksplit ? ybb0 : ybb0 + wave;
    const bool act = bb < b_end && ybb0 < yb_end && ybn * 16 < len;
    const int yin = act ? ybn * 16 + l15 : len;
    if constexpr (MODE == 0) {
      load_reg_frags<T, DK>(n1, base_q(a.q, bb, o0, a.qsb, a.qst), a.qst, yin, len, lane);
      load_reg_frags<T, DK>(n2, base_q(a.dout, bb, o0, a.osb, a.ost), a.ost, yin, len, lane);
      load_reg_frags<T, DK>(n3, base_q(a.o, bb, o0, a.osb, a.ost), a.ost, yin, len, lane);
      ns0 = ns1 = 0.f;
      if (yin < len) {
        const float* st = a.stats + (((int64_t)bb * a.H + h) * a.Lq + yin) * 4;
        ns0 = st[0];
        ns1 = st[1];
      }
    } else {
      load_reg_frags<T, DK>(n1, base_k(a.k, bb, o0, a.ksb, a.kst), a.kst, yin, len, lane);
      load_reg_frags<T, DK>(n2, base_k(a.v, bb, o0, a.vsb, a.vst), a.vst, yin, len, lane);
      nkadd = 0.f;
      if (yin >= len) nkadd = -INFINITY;
      else if (a.key_mask && !a.key_mask[(int64_t)bb * a.Lk + yin]) nkadd = -FLT_MAX;
    }
  };

  // sequences that have nothing for this workgroup (ragged: shorter than its first block, or empty) are stepped over, so
  // that the operands requested ahead are always those of the round that runs next
  auto next_with_work = [&](int bb) {
    for (; bb < b_end; ++bb) {
      int oq, lq, ok, lk;
      seq_q(bb, oq, lq);
      seq_k(bb, ok, lk);
      if (((MODE == 0 ? lq : lk) + 15) >> 4 > yb_begin && (MODE == 0 ? lk : lq) > 0) break;
    }
    return bb;
  };
  int next_b = next_with_work(b_begin);
  request(next_b, yb_begin);
  for (int b = next_b; b < b_end; b = next_b) {
  int q0, k0, Lq, Lk;
  seq_q(b, q0, Lq);
  seq_k(b, k0, Lk);
  const int LX = MODE == 0 ? Lk : Lq, LY = MODE == 0 ? Lq : Lk;   // this sequence's own lengths (bounds)
  const int ybe = min(yb_end, (LY + 15) >> 4);
  const int nchunks = (LX + CH - 1) / CH;
  const char* qbase = base_q(a.q, b, q0, a.qsb, a.qst);
  const char* kbase = base_k(a.k, b, k0, a.ksb, a.kst);
  const char* vbase = base_k(a.v, b, k0, a.vsb, a.vst);
  const char* dobase = base_q(a.dout, b, q0, a.osb, a.ost);
  float* stats = a.stats + ((int64_t)b * a.H + h) * a.Lq * 4;
  bool seq_any = true;        // MODE 1: does this sequence have an attendable key at all?
  if (MODE == 1 && a.key_mask) {
    __syncthreads();
    if (threadIdx.x == 0) pair_any[0] = 0;
    __syncthreads();
    bool mine = false;
    for (int j = threadIdx.x; j < Lk; j += 256) mine |= a.key_mask[(int64_t)b * a.Lk + j] != 0;
    if (__ballot(mine) != 0 && (threadIdx.x & 63) == 0) pair_any[0] = 1;
    __syncthreads();
    seq_any = pair_any[0] != 0;
  }
  for (int yb0 = yb_begin; yb0 < ybe; yb0 += 4) {
    const int yb = ksplit ? yb0 : yb0 + wave;
    const bool active = yb < ybe;
    const int yi = active ? yb * 16 + l15 : LY;  // this lane's register-side row (query in MODE 0, key in MODE 1)
    u32x4 y1[C::NF], y2[C::NF];
    float m_q = 0.f, invl_q = 0.f, delta_q = 0.f;   // MODE 0: this lane's query statistics (inv = 0 ⇒ p = 0)
    float kadd_lane = 0.f;                           // MODE 1: this lane's key mask term
    u32x4 of[C::NF];
#pragma unroll
    for (int i = 0; i < C::NF; ++i) {
      y1[i] = n1[i];
      y2[i] = n2[i];
      if constexpr (MODE == 0) of[i] = n3[i];
    }
    const float st_m = ns0, st_il = ns1;
    kadd_lane = nkadd;
    if (yb0 + 4 < ybe) request(b, yb0 + 4);
    else request(next_b = next_with_work(b + 1), yb_begin);
    if constexpr (MODE == 0) {
      float part = 0.f;
#pragma unroll
      for (int i = 0; i < C::NF; ++i) {
        if constexpr (sizeof(T) == 2) {
          bf16x8 dv = __builtin_bit_cast(bf16x8, y2[i]), ov = __builtin_bit_cast(bf16x8, of[i]);
#pragma unroll
          for (int e = 0; e < 8; ++e) part += (float)dv[e] * (float)ov[e];
        } else {
          f32x4 dv = __builtin_bit_cast(f32x4, y2[i]), ov = __builtin_bit_cast(f32x4, of[i]);
#pragma unroll
          for (int e = 0; e < 4; ++e) part += dv[e] * ov[e];
        }
      }
      delta_q = group_sum(part);
      if (yi < Lq) {
        m_q = st_m;
        invl_q = st_il;
        if (g == 0) stats[yi * 4 + 2] = delta_q;   // handed to the dK/dV pass (launched after this one)
      }
    }
    // MODE 1: a block of 16 keys none of which is attendable has dK = dV = 0 exactly (p = 0 for them on every row
    // that has an attendable key): nothing to compute, the zero accumulators are stored
    const bool compute = active && !(MODE == 1 && seq_any && __ballot(kadd_lane == 0.f) == 0);
    f32x4 acc1[C::NDB], acc2[C::NDB];  // MODE 0: dQ (acc1)   MODE 1: dK (acc1), dV (acc2)
#pragma unroll
    for (int db = 0; db < C::NDB; ++db) {
      acc1[db] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc2[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    for (int ch = 0; ch < nchunks; ++ch) {
      const int x0 = ch * CH;
      const int nx = min(CH, ((LX - x0 + 31) >> 5) << 5);
      if (nchunks > 1 || yb0 == yb_begin) {
        __syncthreads();
        if constexpr (MODE == 0) {
          stage_image<T, DK>(X1, kbase, a.kst, x0, nx, Lk);
          stage_image<T, DK>(X2, vbase, a.vst, x0, nx, Lk);
          stage_key_add(aux0, a.key_mask, b, x0, nx, Lk, a.Lk, pair_any);
        } else {
          stage_image<T, DK>(X1, qbase, a.qst, x0, nx, Lq);
          stage_image<T, DK>(X2, dobase, a.ost, x0, nx, Lq);
          for (int i = threadIdx.x; i < nx; i += 256) {
            const int qi = x0 + i;
            f32x4 st4 = {0.f, 0.f, 0.f, 0.f};
            if (qi < Lq) st4 = *reinterpret_cast<const f32x4*>(stats + qi * 4);
            aux0[i] = st4[0];
            aux1[i] = st4[1];   // 0 for padded rows ⇒ p = 0 there
            aux2[i] = st4[2];   // delta = rowsum(dO∘O), written by the dQ pass
          }
        }
        __syncthreads();
        if (MODE == 0 && nchunks == 1) {
          okbits = pair_bits(pair_any);
          if (okbits == 0) okbits = 0xff;   // fully padded sequence: every key takes part (uniform attention)
        }
      }
      if (!compute) continue;
      const int npair = nx >> 5;
#pragma unroll
      for (int tp = 0; tp < 8; ++tp) {
        if (tp >= npair) continue;
        if (MODE == 0 && !((okbits >> tp) & 1)) continue;
        if (ksplit && (tp & 3) != wave) continue;
        f32x4 pt[2], ds[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int row0 = tp * 32 + t * 16;
          const int x_first = x0 + row0 + 4 * g;          // LDS-side index of element r = 0
          // additive terms as accumulator init: bias + key mask
          f32x4 init = {kadd_lane, kadd_lane, kadd_lane, kadd_lane};
          if constexpr (MODE == 0) init = *reinterpret_cast<const f32x4*>(aux0 + row0 + 4 * g);
          int bi0 = 0;
          if (has_bias) {
            bi0 = (MODE == 0 ? x_first - yi : yi - x_first) + a.rel_off;   // key − query + off of element r = 0
#pragma unroll
            for (int r = 0; r < 4; ++r) init[r] += bias_l[clampi(MODE == 0 ? bi0 + r : bi0 - r, 0, a.R - 1)];
          }
          f32x4 sv = score_tile<T, DK>(X1, row0, y1, lane, init);
          f32x4 dp = score_tile<T, DK>(X2, row0, y2, lane, f32x4{0.f, 0.f, 0.f, 0.f});
          f32x4 mq, il, dl;
          if constexpr (MODE == 0) {
            mq = f32x4{m_q, m_q, m_q, m_q};
            il = f32x4{invl_q, invl_q, invl_q, invl_q};
            dl = f32x4{delta_q, delta_q, delta_q, delta_q};
          } else {
            mq = *reinterpret_cast<const f32x4*>(aux0 + row0 + 4 * g);
            il = *reinterpret_cast<const f32x4*>(aux1 + row0 + 4 * g);
            dl = *reinterpret_cast<const f32x4*>(aux2 + row0 + 4 * g);
          }
          if (a.causal) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int key = MODE == 0 ? x_first + r : yi, qi = MODE == 0 ? yi : x_first + r;
              sv[r] = key > qi + a.causal_off ? -FLT_MAX : sv[r];
            }
          }
          f32x4 p;
#pragma unroll
          for (int r = 0; r < 4; ++r) p[r] = __expf(sv[r] - mq[r]) * il[r];
          f32x4 pd = p, dpd = dp;
          if (a.drop_t16) {
            bool kp[4];
            if constexpr (MODE == 0) {   // 4 consecutive keys of one query: one block row (2 words)
              drop_keep_row(drop_base((uint32_t)((b * a.H + h) * hqq + (yi >> 2)) * (uint32_t)hq + (uint32_t)(x_first >> 2), a.drop_key),
                            drop_row_consts(yi), t_hi, kp);
            } else {                      // one key, 4 consecutive queries: one block column (4 words)
              drop_keep_col(drop_base((uint32_t)((b * a.H + h) * hqq + (x_first >> 2)) * (uint32_t)hq + (uint32_t)(yi >> 2), a.drop_key),
                            drop_col_consts(yi), t_hi, kp);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              pd[r] = kp[r] ? p[r] * a.drop_scale : 0.f;
              dpd[r] = kp[r] ? dp[r] * a.drop_scale : 0.f;
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) ds[t][r] = p[r] * (dpd[r] - dl[r]);
          pt[t] = pd;
          if (MODE == 0 && fast_drel) dsacc[2 * tp + t] += ds[t];
          else if (want_drel) {
#pragma unroll
            for (int r = 0; r < 4; ++r) drel_add(clampi(bi0 + r, 0, a.R - 1), ds[t][r]);
          }
        }
        if constexpr (MODE == 0) {
          pv_accumulate<T, DK>(acc1, ds[0], ds[1], X1, tp * 32, lane);  // dQᵀ += Kᵀ·dSᵀ
        } else {
          pv_accumulate<T, DK>(acc1, ds[0], ds[1], X1, tp * 32, lane);  // dKᵀ += Qᵀ·dS
          pv_accumulate<T, DK>(acc2, pt[0], pt[1], X2, tp * 32, lane);  // dVᵀ += dOᵀ·P̃
        }
      }
    }
    if (ksplit) {
      float* mine = mrg + (wave * 16 + l15) * DK;
#pragma unroll
      for (int db = 0; db < C::NDB; ++db)
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[db * 16 + 4 * g + r] = acc1[db][r];
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int db = 0; db < C::NDB; ++db)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float acc = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) acc += mrg[(w * 16 + l15) * DK + db * 16 + 4 * g + r];
            acc1[db][r] = acc;
          }
      }
    }
    if (active && yi < LY && (!ksplit || wave == 0)) {
      if constexpr (MODE == 0) {
        T* op = reinterpret_cast<T*>(const_cast<char*>(base_q(a.dq, b, q0, a.qsb, a.qst)) + (int64_t)yi * a.qst * C::ES);
#pragma unroll
        for (int db = 0; db < C::NDB; ++db) store4(op + db * 16 + 4 * g, acc1[db]);
      } else {
        T* kp = reinterpret_cast<T*>(const_cast<char*>(base_k(a.dk, b, k0, a.ksb, a.kst)) + (int64_t)yi * a.kst * C::ES);
        T* vp = reinterpret_cast<T*>(const_cast<char*>(base_k(a.dv, b, k0, a.vsb, a.vst)) + (int64_t)yi * a.vst * C::ES);
#pragma unroll
        for (int db = 0; db < C::NDB; ++db) {
          store4(kp + db * 16 + 4 * g, acc1[db]);
          store4(vp + db * 16 + 4 * g, acc2[db]);
        }
      }
    }
  }
  }  // batch rows of this workgroup
  if (MODE == 0 && fast_drel && !ATTN_DBG(a, 8)) {
    // Reduce the register-accumulated dS tiles over diagonals (key − query).  One LDS float atomic per element (64 per lane)
    // cost ≈95 µs per layer; instead each wave lays its 16-query × CH-key stripe out in LDS — the K / V images are dead by
    // now: 64·CH bytes per wave of the 288·CH they occupy — row q rotated by 4q floats so that both the 16-B tile writes and
    // the reads ALONG a diagonal (consecutive lanes = consecutive diagonals = consecutive addresses) are conflict-free, sums
    // each diagonal with 16 plain reads and issues ONE atomic per diagonal.
    __syncthreads();   // every wave is done with the images
    if constexpr (2 * C::ROWB < 256) {   // bf16 with d_head 32: the images are smaller than four stripes — per-element atomics
      const int yq = (yb_begin + wave) * 16 + l15;
      if (yb_begin + wave < yb_end) {
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            drel_add(clampi(t * 16 + 4 * g + r - yq + a.rel_off, 0, a.R - 1), dsacc[t][r]);
      }
    } else if (yb_begin + wave < yb_end) {
      float* reg = reinterpret_cast<float*>(smem) + wave * 16 * CH;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        if (t * 16 < CH) {
          int pos = t * 16 + 4 * g + 4 * l15;
          pos = pos >= CH ? pos - CH : pos;     // 4·l15 <= 60 < CH + 1 only when CH >= 64; CH = 32: second wrap below
          pos = pos >= CH ? pos - CH : pos;
          *reinterpret_cast<f32x4*>(reg + l15 * CH + pos) = dsacc[t];     // 4-float groups never straddle the wrap (CH % 4 == 0)
        }
      }
      const int yq0 = (yb_begin + wave) * 16;
      for (int dd = lane; dd < CH + 15; dd += 64) {
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int key = q + dd - 15;
          int pos = key + (4 * q) % CH;
          pos = pos >= CH ? pos - CH : pos;
          if (key >= 0 && key < CH) sum += reg[q * CH + pos];
        }
        if (sum != 0.f) drel_add(clampi(dd - 15 - yq0 + a.rel_off, 0, a.R - 1), sum);
      }
    }
  }
  if (want_drel && a.det) {
    // (the XCD-grouped 1-D grid of the dQ pass carries surplus workgroups that left at the top: they draw no ticket)
    const unsigned int nblocks = a.grid_x > 0 ? (unsigned)(a.grid_x * a.H * a.grid_z) : gridDim.x * gridDim.y * gridDim.z;
    // (the flag: first word of the dynamic allocation, dead after the barrier inside last_block — no static LDS next to `smem`)
    if (lako_det::last_block(&lako_det_ticket, nblocks, reinterpret_cast<int*>(smem)))
      for (int i = threadIdx.x; i < a.H * a.R; i += 256) a.drel[i] += lako_det::take(&lako_det_slots[i]);
    return;
  }
  if (want_drel && !ATTN_DBG(a, 4)) {
    __syncthreads();
    for (int i = threadIdx.x; i < a.R; i += 256) {
      float v = drel_l[i];
      if (v != 0.f) atomicAdd(a.drel + (int64_t)h * a.R + i, v);
    }
  }
}

template <typename T, int DK>
int lds_bytes_fwd(int ch, int R, bool merge) {
  return 2 * img_bytes<T, DK>(ch) + (ch + R + 16 + (merge ? 4 * 16 * (DK + 2) : 0)) * 4;
}
template <typename T, int DK>
int lds_bytes_bwd(int ch, int R, bool merge) {
  return 2 * img_bytes<T, DK>(ch) + (3 * ch + 2 * R + 16 + 2 * OFFS_MAX + (merge ? 4 * 16 * DK : 0)) * 4;
}

inline int pick_chunk(int L) {
  int ch = ((L + 31) / 32) * 32;
  return ch > CH_MAX ? CH_MAX : ch;
}

// raise the kernel's dynamic-LDS limit only when a launch needs more than any earlier one ON THIS DEVICE (`cur` is a per-instantiation
// static owned by the caller; the attribute is per device: common.h)
template <typename K>
void set_lds_attr(K kern, int bytes, lako_lds_cur_t& cur) {
  lako_raise_max_lds(reinterpret_cast<const void*>(kern), bytes, cur);
}

// choose register-side blocks per workgroup: all of them when there are already plenty of workgroups,
// otherwise split so the grid reaches ~1024 workgroups (no LDS-side merge needed: blocks are independent)
inline int pick_blocks_per_wg(int nblocks, int64_t bh) {
  int64_t want = 1024;
  if (bh >= want || nblocks <= 4) return ((nblocks + 3) / 4) * 4;
  int groups = (int)((want + bh - 1) / bh);
  int per = (nblocks + groups - 1) / groups;
  per = ((per + 3) / 4) * 4;
  return per < 4 ? 4 : per;
}

template <typename T, int DK>
int run_fwd(AttnArgs& a, hipStream_t s) {
  a.chunk_rows = pick_chunk(a.Lk);
  const int nqb = (a.Lq + 15) / 16;
  a.blocks_per_wg = pick_blocks_per_wg(nqb, (int64_t)a.Bn * a.H);
  const int lds = lds_bytes_fwd<T, DK>(a.chunk_rows, a.R, nqb == 1);
  dim3 grid((nqb + a.blocks_per_wg - 1) / a.blocks_per_wg, a.H, a.Bn);
  if (a.scores_out) {
    static lako_lds_cur_t curc;
    set_lds_attr(&attn_fwd_kernel<T, DK, true>, lds, curc);
    hipLaunchKernelGGL((attn_fwd_kernel<T, DK, true>), grid, dim3(256), lds, s, a);
  } else {
    static lako_lds_cur_t cur;
    set_lds_attr(&attn_fwd_kernel<T, DK, false>, lds, cur);
    hipLaunchKernelGGL((attn_fwd_kernel<T, DK, false>), grid, dim3(256), lds, s, a);
  }
  return 0;
}

template <typename T, int DK>
int run_bwd(AttnArgs& a, hipStream_t s) {
  {  // dQ pass
    AttnArgs q = a;
    q.chunk_rows = pick_chunk(a.Lk);
    const int nqb = (a.Lq + 15) / 16;
    q.blocks_per_wg = pick_blocks_per_wg(nqb, (int64_t)a.Bn * a.H);
    q.bn_per_wg = 1;
#ifdef LAKO_EXPERIMENTS   // timing experiments (bits 4 / 8 drop the bias-gradient flushes: wrong results) — never in the release library
    static const int dbg = getenv("LAKO_ATTN_DEBUG") ? atoi(getenv("LAKO_ATTN_DEBUG")) : 0;
#else
    constexpr int dbg = 0;
#endif
    if ((a.drel || (dbg & 1)) && a.Lk <= CH_MAX) {
      // bias gradient: one query block per wave, several batch rows per workgroup (register accumulation of dS)
      q.blocks_per_wg = 4;
      const int64_t wgs = (int64_t)a.Bn * a.H * ((nqb + 3) / 4);
      static const int wg_target = getenv("LAKO_ATTN_WGS") ? atoi(getenv("LAKO_ATTN_WGS")) : 1024;
      q.bn_per_wg = (int)(wgs / wg_target > 1 ? wgs / wg_target : 1);
      if (q.bn_per_wg > OFFS_MAX - 1) q.bn_per_wg = OFFS_MAX - 1;   // the kernel stages bn_per_wg + 1 row offsets in LDS
    }
    const int lds = lds_bytes_bwd<T, DK>(q.chunk_rows, a.R, nqb == 1);
    static lako_lds_cur_t cur0;
    set_lds_attr(&attn_bwd_kernel<T, DK, 0>, lds, cur0);
    dim3 grid((nqb + q.blocks_per_wg - 1) / q.blocks_per_wg, a.H, (a.Bn + q.bn_per_wg - 1) / q.bn_per_wg);
    q.grid_x = 0;
    q.dbg_flags = dbg;
    if (grid.x > 1 && !(dbg & 2)) {
      q.grid_x = (int)grid.x;
      q.grid_z = (int)grid.z;
      const int groups = (int)(grid.y * grid.z);
      grid = dim3((unsigned)(((groups + 7) / 8) * 8 * q.grid_x), 1, 1);
    }
    hipLaunchKernelGGL((attn_bwd_kernel<T, DK, 0>), grid, dim3(256), lds, s, q);
  }
  {  // dK/dV pass
    AttnArgs k = a;
    k.bn_per_wg = 1;
    k.grid_x = 0;
    k.chunk_rows = pick_chunk(a.Lq);
    const int nkb = (a.Lk + 15) / 16;
    k.blocks_per_wg = pick_blocks_per_wg(nkb, (int64_t)a.Bn * a.H);
    const int lds = lds_bytes_bwd<T, DK>(k.chunk_rows, a.R, false);
    static lako_lds_cur_t cur1;
    set_lds_attr(&attn_bwd_kernel<T, DK, 1>, lds, cur1);
    dim3 grid((nkb + k.blocks_per_wg - 1) / k.blocks_per_wg, a.H, a.Bn);
    hipLaunchKernelGGL((attn_bwd_kernel<T, DK, 1>), grid, dim3(256), lds, s, k);
  }
  return 0;
}

int check_common(const char* fn, int Bn, int H, int Lq, int Lk, int d_head, int dtype, int64_t qst, int64_t kst,
                 int64_t vst, int64_t ost, const void* rel_bias, int R, float p) {
  if (!(Bn > 0 && H > 0 && Lq > 0 && Lk > 0)) { lako_set_error("%s: bad dims", fn); return LAKO_E_BADARG; }
  if (!(Bn < 65536 && H < 65536)) { lako_set_error("%s: Bn/H too large for the grid", fn); return LAKO_E_BADARG; }
  if (!(dtype == LAKO_F32 || dtype == LAKO_BF16)) { lako_set_error("%s: bad dtype", fn); return LAKO_E_BADARG; }
  if (!(d_head == 32 || d_head == 64)) {
    lako_set_error("%s: d_head=%d unsupported (32 or 64)", fn, d_head);
    return LAKO_E_UNSUPPORTED;
  }
  const int esz = dtype == LAKO_F32 ? 4 : 2;
  if ((qst * esz) % 16 || (kst * esz) % 16 || (vst * esz) % 16 || (ost * esz) % 16) {
    lako_set_error("%s: token strides must be multiples of 16 bytes", fn);
    return LAKO_E_ALIGN;
  }
  if (rel_bias && !(R > 0 && R <= 4096)) { lako_set_error("%s: bad R=%d", fn, R); return LAKO_E_BADARG; }
  if (!(p >= 0.f && p < 1.f)) { lako_set_error("%s: dropout p out of range", fn); return LAKO_E_BADARG; }
  if (p > 0.f && (int64_t)Bn * H * ((Lq + 3) / 4) * ((Lk + 3) / 4) >= (1ll << 32)) {
    lako_set_error("%s: dropout block index exceeds 32 bits", fn);
    return LAKO_E_UNSUPPORTED;
  }
  return LAKO_OK;
}

void set_drop(AttnArgs& a, const lako_dropout_t& d) {
  a.drop_t16 = 0;
  a.drop_scale = 1.0f;
  if (d.p > 0.f) {
    uint32_t t = (uint32_t)((double)d.p * 65536.0 + 0.5);
    a.drop_t16 = t < 1 ? 1 : (t > 65535 ? 65535 : t);
    a.drop_scale = 1.0f / (1.0f - d.p);
  }
  a.drop_key = lako_drop_key(d.seed, d.site);
  a.det = lako_det::on() ? 1 : 0;
}

}  // namespace

#define ATTN_DISPATCH(dtype, dhead, FN, args, s)                    \
  if ((dtype) == LAKO_BF16) {                                       \
    if ((dhead) == 64) FN<bf16_t, 64>(args, s);                     \
    else FN<bf16_t, 32>(args, s);                                   \
  } else {                                                          \
    if ((dhead) == 64) FN<float, 64>(args, s);                      \
    else FN<float, 32>(args, s);                                    \
  }

extern "C" int lako_attn_fwd(const lako_attn_fwd_t* p, lako_stream_t stream) {
  LAKO_CHECK_ARG(p != nullptr, "lako_attn_fwd: null params");
  int rc = check_common("lako_attn_fwd", p->Bn, p->H, p->Lq, p->Lk, p->d_head, p->dtype, p->q_stride_t, p->k_stride_t,
                        p->v_stride_t, p->o_stride_t, p->rel_bias, p->R, p->drop.p);
  if (rc) return rc;
  LAKO_CHECK_ALIGN(p->q, 16);
  LAKO_CHECK_ALIGN(p->k, 16);
  LAKO_CHECK_ALIGN(p->v, 16);
  LAKO_CHECK_ALIGN(p->out, 8);
  AttnArgs a = {};
  a.q = (const char*)p->q;
  a.k = (const char*)p->k;
  a.v = (const char*)p->v;
  a.out = (char*)p->out;
  a.stats = p->lse;
  a.rel_bias = p->rel_bias;
  a.key_mask = p->key_mask;
  a.scores_out = p->scores_out;
  a.qsb = p->q_stride_b; a.qst = p->q_stride_t; a.ksb = p->k_stride_b; a.kst = p->k_stride_t;
  a.vsb = p->v_stride_b; a.vst = p->v_stride_t; a.osb = p->o_stride_b; a.ost = p->o_stride_t;
  a.R = p->rel_bias ? p->R : 0;
  a.rel_off = p->rel_off;
  a.causal = p->causal;
  a.causal_off = p->causal_off;
  a.Bn = p->Bn; a.H = p->H; a.Lq = p->Lq; a.Lk = p->Lk;
  a.q_off = p->q_off;
  a.k_off = p->k_off;
  a.order = p->order;
  LAKO_CHECK_ARG(!(p->scores_out && (p->q_off || p->k_off)), "lako_attn_fwd: score capture needs the padded layout");
  LAKO_CHECK_ARG(!(p->key_mask && p->k_off), "lako_attn_fwd: ragged keys carry no padding — pass key_mask = NULL");
  set_drop(a, p->drop);
  if (lako_attn_enc_supported(a, p->dtype, p->d_head)) lako_attn_enc_fwd(a, (hipStream_t)stream);
  else ATTN_DISPATCH(p->dtype, p->d_head, run_fwd, a, (hipStream_t)stream);
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

extern "C" int lako_attn_bwd(const lako_attn_bwd_t* p, lako_stream_t stream) {
  LAKO_CHECK_ARG(p != nullptr, "lako_attn_bwd: null params");
  int rc = check_common("lako_attn_bwd", p->Bn, p->H, p->Lq, p->Lk, p->d_head, p->dtype, p->q_stride_t, p->k_stride_t,
                        p->v_stride_t, p->o_stride_t, p->rel_bias, p->R, p->drop.p);
  if (rc) return rc;
  LAKO_CHECK_ARG(p->lse && p->out && p->dout && p->dq_out && p->dk_out && p->dv_out, "lako_attn_bwd: null tensor");
  LAKO_CHECK_ARG(!p->drel || p->rel_bias, "lako_attn_bwd: drel without rel_bias");
  AttnArgs a = {};
  a.q = (const char*)p->q;
  a.k = (const char*)p->k;
  a.v = (const char*)p->v;
  a.o = (const char*)p->out;
  a.dout = (const char*)p->dout;
  a.dq = (char*)p->dq_out;
  a.dk = (char*)p->dk_out;
  a.dv = (char*)p->dv_out;
  a.stats = p->lse;
  a.rel_bias = p->rel_bias;
  a.drel = p->drel;
  a.key_mask = p->key_mask;
  a.qsb = p->q_stride_b; a.qst = p->q_stride_t; a.ksb = p->k_stride_b; a.kst = p->k_stride_t;
  a.vsb = p->v_stride_b; a.vst = p->v_stride_t; a.osb = p->o_stride_b; a.ost = p->o_stride_t;
  a.R = p->rel_bias ? p->R : 0;
  a.rel_off = p->rel_off;
  a.causal = p->causal;
  a.causal_off = p->causal_off;
  a.Bn = p->Bn; a.H = p->H; a.Lq = p->Lq; a.Lk = p->Lk;
  a.q_off = p->q_off;
  a.k_off = p->k_off;
  a.order = p->order;
  LAKO_CHECK_ARG(!(p->key_mask && p->k_off), "lako_attn_bwd: ragged keys carry no padding — pass key_mask = NULL");
  set_drop(a, p->drop);
  if (a.det && a.drel && (int64_t)a.H * a.R > lako_det::SLOTS) {
    lako_set_error("lako_attn_bwd: LAKO_DETERMINISTIC holds the bias gradient of %d heads x %d offsets in a table of %d slots", a.H, a.R, lako_det::SLOTS);
    return LAKO_E_UNSUPPORTED;
  }
  bool done = false;      // (the encoder's fast path declines in LAKO_DETERMINISTIC mode where its one-pass kernel does not apply)
  if (lako_attn_enc_supported(a, p->dtype, p->d_head)) done = lako_attn_enc_bwd(a, (hipStream_t)stream) == 0;
  if (!done) ATTN_DISPATCH(p->dtype, p->d_head, run_bwd, a, (hipStream_t)stream);
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}
