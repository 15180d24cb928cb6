// Row-wise BACKWARD kernels of the retriever's BERT bi-encoder — SURVEY.md §8 f4, the training half: the autograd of
// src/model.py:375-483 (`Retriever.forward` with `gold_score`: KL distillation of the reader's cross-attention scores into the
// bi-encoder, train_retriever.py:37-71) through HF `BertModel` (modeling_bert.py: post-LN blocks, biased linears, exact GELU).
// The matrix products are the reader's kernels (lako_gemm_nt for dX with its residual epilogue, lako_gemm_tn for dW), the attention
// backward is lako_attn_bwd; what is here is everything between them.  All HBM-bound: one wave per row or a thread per column
// group, 16-byte accesses, wave64 shuffle reductions; parameter-gradient column sums are kept in registers per wave / thread,
// reduced per block in LDS and leave as ONE float atomic per column and block.
#include "common.h"

namespace {

__device__ __forceinline__ void ld8(const float* p, float (&v)[8]) {
  f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}
__device__ __forceinline__ void ld8(const bf16_t* p, float (&v)[8]) {
  bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
}
__device__ __forceinline__ void st8(float* p, const float (&v)[8]) {
  *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
__device__ __forceinline__ void st8(bf16_t* p, const float (&v)[8]) {
  bf16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (bf16_t)v[i];
  *reinterpret_cast<bf16x8*>(p) = o;
}

constexpr int LNB_MAXD = 1024;

// block-level reduction of per-wave column partials (4 waves × d columns) and one atomic per column
__device__ __forceinline__ void flush_cols(float (*red)[LNB_MAXD], const float (&acc)[2][8], float* dst, int d, int lane, int wave) {
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int c = it * 512 + lane * 8;
    if (c < d) {
#pragma unroll
      for (int i = 0; i < 8; ++i) red[wave][c + i] = acc[it][i];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < d; c += 256) atomicAdd(dst + c, red[0][c] + red[1][c] + red[2][c] + red[3][c]);
  __syncthreads();
}

// Backward of y = LayerNorm(z)·γ + β with z = x + lin_bias + resid (torch.nn.LayerNorm; the forward is lako_layernorm_fwd, which
// keeps nothing: z, its mean and 1/std are recomputed from the same inputs):
//   zh = (z − mean)·rstd, g = dy·γ, dz = rstd·(g − mean(g) − zh·mean(g·zh));  dγ += Σ_rows dy·zh, dβ += Σ_rows dy, d(lin_bias) += Σ_rows dz.
// dz is the gradient of x AND of resid.
template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ lin_bias,
                                                            const T* __restrict__ resid, const float* __restrict__ gamma, T* __restrict__ dz,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dbias,
                                                            int64_t rows, int d, float eps) {
  __shared__ float red[4][LNB_MAXD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t wid = (int64_t)blockIdx.x * 4 + wave, nw = (int64_t)gridDim.x * 4;
  float ag[2][8], ab[2][8], al[2][8], gm[2][8];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int c = it * 512 + lane * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) ag[it][i] = ab[it][i] = al[it][i] = gm[it][i] = 0.f;
    if (c < d) ld8(gamma + c, gm[it]);
  }
  for (int64_t row = wid; row < rows; row += nw) {
    float z[2][8], g[2][8];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int c = it * 512 + lane * 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) z[it][i] = g[it][i] = 0.f;
      if (c < d) {
        ld8(x + row * d + c, z[it]);
        ld8(dy + row * d + c, g[it]);
        if (lin_bias) {
          float b[8];
          ld8(lin_bias + c, b);
#pragma unroll
          for (int i = 0; i < 8; ++i) z[it][i] += b[i];
        }
        if (resid) {
          float r[8];
          ld8(resid + row * d + c, r);
#pragma unroll
          for (int i = 0; i < 8; ++i) z[it][i] += r[i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) s += z[it][i];
      }
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int c = it * 512 + lane * 8;
      if (c < d) {
#pragma unroll
        for (int i = 0; i < 8; ++i) q += (z[it][i] - mean) * (z[it][i] - mean);
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + eps);
    float sg = 0.f, sgz = 0.f;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int c = it * 512 + lane * 8;
      if (c < d) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float zh = (z[it][i] - mean) * rstd, dyv = g[it][i];
          ag[it][i] += dyv * zh;
          ab[it][i] += dyv;
          z[it][i] = zh;
          g[it][i] = dyv * gm[it][i];
          sg += g[it][i];
          sgz += g[it][i] * zh;
        }
      }
    }
    const float mg = wave_sum(sg) / (float)d, mgz = wave_sum(sgz) / (float)d;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int c = it * 512 + lane * 8;
      if (c < d) {
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          o[i] = rstd * (g[it][i] - mg - z[it][i] * mgz);
          al[it][i] += o[i];
        }
        st8(dz + row * d + c, o);
      }
    }
  }
  flush_cols(red, ag, dgamma, d, lane, wave);
  flush_cols(red, ab, dbeta, d, lane, wave);
  if (dbias) flush_cols(red, al, dbias, d, lane, wave);
}

// Backward of y = act(x + bias): dx = dy·act'(x + bias) (dx may be NULL: only the bias gradient is wanted; dx == dy allowed),
// dbias += Σ_rows dx.  act 0 identity, 1 exact GELU: d/dv [½ v (1 + erf(v/√2))] = ½(1 + erf(v/√2)) + v·exp(−v²/2)/√(2π).
// dy, x and dx are [rows, n] slices with row stride ld (a column block of a fused buffer); dbias += scale·Σ (the 1/sqrt(d_head) that
// the host folds into the query weights).  A thread owns 8 columns and strides over the rows of its block's row range.
template <typename T>
__global__ __launch_bounds__(256) void bias_act_bwd_kernel(const T* dy, const T* __restrict__ x, const float* __restrict__ bias, T* dx,
                                                           float* __restrict__ dbias, int64_t rows, int n, int64_t ld, int act, float scale) {
  const int cg = blockIdx.y * 256 + threadIdx.x;
  if (cg * 8 >= n) return;
  const int c = cg * 8;
  float b[8], acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  ld8(bias + c, b);
  for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
    float g[8];
    ld8(dy + row * ld + c, g);
    if (act == 1) {
      float v[8];
      ld8(x + row * ld + c, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float t = v[e] + b[e];
        g[e] *= 0.5f * (1.0f + erff(t * 0.70710678118654752440f)) + t * 0.39894228040143267794f * __expf(-0.5f * t * t);
      }
    }
    if (dx) st8(dx + row * ld + c, g);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += g[e];
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) atomicAdd(dbias + c + e, acc[e] * scale);
}

// dx[b, l, :] = dout[b, :] / count_b on the rows that took part in the mean (mask != 0, or all L), 0 elsewhere
template <typename T>
__global__ __launch_bounds__(256) void seq_mean_bwd_kernel(const float* __restrict__ dout, const uint8_t* __restrict__ mask, T* __restrict__ dx,
                                                           int L, int d) {
  const int b = blockIdx.x;
  __shared__ float cnt_s;
  if (threadIdx.x < 64) {
    float c = 0.f;
    for (int l = threadIdx.x; l < L; l += 64) c += (!mask || mask[(int64_t)b * L + l]) ? 1.f : 0.f;
    c = wave_sum(c);
    if (threadIdx.x == 0) cnt_s = c;
  }
  __syncthreads();
  const float inv = 1.0f / cnt_s;
  const int per_row = d / 8;
  for (int i = blockIdx.y * 256 + threadIdx.x; i < L * per_row; i += gridDim.y * 256) {
    const int l = i / per_row, c = (i % per_row) * 8;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (!mask || mask[(int64_t)b * L + l]) {
      ld8(dout + (int64_t)b * d + c, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= inv;
    }
    st8(dx + ((int64_t)b * L + l) * d + c, v);
  }
}

// score[b][i] = scale·<q[b], p[b][i]>:  dp[b][i] = scale·dscore[b][i]·q[b],  dq[b] = scale·Σ_i dscore[b][i]·p[b][i].  One block per b.
__global__ __launch_bounds__(256) void bi_score_bwd_kernel(const float* __restrict__ dscore, const float* __restrict__ q, const float* __restrict__ p,
                                                           float* __restrict__ dq, float* __restrict__ dp, int n, int d, float scale) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < d; c += 256) {
    const float qv = q[(int64_t)b * d + c];
    float acc = 0.f;
    for (int i = 0; i < n; ++i) {
      const float ds = dscore[(int64_t)b * n + i] * scale;
      acc += ds * p[((int64_t)b * n + i) * d + c];
      dp[((int64_t)b * n + i) * d + c] = ds * qv;
    }
    dq[(int64_t)b * d + c] = acc;
  }
}

// loss = 1/(B·n)·Σ xlogy(g, g) − g·log_softmax(score):  dscore[b][i] = upstream/(B·n)·(softmax(score)[b][i]·Σ_j g[b][j] − g[b][i])
__global__ __launch_bounds__(256) void kldiv_bwd_kernel(const float* __restrict__ score, const float* __restrict__ gold, float* __restrict__ dscore,
                                                        const float* __restrict__ upstream, int B, int n) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const float up = (upstream ? upstream[0] : 1.0f) / ((float)B * (float)n);
  float mx = -INFINITY, gs = 0.f;
  for (int i = lane; i < n; i += 64) {
    mx = fmaxf(mx, score[(int64_t)b * n + i]);
    gs += gold[(int64_t)b * n + i];
  }
  mx = wave_max(mx);
  gs = wave_sum(gs);
  float se = 0.f;
  for (int i = lane; i < n; i += 64) se += expf(score[(int64_t)b * n + i] - mx);
  const float inv = 1.0f / wave_sum(se);
  for (int i = lane; i < n; i += 64)
    dscore[(int64_t)b * n + i] = up * (expf(score[(int64_t)b * n + i] - mx) * inv * gs - gold[(int64_t)b * n + i]);
}

// Backward of BertEmbeddings: out = LayerNorm(word[id] + pos[t mod L] + type0)·γ + β.  z and its statistics are recomputed;
// dz is scattered into the word / position rows by float atomics (one 32-byte run per lane), summed over the rows for type0.
template <typename T>
__global__ __launch_bounds__(256) void bert_embed_bwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ word,
                                                             const float* __restrict__ pos, const float* __restrict__ type0,
                                                             const float* __restrict__ gamma, const T* __restrict__ dy, float* __restrict__ dword,
                                                             float* __restrict__ dpos, float* __restrict__ dtype0, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, int64_t n_tok, int L, int d, int64_t vocab, float eps) {
  __shared__ float red[4][LNB_MAXD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t wid = (int64_t)blockIdx.x * 4 + wave, nw = (int64_t)gridDim.x * 4;
  float ag[2][8], ab[2][8], at[2][8], gm[2][8], ty[2][8];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int c = it * 512 + lane * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) ag[it][i] = ab[it][i] = at[it][i] = gm[it][i] = ty[it][i] = 0.f;
    if (c < d) {
      ld8(gamma + c, gm[it]);
      ld8(type0 + c, ty[it]);
    }
  }
  for (int64_t t = wid; t < n_tok; t += nw) {
    int64_t id = ids[t];
    if (id < 0 || id >= vocab) id = 0;
    const int p = (int)(t % L);
    float z[2][8], g[2][8];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int c = it * 512 + lane * 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) z[it][i] = g[it][i] = 0.f;
      if (c < d) {
        float a[8], b[8];
        ld8(word + id * d + c, a);
        ld8(pos + (int64_t)p * d + c, b);
        ld8(dy + t * d + c, g[it]);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          z[it][i] = a[i] + b[i] + ty[it][i];
          s += z[it][i];
        }
      }
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int c = it * 512 + lane * 8;
      if (c < d) {
#pragma unroll
        for (int i = 0; i < 8; ++i) q += (z[it][i] - mean) * (z[it][i] - mean);
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + eps);
    float sg = 0.f, sgz = 0.f;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int c = it * 512 + lane * 8;
      if (c < d) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float zh = (z[it][i] - mean) * rstd, dyv = g[it][i];
          ag[it][i] += dyv * zh;
          ab[it][i] += dyv;
          z[it][i] = zh;
          g[it][i] = dyv * gm[it][i];
          sg += g[it][i];
          sgz += g[it][i] * zh;
        }
      }
    }
    const float mg = wave_sum(sg) / (float)d, mgz = wave_sum(sgz) / (float)d;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int c = it * 512 + lane * 8;
      if (c < d) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float o = rstd * (g[it][i] - mg - z[it][i] * mgz);
          at[it][i] += o;
          atomicAdd(dword + id * d + c + i, o);
          atomicAdd(dpos + (int64_t)p * d + c + i, o);
        }
      }
    }
  }
  flush_cols(red, ag, dgamma, d, lane, wave);
  flush_cols(red, ab, dbeta, d, lane, wave);
  flush_cols(red, at, dtype0, d, lane, wave);
}

inline int rows_blocks(int64_t rows, int cap) {
  int64_t g = (rows + 3) / 4;
  return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}

}  // namespace

#define BERT_DISPATCH(dtype, ...)     \
  do {                                \
    if ((dtype) == LAKO_F32) {        \
      using T = float;                \
      __VA_ARGS__;                    \
    } else {                          \
      using T = bf16_t;               \
      __VA_ARGS__;                    \
    }                                 \
  } while (0)
#define BERT_CHECK_DTYPE(fn, dtype) LAKO_CHECK_ARG((dtype) == LAKO_F32 || (dtype) == LAKO_BF16, fn ": bad dtype %d", (int)(dtype))

extern "C" int lako_layernorm_bwd(const void* dy, const void* x, const float* lin_bias, const void* resid, const float* gamma, void* dz,
                                  float* dgamma, float* dbeta, float* dbias, int64_t rows, int d, float eps, int dtype, lako_stream_t stream) {
  BERT_CHECK_DTYPE("lako_layernorm_bwd", dtype);
  LAKO_CHECK_ARG(dy && x && gamma && dz && dgamma && dbeta && rows > 0 && d > 0 && d % 8 == 0 && d <= LNB_MAXD,
                 "lako_layernorm_bwd: rows=%lld d=%d (d %% 8 == 0, <= %d)", (long long)rows, d, LNB_MAXD);
  LAKO_CHECK_ARG(!dbias || lin_bias, "lako_layernorm_bwd: dbias without lin_bias");
  LAKO_CHECK_ALIGN(dy, 16);
  LAKO_CHECK_ALIGN(x, 16);
  LAKO_CHECK_ALIGN(dz, 16);
  BERT_DISPATCH(dtype, hipLaunchKernelGGL((layernorm_bwd_kernel<T>), dim3(rows_blocks(rows, 512)), dim3(256), 0, (hipStream_t)stream, (const T*)dy,
                                          (const T*)x, lin_bias, (const T*)resid, gamma, (T*)dz, dgamma, dbeta, dbias, rows, d, eps));
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

extern "C" int lako_bias_act_bwd(const void* dy, const void* x, const float* bias, void* dx, float* dbias, int64_t rows, int n, int64_t ld,
                                 int act, float dbias_scale, int dtype, lako_stream_t stream) {
  BERT_CHECK_DTYPE("lako_bias_act_bwd", dtype);
  LAKO_CHECK_ARG(dy && bias && dbias && rows > 0 && n > 0 && n % 8 == 0 && ld >= n && ld % 8 == 0 && (act == 0 || (act == 1 && x)),
                 "lako_bias_act_bwd: rows=%lld n=%d ld=%lld act=%d", (long long)rows, n, (long long)ld, act);
  LAKO_CHECK_ALIGN(dy, 16);
  int64_t gx = rows < 256 ? rows : 256;
  BERT_DISPATCH(dtype, hipLaunchKernelGGL((bias_act_bwd_kernel<T>), dim3((unsigned)gx, (unsigned)((n / 8 + 255) / 256)), dim3(256), 0,
                                          (hipStream_t)stream, (const T*)dy, (const T*)x, bias, (T*)dx, dbias, rows, n, ld, act, dbias_scale));
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

extern "C" int lako_seq_mean_bwd(const float* dout, const uint8_t* mask, void* dx, int B, int L, int d, int dtype, lako_stream_t stream) {
  BERT_CHECK_DTYPE("lako_seq_mean_bwd", dtype);
  LAKO_CHECK_ARG(dout && dx && B > 0 && L > 0 && d > 0 && d % 8 == 0, "lako_seq_mean_bwd: B=%d L=%d d=%d", B, L, d);
  LAKO_CHECK_ALIGN(dx, 16);
  LAKO_CHECK_ALIGN(dout, 16);
  int gy = (L * (d / 8) + 255) / 256;
  if (gy > 64) gy = 64;
  BERT_DISPATCH(dtype, hipLaunchKernelGGL((seq_mean_bwd_kernel<T>), dim3((unsigned)B, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, dout,
                                          mask, (T*)dx, L, d));
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

extern "C" int lako_bi_score_bwd(const float* dscore, const float* q, const float* p, float* dq, float* dp, int B, int n, int d, float scale,
                                 lako_stream_t stream) {
  LAKO_CHECK_ARG(dscore && q && p && dq && dp && B > 0 && n > 0 && d > 0, "lako_bi_score_bwd: B=%d n=%d d=%d", B, n, d);
  hipLaunchKernelGGL(bi_score_bwd_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, dscore, q, p, dq, dp, n, d, scale);
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

extern "C" int lako_kldiv_bwd(const float* score, const float* gold, float* dscore, const float* upstream, int B, int n, lako_stream_t stream) {
  LAKO_CHECK_ARG(score && gold && dscore && B > 0 && n > 0, "lako_kldiv_bwd: B=%d n=%d", B, n);
  hipLaunchKernelGGL(kldiv_bwd_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, score, gold, dscore, upstream, B, n);
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

extern "C" int lako_bert_embed_bwd(const int64_t* ids, const float* word, const float* pos, const float* type0, const float* gamma,
                                   const void* dy, float* dword, float* dpos, float* dtype0, float* dgamma, float* dbeta, int64_t n_tok, int L,
                                   int d, int64_t vocab, float eps, int dtype, lako_stream_t stream) {
  BERT_CHECK_DTYPE("lako_bert_embed_bwd", dtype);
  LAKO_CHECK_ARG(ids && word && pos && type0 && gamma && dy && dword && dpos && dtype0 && dgamma && dbeta && n_tok > 0 && L > 0 && d > 0 &&
                     d % 8 == 0 && d <= LNB_MAXD && vocab > 0,
                 "lako_bert_embed_bwd: bad arguments (d %% 8 == 0, <= %d)", LNB_MAXD);
  LAKO_CHECK_ALIGN(dy, 16);
  BERT_DISPATCH(dtype, hipLaunchKernelGGL((bert_embed_bwd_kernel<T>), dim3(rows_blocks(n_tok, 512)), dim3(256), 0, (hipStream_t)stream, ids, word,
                                          pos, type0, gamma, (const T*)dy, dword, dpos, dtype0, dgamma, dbeta, n_tok, L, d, vocab, eps));
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}
