// Row-wise kernels of the retriever's BERT bi-encoder, FORWARD (inference) only — SURVEY.md §8 f4:
// src/model.py:375-483 (`Retriever.embed_text` / `forward`) on top of HF `BertModel` (modeling_bert.py: embeddings =
// LayerNorm(word + position + token_type), post-LN blocks, biased linears, exact erf GELU).  The matrix products and the
// attention are the reader's kernels (lako_gemm_nt, lako_attn_fwd: the 1/sqrt(d_head) score scale is folded into the query
// weights by the host).  All of these are HBM-bound: one wave per row, 16-byte accesses, wave64 shuffle reductions.
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

constexpr int LN_MAXD = 1024;     // one wave holds a row: 64 lanes × 8 × 2 iterations

// y = LayerNorm(x + lin_bias + resid) * gamma + beta   (torch.nn.LayerNorm: biased variance, eps inside the sqrt)
template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, const float* __restrict__ lin_bias,
                                                        const T* __restrict__ resid, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, T* __restrict__ y, int64_t rows, int d,
                                                        float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
  for (int64_t row = wid; row < rows; row += nw) {
    float v[2][8];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int c = it * 512 + lane * 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) v[it][i] = 0.f;
      if (c < d) {
        ld8(x + row * d + c, v[it]);
        if (lin_bias) {
          float b[8];
          ld8(lin_bias + c, b);
#pragma unroll
          for (int i = 0; i < 8; ++i) v[it][i] += b[i];
        }
        if (resid) {
          float r[8];
          ld8(resid + row * d + c, r);
#pragma unroll
          for (int i = 0; i < 8; ++i) v[it][i] += r[i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[it][i];
      }
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int c = it * 512 + lane * 8;
      if (c < d) {
#pragma unroll
        for (int i = 0; i < 8; ++i) q += (v[it][i] - mean) * (v[it][i] - mean);
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + eps);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int c = it * 512 + lane * 8;
      if (c < d) {
        float g[8], b[8], o[8];
        ld8(gamma + c, g);
        ld8(beta + c, b);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (v[it][i] - mean) * rstd * g[i] + b[i];
        st8(y + row * d + c, o);
      }
    }
  }
}

// BertEmbeddings: out = LayerNorm(word[ids[t]] + pos[t mod L] + type0) (token_type_ids = 0, absolute positions 0..L-1)
template <typename T>
__global__ __launch_bounds__(256) void bert_embed_kernel(const int64_t* __restrict__ ids, const float* __restrict__ word,
                                                         const float* __restrict__ pos, const float* __restrict__ type0,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         T* __restrict__ out, int64_t n_tok, int L, int d, int64_t vocab,
                                                         float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
  for (int64_t t = wid; t < n_tok; t += nw) {
    int64_t id = ids[t];
    if (id < 0 || id >= vocab) id = 0;
    const int p = (int)(t % L);
    float v[2][8];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int c = it * 512 + lane * 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) v[it][i] = 0.f;
      if (c < d) {
        float a[8], b[8], e[8];
        ld8(word + id * d + c, a);
        ld8(pos + (int64_t)p * d + c, b);
        ld8(type0 + c, e);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          v[it][i] = a[i] + b[i] + e[i];
          s += v[it][i];
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
        for (int i = 0; i < 8; ++i) q += (v[it][i] - mean) * (v[it][i] - mean);
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + eps);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int c = it * 512 + lane * 8;
      if (c < d) {
        float g[8], b[8], o[8];
        ld8(gamma + c, g);
        ld8(beta + c, b);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (v[it][i] - mean) * rstd * g[i] + b[i];
        st8(out + t * d + c, o);
      }
    }
  }
}

// y = act(x + bias): act 0 = identity, 1 = GELU (exact: 0.5·v·(1 + erf(v/√2)), HF "gelu")
template <typename T>
__global__ __launch_bounds__(256) void bias_act_kernel(const T* x, const float* __restrict__ bias, T* y,   // x == y allowed
                                                       int64_t rows, int n, int act) {
  const int64_t total8 = rows * (int64_t)(n / 8);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / (n / 8);
    const int c = (int)(i % (n / 8)) * 8;
    float v[8], b[8];
    ld8(x + row * n + c, v);
    ld8(bias + c, b);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = v[e] + b[e];
      if (act == 1) t = 0.5f * t * (1.0f + erff(t * 0.70710678118654752440f));
      v[e] = t;
    }
    st8(y + row * n + c, v);
  }
}

// out[b] (fp32 [d]) = mean over the L rows of x[b] — over the rows with mask != 0 when a mask is given
// (src/model.py:471-476: masked_fill + sum / mask.sum; no mask: torch.mean).  One workgroup per (b, 8-column group of 32).
template <typename T>
__global__ __launch_bounds__(256) void seq_mean_kernel(const T* __restrict__ x, const uint8_t* __restrict__ mask, float* __restrict__ out,
                                                       int L, int d) {
  const int b = blockIdx.x;
  const int cg = threadIdx.x & 31, rg = threadIdx.x >> 5;      // 32 column groups of 8 per pass × 8 row groups
  __shared__ float red[8][32][8];
  __shared__ float cnt_s[8];
  for (int c0 = blockIdx.y * 256; c0 < d; c0 += gridDim.y * 256) {
    const int c = c0 + cg * 8;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float cnt = 0.f;
    for (int r = rg; r < L; r += 8) {
      const bool use = !mask || mask[(int64_t)b * L + r];
      if (use) {
        cnt += 1.f;
        if (c < d) {
          float v[8];
          ld8(x + ((int64_t)b * L + r) * d + c, v);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += v[e];
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rg][cg][e] = acc[e];
    if (cg == 0) cnt_s[rg] = cnt;
    __syncthreads();
    if (rg == 0 && c < d) {
      float n = 0.f;
      for (int k = 0; k < 8; ++k) n += cnt_s[k];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = 0.f;
        for (int k = 0; k < 8; ++k) t += red[k][cg][e];
        out[(int64_t)b * d + c + e] = t / n;
      }
    }
    __syncthreads();
  }
}

// score[b][i] = scale · Σ_c q[b][c] · p[b][i][c]   (src/model.py:443-448: einsum('bd,bid->bi') / sqrt(d)); one wave per (b, i)
__global__ __launch_bounds__(256) void bi_score_kernel(const float* __restrict__ q, const float* __restrict__ p, float* __restrict__ out,
                                                       int B, int n, int d, float scale) {
  const int lane = threadIdx.x & 63;
  const int64_t pair = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= (int64_t)B * n) return;
  const int b = (int)(pair / n);
  float s = 0.f;
  for (int c = lane * 4; c < d; c += 256) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(q + (int64_t)b * d + c);
    const f32x4 w = *reinterpret_cast<const f32x4*>(p + pair * d + c);
    s += a[0] * w[0] + a[1] * w[1] + a[2] * w[2] + a[3] * w[3];
  }
  s = wave_sum(s);
  if (lane == 0) out[pair] = s * scale;
}

// torch.nn.KLDivLoss() (reduction 'mean' = over all B·n elements) of log_softmax(score) against gold (src/model.py:480-483):
// loss = 1/(B·n) · Σ xlogy(g, g) − g·(score − logsumexp(score)).  One workgroup, one wave per row, fixed summation order.
__global__ __launch_bounds__(256) void kldiv_kernel(const float* __restrict__ score, const float* __restrict__ gold, float* __restrict__ loss,
                                                    int B, int n) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __shared__ float part[4];
  float acc = 0.f;
  for (int b = w; b < B; b += 4) {
    float mx = -INFINITY;
    for (int i = lane; i < n; i += 64) mx = fmaxf(mx, score[(int64_t)b * n + i]);
    mx = wave_max(mx);
    float se = 0.f;
    for (int i = lane; i < n; i += 64) se += expf(score[(int64_t)b * n + i] - mx);
    const float lse = mx + logf(wave_sum(se));
    float t = 0.f;
    for (int i = lane; i < n; i += 64) {
      const float g = gold[(int64_t)b * n + i];
      const float lp = score[(int64_t)b * n + i] - lse;
      t += (g > 0.f ? g * logf(g) : 0.f) - g * lp;
    }
    acc += wave_sum(t);
  }
  if (lane == 0) part[w] = acc;
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = (part[0] + part[1] + part[2] + part[3]) / ((float)B * (float)n);
}

inline int rows_blocks(int64_t rows) {
  int64_t g = (rows + 3) / 4;
  return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
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

extern "C" int lako_layernorm_fwd(const void* x, const float* lin_bias, const void* resid, const float* gamma, const float* beta,
                                  void* y, int64_t rows, int d, float eps, int dtype, lako_stream_t stream) {
  BERT_CHECK_DTYPE("lako_layernorm_fwd", dtype);
  LAKO_CHECK_ARG(x && gamma && beta && y && rows > 0 && d > 0 && d % 8 == 0 && d <= LN_MAXD, "lako_layernorm_fwd: rows=%lld d=%d (d %% 8 == 0, <= %d)",
                 (long long)rows, d, LN_MAXD);
  LAKO_CHECK_ALIGN(x, 16);
  LAKO_CHECK_ALIGN(y, 16);
  BERT_DISPATCH(dtype, hipLaunchKernelGGL((layernorm_kernel<T>), dim3(rows_blocks(rows)), dim3(256), 0, (hipStream_t)stream, (const T*)x,
                                          lin_bias, (const T*)resid, gamma, beta, (T*)y, rows, d, eps));
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

extern "C" int lako_bert_embed(const int64_t* ids, const float* word, const float* pos, const float* type0, const float* gamma,
                               const float* beta, void* out, int64_t n_tok, int L, int d, int64_t vocab, float eps, int dtype,
                               lako_stream_t stream) {
  BERT_CHECK_DTYPE("lako_bert_embed", dtype);
  LAKO_CHECK_ARG(ids && word && pos && type0 && gamma && beta && out && n_tok > 0 && L > 0 && d > 0 && d % 8 == 0 && d <= LN_MAXD && vocab > 0,
                 "lako_bert_embed: bad arguments (d %% 8 == 0, <= %d)", LN_MAXD);
  LAKO_CHECK_ALIGN(out, 16);
  BERT_DISPATCH(dtype, hipLaunchKernelGGL((bert_embed_kernel<T>), dim3(rows_blocks(n_tok)), dim3(256), 0, (hipStream_t)stream, ids, word, pos,
                                          type0, gamma, beta, (T*)out, n_tok, L, d, vocab, eps));
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

extern "C" int lako_bias_act(const void* x, const float* bias, void* y, int64_t rows, int n, int act, int dtype, lako_stream_t stream) {
  BERT_CHECK_DTYPE("lako_bias_act", dtype);
  LAKO_CHECK_ARG(x && bias && y && rows > 0 && n > 0 && n % 8 == 0 && (act == 0 || act == 1), "lako_bias_act: rows=%lld n=%d act=%d", (long long)rows, n, act);
  LAKO_CHECK_ALIGN(x, 16);
  LAKO_CHECK_ALIGN(y, 16);
  int64_t blocks = (rows * (n / 8) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  BERT_DISPATCH(dtype, hipLaunchKernelGGL((bias_act_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const T*)x, bias,
                                          (T*)y, rows, n, act));
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

extern "C" int lako_seq_mean(const void* x, const uint8_t* mask, float* out, int B, int L, int d, int dtype, lako_stream_t stream) {
  BERT_CHECK_DTYPE("lako_seq_mean", dtype);
  LAKO_CHECK_ARG(x && out && B > 0 && L > 0 && d > 0 && d % 8 == 0, "lako_seq_mean: B=%d L=%d d=%d", B, L, d);
  LAKO_CHECK_ALIGN(x, 16);
  BERT_DISPATCH(dtype, hipLaunchKernelGGL((seq_mean_kernel<T>), dim3((unsigned)B, (unsigned)((d + 255) / 256)), dim3(256), 0,
                                          (hipStream_t)stream, (const T*)x, mask, out, L, d));
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

extern "C" int lako_bi_score(const float* q, const float* p, float* out, int B, int n, int d, float scale, lako_stream_t stream) {
  LAKO_CHECK_ARG(q && p && out && B > 0 && n > 0 && d > 0 && d % 4 == 0, "lako_bi_score: B=%d n=%d d=%d", B, n, d);
  LAKO_CHECK_ALIGN(q, 16);
  LAKO_CHECK_ALIGN(p, 16);
  hipLaunchKernelGGL(bi_score_kernel, dim3((unsigned)(((int64_t)B * n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, q, p, out, B, n, d, scale);
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

extern "C" int lako_kldiv_fwd(const float* score, const float* gold, float* loss, int B, int n, lako_stream_t stream) {
  LAKO_CHECK_ARG(score && gold && loss && B > 0 && n > 0, "lako_kldiv_fwd: B=%d n=%d", B, n);
  hipLaunchKernelGGL(kldiv_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, score, gold, loss, B, n);
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}
