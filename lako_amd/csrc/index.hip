// Exact inner-product search, the GPU side of SURVEY.md §8 f4 (reference: src/index.py:19-50 — faiss.IndexFlatIP —
// and the per-example re-ranking of fact_retrieval_small_range.py:64-89).  The scores are one fp32 GEMM
// (lako_gemm_nt: queries [nq, d] × embeddings [n, d]ᵀ); this file holds the per-row top-k.
//
// lako_topk: one workgroup per row.  HBM-bound integer work — no MFMA: the row is read 5 times from L2/HBM
// (4 radix-select passes of 8 bits over the order-preserving integer image of the floats + 1 compaction pass),
// 16 B per lane per load; the k survivors are sorted in LDS (bitonic, 64-bit keys = value ≫ index).
// Ties: equal scores come out in ascending index order (deterministic; faiss leaves the order unspecified).
#include "common.h"

namespace {

constexpr int TOPK_MAX = 1024;
constexpr int TOPK_THREADS = 1024;

// order-preserving map float → uint32 (larger float ⇒ larger uint)
__device__ __forceinline__ uint32_t f2key(float f) {
  const uint32_t b = __builtin_bit_cast(uint32_t, f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
  const uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __builtin_bit_cast(float, b);
}

__global__ __launch_bounds__(TOPK_THREADS) void topk_kernel(const float* __restrict__ scores, int64_t n, int64_t ld,
                                                             int k, float* __restrict__ out_vals,
                                                             int64_t* __restrict__ out_idx) {
  __shared__ uint32_t hist[256];
  __shared__ uint64_t sel[TOPK_MAX];        // (key << 32) | ~index : descending sort = value desc, index asc
  __shared__ uint32_t eq_idx[TOPK_MAX];     // indices of elements equal to the threshold (the lowest `need` are taken)
  __shared__ uint32_t s_prefix, s_mask, s_need, s_cnt, s_eq;
  const float* row = scores + (int64_t)blockIdx.x * ld;
  const int tid = threadIdx.x;
  if (tid == 0) {
    s_prefix = 0;
    s_mask = 0;
    s_need = (uint32_t)k;
  }
  // ---- radix select: the k-th largest key, 8 bits at a time from the top ------------------------------------------
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (int i = tid; i < 256; i += TOPK_THREADS) hist[i] = 0;
    __syncthreads();
    const uint32_t prefix = s_prefix, mask = s_mask;
    for (int64_t i = (int64_t)tid * 4; i < n; i += (int64_t)TOPK_THREADS * 4) {
      float v[4];
      if (i + 3 < n && (ld % 4 == 0)) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(row + i);
        v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = i + e < n ? row[i + e] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (i + e >= n) break;
        const uint32_t key = f2key(v[e]);
        if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
      }
    }
    __syncthreads();
    if (tid == 0) {      // walk the bins from the top until the k-th element falls inside one
      uint32_t need = s_need, d = 255;
      for (;; --d) {
        if (hist[d] >= need || d == 0) break;
        need -= hist[d];
      }
      s_prefix = prefix | (d << shift);
      s_mask = mask | (255u << shift);
      s_need = need;     // how many elements with this digit (finally: equal to the threshold) are still wanted
    }
    __syncthreads();
  }
  const uint32_t thr = s_prefix, need_eq = s_need;
  if (tid == 0) {
    s_cnt = 0;
    s_eq = 0;
  }
  __syncthreads();
  // ---- compaction: everything above the threshold, and the candidates equal to it ------------------------------------
  for (int64_t i = tid; i < n; i += TOPK_THREADS) {
    const uint32_t key = f2key(row[i]);
    if (key > thr) {
      const uint32_t p = atomicAdd(&s_cnt, 1u);
      sel[p] = ((uint64_t)key << 32) | (uint32_t)~(uint32_t)i;
    } else if (key == thr) {
      const uint32_t p = atomicAdd(&s_eq, 1u);
      if (p < (uint32_t)TOPK_MAX) eq_idx[p] = (uint32_t)i;
    }
  }
  __syncthreads();
  const uint32_t n_gt = s_cnt, n_eq = s_eq;
  if (n_eq <= (uint32_t)TOPK_MAX) {
    // the `need_eq` LOWEST indices among the ties: rank each candidate among the collected ones (ties are few)
    for (uint32_t c = tid; c < n_eq; c += TOPK_THREADS) {
      const uint32_t me = eq_idx[c];
      uint32_t rank = 0;
      for (uint32_t o = 0; o < n_eq; ++o) rank += eq_idx[o] < me;
      if (rank < need_eq) sel[n_gt + rank] = ((uint64_t)thr << 32) | (uint32_t)~me;
    }
  } else {
    // more ties than the candidate buffer holds (constant rows): walk the row in index order, 1024 elements at a time,
    // with a block-wide count of the ties seen so far, until `need_eq` of them are taken
    __shared__ uint32_t wave_cnt[TOPK_THREADS / 64];
    uint32_t base = 0;
    for (int64_t c0 = 0; c0 < n && base < need_eq; c0 += TOPK_THREADS) {
      const int64_t i = c0 + tid;
      const bool is = i < n && f2key(row[i]) == thr;
      const uint64_t bal = __ballot(is);
      const int lane = tid & 63, wv = tid >> 6;
      if (lane == 0) wave_cnt[wv] = (uint32_t)__popcll(bal);
      __syncthreads();
      uint32_t before = 0, total = 0;
      for (int w = 0; w < TOPK_THREADS / 64; ++w) {
        before += w < wv ? wave_cnt[w] : 0u;
        total += wave_cnt[w];
      }
      const uint32_t rank = base + before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
      if (is && rank < need_eq) sel[n_gt + rank] = ((uint64_t)thr << 32) | (uint32_t)~(uint32_t)i;
      base += total;
      __syncthreads();
    }
  }
  for (int i = (int)n_gt + (int)need_eq + tid; i < TOPK_MAX; i += TOPK_THREADS) sel[i] = 0ull;   // padding sorts last
  __syncthreads();
  // ---- bitonic sort of the TOPK_MAX slots, descending ---------------------------------------------------------------------
  for (int size = 2; size <= TOPK_MAX; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const int i = tid, j = i ^ stride;
      if (j > i) {
        const uint64_t a = sel[i], b = sel[j];
        const bool desc = (i & size) == 0;
        if (desc ? a < b : a > b) {
          sel[i] = b;
          sel[j] = a;
        }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < k; i += TOPK_THREADS) {
    const uint64_t e = sel[i];
    out_vals[(int64_t)blockIdx.x * k + i] = key2f((uint32_t)(e >> 32));
    out_idx[(int64_t)blockIdx.x * k + i] = (int64_t)(uint32_t)~(uint32_t)e;
  }
}

}  // namespace

extern "C" int lako_topk(const float* scores, int64_t rows, int64_t n, int64_t ld, int k, float* out_vals,
                         int64_t* out_idx, lako_stream_t stream) {
  LAKO_CHECK_ARG(scores && out_vals && out_idx, "lako_topk: null pointer");
  LAKO_CHECK_ARG(rows > 0 && n > 0 && ld >= n && n < (1ll << 31), "lako_topk: bad dims rows=%lld n=%lld ld=%lld", (long long)rows,
                 (long long)n, (long long)ld);
  LAKO_CHECK_ARG(k >= 1 && k <= TOPK_MAX && k <= n, "lako_topk: k must be in [1, min(n, %d)]", TOPK_MAX);
  LAKO_CHECK_ALIGN(scores, 16);
  hipLaunchKernelGGL(topk_kernel, dim3((unsigned)rows), dim3(TOPK_THREADS), 0, (hipStream_t)stream, scores, n, ld, k, out_vals,
                     out_idx);
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}
