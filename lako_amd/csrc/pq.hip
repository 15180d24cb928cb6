// Product-quantised inner-product search — the GPU side of `faiss.IndexPQ(d, M, nbits, METRIC_INNER_PRODUCT)` behind
// src/index.py:21-23 (`Indexer(vector_sz, n_subquantizers > 0, n_bits)`; LaKo's own drivers construct the flat index,
// fact_retrieval.py:149-150).  faiss is absent from the image: these kernels follow its published definition —
//   * a vector is M sub-vectors of dsub = d / M floats; sub-quantiser m owns ksub = 2^nbits centroids [ksub, dsub], trained by
//     k-means on the sub-vectors; a vector's code is, per m, the index of the nearest centroid in L2 (lowest index on ties);
//   * search (asymmetric distance computation): per query a table LUT[m][c] = <q_m, centroid[m][c]>, the score of a stored
//     vector is Σ_m LUT[m][code_m] summed in ascending m in fp32, the result the k largest scores.
// HBM / LDS-gather bound byte work, no MFMA: the codes are streamed once per group of queries (16 B per lane per load), the
// tables of QT queries sit in LDS, interleaved by query, and a lookup of four queries is one `ds_read_b128` at a data-dependent bank.  The k-means loop around
// lako_pq_assign (initialisation, empty clusters, convergence) is host logic in lako_amd/index.py.
#include "common.h"

namespace {

constexpr int PQ_MAX_DSUB = 64;
constexpr int PQ_LDS_BYTES = 128 * 1024;
constexpr int PQ_SCAN_THREADS = 1024;      // one workgroup per CU (its tables fill the LDS): 16 waves to cover the gathers' latency

// One thread per (vector, sub-quantiser): the centroids of sub-quantiser m in LDS (every lane reads the same address: broadcast),
// the sub-vector in registers.  Optionally adds the sub-vector to its centroid's running sum (the k-means update) and the squared
// distance to *err.
template <int DS>
__global__ __launch_bounds__(256) void pq_assign_kernel(const float* __restrict__ x, int64_t n, int64_t ldx,
                                                        const float* __restrict__ cent, int M, int ksub, uint8_t* __restrict__ codes,
                                                        float* __restrict__ sums, int32_t* __restrict__ counts, float* __restrict__ err) {
  extern __shared__ __attribute__((aligned(16))) float cs[];     // [ksub][DS]
  const int m = blockIdx.y;
  const float* cm = cent + (int64_t)m * ksub * DS;
  for (int i = threadIdx.x; i < ksub * DS; i += 256) cs[i] = cm[i];
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float best = INFINITY;
  int bi = 0;
  float v[DS];
  if (i < n) {
#pragma unroll
    for (int j = 0; j < DS; ++j) v[j] = x[i * ldx + (int64_t)m * DS + j];
    for (int c = 0; c < ksub; ++c) {
      float dist = 0.f;
#pragma unroll
      for (int j = 0; j < DS; ++j) {
        const float t = v[j] - cs[c * DS + j];
        dist += t * t;
      }
      if (dist < best) {      // strict: the lowest index wins a tie
        best = dist;
        bi = c;
      }
    }
    if (codes) codes[i * M + m] = (uint8_t)bi;
    if (sums) {
      float* s = sums + ((int64_t)m * ksub + bi) * DS;
#pragma unroll
      for (int j = 0; j < DS; ++j) atomicAdd(s + j, v[j]);
      atomicAdd(counts + m * ksub + bi, 1);
    }
  }
  if (err) {
    float e = i < n ? best : 0.f;
    e = wave_sum(e);
    if ((threadIdx.x & 63) == 0) atomicAdd(err, e);
  }
}

// LUT[q][m][c] = <q_m, centroid[m][c]>: a workgroup per (query, sub-quantiser), a thread per centroid
__global__ __launch_bounds__(256) void pq_lut_kernel(const float* __restrict__ q, int64_t ldq, const float* __restrict__ cent, int M,
                                                     int ksub, int dsub, float* __restrict__ lut) {
  __shared__ float qs[PQ_MAX_DSUB];
  const int iq = blockIdx.x, m = blockIdx.y;
  if (threadIdx.x < dsub) qs[threadIdx.x] = q[(int64_t)iq * ldq + (int64_t)m * dsub + threadIdx.x];
  __syncthreads();
  for (int c = threadIdx.x; c < ksub; c += 256) {
    const float* cc = cent + ((int64_t)m * ksub + c) * dsub;
    float acc = 0.f;
    for (int j = 0; j < dsub; ++j) acc += qs[j] * cc[j];
    lut[((int64_t)iq * M + m) * ksub + c] = acc;
  }
}

// scores[q][i] = Σ_m LUT[q][m][codes[i][m]] (ascending m, fp32).  The tables of QT queries in LDS, INTERLEAVED by query
// (ls[(m·ksub + c)·QT + q]): the QT entries one code selects are adjacent, so a lane fetches four queries' entries with one
// ds_read_b128 — the scan is bound by LDS gathers at data-dependent banks, and this quarters their number (measured, 1024 queries ×
// 300 600 codes, with 1024-thread workgroups instead of 256: M = 16 1 675 → 653 µs, M = 32 2 340 → 943, M = 64 4 250 → 1 993;
// profiles/r03m_bench_ops_index.txt; the exact fp32 GEMM of the flat index takes 1 540 µs).  A thread per stored vector, whose M code bytes are
// consecutive (16-byte loads when M % 16 == 0); a wave writes 256 consecutive bytes of one score row.
template <int QT>
__global__ __launch_bounds__(PQ_SCAN_THREADS) void pq_scan_kernel(const float* __restrict__ lut, const uint8_t* __restrict__ codes, int64_t n,
                                                      int64_t nq, int M, int ksub, float* __restrict__ scores, int64_t ld,
                                                      int per_block) {
  extern __shared__ __attribute__((aligned(16))) float ls[];     // [M·ksub][QT]
  const int64_t q0 = (int64_t)blockIdx.y * QT;
  const int tab = M * ksub;
  for (int t = threadIdx.x; t < QT * tab; t += PQ_SCAN_THREADS) {            // (coalesced reads of each query's table, strided LDS writes)
    const int qi = t / tab, o = t % tab;
    ls[o * QT + qi] = q0 + qi < nq ? lut[(q0 + qi) * tab + o] : 0.f;
  }
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * per_block;
  const int64_t end = base + per_block < n ? base + per_block : n;
  for (int64_t i = base + threadIdx.x; i < end; i += PQ_SCAN_THREADS) {
    const uint8_t* c = codes + i * M;
    float acc[QT];
#pragma unroll
    for (int qi = 0; qi < QT; ++qi) acc[qi] = 0.f;
    auto add = [&](int o) {
      if constexpr (QT >= 4) {
#pragma unroll
        for (int h = 0; h < QT / 4; ++h) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(ls + o * QT + 4 * h);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[4 * h + r] += t[r];
        }
      } else {
#pragma unroll
        for (int qi = 0; qi < QT; ++qi) acc[qi] += ls[o * QT + qi];
      }
    };
    int m = 0;
    if ((M & 15) == 0) {
      for (; m < M; m += 16) {
        const u32x4 w = *reinterpret_cast<const u32x4*>(c + m);
#pragma unroll
        for (int b = 0; b < 16; ++b) add((m + b) * ksub + (int)((w[b >> 2] >> (8 * (b & 3))) & 0xff));
      }
    } else {
      for (; m < M; ++m) add(m * ksub + c[m]);
    }
#pragma unroll
    for (int qi = 0; qi < QT; ++qi)
      if (q0 + qi < nq) scores[(q0 + qi) * ld + i] = acc[qi];
  }
}

template <int DS>
void launch_assign(const float* x, int64_t n, int64_t ldx, const float* cent, int M, int ksub, uint8_t* codes, float* sums,
                   int32_t* counts, float* err, hipStream_t s) {
  LAKO_SET_MAX_LDS((&pq_assign_kernel<DS>), 256 * DS * 4);
  hipLaunchKernelGGL((pq_assign_kernel<DS>), dim3((unsigned)((n + 255) / 256), (unsigned)M), dim3(256), (size_t)ksub * DS * 4, s, x, n, ldx,
                     cent, M, ksub, codes, sums, counts, err);
}

template <int QT>
void launch_scan(const float* lut, const uint8_t* codes, int64_t n, int64_t nq, int M, int ksub, float* scores, int64_t ld, hipStream_t s) {
  LAKO_SET_MAX_LDS((&pq_scan_kernel<QT>), PQ_LDS_BYTES);
  // ≥ 4 blocks per CU along the stored vectors when there are few query groups, whole multiples of 256 vectors per block
  const int64_t qgroups = (nq + QT - 1) / QT;
  int64_t blocks = (n + 255) / 256;
  const int64_t want = qgroups >= 1024 ? 1 : (1024 + qgroups - 1) / qgroups;
  if (blocks > want) blocks = want;
  const int64_t per_block = ((n + blocks - 1) / blocks + 255) / 256 * 256;
  blocks = (n + per_block - 1) / per_block;
  hipLaunchKernelGGL((pq_scan_kernel<QT>), dim3((unsigned)blocks, (unsigned)qgroups), dim3(PQ_SCAN_THREADS), (size_t)QT * M * ksub * 4, s, lut, codes, n, nq,
                     M, ksub, scores, ld, (int)per_block);
}

}  // namespace

extern "C" int lako_pq_assign(const float* x, int64_t n, int64_t ldx, const float* centroids, int M, int ksub, int dsub, uint8_t* codes,
                              float* sums, int32_t* counts, float* err, lako_stream_t stream) {
  LAKO_CHECK_ARG(x && centroids && n > 0 && M > 0 && M <= 65535 && ksub >= 1 && ksub <= 256 && dsub >= 1 && ldx >= (int64_t)M * dsub,
                 "lako_pq_assign: n=%lld M=%d ksub=%d dsub=%d ldx=%lld", (long long)n, M, ksub, dsub, (long long)ldx);
  LAKO_CHECK_ARG(codes || sums, "lako_pq_assign: neither codes nor sums requested");
  LAKO_CHECK_ARG(!sums == !counts, "lako_pq_assign: sums and counts go together");
  hipStream_t s = (hipStream_t)stream;
  switch (dsub) {
#define PQ_CASE(D) \
  case D: launch_assign<D>(x, n, ldx, centroids, M, ksub, codes, sums, counts, err, s); break;
    PQ_CASE(1) PQ_CASE(2) PQ_CASE(4) PQ_CASE(6) PQ_CASE(8) PQ_CASE(12) PQ_CASE(16) PQ_CASE(24) PQ_CASE(32) PQ_CASE(48) PQ_CASE(64)
#undef PQ_CASE
    default:
      LAKO_CHECK_ARG(false, "lako_pq_assign: sub-vector length %d not one of 1, 2, 4, 6, 8, 12, 16, 24, 32, 48, 64", dsub);
  }
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

extern "C" int lako_pq_lut(const float* q, int64_t nq, int64_t ldq, const float* centroids, int M, int ksub, int dsub, float* lut,
                           lako_stream_t stream) {
  LAKO_CHECK_ARG(q && centroids && lut && nq > 0 && nq < (1ll << 31) && M > 0 && M <= 65535 && ksub >= 1 && ksub <= 256 && dsub >= 1 &&
                     dsub <= PQ_MAX_DSUB && ldq >= (int64_t)M * dsub,
                 "lako_pq_lut: nq=%lld M=%d ksub=%d dsub=%d (dsub <= %d)", (long long)nq, M, ksub, dsub, PQ_MAX_DSUB);
  hipLaunchKernelGGL(pq_lut_kernel, dim3((unsigned)nq, (unsigned)M), dim3(256), 0, (hipStream_t)stream, q, ldq, centroids, M, ksub, dsub, lut);
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

extern "C" int lako_pq_scan(const float* lut, const uint8_t* codes, int64_t n, int64_t nq, int M, int ksub, float* scores, int64_t ld,
                            lako_stream_t stream) {
  LAKO_CHECK_ARG(lut && codes && scores && n > 0 && nq > 0 && M > 0 && ksub >= 1 && ksub <= 256 && ld >= n,
                 "lako_pq_scan: n=%lld nq=%lld M=%d ksub=%d ld=%lld", (long long)n, (long long)nq, M, ksub, (long long)ld);
  const int64_t tab = (int64_t)M * ksub * 4;
  LAKO_CHECK_ARG(tab <= PQ_LDS_BYTES, "lako_pq_scan: one query's table (M * ksub * 4 = %lld bytes) must fit %d bytes of LDS", (long long)tab,
                 PQ_LDS_BYTES);
  if ((M & 15) == 0) LAKO_CHECK_ALIGN(codes, 16);
  hipStream_t s = (hipStream_t)stream;
  const int fit = (int)(PQ_LDS_BYTES / tab);
  if (fit >= 8 && nq >= 8) launch_scan<8>(lut, codes, n, nq, M, ksub, scores, ld, s);
  else if (fit >= 4 && nq >= 4) launch_scan<4>(lut, codes, n, nq, M, ksub, scores, ld, s);
  else if (fit >= 2 && nq >= 2) launch_scan<2>(lut, codes, n, nq, M, ksub, scores, ld, s);
  else launch_scan<1>(lut, codes, n, nq, M, ksub, scores, ld, s);
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}
