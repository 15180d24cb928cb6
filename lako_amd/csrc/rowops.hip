// Row-wise and element-wise kernels of the FiD-T5 reader path (all HBM-bound): T5LayerNorm
// (HF5:59-72), embedding gather/scatter (HF5:678), dropout (HF5:725,745,400,141), LM-head
// cross-entropy (HF5:1051-1054), global-norm clip + HF AdamW(correct_bias=False) (train_reader.py:76-79,
// src/util.py:185-227), _shift_right (HF5:618-637), greedy token selection (HF generate, greedy).
// 16-byte vector accesses per lane everywhere; wave-level (64-lane) shuffle reductions.
#include <stdarg.h>

#include <map>
#include <mutex>
#include <utility>

#include "common.h"
#include "det.h"

// ---- error string (thread local) ------------------------------------------------------------
static thread_local char g_err[512] = "";
void lako_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" int lako_last_error(char* buf, size_t n) {
  size_t len = strlen(g_err);
  if (buf && n) {
    size_t c = len < n - 1 ? len : n - 1;
    memcpy(buf, g_err, c);
    buf[c] = 0;
  }
  return (int)len;
}
extern "C" int lako_version(void) { return LAKO_ABI_VERSION; }

namespace {

LAKO_DET_STORAGE      // LAKO_DETERMINISTIC=1: see det.h

// keep decisions of 8 consecutive elements starting at idx0 (a multiple of 4): two quads
__device__ __forceinline__ void keep8(uint32_t key, uint64_t idx0, uint32_t thresh, bool (&k)[8]) {
  bool a[4], b[4];
  lako_keep4(key, idx0 >> 2, thresh, a);
  lako_keep4(key, (idx0 >> 2) + 1, thresh, b);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    k[i] = a[i];
    k[4 + i] = b[i];
  }
}

struct DropDev {
  uint32_t thresh, key;
  float scale;
};
inline DropDev make_drop(const lako_dropout_t& d) {
  DropDev r;
  r.thresh = d.p > 0.f ? lako_drop_thresh(d.p) : 0u;
  r.key = lako_drop_key(d.seed, d.site);
  r.scale = d.p > 0.f ? 1.0f / (1.0f - d.p) : 1.0f;
  return r;
}

// 8 consecutive elements as floats
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
  f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}
__device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
  bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
  *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
  bf16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (bf16_t)v[i];
  *reinterpret_cast<bf16x8*>(p) = o;
}

// ---- RMSNorm ----------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                          T* __restrict__ y, float* __restrict__ rstd, int64_t rows,
                                                          int d, float eps, DropDev dr) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t row = wid; row < rows; row += nw) {
    const T* xr = x + row * d;
    float ss = 0.f;
    for (int c = lane * 8; c < d; c += 512) {
      float v[8];
      load8(xr + c, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) ss += v[i] * v[i];
    }
    ss = wave_sum(ss);
    const float rs = rsqrtf(ss / (float)d + eps);
    if (lane == 0) rstd[row] = rs;
    T* yr = y + row * d;
    for (int c = lane * 8; c < d; c += 512) {
      float v[8], wv[8];
      load8(xr + c, v);
      load8(w + c, wv);
      bool kp[8];
      if (dr.thresh) keep8(dr.key, (uint64_t)row * d + c, dr.thresh, kp);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float o = wv[i] * (v[i] * rs);
        if (dr.thresh) o = kp[i] ? o * dr.scale : 0.f;
        v[i] = o;
      }
      store8(yr + c, v);
    }
  }
}

// The same with the MX quantiser of the normalised row fused in (round 6, bf16, d % 128 == 0, no dropout): y as above, plus its e4m3 bytes and
// E8M0 block scales in lako_mx_quantize's layout — what the fp8 forward product of the row consumes next, without a second pass over y
// (the quantiser reads the ROUNDED bf16 values: bit-identical to lako_rmsnorm_fwd + lako_mx_quantize).  Lane l holds the chunks l + 64 k
// of 8 elements: the four lanes of a 32-element block are neighbours.
__global__ __launch_bounds__(256) void rmsnorm_fwd_mx_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w, bf16_t* __restrict__ y,
                                                             float* __restrict__ rstd, uint8_t* __restrict__ q, uint8_t* __restrict__ sc, int ksp,
                                                             int64_t rows, int d, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t row = wid; row < rows; row += nw) {
    const bf16_t* xr = x + row * d;
    float ss = 0.f;
    for (int c = lane * 8; c < d; c += 512) {
      float v[8];
      load8(xr + c, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) ss += v[i] * v[i];
    }
    ss = wave_sum(ss);
    const float rs = rsqrtf(ss / (float)d + eps);
    if (lane == 0) rstd[row] = rs;
    bf16_t* yr = y + row * d;
    for (int c0 = 0; c0 < d; c0 += 512) {        // (every lane takes every iteration: the quantiser's shuffles need whole blocks of four lanes; d % 32 == 0)
      const int c = c0 + lane * 8;
      const bool in = c < d;
      float v[8], wv[8];
      if (in) {
        load8(xr + c, v);
        load8(w + c, wv);
      }
      bf16x8 o;
      float f[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        o[i] = (bf16_t)(in ? wv[i] * (v[i] * rs) : 0.f);
        f[i] = (float)o[i];
      }
      uint32_t wq[2];
      const uint8_t scale = mx_quant8(f, wq);
      if (in) {
        *reinterpret_cast<bf16x8*>(yr + c) = o;
        *reinterpret_cast<u32x2*>(q + row * d + c) = u32x2{wq[0], wq[1]};
        if ((lane & 3) == 0) {
          const int blk = c >> 5;
          sc[row * 4 * ksp + (blk & 3) * ksp + (blk >> 2)] = scale;
        }
      }
    }
  }
}

// 8 consecutive elements as they sit in memory (a prefetched row costs 4 registers per 8 bf16, converted when it is used)
template <typename T>
struct Raw8;
template <>
struct Raw8<bf16_t> {
  bf16x8 v;
  __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const bf16x8*>(p); }
  __device__ __forceinline__ void get(float (&o)[8]) const {
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
  }
};
template <>
struct Raw8<float> {
  f32x4 a, b;
  __device__ __forceinline__ void load(const float* p) {
    a = *reinterpret_cast<const f32x4*>(p);
    b = *reinterpret_cast<const f32x4*>(p + 4);
  }
  __device__ __forceinline__ void get(float (&o)[8]) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[i] = a[i]; o[4 + i] = b[i]; }
  }
};

// dx = dres + rstd*(w∘dy − x·rstd²·mean(w∘dy∘x));  dw += Σ_rows dy∘x·rstd.   d <= 1024.
// A wave walks its rows one at a time; all three operands of the NEXT row (dy, x, dres) are requested before the current row's
// reduction, so a row costs one memory round trip that overlaps the previous row's arithmetic instead of two exposed ones (the first
// version fetched dres only after the wave reduction: 3.6-4.6 TB/s).  dx may alias dres: a row is read and written by one wave only.
template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const T* dy, const T* x, const float* __restrict__ w,
                                                          const float* __restrict__ rstd, const T* dres, T* dx,
                                                          float* __restrict__ dw, int64_t rows, int d, DropDev dr,
                                                          T* __restrict__ dx_drop, DropDev dr_out, int det) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  float dwacc[2][8];
  float wv[2][8];
  const int c0 = lane * 8, c1 = 512 + lane * 8;
  const bool has1 = c1 < d, has0 = c0 < d;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    int c = it * 512 + lane * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) { dwacc[it][i] = 0.f; wv[it][i] = 0.f; }
    if (c < d) load8(w + c, wv[it]);
  }
  Raw8<T> ng[2], nx[2], nr[2];
  float nrs = 0.f;
  auto fetch = [&](int64_t row) {
    if (row < rows) {
      nrs = rstd[row];
      const int64_t base = row * d;
      if (has0) {
        ng[0].load(dy + base + c0);
        nx[0].load(x + base + c0);
        if (dres) nr[0].load(dres + base + c0);
      }
      if (has1) {
        ng[1].load(dy + base + c1);
        nx[1].load(x + base + c1);
        if (dres) nr[1].load(dres + base + c1);
      }
    }
  };
  fetch(wid);
  for (int64_t row = wid; row < rows; row += nw) {
    const float rs = nrs;
    float g[2][8], xv[2][8], o[2][8];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) g[it][i] = xv[it][i] = o[it][i] = 0.f;
      if (it == 0 ? has0 : has1) {
        ng[it].get(g[it]);
        nx[it].get(xv[it]);
        if (dres) nr[it].get(o[it]);
      }
    }
    fetch(row + nw);
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      int c = it * 512 + lane * 8;
      if (c < d) {
        bool kp[8];
        if (dr.thresh) keep8(dr.key, (uint64_t)row * d + c, dr.thresh, kp);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (dr.thresh) g[it][i] = kp[i] ? g[it][i] * dr.scale : 0.f;
          s += wv[it][i] * g[it][i] * xv[it][i];
          dwacc[it][i] += g[it][i] * xv[it][i] * rs;
        }
      }
    }
    s = wave_sum(s);
    const float k = rs * rs * rs * s / (float)d;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      int c = it * 512 + lane * 8;
      if (c < d) {
#pragma unroll
        for (int i = 0; i < 8; ++i) o[it][i] += rs * wv[it][i] * g[it][i] - xv[it][i] * k;
        store8(dx + row * d + c, o[it]);
        if (dx_drop) {   // the consumer's dropout_bwd(dx) in the same pass: exactly lako_dropout_apply of the ROUNDED dx
          bool kp[8];
          keep8(dr_out.key, (uint64_t)row * d + c, dr_out.thresh, kp);
          float od[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) od[i] = kp[i] ? (float)(T)o[it][i] * dr_out.scale : 0.f;
          store8(dx_drop + row * d + c, od);
        }
      }
    }
  }
  // block-level reduction of the 4 waves' partial dw in LDS, then ONE atomic per column per block (all
  // blocks add into the same d addresses, so the number of adders per address is what costs)
  __shared__ float red[4][1024];
  const int wave_id = threadIdx.x >> 6;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    int c = it * 512 + lane * 8;
    if (c < d) {
#pragma unroll
      for (int i = 0; i < 8; ++i) red[wave_id][c + i] = dwacc[it][i];
    }
  }
  __syncthreads();
  if (det) {      // LAKO_DETERMINISTIC: fixed-point adds, converted once by the workgroup that arrives last (det.h)
    __shared__ int last;
    for (int c = threadIdx.x; c < d; c += 256) lako_det::add(&lako_det_slots[c], red[0][c] + red[1][c] + red[2][c] + red[3][c]);
    if (lako_det::last_block(&lako_det_ticket, gridDim.x, &last))
      for (int c = threadIdx.x; c < d; c += 256) dw[c] += lako_det::take(&lako_det_slots[c]);
    return;
  }
  for (int c = threadIdx.x; c < d; c += 256) atomicAdd(dw + c, red[0][c] + red[1][c] + red[2][c] + red[3][c]);
}

#ifdef LAKO_EXPERIMENTS   // round 2's kernel (dres fetched after the reduction, no prefetch) for A/B timing: LAKO_RMS_OLD=1
template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_bwd_old_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                          const float* __restrict__ w,
                                                          const float* __restrict__ rstd, const T* __restrict__ dres,
                                                          T* __restrict__ dx, float* __restrict__ dw, int64_t rows,
                                                          int d, DropDev dr, T* __restrict__ dx_drop, DropDev dr_out) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  float dwacc[2][8];
  float wv[2][8];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    int c = it * 512 + lane * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) { dwacc[it][i] = 0.f; wv[it][i] = 0.f; }
    if (c < d) load8(w + c, wv[it]);
  }
  for (int64_t row = wid; row < rows; row += nw) {
    const float rs = rstd[row];
    float g[2][8], xv[2][8];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      int c = it * 512 + lane * 8;
      if (c < d) {
        load8(dy + row * d + c, g[it]);
        load8(x + row * d + c, xv[it]);
        bool kp[8];
        if (dr.thresh) keep8(dr.key, (uint64_t)row * d + c, dr.thresh, kp);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (dr.thresh) g[it][i] = kp[i] ? g[it][i] * dr.scale : 0.f;
          s += wv[it][i] * g[it][i] * xv[it][i];
          dwacc[it][i] += g[it][i] * xv[it][i] * rs;
        }
      }
    }
    s = wave_sum(s);
    const float k = rs * rs * rs * s / (float)d;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      int c = it * 512 + lane * 8;
      if (c < d) {
        float o[8];
        if (dres) load8(dres + row * d + c, o);
        else {
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] += rs * wv[it][i] * g[it][i] - xv[it][i] * k;
        store8(dx + row * d + c, o);
        if (dx_drop) {   // the consumer's dropout_bwd(dx) in the same pass: exactly lako_dropout_apply of the ROUNDED dx
          bool kp[8];
          keep8(dr_out.key, (uint64_t)row * d + c, dr_out.thresh, kp);
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] = kp[i] ? (float)(T)o[i] * dr_out.scale : 0.f;
          store8(dx_drop + row * d + c, o);
        }
      }
    }
  }
  // block-level reduction of the 4 waves' partial dw in LDS, then ONE atomic per column per block (all
  // blocks add into the same d addresses, so the number of adders per address is what costs)
  __shared__ float red[4][1024];
  const int wave_id = threadIdx.x >> 6;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    int c = it * 512 + lane * 8;
    if (c < d) {
#pragma unroll
      for (int i = 0; i < 8; ++i) red[wave_id][c + i] = dwacc[it][i];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < d; c += 256) atomicAdd(dw + c, red[0][c] + red[1][c] + red[2][c] + red[3][c]);
}
#endif

// ---- embedding ----------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* __restrict__ ids, const T* __restrict__ table,
                                                        T* __restrict__ out, int64_t n_tok, int d, int64_t vocab,
                                                        DropDev dr) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t t = wid; t < n_tok; t += nw) {
    int64_t id = ids[t];
    if (id < 0 || id >= vocab) id = 0;
    for (int c = lane * 8; c < d; c += 512) {
      float v[8];
      load8(table + id * d + c, v);
      if (dr.thresh) {
        bool kp[8];
        keep8(dr.key, (uint64_t)t * d + c, dr.thresh, kp);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = kp[i] ? v[i] * dr.scale : 0.f;
      }
      store8(out + t * d + c, v);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int64_t* __restrict__ ids, const T* __restrict__ dout,
                                                        float* __restrict__ dtable, int64_t n_tok, int d,
                                                        int64_t vocab, DropDev dr) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t t = wid; t < n_tok; t += nw) {
    int64_t id = ids[t];
    if (id < 0 || id >= vocab) id = 0;
    // one dword per lane per instruction: 256 contiguous bytes per wave atomic (full-rate shape).  Wave-instructions
    // whose 64 values are all zero are skipped: adding 0 changes nothing, and the gradient at padded positions is
    // exactly 0 — a quarter of the config-2 tokens, every one of them aimed at the SAME table row (pad id 0).
    for (int c = lane; c < d; c += 64) {
      float v = (float)dout[t * d + c];
      if (dr.thresh) v = lako_keep(dr.key, (uint64_t)t * d + c, dr.thresh) ? v * dr.scale : 0.f;
      if (__ballot(v != 0.f) == 0) continue;
      atomicAdd(dtable + id * d + c, v);
    }
  }
}

// LAKO_DETERMINISTIC: the same sums with ONE writer per table row and a fixed order.  `perm` is a stable ascending sort of the token
// positions by id (index plumbing of the caller: torch.argsort); the wave of sorted position j works only if j opens a run of equal ids,
// and then adds that run's rows in run order — position order — into registers and the row once.
template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_ordered_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ perm,
                                                                const T* __restrict__ dout, float* __restrict__ dtable, int64_t n_tok,
                                                                int d, int64_t vocab, DropDev dr) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  auto id_of = [&](int64_t j) {
    int64_t id = ids[perm[j]];
    return (id < 0 || id >= vocab) ? (int64_t)0 : id;
  };
  for (int64_t j = wid; j < n_tok; j += nw) {
    const int64_t id = id_of(j);
    if (j > 0 && id_of(j - 1) == id) continue;          // not the first of its run
    for (int c0 = 0; c0 < d; c0 += 64 * 4) {             // 4 columns per lane and sweep: the run is walked once per 256 columns
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      for (int64_t jj = j; jj < n_tok && id_of(jj) == id; ++jj) {
        const int64_t t = perm[jj];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int c = c0 + u * 64 + lane;
          if (c < d) {
            float v = (float)dout[t * d + c];
            if (dr.thresh) v = lako_keep(dr.key, (uint64_t)t * d + c, dr.thresh) ? v * dr.scale : 0.f;
            acc[u] += v;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + u * 64 + lane;
        if (c < d) dtable[id * d + c] += acc[u];
      }
    }
  }
}

// ---- dropout apply / cast ---------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void dropout_apply_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n,
                                                            DropDev dr) {
  int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
  const int64_t stride = (int64_t)gridDim.x * 256 * 8;
  for (; i < n; i += stride) {
    float v[8];
    load8(x + i, v);
    bool kp[8];
    keep8(dr.key, (uint64_t)i, dr.thresh, kp);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = kp[j] ? v[j] * dr.scale : 0.f;
    store8(y + i, v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, T* __restrict__ dst, int64_t n) {
  int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
  const int64_t stride = (int64_t)gridDim.x * 256 * 8;
  for (; i + 8 <= n; i += stride) {
    float v[8];
    load8(src + i, v);
    store8(dst + i, v);
  }
  if (i < n && i + 8 > n)
    for (int64_t j = i; j < n; ++j) dst[j] = (T)src[j];
}

template <typename T>
__global__ __launch_bounds__(256) void transpose_cast_kernel(const float* __restrict__ src, T* __restrict__ dst,
                                                             int64_t rows, int64_t cols) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 × 8
  const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int64_t r = r0 + ty + 8 * j, c = c0 + tx;
    tile[ty + 8 * j][tx] = (r < rows && c < cols) ? src[r * cols + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int64_t c = c0 + ty + 8 * j, r = r0 + tx;
    if (r < rows && c < cols) dst[c * rows + r] = (T)tile[tx][ty + 8 * j];
  }
}

// All weight matrices of the model in ONE launch (122 matrices at T5-base: one launch instead of 122 of ≈8 µs each).
// desc[i] = {src offset, dst offset (elements), rows, cols}; tile_prefix[i] = first 64×64 tile of matrix i.
template <typename T, typename TS>
__global__ __launch_bounds__(256) void transpose_cast_batched_kernel(const TS* __restrict__ src_base, T* __restrict__ dst_base,
                                                                     const int64_t* __restrict__ desc,
                                                                     const int32_t* __restrict__ tile_prefix, int n) {
  __shared__ float tile[64][65];
  int lo = 0, hi = n - 1;                      // last i with tile_prefix[i] <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tile_prefix[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const int64_t* dd = desc + 4 * lo;
  const TS* src = src_base + dd[0];
  T* dst = dst_base + dd[1];
  const int64_t rows = dd[2], cols = dd[3];
  const int t = blockIdx.x - tile_prefix[lo];
  const int tiles_c = (int)((cols + 63) >> 6);
  const int64_t r0 = (int64_t)(t / tiles_c) * 64, c0 = (int64_t)(t % tiles_c) * 64;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;   // 16 × 16; a thread reads 4 consecutive elements
  const bool vec = (cols % 4 == 0) && (rows % 4 == 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t r = r0 + ty + 16 * j, c = c0 + 4 * tx;
    if (vec && r < rows && c + 3 < cols) {
      const f32x4 v = load4(src + r * cols + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) tile[ty + 16 * j][4 * tx + e] = v[e];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) tile[ty + 16 * j][4 * tx + e] = (r < rows && c + e < cols) ? to_f32(src[r * cols + c + e]) : 0.f;
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t c = c0 + ty + 16 * j, r = r0 + 4 * tx;    // dst row c, 4 consecutive r
    if (c >= cols) continue;
    if (vec && r + 3 < rows) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = tile[4 * tx + e][ty + 16 * j];
      store4(dst + c * rows + r, v);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (r + e < rows) dst[c * rows + r + e] = (T)tile[4 * tx + e][ty + 16 * j];
    }
  }
}

// ---- RMSNorm at d = 768 (T5-base): TWO rows per wave ------------------------------------------------------------------
// One wave per row leaves a third of the lanes idle on the row's second 512-element slice (768 = 512 + 256), reads the row twice
// in the forward and runs one dependent load → reduce → load → store chain per row.  Here a wave takes rows 2p and 2p + 1 as three
// full slices of 512 elements (row A: columns 0…511 | row A: 512…767 on lanes 0…31, row B: 0…255 on lanes 32…63 | row B: 256…767),
// keeps them in registers, and requests the next pair's slices before it reduces the current pair.  Same arithmetic per element
// as the generic kernel; the sums run over the same elements in another order.  Forward at 48 k rows: 43 → 28 µs (5.3 TB/s).
// (The backward in the same form measured 82 µs against 83: with three input and two output streams it sits at the 4.5 TB/s the
// mixed read / write traffic reaches either way — and 128 µs with the next pair double-buffered in 256 registers; not kept.)
struct Pair768 {
  int c0, c1, c2;      // the lane's first column of each slice
  bool lo;             // lanes 0…31: slice 1 belongs to row A
};
__device__ __forceinline__ Pair768 pair768(int lane) {
  Pair768 p;
  p.lo = lane < 32;
  p.c0 = lane * 8;
  p.c1 = p.lo ? 512 + lane * 8 : (lane - 32) * 8;
  p.c2 = 256 + lane * 8;
  return p;
}

template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_fwd768_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                             T* __restrict__ y, float* __restrict__ rstd, int64_t rows,
                                                             float eps, DropDev dr) {
  constexpr int D = 768;
  const int lane = threadIdx.x & 63;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
  const int64_t npairs = (rows + 1) >> 1;
  const Pair768 pc = pair768(lane);
  float w0[8], w1[8], w2[8];
  load8(w + pc.c0, w0);
  load8(w + pc.c1, w1);
  load8(w + pc.c2, w2);
  auto fetch = [&](int64_t pair, float (&v0)[8], float (&v1)[8], float (&v2)[8]) {
    const int64_t rA = 2 * pair, rB = rA + 1;
    const bool hasB = rB < rows;
#pragma unroll
    for (int i = 0; i < 8; ++i) v0[i] = v1[i] = v2[i] = 0.f;
    if (pair < npairs) {
      load8(x + rA * D + pc.c0, v0);
      if (pc.lo) load8(x + rA * D + pc.c1, v1);
      else if (hasB) load8(x + rB * D + pc.c1, v1);
      if (hasB) load8(x + rB * D + pc.c2, v2);
    }
  };
  float n0[8], n1[8], n2[8];
  fetch(wid, n0, n1, n2);
  for (int64_t pair = wid; pair < npairs; pair += nw) {
    float v0[8], v1[8], v2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { v0[i] = n0[i]; v1[i] = n1[i]; v2[i] = n2[i]; }
    fetch(pair + nw, n0, n1, n2);
    const int64_t rA = 2 * pair, rB = rA + 1;
    const bool hasB = rB < rows;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { s0 += v0[i] * v0[i]; s1 += v1[i] * v1[i]; s2 += v2[i] * v2[i]; }
    const float ssA = wave_sum(s0 + (pc.lo ? s1 : 0.f)), ssB = wave_sum(s2 + (pc.lo ? 0.f : s1));
    const float rsA = rsqrtf(ssA / (float)D + eps), rsB = rsqrtf(ssB / (float)D + eps);
    if (lane == 0) {
      rstd[rA] = rsA;
      if (hasB) rstd[rB] = rsB;
    }
    const float rs1 = pc.lo ? rsA : rsB;
    const int64_t r1 = pc.lo ? rA : rB;
    bool k0[8], k1[8], k2[8];
    if (dr.thresh) {
      keep8(dr.key, (uint64_t)rA * D + pc.c0, dr.thresh, k0);
      keep8(dr.key, (uint64_t)r1 * D + pc.c1, dr.thresh, k1);
      keep8(dr.key, (uint64_t)rB * D + pc.c2, dr.thresh, k2);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float o0 = w0[i] * (v0[i] * rsA), o1 = w1[i] * (v1[i] * rs1), o2 = w2[i] * (v2[i] * rsB);
      if (dr.thresh) {
        o0 = k0[i] ? o0 * dr.scale : 0.f;
        o1 = k1[i] ? o1 * dr.scale : 0.f;
        o2 = k2[i] ? o2 * dr.scale : 0.f;
      }
      v0[i] = o0; v1[i] = o1; v2[i] = o2;
    }
    store8(y + rA * D + pc.c0, v0);
    if (pc.lo || hasB) store8(y + r1 * D + pc.c1, v1);
    if (hasB) store8(y + rB * D + pc.c2, v2);
  }
}

// ---- cross-entropy --------------------------------------------------------------------------
// The mean is accumulated in 2^-40 fixed point (a 64-bit integer add is order-independent, so the loss does not depend on which
// row's workgroup finishes first); the last row to finish converts it.  scratch = {u64 sum, u32 rows done, u32 non-finite flags}.
// A row whose loss is NaN / ±Inf (diverged run) or too large for the fixed-point sum (|loss| > 2^22 / rows) is not added: it sets a
// flag bit instead (1 NaN, 2 +Inf / overflow, 4 −Inf) and the reported mean is then NaN / ±Inf like a float sum's would be.
struct CeScratch {
  unsigned long long sum;
  unsigned done;
  unsigned bad;
};
constexpr float CE_FIX = 1099511627776.0f;   // 2^40

__global__ void ce_init_kernel(float* loss_out, const int64_t* labels, int64_t M, CeScratch* sc) {
  // single block: count valid labels
  __shared__ int cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  int c = 0;
  for (int64_t i = threadIdx.x; i < M; i += blockDim.x) c += labels[i] != -100;
  atomicAdd(&cnt, c);
  __syncthreads();
  if (threadIdx.x == 0) {
    loss_out[0] = 0.f;
    loss_out[1] = (float)cnt;
    sc->sum = 0ull;
    sc->done = 0u;
    sc->bad = 0u;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                 float* __restrict__ loss_out, T* __restrict__ dlogits,
                                                 const float* __restrict__ upstream, int64_t V, CeScratch* sc) {
  __shared__ float red[4];
  __shared__ float bc;
  const int64_t row = blockIdx.x;
  const float* lr = logits + row * V;
  const int64_t label = labels[row];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float n_valid = loss_out[1];
  const bool valid = label != -100;
  float mx = -INFINITY;
  for (int64_t c = threadIdx.x * 4; c < V; c += 1024) {
    f32x4 v = *reinterpret_cast<const f32x4*>(lr + c);
    mx = fmaxf(fmaxf(mx, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
  }
  mx = wave_max(mx);
  if (lane == 0) red[wv] = mx;
  __syncthreads();
  if (threadIdx.x == 0) bc = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  mx = bc;
  float se = 0.f;
  for (int64_t c = threadIdx.x * 4; c < V; c += 1024) {
    f32x4 v = *reinterpret_cast<const f32x4*>(lr + c);
    se += __expf(v[0] - mx) + __expf(v[1] - mx) + __expf(v[2] - mx) + __expf(v[3] - mx);
  }
  se = wave_sum(se);
  __syncthreads();
  if (lane == 0) red[wv] = se;
  __syncthreads();
  if (threadIdx.x == 0) bc = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  se = bc;
  const float lse = mx + __logf(se);
  if (threadIdx.x == 0) {
    if (valid) {
      const float lrow = lse - lr[label];
      const float lim = 4194304.0f / (float)gridDim.x;     // Σ rows·|loss|·2^40 stays below 2^62
      if (lrow != lrow) atomicOr(&sc->bad, 1u);
      else if (lrow > lim) atomicOr(&sc->bad, 2u);
      else if (lrow < -lim) atomicOr(&sc->bad, 4u);
      else atomicAdd(&sc->sum, (unsigned long long)(long long)((double)lrow * (double)CE_FIX));
    }
    __threadfence();
    if (atomicAdd(&sc->done, 1u) == gridDim.x - 1) {
      const long long tot = (long long)atomicAdd(&sc->sum, 0ull);
      const unsigned bad = atomicOr(&sc->bad, 0u);
      float mean = (float)((double)tot / (double)CE_FIX / (double)n_valid);
      if ((bad & 1u) || (bad & 6u) == 6u) mean = __builtin_nanf("");
      else if (bad & 2u) mean = INFINITY;
      else if (bad & 4u) mean = -INFINITY;
      loss_out[0] = mean;
    }
  }
  if (dlogits) {
    T* dr = dlogits + row * V;
    const float inv = valid ? (upstream ? upstream[0] : 1.0f) / n_valid : 0.f;
    for (int64_t c = threadIdx.x * 4; c < V; c += 1024) {
      f32x4 v = *reinterpret_cast<const f32x4*>(lr + c);
      f32x4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float p = __expf(v[i] - lse);
        o[i] = (p - ((c + i) == label ? 1.f : 0.f)) * inv;
      }
      store4(dr + c, o);
    }
  }
}

// ---- optimizer ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ out, int det) {
  __shared__ float red[4];
  float s = 0.f;
  int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const int64_t stride = (int64_t)gridDim.x * 256 * 4;
  for (; i + 4 <= n; i += stride) {
    f32x4 v = *reinterpret_cast<const f32x4*>(g + i);
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (i < n)
    for (int64_t j = i; j < n && j < i + 4; ++j) s += g[j] * g[j];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (det) {      // LAKO_DETERMINISTIC (det.h): the squared norm in fixed point — block sums of squares stay far below 2^23
    __shared__ int last;
    if (threadIdx.x == 0) lako_det::add(&lako_det_slots[lako_det::SLOTS - 1], red[0] + red[1] + red[2] + red[3]);
    if (lako_det::last_block(&lako_det_ticket, gridDim.x, &last) && threadIdx.x == 0) out[0] += lako_det::take(&lako_det_slots[lako_det::SLOTS - 1]);
    return;
  }
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

template <typename T>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    T* __restrict__ shadow, int64_t n, float lr, float b1, float b2,
                                                    float eps, float wd, const float* __restrict__ gnorm_sq,
                                                    float max_norm, float grad_scale) {
  float coef = grad_scale;
  if (gnorm_sq) {
    float total = sqrtf(gnorm_sq[0]) * grad_scale;
    float c = max_norm / (total + 1e-6f);
    coef *= fminf(c, 1.0f);
  }
  // two 16-byte vectors per thread and iteration, all eight loads in flight before the first use; m / v / g are streamed once per
  // step: non-temporal accesses keep them from displacing the weights the next forward reads (measured in bench.py's breakdown)
  const int64_t stride = (int64_t)gridDim.x * 256 * 4;
  for (int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i0 < n; i0 += 2 * stride) {  // n % 4 == 0 (checked on host)
    f32x4 pv[2], gv[2], mv[2], vv[2];
    bool ok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t i = i0 + u * stride;
      ok[u] = i < n;
      if (ok[u]) {
        pv[u] = *reinterpret_cast<f32x4*>(p + i);
        gv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g + i));
        mv[u] = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(m + i));
        vv[u] = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(v + i));
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (!ok[u]) continue;
      const int64_t i = i0 + u * stride;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float gj = gv[u][j] * coef;
        mv[u][j] = mv[u][j] * b1 + gj * (1.f - b1);
        vv[u][j] = vv[u][j] * b2 + gj * gj * (1.f - b2);
        float q = pv[u][j] - lr * (mv[u][j] / (sqrtf(vv[u][j]) + eps));
        if (wd > 0.f) q = q - lr * wd * q;
        pv[u][j] = q;
      }
      *reinterpret_cast<f32x4*>(p + i) = pv[u];
      __builtin_nontemporal_store(mv[u], reinterpret_cast<f32x4*>(m + i));
      __builtin_nontemporal_store(vv[u], reinterpret_cast<f32x4*>(v + i));
      if (shadow) store4(shadow + i, pv[u]);
    }
  }
}

// ---- relative position bias table -------------------------------------------------------------
__global__ void relpos_expand_kernel(const float* table, const int32_t* lut, float* rel, int H, int R, int nb) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= H * R) return;
  int h = i / R, r = i % R;
  int b = lut[r];
  rel[i] = (b >= 0 && b < nb) ? table[b * H + h] : 0.f;
}
// (round 6) a GATHER: thread (bucket b, head h) sums the offsets r whose bucket is b in ascending r and adds the sum to its table entry —
// one writer per entry, a fixed order, no atomics (rounds 1-5 scattered every offset into its bucket by a float atomic: ≤ 2L − 1 adders
// per entry in arrival order).  nb · H threads of ≤ 2L − 1 iterations: nothing next to the launch itself.
__global__ void relpos_reduce_kernel(const float* drel, const int32_t* lut, float* dtable, int H, int R, int nb) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nb * H) return;
  int b = i / H, h = i % H;
  float s = 0.f;
  for (int r = 0; r < R; ++r)
    if (lut[r] == b) s += drel[(int64_t)h * R + r];
  dtable[b * H + h] += s;
}

// ---- int helpers ------------------------------------------------------------------------------
__global__ void shift_right_kernel(const int64_t* labels, int64_t* dec, int B, int T) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * T) return;
  int t = i % T;
  int64_t v = t == 0 ? 0 : labels[i - 1];
  dec[i] = v == -100 ? 0 : v;
}

// token ids of the VALID positions of a padded [BN, L] batch, packed in (passage, position) order: out[off[j] + t] = ids[j·L + t], t < len(j).
// One launch instead of the arange / cumsum / repeat_interleave / gather chain that built and applied a packed-row index on the device.
__global__ __launch_bounds__(256) void pack_ids_kernel(const int64_t* __restrict__ ids, const int32_t* __restrict__ off, int64_t* __restrict__ out,
                                                       int BN, int L) {
  const int j = blockIdx.x;
  const int o0 = off[j], n = off[j + 1] - o0;
  for (int t = threadIdx.x; t < n && t < L; t += 256) out[o0 + t] = ids[(int64_t)j * L + t];
}

__global__ __launch_bounds__(256) void greedy_step_kernel(const float* __restrict__ logits, int64_t V,
                                                          int64_t* seq, int64_t seq_ld, int pos, int64_t* next_ids,
                                                          uint8_t* done, int64_t eos, int64_t pad) {
  __shared__ float bv[4];
  __shared__ int64_t bi[4];
  const int b = blockIdx.x;
  const float* lr = logits + (int64_t)b * V;
  float best = -INFINITY;
  int64_t besti = V;
  for (int64_t c = threadIdx.x; c < V; c += 256) {
    float v = lr[c];
    if (v > best) { best = v; besti = c; }  // strided scan keeps the lowest index per thread on ties
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ov = __shfl_xor(best, o, 64);
    int64_t oi = __shfl_xor(besti, o, 64);
    if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
  }
  if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = besti; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (bv[w] > best || (bv[w] == best && bi[w] < besti)) { best = bv[w]; besti = bi[w]; }
    if (besti >= V) besti = 0;   // a row of NaNs compares false everywhere: emit id 0 (what torch.argmax gives is unspecified), never an out-of-range id
    int64_t nxt = done[b] ? pad : besti;
    seq[(int64_t)b * seq_ld + pos] = nxt;
    next_ids[b] = nxt;
    if (nxt == eos) done[b] = 1;
  }
}
__global__ void count_done_kernel(const uint8_t* done, int B, int32_t* n_done) {
  __shared__ int cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  int c = 0;
  for (int i = threadIdx.x; i < B; i += blockDim.x) c += done[i] != 0;
  atomicAdd(&cnt, c);
  __syncthreads();
  if (threadIdx.x == 0) n_done[0] = cnt;
}

// ---- per-fact aggregation of captured cross-attention scores (src/model.py:100-115,143-204) ----------------
// One workgroup per sample.  vals[l] = Σ_{head, layer} scores[b, h, layer, p0·L + l] over the layers [ly0, ly0 + nly), masked
// positions 0 (the reference's masked_fill + sum(dim=[1,2]), rounded to fp32 like its tensor).  The token ids that define the spans
// are those of passage `pid` — the reference reads context_ids[b][1] (src/model.py:173) while the scores come from the first
// passage of the second chunk, p0 = ceil(N / 2) (:164,174): the same passage only for N = 2 (stream 2).  Passage p0 is then cut
// into spans ending at token id 5 ('.'), starting at index 2 (after '▁fact', ':'); an unterminated last span counts when the
// passage is not padded; each of the first n_ctx spans is reduced by mean | max | mean of the larger half ("21mean": the
// reference's heapq.nlargest(max(int((len+1)/2), 1))), in double and in the reference's summation order; missing facts get −5;
// everything is divided by (layers · heads).
__global__ __launch_bounds__(256) void fact_scores_kernel(const float* __restrict__ scores, const uint8_t* __restrict__ mask,
                                                          const int64_t* __restrict__ ids, double* __restrict__ out, int H, int nl,
                                                          int ly0, int nly, int N, int L, int p0, int pid, int n_ctx, int style) {
  extern __shared__ __attribute__((aligned(16))) char fs_smem[];
  float* vals = reinterpret_cast<float*>(fs_smem);        // [L]
  int* tok = reinterpret_cast<int*>(vals + L);             // [L]  (only == 5 / != 0 matter)
  int* sp = tok + L;                                       // [2·n_ctx] (start, end) + [1] count
  const int b = blockIdx.x;
  const int64_t S = (int64_t)N * L;
  for (int l = threadIdx.x; l < L; l += 256) {
    double acc = 0.0;
    if (mask[((int64_t)b * N + p0) * L + l]) {
      for (int h = 0; h < H; ++h)
        for (int ly = ly0; ly < ly0 + nly; ++ly) acc += (double)scores[(((int64_t)b * H + h) * nl + ly) * S + (int64_t)p0 * L + l];
    }
    vals[l] = (float)acc;
    const int64_t t = ids[((int64_t)b * N + pid) * L + l];
    tok[l] = t == 5 ? 5 : (t == 0 ? 0 : 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int n = 0, start = 2;
    while (n < n_ctx) {
      int e = start;
      while (e < L && tok[e] != 5) ++e;
      if (e >= L) break;
      sp[2 * n] = start;
      sp[2 * n + 1] = e + 1;
      ++n;
      start = e + 1;
    }
    if (n < n_ctx && tok[L - 1] != 0 && L > start) {
      sp[2 * n] = start;
      sp[2 * n + 1] = L;
      ++n;
    }
    sp[2 * n_ctx] = n;
  }
  __syncthreads();
  const int nspan = sp[2 * n_ctx];
  const double denom = (double)nly * (double)H;
  for (int j = threadIdx.x; j < n_ctx; j += 256) {
    double r = -5.0;
    if (j < nspan) {
      const int s0 = sp[2 * j], e0 = sp[2 * j + 1], len = e0 - s0;
      if (style == 0) {           // mean
        double a = 0.0;
        for (int i = s0; i < e0; ++i) a += (double)vals[i];
        r = a / (double)len;
      } else if (style == 1) {    // max
        double m = (double)vals[s0];
        for (int i = s0 + 1; i < e0; ++i) m = fmax(m, (double)vals[i]);
        r = m;
      } else {                    // 21mean: the num largest, summed from the largest down
        const int num = max((len + 1) / 2, 1);
        double a = 0.0;
        for (int rk = 0; rk < num; ++rk) {
          for (int i = s0; i < e0; ++i) {
            int rank = 0;
            for (int k = s0; k < e0; ++k) rank += (vals[k] > vals[i]) || (vals[k] == vals[i] && k < i);
            if (rank == rk) { a += (double)vals[i]; break; }
          }
        }
        r = a / (double)num;
      }
    }
    out[(int64_t)b * n_ctx + j] = r / denom;
  }
}

inline int rows_grid(int64_t rows) {
  int64_t g = (rows + 3) / 4;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}
inline int elems_grid(int64_t n, int per_thread) {
  int64_t g = (n + 256 * per_thread - 1) / (256 * per_thread);
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

#define DISPATCH_T(dtype, ...)                 \
  if ((dtype) == LAKO_BF16) {                  \
    using T = bf16_t;                          \
    __VA_ARGS__;                               \
  } else {                                     \
    using T = float;                           \
    __VA_ARGS__;                               \
  }
#define CHECK_DTYPE(fn, dtype) LAKO_CHECK_ARG((dtype) == LAKO_F32 || (dtype) == LAKO_BF16, fn ": bad dtype %d", (int)(dtype))

extern "C" int lako_rmsnorm_fwd(const void* x, const float* w, void* y, float* rstd, int64_t rows, int d, float eps,
                                int dtype, lako_dropout_t drop, lako_stream_t stream) {
  CHECK_DTYPE("lako_rmsnorm_fwd", dtype);
  LAKO_CHECK_ARG(rows > 0 && d > 0 && d % 8 == 0, "lako_rmsnorm_fwd: rows=%lld d=%d (d %% 8 == 0)", (long long)rows, d);
  LAKO_CHECK_ALIGN(x, 16);
  LAKO_CHECK_ALIGN(y, 16);
  LAKO_CHECK_ALIGN(w, 16);
  DropDev dr = make_drop(drop);
  if (d == 768) {      // two rows per wave
    DISPATCH_T(dtype, hipLaunchKernelGGL((rmsnorm_fwd768_kernel<T>), dim3(rows_grid((rows + 1) / 2)), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)x, w, (T*)y, rstd, rows, eps, dr));
    LAKO_LAUNCH_CHECK();
    return LAKO_OK;
  }
  DISPATCH_T(dtype, hipLaunchKernelGGL((rmsnorm_fwd_kernel<T>), dim3(rows_grid(rows)), dim3(256), 0,
                                       (hipStream_t)stream, (const T*)x, w, (T*)y, rstd, rows, d, eps, dr));
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

extern "C" int lako_rmsnorm_fwd_mx(const void* x, const float* w, void* y, float* rstd, void* q, uint8_t* scales, int64_t rows, int d, float eps,
                                   lako_stream_t stream) {
  LAKO_CHECK_ARG(rows > 0 && d > 0 && d % 128 == 0, "lako_rmsnorm_fwd_mx: rows=%lld d=%d (d %% 128 == 0)", (long long)rows, d);
  LAKO_CHECK_ARG(x && w && y && rstd && q && scales, "lako_rmsnorm_fwd_mx: null tensor");
  LAKO_CHECK_ALIGN(x, 16);
  LAKO_CHECK_ALIGN(y, 16);
  LAKO_CHECK_ALIGN(w, 16);
  LAKO_CHECK_ALIGN(q, 8);
  const int ksp = (d / 128 + 3) / 4 * 4;      // scale columns per 32-k lane group, as lako_mx_scale_cols(d) / 4
  if (d == 768) {
    // lako_rmsnorm_fwd has its own kernel for this width (two rows per wave: another summation order of the squares): "bit for bit" means
    // that kernel followed by the quantiser — two launches, only the call is fused (T5-base is no fp8 configuration of the benchmark)
    lako_dropout_t none{};
    int rc = lako_rmsnorm_fwd(x, w, y, rstd, rows, d, eps, LAKO_BF16, none, stream);
    if (rc != LAKO_OK) return rc;
    return lako_mx_quantize(y, rows, d, d, q, scales, stream);
  }
  hipLaunchKernelGGL(rmsnorm_fwd_mx_kernel, dim3(rows_grid(rows)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, w, (bf16_t*)y, rstd, (uint8_t*)q, scales,
                     ksp, rows, d, eps);
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

extern "C" int lako_rmsnorm_bwd(const void* dy, const void* x, const float* w, const float* rstd, const void* dres,
                                void* dx, float* dw, int64_t rows, int d, int dtype, lako_dropout_t drop, void* dx_drop,
                                lako_dropout_t drop_out, lako_stream_t stream) {
  CHECK_DTYPE("lako_rmsnorm_bwd", dtype);
  LAKO_CHECK_ARG(rows > 0 && d > 0 && d % 8 == 0 && d <= 1024, "lako_rmsnorm_bwd: rows=%lld d=%d (d %% 8 == 0, d <= 1024)",
                 (long long)rows, d);
  LAKO_CHECK_ALIGN(dy, 16);
  LAKO_CHECK_ALIGN(x, 16);
  LAKO_CHECK_ALIGN(dx, 16);
  LAKO_CHECK_ALIGN(w, 16);
  LAKO_CHECK_ARG(!dx_drop || drop_out.p > 0.f, "lako_rmsnorm_bwd: dx_drop without a dropout");
  LAKO_CHECK_ALIGN(dx_drop, 16);
  DropDev dr = make_drop(drop), dr_out = make_drop(drop_out);
  int grid = rows_grid(rows);
  // 3 blocks per CU = the 12 waves the kernel's 148 registers allow: every wave resident from the start, each keeping its next row
  // in flight.  Measured at 64 k rows × 768, in place + dropout_bwd(dx) (run r03j, 491 MB per launch): round 2's kernel 101 µs; this
  // one with 512 / 768 / 1024 / 1536 / 2048 blocks 90 / 85 / 95 / 89 / 90 µs (5.8 TB/s at 768); without the second output 77 → 64.5 µs
  // (6.1 TB/s — the float4-copy rate of the chip is 6.3).  Few adders per dw address as a side effect.
  if (grid > 768) grid = 768;
#ifdef LAKO_EXPERIMENTS
  if (const char* e = getenv("LAKO_RMS_GRID")) grid = std::min(rows_grid(rows), atoi(e));
  if (const char* e = getenv("LAKO_RMS_OLD"); e && e[0] == '1') {
    DISPATCH_T(dtype, hipLaunchKernelGGL((rmsnorm_bwd_old_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                                         (const T*)dy, (const T*)x, w, rstd, (const T*)dres, (T*)dx, dw, rows, d, dr, (T*)dx_drop,
                                         dr_out));
    LAKO_LAUNCH_CHECK();
    return LAKO_OK;
  }
#endif
  DISPATCH_T(dtype, hipLaunchKernelGGL((rmsnorm_bwd_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                                       (const T*)dy, (const T*)x, w, rstd, (const T*)dres, (T*)dx, dw, rows, d, dr, (T*)dx_drop,
                                       dr_out, lako_det::on() ? 1 : 0));
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

extern "C" int lako_embed_fwd(const int64_t* ids, const void* table, void* out, int64_t n_tok, int d, int64_t vocab,
                              int dtype, lako_dropout_t drop, lako_stream_t stream) {
  CHECK_DTYPE("lako_embed_fwd", dtype);
  LAKO_CHECK_ARG(n_tok > 0 && d > 0 && d % 8 == 0 && vocab > 0, "lako_embed_fwd: bad dims");
  LAKO_CHECK_ALIGN(table, 16);
  LAKO_CHECK_ALIGN(out, 16);
  DropDev dr = make_drop(drop);
  DISPATCH_T(dtype, hipLaunchKernelGGL((embed_fwd_kernel<T>), dim3(rows_grid(n_tok)), dim3(256), 0, (hipStream_t)stream,
                                       ids, (const T*)table, (T*)out, n_tok, d, vocab, dr));
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

extern "C" int lako_embed_bwd(const int64_t* ids, const void* dout, float* dtable, int64_t n_tok, int d, int64_t vocab,
                              int dtype, lako_dropout_t drop, lako_stream_t stream) {
  CHECK_DTYPE("lako_embed_bwd", dtype);
  LAKO_CHECK_ARG(n_tok > 0 && d > 0 && vocab > 0, "lako_embed_bwd: bad dims");
  DropDev dr = make_drop(drop);
  DISPATCH_T(dtype, hipLaunchKernelGGL((embed_bwd_kernel<T>), dim3(rows_grid(n_tok)), dim3(256), 0, (hipStream_t)stream,
                                       ids, (const T*)dout, dtable, n_tok, d, vocab, dr));
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

extern "C" int lako_embed_bwd_ordered(const int64_t* ids, const int64_t* perm, const void* dout, float* dtable, int64_t n_tok, int d,
                                      int64_t vocab, int dtype, lako_dropout_t drop, lako_stream_t stream) {
  CHECK_DTYPE("lako_embed_bwd_ordered", dtype);
  LAKO_CHECK_ARG(n_tok > 0 && d > 0 && vocab > 0 && ids && perm, "lako_embed_bwd_ordered: bad arguments");
  DropDev dr = make_drop(drop);
  DISPATCH_T(dtype, hipLaunchKernelGGL((embed_bwd_ordered_kernel<T>), dim3(rows_grid(n_tok)), dim3(256), 0, (hipStream_t)stream,
                                       ids, perm, (const T*)dout, dtable, n_tok, d, vocab, dr));
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

extern "C" int lako_deterministic(void) { return lako_det::on() ? 1 : 0; }

extern "C" int lako_dropout_apply(const void* x, void* y, int64_t n, int dtype, lako_dropout_t drop,
                                  lako_stream_t stream) {
  CHECK_DTYPE("lako_dropout_apply", dtype);
  LAKO_CHECK_ARG(n > 0 && n % 8 == 0, "lako_dropout_apply: n must be a positive multiple of 8");
  LAKO_CHECK_ALIGN(x, 16);
  LAKO_CHECK_ALIGN(y, 16);
  DropDev dr = make_drop(drop);
  DISPATCH_T(dtype, hipLaunchKernelGGL((dropout_apply_kernel<T>), dim3(elems_grid(n, 8)), dim3(256), 0,
                                       (hipStream_t)stream, (const T*)x, (T*)y, n, dr));
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

extern "C" int lako_cast(const float* src, void* dst, int64_t n, int dtype, lako_stream_t stream) {
  CHECK_DTYPE("lako_cast", dtype);
  LAKO_CHECK_ARG(n > 0, "lako_cast: n <= 0");
  LAKO_CHECK_ALIGN(src, 16);
  LAKO_CHECK_ALIGN(dst, 16);
  DISPATCH_T(dtype, hipLaunchKernelGGL((cast_kernel<T>), dim3(elems_grid(n, 8)), dim3(256), 0, (hipStream_t)stream, src,
                                       (T*)dst, n));
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

extern "C" int lako_transpose_cast(const float* src, void* dst, int64_t rows, int64_t cols, int dtype,
                                   lako_stream_t stream) {
  CHECK_DTYPE("lako_transpose_cast", dtype);
  LAKO_CHECK_ARG(rows > 0 && cols > 0, "lako_transpose_cast: bad dims");
  dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
  LAKO_CHECK_ARG(grid.y < 65536, "lako_transpose_cast: too many rows");
  DISPATCH_T(dtype, hipLaunchKernelGGL((transpose_cast_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, src, (T*)dst,
                                       rows, cols));
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

extern "C" int lako_transpose_cast_batched(const void* src_base, int src_dtype, void* dst_base, const int64_t* desc,
                                           const int32_t* tile_prefix, int n, int total_tiles, int dtype,
                                           lako_stream_t stream) {
  CHECK_DTYPE("lako_transpose_cast_batched", dtype);
  CHECK_DTYPE("lako_transpose_cast_batched (source)", src_dtype);
  LAKO_CHECK_ARG(src_dtype == LAKO_F32 || dtype == LAKO_BF16, "lako_transpose_cast_batched: a bf16 source has bf16 copies only");
  LAKO_CHECK_ARG(n > 0 && total_tiles > 0 && desc && tile_prefix, "lako_transpose_cast_batched: empty table");
  LAKO_CHECK_ALIGN(src_base, 8);
  LAKO_CHECK_ALIGN(dst_base, 16);
  if (src_dtype == LAKO_BF16) {
    hipLaunchKernelGGL((transpose_cast_batched_kernel<bf16_t, bf16_t>), dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)src_base, (bf16_t*)dst_base, desc, tile_prefix, n);
  } else {
    DISPATCH_T(dtype, hipLaunchKernelGGL((transpose_cast_batched_kernel<T, float>), dim3((unsigned)total_tiles), dim3(256), 0,
                                         (hipStream_t)stream, (const float*)src_base, (T*)dst_base, desc, tile_prefix, n));
  }
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

// 16 bytes of accumulator per (device, stream): calls on one stream are ordered, calls on different streams get different scratch
static CeScratch* ce_scratch(hipStream_t stream) {
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, CeScratch*> tab;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  CeScratch*& p = tab[{dev, stream}];
  if (!p && hipMalloc(reinterpret_cast<void**>(&p), sizeof(CeScratch)) != hipSuccess) p = nullptr;
  return p;
}

extern "C" int lako_ce_fwd_bwd(const float* logits, const int64_t* labels, float* loss_out, void* dlogits,
                               const float* upstream, int64_t M, int64_t V, int dtype, lako_stream_t stream) {
  CHECK_DTYPE("lako_ce_fwd_bwd", dtype);
  LAKO_CHECK_ARG(M > 0 && V > 0 && V % 4 == 0, "lako_ce_fwd_bwd: M=%lld V=%lld (V %% 4 == 0)", (long long)M, (long long)V);
  LAKO_CHECK_ALIGN(logits, 16);
  CeScratch* sc = ce_scratch((hipStream_t)stream);
  if (!sc) {
    lako_set_error("lako_ce_fwd_bwd: could not allocate the loss accumulator");
    return LAKO_E_LAUNCH;
  }
  hipLaunchKernelGGL(ce_init_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, loss_out, labels, M, sc);
  DISPATCH_T(dtype, hipLaunchKernelGGL((ce_kernel<T>), dim3((unsigned)M), dim3(256), 0, (hipStream_t)stream, logits,
                                       labels, loss_out, (T*)dlogits, upstream, V, sc));
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

extern "C" int lako_sumsq(const float* g, int64_t n, float* out, lako_stream_t stream) {
  LAKO_CHECK_ARG(n > 0, "lako_sumsq: n <= 0");
  LAKO_CHECK_ALIGN(g, 16);
  int grid = elems_grid(n, 4);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, g, n, out, lako_det::on() ? 1 : 0);
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

extern "C" int lako_adamw_step(float* p, const float* g, float* m, float* v, void* shadow, int64_t n, float lr,
                               float beta1, float beta2, float eps, float weight_decay, const float* gnorm_sq,
                               float max_norm, float grad_scale, int shadow_dtype, lako_stream_t stream) {
  CHECK_DTYPE("lako_adamw_step", shadow_dtype);
  LAKO_CHECK_ARG(n > 0 && n % 4 == 0, "lako_adamw_step: n must be a positive multiple of 4");
  LAKO_CHECK_ALIGN(p, 16);
  LAKO_CHECK_ALIGN(g, 16);
  LAKO_CHECK_ALIGN(m, 16);
  LAKO_CHECK_ALIGN(v, 16);
  if (shadow) LAKO_CHECK_ALIGN(shadow, 8);
  DISPATCH_T(shadow_dtype,
             hipLaunchKernelGGL((adamw_kernel<T>), dim3(elems_grid(n, 4)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                                (T*)shadow, n, lr, beta1, beta2, eps, weight_decay, gnorm_sq, max_norm, grad_scale));
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

extern "C" int lako_relpos_expand(const float* table, const int32_t* lut, float* rel, int H, int R, int nb,
                                  lako_stream_t stream) {
  LAKO_CHECK_ARG(H > 0 && R > 0 && nb > 0, "lako_relpos_expand: bad dims");
  hipLaunchKernelGGL(relpos_expand_kernel, dim3((H * R + 255) / 256), dim3(256), 0, (hipStream_t)stream, table, lut, rel,
                     H, R, nb);
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}
extern "C" int lako_relpos_reduce(const float* drel, const int32_t* lut, float* dtable, int H, int R, int nb,
                                  lako_stream_t stream) {
  LAKO_CHECK_ARG(H > 0 && R > 0 && nb > 0, "lako_relpos_reduce: bad dims");
  hipLaunchKernelGGL(relpos_reduce_kernel, dim3((nb * H + 255) / 256), dim3(256), 0, (hipStream_t)stream, drel, lut,
                     dtable, H, R, nb);
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

extern "C" int lako_shift_right(const int64_t* labels, int64_t* dec_ids, int B, int T, lako_stream_t stream) {
  LAKO_CHECK_ARG(B > 0 && T > 0, "lako_shift_right: bad dims");
  hipLaunchKernelGGL(shift_right_kernel, dim3((B * T + 255) / 256), dim3(256), 0, (hipStream_t)stream, labels, dec_ids,
                     B, T);
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

extern "C" int lako_pack_ids(const int64_t* ids, const int32_t* off, int64_t* out, int BN, int L, lako_stream_t stream) {
  LAKO_CHECK_ARG(ids && off && out && BN > 0 && L > 0, "lako_pack_ids: bad arguments");
  hipLaunchKernelGGL(pack_ids_kernel, dim3(BN), dim3(256), 0, (hipStream_t)stream, ids, off, out, BN, L);
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

extern "C" int lako_greedy_step(const float* logits, int64_t V, int B, int64_t* seq, int64_t seq_ld, int pos,
                                int64_t* next_ids, uint8_t* done, int32_t* n_done, int64_t eos_id, int64_t pad_id,
                                lako_stream_t stream) {
  LAKO_CHECK_ARG(B > 0 && V > 0 && pos >= 0 && pos < seq_ld, "lako_greedy_step: bad dims");
  hipLaunchKernelGGL(greedy_step_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, V, seq, seq_ld, pos,
                     next_ids, done, eos_id, pad_id);
  hipLaunchKernelGGL(count_done_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, done, B, n_done);
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

extern "C" int lako_fact_scores(const float* scores, const uint8_t* mask, const int64_t* ids, double* out, int B, int H, int n_layers,
                                int layer0, int layers_used, int N, int L, int passage, int ids_passage, int n_context, int style,
                                lako_stream_t stream) {
  LAKO_CHECK_ARG(scores && mask && ids && out, "lako_fact_scores: null tensor");
  LAKO_CHECK_ARG(B > 0 && H > 0 && n_layers > 0 && N > 0 && L >= 3 && n_context > 0, "lako_fact_scores: bad dims");
  LAKO_CHECK_ARG(layer0 >= 0 && layers_used > 0 && layer0 + layers_used <= n_layers, "lako_fact_scores: bad layer range");
  LAKO_CHECK_ARG(passage >= 0 && passage < N && ids_passage >= 0 && ids_passage < N, "lako_fact_scores: bad passage index");
  LAKO_CHECK_ARG(style >= 0 && style <= 2, "lako_fact_scores: style 0 mean | 1 max | 2 21mean");
  const size_t lds = (size_t)L * 8 + (size_t)(2 * n_context + 1) * 4;
  LAKO_CHECK_ARG(lds <= 60 * 1024, "lako_fact_scores: L / n_context too large for one workgroup's LDS");
  hipLaunchKernelGGL(fact_scores_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, scores, mask, ids, out, H, n_layers, layer0,
                     layers_used, N, L, passage, ids_passage, n_context, style);
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}
