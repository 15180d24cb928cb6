// MFMA GEMMs for the FiD-T5 reader: every nn.Linear(bias=False) of HF5:304,325-326,367 (q,k,v,o),
// HF5:83-94 (wi,wo) and HF5:1047 (lm_head), forward and backward.
//
//   gemm_nt : C[M,N] = epilogue(alpha * A[M,K] · B[N,K]ᵀ)        activations × weights ([out,in] layout)
//   gemm_tn : C[M,N] += alpha * A[K,M]ᵀ · B[K,N]   (fp32)          weight gradients dW = dYᵀ·X
//
// Kernels (gfx950), all on one LDS image: K stepped in 128-BYTE slices so that bf16 (k = 64, v_mfma_f32_16x16x32_bf16) and fp32
// (k = 32, the exact-f32 v_mfma_f32_16x16x4_f32: parity mode) share the staging code; global → LDS by LDS-DMA (16 B per lane, no
// register round trip; ragged edges zero-filled by the buffer range check or clamped), double-buffered; the image is lane-linear
// (a DMA constraint), so the bank-conflict swizzle sits on the per-lane SOURCE address and again on the fragment read.
//   gemm_nt_kernel        persistent; 256 × 256 tiles on 8 waves (2 × 4, 128 × 64 per wave) for everything with >= 256 tiles, 288 × 256
//                         (MT = 9) where that saves a round of the chip or the tail launch (launch_nt prices both per call), 128 × 128
//                         on 4 waves for small problems; fused epilogues (alpha, ReLU, dropout, residual, ReLU/dropout-backward
//                         mask), LDS-transposed 16-byte stores; XCD-aware tile order
//   gemm_nt_skinny_kernel M <= 256 rows (the decoder): 32 × 32 tiles, eight waves split K, operands straight into MFMA fragments
//   gemm_nt_ring_kernel   a handful of 128 × 128 tiles: 4-slot LDS ring (LM head, row tails where the 288-row plan does not apply)
//   gemm_tn256_kernel     256 × 256 tiles, both operands k-strided → ds_read_b64_tr_b16; grouped launches of up to 48 problems; K-splits
//                         by atomics, or (round 4) whole rounds of full-K tiles by plain 16-byte adds + a finely cut remainder
//   gemm_nt_mx_kernel     MX block-scaled fp8 operands (config 5)
#include <type_traits>

#include <algorithm>
#include <map>
#include <mutex>

#include "common.h"

// Timing experiments (bench_ops / tools/*_probe.py) that produce WRONG results — K-loop DMA switched off, stores dropped or
// redirected, cache-policy bits on the stores, atomics skipped — exist only in a build with -DLAKO_EXPERIMENTS
// (csrc/build.sh LAKO_EXPERIMENTS=1 → liblako_hip_exp.so, never loaded by the product).  In the release library the
// tests below are compile-time zero and lako_set_tuning rejects the keys that would set them.
// cache-policy bits of the NT kernels' LDS-DMA loads (gfx940 encoding of the builtin's aux operand: 1 = sc0, 2 = nt, 16 = sc1): 0 in the
// product; other values only in throw-away A/B builds (-DLAKO_LOAD_AUX=…, tools/gemm_pp_probe.py)
// m-tile rows of MFMAs of a K-step's second half that the TN kernel issues BEFORE the wait for the next K-slice + barrier (8 = all of them);
// 6 in the product, other values in throw-away A/B builds (-DLAKO_TN_WAIT_ROWS=…)
#ifndef LAKO_TN_WAIT_ROWS
#define LAKO_TN_WAIT_ROWS 6
#endif
// m-tile rows of MFMAs of a K-step's second half that the NT kernel issues AFTER the wait for the next K-slice + barrier (0 = MT / 4, the
// product; other values in throw-away A/B builds)
#ifndef LAKO_NT_WAIT_Q
#define LAKO_NT_WAIT_Q 0
#endif
#ifndef LAKO_LOAD_AUX
#define LAKO_LOAD_AUX 0
#endif
#ifndef LAKO_NT_ROLL
#define LAKO_NT_ROLL 0        // 1: rolling A fragments in the plain bf16 256² / 288-row kernels (see the K loop).  Measured: no gain (± 1 % per launch,
                              // profiles/r05c_gemm_roll.txt); A/B build only
#endif
#ifndef LAKO_TN_SPREAD
#define LAKO_TN_SPREAD 0      // the same for the 256² weight-gradient kernel: measured SLOWER (7.14 -> 7.65 ms per step, two alternations on one box,
                              // profiles/r05c_gemm_spread_dma.txt) — its K loop is 747 steps deep and already runs at 0.47 of peak; A/B build only
#endif
#ifndef LAKO_NT_SPREAD
// 1 (default, round 5): the K-slice DMA of the PLAIN 256² / 288-row kernels is dealt out one piece per row of K-half 0's MFMAs instead of a
// burst of eight per wave.  The stamps of one workgroup (tools/gemm_stamps.py, profiles/r05c_gemm_stamps.txt) show a wave spending ≈ 1 300 of a
// K-step's ≈ 3 900 cycles issuing its eight pieces back to back — the vector-memory queue is full, the wave's MFMAs wait behind the
// issue — and the early half of the waves then ≈ 830 cycles at the barrier for the late half.  Same-box A/B of two builds
// (tools/gemm_lib_ab.sh, profiles/r05c_gemm_spread_dma.txt): QKV 193 → 186 µs, dX of QKV 155 → 150, dX of FFN 200 → 195, 8192³ 870 → 860;
// the step 37.16 → 36.90 ms.  0: the staggered bursts of rounds 1 – 4.  2: also in the side-operand kernels — measured SLOWER there
// (dpre 278 → 286 µs: the side pass's counted waits want the pieces together) and not correct yet; A/B only.
#define LAKO_NT_SPREAD 1
#endif
#ifdef LAKO_EXPERIMENTS
#define NT_DBG(a, bits) ((a).debug & (bits))
#else
#define NT_DBG(a, bits) 0
#endif

namespace {

// f(integral_constant<int, 0>) … f(integral_constant<int, N-1>): a compile-time-indexed unrolled loop
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

constexpr int TM = 128, TN_ = 128, TKB = 128;  // TN-kernel tile rows / cols; K bytes per step (both kernels)
constexpr int TILE_BYTES = TM * TKB;            // 16 KiB per operand per buffer
constexpr int GEMM_LDS = 4 * TILE_BYTES;        // A,B × 2 buffers (TN kernel)

// (experiments build, gemm_nt_debug bit 6) s_memtime stamps of workgroup 0 of the 256² / 288-row kernel: [wave][K-step < 128][point < 8] in
// the buffer the (unused, non-QUEUE) queue pointer carries — tools/gemm_stamps.py reads it back
#define NT_STAMP(P)                                                                                                              \
  do {                                                                                                                           \
    if (NT_DBG(a, 64) && blockIdx.x == 0 && (threadIdx.x & 63) == 0 && kst < 128)                                               \
      reinterpret_cast<unsigned long long*>(a.queue)[((threadIdx.x >> 6) * 128 + kst) * 8 + (P)] = __builtin_amdgcn_s_memtime();   \
  } while (0)

struct NtArgs {
  const char* A;
  const char* B;
  char* C;
  const char* resid;
  const char* aux;
  int M, N, K;
  int64_t lda, ldb, ldc, ldr, ldaux;  // in elements
  float alpha, aux_scale, drop_scale;
  int flags;
  uint32_t drop_thresh, drop_key;
  int tiles_m, tiles_n;
  int stagger;
  int debug;      // timing experiments only (bench_ops): bit 0 = skip the K-loop LDS-DMA, bit 1 = all workgroups stream tile (0,0), bit 2 = never wait for the DMA, bit 3 = no epilogue stores (results are garbage)
  int split_k, k_chunk;   // ring kernel with LAKO_EPI_ATOMIC only: K split over split_k workgroups per tile, k_chunk BYTES each
  int64_t row0;   // rows [row0, row0 + M) of a larger problem (tail launch): only the dropout element index needs it
  int group_m;    // >0: tile ids walk bands of group_m tile-rows column-major (an XCD's 32 resident tiles form a ≈group_m × 32/group_m block)
  int store_aux;  // cache-policy bits for the 256² kernel's 16-byte output stores (store_b128_policy)
  int epi_direct; // gemm_nt4_kernel: 1 = the dropout epilogues without a side operand straight from the accumulator layout, 0 = through the LDS transposition
  int dephase;    // (phases << 16) | ticks: workgroup w of an XCD starts (w mod phases)·ticks·10 ns late (breaks the lockstep of main loops / epilogues)
  int wide_epi;   // 256² bf16 tile: LDS-transposed epilogue with 16-B row-major stores (needs N, ldc, ldr, ldaux % 8 == 0)
  int glds;       // 1: K-slice pieces by global_load_lds where the slice is whole (tuning nt_glds)
  const float* norm_w;   // LAKO_EPI_NORM_A (skinny kernel): T5 RMSNorm weight [K]; A is the un-normalised input, the operand is bf16(w · (A · rstd))
  float norm_eps;
  char* norm_out;        // optional bf16 [M, K] (row stride norm_ld): the normalised rows, written by the workgroups of the first tile column
  int64_t norm_ld;
  float* norm_rs;        // optional fp32 [M]: rstd
  int* queue;     // QUEUE instantiation: [0..7] per-XCD tile tickets, [8] workgroups finished (all zero between launches)
};

#ifdef LAKO_EXPERIMENTS
constexpr size_t NT_STAMP_BYTES = 8 * 128 * 8 * 8;
static void* lako_exp_nt_stamp_buffer() {
  static void* buf = nullptr;
  if (!buf) {
    (void)hipMalloc(&buf, NT_STAMP_BYTES);
    (void)hipMemset(buf, 0, NT_STAMP_BYTES);
  }
  return buf;
}
#endif

// bijective XCD-aware remap (blocks b and b+8 share an XCD): give each XCD a contiguous id range
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (bid >> 3);
}

// tile id → (tile row, tile col).  Row-major ids make the 32 tiles resident on one XCD a 1×32 strip when tiles_n ≥ 32:
// 33 distinct operand panels for 32 tiles.  Banded ids (group_m rows per band, column-major inside the band) make
// them a group_m × (32/group_m) block: 12 panels at group_m = 4 or 8, so ≥ 80 % of the LDS-DMA requests can hit L2.
__device__ __forceinline__ void tile_coords(int tile, int tiles_m, int tiles_n, int group_m, int& tm, int& tn) {
  if (group_m <= 0) {
    tm = tile / tiles_n;
    tn = tile % tiles_n;
    return;
  }
  const int band = group_m * tiles_n, b = tile / band, r = tile - b * band;
  const int rows = min(group_m, tiles_m - b * group_m);
  tm = b * group_m + r % rows;
  tn = r / rows;
}

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0,
                                                   0, 0);
  }
};
template <> struct Mma<float> {
  static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
    f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
    for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e], bf[e], c, 0, 0, 0);
    return c;
  }
};

// Stage one [ROWS rows][128 B] operand slice: LDS chunk (row, cp) holds logical 16-B chunk
// c = cp ^ ((row >> 1) & 7) of that row.  `base` points at (row0, kbyte0) of the operand.  Each
// wave-instruction moves 8 rows (1 KiB); the ROWS/8 instructions are dealt round-robin to the waves.
template <int ROWS, int NWAVES>
__device__ __forceinline__ void stage_rows(char* lds_tile, const char* base, int rows_valid, int64_t ld_bytes,
                                           int kbytes_left, int wave, int lane) {
  uint32_t nrec = (rows_valid > 0 && kbytes_left > 0)
                      ? (uint32_t)((int64_t)(rows_valid - 1) * ld_bytes + (kbytes_left < TKB ? kbytes_left : TKB))
                      : 0u;
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)nrec, 0x00020000);
#pragma unroll
  for (int i = 0; i < (ROWS / 8 + NWAVES - 1) / NWAVES; ++i) {
    int inst = wave + i * NWAVES;
    if ((ROWS / 8) % NWAVES != 0 && inst >= ROWS / 8) break;      // 288-row tiles: 36 pieces over 8 waves (wave-uniform)
    int row = inst * 8 + (lane >> 3);
    int cp = lane & 7;
    int c = cp ^ ((row >> 1) & 7);
    bool ok = (row < rows_valid) && (c * 16 < kbytes_left);
    uint32_t voff = ok ? (uint32_t)(row * ld_bytes + c * 16) : 0xFFFFFFF0u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_tile + inst * 1024), 16, (int)voff, 0, 0, LAKO_LOAD_AUX);
  }
}

// the same, one 1-KiB piece at a time (piece i of this wave), so that the K-loop can deal the pieces out between MFMAs.
// piece_base / stage_piece_at (SIDE kernel): piece i of a wave is piece 0 moved down by i·NWAVES·8 rows (the chunk swizzle repeats
// every 16 rows), so ONE per-lane offset serves all pieces, each with its own descriptor starting at its first row.  Rows past
// the edge fall out of the descriptor's range by themselves (slice_rsrc: the range ends inside row rows_valid − 1 and a row
// stride is at least a row long); chunks past the end of K start from an out-of-range offset.
__device__ __forceinline__ uint32_t piece_base(int64_t ld_bytes, int kbytes_left, int wave, int lane) {
  const int row = wave * 8 + (lane >> 3);
  const int c = (lane & 7) ^ ((row >> 1) & 7);
  return c * 16 < kbytes_left ? (uint32_t)(row * ld_bytes + c * 16) : 0x80000000u;
}
template <int NWAVES>
__device__ __forceinline__ void stage_piece(char* lds_tile, __amdgpu_buffer_rsrc_t rsrc, int rows_valid, int64_t ld_bytes,
                                            int kbytes_left, int wave, int lane, int i) {
  int inst = wave + i * NWAVES;
  int row = inst * 8 + (lane >> 3);
  int cp = lane & 7;
  int c = cp ^ ((row >> 1) & 7);
  bool ok = (row < rows_valid) && (c * 16 < kbytes_left);
  uint32_t voff = ok ? (uint32_t)(row * ld_bytes + c * 16) : 0xFFFFFFF0u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_tile + inst * 1024), 16, (int)voff, 0, 0, LAKO_LOAD_AUX);
}
// the same piece by `global_load_lds` where the whole 128-byte K-slice exists (`full`): rows past the edge are clamped to the last valid
// row (they feed accumulator rows that are never stored), nothing needs the buffer form's zero fill — measured 4–14 % faster per
// launch than `buffer_load … lds` in the same loop (round 4, profiles/r04g_gemm_eight_phase.txt); a partial K-slice (K % 64 != 0: its
// missing chunks must read as zeros) keeps the buffer form
template <int NWAVES>
__device__ __forceinline__ void stage_piece_g(char* lds_tile, const char* base, __amdgpu_buffer_rsrc_t rsrc, int rows_valid, int64_t ld_bytes,
                                              int kbytes_left, int wave, int lane, int i, bool full) {
  const int inst = wave + i * NWAVES;
  const int row = inst * 8 + (lane >> 3);
  const int c = (lane & 7) ^ ((row >> 1) & 7);
  if (full) {
    const uint32_t voff = (uint32_t)(min(row, rows_valid - 1) * ld_bytes + c * 16);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + voff), LDS_PTR(lds_tile + inst * 1024), 16, 0, LAKO_LOAD_AUX);
  } else {
    const bool ok = (row < rows_valid) && (c * 16 < kbytes_left);
    const uint32_t voff = ok ? (uint32_t)(row * ld_bytes + c * 16) : 0xFFFFFFF0u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_tile + inst * 1024), 16, (int)voff, 0, 0, LAKO_LOAD_AUX);
  }
}
// SIDE kernel: ONE per-lane offset (piece_base) serves all pieces of a wave, each piece with its own descriptor
template <int NWAVES>
__device__ __forceinline__ void stage_piece_at(char* lds_tile, __amdgpu_buffer_rsrc_t rsrc, uint32_t vbase, int wave, int i) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_tile + (wave + i * NWAVES) * 1024), 16, (int)vbase, 0, 0, LAKO_LOAD_AUX);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t slice_rsrc(const char* base, int rows_valid, int64_t ld_bytes, int kbytes_left) {
  uint32_t nrec = (rows_valid > 0 && kbytes_left > 0)
                      ? (uint32_t)((int64_t)(rows_valid - 1) * ld_bytes + (kbytes_left < TKB ? kbytes_left : TKB))
                      : 0u;
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)nrec, 0x00020000);
}

// the lane id, recomputed where it is needed (v_mbcnt of an opaque zero: cannot be hoisted, keeps no register alive across the K loop)
__device__ __forceinline__ int fresh_lane() {
  int z = 0;
  asm volatile("" : "+v"(z));
  return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)z));
}

__device__ __forceinline__ u32x4 read_frag_rows(const char* lds_tile, int row, int chunk) {
  int cp = chunk ^ ((row >> 1) & 7);
  return *reinterpret_cast<const u32x4*>(lds_tile + row * TKB + cp * 16);
}

#ifdef LAKO_EXPERIMENTS
// 16-byte buffer store with run-time selectable cache-policy bits (gfx940 encoding of the builtin's aux operand: 1 = sc0, 2 = nt,
// 16 = sc1).  EXPERIMENT, off by default ("gemm_nt_store_aux"): a round of the 256² kernel leaves 4 MB of output per XCD dirty in
// the 4 MB L2 — the same stores aimed at one L2-resident tile (debug bit 4) run the K = 768 shapes 30 % faster, so the cost of
// the epilogue is the write-back traffic, not the store instructions.  Non-temporal / sc1 stores helped on one box
// ([.,768]×[2304,768] 248 → 221 µs, ×[3072,768] 333 → 292 µs) and hurt on another (×[18432,768] 1 754 → 1 855 µs); the whole
// training step moved by 0.2 ms.  tools/store_policy_probe.py repeats the measurement.
__device__ __forceinline__ void store_b128_policy(u32x4 v, __amdgpu_buffer_rsrc_t rs, int voff, int aux) {
  switch (aux) {
    case 0: __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, 0, 0); break;
    case 1: __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, 0, 1); break;
    case 2: __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, 0, 2); break;
    case 3: __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, 0, 3); break;
    case 16: __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, 0, 16); break;
    case 17: __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, 0, 17); break;
    case 18: __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, 0, 18); break;
    default: __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, 0, 19); break;
  }
}
#endif

// Generic epilogue in the accumulator layout (element r of lane (l&15, g) of sub-tile (mt, nt) is
// C[m0 + (wr·MT + mt)·16 + (l&15)][n0 + (wc·NT + nt)·16 + 4g + r]): alpha, ReLU, aux mask, dropout, residual, store / atomic.
template <typename T, typename TO, int WM, int WN, int MT, int NT>
__device__ __forceinline__ void nt_store_tile(const NtArgs& a, f32x4 (&acc)[NT][MT], int m0, int n0, int wr, int wc, int lane) {
  const int r16 = lane & 15, g = lane >> 4;
  TO* C = reinterpret_cast<TO*>(a.C);
  const TO* R = reinterpret_cast<const TO*>(a.resid);
  const T* X = reinterpret_cast<const T*>(a.aux);
  const bool relu = a.flags & LAKO_EPI_RELU, has_res = a.flags & LAKO_EPI_RESID,
             auxm = a.flags & LAKO_EPI_AUXMASK, atomic = a.flags & LAKO_EPI_ATOMIC;
  const bool drop = a.drop_thresh != 0;
  // pass 1: issue EVERY residual / aux load of the wave's sub-tiles before the first use (a load → wait → use
  // chain per sub-tile would pay one memory round trip per 16×16 block); addresses clamped in-bounds
  // (kept packed: 2 registers per sub-tile when both dtypes are bf16)
  constexpr bool PACKED = sizeof(T) == 2 && sizeof(TO) == 2;
  using SideT = typename std::conditional<PACKED, bf16x4, f32x4>::type;
  constexpr bool PREFETCH = MT * NT <= 16;   // 256² tile: 64 more live registers would spill next to the wide epilogue (measured equal without)
  SideT side[PREFETCH ? NT : 1][PREFETCH ? MT : 1];
  if (PREFETCH && (has_res || auxm)) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = min(m0 + (wr * MT + mt) * 16 + r16, a.M - 1);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = min(n0 + (wc * NT + nt) * 16 + 4 * g, a.N - 4);
        if constexpr (PACKED && PREFETCH) {
          side[nt][mt] = has_res ? *reinterpret_cast<const bf16x4*>(R + (int64_t)m * a.ldr + n)
                                 : *reinterpret_cast<const bf16x4*>(X + (int64_t)m * a.ldaux + n);
        } else if constexpr (PREFETCH) {
          side[nt][mt] = has_res ? load4(R + (int64_t)m * a.ldr + n) : load4(X + (int64_t)m * a.ldaux + n);
        }
      }
    }
  }
  auto side_f32 = [&](int nt, int mt, int m, int n) -> f32x4 {
    if constexpr (PREFETCH && PACKED) {
      const bf16x4 q = side[nt][mt];
      return f32x4{(float)q[0], (float)q[1], (float)q[2], (float)q[3]};
    } else if constexpr (PREFETCH) {
      return side[nt][mt];
    } else {
      return has_res ? load4(R + (int64_t)m * a.ldr + n) : load4(X + (int64_t)m * a.ldaux + n);
    }
  };
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + (wr * MT + mt) * 16 + r16;
    if (m >= a.M) continue;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = n0 + (wc * NT + nt) * 16 + 4 * g;
      if (n >= a.N) continue;  // N % 4 == 0: a group of 4 is all in or all out
      f32x4 v = acc[nt][mt] * a.alpha;
      if (relu) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
      }
      if (auxm) {
        const f32x4 x = side_f32(nt, mt, m, n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = x[r] > 0.f ? v[r] * a.aux_scale : 0.f;
      }
      if (drop) {
        uint64_t idx = (uint64_t)(a.row0 + m) * (uint64_t)a.N + (uint64_t)n;
        bool kp[4];
        lako_keep4(a.drop_key, idx >> 2, a.drop_thresh, kp);   // n % 4 == 0: one quad
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = kp[r] ? v[r] * a.drop_scale : 0.f;
      }
      if (has_res) v += side_f32(nt, mt, m, n);
      TO* cp = C + (int64_t)m * a.ldc + n;
      if constexpr (sizeof(TO) == 4) {
        if (atomic) {
#pragma unroll
          for (int r = 0; r < 4; ++r) atomicAdd(reinterpret_cast<float*>(cp) + r, v[r]);
          continue;
        }
      }
      if (!NT_DBG(a, 8)) store4(cp, v);
    }
  }
}

// Workgroup tile = (WM·MT·16) × (WN·NT·16); WM×WN waves, each owning MT×NT MFMA 16×16 tiles.
//   <2,2,4,4> 128×128, 4 waves, 64 KiB LDS (2 workgroups / CU)      — small / skinny problems
//   <2,4,8,4> 256×256, 8 waves, 128 KiB LDS (1 workgroup / CU)      — half the LDS+L2 bytes per FLOP
//   SIDE (256² bf16 only): the residual / aux operand of the epilogue is LDS-DMA'd in four 32-row passes (see the epilogue)
//   QUEUE: the tiles after a workgroup's first are PULLED from per-XCD ticket counters instead of strided by the grid size (below)
//   PP (round 4; 8 waves, bf16, K >= 128): the 8-phase main loop — the two waves of a SIMD (w and w + 4) alternate 16-MFMA clusters and
//       memory segments across barriers, counted waits keep three half-tiles of LDS-DMA in flight (see the loop below)
template <typename T, typename TO, int WM, int WN, int MT, int NT, bool SIDE = false, bool QUEUE = false, bool PP = false>
__global__ __launch_bounds__(WM * WN * 64) void gemm_nt_kernel(NtArgs a) {
  static_assert(!PP || (WM * WN == 8 && !QUEUE), "PP: 8 waves, static tile order");
  constexpr int NW = WM * WN, BM = WM * MT * 16, BN = WN * NT * 16;
  constexpr int A_BYTES = BM * TKB, B_BYTES = BN * TKB, BUF = A_BYTES + B_BYTES;
  // SIDE: rows of a side pass (32 = two m-tiles; odd MT — the 288-row tile — takes one m-tile per pass, which also halves the slots so that
  // they fit behind the two 68-KiB K-slice buffers), its DMA pieces (= the 16-byte stores of a finished pass) and the bytes of one slot
  constexpr int PH = MT % 2 == 0 ? 32 : 16, SP = PH / 8, SLOT = PH * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave / WN, wc = wave % WN;
  const int nwg = a.tiles_m * a.tiles_n;
  const int64_t lda_b = a.lda * sizeof(T), ldb_b = a.ldb * sizeof(T);
  const int kbytes = a.K * (int)sizeof(T);
  const int nk = (kbytes + TKB - 1) / TKB;
  const int r16 = lane & 15, g = lane >> 4;

  // PERSISTENT: the grid is at most (CUs × resident workgroups); each workgroup walks tiles
  // v, v + G, v + 2G, …  The first K-slice of the NEXT tile is DMA'd into the free LDS buffer during the
  // last K-step of the current tile, so neither its HBM latency nor this tile's epilogue stalls the MFMAs.
  // v = xcd_remap(blockIdx): workgroups of one XCD get consecutive tile ids (they share A row-panels in L2).
  int tile = xcd_remap(blockIdx.x, gridDim.x);
  int tm_, tn_;
  tile_coords(tile, a.tiles_m, a.tiles_n, a.group_m, tm_, tn_);
  int m0 = tm_ * BM, n0 = tn_ * BN;
  int rows_a = min(BM, a.M - m0), rows_b = min(BN, a.N - n0);
  if (a.dephase > 0) {
    const int ph = (blockIdx.x >> 3) % (a.dephase >> 16);
    if (ph) {
      const uint64_t t0 = __builtin_amdgcn_s_memrealtime(), dt = (uint64_t)((a.dephase & 0xffff) * ph);   // 100 MHz
      while (__builtin_amdgcn_s_memrealtime() - t0 < dt) __builtin_amdgcn_s_sleep(8);
    }
  }
  stage_rows<BM, NW>(smem, a.A + (int64_t)m0 * lda_b, rows_a, lda_b, kbytes, wave, lane);
  stage_rows<BN, NW>(smem + A_BYTES, a.B + (int64_t)n0 * ldb_b, rows_b, ldb_b, kbytes, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;
  // epilogue scratch of a wave: 8 KiB of the K-slice buffer the finished tile no longer reads; where that buffer is smaller than
  // 8 waves × 8 KiB (192-row tiles: 56 KiB) the last wave(s) use an extra region behind the buffers (and the SIDE slots)
  constexpr int SCR_IN_BUF = BUF / 8192 < NW ? BUF / 8192 : NW;
  auto wave_scratch = [&](int free_buf) -> char* {
    return wave < SCR_IN_BUF ? smem + free_buf * BUF + wave * 8192 : smem + 2 * BUF + (SIDE ? NW * SLOT : 0) + (wave - SCR_IN_BUF) * 8192;
  };

  // SIDE: pass p = rows [32p, 32p + 32) of this wave's 128×64 region of the residual / aux tile → 4 KiB of LDS, 16-B chunk c of
  // row r at position c ^ (r & 7) (the accumulator-layout reads below are then conflict-free); 4 DMA instructions, always
  // issued (rows / columns past the edge are out of the descriptor's range: they move nothing), so that the waits further
  // down can be COUNTED.  The per-lane offset is rebuilt from an opaque lane id at every call: shared with the epilogue's
  // calls it would stay live across the main loop (256 registers, spills).
  auto side_issue = [&](int pass, char* slot, int tm0, int tn0) {
    if constexpr (SIDE) {
      const bool res = a.flags & LAKO_EPI_RESID;
      const char* sp = res ? a.resid : a.aux;
      const int ld_b = (int)(res ? a.ldr : a.ldaux) * 2;
      const int rows_v = min(BM, a.M - tm0), cols_v = min(BN, a.N - tn0);
      const auto rs = lds_dma_rsrc(sp + (int64_t)tm0 * ld_b + (int64_t)tn0 * 2, (uint32_t)((rows_v - 1) * ld_b + cols_v * 2));
      const int l = fresh_lane();
      const int col_t = wc * 64 + (((l & 7) ^ ((l >> 3) & 7)) * 8);
      const uint32_t vb = col_t < cols_v ? (uint32_t)((wr * (MT * 16) + (l >> 3)) * ld_b + col_t * 2) : 0x80000000u;
#pragma unroll
      for (int j = 0; j < SP; ++j) lds_dma16(slot + j * 1024, rs, vb + (uint32_t)((pass * PH + j * 8) * ld_b));
    }
  };

  int kst = 0;      // (stamps only) K-steps so far, over all tiles of the workgroup
  while (true) {
    int next_tile = tile + gridDim.x;
    bool has_next = next_tile < nwg;
    tile_coords(has_next ? next_tile : 0, a.tiles_m, a.tiles_n, a.group_m, tm_, tn_);
    int nm0 = tm_ * BM, nn0 = tn_ * BN;
    // QUEUE (data-parallel overlap: RCCL kernels hold some CUs while this launch runs, so not all of its 256 workgroups are
    // resident at once).  With tiles strided by the grid size a workgroup that starts late still owes its whole list and the launch
    // ends a full pass later; here only a workgroup's FIRST tile is static, the rest of an XCD's share — the same tile ids the
    // striding gives that XCD, so the operand panels stay in its L2 — is handed out by a ticket counter per XCD: late workgroups
    // find the counter exhausted.  The ticket (one returning atomic by thread 0) is requested at the tile's start and consumed at the
    // top of its last K-step, in time for the next tile's first prefetch.
    int ticket = 0;
    if constexpr (QUEUE) {
      if (threadIdx.x == 0) ticket = atomicAdd(&a.queue[blockIdx.x & 7], 1);
    }
    // debug bit 1: every workgroup streams tile (0, 0)'s operands — all requests hit L2 (timing experiment)
    const char* Abase = a.A + (NT_DBG(a, 2) ? 0 : (int64_t)m0 * lda_b);
    const char* Bbase = a.B + (NT_DBG(a, 2) ? 0 : (int64_t)n0 * ldb_b);

    f32x4 acc[NT][MT];  // element r of lane (l&15, g): C[m = mt*16 + (l&15)][n = nt*16 + 4g + r]
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if constexpr (PP) {
      // 8-PHASE main loop (round 4; the schedule of cdna_hip_programming.md § "The 256² 8-phase template", rebuilt on this kernel's
      // LDS image, accumulator layout and epilogues).  A K-step (64 k) is four phases; in each a wave
      //     reads the fragments of ONE quadrant of its 128 × 64 tile  →  issues 2 LDS-DMA pieces (ONE half-tile per workgroup and
      //     phase)  →  s_barrier  →  16 MFMAs (the quadrant × both K-halves)  →  s_barrier
      // and waves 4–7 — the SIMD partners of waves 0–3 — run ONE BARRIER behind, so that on every SIMD one wave is inside its MFMA
      // cluster while the other reads fragments and issues DMA (16-MFMA granularity, enforced by the barriers: the coarse
      // half-K-step stagger without them measured no gain, profiles/r04c_gemm_ping_pong.txt).  The DMA stream runs 1¾ K-steps
      // ahead and is retired by ONE counted wait per K-step (vmcnt(6): three half-tiles stay in flight across the barriers).
      //   quadrants of a wave (mh: m-tiles 4mh…4mh+3, nh: n-tiles 2nh, 2nh+1):  phase 0 (0,0) · 1 (0,1) · 2 (1,1) · 3 (1,0)
      //   fragment reads:  phase 0: B-half 0 (4, issued first, retired by lgkmcnt(8) before the barrier) + A-half 0 (8) · 1: B-half 1 (4)
      //                    · 2: A-half 1 (8) · 3: none (B-half 0 is still in registers)
      //   half-tiles (128 rows of the A or B image each = the rows ALL waves read for that half):
      //     phase 0 of K-step k stages A-half 1 of k + 1;  phases 1, 2, 3 stage B-half 0, A-half 0, B-half 1 of k + 2
      //     (each into a slot whose last fragment reads are at least two phases old, or one phase and retired before the barrier)
      //   phase 3 of k waits vmcnt(6): everything but the three half-tiles staged in phases 1–3 has landed — all of K-step k + 1,
      //     which is read from the next phase on (a staged buffer is read one phase AFTER the wait that retires it, never in it)
      // At the end of a tile the stream continues with the next tile's first K-step (staged during this tile's last two K-steps);
      // the three half-tiles of its second K-step are staged after the epilogue, whose scratch is the buffer they go to.
      const bool late = wave >= NW / 2;
      u32x4 af[4][2], b0f[2][2], b1f[2][2];       // [tile in the half][K-half]
      // rows a wave stages: piece i (0, 1) of a half-tile, 8 image rows each (see the half-tile definition above)
      auto a_row = [&](int mh, int i) { return i * 128 + mh * 64 + wave * 8; };
      auto b_row = [&](int nh, int i) { const int j = wave + i * 8; return (j >> 2) * 64 + nh * 32 + (j & 3) * 8; };
      // stage half-tile `which` (0 B0, 1 A0, 2 B1, 3 A1) of K-step kk of this tile — or, when kk == nk and the workgroup has another
      // tile, of that tile's first K-step — into the buffer of parity (cur + kk); returns false when there is nothing to stage
      auto stage_half = [&](int k, int d, int which) -> bool {
        const int kk = k + d;                 // K-step of this tile (kk == nk: the next tile's first)
        const bool nxt = kk >= nk;
        if (nxt && !(has_next && kk == nk)) return false;
        const int koff = nxt ? 0 : kk * TKB;
        const int rows_v = (which & 1) ? (nxt ? min(BM, a.M - nm0) : rows_a) : (nxt ? min(BN, a.N - nn0) : rows_b);
        const int64_t ld = (which & 1) ? lda_b : ldb_b;
        const char* base = (which & 1) ? (nxt ? a.A + (int64_t)nm0 * lda_b : Abase) : (nxt ? a.B + (int64_t)nn0 * ldb_b : Bbase);
        char* img = smem + (cur ^ (d & 1)) * BUF + ((which & 1) ? 0 : A_BYTES);
        const auto rs = slice_rsrc(base + koff, rows_v, ld, kbytes - koff);
        const bool full = a.glds && kbytes - koff >= TKB && rows_v > 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row0 = (which & 1) ? a_row(which >> 1, i) : b_row(which >> 1, i);
          const int row = row0 + (lane >> 3);
          const int c = (lane & 7) ^ ((row >> 1) & 7);
          if (full) {      // (global_load_lds: see stage_piece_g)
            const uint32_t voff = (uint32_t)(min(row, rows_v - 1) * ld + c * 16);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + koff + voff), LDS_PTR(img + row0 * TKB), 16, 0, LAKO_LOAD_AUX);
          } else {
            const bool ok = (row < rows_v) && (c * 16 < kbytes - koff);
            const uint32_t voff = ok ? (uint32_t)(row * ld + c * 16) : 0xFFFFFFF0u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(img + row0 * TKB), 16, (int)voff, 0, 0, LAKO_LOAD_AUX);
          }
        }
        return true;
      };
      auto mma_quad = [&](int mh, u32x4 (&bq)[2][2], int nh) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[nh * 2 + nt][mh * 4 + mt] = Mma<T>::run(bq[nt][kh], af[mt][kh], acc[nh * 2 + nt][mh * 4 + mt]);
        __builtin_amdgcn_s_setprio(0);
      };
      auto read_a = [&](const char* As, int mh) {
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) af[mt][kh] = read_frag_rows(As, (wr * MT + mh * 4 + mt) * 16 + r16, kh * 4 + g);
      };
      auto read_b = [&](const char* Bs, int nh, u32x4 (&bq)[2][2]) {
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) bq[nt][kh] = read_frag_rows(Bs, (wc * NT + nh * 2 + nt) * 16 + r16, kh * 4 + g);
      };
      auto bar = [&]() { __builtin_amdgcn_s_barrier(); };
      auto fence = [&]() { __builtin_amdgcn_sched_barrier(0); };
      // the second K-step's first three half-tiles (first tile: behind the prologue; later tiles: behind the epilogue, whose scratch
      // this buffer was)
      stage_half(0, 1, 0);
      stage_half(0, 1, 1);
      stage_half(0, 1, 2);
      fence();
      if (late) bar();                       // waves 4–7 run one barrier behind from here to the end of the tile's K loop
      for (int k = 0; k < nk; ++k) {
        const char* As = smem + cur * BUF;
        const char* Bs = As + A_BYTES;
        // ---- phase 0: quadrant (0, 0) ----
        read_b(Bs, 0, b0f);
        fence();
        read_a(As, 0);
        fence();
        stage_half(k, 1, 3);
        fence();
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");      // the B-half-0 reads are done: its slot is restaged in the next phase
        bar();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        fence();
        mma_quad(0, b0f, 0);
        fence();
        bar();
        // ---- phase 1: quadrant (0, 1) ----
        read_b(Bs, 1, b1f);
        fence();
        const bool s1 = stage_half(k, 2, 0);
        fence();
        bar();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        fence();
        mma_quad(0, b1f, 1);
        fence();
        bar();
        // ---- phase 2: quadrant (1, 1) ----
        read_a(As, 1);
        fence();
        stage_half(k, 2, 1);
        fence();
        bar();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        fence();
        mma_quad(1, b1f, 1);
        fence();
        bar();
        // ---- phase 3: quadrant (1, 0) ----
        stage_half(k, 2, 2);
        fence();
        if (s1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");     // all but the three half-tiles of phases 1–3 have landed
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (SIDE && k == 0) side_issue(0, smem + 2 * BUF + wave * 4096, m0, n0);   // retired by the next K-step's wait (nk >= 2)
        fence();
        bar();
        mma_quad(1, b0f, 0);
        fence();
        bar();
        cur ^= 1;
      }
      if (!late) bar();                      // the two halves of the workgroup are in step again
      fence();
    } else
    for (int t = 0; t < nk; ++t, ++kst) {
      NT_STAMP(0);
      const char* As = smem + cur * BUF;
      const char* Bs = As + A_BYTES;
      char* An = smem + (cur ^ 1) * BUF;
      if constexpr (QUEUE) {
        // the ticket becomes the next tile id at the top of the tile's LAST K-step, handed to the other waves through the first word
        // of the K-slice buffer that this step is about to refill (free since the barrier that ended the previous step; the SIDE
        // instantiation has no other byte of its 160 KiB to spare): write → barrier → read → barrier → the prefetch may overwrite it
        if (t == nk - 1) {
          int* sh_next = reinterpret_cast<int*>(An);
          if (threadIdx.x == 0) {
            const int G = gridDim.x, x = blockIdx.x & 7, q = G >> 3, r = G & 7;
            const int qx = q + (x < r ? 1 : 0), base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
            *sh_next = base + ticket % qx + (ticket / qx + 1) * G;       // the XCD's share in the order the striding walks it
          }
          __syncthreads();
          next_tile = *sh_next;
          __syncthreads();
          has_next = next_tile < nwg;
          tile_coords(has_next ? next_tile : 0, a.tiles_m, a.tiles_n, a.group_m, tm_, tn_);
          nm0 = tm_ * BM;
          nn0 = tn_ * BN;
        }
      }
      // source of the NEXT K-slice: this tile's slice t+1, or the first slice of the workgroup's next tile
      const bool more_k = t + 1 < nk;
      const bool pf = !NT_DBG(a, 1) && (more_k || has_next);
      const int koff = more_k ? (t + 1) * TKB : 0;
      const int pf_rows_a = more_k ? rows_a : min(BM, a.M - nm0), pf_rows_b = more_k ? rows_b : min(BN, a.N - nn0);
      const char* pf_a = more_k ? Abase + koff : a.A + (int64_t)nm0 * lda_b;
      const char* pf_b = more_k ? Bbase + koff : a.B + (int64_t)nn0 * ldb_b;
      constexpr int PA = (BM / 8 + NW - 1) / NW, PB = BN / 8 / NW;    // 1-KiB DMA pieces per wave and K-step (288-row tiles: the last A piece exists for waves 0–3 only)
      const auto rsrc_a = slice_rsrc(pf_a, pf_rows_a, lda_b, kbytes - koff);
      const auto rsrc_b = slice_rsrc(pf_b, pf_rows_b, ldb_b, kbytes - koff);
      // (SIDE: the per-lane offset is rebuilt from an opaque copy of the lane id in every K-step — hoisted out of the K loop its parts
      //  are four more live registers, which the 288-row instantiation spills and reloads, with a vmcnt(0) each, inside the loop)
      uint32_t vb_a = 0, vb_b = 0;
      if constexpr (SIDE) {
        const int lo = fresh_lane();
        vb_a = piece_base(lda_b, kbytes - koff, wave, lo);
        vb_b = piece_base(ldb_b, kbytes - koff, wave, lo);
      }
      auto piece = [&](int j) {
        if (!pf) return;
        if constexpr (SIDE) {
          // one per-lane offset, a descriptor per piece (scalar arithmetic): the eight per-lane offsets the plain kernel keeps
          // in registers across the K-loop do not fit beside the SIDE epilogue's state (spills inside the K-loop); the plain
          // kernel keeps them — the scalar variant measured 2 % slower there (SGPR pressure spills into VGPR lanes)
          if (j < PA) {
            if ((BM / 8) % NW == 0 || wave + j * NW < BM / 8)
              stage_piece_at<NW>(An, slice_rsrc(pf_a + (int64_t)j * NW * 8 * lda_b, pf_rows_a - j * NW * 8, lda_b, kbytes - koff), vb_a, wave, j);
          } else if (j < PA + PB)
            stage_piece_at<NW>(An + A_BYTES, slice_rsrc(pf_b + (int64_t)(j - PA) * NW * 8 * ldb_b, pf_rows_b - (j - PA) * NW * 8, ldb_b, kbytes - koff),
                               vb_b, wave, j - PA);
        } else {
          const bool full = a.glds && kbytes - koff >= TKB && pf_rows_a > 0 && pf_rows_b > 0;
          if (j < PA) { if ((BM / 8) % NW == 0 || wave + j * NW < BM / 8) stage_piece_g<NW>(An, pf_a, rsrc_a, pf_rows_a, lda_b, kbytes - koff, wave, lane, j, full); }
          else if (j < PA + PB) stage_piece_g<NW>(An + A_BYTES, pf_b, rsrc_b, pf_rows_b, ldb_b, kbytes - koff, wave, lane, j - PA, full);
        }
      };
      auto prefetch = [&]() {
#pragma unroll
        for (int j = 0; j < PA + PB; ++j) piece(j);
      };
      // DMA placement: STAGGER — waves w and w + NW/2 share a SIMD; the first half issues its burst before the
      // K-step's fragment reads, the second half after the first MT/4 rows of MFMAs.  (Dealing the pieces out one per
      // row of MFMAs instead measured equal to 3 % slower: the L2→LDS path, not the issue slot, is the limit.)
      // stagger 2 (round 5 A/B): no burst at all — ONE piece after every row of K-half 0's MFMAs (8 / 9 pieces over its 8 / 9 rows): the
      // stamps show ≈ 1 300 cycles of a 3 900-cycle K-step spent issuing a wave's eight pieces back to back (profiles/r05c_gemm_stamps.txt)
      constexpr bool spread = LAKO_NT_SPREAD == 1 ? !SIDE : LAKO_NT_SPREAD == 2;      // a BUILD-time constant (1: plain kernels, 2: all): as a run-time branch the two copies of the MFMA rows
                                                                // make the accumulators meet at a control-flow merge — 244-341 VGPRs of spills
      const bool late = a.stagger == 1 && NW == 8 && wave >= NW / 2;
      if (!late && !spread) prefetch();
      NT_STAMP(1);
      // (288-row tile with a side operand: 256 registers — the lane parts of the fragment addresses are rebuilt per K-step; kept across the
      //  loop one of them is spilled and its reload sits, with a vmcnt(0), between the DMA issue and the fragment reads)
      int r16k = r16, gk = g;
      if constexpr (SIDE && MT > 8) {
        const int lf = fresh_lane();
        r16k = lf & 15;
        gk = lf >> 4;
      }
      // LAKO_NT_ROLL (round 5, A/B build; plain bf16 kernels): the A fragments ROLL through a ring of four register quads — row r + 4 of the
      // K-step's 2·MT MFMA rows is requested as soon as row r's MFMAs have been issued, every row waits for exactly its own fragment
      // (counted lgkmcnt: LDS returns in order) — instead of twelve reads, a wait and 32 MFMAs twice per K-step.  The stamps
      // (profiles/r05c_gemm_stamps.txt) put ≈ 600 cycles of exposed read latency in front of each K-half.  The reads are inline asm (the
      // compiler would fold the ring into one quad and wait per read); nothing else touches LDS inside the K loop.
      constexpr bool ROLL = LAKO_NT_ROLL != 0 && !SIDE && !PP && sizeof(T) == 2 && (NW == 8 || (NW == 4 && MT == 8 && NT == 8));
      if constexpr (ROLL) {
        constexpr int NR = 2 * MT, RD = 4;
        const uint32_t key16 = (uint32_t)((g ^ ((r16 >> 1) & 7)) << 4);
        const uint32_t aA = (uint32_t)(uintptr_t)LDS_PTR(As) + (uint32_t)((wr * MT * 16 + r16) * TKB) + key16;      // K-half 1: ^ 64
        const uint32_t aB = (uint32_t)(uintptr_t)LDS_PTR(Bs) + (uint32_t)((wc * NT * 16 + r16) * TKB) + key16;
        const uint32_t aA1 = aA ^ 64u, aB1 = aB ^ 64u;
        u32x4 bfr[2][NT], ar[RD];
        auto rd = [](u32x4& dst, uint32_t addr, auto OFF) {
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(decltype(OFF)::value) : "memory");
        };
        static_for<NT>([&](auto N_) { rd(bfr[0][decltype(N_)::value], aB, std::integral_constant<int, decltype(N_)::value * 16 * TKB>{}); });
        static_for<NT>([&](auto N_) { rd(bfr[1][decltype(N_)::value], aB1, std::integral_constant<int, decltype(N_)::value * 16 * TKB>{}); });
        static_for<RD>([&](auto R_) {
          constexpr int r = decltype(R_)::value;
          rd(ar[r], r / MT ? aA1 : aA, std::integral_constant<int, (r % MT) * 16 * TKB>{});
        });
        constexpr int Q = LAKO_NT_WAIT_Q > 0 ? LAKO_NT_WAIT_Q : MT / 4;
        static_for<NR>([&](auto R_) {
          constexpr int r = decltype(R_)::value, kh = r / MT, mt = r % MT;
          if constexpr (r == NR - Q) {
            // the wait for the next K-slice and the barrier, Q rows before the end of the step (as in the two-phase loop); every fragment read of
            // this buffer has been issued by now (RD > Q) and must have LANDED before another wave's next K-step may DMA into the buffer
            if (!NT_DBG(a, 4)) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
          } else if constexpr (r < NR - Q) {
            constexpr int newer = (NR - 1 - r) < (RD - 1) ? (NR - 1 - r) : (RD - 1);      // reads issued after row r's
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(newer) : "memory");
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[nt][mt] = Mma<T>::run(bfr[kh][nt], ar[r % RD], acc[nt][mt]);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (r + RD < NR) rd(ar[r % RD], (r + RD) / MT ? aA1 : aA, std::integral_constant<int, ((r + RD) % MT) * 16 * TKB>{});
          if constexpr (kh == 0) {
#pragma unroll
            for (int j = 0; j < PA + PB; ++j)
              if (j * MT / (PA + PB) == mt) piece(j);
          }
          __builtin_amdgcn_sched_barrier(0);
        });
      } else
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        u32x4 af[MT], bf[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[nt] = read_frag_rows(Bs, (wc * NT + nt) * 16 + r16k, kh * 4 + gk);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          // debug bit 5 (timing experiment, wrong results): every second A fragment is a copy of its neighbour — a third fewer LDS
          // reads at unchanged DMA and MFMA work: does the fragment-read traffic slow the DMA's LDS writes?
          if (NT_DBG(a, 32) && (mt & 1)) af[mt] = af[mt - 1];
          else af[mt] = read_frag_rows(As, (wr * MT + mt) * 16 + r16k, kh * 4 + gk);
        }
        // all fragment reads of the K-half go out back to back; left alone, the machine scheduler folds every
        // A fragment into ONE register quad (read → s_waitcnt lgkmcnt(0) → 4 MFMAs, MT times per K-half),
        // exposing a full LDS round trip per 64 MFMA cycles
        __builtin_amdgcn_sched_barrier(0);
        auto mma_rows = [&](int lo, int hi) {
#pragma unroll
          for (int mt = lo; mt < hi; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt][mt] = Mma<T>::run(bf[nt], af[mt], acc[nt][mt]);
        };
        constexpr int Q = LAKO_NT_WAIT_Q > 0 ? LAKO_NT_WAIT_Q : MT / 4;
        if (kh == 0) {
          if constexpr (spread) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            mma_rows(mt, mt + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < PA + PB; ++j)      // the pieces dealt evenly over the MT rows (256-row tiles: 8 over 8; 128-row tiles: 8 over 4)
              if (j * MT / (PA + PB) == mt) piece(j);
            __builtin_amdgcn_sched_barrier(0);
          }
          if (SIDE && t == 0) {   // pass 0 of the side operand behind this step's K-slice pieces (the counted wait below assumes that order)
            __builtin_amdgcn_sched_barrier(0);
            side_issue(0, smem + 2 * BUF + wave * SLOT, m0, n0);
            __builtin_amdgcn_sched_barrier(0);
          }
          } else {
          mma_rows(0, Q);
          if (late) {
            __builtin_amdgcn_sched_barrier(0);
            prefetch();
            __builtin_amdgcn_sched_barrier(0);
          }
          if (SIDE && t == 0) {   // pass 0 of the side operand → the spare 32 KiB, behind this step's K-slice DMA
            __builtin_amdgcn_sched_barrier(0);
            side_issue(0, smem + 2 * BUF + wave * SLOT, m0, n0);
            __builtin_amdgcn_sched_barrier(0);
          }
          mma_rows(Q, MT);
          }
          if (NT_DBG(a, 64)) { asm volatile("" :: "v"(acc[0][MT - 1])); NT_STAMP(2); }
        } else {
          // the wait for the next K-slice and the barrier sit Q rows of MFMAs before the end of the step: late enough
          // that the DMA had the step to land (the scheduler would hoist them to the top of this K-half), early
          // enough that the barrier round trip is covered by MFMAs already queued
          mma_rows(0, MT - Q);
          __builtin_amdgcn_sched_barrier(0);
          if (NT_DBG(a, 64)) { asm volatile("" :: "v"(acc[0][MT - Q - 1])); NT_STAMP(3); }
          if (SIDE && t == 0 && nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SP) : "memory");   // the side pass may land during the next K-step
          else if (!NT_DBG(a, 4)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          NT_STAMP(4);
          __syncthreads();
          NT_STAMP(5);
          __builtin_amdgcn_sched_barrier(0);
          mma_rows(MT - Q, MT);
          if (NT_DBG(a, 64)) { asm volatile("" :: "v"(acc[0][MT - 1])); NT_STAMP(6); }
        }
      }
      cur ^= 1;
    }

  // ---- epilogue -------------------------------------------------------------------------------
  TO* C = reinterpret_cast<TO*>(a.C);
  const TO* R = reinterpret_cast<const TO*>(a.resid);
  const T* X = reinterpret_cast<const T*>(a.aux);
  const bool relu = a.flags & LAKO_EPI_RELU, has_res = a.flags & LAKO_EPI_RESID,
             auxm = a.flags & LAKO_EPI_AUXMASK, atomic = a.flags & LAKO_EPI_ATOMIC;
  const bool drop = a.drop_thresh != 0;
  if constexpr (SIDE) {
    // SIDE epilogue (256² bf16 tile with a residual OR an aux-mask operand).  The generic epilogue fetches the operand in the
    // accumulator layout (32 dependent 8-B loads per lane, 32-B granules) and stores the same way: ≈24 µs per tile next to a
    // 20 µs main loop at K = 768.  Here the operand arrives by LDS-DMA in four 32-row passes, two passes ahead of its use
    // (slot A = the spare 32 KiB, slot B and the transposition scratch = the K-slice buffer this tile no longer reads; all
    // wave-private, no barriers; retired by counted waits — stores and DMA share one in-order counter, so every store below
    // is an always-issued buffer store); the arithmetic is the generic epilogue's, in the accumulator layout (identical
    // results); the finished 16 rows then go through 4 KiB of scratch and leave as row-major 16-B stores.
    // every lane-derived address below is rebuilt from an opaque copy of the lane id: as loop invariants of the persistent tile
    // loop they would be hoisted to the kernel prologue and stay live across the main loop (spills)
    const int le = fresh_lane();
    const int r16e = le & 15, ge = le >> 4;
    char* slot_a = smem + 2 * BUF + wave * SLOT;
    char* fr = wave_scratch(cur ^ 1);
    char* slot_b = fr + 4096;
    constexpr int NP = MT * 16 / PH;    // PH-row passes of the wave's rows
    constexpr int MPP = PH / 16;        // m-tiles per pass
    const int rows_v = min(BM, a.M - m0), cols_v = min(BN, a.N - n0), ldc_b = (int)a.ldc * 2;
    // rows past the edge fall out of the descriptor's range by themselves; columns past it start from an out-of-range base
    const auto crs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(C + (int64_t)m0 * a.ldc + n0), 0,
                                                        (rows_v - 1) * ldc_b + cols_v * 2, 0x00020000);
    const uint32_t cvb = (wc * 64 + (le & 7) * 8 < cols_v && !NT_DBG(a, 8))
                             ? (uint32_t)((wr * (MT * 16) + (le >> 3)) * ldc_b + (wc * 64 + (le & 7) * 8) * 2) : 0x80000000u;
    side_issue(1, slot_b, m0, n0);
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
      char* slot = (pass & 1) ? slot_b : slot_a;
      // in flight behind the side pass needed now (SP DMA pieces per side pass, SP stores per finished pass) — pass 1: side 2 + stores 0;
      // pass p >= 2: stores p−2, side p+1 (if there is one), stores p−1
      if (pass == 1) {
        if (NP > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * SP) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SP) : "memory");
      } else if (pass >= 2) {
        if (pass + 1 < NP) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * SP) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * SP) : "memory");
      }
      bf16x4 sv[MPP][4];
#pragma unroll
      for (int mi = 0; mi < MPP; ++mi)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const int row_l = mi * 16 + r16e;
          sv[mi][nt] = *reinterpret_cast<const bf16x4*>(slot + row_l * 128 + (((nt * 2 + (ge >> 1)) ^ (row_l & 7)) * 16) + (ge & 1) * 8);
        }
      if (pass + 2 < NP) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slot has been read: it may be overwritten
        side_issue(pass + 2, slot, m0, n0);
      }
#pragma unroll
      for (int mi = 0; mi < MPP; ++mi) {
        const int mt = MPP * pass + mi;
        const int m = m0 + (wr * MT + mt) * 16 + r16e;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const int n = n0 + (wc * NT + nt) * 16 + 4 * ge;
          const bf16x4 q = sv[mi][nt];
          const f32x4 x = {(float)q[0], (float)q[1], (float)q[2], (float)q[3]};
          f32x4 v = acc[nt][mt] * a.alpha;
          if (relu) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
          }
          if (auxm) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = x[r] > 0.f ? v[r] * a.aux_scale : 0.f;
          }
          if (drop) {
            const uint64_t idx = (uint64_t)(a.row0 + m) * (uint64_t)a.N + (uint64_t)n;
            bool kp[4];
            lako_keep4(a.drop_key, idx >> 2, a.drop_thresh, kp);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = kp[r] ? v[r] * a.drop_scale : 0.f;
          }
          if (has_res) v += x;
          *reinterpret_cast<f32x4*>(fr + r16e * 256 + (((nt * 4 + ge) ^ r16e) * 16)) = v;
        }
        const int cj = le & 7;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int row_l = it * 8 + (le >> 3);
          const f32x4 lo = *reinterpret_cast<const f32x4*>(fr + row_l * 256 + (((2 * cj) ^ row_l) * 16));
          const f32x4 hi = *reinterpret_cast<const f32x4*>(fr + row_l * 256 + (((2 * cj + 1) ^ row_l) * 16));
          bf16x8 o;
          o[0] = (bf16_t)lo[0]; o[1] = (bf16_t)lo[1]; o[2] = (bf16_t)lo[2]; o[3] = (bf16_t)lo[3];
          o[4] = (bf16_t)hi[0]; o[5] = (bf16_t)hi[1]; o[6] = (bf16_t)hi[2]; o[7] = (bf16_t)hi[3];
#ifdef LAKO_EXPERIMENTS
          store_b128_policy(__builtin_bit_cast(u32x4, o), crs, (int)(cvb + (uint32_t)((mt * 16 + it * 8) * ldc_b)), a.store_aux);
#else
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), crs, (int)(cvb + (uint32_t)((mt * 16 + it * 8) * ldc_b)), 0, 0);
#endif
        }
      }
    }
    if (!has_next) break;
    __syncthreads();   // the next tile's first prefetch DMA-writes the buffer the waves just used as scratch / slot B
    tile = next_tile;
    m0 = nm0;
    n0 = nn0;
    rows_a = min(BM, a.M - m0);
    rows_b = min(BN, a.N - n0);
    continue;
  }
  if constexpr (sizeof(T) == 2 && sizeof(TO) == 2 && (MT == 8 || MT == 6 || MT == 9) && NT == 4 && WM == 2 && WN == 4) {
    if (a.wide_epi) {
      // WIDE epilogue (256² bf16 tile).  In the accumulator layout a lane owns 4 consecutive columns of 32
      // different sub-tiles: 32 eight-byte stores per lane that land as 32-B granules — store-ISSUE bound, ≈9 µs
      // per tile, i.e. 5 K-steps' worth when K = 768.  Instead each wave transposes its 128×64 fp32 tile through
      // its private 8 KiB of the free LDS buffer, 32 rows at a time (XOR-swizzled 16-B chunks), and leaves with
      // row-major data: 8 consecutive columns per lane, 8 rows × 128 B per wave-instruction, 16-B stores (16 store
      // instructions per wave instead of 32).  Only for epilogues WITHOUT a residual / aux operand (host-selected).
      // (round 4) the tile goes through the scratch as BF16: alpha, ReLU and dropout happen in the accumulator layout (a lane's four consecutive
      // columns are one dropout quad — the generic epilogue's arithmetic), the four values are rounded and leave as ONE ds_write_b64 (6 cycles
      // per wave-instruction against 13 for the 16-byte fp32 form, whose ≈ 79 B/clk made the transposition of a 288 × 256 tile 1.5 µs of LDS
      // time); a finished row is then one ds_read_b128 per lane.  16-byte slot C of row r sits at C ^ (r & 7); a pass (32 rows × 128 B) is 4 KiB,
      // so the wave's 8 KiB hold two passes: the next pass's writes do not wait for this pass's reads.
      char* ep = wave_scratch(cur ^ 1);
      const int cj = lane & 7, n = n0 + wc * 64 + cj * 8;
#pragma unroll
      for (int pass = 0; pass < (MT + 1) / 2; ++pass) {
        char* buf = ep + (pass & 1) * 4096;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int mt = 2 * pass + mi, row_l = mi * 16 + r16;
          if (mt >= MT) continue;            // odd MT (288-row tiles): the last pass is 16 rows
          const int m = m0 + (wr * MT + mt) * 16 + r16;
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            f32x4 v = acc[nt][mt] * a.alpha;
            if (relu) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if (drop) {
              const uint64_t idx = (uint64_t)(a.row0 + m) * (uint64_t)a.N + (uint64_t)(n0 + (wc * NT + nt) * 16 + 4 * g);
              bool kp[4];
              lako_keep4(a.drop_key, idx >> 2, a.drop_thresh, kp);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = kp[r] ? v[r] * a.drop_scale : 0.f;
            }
            const bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
            const int c = nt * 4 + g;          // 8-byte chunk of the row
            *reinterpret_cast<bf16x4*>(buf + row_l * 128 + (((c >> 1) ^ (row_l & 7)) * 16) + (c & 1) * 8) = o;
          }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          if (2 * pass + it / 2 >= MT) continue;
          const int row_l = it * 8 + (lane >> 3);
          const u32x4 o = *reinterpret_cast<const u32x4*>(buf + row_l * 128 + ((cj ^ (row_l & 7)) * 16));
          const int m = m0 + wr * (MT * 16) + pass * 32 + row_l;
          if (m < a.M && n < a.N) {
#ifdef LAKO_EXPERIMENTS
            if (a.debug & 16) *reinterpret_cast<u32x4*>(C + (int64_t)(m - m0) * a.ldc + (n - n0)) = o;   // timing experiment: every tile stores to tile (0, 0)
            else if (a.store_aux) {   // streaming cache policy (see store_b128_policy)
              const auto crs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(C + (int64_t)m0 * a.ldc + n0), 0, 0x7fffffff, 0x00020000);
              if (!(a.debug & 8))
                store_b128_policy(o, crs, (int)(((int64_t)(m - m0) * a.ldc + (n - n0)) * 2), a.store_aux);
            } else if (!(a.debug & 8))
#endif
            *reinterpret_cast<u32x4*>(C + (int64_t)m * a.ldc + n) = o;
          }
        }
      }
      if (!has_next) break;
      __syncthreads();   // the next tile's first prefetch DMA-writes the buffer the waves just used as scratch
      tile = next_tile;
      m0 = nm0;
      n0 = nn0;
      rows_a = min(BM, a.M - m0);
      rows_b = min(BN, a.N - n0);
      continue;
    }
  }
  nt_store_tile<T, TO, WM, WN, MT, NT>(a, acc, m0, n0, wr, wc, lane);
    if (!has_next) break;
    tile = next_tile;
    m0 = nm0;
    n0 = nn0;
    rows_a = min(BM, a.M - m0);
    rows_b = min(BN, a.N - n0);
  }
  if constexpr (QUEUE) {
    // the workgroup that finishes last (every ticket of the launch has been drawn by then) leaves the counters zero for the next
    // launch on this stream
    if (threadIdx.x == 0 && atomicAdd(&a.queue[8], 1) == (int)gridDim.x - 1) {
#pragma unroll
      for (int i = 0; i < 9; ++i) atomicExch(&a.queue[i], 0);
    }
  }
}

#include "det.h"
#include "gemm_nt4.h"

// ---------------------------------------------------------------------------------------------
// NT, VERY SKINNY problems (M <= 256 rows: every GEMM of the decoder at the reader's batch sizes) — bf16.  One K-step of the
// ring kernel below costs a global→LDS round trip however small the tile, and K is walked sequentially: 15 µs for
// [128,768]×[2304,768], 36 µs at K = 3072, ≈150 such launches per training step.  Here a workgroup owns a 64×64 tile and its
// EIGHT WAVES SPLIT K: each wave streams its quarter of K straight from global memory into MFMA fragments (a lane's 16 bytes
// are 8 consecutive k of one row — exactly the fragment layout; no LDS, no barriers, three K-steps in flight), accumulates
// a full 64×64 partial tile, and the partials meet in LDS (swizzled 16-B chunks); each wave then finishes 16 rows in
// row-major order: alpha, ReLU, aux mask, dropout, residual, 16-byte stores.  Needs K % 32 == 0, K <= 4096 and 16-byte rows (host).
// ---------------------------------------------------------------------------------------------
// NORM (round 4, MAXS <= NS: every K-step of the wave resident at once): the A operand is the T5-RMSNorm of the rows of a.A, formed in the
// fragments — the waves' partial sums of squares meet in LDS (fixed order), a_frag = bf16(w · (x · rstd)) exactly as rmsnorm_fwd writes it;
// the workgroups of tile column 0 also write the normalised rows and rstd (the backward's operands).  Saves the decoder one launch per
// projection that follows a norm (3 per layer: each ≈ 4.5 µs of kernel and ≈ 15 µs of the dependent chain).
template <typename TO, int NW, int MAXS, int TB, bool NORM = false>   // NW waves split K; at most MAXS K-steps of 32 per wave (fully unrolled); tile = (16·TB)²
__global__ __launch_bounds__(NW * 64) void gemm_nt_skinny_kernel(NtArgs a) {
  constexpr int NS = 4;   // register stages: NS - 1 K-steps in flight per wave
  constexpr int TS = TB * 16, CPR = TS / 4, PART = TS * TS * 4;   // tile side, 16-B chunks per fp32 row, bytes of a partial tile
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [NW waves][TS rows][TS fp32]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.y * TS, n0 = blockIdx.x * TS;
  const int nks = a.K / 32, per = (nks + NW - 1) / NW;
  const int ks0 = wave * per, ks1 = min(nks, ks0 + per);
  // Branch-free K loop: every load is a bounds-checked buffer load whose offset is pushed out of range once the wave's K
  // range is exhausted (zeros: the MFMAs of a padding step add nothing), so the trip count is uniform, the loop body is
  // straight-line code and the compiler's load counter stays exact (conditional loads made it wait for the newest request).
  const auto ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.A), 0, (int)((int64_t)(a.M - 1) * a.lda * 2 + a.K * 2), 0x00020000);
  const auto rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.B), 0, (int)((int64_t)(a.N - 1) * a.ldb * 2 + a.K * 2), 0x00020000);
  uint32_t ao[TB], bo[TB];
#pragma unroll
  for (int i = 0; i < TB; ++i) {   // rows past the edge are clamped: their products are never stored
    ao[i] = (uint32_t)((min(m0 + i * 16 + r16, a.M - 1) * a.lda + g * 8) * 2);
    bo[i] = (uint32_t)((min(n0 + i * 16 + r16, a.N - 1) * a.ldb + g * 8) * 2);
  }
  f32x4 acc[TB][TB];   // [nt][mt], element r of lane (r16, g): C[mt*16 + r16][nt*16 + 4g + r]
#pragma unroll
  for (int i = 0; i < TB; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 fa[NS][TB], fb[NS][TB];
  auto load = [&](int st, int ks) {
    const uint32_t ko = ks < ks1 ? (uint32_t)ks * 64u : 0x80000000u;
#pragma unroll
    for (int i = 0; i < TB; ++i) {
      fa[st][i] = __builtin_amdgcn_raw_buffer_load_b128(ra, (int)(ao[i] + ko), 0, 0);
      fb[st][i] = __builtin_amdgcn_raw_buffer_load_b128(rb, (int)(bo[i] + ko), 0, 0);
    }
  };
  // fully unrolled over the MAXS K-steps a wave can have (steps past its range load zeros): straight-line code, so the
  // compiler's load counter is exact — as a rolled loop with requests in flight across the back edge it drained the queue
  // (s_waitcnt vmcnt(0)) at every loop header
#pragma unroll
  for (int u = 0; u < NS - 1; ++u) load(u, ks0 + u);
  if constexpr (NORM) {
    static_assert(MAXS <= NS, "NORM: all K-steps of a wave in registers");
    if (MAXS == NS) load(NS - 1, ks0 + NS - 1);
    // norm weights of the wave's K-steps (8 consecutive k per lane and step), requested with the operands
    f32x4 wv[MAXS][2];
#pragma unroll
    for (int st = 0; st < MAXS; ++st) {
      const int k = min((ks0 + st) * 32 + g * 8, a.K - 8);
      wv[st][0] = *reinterpret_cast<const f32x4*>(a.norm_w + k);
      wv[st][1] = *reinterpret_cast<const f32x4*>(a.norm_w + k + 4);
    }
    float* ssl = reinterpret_cast<float*>(smem + NW * PART);      // [NW][TS] partial sums of squares
    float ss[TB];
#pragma unroll
    for (int i = 0; i < TB; ++i) {
      ss[i] = 0.f;
#pragma unroll
      for (int st = 0; st < MAXS; ++st) {       // (steps past the wave's range hold zeros)
        const bf16x8 x = __builtin_bit_cast(bf16x8, fa[st][i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) ss[i] += (float)x[e] * (float)x[e];
      }
      ss[i] += __shfl_xor(ss[i], 16, 64);
      ss[i] += __shfl_xor(ss[i], 32, 64);
      if (g == 0) ssl[wave * TS + i * 16 + r16] = ss[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TB; ++i) {
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) tot += ssl[w * TS + i * 16 + r16];
      const float rs = rsqrtf(tot / (float)a.K + a.norm_eps);
      const int m = m0 + i * 16 + r16;
      if (blockIdx.x == 0 && wave == 0 && g == 0 && m < a.M && a.norm_rs) a.norm_rs[m] = rs;
#pragma unroll
      for (int st = 0; st < MAXS; ++st) {
        const bf16x8 x = __builtin_bit_cast(bf16x8, fa[st][i]);
        bf16x8 y;
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = (bf16_t)(wv[st][e >> 2][e & 3] * ((float)x[e] * rs));
        fa[st][i] = __builtin_bit_cast(u32x4, y);
        if (blockIdx.x == 0 && a.norm_out && m < a.M && ks0 + st < ks1)
          *reinterpret_cast<bf16x8*>(a.norm_out + ((int64_t)m * a.norm_ld + (ks0 + st) * 32 + g * 8) * 2) = y;
      }
    }
  }
#pragma unroll
  for (int st = 0; st < MAXS; ++st) {
    if (!NORM && st + NS - 1 < MAXS) load((st + NS - 1) % NS, ks0 + st + NS - 1);
#pragma unroll
    for (int mt = 0; mt < TB; ++mt)
#pragma unroll
      for (int nt = 0; nt < TB; ++nt) acc[nt][mt] = Mma<bf16_t>::run(fb[st % NS][nt], fa[st % NS][mt], acc[nt][mt]);
    __builtin_amdgcn_sched_barrier(0);
  }
  char* mine = smem + wave * PART;
#pragma unroll
  for (int mt = 0; mt < TB; ++mt)
#pragma unroll
    for (int nt = 0; nt < TB; ++nt)
      *reinterpret_cast<f32x4*>(mine + (mt * 16 + r16) * (TS * 4) + (((nt * 4 + g) ^ (r16 & (CPR - 1))) * 16)) = acc[nt][mt];
  __syncthreads();

  TO* C = reinterpret_cast<TO*>(a.C);
  const bf16_t* X = reinterpret_cast<const bf16_t*>(a.aux);
  const TO* R = reinterpret_cast<const TO*>(a.resid);
  const bool relu = a.flags & LAKO_EPI_RELU, has_res = a.flags & LAKO_EPI_RESID, auxm = a.flags & LAKO_EPI_AUXMASK;
  for (int idx = threadIdx.x; idx < TS * TS / 8; idx += NW * 64) {   // 8 consecutive columns of one row per thread
    const int row = idx / (TS / 8), cj = idx % (TS / 8), m = m0 + row, n = n0 + cj * 8;
    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NW; ++w) {   // fixed summation order: deterministic
      lo += *reinterpret_cast<const f32x4*>(smem + w * PART + row * (TS * 4) + (((2 * cj) ^ (row & (CPR - 1))) * 16));
      hi += *reinterpret_cast<const f32x4*>(smem + w * PART + row * (TS * 4) + (((2 * cj + 1) ^ (row & (CPR - 1))) * 16));
    }
    if (m >= a.M || n >= a.N) continue;   // N % 8 == 0: a group of 8 columns is all in or all out
    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[e] *= a.alpha;
      if (relu) v[e] = fmaxf(v[e], 0.f);
    }
    if (auxm) {
      const bf16x8 x = *reinterpret_cast<const bf16x8*>(X + (int64_t)m * a.ldaux + n);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (float)x[e] > 0.f ? v[e] * a.aux_scale : 0.f;
    }
    if (a.drop_thresh != 0) {
      const uint64_t idx = (uint64_t)(a.row0 + m) * (uint64_t)a.N + (uint64_t)n;
      bool kp[2][4];
      lako_keep4(a.drop_key, idx >> 2, a.drop_thresh, kp[0]);
      lako_keep4(a.drop_key, (idx >> 2) + 1, a.drop_thresh, kp[1]);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = kp[e >> 2][e & 3] ? v[e] * a.drop_scale : 0.f;
    }
    if constexpr (sizeof(TO) == 2) {
      if (has_res) {
        const bf16x8 r = *reinterpret_cast<const bf16x8*>(R + (int64_t)m * a.ldr + n);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)r[e];
      }
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
      *reinterpret_cast<bf16x8*>(C + (int64_t)m * a.ldc + n) = o;
    } else {
      if (has_res) {
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(R + (int64_t)m * a.ldr + n);
        const f32x4 r1 = *reinterpret_cast<const f32x4*>(R + (int64_t)m * a.ldr + n + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] += r0[e];
          v[4 + e] += r1[e];
        }
      }
      *reinterpret_cast<f32x4*>(C + (int64_t)m * a.ldc + n) = f32x4{v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(C + (int64_t)m * a.ldc + n + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
  }
}

// ---------------------------------------------------------------------------------------------
// NT, SKINNY problems (the decoder's 128-row GEMMs, the LM head): a handful of 128×128 tiles, so every workgroup
// is alone on its CU and nothing hides the global→LDS latency of the 2-buffer kernel above — ≈2 µs per K-step,
// 16–30 µs per GEMM of ≈0.5 GFLOP, 145 of them per training step.  Here the K-slices run through a 4-slot ring
// (32 KiB per slot): three slices are always in flight, retired by a COUNTED s_waitcnt (every slice is exactly
// RING_P wave-instructions per wave — slices past K are issued too, fully out of bounds, they move no bytes),
// one barrier per K-step.  2×2 waves of 64×64; generic epilogue.
// ---------------------------------------------------------------------------------------------
constexpr int RING_NST = 4, RING_BM = 128, RING_STAGE = 2 * RING_BM * TKB, RING_P = 2 * RING_BM / 8 / 4;
static_assert(RING_P == 8, "the counted waits below assume 8 DMA pieces per wave and slice");

template <typename T, typename TO>
__global__ __launch_bounds__(256) void gemm_nt_ring_kernel(NtArgs a) {
  constexpr int WM = 2, WN = 2, MT = 4, NT = 4, NW = 4, A_BYTES = RING_BM * TKB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave / WN, wc = wave % WN;
  const int64_t lda_b = a.lda * sizeof(T), ldb_b = a.ldb * sizeof(T);
  const int r16 = lane & 15, g = lane >> 4;
  const int ntile = a.tiles_m * a.tiles_n;
  const int tile = blockIdx.x % ntile, split = blockIdx.x / ntile;
  // split-K (atomic fp32 accumulation only): this workgroup owns K bytes [kb0, kb0 + kbytes)
  const int kb0 = split * a.k_chunk;
  const int kbytes = min(a.K * (int)sizeof(T), kb0 + a.k_chunk) - kb0;
  const int nk = (kbytes + TKB - 1) / TKB;
  const int m0 = (tile / a.tiles_n) * RING_BM, n0 = (tile % a.tiles_n) * RING_BM;
  const int rows_a = min(RING_BM, a.M - m0), rows_b = min(RING_BM, a.N - n0);
  const char* Abase = a.A + (int64_t)m0 * lda_b + kb0;
  const char* Bbase = a.B + (int64_t)n0 * ldb_b + kb0;
  auto issue = [&](int t) {     // slice t → slot t % RING_NST (all out of bounds once t >= nk)
    char* slot = smem + (t % RING_NST) * RING_STAGE;
    const int koff = t * TKB;
    stage_rows<RING_BM, NW>(slot, Abase + koff, rows_a, lda_b, kbytes - koff, wave, lane);
    stage_rows<RING_BM, NW>(slot + A_BYTES, Bbase + koff, rows_b, ldb_b, kbytes - koff, wave, lane);
  };
#pragma unroll
  for (int t = 0; t < RING_NST - 1; ++t) issue(t);

  f32x4 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int t = 0; t < nk; ++t) {
    // slices t+1, t+2 may stay in flight: 2 × RING_P wave-instructions; then every wave's pieces of slice t are in
    // LDS and every wave has finished reading slice t-1, whose slot the next issue overwrites
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue(t + RING_NST - 1);
    const char* As = smem + (t % RING_NST) * RING_STAGE;
    const char* Bs = As + A_BYTES;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      u32x4 af[MT], bf[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bf[nt] = read_frag_rows(Bs, (wc * NT + nt) * 16 + r16, kh * 4 + g);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) af[mt] = read_frag_rows(As, (wr * MT + mt) * 16 + r16, kh * 4 + g);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt][mt] = Mma<T>::run(bf[nt], af[mt], acc[nt][mt]);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the (empty) slices issued past K
  nt_store_tile<T, TO, WM, WN, MT, NT>(a, acc, m0, n0, wr, wc, lane);
}

// ---------------------------------------------------------------------------------------------
// TN: C[M,N] += alpha * Aᵀ·B, A [K,M], B [K,N]: the reduction index is the ROW of both operands.
// LDS image per operand: [KR k-rows][128 cols] (bf16: 64 rows × 256 B, fp32: 32 rows × 512 B).
// bf16 fragments come from ds_read_b64_tr_b16 (hardware 4×16 transpose read); fp32 fragments are
// single dwords.  bf16 image swizzle: 32-B slot s of row r sits at slot s ^ key(r),
// key(r) = (r & 3) | (((r >> 3) & 1) << 2): the 8 rows one half-wave touches land in 8 slots.
// ---------------------------------------------------------------------------------------------
struct TnArgs {
  const char* A;
  const char* B;
  float* C;
  int M, N, K;
  int64_t lda, ldb, ldc;
  float alpha;
  int tiles_m, tiles_n, split_k, k_chunk;  // k_chunk: rows of K per split (multiple of 64)
  int no_atomics;   // timing experiment (gemm_tn_big = 2): skip the accumulation
  int glds;         // 1: whole K-steps staged by global_load_lds (tuning nt_glds)
  int Mout;         // output rows actually written (<= M; single-problem launches)
  float* slabs;     // several K-splits WITHOUT atomics: [tile][split][64 Ki floats] partial tiles + tickets (below); nullptr = atomics
  int* tickets;     // [tiles] arrival counters, zeroed before the launch
  int wide_out;     // out_mode 1 / 2 with 16-byte aligned rows (every item: C % 16 == 0, ldc % 4 == 0, N % 4 == 0): 16-byte epilogue accesses
  int no_stagger;   // 1 (default): every wave issues its DMA at the top of the K-step; 0 (LAKO_TN_STAGGER=1): waves 4-7 two MFMA rows later
  int t_full;       // HYBRID schedule (grouped launch, split_k −3): the first t_full tiles (a multiple of 256: whole rounds of the chip) run their
                    // WHOLE K in one workgroup each (plain C += v, no atomics), the remaining tiles are cut into split_k pieces of k_chunk that
                    // add by atomics; 0 = uniform splits
  int out_mode;     // 256² kernel with ONE K-split: 0 fp32 atomics, 1 plain read-modify-write (C += v), 2 overwrite (C = v)
  // grouped launch (gemm_tn256_kernel): n_items > 0 → tile id t belongs to the last item with tile_start <= t
  int n_items;
  struct Item {
    const char *A, *B;
    float* C;
    int M, N, tiles_n, tile_start;
    int64_t lda, ldb, ldc;
    float alpha;
    int Mout;     // output rows actually written (lako_gemm_tn_item_t.rows_out)
  } items[LAKO_TN_GROUP_MAX];
};

template <typename T> struct TnGeom;
template <> struct TnGeom<bf16_t> { static constexpr int KR = 64, ROWB = 256, ROWS_PER_INST = 4; };
template <> struct TnGeom<float> { static constexpr int KR = 32, ROWB = 512, ROWS_PER_INST = 2; };

__device__ __forceinline__ int tn_key(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

template <typename T>
__device__ __forceinline__ void stage_cols(char* lds_tile, const char* base, int krows_valid, int64_t ld_bytes,
                                           int colbytes_valid, int wave, int lane) {
  using G = TnGeom<T>;
  uint32_t nrec = (krows_valid > 0 && colbytes_valid > 0)
                      ? (uint32_t)((int64_t)(krows_valid - 1) * ld_bytes + min(colbytes_valid, G::ROWB))
                      : 0u;
  const auto rsrc = lds_dma_rsrc(base, nrec);   // asm LDS-DMA: see common.h (the builtin makes hipcc wait vmcnt(0) before the tr reads)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int inst = wave * 4 + i;
    int row, cb;
    if constexpr (sizeof(T) == 2) {
      row = inst * 4 + (lane >> 4);
      int cp = lane & 15;
      int c = cp ^ (tn_key(row) << 1);
      cb = c * 16;
    } else {
      row = inst * 2 + (lane >> 5);
      cb = (lane & 31) * 16;
    }
    bool ok = (row < krows_valid) && (cb < colbytes_valid);
    uint32_t voff = ok ? (uint32_t)(row * ld_bytes + cb) : 0xFFFFFFF0u;
    lds_dma16(lds_tile + inst * 1024, rsrc, voff);
  }
}

// bf16: 8 k-values (k = kk*32 + 8g + e) of column c0 + (lane & 15)
__device__ __forceinline__ u32x4 read_frag_tr(const char* lds_tile, int kk, int c0, int lane) {
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int s = c0 >> 4;
  u32x4 out;
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    int row = kk * 32 + 8 * g + 4 * blk + q;
    const char* addr = lds_tile + row * 256 + ((s ^ tn_key(row)) * 32) + p * 8;
    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(addr));
    u32x2 u = __builtin_bit_cast(u32x2, t);
    out[2 * blk] = u[0];
    out[2 * blk + 1] = u[1];
  }
  return out;
}

template <typename T>
__global__ __launch_bounds__(256) void gemm_tn_kernel(TnArgs a) {
  using G = TnGeom<T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int ntile = a.tiles_m * a.tiles_n;
  const int split = blockIdx.x / ntile;
  const int tid = blockIdx.x % ntile;
  const int tile_m = tid / a.tiles_n, tile_n = tid % a.tiles_n;
  const int m0 = tile_m * TM, n0 = tile_n * TN_;
  const int k_begin = split * a.k_chunk;
  const int k_end = min(a.K, k_begin + a.k_chunk);
  if (k_begin >= k_end) return;
  const int64_t lda_b = a.lda * sizeof(T), ldb_b = a.ldb * sizeof(T);
  const int acols_b = (a.M - m0) * (int)sizeof(T), bcols_b = (a.N - n0) * (int)sizeof(T);
  const char* Abase = a.A + (int64_t)k_begin * lda_b + (int64_t)m0 * sizeof(T);
  const char* Bbase = a.B + (int64_t)k_begin * ldb_b + (int64_t)n0 * sizeof(T);
  const int nk = (k_end - k_begin + G::KR - 1) / G::KR;

  f32x4 acc[4][4];  // [mt][nt]; element r of lane (l&15, g): C[m = mt*16 + 4g + r][n = nt*16 + (l&15)]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  stage_cols<T>(smem, Abase, k_end - k_begin, lda_b, acols_b, wave, lane);
  stage_cols<T>(smem + TILE_BYTES, Bbase, k_end - k_begin, ldb_b, bcols_b, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int r16 = lane & 15, g = lane >> 4;
  for (int t = 0; t < nk; ++t) {
    const int cur = t & 1;
    const char* As = smem + cur * 2 * TILE_BYTES;
    const char* Bs = As + TILE_BYTES;
    if (t + 1 < nk) {
      char* An = smem + (cur ^ 1) * 2 * TILE_BYTES;
      int kr = (t + 1) * G::KR;
      stage_cols<T>(An, Abase + (int64_t)kr * lda_b, k_end - k_begin - kr, lda_b, acols_b, wave, lane);
      stage_cols<T>(An + TILE_BYTES, Bbase + (int64_t)kr * ldb_b, k_end - k_begin - kr, ldb_b, bcols_b, wave, lane);
    }
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        u32x4 af[4], bf[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) af[mt] = read_frag_tr(As, kk, wr * 64 + mt * 16, lane);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bf[nt] = read_frag_tr(Bs, kk, wc * 64 + nt * 16, lane);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = Mma<bf16_t>::run(af[mt], bf[nt], acc[mt][nt]);
      }
    } else {
#pragma unroll
      for (int kq = 0; kq < 8; ++kq) {  // 8 × (k = 4) per 32-row slice
        float af[4], bf[4];
        const int row = kq * 4 + g;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          af[mt] = *reinterpret_cast<const float*>(As + row * 512 + (wr * 64 + mt * 16 + r16) * 4);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          bf[nt] = *reinterpret_cast<const float*>(Bs + row * 512 + (wc * 64 + nt * 16 + r16) * 4);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt], bf[nt], acc[mt][nt], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wr * 64 + mt * 16 + 4 * g + r;
      if (m >= a.Mout) continue;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int n = n0 + wc * 64 + nt * 16 + r16;
        if (n < a.N) atomicAdd(a.C + (int64_t)m * a.ldc + n, acc[mt][nt][r] * a.alpha);
      }
    }
}

// ---------------------------------------------------------------------------------------------
// TN, bf16, 256×256 tile / 8 waves (2 along M × 4 along N, 128×64 per wave): half the operand bytes
// per FLOP of the 128² kernel — the weight-gradient GEMMs (K = all tokens, tiny M×N) are bound by the
// global→LDS ingest rate, not by the MFMAs.  Split-K workgroups are placed XCD-locally (xcd_remap over
// split-major ids) so the tiles that share a K-range share one L2.  The epilogue transposes each
// 16×64 accumulator slab through LDS so that every atomic wave-instruction adds 256 contiguous bytes
// of one C row (the full-rate float-atomic shape).
// ---------------------------------------------------------------------------------------------
constexpr int TN2_ROWB = 512, TN2_IMG = 64 * TN2_ROWB;  // 64 k-rows × 256 columns of bf16 = 32 KiB

// GLDS: every k-row of the step exists (the kernel stages a K range's incomplete step by the buffer form, see there; every split is a
// multiple of 64 rows): global_load_lds, no descriptor (see lds_dma16_g) — k-rows past the end of K would have to read as zeros, which
// only the buffer form's range check provides.  Columns past the edge are clamped to the last valid 16 bytes (they feed outputs that
// are never added).
template <bool GLDS>
__device__ __forceinline__ void stage_cols256(char* img, const char* base, int krows_valid, int64_t ld_bytes,
                                              int colbytes_valid, int wave, int lane) {
  if constexpr (GLDS) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int inst = wave + i * 8;
      const int row = inst * 2 + (lane >> 5);
      const int cb = min(((lane & 31) ^ (tn_key(row) << 1)) * 16, colbytes_valid - 16);
      lds_dma16_g(img + inst * 1024, base, (uint32_t)(row * ld_bytes + cb));
    }
  } else {
    uint32_t nrec = (krows_valid > 0 && colbytes_valid > 0)
                        ? (uint32_t)((int64_t)(krows_valid - 1) * ld_bytes + min(colbytes_valid, TN2_ROWB))
                        : 0u;
    const auto rsrc = lds_dma_rsrc(base, nrec);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int inst = wave + i * 8;            // 32 wave-instructions of 2 rows each
      const int row = inst * 2 + (lane >> 5);
      const int cp = lane & 31;
      const int cb = (cp ^ (tn_key(row) << 1)) * 16;
      const bool ok = (row < krows_valid) && (cb < colbytes_valid);
      const uint32_t voff = ok ? (uint32_t)(row * ld_bytes + cb) : 0xFFFFFFF0u;
      lds_dma16(img + inst * 1024, rsrc, voff);
    }
  }
}

// one of the four pieces a wave stages of an operand's K-step (global_load_lds form: whole steps only, columns clamped)
__device__ __forceinline__ void stage_cols256_piece(char* img, const char* base, int64_t ld_bytes, int colbytes_valid, int wave, int lane, int i) {
  const int inst = wave + i * 8;
  const int row = inst * 2 + (lane >> 5);
  const int cb = min(((lane & 31) ^ (tn_key(row) << 1)) * 16, colbytes_valid - 16);
  lds_dma16_g(img + inst * 1024, base, (uint32_t)(row * ld_bytes + cb));
}

__device__ __forceinline__ u32x4 read_frag_tr256(const char* img, int kk, int c0, int lane) {
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int s = c0 >> 4;
  u32x4 out;
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    const int row = kk * 32 + 8 * g + 4 * blk + q;
    const char* addr = img + row * TN2_ROWB + ((s ^ tn_key(row)) * 32) + p * 8;
    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(addr));
    u32x2 u = __builtin_bit_cast(u32x2, t);
    out[2 * blk] = u[0];
    out[2 * blk + 1] = u[1];
  }
  return out;
}

template <bool GLDS>
__global__ __launch_bounds__(512) void gemm_tn256_kernel(TnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int ntile = a.tiles_m * a.tiles_n;               // grouped: the tiles of all items
  int split, tid, k_begin, k_end, unit_mode = a.out_mode;
  if (a.t_full > 0) {
    // HYBRID: workgroup b belongs to XCD x = b mod 8 and is its i-th; every XCD first walks its eighth of the full-K tiles (consecutive
    // tile ids: neighbours share operand panels in its L2), then its share of the short pieces — long units first, short ones fill the end
    const int x = blockIdx.x & 7, i = blockIdx.x >> 3, fpx = a.t_full >> 3;
    if (i < fpx) {
      tid = x * fpx + i;
      split = 0;
      k_begin = 0;
      k_end = a.K;
      unit_mode = 1;
    } else {
      const int j = (i - fpx) * 8 + x;
      if (j >= (ntile - a.t_full) * a.split_k) return;
      tid = a.t_full + j / a.split_k;
      split = j % a.split_k;
      k_begin = split * a.k_chunk;
      k_end = min(a.K, k_begin + a.k_chunk);
      unit_mode = 0;
    }
  } else {
    const int flat = xcd_remap(blockIdx.x, gridDim.x);   // one XCD ← consecutive (split, tile) ids
    split = flat / ntile;
    tid = flat % ntile;
    k_begin = split * a.k_chunk;
    k_end = min(a.K, k_begin + a.k_chunk);
  }
  const int gtile = tid;      // tile id over all problems of the launch (slab / ticket index)
  // the problem this tile belongs to (one problem, or up to LAKO_TN_GROUP_MAX weight gradients sharing K = tokens: with
  // more tiles per launch fewer K-splits fill the chip, and every split costs one fp32 atomic pass over the output)
  TnArgs::Item it;
  if (a.n_items > 0) {
    int i = 0;
#pragma unroll
    for (int j = 1; j < LAKO_TN_GROUP_MAX; ++j)
      if (j < a.n_items && a.items[j].tile_start <= tid) i = j;
    it = a.items[i];
    tid -= it.tile_start;
  } else {
    it.A = a.A; it.B = a.B; it.C = a.C; it.M = a.M; it.N = a.N; it.tiles_n = a.tiles_n;
    it.lda = a.lda; it.ldb = a.ldb; it.ldc = a.ldc; it.alpha = a.alpha; it.Mout = a.Mout;
  }
  const int tile_m = tid / it.tiles_n, tile_n = tid % it.tiles_n;
  const int m0 = tile_m * 256, n0 = tile_n * 256;
  if (k_begin >= k_end) return;
  const int64_t lda_b = it.lda * 2, ldb_b = it.ldb * 2;
  const int acols_b = (it.M - m0) * 2, bcols_b = (it.N - n0) * 2;
  const char* Abase = it.A + (int64_t)k_begin * lda_b + (int64_t)m0 * 2;
  const char* Bbase = it.B + (int64_t)k_begin * ldb_b + (int64_t)n0 * 2;
  const int krows = k_end - k_begin;
  const int nk = (krows + 63) / 64;

  f32x4 acc[8][4];  // [mt][nt]; element r of lane (l&15, g): C[m = mt*16 + 4g + r][n = nt*16 + (l&15)]
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // GLDS: the K range's incomplete last step (krows % 64 k-rows: they must be zero-filled, which only the buffer form does) is
  // processed FIRST, staged here by the buffer form; every step staged inside the loop is then whole and goes by global_load_lds —
  // both forms inside the loop cost 40 registers (spills).  The sum over k does not care about the order of the steps.
  const int tail = GLDS ? (krows & 63) : 0;
  if (tail) {
    stage_cols256<false>(smem, Abase + (int64_t)(krows - tail) * lda_b, tail, lda_b, acols_b, wave, lane);
    stage_cols256<false>(smem + TN2_IMG, Bbase + (int64_t)(krows - tail) * ldb_b, tail, ldb_b, bcols_b, wave, lane);
  } else {
    stage_cols256<GLDS>(smem, Abase, krows, lda_b, acols_b, wave, lane);
    stage_cols256<GLDS>(smem + TN2_IMG, Bbase, krows, ldb_b, bcols_b, wave, lane);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int t = 0; t < nk; ++t) {
    const int cur = t & 1;
    const char* As = smem + cur * 2 * TN2_IMG;
    const char* Bs = As + TN2_IMG;
    auto prefetch = [&]() {
      if (t + 1 < nk) {
        char* An = smem + (cur ^ 1) * 2 * TN2_IMG;
        const int kr = (t + 1 - (tail ? 1 : 0)) * 64;      // (with a tail step in front, loop step t + 1 is whole step t)
        stage_cols256<GLDS>(An, Abase + (int64_t)kr * lda_b, krows - kr, lda_b, acols_b, wave, lane);
        stage_cols256<GLDS>(An + TN2_IMG, Bbase + (int64_t)kr * ldb_b, krows - kr, ldb_b, bcols_b, wave, lane);
      }
    };
    // K-step schedule as in gemm_nt_kernel: fragment reads of a K-half back to back (sched_barrier: the machine
    // scheduler would fold them into one register quad and wait per read), the second half of the workgroup issues
    // its DMA after two rows of MFMAs, and the wait + barrier sit two rows before the end of the step
    // LAKO_TN_SPREAD (round 5, as LAKO_NT_SPREAD): the eight pieces go out one per row of K-half 0's MFMAs instead of in a burst
    constexpr bool spread = LAKO_TN_SPREAD != 0 && GLDS;
    auto piece = [&](int j) {
      if (t + 1 >= nk) return;
      char* An = smem + (cur ^ 1) * 2 * TN2_IMG;
      const int kr = (t + 1 - (tail ? 1 : 0)) * 64;
      if (j < 4) stage_cols256_piece(An, Abase + (int64_t)kr * lda_b, lda_b, acols_b, wave, lane, j);
      else stage_cols256_piece(An + TN2_IMG, Bbase + (int64_t)kr * ldb_b, ldb_b, bcols_b, wave, lane, j - 4);
    };
    const bool late = wave >= 4 && !a.no_stagger;
    if (!late && !spread) prefetch();
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      u32x4 af[8], bf[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) bf[nt] = read_frag_tr256(Bs, kk, wc * 64 + nt * 16, lane);
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) af[mt] = read_frag_tr256(As, kk, wr * 128 + mt * 16, lane);
      __builtin_amdgcn_sched_barrier(0);
      auto mma_rows = [&](int lo, int hi) {
#pragma unroll
        for (int mt = lo; mt < hi; ++mt)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = Mma<bf16_t>::run(af[mt], bf[nt], acc[mt][nt]);
      };
      if (kk == 0) {
        if constexpr (spread) {
#pragma unroll
          for (int mt = 0; mt < 8; ++mt) {
            mma_rows(mt, mt + 1);
            __builtin_amdgcn_sched_barrier(0);
            piece(mt);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
        mma_rows(0, 2);
        if (late) {
          __builtin_amdgcn_sched_barrier(0);
          prefetch();
          __builtin_amdgcn_sched_barrier(0);
        }
        mma_rows(2, 8);
        }
      } else {
        mma_rows(0, LAKO_TN_WAIT_ROWS);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        mma_rows(LAKO_TN_WAIT_ROWS, 8);
      }
    }
  }

  // SLAB REDUCTION of the K-splits (round 4; cdna_hip_programming.md "in-launch split-K reduction", counter form): every split writes its
  // fp32 partial tile with plain 16-byte stores (accumulator order: a wave-instruction writes 1 KiB), drains them, ONE lane issues an
  // agent-scope release fence and draws a ticket; the workgroup that draws the last ticket acquires, sums ALL splits' slabs in split
  // order (its own included: the sum does not depend on who arrived last — the gradients are bit-reproducible, which float atomics
  // were not) and adds the tile to C by plain loads / stores.  Replaces two atomic passes over every tile (≈ 1.3 TB/s chip-wide).
  if (a.slabs != nullptr && a.split_k > 1 && a.t_full == 0) {      // (never with the hybrid schedule: its full-K units draw no ticket)
    float* mine = a.slabs + ((int64_t)gtile * a.split_k + split) * 65536 + (int64_t)wave * 8192;
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) *reinterpret_cast<f32x4*>(mine + ((mt * 4 + nt) * 64 + lane) * 4) = acc[mt][nt];
      if (mt & 1) __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* flag = reinterpret_cast<int*>(smem);          // (the staging buffers are free after the K loop's last barrier)
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      *flag = __hip_atomic_fetch_add(a.tickets + gtile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const int ticket = *flag;
    __syncthreads();                                    // (smem is reused as the transposition scratch below)
    if (ticket != a.split_k - 1) return;
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      a.tickets[gtile] = 0;                             // (the host zeroes them before every launch as well)
    }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int sp = 0; sp < a.split_k; ++sp) {
      const float* part = a.slabs + ((int64_t)gtile * a.split_k + sp) * 65536 + (int64_t)wave * 8192;
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) {      // two rows of tiles' loads in flight at a time: all 32 at once cost 40 registers (spills)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] += *reinterpret_cast<const f32x4*>(part + ((mt * 4 + nt) * 64 + lane) * 4);
        if (mt & 1) __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const int out_mode = (a.slabs != nullptr && a.split_k > 1 && a.t_full == 0) ? 1 : unit_mode;
  // epilogue: per wave a private 16×64 fp32 slab in LDS (staging buffers are free after the last barrier)
  float* slab = reinterpret_cast<float*>(smem) + wave * 1024;
  const int r16 = lane & 15, g = lane >> 4;
  const int n = n0 + wc * 64 + lane;
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) slab[(4 * g + r) * 64 + nt * 16 + r16] = acc[mt][nt][r] * it.alpha;
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes have landed
    __builtin_amdgcn_wave_barrier();
    if (out_mode != 0 && a.wide_out) {
      // plain adds / stores (one contributor): FOUR consecutive columns per lane — 4 rows × 256 B per wave-instruction, 16-byte accesses:
      // a quarter of the memory instructions of the dword form (the decoder's K = 128 products and dE are all epilogue:
      // 1 024 loads + 1 024 stores per tile and CU in the dword form — round 4: 318 + 167 + 236 + 154 µs per step for ≈ 30 GFLOP)
      const int c4 = (lane & 15) * 4, nq = n0 + wc * 64 + c4;
      f32x4 v4[4], old[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = (lane >> 4) + 4 * j, m = m0 + wr * 128 + mt * 16 + row;
        v4[j] = *reinterpret_cast<const f32x4*>(slab + row * 64 + c4);
        if (out_mode == 1 && m < it.Mout && nq < it.N) old[j] = *reinterpret_cast<const f32x4*>(it.C + (int64_t)m * it.ldc + nq);
        else old[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = (lane >> 4) + 4 * j, m = m0 + wr * 128 + mt * 16 + row;
        if (m < it.Mout && nq < it.N) *reinterpret_cast<f32x4*>(it.C + (int64_t)m * it.ldc + nq) = v4[j] + old[j];
      }
      __builtin_amdgcn_wave_barrier();
      continue;
    }
#pragma unroll
    for (int row = 0; row < 16; ++row) {
      const float v = slab[row * 64 + lane];
      const int m = m0 + wr * 128 + mt * 16 + row;
      #ifdef LAKO_EXPERIMENTS
      if (a.no_atomics) continue;
#endif
      if (m < it.Mout && n < it.N) {
        float* cp = it.C + (int64_t)m * it.ldc + n;
        // one K-split and nobody else adding to C (the caller's promise, split_k < 0): 256 contiguous bytes per wave-instruction as
        // plain loads / stores at the chip's copy rate instead of float atomics at ≈1.3 TB/s
        if (out_mode == 2) *cp = v;
        else if (out_mode == 1) *cp += v;
        else atomicAdd(cp, v);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

#include "gemm_tn4.h"

// the four-wave weight-gradient kernel takes whole-step K ranges (every unit's range a multiple of 64 rows except the last) of at least 128
// rows by global_load_lds, without the slab reduction
inline bool tn4_ok(const TnArgs& a, const lako_tuning_t& tu) {
  const int last = a.K - (a.split_k - 1) * a.k_chunk;
  return tu.tn_four && a.glds && a.k_chunk % 64 == 0 && a.slabs == nullptr && a.k_chunk >= 128 && last >= 128;
}

// ---------------------------------------------------------------------------------------------
// NT with MX-scaled fp8 operands (BASELINE config 5: "fp8 MFMA GEMMs"): C = epilogue(alpha · A·Bᵀ), A [M,K] and B [N,K] in
// OCP e4m3 bytes, every 32 consecutive k of a row sharing one E8M0 scale (OCP microscaling: element = q · 2^(s − 127)).
// v_mfma_scale_f32_16x16x128_f8f6f4 applies both operands' block scales inside the matrix core and runs at twice the bf16
// rate per clock; the non-scaled fp8 MFMA runs at the bf16 rate (MI355X_MICROARCH.md § Matrix cores), so this is the only
// fp8 form worth having.  Same 256×256 persistent structure as gemm_nt_kernel<bf16, …, 2, 4, 8, 4>: one 128-BYTE K-slice
// per step is now 128 k — exactly one MFMA deep — so a step moves the same bytes through LDS and does twice the FLOPs.
// Operand layout (measured with exact integer data and per-block scales, tools/probe/mx_probe.hip): lane (row = lane & 15,
// g = lane >> 4) holds, in its 32 operand bytes, k = 16g … 16g + 15 and k = 64 + 16g … 64 + 16g + 15 of its row — the two
// 16-byte LDS chunks g and 4 + g, exactly the two K-halves of the bf16 kernel — while its SCALE byte is the one of block g,
// k = 32g … 32g + 31.  Scales are stored [rows][4 (kb)][KSP] bytes, KSP = ⌈K/128⌉ rounded up to 4:
// the scales of (row, kb) for 4 consecutive K-steps are ONE aligned dword, fetched a group ahead.
// Epilogues: the wide LDS-transposed store (plain / ReLU / dropout) and the accumulator-layout generic one (residual, aux mask).
// ---------------------------------------------------------------------------------------------
typedef int v8i_t __attribute__((ext_vector_type(8)));
struct MxArgs {
  NtArgs nt;
  const uint8_t* sa;   // [M][4][ksp]
  const uint8_t* sb;   // [N][4][ksp]
  int ksp;
};

__global__ __launch_bounds__(512) void gemm_nt_mx_kernel(MxArgs ma) {
  const NtArgs& a = ma.nt;
  constexpr int WM = 2, WN = 4, MT = 8, NT = 4, NW = 8, BM = 256, BN = 256;
  constexpr int A_BYTES = BM * TKB, B_BYTES = BN * TKB, BUF = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave / WN, wc = wave % WN;
  const int nwg = a.tiles_m * a.tiles_n;
  const int64_t lda_b = a.lda, ldb_b = a.ldb;          // bytes = elements
  const int kbytes = a.K;
  const int nk = (kbytes + TKB - 1) / TKB;
  const int r16 = lane & 15, g = lane >> 4;
  const int srow = 4 * ma.ksp;                          // scale bytes per operand row

  int tile = xcd_remap(blockIdx.x, gridDim.x);
  int tm_, tn_;
  tile_coords(tile, a.tiles_m, a.tiles_n, a.group_m, tm_, tn_);
  int m0 = tm_ * BM, n0 = tn_ * BN;
  int rows_a = min(BM, a.M - m0), rows_b = min(BN, a.N - n0);
  if (a.dephase > 0) {
    const int ph = (blockIdx.x >> 3) % (a.dephase >> 16);
    if (ph) {
      const uint64_t t0 = __builtin_amdgcn_s_memrealtime(), dt = (uint64_t)((a.dephase & 0xffff) * ph);
      while (__builtin_amdgcn_s_memrealtime() - t0 < dt) __builtin_amdgcn_s_sleep(8);
    }
  }
  stage_rows<BM, NW>(smem, a.A + (int64_t)m0 * lda_b, rows_a, lda_b, kbytes, wave, lane);
  stage_rows<BN, NW>(smem + A_BYTES, a.B + (int64_t)n0 * ldb_b, rows_b, ldb_b, kbytes, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;

  while (true) {
    const int next_tile = tile + gridDim.x;
    const bool has_next = next_tile < nwg;
    tile_coords(has_next ? next_tile : 0, a.tiles_m, a.tiles_n, a.group_m, tm_, tn_);
    const int nm0 = tm_ * BM, nn0 = tn_ * BN;
    const char* Abase = a.A + (int64_t)m0 * lda_b;
    const char* Bbase = a.B + (int64_t)n0 * ldb_b;
    // the lane's scale rows as 32-bit byte offsets (clamped at the edge: those operand rows are zero-filled, their scale is irrelevant)
    uint32_t sao[MT], sbo[NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) sao[mt] = (uint32_t)(min(m0 + (wr * MT + mt) * 16 + r16, a.M - 1) * srow + g * ma.ksp);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) sbo[nt] = (uint32_t)(min(n0 + (wc * NT + nt) * 16 + r16, a.N - 1) * srow + g * ma.ksp);
    uint32_t sca[MT], scb[NT], sca_n[MT], scb_n[NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) sca_n[mt] = *reinterpret_cast<const uint32_t*>(ma.sa + sao[mt]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) scb_n[nt] = *reinterpret_cast<const uint32_t*>(ma.sb + sbo[nt]);

    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int t = 0; t < nk; ++t) {
      const char* As = smem + cur * BUF;
      const char* Bs = As + A_BYTES;
      char* An = smem + (cur ^ 1) * BUF;
      const bool more_k = t + 1 < nk;
      const bool pf = more_k || has_next;
      const int koff = more_k ? (t + 1) * TKB : 0;
      const int pf_rows_a = more_k ? rows_a : min(BM, a.M - nm0), pf_rows_b = more_k ? rows_b : min(BN, a.N - nn0);
      const char* pf_a = more_k ? Abase + koff : a.A + (int64_t)nm0 * lda_b;
      const char* pf_b = more_k ? Bbase + koff : a.B + (int64_t)nn0 * ldb_b;
      constexpr int PA = BM / 8 / NW, PB = BN / 8 / NW;
      const auto rsrc_a = slice_rsrc(pf_a, pf_rows_a, lda_b, kbytes - koff);
      const auto rsrc_b = slice_rsrc(pf_b, pf_rows_b, ldb_b, kbytes - koff);
      auto prefetch = [&]() {
        if (!pf) return;
#pragma unroll
        for (int j = 0; j < PA; ++j) stage_piece<NW>(An, rsrc_a, pf_rows_a, lda_b, kbytes - koff, wave, lane, j);
#pragma unroll
        for (int j = 0; j < PB; ++j) stage_piece<NW>(An + A_BYTES, rsrc_b, pf_rows_b, ldb_b, kbytes - koff, wave, lane, j);
      };
      if ((t & 3) == 0) {       // a new group of 4 K-steps: its scale dwords were requested a group ago; request the next group's
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) sca[mt] = sca_n[mt];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) scb[nt] = scb_n[nt];
        if (t + 4 < nk) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) sca_n[mt] = *reinterpret_cast<const uint32_t*>(ma.sa + sao[mt] + (uint32_t)(t + 4));
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) scb_n[nt] = *reinterpret_cast<const uint32_t*>(ma.sb + sbo[nt] + (uint32_t)(t + 4));
        }
      }
      const int sh = (t & 3) * 8;
      const bool late = wave >= NW / 2;
      if (!late) prefetch();
      v8i_t bfr[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const u32x4 lo = read_frag_rows(Bs, (wc * NT + nt) * 16 + r16, g), hi = read_frag_rows(Bs, (wc * NT + nt) * 16 + r16, 4 + g);
        bfr[nt] = v8i_t{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
      }
      // A fragments two m-tiles at a time (16 registers; all eight at once would be 64 beside the 128 of the accumulators)
#pragma unroll
      for (int qd = 0; qd < MT / 2; ++qd) {
        v8i_t afr[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = (wr * MT + 2 * qd + i) * 16 + r16;
          const u32x4 lo = read_frag_rows(As, row, g), hi = read_frag_rows(As, row, 4 + g);
          afr[i] = v8i_t{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
        }
        if (qd == MT / 2 - 1) {     // the wait for the next K-slice and the barrier sit before the last quarter of the step's MFMAs
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int mt = 2 * qd + i;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[nt][mt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bfr[nt], afr[i], acc[nt][mt], 0, 0, 0, (int)(scb[nt] >> sh), 0,
                                                                           (int)(sca[mt] >> sh));
        }
        if (qd == 0 && late) {
          __builtin_amdgcn_sched_barrier(0);
          prefetch();
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      cur ^= 1;
    }

    // ---- epilogue ---------------------------------------------------------------------------
    bf16_t* C = reinterpret_cast<bf16_t*>(a.C);
    const bool relu = a.flags & LAKO_EPI_RELU;
    const bool drop = a.drop_thresh != 0;
    if (a.wide_epi) {       // as gemm_nt_kernel's WIDE epilogue: 32 rows at a time through the wave's 8 KiB of the free buffer
      char* ep = smem + (cur ^ 1) * BUF + wave * 8192;
      const int cj = lane & 7, n = n0 + wc * 64 + cj * 8;
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int mt = 2 * pass + mi, row_l = mi * 16 + r16;
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            f32x4 v = acc[nt][mt] * a.alpha;
            if (relu) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            *reinterpret_cast<f32x4*>(ep + row_l * 256 + (((nt * 4 + g) ^ (row_l & 15)) * 16)) = v;
          }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int row_l = it * 8 + (lane >> 3);
          const f32x4 lo = *reinterpret_cast<const f32x4*>(ep + row_l * 256 + (((2 * cj) ^ (row_l & 15)) * 16));
          const f32x4 hi = *reinterpret_cast<const f32x4*>(ep + row_l * 256 + (((2 * cj + 1) ^ (row_l & 15)) * 16));
          const int m = m0 + wr * 128 + pass * 32 + row_l;
          if (m < a.M && n < a.N) {
            float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            if (drop) {
              const uint64_t idx = (uint64_t)(a.row0 + m) * (uint64_t)a.N + (uint64_t)n;
              bool kp[2][4];
              lako_keep4(a.drop_key, idx >> 2, a.drop_thresh, kp[0]);
              lako_keep4(a.drop_key, (idx >> 2) + 1, a.drop_thresh, kp[1]);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = kp[e >> 2][e & 3] ? v[e] * a.drop_scale : 0.f;
            }
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
            *reinterpret_cast<bf16x8*>(C + (int64_t)m * a.ldc + n) = o;
          }
        }
      }
      if (!has_next) break;
      __syncthreads();
    } else {
      nt_store_tile<bf16_t, bf16_t, WM, WN, MT, NT>(a, acc, m0, n0, wr, wc, lane);
      if (!has_next) break;
    }
    tile = next_tile;
    m0 = nm0;
    n0 = nn0;
    rows_a = min(BM, a.M - m0);
    rows_b = min(BN, a.N - n0);
  }
}

#include "gemm_nt4_mx.h"

// MX quantisation of a bf16 matrix: per row and per 32 consecutive k one E8M0 scale 2^(e − 127), e − 127 = floor(log2(amax)) − 8
// (8 = the exponent of e4m3's largest normal, 448 = 1.75·2^8: the OCP microscaling rule), elements q = sat_e4m3(x · 2^−(e−127)).
// One lane = 8 consecutive k (a 16-byte load, an 8-byte store); the 4 lanes of a block agree on amax by two shuffles.
__global__ __launch_bounds__(256) void mx_quantize_kernel(const bf16_t* __restrict__ x, int64_t rows, int K, int64_t ld,
                                                          uint8_t* __restrict__ q, uint8_t* __restrict__ sc, int ksp) {
  const int per_row = K >> 3;
  const int64_t total = rows * per_row;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / per_row;
    const int c = (int)(i - row * per_row);                 // 8-element chunk of the row
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + row * ld + c * 8);
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
    uint32_t w[2];
    const uint8_t scale = mx_quant8(f, w);
    *reinterpret_cast<u32x2*>(q + row * K + c * 8) = u32x2{w[0], w[1]};
    if ((c & 3) == 0) {
      const int blk = c >> 2;                               // 32-element block of the row
      sc[row * 4 * ksp + (blk & 3) * ksp + (blk >> 2)] = scale;
    }
  }
}

// ---- kernel-selection knobs -------------------------------------------------------------------------------------------------
// The library keeps NO mutable tuning state: a caller that wants anything but the defaults owns a lako_tuning_t (lako_tuning_init
// fills in the defaults and the LAKO_TUNING environment string, lako_tuning_set changes one field of the CALLER's struct) and hands
// it to the GEMM entry points; NULL means the process defaults, built once (thread-safe static initialisation) and read-only after.
struct TuneKey {
  const char* name;
  int32_t lako_tuning_t::*field;
  bool experiment;     // a timing experiment with wrong results / non-default cache policy: -DLAKO_EXPERIMENTS builds only
};
const TuneKey TUNE_KEYS[] = {
    {"gemm_nt_variant", &lako_tuning_t::nt_variant, false},       {"gemm_nt_tail_split", &lako_tuning_t::nt_tail_split, false},
    {"gemm_nt_ring", &lako_tuning_t::nt_ring, false},             {"gemm_nt_skinny", &lako_tuning_t::nt_skinny, false},
    {"gemm_nt_side_lds", &lako_tuning_t::nt_side_lds, false},     {"gemm_nt_wide_epi", &lako_tuning_t::nt_wide_epi, false},
    {"gemm_nt_group_m", &lako_tuning_t::nt_group_m, false},       {"gemm_nt_persistent", &lako_tuning_t::nt_persistent, false},
    {"gemm_nt_stagger", &lako_tuning_t::nt_stagger, false},       {"gemm_nt_dephase", &lako_tuning_t::nt_dephase, false},
    {"gemm_nt_dephase_n", &lako_tuning_t::nt_dephase_n, false},   {"gemm_tn_big", &lako_tuning_t::tn_big, false},
    {"gemm_tn_split", &lako_tuning_t::tn_split, false},           {"gemm_nt_debug", &lako_tuning_t::nt_debug, true},
    {"gemm_nt_store_aux", &lako_tuning_t::nt_store_aux, true},    {"gemm_nt_tile192", &lako_tuning_t::nt_tile192, false},
    {"gemm_nt_tile288", &lako_tuning_t::nt_tile288, false},
    {"gemm_nt_queue", &lako_tuning_t::nt_queue, false},           {"gemm_nt_pp", &lako_tuning_t::nt_pp, false},
    {"gemm_nt_glds", &lako_tuning_t::nt_glds, false},             {"gemm_nt_four", &lako_tuning_t::nt_four, false},
    {"gemm_tn_four", &lako_tuning_t::tn_four, false},
};

void tuning_defaults(lako_tuning_t* t) {
  memset(t, 0, sizeof(*t));
  t->nt_variant = -1;      // -1 auto; 0: 128x128 / 4 waves; 1: 256x128 / 8 waves; 2: 256x256 / 8 waves; 4: 128x128 ring; 5: split-K skinny; 6: 256x256 / 4 waves (experiment); 7: 192x256 / 8 waves; 8: 288x256 / 8 waves (plain epilogues)
  t->nt_tail_split = 1;
  t->nt_ring = 1;          // skinny problems go to gemm_nt_ring_kernel (0 disables; variant 4 forces it)
  t->nt_skinny = 1;        // M <= 256 rows: gemm_nt_skinny_kernel (0 disables, 2 / 3 / 4 force 64² / 32² / 16² tiles; variant 5 forces the kernel)
  t->nt_side_lds = 1;      // 256² bf16 tiles with ONE side operand (residual or aux mask): LDS-DMA'd operand + row-major stores
  t->nt_wide_epi = 1;
  t->nt_group_m = 8;
  t->nt_persistent = 1;
  t->nt_stagger = 1;
  t->nt_dephase = 100;     // 10-ns ticks (s_memrealtime)
  t->nt_dephase_n = 2;     // phases
  t->tn_big = 1;
  t->tn_split = 0;         // > 0: force the number of K-splits of the 256x256 TN kernel (A/B measurements)
  t->nt_glds = 1;          // K-slice staging of the 256² kernels by global_load_lds (0: buffer_load … lds everywhere)
  t->nt_pp = 0;            // 1: the 8-phase main loop of the 256² / 8-wave bf16 kernel (measured slower, A/B only); 0: the two-phase loop
  t->nt_tile288 = 1;       // 1: 288-row tiles where the round count favours them (launch_nt; plain epilogues)
  t->nt_tile192 = 0;       // 1: 192-row tiles where the round count favours them (launch_nt) — measured no faster, off
  t->nt_four = 1;          // 1: the four-wave kernels with the hand-placed K loop (gemm_nt4.h) wherever they apply
  t->tn_four = 1;          // 1: the same for the 256² weight-gradient kernel (gemm_tn4.h)
}

int tuning_set(lako_tuning_t* t, const char* key, int value) {
  if (!t || !key) {
    lako_set_error("lako_tuning_set: null argument");
    return LAKO_E_BADARG;
  }
  for (const TuneKey& k : TUNE_KEYS) {
    if (strcmp(key, k.name)) continue;
#ifndef LAKO_EXPERIMENTS
    if (k.experiment || (k.field == &lako_tuning_t::tn_big && value == 2)) {
      lako_set_error("lako_tuning_set: '%s' is a timing experiment with wrong results / non-default cache policy; it exists only in the -DLAKO_EXPERIMENTS build", key);
      return LAKO_E_UNSUPPORTED;
    }
#endif
    if (k.field == &lako_tuning_t::nt_dephase_n && value < 2) value = 2;
    t->*(k.field) = value;
    return LAKO_OK;
  }
  lako_set_error("lako_tuning_set: unknown key '%s'", key);
  return LAKO_E_BADARG;
}

// defaults + the LAKO_TUNING environment string "key=value,key=value" (A/B measurements); malformed or refused entries are an error
int tuning_init(lako_tuning_t* t) {
  tuning_defaults(t);
  const char* env = getenv("LAKO_TUNING");
  if (!env) return LAKO_OK;
  char buf[512];
  strncpy(buf, env, sizeof(buf) - 1);
  buf[sizeof(buf) - 1] = 0;
  for (char* tok = strtok(buf, ","); tok; tok = strtok(nullptr, ",")) {
    while (*tok == ' ') ++tok;
    if (!*tok) continue;
    char* eq = strchr(tok, '=');
    if (!eq) {
      lako_set_error("LAKO_TUNING: '%s' is not key=value", tok);
      return LAKO_E_BADARG;
    }
    *eq = 0;
    char* end = eq;
    while (end > tok && end[-1] == ' ') *--end = 0;
    const int rc = tuning_set(t, tok, atoi(eq + 1));
    if (rc != LAKO_OK) return rc;
  }
  return LAKO_OK;
}

const lako_tuning_t& process_tuning() {
  static const lako_tuning_t t = [] {
    lako_tuning_t x;
    if (tuning_init(&x) != LAKO_OK) tuning_defaults(&x);      // a bad LAKO_TUNING is reported by lako_tuning_init to callers that ask
    return x;
  }();
  return t;
}

// ticket counters of the QUEUE instantiation: 9 ints per (device, stream); launches on one stream run one after the other, launches on
// different streams never share counters.  Every QUEUE launch is preceded by its own hipMemsetAsync of the counters on the same stream
// (checked): a launch that faulted or was aborted half-way cannot leave tickets behind for the next one (round 3 relied on each launch's
// last workgroup to reset them).  The first use on a (device, stream) allocates — not possible while the stream is being captured:
// then (or when the allocation or the memset fails) the caller falls back to the static tile order, which computes the same result.
static int* nt_queue_counters(hipStream_t stream) {
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, int*> tab;
  int dev = 0;
  (void)hipGetDevice(&dev);
  int* p = nullptr;
  {
    std::lock_guard<std::mutex> lock(mu);
    int*& slot = tab[{dev, stream}];
    if (!slot) {
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return nullptr;
      }
      if (hipMalloc(reinterpret_cast<void**>(&slot), 16 * sizeof(int)) != hipSuccess) {
        (void)hipGetLastError();
        slot = nullptr;
        return nullptr;
      }
    }
    p = slot;
  }
  if (hipMemsetAsync(p, 0, 16 * sizeof(int), stream) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}

// the 8-phase main loop of the 256² / 8-wave kernel (tuning nt_pp) needs two K-steps per tile
inline int nt_pp_mode(const lako_tuning_t& tu, const NtArgs& a) { return tu.nt_pp && a.K >= 128 ? 1 : 0; }

template <typename T, typename TO, int WM, int WN, int MT, int NT, bool SIDE = false, bool QUEUE = false, bool PP = false>
void launch_nt_cfg(NtArgs a, const lako_tuning_t& tu, hipStream_t s) {
  constexpr int BM = WM * MT * 16, BN = WN * NT * 16, NW_ = WM * WN, BUF_ = (BM + BN) * TKB;
  // + the epilogue scratch of the waves that do not fit the free K-slice buffer (8 KiB each; 192-row tiles: one wave)
  constexpr int EXTRA = (sizeof(T) == 2 && NW_ == 8 && BUF_ / 8192 < NW_) ? (NW_ - BUF_ / 8192) * 8192 : 0;
  constexpr int LDS = 2 * BUF_ + (SIDE ? NW_ * (MT % 2 == 0 ? 32 : 16) * 128 : 0) + EXTRA;
  static_assert(LDS <= 160 * 1024, "LDS");
  LAKO_SET_MAX_LDS((&gemm_nt_kernel<T, TO, WM, WN, MT, NT, SIDE, QUEUE, PP>), LDS);
  a.tiles_m = cdiv(a.M, BM);
  a.tiles_n = cdiv(a.N, BN);
  a.stagger = tu.nt_stagger;
  a.glds = tu.nt_glds && (int64_t)BM * std::max(a.lda, a.ldb) * (int64_t)sizeof(T) < (1ll << 31);   // 32-bit row offsets inside a tile
  a.store_aux = (int64_t)256 * a.ldc * 2 < (1ll << 31) ? tu.nt_store_aux : 0;   // tile-relative 32-bit store offsets
  a.debug = tu.nt_debug;
  // narrow outputs (≤ 7 tile columns) already give an XCD a compact block; a negative knob forces |value| on every shape (tests).
  // Measured at 48 000 rows (profiles/r02g_gemm_group_m_traffic.txt): banding takes the wi projection's L2 misses (FETCH_SIZE) from
  // 217 to 163 MiB-units per launch with no change in time (244 → 241 µs) — the L2-miss traffic is not what bounds the kernel —
  // and the QKV projection from 185 to 178 µs at unchanged misses
  a.group_m = tu.nt_group_m < 0 ? -tu.nt_group_m : (a.tiles_n >= 8 ? tu.nt_group_m : 0);
  // measured (tools/bench_ops.py --variants 2,32): +6…12 % on plain stores, a LOSS when a residual / aux operand must
  // be fetched in the row-major layout too — those keep the accumulator-layout epilogue
  a.wide_epi = tu.nt_wide_epi && !(a.flags & (LAKO_EPI_ATOMIC | LAKO_EPI_RESID | LAKO_EPI_AUXMASK)) && a.N % 8 == 0 &&
               a.ldc % 8 == 0 && reinterpret_cast<uintptr_t>(a.C) % 16 == 0;
  // persistent grid: resident workgroups only (LDS-limited: 160 KiB / CU), 256 CUs
  const int per_cu = (160 * 1024) / LDS > 0 ? (160 * 1024) / LDS : 1;
  int grid = a.tiles_m * a.tiles_n;
  if (tu.nt_persistent && grid > 256 * per_cu) grid = 256 * per_cu;
#ifdef LAKO_EXPERIMENTS
  if ((tu.nt_debug >> 8) > 0 && grid > (tu.nt_debug >> 8)) grid = tu.nt_debug >> 8;   // timing experiment: fewer resident workgroups
#endif
  // DEPHASE: every other workgroup of an XCD starts 1 µs late when the workgroups walk several tiles.  Measured (tools/bench_ops.py
  // --dephase 0,100,…): [64000,768]×[2304,768] 287 → 250 µs, o+res 140 → 134, wi 374 → 366, long launches unchanged; the size of
  // the delay (1…16 µs) and the number of phases (2, 4, 8) do not matter — the lockstep start is what costs.
  a.dephase = (tu.nt_dephase > 0 && a.tiles_m * a.tiles_n > grid) ? ((tu.nt_dephase_n << 16) | (tu.nt_dephase & 0xffff)) : 0;
  if constexpr (QUEUE) {
    a.queue = nt_queue_counters(s);
    // (several K-steps per tile to cover the ticket's round trip, several tiles per workgroup to be of any use, a persistent grid)
    if (!a.queue || a.K * (int)sizeof(T) <= TKB || a.tiles_m * a.tiles_n <= grid || !tu.nt_persistent) {
      launch_nt_cfg<T, TO, WM, WN, MT, NT, SIDE, false>(a, tu, s);
      return;
    }
  } else {
    a.queue = nullptr;
#ifdef LAKO_EXPERIMENTS
    if (tu.nt_debug & 64) a.queue = reinterpret_cast<int*>(lako_exp_nt_stamp_buffer());
#endif
  }
  hipLaunchKernelGGL((gemm_nt_kernel<T, TO, WM, WN, MT, NT, SIDE, QUEUE, PP>), dim3(grid), dim3(WM * WN * 64), LDS, s, a);
}

// 288-row tiles (MT = 9): bf16 in and out, static tile order (the QUEUE instantiations are 256-row)
template <typename T, typename TO>
bool nt288_ok(const NtArgs& a, const lako_tuning_t& tu) {
  return sizeof(T) == 2 && sizeof(TO) == 2 && !(a.flags & LAKO_EPI_ATOMIC) && !tu.nt_queue;
}

// the 256² kernel, with the LDS-staged side operand where the epilogue has exactly one (bf16 in and out, 16-B aligned rows)
template <typename T, typename TO, int MT = 8>
void launch_nt_256(const NtArgs& a, const lako_tuning_t& tu, hipStream_t s) {
  if constexpr (sizeof(T) == 2 && sizeof(TO) == 2) {
    const int side = a.flags & (LAKO_EPI_RESID | LAKO_EPI_AUXMASK);
    const char* sp = side == LAKO_EPI_RESID ? a.resid : a.aux;
    const int64_t ld = side == LAKO_EPI_RESID ? a.ldr : a.ldaux;
    if (tu.nt_side_lds && (side == LAKO_EPI_RESID || side == LAKO_EPI_AUXMASK) && !(a.flags & LAKO_EPI_ATOMIC) && a.N % 8 == 0 &&
        a.ldc % 8 == 0 && ld % 8 == 0 && reinterpret_cast<uintptr_t>(a.C) % 16 == 0 && reinterpret_cast<uintptr_t>(sp) % 16 == 0 &&
        (int64_t)256 * a.ldc * 2 < (1ll << 31) && (int64_t)256 * ld * 2 < (1ll << 31)) {
      if (MT == 8 && tu.nt_queue) launch_nt_cfg<T, TO, 2, 4, 8, 4, true, true>(a, tu, s);
      else if (MT == 8 && nt_pp_mode(tu, a)) launch_nt_cfg<T, TO, 2, 4, 8, 4, true, false, true>(a, tu, s);
      else launch_nt_cfg<T, TO, 2, 4, MT, 4, true>(a, tu, s);
      return;
    }
    if (MT == 8 && tu.nt_queue) {
      launch_nt_cfg<T, TO, 2, 4, 8, 4, false, true>(a, tu, s);
      return;
    }
    if (MT == 8 && nt_pp_mode(tu, a)) {
      launch_nt_cfg<T, TO, 2, 4, 8, 4, false, false, true>(a, tu, s);
      return;
    }
  }
  launch_nt_cfg<T, TO, 2, 4, MT, 4>(a, tu, s);
}

// the four-wave hand-scheduled kernels (gemm_nt4.h): bf16 in and out, whole 128-byte K-slices in pairs, every epilogue but the atomic one
// and the fused norm; a residual OR an aux operand (16-byte aligned rows), not both
template <typename T, typename TO>
bool nt4_ok(const NtArgs& a) {
  if (sizeof(T) != 2 || sizeof(TO) != 2 || a.K % 128 != 0 || a.K < 256 || (a.flags & (LAKO_EPI_ATOMIC | LAKO_EPI_NORM_A)) || a.N % 8 != 0 || a.ldc % 8 != 0 ||
      reinterpret_cast<uintptr_t>(a.C) % 16 != 0 || (int64_t)256 * std::max(a.lda, a.ldb) * 2 >= (1ll << 31))
    return false;
  // epilogues built into the kernels: alpha alone; {ReLU, dropout}; residual with or without dropout; the aux mask alone (all with alpha = 1)
  const int side = a.flags & (LAKO_EPI_RESID | LAKO_EPI_AUXMASK);
  if (a.alpha != 1.0f && (a.flags != 0 || a.drop_thresh != 0)) return false;
  if (a.drop_thresh != 0 && (uint64_t)(a.row0 + a.M + 256) * (uint64_t)a.N >= (1ull << 34)) return false;      // dropout quads are indexed in 32 bits
  if (side == (LAKO_EPI_RESID | LAKO_EPI_AUXMASK)) return false;
  if (side && (a.flags & LAKO_EPI_RELU)) return false;
  if (side == LAKO_EPI_AUXMASK && a.drop_thresh != 0) return false;
  if (side == LAKO_EPI_RESID && (a.ldr % 8 != 0 || reinterpret_cast<uintptr_t>(a.resid) % 16 != 0 || (int64_t)128 * a.ldr * 2 >= (1ll << 31))) return false;
  if (side == LAKO_EPI_AUXMASK && (a.ldaux % 8 != 0 || reinterpret_cast<uintptr_t>(a.aux) % 16 != 0 || (int64_t)128 * a.ldaux * 2 >= (1ll << 31))) return false;
  if ((int64_t)128 * a.ldc * 2 >= (1ll << 31)) return false;      // (32-bit offsets inside a wave's region)
  return true;
}
template <int MT>
void launch_nt4(NtArgs a, const lako_tuning_t& tu, hipStream_t s) {
  const bool side = a.flags & (LAKO_EPI_RESID | LAKO_EPI_AUXMASK);
  a.tiles_m = cdiv(a.M, MT * 32);
  a.tiles_n = cdiv(a.N, 256);
  a.group_m = tu.nt_group_m < 0 ? -tu.nt_group_m : (a.tiles_n >= 8 ? tu.nt_group_m : 0);
  int grid = a.tiles_m * a.tiles_n;
  if (grid > 256) grid = 256;
  // no start offsets between the workgroups: with this loop every offset costs at the end of the launch what it saves in the store
  // phases (tools/gemm_nt4_dephase.py, profiles/r06d_nt4_dephase.txt: lockstep is the fastest of nine settings on all five shapes)
  a.dephase = 0;
  if (tu.nt_debug & 256) a.dephase = tu.nt_dephase > 0 ? ((tu.nt_dephase_n << 16) | (tu.nt_dephase & 0xffff)) : 0;      // (A/B hook: tools/gemm_nt4_dephase.py)
  a.debug = tu.nt_debug;
  a.queue = nullptr;
  // gemm_nt_four = 1: the dropout epilogues without a side operand straight from the accumulator layout (gemm_nt4.h), 3: through the LDS
  // transposition like the others (A/B measurements)
  a.epi_direct = (tu.nt_four >> 1) ? 0 : 1;
#ifdef LAKO_EXPERIMENTS
  if (tu.nt_debug & 64) a.queue = reinterpret_cast<int*>(lako_exp_nt_stamp_buffer());
#endif
  const bool relu = a.flags & LAKO_EPI_RELU, drop = a.drop_thresh != 0;
  const int epi = (!side && drop && a.epi_direct) ? (relu ? 2 : 1) : 0;
#define LAKO_NT4_DIRECT(E)                                                                                    \
  case E:                                                                                                     \
    LAKO_SET_MAX_LDS((&gemm_nt4_kernel<MT, false, E>), nt4::LDS_BYTES);                                       \
    hipLaunchKernelGGL((gemm_nt4_kernel<MT, false, E>), dim3(grid), dim3(256), nt4::LDS_BYTES, s, a);        \
    return;
  switch (epi) {
    LAKO_NT4_DIRECT(1)
    LAKO_NT4_DIRECT(2)
    default: break;
  }
#undef LAKO_NT4_DIRECT
  if (side) {
    LAKO_SET_MAX_LDS((&gemm_nt4_kernel<MT, true>), nt4::LDS_BYTES);
    hipLaunchKernelGGL((gemm_nt4_kernel<MT, true>), dim3(grid), dim3(256), nt4::LDS_BYTES, s, a);
  } else {
    LAKO_SET_MAX_LDS((&gemm_nt4_kernel<MT, false>), nt4::LDS_BYTES);
    hipLaunchKernelGGL((gemm_nt4_kernel<MT, false>), dim3(grid), dim3(256), nt4::LDS_BYTES, s, a);
  }
}

template <typename T, typename TO>
int launch_nt(const NtArgs& a_in, const lako_tuning_t& tu, hipStream_t s) {
  NtArgs a = a_in;
  int v = tu.nt_variant;
  if (a.flags & LAKO_EPI_NORM_A) {
    // the norm is formed by the M <= 256 kernel only: decide BEFORE any planning or launch (round 5, ADVICE: with a tail plan the head
    // rows used to be launched un-normalised before the flag was refused) — the conditions of the skinny branch below, restated
    bool ok = false;
    if constexpr (sizeof(T) == 2) {
      ok = a.K % 32 == 0 && a.K >= 128 && a.K <= 1024 && a.N % 8 == 0 && (int64_t)a.M * a.lda * 2 < (1ll << 31) && (int64_t)a.N * a.ldb * 2 < (1ll << 31) &&
           a.lda % 8 == 0 && a.ldb % 8 == 0 && a.ldc % 8 == 0 && !(a.flags & LAKO_EPI_ATOMIC) &&
           (!(a.flags & LAKO_EPI_RESID) || (a.ldr % 8 == 0 && reinterpret_cast<uintptr_t>(a.resid) % 16 == 0)) &&
           (!(a.flags & LAKO_EPI_AUXMASK) || (a.ldaux % 8 == 0 && reinterpret_cast<uintptr_t>(a.aux) % 16 == 0)) &&
           reinterpret_cast<uintptr_t>(a.A) % 16 == 0 && reinterpret_cast<uintptr_t>(a.B) % 16 == 0 && reinterpret_cast<uintptr_t>(a.C) % 16 == 0 &&
           (v == 5 || (v < 0 && tu.nt_skinny && a.M <= 256 && a.N <= 4096));
    }
    if (!ok) return LAKO_E_BADARG;
  }
  if (v < 0 && tu.nt_four && !tu.nt_queue && nt4_ok<T, TO>(a) && (int64_t)cdiv(a.M, 256) * cdiv(a.N, 256) >= 256) {
    // ROUND 6: the four-wave kernels.  A launch walks its tiles in rounds of 256 (one workgroup per CU), a partly filled round costs a
    // whole one, and a tile costs ≈ 4.5 µs + nk · (1.32 µs for 256 rows | 0.99 µs for 192 rows) (nk = K-steps; fitted to
    // profiles/r06c_nt4_probe.txt — the K loop is MFMA-bound, so three quarters of the rows take three quarters of the K-step): take the
    // height whose rounds cost less.  47 757 × 768 outputs: 561 tiles of 256 rows = 3 rounds, 747 of 192 rows = 3 shorter ones.
    const int nk = a.K / 64, tn = cdiv(a.N, 256);
    const double c8 = (double)cdiv((int64_t)cdiv(a.M, 256) * tn, 256) * (4.5 + 1.32 * nk);
    const double c6 = (double)cdiv((int64_t)cdiv(a.M, 192) * tn, 256) * (4.5 + 0.99 * nk);
    if (c6 < c8) launch_nt4<6>(a, tu, s);
    else launch_nt4<8>(a, tu, s);
    return 0;
  }
  if (v < 0) {
    // big tiles once there is enough work to fill the chip with them (>= 1 tile per CU), else 128x128
    const int tn = cdiv(a.N, 256);
    const int64_t t256 = (int64_t)cdiv(a.M, 256) * tn;
    v = (t256 >= 256 && sizeof(T) == 2) ? 2 : 0;
    // TAIL SPLIT.  256 persistent workgroups walk the 256² tiles in rounds; t256 = 564 (an unpadded batch of ≈48 k
    // tokens × 768 columns) means 2 full rounds and a third with 52 tiles on 52 CUs: 3 tile-times for 2.2 of work.
    // The rows of the full rounds go to the 256² kernel, the remaining rows to a second launch with 128² tiles
    // (4 × as many, 2 workgroups per CU): ≈2.35 tile-times.  Only when the last round would be less than half full.
    // TILE HEIGHT (round 3).  The same launch with 192-row tiles (the kernel's MT = 6 instantiation: ≈0.8 of a 256-row tile's time —
    // three quarters of the FLOPs at 17 % more operand bytes per FLOP) can land on a whole number of rounds where the 256-row tiles
    // do not: 47 757 rows × 768 columns are 747 tiles = 2.92 rounds → 3 × 0.8 = 2.4 tile-times and NO tail launch, against 2 + ≈0.7.
    // Both plans are priced in 256-row tile-times and the cheaper one runs — when "gemm_nt_tile192" is 1.  It is 0 by default:
    // measured (profiles/r03e_gemm_tile192.txt, 47 757 rows) a 192-row tile takes ≈0.93 of a 256-row tile's time, not 0.8 — the
    // K-step is bound by the L2 → LDS path, whose bytes per tile fall by an eighth only — so three rounds of them (72.9 / 235 /
    // 172 µs at K = 768 / 3072 / 2304) lose to two rounds + tail (71.4 / 228 / 165 µs).  Variant 7 forces the tile (tests).
    struct Plan { double cost; int64_t full_rows; bool tail; };
    auto plan = [&](int bm, double tile_time) {
      const int64_t tiles = (int64_t)cdiv(a.M, bm) * tn;
      const int64_t rounds = tiles / 256;
      const int64_t full_rows = rounds * 256 / tn;                   // tile-rows covered by the full rounds
      const int64_t rest = tiles - full_rows * tn;                   // tiles left for the last round
      if (rest == 0) return Plan{rounds * tile_time, full_rows, false};
      // (nt_tail_split: 1 = split when the last round would be less than half full; a value > 1 is the threshold in 256² tiles)
      if (tu.nt_tail_split && rest < (tu.nt_tail_split > 1 ? tu.nt_tail_split : 128) && full_rows > 0 && !(a.flags & LAKO_EPI_ATOMIC)) return Plan{rounds * tile_time + 0.7, full_rows, true};
      return Plan{(rounds + 1) * tile_time, full_rows, false};
    };
    if (v == 2 && t256 > 256) {
      Plan pl = plan(256, 1.0);
      int bm = 256;
      if constexpr (sizeof(T) == 2 && sizeof(TO) == 2) {
        if (tu.nt_tile192) {
          const Plan p192 = plan(192, 0.8);
          if (p192.cost < 0.97 * pl.cost) {
            pl = p192;
            bm = 192;
          }
        }
        // 288-ROW TILES (round 4, MT = 9; profiles/r04h_gemm_tile_height.txt).  The other direction pays: a 288-row tile takes ≈1.13–1.16 of a
        // 256-row tile's time for 1.125 of its work, and 47 757 rows × 768 columns are 498 of them — two rounds at 97 % instead of two rounds +
        // a 128²-tile tail launch: −10…−12 % per call on the three N = 768 gradient products of a layer, −1…−3 % at N = 2304 / 3072 (one round
        // fewer).  144 accumulator registers: 243 VGPRs with the plain epilogues; with a side operand 288-row tiles need the 16-row side passes
        // (launch_nt_256<…, 9>).
        if (tu.nt_tile288 && !(a.flags & LAKO_EPI_ATOMIC) && nt288_ok<T, TO>(a, tu)) {
          const Plan p288 = plan(288, tu.nt_tile288 >= 2 ? 1.0 + 0.01 * tu.nt_tile288 : 1.13);      // (values >= 2: the price in percent over a 256-row tile, A/B)
          if (p288.cost < 0.99 * pl.cost) {
            pl = p288;
            bm = 288;
          }
        }
      }
      if (pl.tail) {
        NtArgs head = a;
        head.M = (int)(pl.full_rows * bm);
        if (bm == 192) {
          if constexpr (sizeof(T) == 2 && sizeof(TO) == 2) launch_nt_256<T, TO, 6>(head, tu, s);
        } else if (bm == 288) {
          if constexpr (sizeof(T) == 2 && sizeof(TO) == 2) launch_nt_256<T, TO, 9>(head, tu, s);
        } else {
          launch_nt_256<T, TO>(head, tu, s);
        }
        const int64_t r0 = pl.full_rows * bm;
        a.M -= (int)r0;
        a.row0 += r0;
        a.A += r0 * a.lda * (int64_t)sizeof(T);
        a.C += r0 * a.ldc * (int64_t)sizeof(TO);
        if (a.resid) a.resid += r0 * a.ldr * (int64_t)sizeof(TO);
        if (a.aux) a.aux += r0 * a.ldaux * (int64_t)sizeof(T);
        v = 0;      // the tail: 128² tiles (round 4: 256 × 128 / 8-wave tiles for the tail measured +0.2 ms on the step)
      } else if (bm == 192) {
        v = 7;
      } else if (bm == 288) {
        v = 8;
      }
    }
  }
  if ((v == 3 || v == 9) && !nt4_ok<T, TO>(a)) v = 2;   // (forced four-wave kernels: shapes / epilogues they do not take run on the eight-wave kernel)
  if (v == 5 && sizeof(T) != 2) v = 4;
  if constexpr (sizeof(T) == 2) {
    // very skinny (the decoder): 64×64 tiles, K split over the workgroup's four waves — see gemm_nt_skinny_kernel
    const bool fits = a.K % 32 == 0 && a.K >= 128 && a.K <= 4096 && a.N % 8 == 0 && (int64_t)a.M * a.lda * 2 < (1ll << 31) && (int64_t)a.N * a.ldb * 2 < (1ll << 31) && a.lda % 8 == 0 && a.ldb % 8 == 0 && a.ldc % 8 == 0 &&
                      !(a.flags & LAKO_EPI_ATOMIC) && (!(a.flags & LAKO_EPI_RESID) || (a.ldr % 8 == 0 && reinterpret_cast<uintptr_t>(a.resid) % 16 == 0)) &&
                      (!(a.flags & LAKO_EPI_AUXMASK) || (a.ldaux % 8 == 0 && reinterpret_cast<uintptr_t>(a.aux) % 16 == 0)) &&
                      reinterpret_cast<uintptr_t>(a.A) % 16 == 0 && reinterpret_cast<uintptr_t>(a.B) % 16 == 0 &&
                      reinterpret_cast<uintptr_t>(a.C) % 16 == 0;
    if (fits && (v == 5 || (tu.nt_variant < 0 && tu.nt_skinny && a.M <= 256 && a.N <= 4096))) {
      LAKO_SET_MAX_LDS((&gemm_nt_skinny_kernel<TO, 8, 4, 4>), 8 * 16384);
LAKO_SET_MAX_LDS((&gemm_nt_skinny_kernel<TO, 8, 8, 4>), 8 * 16384);
LAKO_SET_MAX_LDS((&gemm_nt_skinny_kernel<TO, 8, 16, 4>), 8 * 16384);
      // a CU pulls ≈45 GB/s through its L1 whatever the other CUs do, so the bytes have to be spread over many CUs: 32² tiles
      // (4 × the workgroups of 64² tiles, half the operand bytes each) — measured on the whole training step: 64² 50.4 ms,
      // 32² 49.4 ms, 16² 49.3 ms ("gemm_nt_skinny" 2 / 3 / 4 force them)
      const bool small = tu.nt_skinny != 2;
      const int ts = small ? 32 : 64;
      const dim3 grid(cdiv(a.N, ts), cdiv(a.M, ts));
      const int per = cdiv(a.K / 32, 8);
      if (a.flags & LAKO_EPI_NORM_A) {      // RMSNorm of A in the fragments: K <= 1024 (all K-steps of a wave resident), 32² tiles
        if (per > 4) return LAKO_E_BADARG;
        const dim3 g32(cdiv(a.N, 32), cdiv(a.M, 32));
        hipLaunchKernelGGL((gemm_nt_skinny_kernel<TO, 8, 4, 2, true>), g32, dim3(512), 8 * 4096 + 8 * 32 * 4, s, a);
        return 0;
      }
      if (tu.nt_skinny == 4) {
        const dim3 g16(cdiv(a.N, 16), cdiv(a.M, 16));
        if (per <= 4) hipLaunchKernelGGL((gemm_nt_skinny_kernel<TO, 8, 4, 1>), g16, dim3(512), 8 * 1024, s, a);
        else if (per <= 8) hipLaunchKernelGGL((gemm_nt_skinny_kernel<TO, 8, 8, 1>), g16, dim3(512), 8 * 1024, s, a);
        else hipLaunchKernelGGL((gemm_nt_skinny_kernel<TO, 8, 16, 1>), g16, dim3(512), 8 * 1024, s, a);
      } else if (small) {
        if (per <= 4) hipLaunchKernelGGL((gemm_nt_skinny_kernel<TO, 8, 4, 2>), grid, dim3(512), 8 * 4096, s, a);
        else if (per <= 8) hipLaunchKernelGGL((gemm_nt_skinny_kernel<TO, 8, 8, 2>), grid, dim3(512), 8 * 4096, s, a);
        else hipLaunchKernelGGL((gemm_nt_skinny_kernel<TO, 8, 16, 2>), grid, dim3(512), 8 * 4096, s, a);
      } else {
        if (per <= 4) hipLaunchKernelGGL((gemm_nt_skinny_kernel<TO, 8, 4, 4>), grid, dim3(512), 8 * 16384, s, a);
        else if (per <= 8) hipLaunchKernelGGL((gemm_nt_skinny_kernel<TO, 8, 8, 4>), grid, dim3(512), 8 * 16384, s, a);
        else hipLaunchKernelGGL((gemm_nt_skinny_kernel<TO, 8, 16, 4>), grid, dim3(512), 8 * 16384, s, a);
      }
      return 0;
    }
    if (v == 5) v = 4;
  }
  if (a.flags & LAKO_EPI_NORM_A) return LAKO_E_BADARG;      // only the skinny kernel forms the norm in its fragments
  if (v == 4 || (tu.nt_variant < 0 && tu.nt_ring && (int64_t)cdiv(a.M, RING_BM) * cdiv(a.N, RING_BM) <= 256)) {
    // skinny: at most one 128² tile per CU → the 4-slot ring hides the global→LDS latency inside the workgroup
    LAKO_SET_MAX_LDS((&gemm_nt_ring_kernel<T, TO>), RING_NST * RING_STAGE);
    NtArgs b = a;
    b.tiles_m = cdiv(a.M, RING_BM);
    b.tiles_n = cdiv(a.N, RING_BM);
    b.debug = 0;
    const int tiles = b.tiles_m * b.tiles_n, nk_all = cdiv(a.K * (int)sizeof(T), TKB);
    b.split_k = 1;
    // a few tiles with a very long K (LM-head backward: 6 tiles × 502 K-steps) would run on 6 CUs: with fp32 atomic
    // accumulation the K range may be cut into pieces of >= 8 K-steps over idle CUs (alpha is linear, no other epilogue)
    if (a.flags == LAKO_EPI_ATOMIC && sizeof(TO) == 4 && tiles < 128 && nk_all >= 32)
      b.split_k = std::max(1, std::min(256 / tiles, nk_all / 8));
    b.k_chunk = cdiv(nk_all, b.split_k) * TKB;
    b.split_k = cdiv(nk_all * TKB, b.k_chunk);
    hipLaunchKernelGGL((gemm_nt_ring_kernel<T, TO>), dim3(tiles * b.split_k), dim3(256), RING_NST * RING_STAGE, s, b);
    return 0;
  }
  if ((v == 7 || v == 8) && (sizeof(T) != 2 || sizeof(TO) != 2)) v = 2;
  if (v == 2) launch_nt_256<T, TO>(a, tu, s);
  else if (v == 7) {
    if constexpr (sizeof(T) == 2 && sizeof(TO) == 2) launch_nt_256<T, TO, 6>(a, tu, s);     // 192 x 256 tiles
  } else if (v == 8) {
    if constexpr (sizeof(T) == 2 && sizeof(TO) == 2) {                                          // 288 x 256 tiles
      if (nt288_ok<T, TO>(a, tu)) launch_nt_256<T, TO, 9>(a, tu, s);
      else launch_nt_256<T, TO>(a, tu, s);
    }
  } else if (v == 9 && nt4_ok<T, TO>(a)) {
    launch_nt4<8>(a, tu, s);
  } else if (v == 3 && nt4_ok<T, TO>(a)) {
    launch_nt4<6>(a, tu, s);
  } else if (v == 6 && sizeof(T) == 2) launch_nt_cfg<T, TO, 2, 2, 8, 8>(a, tu, s);   // EXPERIMENT: 256x256 on 4 waves of 128x128 (hipBLASLt's MT256x256x64 MIWT8_8 shape)
  else if (v == 1) launch_nt_cfg<T, TO, 4, 2, 4, 4>(a, tu, s);
  else launch_nt_cfg<T, TO, 2, 2, 4, 4>(a, tu, s);
  return 0;
}

template <typename T>
int launch_tn(const TnArgs& a, hipStream_t s) {
  LAKO_SET_MAX_LDS((&gemm_tn_kernel<T>), GEMM_LDS);
  hipLaunchKernelGGL((gemm_tn_kernel<T>), dim3(a.tiles_m * a.tiles_n * a.split_k), dim3(256), GEMM_LDS, s, a);
  return 0;
}

}  // namespace

extern "C" int lako_gemm_nt(const lako_gemm_nt_t* p, lako_stream_t stream) {
  LAKO_CHECK_ARG(p != nullptr, "lako_gemm_nt: null params");
  LAKO_CHECK_ARG(p->M > 0 && p->N > 0 && p->K > 0, "lako_gemm_nt: bad dims M=%lld N=%lld K=%lld", (long long)p->M,
                 (long long)p->N, (long long)p->K);
  LAKO_CHECK_ARG(p->in_dtype == LAKO_F32 || p->in_dtype == LAKO_BF16, "lako_gemm_nt: bad in_dtype");
  LAKO_CHECK_ARG(p->out_dtype == LAKO_F32 || p->out_dtype == LAKO_BF16, "lako_gemm_nt: bad out_dtype");
  const int esz = p->in_dtype == LAKO_F32 ? 4 : 2;
  LAKO_CHECK_ARG((p->K * esz) % 16 == 0 && (p->lda * esz) % 16 == 0 && (p->ldb * esz) % 16 == 0,
                 "lako_gemm_nt: K/lda/ldb must make 16-byte rows (K=%lld lda=%lld ldb=%lld)", (long long)p->K,
                 (long long)p->lda, (long long)p->ldb);
  LAKO_CHECK_ARG(p->N % 4 == 0 && p->ldc % 4 == 0, "lako_gemm_nt: N and ldc must be multiples of 4");
  LAKO_CHECK_ARG(p->M < (1 << 30) && p->N < (1 << 30) && p->K < (1 << 28), "lako_gemm_nt: dims too large");
  LAKO_CHECK_ALIGN(p->A, 16);
  LAKO_CHECK_ALIGN(p->B, 16);
  LAKO_CHECK_ALIGN(p->C, 8);
  if (p->flags & LAKO_EPI_RESID) {
    LAKO_CHECK_ARG(p->resid != nullptr && p->ldr % 4 == 0, "lako_gemm_nt: RESID needs resid, ldr %% 4 == 0");
  }
  if (p->flags & LAKO_EPI_AUXMASK) {
    LAKO_CHECK_ARG(p->aux != nullptr && p->ldaux % 4 == 0, "lako_gemm_nt: AUXMASK needs aux, ldaux %% 4 == 0");
  }
  if (p->flags & LAKO_EPI_ATOMIC) LAKO_CHECK_ARG(p->out_dtype == LAKO_F32, "lako_gemm_nt: ATOMIC needs fp32 C");
  LAKO_CHECK_ARG(p->drop.p >= 0.f && p->drop.p < 1.f, "lako_gemm_nt: dropout p out of range");
  NtArgs a{};
  a.row0 = 0;
  a.A = (const char*)p->A;
  a.B = (const char*)p->B;
  a.C = (char*)p->C;
  a.resid = (const char*)p->resid;
  a.aux = (const char*)p->aux;
  a.M = (int)p->M;
  a.N = (int)p->N;
  a.K = (int)p->K;
  a.lda = p->lda;
  a.ldb = p->ldb;
  a.ldc = p->ldc;
  a.ldr = p->ldr;
  a.ldaux = p->ldaux;
  a.alpha = p->alpha;
  a.aux_scale = p->aux_scale;
  a.flags = p->flags;
  a.drop_thresh = p->drop.p > 0.f ? lako_drop_thresh(p->drop.p) : 0u;
  a.drop_scale = p->drop.p > 0.f ? 1.0f / (1.0f - p->drop.p) : 1.0f;
  a.drop_key = lako_drop_key(p->drop.seed, p->drop.site);
  if (p->flags & LAKO_EPI_NORM_A) {
    if (p->in_dtype != LAKO_BF16) {
      lako_set_error("lako_gemm_nt: LAKO_EPI_NORM_A is implemented for bf16 operands only");
      return LAKO_E_UNSUPPORTED;
    }
    LAKO_CHECK_ARG(p->norm_w != nullptr && p->norm_eps > 0.f, "lako_gemm_nt: NORM_A needs norm_w and norm_eps > 0");
    LAKO_CHECK_ARG(p->norm_out == nullptr || (p->norm_ld >= p->K && p->norm_ld % 8 == 0 && reinterpret_cast<uintptr_t>(p->norm_out) % 16 == 0),
                   "lako_gemm_nt: norm_out needs norm_ld >= K, norm_ld %% 8 == 0 and a 16-byte aligned pointer");
    LAKO_CHECK_ALIGN(p->norm_w, 16);
    a.norm_w = p->norm_w;
    a.norm_eps = p->norm_eps;
    a.norm_out = (char*)p->norm_out;
    a.norm_ld = p->norm_ld;
    a.norm_rs = p->norm_rstd;
  }
  hipStream_t s = (hipStream_t)stream;
  const lako_tuning_t& tu = p->tuning ? *p->tuning : process_tuning();
  int rc = 0;
  if (p->in_dtype == LAKO_BF16) {
    if (p->out_dtype == LAKO_BF16) rc = launch_nt<bf16_t, bf16_t>(a, tu, s);
    else rc = launch_nt<bf16_t, float>(a, tu, s);
  } else {
    if (p->out_dtype == LAKO_BF16) rc = launch_nt<float, bf16_t>(a, tu, s);
    else rc = launch_nt<float, float>(a, tu, s);
  }
  if (rc != 0) {      // (nothing was launched)
    lako_set_error("lako_gemm_nt: LAKO_EPI_NORM_A is formed by the M <= 256 kernel only (bf16, K %% 32 == 0, K <= 1024, 16-byte aligned rows): M=%lld N=%lld K=%lld",
                   (long long)p->M, (long long)p->N, (long long)p->K);
    return LAKO_E_UNSUPPORTED;
  }
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

// K-splits of the 256×256 weight-gradient kernel.  One workgroup per CU is resident, so `tiles·sk` workgroups run in
// ⌈tiles·sk / 256⌉ rounds of K/sk rows each, and every split pays one fp32 atomic pass over its 256 KiB tile (≈1.3 TB/s chip-wide,
// i.e. ≈51 µs per tile and split against ≈1.65 µs per 64-row K-step).  Pick the split that minimises rounds × (steps/sk × 1.65 + 51):
// 108 tiles (a T5-base layer's four weight gradients) → 2 splits as before; 192 tiles (T5-large) → 4 splits, three full rounds of a
// quarter of K instead of one round on 75 % of the CUs (config 4: −16 % on the weight-gradient launches).
static int tn_pick_split(int tiles, int64_t K, int max_split) {
  const double steps = (double)((K + 63) / 64);
  int best = 1;
  double best_cost = 1e300;
  for (int sk = 1; sk <= 16 && sk <= max_split; ++sk) {
    const int rounds = (tiles * sk + 255) / 256;
    const double cost = rounds * (steps / sk * 1.65 + (sk > 1 ? 51.0 : 25.0));
    if (cost < best_cost * 0.97) {      // a larger split must win by 3 %: more atomics, more launch
      best_cost = cost;
      best = sk;
    }
  }
  return best;
}

static int tn_single(const void* A, const void* B, float* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                     int in_dtype, float alpha, int split_k, const lako_tuning_t* tuning, lako_stream_t stream, int rows_out);

extern "C" int lako_gemm_tn(const void* A, const void* B, float* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                            int64_t ldb, int64_t ldc, int in_dtype, float alpha, int split_k, const lako_tuning_t* tuning,
                            lako_stream_t stream) {
  return tn_single(A, B, C, M, N, K, lda, ldb, ldc, in_dtype, alpha, split_k, tuning, stream, 0);
}

// rows_out: 0 = M, else only the first rows_out rows of C are written (lako_gemm_tn_item_t.rows_out)
static int tn_single(const void* A, const void* B, float* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                     int in_dtype, float alpha, int split_k, const lako_tuning_t* tuning, lako_stream_t stream, int rows_out) {
  const lako_tuning_t& tu = tuning ? *tuning : process_tuning();
  LAKO_CHECK_ARG(M > 0 && N > 0 && K > 0, "lako_gemm_tn: bad dims");
  LAKO_CHECK_ARG(split_k >= -2, "lako_gemm_tn: split_k %d", split_k);
  const int out_mode = split_k < 0 ? -split_k : 0;      // -1: exclusive read-modify-write, -2: overwrite (one K-split either way)
  if (split_k < 0) split_k = 1;
  if (lako_det::on() && split_k == 0) split_k = 1;      // LAKO_DETERMINISTIC=1: K-splits meet by float atomics in arrival order — one contributor
  LAKO_CHECK_ARG(in_dtype == LAKO_F32 || in_dtype == LAKO_BF16, "lako_gemm_tn: bad in_dtype");
  const int esz = in_dtype == LAKO_F32 ? 4 : 2;
  LAKO_CHECK_ARG((M * esz) % 16 == 0 && (N * esz) % 16 == 0 && (lda * esz) % 16 == 0 && (ldb * esz) % 16 == 0,
                 "lako_gemm_tn: M/N/lda/ldb must make 16-byte rows");
  LAKO_CHECK_ARG(M < (1 << 30) && N < (1 << 30) && K < (1 << 30), "lako_gemm_tn: dims too large");
  LAKO_CHECK_ALIGN(A, 16);
  LAKO_CHECK_ALIGN(B, 16);
  LAKO_CHECK_ALIGN(C, 4);
  TnArgs a{};
  a.n_items = 0;
  a.A = (const char*)A;
  a.B = (const char*)B;
  a.C = C;
  a.M = (int)M;
  a.N = (int)N;
  a.K = (int)K;
  a.lda = lda;
  a.ldb = ldb;
  a.ldc = ldc;
  a.alpha = alpha;
  LAKO_CHECK_ARG(rows_out >= 0 && rows_out <= M, "lako_gemm_tn: rows_out");
  a.Mout = rows_out > 0 ? rows_out : (int)M;
  a.tiles_m = cdiv(M, TM);
  a.tiles_n = cdiv(N, TN_);
  hipStream_t s = (hipStream_t)stream;
  if (in_dtype == LAKO_BF16 && M >= 256 && N >= 256 && tu.tn_big) {
    a.tiles_m = cdiv(M, 256);
    a.tiles_n = cdiv(N, 256);
    const int tiles = a.tiles_m * a.tiles_n;
    int sk = split_k;
    if (sk <= 0) {   // one workgroup per CU (128 KiB LDS): aim at ~256 workgroups, >= 4 K-steps per split
      const int max_split = cdiv(K, 64 * 4);
      sk = tn_pick_split(tiles, K, max_split);
      if (tu.tn_split > 0) sk = tu.tn_split;
      if (sk > max_split) sk = max_split;
      if (sk < 1) sk = 1;
    }
    int chunk = cdiv(cdiv(K, sk), 64) * 64;
    a.split_k = cdiv(K, chunk);
    a.k_chunk = chunk;
    a.no_atomics = tu.tn_big == 2;
    a.glds = tu.nt_glds;
    a.no_stagger = 1;      // (as the grouped launch: see lako_gemm_tn_grouped)
    a.out_mode = a.split_k == 1 ? out_mode : 0;
    a.wide_out = reinterpret_cast<uintptr_t>(C) % 16 == 0 && ldc % 4 == 0 && N % 4 == 0;
    a.slabs = nullptr;
    a.tickets = nullptr;
    LAKO_CHECK_ARG((int64_t)64 * lda * 2 < (1ll << 31) && (int64_t)64 * ldb * 2 < (1ll << 31),
                   "lako_gemm_tn: leading dimension too large");
    if (tn4_ok(a, tu)) {
      LAKO_SET_MAX_LDS((&gemm_tn4_kernel), tn4::LDS_BYTES);
      hipLaunchKernelGGL(gemm_tn4_kernel, dim3(tiles * a.split_k), dim3(256), tn4::LDS_BYTES, s, a);
    } else if (a.glds && a.k_chunk % 64 == 0) {
      LAKO_SET_MAX_LDS((&gemm_tn256_kernel<true>), 4 * TN2_IMG);
      hipLaunchKernelGGL(gemm_tn256_kernel<true>, dim3(tiles * a.split_k), dim3(512), 4 * TN2_IMG, s, a);
    } else {
      LAKO_SET_MAX_LDS((&gemm_tn256_kernel<false>), 4 * TN2_IMG);
      hipLaunchKernelGGL(gemm_tn256_kernel<false>, dim3(tiles * a.split_k), dim3(512), 4 * TN2_IMG, s, a);
    }
    LAKO_LAUNCH_CHECK();
    return LAKO_OK;
  }
  if (out_mode == 2) {      // the small-tile kernel only adds: overwrite = zero, then add
    if (hipMemset2DAsync(C, (size_t)ldc * 4, 0, (size_t)N * 4, (size_t)a.Mout, (hipStream_t)stream) != hipSuccess) {
      lako_set_error("lako_gemm_tn: could not zero the output");
      return LAKO_E_LAUNCH;
    }
  }
  const int kr = in_dtype == LAKO_BF16 ? 64 : 32;
  if (split_k <= 0) {  // auto: aim at >= 2 workgroups per CU, at least 4 K-steps per split
    int tiles = a.tiles_m * a.tiles_n;
    split_k = cdiv(512, tiles);
    int max_split = cdiv(K, (int64_t)kr * 4);
    if (split_k > max_split) split_k = max_split;
    if (split_k < 1) split_k = 1;
  }
  int chunk = cdiv(K, split_k);
  chunk = cdiv(chunk, kr) * kr;
  a.split_k = cdiv(K, chunk);
  a.k_chunk = chunk;
  // voffset (row * ld_bytes) must stay below 2^32 within one K-step: rows < 64
  LAKO_CHECK_ARG((int64_t)64 * lda * esz < (1ll << 31) && (int64_t)64 * ldb * esz < (1ll << 31),
                 "lako_gemm_tn: leading dimension too large");
  if (in_dtype == LAKO_BF16) launch_tn<bf16_t>(a, s);
  else launch_tn<float>(a, s);
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

// K-splits the grouped launch would use (the same arithmetic as the launch itself)
static int tn_grouped_plan(const lako_gemm_tn_item_t* items, int n_items, int64_t K, int split_k, const lako_tuning_t& tu, int* tiles_out) {
  int tiles = 0;
  for (int i = 0; i < n_items; ++i) tiles += cdiv((int)items[i].M, 256) * cdiv((int)items[i].N, 256);
  const int max_split = cdiv(K, 64 * 4);
  if (lako_det::on() && split_k == -3) split_k = 0;      // (as the launch)
  int sk = tn_pick_split(tiles, K, max_split);
  if (tu.tn_split > 0) sk = tu.tn_split;
  if (split_k > 0) sk = split_k;
  if (split_k < 0) sk = 1;
  if (sk > max_split) sk = max_split;
  if (sk < 1) sk = 1;
  const int chunk = cdiv(cdiv(K, sk), 64) * 64;
  *tiles_out = tiles;
  return cdiv(K, chunk);
}

extern "C" int64_t lako_gemm_tn_grouped_workspace(const lako_gemm_tn_item_t* items, int n_items, int64_t K, int in_dtype, int split_k,
                                                  const lako_tuning_t* tuning) {
  const lako_tuning_t& tu = tuning ? *tuning : process_tuning();
  if (!items || n_items < 1 || n_items > LAKO_TN_GROUP_MAX || K <= 0 || in_dtype != LAKO_BF16 || !tu.tn_big) return 0;
  for (int i = 0; i < n_items; ++i)
    if (items[i].M < 256 || items[i].N < 256) return 0;
  int tiles = 0;
  const int sk = tn_grouped_plan(items, n_items, K, split_k, tu, &tiles);
  if (sk <= 1) return 0;
  return (int64_t)tiles * sk * 65536 * 4 + (int64_t)((tiles + 63) / 64) * 256;      // slabs + tickets (256-byte granules)
}

extern "C" int lako_gemm_tn_grouped(const lako_gemm_tn_item_t* items, int n_items, int64_t K, int in_dtype, int split_k,
                                    const lako_tuning_t* tuning, void* workspace, int64_t workspace_bytes, lako_stream_t stream) {
  const lako_tuning_t& tu = tuning ? *tuning : process_tuning();
  LAKO_CHECK_ARG(items && n_items >= 1 && n_items <= LAKO_TN_GROUP_MAX, "lako_gemm_tn_grouped: 1..%d items", LAKO_TN_GROUP_MAX);
  LAKO_CHECK_ARG(split_k >= -3, "lako_gemm_tn_grouped: split_k %d", split_k);
  LAKO_CHECK_ARG(K > 0 && K < (1 << 30), "lako_gemm_tn_grouped: bad K");
  if (split_k < 0) {
    // split_k < 0 promises "one contributor per element of C" (plain read-modify-write / overwrite / the hybrid schedule): two items whose
    // outputs share an element would race or overwrite each other silently — refuse them here (n <= 48).  An item touches rows
    // [0, rows_out or M) × columns [0, N) of its C; two items with the same ldc are compared as rectangles of one matrix (column blocks of
    // one gradient are fine), others by their byte ranges.
    auto rows_of = [](const lako_gemm_tn_item_t& p) { return p.rows_out > 0 ? (int64_t)p.rows_out : p.M; };
    for (int i = 0; i < n_items; ++i) {
      const lako_gemm_tn_item_t& p = items[i];
      if (!p.c || p.M <= 0 || p.N <= 0) continue;      // (reported per item below)
      for (int j = i + 1; j < n_items; ++j) {
        const lako_gemm_tn_item_t& q = items[j];
        if (!q.c || q.M <= 0 || q.N <= 0) continue;
        const lako_gemm_tn_item_t &lo = p.c <= q.c ? p : q, &hi = p.c <= q.c ? q : p;
        const int64_t d = hi.c - lo.c;                   // in elements
        bool overlap;
        if (lo.ldc == hi.ldc && lo.ldc > 0 && lo.N <= lo.ldc && d % lo.ldc + hi.N <= lo.ldc)
          overlap = d / lo.ldc < rows_of(lo) && d % lo.ldc < lo.N;
        else
          overlap = d < (rows_of(lo) - 1) * lo.ldc + lo.N;
        LAKO_CHECK_ARG(!overlap, "lako_gemm_tn_grouped: split_k %d needs disjoint outputs, items %d and %d overlap", split_k, i, j);
      }
    }
  }
  // LAKO_DETERMINISTIC=1: no hybrid schedule (its short units meet by float atomics); the uniform splits below go through the
  // workspace slabs, summed in split order — and without a workspace one contributor per element
  const bool det = lako_det::on();
  if (det && split_k == -3) split_k = 0;
  bool big = in_dtype == LAKO_BF16 && tu.tn_big;
  for (int i = 0; i < n_items; ++i) big = big && items[i].M >= 256 && items[i].N >= 256;
  if (!big) {   // shapes the 256×256 kernel does not take: one launch per problem
    for (int i = 0; i < n_items; ++i) {
      int rc = tn_single(items[i].a, items[i].b, items[i].c, items[i].M, items[i].N, K, items[i].lda, items[i].ldb,
                         items[i].ldc, in_dtype, items[i].alpha, split_k == -3 ? 0 : split_k, tuning, stream, items[i].rows_out);
      if (rc != LAKO_OK) return rc;
    }
    return LAKO_OK;
  }
  TnArgs a{};
  a.n_items = n_items;
  a.K = (int)K;
  a.no_atomics = tu.tn_big == 2;
  a.glds = tu.nt_glds;
  int tiles = 0;
  for (int i = 0; i < n_items; ++i) {
    const lako_gemm_tn_item_t& p = items[i];
    LAKO_CHECK_ARG(p.a && p.b && p.c && p.M > 0 && p.N > 0, "lako_gemm_tn_grouped: item %d: null operand / empty problem", i);
    LAKO_CHECK_ARG((p.M * 2) % 16 == 0 && (p.N * 2) % 16 == 0 && (p.lda * 2) % 16 == 0 && (p.ldb * 2) % 16 == 0,
                   "lako_gemm_tn_grouped: item %d: M/N/lda/ldb must make 16-byte rows", i);
    LAKO_CHECK_ARG(p.M < (1 << 30) && p.N < (1 << 30) && (int64_t)64 * p.lda * 2 < (1ll << 31) && (int64_t)64 * p.ldb * 2 < (1ll << 31),
                   "lako_gemm_tn_grouped: item %d: dims too large", i);
    LAKO_CHECK_ALIGN(p.a, 16);
    LAKO_CHECK_ALIGN(p.b, 16);
    LAKO_CHECK_ALIGN(p.c, 4);
    TnArgs::Item& d = a.items[i];
    d.A = (const char*)p.a; d.B = (const char*)p.b; d.C = p.c; d.M = (int)p.M; d.N = (int)p.N;
    d.lda = p.lda; d.ldb = p.ldb; d.ldc = p.ldc; d.alpha = p.alpha;
    LAKO_CHECK_ARG(p.rows_out >= 0 && p.rows_out <= p.M, "lako_gemm_tn_grouped: item %d: rows_out", i);
    d.Mout = p.rows_out > 0 ? p.rows_out : (int)p.M;
    d.tiles_n = cdiv((int)p.N, 256);
    d.tile_start = tiles;
    tiles += cdiv((int)p.M, 256) * d.tiles_n;
  }
  // every wave issues its DMA at the top of the K-step (round 4, with the full-K units: 7.29 -> 7.04 ms per step against waves 4-7 issuing
  // theirs two MFMA rows later, the NT kernel's order; LAKO_TN_STAGGER=1 restores that)
  static const bool tn_stag = getenv("LAKO_TN_STAGGER") && atoi(getenv("LAKO_TN_STAGGER")) == 1;
  a.no_stagger = !tn_stag;
  a.tiles_m = tiles;     // the kernel only uses the product
  a.tiles_n = 1;
  a.wide_out = 1;
  for (int i = 0; i < n_items; ++i)
    if (reinterpret_cast<uintptr_t>(items[i].c) % 16 != 0 || items[i].ldc % 4 != 0 || items[i].N % 4 != 0) a.wide_out = 0;
  const int max_split = cdiv(K, 64 * 4);
  int sk = tn_pick_split(tiles, K, max_split);
  if (tu.tn_split > 0) sk = tu.tn_split;
  if (split_k > 0) sk = split_k;
  if (split_k == -1 || split_k == -2) sk = 1;
  // HYBRID (split_k −3, round 4): ⌊tiles / 256⌋ whole rounds of full-K tiles (one contributor: plain adds, no atomic pass, a third of the
  // prologues) + the remaining tiles cut finely enough to fill one or a few short rounds.  Uniform splits pay a last round of full-length
  // units on a fraction of the chip (1 296 tiles × 3 = 15.19 rounds → 16) and one atomic pass of 256 KiB per unit — ≈ 50 µs of a 450 µs
  // unit at the chip's ≈ 1.3 TB/s, during which the CU's only workgroup does nothing else.
  a.t_full = 0;
  if (split_k == -3) {
    const int full = (tiles / 256) * 256, rem = tiles - full;
    if (full > 0) {
      int best = 1;
      double best_cost = 1e30;
      for (int c = 1; c <= 16 && c <= max_split; ++c) {      // remainder phase in full-unit times: rounds of 1/c-length units
        const double cost = rem ? (double)cdiv((int64_t)rem * c, 256) / c + 0.004 * c : 0.0;
        if (cost < best_cost - 1e-9) { best_cost = cost; best = c; }
      }
      a.t_full = full;
      sk = best;
    } else {
      sk = tn_pick_split(tiles, K, max_split);
    }
  }
  if (sk > max_split) sk = max_split;
  if (sk < 1) sk = 1;
  const int chunk = cdiv(cdiv(K, sk), 64) * 64;
  a.split_k = cdiv(K, chunk);
  a.k_chunk = chunk;
  a.out_mode = ((split_k == -1 || split_k == -2) && a.split_k == 1) ? -split_k : 0;
  a.slabs = nullptr;
  a.tickets = nullptr;
  if (a.split_k > 1 && workspace != nullptr && a.t_full == 0) {      // K-splits meet through partial tiles in the caller's workspace instead of atomics
    const int64_t slab_bytes = (int64_t)tiles * a.split_k * 65536 * 4, tick_bytes = (int64_t)((tiles + 63) / 64) * 256;
    if (workspace_bytes >= slab_bytes + tick_bytes && reinterpret_cast<uintptr_t>(workspace) % 16 == 0) {
      a.slabs = reinterpret_cast<float*>(workspace);
      a.tickets = reinterpret_cast<int*>(reinterpret_cast<char*>(workspace) + slab_bytes);
      if (hipMemsetAsync(a.tickets, 0, (size_t)tick_bytes, (hipStream_t)stream) != hipSuccess) {
        (void)hipGetLastError();
        a.slabs = nullptr;
        a.tickets = nullptr;
      }
    }
  }
  if (det && a.split_k > 1 && a.slabs == nullptr) {
    a.split_k = 1;
    a.k_chunk = cdiv(K, 64) * 64;
  }
  const int grid = a.t_full > 0 ? 8 * (a.t_full / 8 + cdiv((int64_t)(tiles - a.t_full) * a.split_k, 8)) : tiles * a.split_k;
  if (tn4_ok(a, tu)) {
    LAKO_SET_MAX_LDS((&gemm_tn4_kernel), tn4::LDS_BYTES);
    hipLaunchKernelGGL(gemm_tn4_kernel, dim3(grid), dim3(256), tn4::LDS_BYTES, (hipStream_t)stream, a);
  } else if (a.glds && a.k_chunk % 64 == 0) {
    LAKO_SET_MAX_LDS((&gemm_tn256_kernel<true>), 4 * TN2_IMG);
    hipLaunchKernelGGL(gemm_tn256_kernel<true>, dim3(grid), dim3(512), 4 * TN2_IMG, (hipStream_t)stream, a);
  } else {
    LAKO_SET_MAX_LDS((&gemm_tn256_kernel<false>), 4 * TN2_IMG);
    hipLaunchKernelGGL(gemm_tn256_kernel<false>, dim3(grid), dim3(512), 4 * TN2_IMG, (hipStream_t)stream, a);
  }
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

extern "C" int lako_mx_quantize(const void* x, int64_t rows, int64_t K, int64_t ld, void* q, uint8_t* scales, lako_stream_t stream) {
  LAKO_CHECK_ARG(x && q && scales, "lako_mx_quantize: null tensor");
  LAKO_CHECK_ARG(rows > 0 && K > 0 && K % 128 == 0 && K < (1 << 24) && ld % 8 == 0 && ld >= K, "lako_mx_quantize: K must be a positive multiple of 128, ld %% 8 == 0");
  LAKO_CHECK_ALIGN(x, 16);
  LAKO_CHECK_ALIGN(q, 8);
  const int ksp = (int)(((K / 128) + 3) / 4 * 4);
  const int64_t total = rows * (K / 8);
  int64_t grid = (total + 255) / 256;
  if (grid > 256 * 16) grid = 256 * 16;
  hipLaunchKernelGGL(mx_quantize_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, rows, (int)K, ld,
                     (uint8_t*)q, scales, ksp);
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

extern "C" int lako_gemm_nt_mx(const lako_gemm_nt_t* p, const uint8_t* a_scales, const uint8_t* b_scales, lako_stream_t stream) {
  LAKO_CHECK_ARG(p != nullptr && a_scales && b_scales, "lako_gemm_nt_mx: null params");
  LAKO_CHECK_ARG(p->M > 0 && p->N > 0 && p->K > 0, "lako_gemm_nt_mx: bad dims");
  LAKO_CHECK_ARG(p->in_dtype == LAKO_FP8_E4M3 && p->out_dtype == LAKO_BF16, "lako_gemm_nt_mx: e4m3 operands, bf16 output");
  LAKO_CHECK_ARG(p->K % 128 == 0 && p->lda % 16 == 0 && p->ldb % 16 == 0, "lako_gemm_nt_mx: K %% 128 == 0, 16-byte rows");
  LAKO_CHECK_ARG(p->N % 8 == 0 && p->ldc % 8 == 0, "lako_gemm_nt_mx: N and ldc must be multiples of 8");
  LAKO_CHECK_ARG(p->M < (1 << 30) && p->N < (1 << 30) && p->K < (1 << 24), "lako_gemm_nt_mx: dims too large");
  LAKO_CHECK_ARG(!(p->flags & LAKO_EPI_ATOMIC), "lako_gemm_nt_mx: no atomic epilogue");
  LAKO_CHECK_ALIGN(p->A, 16);
  LAKO_CHECK_ALIGN(p->B, 16);
  LAKO_CHECK_ALIGN(p->C, 16);
  LAKO_CHECK_ALIGN(a_scales, 4);
  LAKO_CHECK_ALIGN(b_scales, 4);
  if (p->flags & LAKO_EPI_RESID) LAKO_CHECK_ARG(p->resid != nullptr && p->ldr % 4 == 0, "lako_gemm_nt_mx: RESID needs resid, ldr %% 4 == 0");
  if (p->flags & LAKO_EPI_AUXMASK) LAKO_CHECK_ARG(p->aux != nullptr && p->ldaux % 4 == 0, "lako_gemm_nt_mx: AUXMASK needs aux (bf16), ldaux %% 4 == 0");
  MxArgs m;
  NtArgs& a = m.nt;
  a = NtArgs{};
  a.A = (const char*)p->A;
  a.B = (const char*)p->B;
  a.C = (char*)p->C;
  a.resid = (const char*)p->resid;
  a.aux = (const char*)p->aux;
  a.M = (int)p->M;
  a.N = (int)p->N;
  a.K = (int)p->K;
  a.lda = p->lda;
  a.ldb = p->ldb;
  a.ldc = p->ldc;
  a.ldr = p->ldr;
  a.ldaux = p->ldaux;
  a.alpha = p->alpha;
  a.aux_scale = p->aux_scale;
  a.flags = p->flags;
  a.drop_thresh = p->drop.p > 0.f ? lako_drop_thresh(p->drop.p) : 0u;
  a.drop_scale = p->drop.p > 0.f ? 1.0f / (1.0f - p->drop.p) : 1.0f;
  a.drop_key = lako_drop_key(p->drop.seed, p->drop.site);
  a.tiles_m = cdiv(a.M, 256);
  a.tiles_n = cdiv(a.N, 256);
  const lako_tuning_t& tu = p->tuning ? *p->tuning : process_tuning();
  a.group_m = a.tiles_n >= 16 ? (tu.nt_group_m < 0 ? -tu.nt_group_m : tu.nt_group_m) : (tu.nt_group_m < 0 ? -tu.nt_group_m : 0);
  a.wide_epi = !(a.flags & (LAKO_EPI_RESID | LAKO_EPI_AUXMASK));
  m.sa = a_scales;
  m.sb = b_scales;
  m.ksp = (int)(((p->K / 128) + 3) / 4 * 4);
  // round 6: the four-wave tile with two K-slices in flight (gemm_nt4_mx.h) where it applies — whole groups of four K-steps, at least one
  // round of the chip, the epilogues nt4_epilogue has (nt4_ok's rules), 32-bit offsets; the tile height by the rounds of the chip as for bf16
  {
    const int side = a.flags & (LAKO_EPI_RESID | LAKO_EPI_AUXMASK);
    bool ok = tu.nt_four && a.K % 512 == 0 && (int64_t)cdiv(a.M, 256) * cdiv(a.N, 256) >= 256 && (int64_t)256 * std::max(a.lda, a.ldb) < (1ll << 31) &&
              (int64_t)128 * a.ldc * 2 < (1ll << 31) && !(a.alpha != 1.0f && (a.flags != 0 || a.drop_thresh != 0)) &&
              !(a.drop_thresh != 0 && (uint64_t)(a.M + 256) * (uint64_t)a.N >= (1ull << 34)) && side != (LAKO_EPI_RESID | LAKO_EPI_AUXMASK) &&
              !(side && (a.flags & LAKO_EPI_RELU)) && !(side == LAKO_EPI_AUXMASK && a.drop_thresh != 0);
    if (side == LAKO_EPI_RESID) ok = ok && a.ldr % 8 == 0 && reinterpret_cast<uintptr_t>(a.resid) % 16 == 0 && (int64_t)128 * a.ldr * 2 < (1ll << 31);
    if (side == LAKO_EPI_AUXMASK) ok = ok && a.ldaux % 8 == 0 && reinterpret_cast<uintptr_t>(a.aux) % 16 == 0 && (int64_t)128 * a.ldaux * 2 < (1ll << 31);
    if (ok) {
      const int nkx = a.K / 128, tn = cdiv(a.N, 256);
      const double c8 = (double)cdiv((int64_t)cdiv(a.M, 256) * tn, 256) * (4.5 + 1.6 * nkx);
      const double c6 = (double)cdiv((int64_t)cdiv(a.M, 192) * tn, 256) * (4.5 + 1.25 * nkx);
      const bool six = tu.nt_variant == 3 || (tu.nt_variant != 9 && c6 < c8);
      a.tiles_m = cdiv(a.M, six ? 192 : 256);
      a.group_m = tu.nt_group_m < 0 ? -tu.nt_group_m : (a.tiles_n >= 8 ? tu.nt_group_m : 0);
      a.dephase = 0;
      a.epi_direct = 0;
      int g4 = a.tiles_m * a.tiles_n;
      if (g4 > 256) g4 = 256;
#define LAKO_MX4(MT_, SIDE_)                                                                                         \
  do {                                                                                                               \
    LAKO_SET_MAX_LDS((&gemm_nt4_mx_kernel<MT_, SIDE_>), nt4::LDS_BYTES);                                             \
    hipLaunchKernelGGL((gemm_nt4_mx_kernel<MT_, SIDE_>), dim3(g4), dim3(256), nt4::LDS_BYTES, (hipStream_t)stream, m); \
  } while (0)
      if (six) { if (side) LAKO_MX4(6, true); else LAKO_MX4(6, false); }
      else { if (side) LAKO_MX4(8, true); else LAKO_MX4(8, false); }
#undef LAKO_MX4
      LAKO_LAUNCH_CHECK();
      return LAKO_OK;
    }
  }
  int grid = a.tiles_m * a.tiles_n;
  if (grid > 256) grid = 256;
  a.dephase = (tu.nt_dephase > 0 && a.tiles_m * a.tiles_n > grid) ? ((tu.nt_dephase_n << 16) | (tu.nt_dephase & 0xffff)) : 0;
  constexpr int LDS = 2 * (256 + 256) * TKB;
  LAKO_SET_MAX_LDS((&gemm_nt_mx_kernel), LDS);
  hipLaunchKernelGGL(gemm_nt_mx_kernel, dim3(grid), dim3(512), LDS, (hipStream_t)stream, m);
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

// Kernel-selection knobs for A/B measurements and tests (see the table above): every key the release build accepts selects between
// kernels / tilings that compute the SAME result.  Both functions write only the caller's struct.
extern "C" int lako_tuning_init(lako_tuning_t* t) {
  LAKO_CHECK_ARG(t != nullptr, "lako_tuning_init: null struct");
  return tuning_init(t);
}

extern "C" int lako_tuning_set(lako_tuning_t* t, const char* key, int value) { return tuning_set(t, key, value); }

#ifdef LAKO_EXPERIMENTS
// (experiments build only) the stamps of gemm_nt_debug bit 6, for tools/gemm_stamps.py
extern "C" int lako_exp_nt_stamps(void* host_dst, size_t bytes) {
  (void)hipDeviceSynchronize();
  return (int)hipMemcpy(host_dst, lako_exp_nt_stamp_buffer(), bytes < NT_STAMP_BYTES ? bytes : NT_STAMP_BYTES, hipMemcpyDeviceToHost);
}
#endif
