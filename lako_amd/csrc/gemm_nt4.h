// gemm_nt4_kernel: the 256 × 256 NT tile on FOUR waves of 128 × 128 (one wave per SIMD, 256 accumulator registers in AGPRs) with a
// HAND-PLACED K loop (round 6).  Included by gemm.hip inside its anonymous namespace (NtArgs, xcd_remap, tile_coords, lako_keep4 …).
//
// Why: the eight-wave kernel's K-step costs ≈ 3 900 cycles for 2 048 MFMA cycles per SIMD (profiles/r05c_gemm_stamps.txt) — every wave
// waits `vmcnt(0)` for a K-slice that was requested less than one K-step earlier, and hipcc's schedule issues the fragment reads of a
// K-half as a block.  The schedule below is the one the disassembly of the vendor library's MT256x256x64 kernel shows (its structure, not
// its code: profiles/r06a_vendor_loop_schedule.txt):
//   * TWO K-steps of LDS-DMA in flight with TWO LDS buffers: K-slice t + 2 goes into the buffer of K-slice t, operand by operand, as
//     soon as every wave holds that operand's fragments of step t in registers (barrier 1: B, barrier 2: A);
//   * ONE counted wait per K-step, `vmcnt(13)`, at MFMA 91 of 128: thirteen of the step's own sixteen pieces may be outstanding,
//     everything older — K-slice t + 1, requested a full K-step earlier — has landed; barrier 3 makes that true for all waves, and the
//     k-half-0 fragments of step t + 1 are read under the last 35 MFMAs;
//   * every LDS read / DMA piece / wait / barrier sits at a FIXED MFMA slot (one inline-asm statement per MFMA slot; hipcc's
//     scheduler cannot move volatile asm statements past each other), never more than one memory instruction between two MFMAs;
//   * across tiles the stream never stops: the last two K-steps of a tile request the first two K-slices of the workgroup's next tile,
//     which land under the epilogue (whose transposition scratch is the 32 KiB behind the two 64-KiB buffers).
// LDS image, fragment layout, accumulator layout and the epilogue arithmetic are the eight-wave kernel's (bit-identical results).
#pragma once

namespace nt4 {

constexpr int BM = 256, BN = 256, IMG = 256 * TKB, BUF = 2 * IMG, SCRATCH = 2 * BUF, LDS_BYTES = 2 * BUF + 4 * 8192;
static_assert(LDS_BYTES == 160 * 1024, "LDS");

// side operation of an MFMA slot
enum : int { OP_NONE = 0, OP_RD_B1, OP_RD_A1, OP_RD_B0N, OP_RD_A0N, OP_DMA_B, OP_DMA_A, OP_WAIT_L, OP_BAR, OP_WAIT_V };
struct Slot { int op, idx; };

// the schedule of one K-step: slot s = 8·a + b of part 1 (k-half 0) is MFMA (m-tile a, n-tile b), part 2 (k-half 1) the same 64 later
constexpr Slot slot_of(int s) {
  // k-half-1 B fragments (all eight are needed at slot 64): slots 0 … 14
  if (s < 16 && s % 2 == 0) return {OP_RD_B1, s / 2};
  if (s == 20) return {OP_WAIT_L, 0};
  if (s == 21) return {OP_BAR, 0};          // barrier 1: every wave holds B of this step → its image may be overwritten
  // B pieces 0 … 4 and the k-half-1 A fragments, alternating
  if (s == 22) return {OP_DMA_B, 0};
  if (s == 25) return {OP_DMA_B, 1};
  if (s == 28) return {OP_DMA_B, 2};
  if (s == 31) return {OP_DMA_B, 3};
  if (s == 34) return {OP_DMA_B, 4};
  if (s == 24) return {OP_RD_A1, 0};
  if (s == 27) return {OP_RD_A1, 1};
  if (s == 30) return {OP_RD_A1, 2};
  if (s == 33) return {OP_RD_A1, 3};
  if (s == 36) return {OP_RD_A1, 4};
  if (s == 37) return {OP_RD_A1, 5};
  if (s == 39) return {OP_RD_A1, 6};
  if (s == 41) return {OP_RD_A1, 7};
  if (s == 50) return {OP_WAIT_L, 0};
  if (s == 51) return {OP_BAR, 0};          // barrier 2: every wave holds A of this step
  if (s == 52) return {OP_DMA_B, 5};
  if (s == 55) return {OP_DMA_B, 6};
  if (s == 58) return {OP_DMA_B, 7};
  if (s == 61) return {OP_DMA_A, 0};
  if (s == 64) return {OP_DMA_A, 1};
  if (s == 85) return {OP_DMA_A, 2};
  if (s == 87) return {OP_DMA_A, 3};
  if (s == 89) return {OP_DMA_A, 4};
  if (s == 91) return {OP_WAIT_V, 13};      // 8 + 5 pieces of this step issued: everything older (K-slice t + 1) has landed
  if (s == 92) return {OP_BAR, 0};          // barrier 3: … for every wave
  if (s == 93) return {OP_RD_B0N, 0};
  if (s == 94) return {OP_RD_B0N, 1};
  if (s == 95) return {OP_RD_B0N, 2};
  if (s == 97) return {OP_RD_B0N, 3};
  if (s == 98) return {OP_RD_B0N, 4};
  if (s == 101) return {OP_RD_B0N, 5};
  if (s == 103) return {OP_RD_B0N, 6};
  if (s == 104) return {OP_RD_B0N, 7};
  if (s == 96) return {OP_DMA_A, 5};
  if (s == 100) return {OP_DMA_A, 6};
  if (s == 105) return {OP_RD_A0N, 0};
  if (s == 106) return {OP_RD_A0N, 1};
  if (s == 109) return {OP_RD_A0N, 2};
  if (s == 112) return {OP_RD_A0N, 3};
  if (s == 114) return {OP_RD_A0N, 4};
  if (s == 117) return {OP_RD_A0N, 5};
  if (s == 120) return {OP_RD_A0N, 6};
  if (s == 124) return {OP_RD_A0N, 7};
  if (s == 125) return {OP_DMA_A, 7};
  if (s == 126) return {OP_WAIT_L, 0};
  return {OP_NONE, 0};
}

struct Addr {            // per-lane LDS byte addresses of the fragment reads (m-/n-tile 0; tile i is + i · 2048)
  uint32_t a[2][2];      // [buffer][k-half]
  uint32_t b[2][2];
};

// one LDS-DMA piece outside the K loop (prologue)
__device__ __forceinline__ void dma_piece(uint32_t lds_dst, uint32_t voff, uint64_t sbase) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
}

// One K-step from buffer X (compile-time): acc += A(step)·B(step)ᵀ; requests K-slice t + 2 (pA / pB: its first byte of row 0 of the
// tile, voffA / voffB: the per-lane row · ld + swizzled chunk of piece j) into buffer X; leaves the k-half-0 fragments of step t + 1
// (buffer X ^ 1) in A0 / B0.
template <int X, int s>
__device__ __forceinline__ void step_slot(f32x4 (&acc)[8][8], u32x4 (&A0)[8], u32x4 (&B0)[8], u32x4 (&A1)[8], u32x4 (&B1)[8], const Addr& ad,
                                          uint32_t ldsA, uint32_t ldsB, uint64_t pA, uint64_t pB, const uint32_t (&voffA)[8],
                                          const uint32_t (&voffB)[8]) {
  if constexpr (s < 128) {
    constexpr int part = s / 64, ma = (s % 64) / 8, nb = s % 8;
    constexpr Slot sl = slot_of(s);
    u32x4& af = part ? A1[ma] : A0[ma];
    u32x4& bf = part ? B1[nb] : B0[nb];
    f32x4& c = acc[nb][ma];
    if constexpr (s == 127) {
      // the step's last MFMA: whatever hipcc places behind the asm statement (at the loop exit it copies accumulator registers, and its
      // hazard recogniser does not see the MFMA inside the statement) must find the result written — the wait states sit under the
      // MFMA's own 16 pipe cycles, the next step's first MFMA could not issue earlier anyway
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 3" : "+a"(c) : "v"(bf), "v"(af));
    } else if constexpr (sl.op == OP_NONE) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(bf), "v"(af));
    } else if constexpr (sl.op == OP_RD_B1) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\tds_read_b128 %1, %4 offset:%5"
                   : "+a"(c), "=&v"(B1[sl.idx]) : "v"(bf), "v"(af), "v"(ad.b[X][1]), "n"(sl.idx * 2048) : "memory");
    } else if constexpr (sl.op == OP_RD_A1) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\tds_read_b128 %1, %4 offset:%5"
                   : "+a"(c), "=&v"(A1[sl.idx]) : "v"(bf), "v"(af), "v"(ad.a[X][1]), "n"(sl.idx * 2048) : "memory");
    } else if constexpr (sl.op == OP_RD_B0N) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\tds_read_b128 %1, %4 offset:%5"
                   : "+a"(c), "=&v"(B0[sl.idx]) : "v"(bf), "v"(af), "v"(ad.b[X ^ 1][0]), "n"(sl.idx * 2048) : "memory");
    } else if constexpr (sl.op == OP_RD_A0N) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\tds_read_b128 %1, %4 offset:%5"
                   : "+a"(c), "=&v"(A0[sl.idx]) : "v"(bf), "v"(af), "v"(ad.a[X ^ 1][0]), "n"(sl.idx * 2048) : "memory");
    } else if constexpr (sl.op == OP_DMA_B) {
      asm volatile("s_add_u32 m0, %3, %4\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\tglobal_load_lds_dwordx4 %5, %6"
                   : "+a"(c) : "v"(bf), "v"(af), "s"(ldsB), "n"(sl.idx * 4096), "v"(voffB[sl.idx]), "s"(pB) : "memory");
    } else if constexpr (sl.op == OP_DMA_A) {
      asm volatile("s_add_u32 m0, %3, %4\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\tglobal_load_lds_dwordx4 %5, %6"
                   : "+a"(c) : "v"(bf), "v"(af), "s"(ldsA), "n"(sl.idx * 4096), "v"(voffA[sl.idx]), "s"(pA) : "memory");
    } else if constexpr (sl.op == OP_WAIT_L) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_waitcnt lgkmcnt(0)" : "+a"(c) : "v"(bf), "v"(af) : "memory");
    } else if constexpr (sl.op == OP_BAR) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_barrier" : "+a"(c) : "v"(bf), "v"(af) : "memory");
    } else if constexpr (sl.op == OP_WAIT_V) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_waitcnt vmcnt(%3)" : "+a"(c) : "v"(bf), "v"(af), "n"(sl.idx) : "memory");
    }
    step_slot<X, s + 1>(acc, A0, B0, A1, B1, ad, ldsA, ldsB, pA, pB, voffA, voffB);
  }
}
template <int X>
__device__ __forceinline__ void step(f32x4 (&acc)[8][8], u32x4 (&A0)[8], u32x4 (&B0)[8], u32x4 (&A1)[8], u32x4 (&B1)[8], const Addr& ad,
                                     uint32_t ldsA, uint32_t ldsB, uint64_t pA, uint64_t pB, const uint32_t (&voffA)[8],
                                     const uint32_t (&voffB)[8]) {
  step_slot<X, 0>(acc, A0, B0, A1, B1, ad, ldsA, ldsB, pA, pB, voffA, voffB);
}
template <int I>
__device__ __forceinline__ void read8(u32x4 (&dst)[8], uint32_t addr) {
  if constexpr (I < 8) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[I]) : "v"(addr), "n"(I * 2048) : "memory");
    read8<I + 1>(dst, addr);
  }
}

// per-lane source offsets of a tile's eight pieces per operand (piece j of a wave = image rows 32 j + 8 wave … + 7; rows past the
// edge clamped to the last valid row: they feed accumulator rows that are never stored)
__device__ __forceinline__ void piece_offsets(uint32_t (&voff)[8], int rows_valid, int64_t ld_bytes, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int row = (wave + 4 * j) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    voff[j] = (uint32_t)(min(row, rows_valid - 1) * ld_bytes + c * 16);
  }
}

}  // namespace nt4

__global__ __launch_bounds__(256) void gemm_nt4_kernel(NtArgs a) {
  using namespace nt4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int r16 = lane & 15, g = lane >> 4;
  const int nwg = a.tiles_m * a.tiles_n;
  const int64_t lda_b = a.lda * 2, ldb_b = a.ldb * 2;
  const int nk = a.K / 64;                          // whole 128-byte K-slices, an even number of them (host)
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);

  Addr ad;
  {
    const uint32_t key16 = (uint32_t)((g ^ ((r16 >> 1) & 7)) << 4);
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      ad.a[x][0] = lds0 + x * BUF + (uint32_t)((wr * 128 + r16) * TKB) + key16;
      ad.a[x][1] = ad.a[x][0] ^ 64u;
      ad.b[x][0] = lds0 + x * BUF + IMG + (uint32_t)((wc * 128 + r16) * TKB) + key16;
      ad.b[x][1] = ad.b[x][0] ^ 64u;
    }
  }
  const uint32_t ldsA0 = lds0 + wave * 1024, ldsB0 = ldsA0 + IMG;      // DMA destinations of the wave's piece 0, buffer 0

  int tile = xcd_remap(blockIdx.x, gridDim.x);
  int tm_, tn_;
  tile_coords(tile, a.tiles_m, a.tiles_n, a.group_m, tm_, tn_);
  int m0 = tm_ * BM, n0 = tn_ * BN;
  if (a.dephase > 0) {
    const int ph = (blockIdx.x >> 3) % (a.dephase >> 16);
    if (ph) {
      const uint64_t t0 = __builtin_amdgcn_s_memrealtime(), dt = (uint64_t)((a.dephase & 0xffff) * ph);
      while (__builtin_amdgcn_s_memrealtime() - t0 < dt) __builtin_amdgcn_s_sleep(8);
    }
  }
  uint32_t voffA[8], voffB[8];
  piece_offsets(voffA, min(BM, a.M - m0), lda_b, wave, lane);
  piece_offsets(voffB, min(BN, a.N - n0), ldb_b, wave, lane);
  // prologue: K-slices 0 and 1 of the first tile, then the k-half-0 fragments of step 0
  {
    const uint64_t pA = reinterpret_cast<uint64_t>(a.A + (int64_t)m0 * lda_b), pB = reinterpret_cast<uint64_t>(a.B + (int64_t)n0 * ldb_b);
#pragma unroll
    for (int x = 0; x < 2; ++x) {
#pragma unroll
      for (int j = 0; j < 8; ++j) dma_piece(ldsA0 + x * BUF + j * 4096, voffA[j], pA + x * TKB);
#pragma unroll
      for (int j = 0; j < 8; ++j) dma_piece(ldsB0 + x * BUF + j * 4096, voffB[j], pB + x * TKB);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  u32x4 A0[8], B0[8], A1[8], B1[8];
  read8<0>(B0, ad.b[0][0]);
  read8<0>(A0, ad.a[0][0]);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  while (true) {
    const int next_tile = tile + gridDim.x;
    const bool has_next = next_tile < nwg;
    tile_coords(has_next ? next_tile : tile, a.tiles_m, a.tiles_n, a.group_m, tm_, tn_);
    const int nm0 = tm_ * BM, nn0 = tn_ * BN;
    const char* Abase = a.A + (int64_t)m0 * lda_b;
    const char* Bbase = a.B + (int64_t)n0 * ldb_b;

    f32x4 acc[8][8];      // [n-tile][m-tile]; element r of lane (l & 15, g): C[m = mt·16 + (l & 15)][n = nt·16 + 4g + r]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint64_t pA = reinterpret_cast<uint64_t>(Abase) + 2 * TKB, pB = reinterpret_cast<uint64_t>(Bbase) + 2 * TKB;
    for (int t = 0; t < nk; t += 2) {
      if (t + 2 == nk) {
        // the stream moves on to the workgroup's next tile (its K-slices 0 and 1); the last tile requests its own first slices again
        // (the pieces are always issued: the counted wait depends on it) — they land in buffers nobody reads any more
        pA = reinterpret_cast<uint64_t>(a.A + (int64_t)nm0 * lda_b);
        pB = reinterpret_cast<uint64_t>(a.B + (int64_t)nn0 * ldb_b);
        piece_offsets(voffA, min(BM, a.M - nm0), lda_b, wave, fresh_lane());
        piece_offsets(voffB, min(BN, a.N - nn0), ldb_b, wave, fresh_lane());
      }
      step<0>(acc, A0, B0, A1, B1, ad, ldsA0, ldsB0, pA, pB, voffA, voffB);
      step<1>(acc, A0, B0, A1, B1, ad, ldsA0 + BUF, ldsB0 + BUF, pA + TKB, pB + TKB, voffA, voffB);
      pA += 2 * TKB;
      pB += 2 * TKB;
    }

    // ---- epilogue: alpha, ReLU, dropout in the accumulator layout; the wave's 128 × 128 region leaves through 8 KiB of private LDS
    // scratch, 16 rows (4 KiB as bf16) at a time, as 16-byte row-major stores (4 rows × 256 B per wave-instruction)
    {
      // (every lane-derived value of the epilogue is rebuilt from an opaque lane id: as invariants of the tile loop they would be hoisted
      //  above the K loop and spilled)
      bf16_t* C = reinterpret_cast<bf16_t*>(a.C);
      const bool relu = a.flags & LAKO_EPI_RELU, drop = a.drop_thresh != 0;
      const int le = fresh_lane();
      const int r16e = le & 15, ge = le >> 4;
      char* ep = smem + SCRATCH + wave * 8192;
      const int cj = le & 15, n = n0 + wc * 128 + cj * 8;
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) {
        char* buf = ep + (mt & 1) * 4096;
        const int m_acc = m0 + wr * 128 + mt * 16 + r16e;
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
          f32x4 v = acc[nt][mt] * a.alpha;
          if (relu) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
          }
          if (drop) {
            const uint64_t idx = (uint64_t)(a.row0 + m_acc) * (uint64_t)a.N + (uint64_t)(n0 + wc * 128 + nt * 16 + 4 * ge);
            bool kp[4];
            lako_keep4(a.drop_key, idx >> 2, a.drop_thresh, kp);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = kp[r] ? v[r] * a.drop_scale : 0.f;
          }
          const bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
          const int c8 = nt * 4 + ge;          // 8-byte chunk of the 256-byte row
          *reinterpret_cast<bf16x4*>(buf + r16e * 256 + (((c8 >> 1) ^ r16e) * 16) + (c8 & 1) * 8) = o;
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int row_l = it * 4 + (le >> 4);
          const u32x4 o = *reinterpret_cast<const u32x4*>(buf + row_l * 256 + ((cj ^ row_l) * 16));
          const int m = m0 + wr * 128 + mt * 16 + row_l;
          if (m < a.M && n < a.N) *reinterpret_cast<u32x4*>(C + (int64_t)m * a.ldc + n) = o;
        }
      }
    }
    if (!has_next) break;
    tile = next_tile;
    m0 = nm0;
    n0 = nn0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the last tile's surplus pieces must have landed before the LDS is released
}
