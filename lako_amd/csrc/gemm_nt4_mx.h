// gemm_nt4_mx_kernel: the MX block-scaled fp8 product (lako_gemm_nt_mx: e4m3 bytes + one E8M0 scale per 32 k, bf16 out) on the four-wave
// tile of gemm_nt4.h — 256 × 256 (MT = 8) or 192 × 256 (MT = 6) per workgroup, 128 × 128 / 96 × 128 accumulators per wave, the same LDS image
// (128-byte rows = ONE K-step of 128 k here, XOR-swizzled 16-byte chunks), the same two K-slices of LDS-DMA in flight behind a counted
// `s_waitcnt vmcnt`, the same epilogues (nt4_epilogue).  Included by gemm.hip after MxArgs.
//
// What differs from the bf16 loop: `v_mfma_scale_f32_16x16x128_f8f6f4` consumes BOTH 16-byte halves of a fragment (k = 16g … 16g+15 and
// 64+16g …: the chunk pair the bf16 loop reads as its two k-halves) in one instruction of 8 passes, so a K-step is 64 MFMAs per wave, each
// needing its whole fragments — the bf16 schedule's trick (read k-half 1 under the MFMAs of k-half 0) has nothing to hide behind, and a
// second set of fragment registers for the next step does not fit beside 256 accumulators (2 × 128 + scales + addresses > 256 VGPRs).
// So a step is two phases:
//   R  `vmcnt(P)` (P = the pieces of the step before: K-slice t has landed, t + 1 may be in flight) → barrier → the 2·(MT + 8) fragment reads
//      → barrier (every wave holds its fragments: the buffer of K-slice t is free);
//   M  the 8·MT MFMAs, the MT + 8 LDS-DMA pieces of K-slice t + 2 dealt evenly between them into the buffer just freed.
// The read phase is exposed (≈ 1 000 cycles: 128 KiB per CU through the LDS port) against 2 048 cycles of MFMA per step — 0.65 where the
// eight-wave MX kernel (one slice in flight, `vmcnt(0)` + barrier per step, 3 900 cycles) reaches 0.52; the fragments are compiler-visible
// values here (it must build the 8-register operands of the MFMA from two 16-byte reads without copies), only DMA, waits and barriers are asm.
// Scales: a lane's dword holds the scale bytes of FOUR K-steps of its row and 32-k block ([rows][4][ksp] layout of lako_mx_quantize);
// the byte is picked by the instruction's op_sel bits, the dwords of the next group of four steps are requested (inline asm: a
// compiler-visible load is waited for with vmcnt(0), i.e. for every piece in flight) at the top of a group and are older than every
// piece the next counted wait lets through.  Host: K % 512 == 0 (whole groups), at least one round of the chip.
#pragma once

namespace nt4 {

// One scaled MFMA as an asm statement with the accumulator pinned to AGPRs (as plain builtins hipcc spread 256 accumulators over both register
// files and spilled 128 of them).  BY: the byte of the scale dwords (op_sel | op_sel_hi << 1 per operand).  The matrix operands are swapped
// as in the bf16 kernels: the first one — B's fragment, scaled by B's byte — makes the accumulator rows.
template <int BY, bool FIRST>
__device__ __forceinline__ void mx_mfma(f32x4& c, v8i_t bfr, v8i_t afr, uint32_t sb, uint32_t sa) {
  if constexpr (FIRST) {
    if constexpr (BY == 0) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, 0, %3, %4 op_sel_hi:[0,0,0]" : "=a"(c) : "v"(bfr), "v"(afr), "v"(sb), "v"(sa));
    else static_assert(BY == 0, "a tile starts with the first byte of a scale group");
  } else {
    if constexpr (BY == 0) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+a"(c) : "v"(bfr), "v"(afr), "v"(sb), "v"(sa));
    else if constexpr (BY == 1) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[1,1,0] op_sel_hi:[0,0,0]" : "+a"(c) : "v"(bfr), "v"(afr), "v"(sb), "v"(sa));
    else if constexpr (BY == 2) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+a"(c) : "v"(bfr), "v"(afr), "v"(sb), "v"(sa));
    else asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+a"(c) : "v"(bfr), "v"(afr), "v"(sb), "v"(sa));
  }
}

__device__ __forceinline__ void mx_dma(uint32_t lds_dst, uint32_t voff, uint64_t sbase) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
}

}  // namespace nt4

template <int MT, bool SIDE>
__global__ __launch_bounds__(256) void gemm_nt4_mx_kernel(MxArgs ma) {
  using namespace nt4;
  const NtArgs& a = ma.nt;
  constexpr int BM = MT * 32, NP = MT + 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int r16 = lane & 15, g = lane >> 4;
  const int nwg = a.tiles_m * a.tiles_n;
  const int64_t lda_b = a.lda, ldb_b = a.ldb;      // bytes = elements
  const int nk = a.K / 128;                         // K-steps (a multiple of 4: host)
  const uint32_t srow = 4u * (uint32_t)ma.ksp;      // scale bytes per operand row
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);

  // per-lane LDS offsets of the fragment reads (row r16 of tile 0, chunk g; the second half is ^ 64; tile i is + i · 2048)
  const uint32_t key16 = (uint32_t)((g ^ ((r16 >> 1) & 7)) << 4);
  const uint32_t fa = (uint32_t)((wr * MT * 16 + r16) * TKB) + key16, fb = IMG + (uint32_t)((wc * 128 + r16) * TKB) + key16;
  const uint32_t ldsA0 = lds0 + wave * 1024, ldsB0 = ldsA0 + IMG;

  int tile = xcd_remap(blockIdx.x, gridDim.x);
  int tm_, tn_;
  tile_coords(tile, a.tiles_m, a.tiles_n, a.group_m, tm_, tn_);
  int m0 = tm_ * BM, n0 = tn_ * BN;
  uint32_t voffA[MT], voffB[8];
  piece_offsets<MT>(voffA, min(BM, a.M - m0), lda_b, wave, lane);
  piece_offsets<8>(voffB, min(BN, a.N - n0), ldb_b, wave, lane);
  uint64_t pA = reinterpret_cast<uint64_t>(a.A + (int64_t)m0 * lda_b), pB = reinterpret_cast<uint64_t>(a.B + (int64_t)n0 * ldb_b);
  // prologue: K-slices 0 and 1 of the first tile
#pragma unroll
  for (int x = 0; x < 2; ++x) {
#pragma unroll
    for (int j = 0; j < MT; ++j) mx_dma(ldsA0 + x * BUF + j * 4096, voffA[j], pA + x * TKB);
#pragma unroll
    for (int j = 0; j < 8; ++j) mx_dma(ldsB0 + x * BUF + j * 4096, voffB[j], pB + x * TKB);
  }
  pA += 2 * TKB;
  pB += 2 * TKB;
  // the lane's scale dwords of a group of four K-steps: rows clamped at the edge (those operand rows feed accumulators that are never stored)
  uint32_t sca[MT], scb[8], sca_n[MT], scb_n[8];
  // (32-bit offsets from an opaque lane id against scalar bases: as 64-bit per-lane pointers the sixteen addresses are loop invariants that
  //  hipcc hoists — 32 registers — and spills)
  auto request_scales = [&](int tm0, int tn0, int grp) {
    const int le = fresh_lane();
    const int r16e = le & 15;
    const uint32_t col = (uint32_t)((le >> 4) * ma.ksp + 4 * grp);
    const uint64_t sa_base = reinterpret_cast<uint64_t>(ma.sa), sb_base = reinterpret_cast<uint64_t>(ma.sb);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const uint32_t off = (uint32_t)min(tm0 + (wr * MT + i) * 16 + r16e, a.M - 1) * srow + col;
      asm volatile("global_load_dword %0, %1, %2" : "=v"(sca_n[i]) : "v"(off), "s"(sa_base) : "memory");
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t off = (uint32_t)min(tn0 + (wc * 8 + j) * 16 + r16e, a.N - 1) * srow + col;
      asm volatile("global_load_dword %0, %1, %2" : "=v"(scb_n[j]) : "v"(off), "s"(sb_base) : "memory");
    }
  };
  request_scales(m0, n0, 0);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // (the first tile's K-slices 0 / 1 and scale dwords: the only full wait of the kernel)

  while (true) {
    const int next_tile = tile + gridDim.x;
    const bool has_next = next_tile < nwg;
    tile_coords(has_next ? next_tile : tile, a.tiles_m, a.tiles_n, a.group_m, tm_, tn_);
    const int nm0 = tm_ * BM, nn0 = tn_ * BN;
    f32x4 acc[8][MT];      // [n-tile][m-tile], as in gemm_nt4_kernel
    v8i_t af[MT], bf[8];

    // one K-step from buffer X with scale byte BY of the current group; `first`: the accumulators start at 0
    auto step = [&](auto Xc, auto BYc, auto FIRSTc, auto&& after_wait) {
      constexpr int X = decltype(Xc)::value, BY = decltype(BYc)::value;
      constexpr bool FIRST = decltype(FIRSTc)::value;
      // ---- R: K-slice t has landed — the NP pieces of K-slice t + 1, requested one step ago, are the wave's newest operations and may stay
      // in flight (whatever else the wave issued goes out AFTER this wait — `after_wait` — and is older than the next step's newest NP)
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" : : "n"(NP) : "memory");
      after_wait();
      const char* As = smem + X * BUF;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const u32x4 lo = *reinterpret_cast<const u32x4*>(As + fb + j * 2048), hi = *reinterpret_cast<const u32x4*>(As + (fb ^ 64u) + j * 2048);
        bf[j] = v8i_t{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
      }
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const u32x4 lo = *reinterpret_cast<const u32x4*>(As + fa + i * 2048), hi = *reinterpret_cast<const u32x4*>(As + (fa ^ 64u) + i * 2048);
        af[i] = v8i_t{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave holds its fragments: buffer X is free for K-slice t + 2
      // ---- M: the MFMAs, the DMA pieces of K-slice t + 2 between them (B's eight pieces first, then A's)
#pragma unroll
      for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          mx_mfma<BY, FIRST>(acc[j][i], bf[j], af[i], scb[j], sca[i]);
          // the NP pieces spread evenly over the 8·MT MFMAs: piece pz goes out behind the MFMA at which (s + 1)·NP / (8·MT) reaches pz + 1
          const int s = i * 8 + j;
          const int pz = (s * NP) / (8 * MT);
          if (((s + 1) * NP) / (8 * MT) > pz) {
            if (pz < 8) mx_dma(ldsB0 + X * BUF + pz * 4096, voffB[pz], pB + X * TKB);
            else mx_dma(ldsA0 + X * BUF + (pz - 8) * 4096, voffA[pz - 8], pA + X * TKB);
          }
        }
      }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    using T_ = std::true_type;
    using F_ = std::false_type;
    auto next_tile_stream = [&]() {      // the stream moves on to the workgroup's next tile (the last tile requests its own first slices again)
      pA = reinterpret_cast<uint64_t>(a.A + (int64_t)nm0 * lda_b);
      pB = reinterpret_cast<uint64_t>(a.B + (int64_t)nn0 * ldb_b);
      piece_offsets<MT>(voffA, min(BM, a.M - nm0), lda_b, wave, fresh_lane());
      piece_offsets<8>(voffB, min(BN, a.N - nn0), ldb_b, wave, fresh_lane());
    };
    const int ngrp = nk >> 2;
    auto nothing = []() {};
    for (int gi = 0; gi < ngrp; ++gi) {
      // this group's scale dwords: requested a group ago, behind the counted wait of that group's first step, i.e. older than the NP newest
      // operations of its second step's wait — landed three steps ago
#pragma unroll
      for (int i = 0; i < MT; ++i) sca[i] = sca_n[i];
#pragma unroll
      for (int j = 0; j < 8; ++j) scb[j] = scb_n[j];
      const bool last = gi + 1 == ngrp;
      auto next_scales = [&]() {
        if (!last) request_scales(m0, n0, gi + 1);
        else request_scales(nm0, nn0, 0);
      };
      if (gi == 0) step(I0{}, I0{}, T_{}, next_scales);
      else step(I0{}, I0{}, F_{}, next_scales);
      step(I1{}, I1{}, F_{}, nothing);
      if (last) {
        next_tile_stream();      // K-slices nk and nk + 1 are the next tile's 0 and 1
      } else {
        pA += 2 * TKB;
        pB += 2 * TKB;
      }
      step(I0{}, I2{}, F_{}, nothing);
      step(I1{}, I3{}, F_{}, nothing);
      pA += 2 * TKB;      // (after a tile's last pair: the next tile's K-slice 2)
      pB += 2 * TKB;
    }
    // (the epilogue's first accumulator reads are compiler-generated: its hazard recogniser does not see the MFMAs inside the statements above)
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    nt4_epilogue<MT, SIDE, 0>(a, acc, smem, wave, wr, wc, m0, n0);
    if (!has_next) break;
    tile = next_tile;
    m0 = nm0;
    n0 = nn0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the last tile's surplus pieces and scale requests must have landed before the LDS is released
}
