// gemm_tn4_kernel: the 256 × 256 weight-gradient tile (C += Aᵀ·B, both operands k-strided) on FOUR waves of 128 × 128 with the hand-placed
// K loop of gemm_nt4.h (round 6): two K-steps of LDS-DMA in flight, one counted `vmcnt(13)` per step, three barriers, every memory
// operation at a fixed MFMA slot (nt4::Sched<8>).  Included by gemm.hip inside its anonymous namespace, after gemm_nt4.h.
//
// What differs from the NT kernel: the fragments are TRANSPOSED reads (two `ds_read_b64_tr_b16` per 16 × 32 fragment, the image and
// its swizzle are gemm_tn256_kernel's), issued as compiler-visible builtins between the MFMA statements — hipcc then owns the
// register halves of a fragment and the counted `lgkmcnt` in front of its first use, while the `memory` clobber of every asm
// statement pins the reads to their slots; the kernel's own `lgkmcnt(0)` in front of barriers 1 and 2 is what lets another wave's DMA
// overwrite the image.  A unit (tile × K range) is one workgroup as in gemm_tn256_kernel (uniform K-splits, or the hybrid schedule:
// whole rounds of full-K tiles + finely cut remainder); there is no stream across units: a unit is hundreds of K-steps.
#pragma once

namespace tn4 {

using nt4::OP_BAR;
using nt4::OP_DMA_A;
using nt4::OP_DMA_B;
using nt4::OP_NONE;
using nt4::OP_RD_A0N;
using nt4::OP_RD_A1;
using nt4::OP_RD_B0N;
using nt4::OP_RD_B1;
using nt4::OP_WAIT_L;
using nt4::OP_WAIT_V;

constexpr int IMG = TN2_IMG, BUF = 2 * IMG, LDS_BYTES = 2 * BUF;

struct Addr {                // per-lane LDS pointers of the transposed fragment reads: [buffer][16-column tile of the wave]; k-half kk and
  const char* a[2][8];       // half-fragment blk are + kk · 16 384 + blk · 2 048 (immediates)
  const char* b[2][8];
};

__device__ __forceinline__ u32x4 read_frag(const char* p, int kk) {
  u32x4 out;
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + kk * 16384 + blk * 2048));
    const u32x2 u = __builtin_bit_cast(u32x2, t);
    out[2 * blk] = u[0];
    out[2 * blk + 1] = u[1];
  }
  return out;
}

__device__ __forceinline__ uint64_t uniform64(uint64_t v) {      // a wave-uniform value the compiler could not prove uniform (loaded through an item table)
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;      // (the builtin returns a SIGNED int: or-ed into 64 bits directly it would smear bit 31 over the high word)
}
__device__ __forceinline__ void dma_piece(uint32_t lds_dst, uint32_t voff, uint64_t sbase) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
}

// one K-step from buffer X; see nt4::step_slot.  srcA of the MFMA is the A fragment (m), srcB the B fragment (n):
// element r of lane (l & 15, g) of acc[mt][nt] is C[m = mt·16 + 4g + r][n = nt·16 + (l & 15)]
template <int X, int s, bool FIRST = false>
__device__ __forceinline__ void step_slot(f32x4 (&acc)[8][8], u32x4 (&A0)[8], u32x4 (&B0)[8], u32x4 (&A1)[8], u32x4 (&B1)[8], const Addr& ad,
                                          uint32_t ldsA, uint32_t ldsB, uint64_t pA, uint64_t pB, const uint32_t (&voffA)[8],
                                          const uint32_t (&voffB)[8]) {
  if constexpr (s < 128) {
    constexpr int part = s / 64, ma = (s % 64) / 8, nb = s % 8;
    constexpr nt4::Slot sl = nt4::slot_of<8>(s);
    u32x4& af = part ? A1[ma] : A0[ma];
    u32x4& bf = part ? B1[nb] : B0[nb];
    f32x4& c = acc[ma][nb];
    // the MFMA (+ the slot's scalar side operation)
    if constexpr (FIRST && s < 64) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(af), "v"(bf) : "memory");
      if constexpr (sl.op == OP_WAIT_L) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if constexpr (sl.op == OP_BAR) asm volatile("s_barrier" ::: "memory");
      if constexpr (sl.op == OP_DMA_B)
        asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3" : : "s"(ldsB), "n"(sl.idx * 4096), "v"(voffB[sl.idx]), "s"(pB) : "memory", "scc");
      if constexpr (sl.op == OP_DMA_A)
        asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3" : : "s"(ldsA), "n"(sl.idx * 4096), "v"(voffA[sl.idx]), "s"(pA) : "memory", "scc");
    } else if constexpr (s == 127) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 3" : "+a"(c) : "v"(af), "v"(bf) : "memory");      // (see nt4::step_slot)
    } else if constexpr (sl.op == OP_DMA_B) {
      asm volatile("s_add_u32 m0, %3, %4\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\tglobal_load_lds_dwordx4 %5, %6"
                   : "+a"(c) : "v"(af), "v"(bf), "s"(ldsB), "n"(sl.idx * 4096), "v"(voffB[sl.idx]), "s"(pB) : "memory", "scc");
    } else if constexpr (sl.op == OP_DMA_A) {
      asm volatile("s_add_u32 m0, %3, %4\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\tglobal_load_lds_dwordx4 %5, %6"
                   : "+a"(c) : "v"(af), "v"(bf), "s"(ldsA), "n"(sl.idx * 4096), "v"(voffA[sl.idx]), "s"(pA) : "memory", "scc");
    } else if constexpr (sl.op == OP_WAIT_L) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_waitcnt lgkmcnt(0)" : "+a"(c) : "v"(af), "v"(bf) : "memory");
    } else if constexpr (sl.op == OP_BAR) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_barrier" : "+a"(c) : "v"(af), "v"(bf) : "memory");
    } else if constexpr (sl.op == OP_WAIT_V) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_waitcnt vmcnt(%3)" : "+a"(c) : "v"(af), "v"(bf), "n"(sl.idx) : "memory");
    } else {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(af), "v"(bf) : "memory");
    }
    // the slot's fragment read (behind the MFMA)
    if constexpr (sl.op == OP_RD_B1) B1[sl.idx] = read_frag(ad.b[X][sl.idx], 1);
    if constexpr (sl.op == OP_RD_A1) A1[sl.idx] = read_frag(ad.a[X][sl.idx], 1);
    if constexpr (sl.op == OP_RD_B0N) B0[sl.idx] = read_frag(ad.b[X ^ 1][sl.idx], 0);
    if constexpr (sl.op == OP_RD_A0N) A0[sl.idx] = read_frag(ad.a[X ^ 1][sl.idx], 0);
    step_slot<X, s + 1, FIRST>(acc, A0, B0, A1, B1, ad, ldsA, ldsB, pA, pB, voffA, voffB);
  }
}

// per-lane source offsets of the eight pieces a wave stages of an operand's K-step: piece j = image rows 2 (wave + 4 j), + 1 (two k-rows of
// 512 bytes); columns past the edge clamped to the last valid 16 bytes (they feed outputs that are never added)
__device__ __forceinline__ void piece_offsets(uint32_t (&voff)[8], int64_t ld_bytes, int colbytes_valid, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int row = (wave + 4 * j) * 2 + (lane >> 5);
    const int cb = min(((lane & 31) ^ (tn_key(row) << 1)) * 16, colbytes_valid - 16);
    voff[j] = (uint32_t)(row * ld_bytes + cb);
  }
}
// an INCOMPLETE K-step (krows_valid < 64 k-rows: the rest must read as zeros) by the buffer form, whose range check zero-fills
__device__ __forceinline__ void stage_partial(char* img, const char* base, int krows_valid, int64_t ld_bytes, int colbytes_valid, int wave, int lane) {
  const uint32_t nrec = (krows_valid > 0 && colbytes_valid > 0) ? (uint32_t)((int64_t)(krows_valid - 1) * ld_bytes + min(colbytes_valid, TN2_ROWB)) : 0u;
  const auto rsrc = lds_dma_rsrc(base, nrec);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int inst = wave + 4 * j;
    const int row = inst * 2 + (lane >> 5);
    const int cb = ((lane & 31) ^ (tn_key(row) << 1)) * 16;
    const bool ok = (row < krows_valid) && (cb < colbytes_valid);
    lds_dma16(img + inst * 1024, rsrc, ok ? (uint32_t)(row * ld_bytes + cb) : 0xFFFFFFF0u);
  }
}

}  // namespace tn4

__global__ __launch_bounds__(256) void gemm_tn4_kernel(TnArgs a) {
  using namespace tn4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  // ---- the unit (tile, K range) of this workgroup: as gemm_tn256_kernel
  const int ntile = a.tiles_m * a.tiles_n;
  int split, tid, k_begin, k_end, unit_mode = a.out_mode;
  if (a.t_full > 0) {
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
    const int flat = xcd_remap(blockIdx.x, gridDim.x);
    split = flat / ntile;
    tid = flat % ntile;
    k_begin = split * a.k_chunk;
    k_end = min(a.K, k_begin + a.k_chunk);
  }
  (void)split;
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
  const int krows = k_end - k_begin;          // >= 128 (host)
  const int tail = krows & 63, nfull = krows >> 6, n = nfull + (tail ? 1 : 0);
  // loop step t: the incomplete step FIRST (zero-filled by the buffer form in the prologue; the sum does not care about the order), then
  // the whole steps 0 … nfull − 1; whole step f starts at k-row 64 f
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);
  const uint32_t ldsA0 = lds0 + wave * 1024, ldsB0 = ldsA0 + IMG;

  Addr ad;
  {
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int key = q | ((g & 1) << 2);                      // tn_key of every row this lane reads (rows 32 kk + 8 g + 4 blk + q)
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        ad.a[x][i] = smem + x * BUF + (8 * g + q) * TN2_ROWB + (((wr * 8 + i) ^ key) * 32) + p * 8;
        ad.b[x][i] = smem + x * BUF + IMG + (8 * g + q) * TN2_ROWB + (((wc * 8 + i) ^ key) * 32) + p * 8;
      }
  }
  uint32_t voffA[8], voffB[8];
  piece_offsets(voffA, lda_b, acols_b, wave, lane);
  piece_offsets(voffB, ldb_b, bcols_b, wave, lane);
  const uint64_t gA = uniform64(reinterpret_cast<uint64_t>(Abase)), gB = uniform64(reinterpret_cast<uint64_t>(Bbase));
  const uint64_t sA = uniform64((uint64_t)64 * lda_b), sB = uniform64((uint64_t)64 * ldb_b);       // bytes per whole K-step
  // prologue: loop steps 0 and 1
  if (tail) {
    stage_partial(smem, Abase + (int64_t)nfull * 64 * lda_b, tail, lda_b, acols_b, wave, lane);
    stage_partial(smem + IMG, Bbase + (int64_t)nfull * 64 * ldb_b, tail, ldb_b, bcols_b, wave, lane);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) dma_piece(ldsA0 + j * 4096, voffA[j], gA);
#pragma unroll
    for (int j = 0; j < 8; ++j) dma_piece(ldsB0 + j * 4096, voffB[j], gB);
  }
  {
    const int f1 = tail ? 0 : 1;                 // whole step that is loop step 1
#pragma unroll
    for (int j = 0; j < 8; ++j) dma_piece(ldsA0 + BUF + j * 4096, voffA[j], gA + f1 * sA);
#pragma unroll
    for (int j = 0; j < 8; ++j) dma_piece(ldsB0 + BUF + j * 4096, voffB[j], gB + f1 * sB);
  }
  // (the barrier as an asm statement WITH a memory clobber: the fragment reads below are plain LDS loads to the compiler, which may move
  //  them above `__builtin_amdgcn_s_barrier()` — it has no memory effects — and read images other waves are still filling)
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  u32x4 A0[8], B0[8], A1[8], B1[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) B0[i] = read_frag(ad.b[0][i], 0);
#pragma unroll
  for (int i = 0; i < 8; ++i) A0[i] = read_frag(ad.a[0][i], 0);

  f32x4 acc[8][8];
  // whole step of loop step t + 2 (the pieces a step requests): t + 2 − (tail ? 1 : 0); past the end the first whole step again (the
  // pieces are always issued: the counted wait depends on it) — it lands in a buffer nobody reads any more
  // (the pointers advance by additions — a 64-bit multiply would run on the vector unit and hand the asm statements a VGPR pair)
  uint64_t pA = gA + (tail ? sA : 2 * sA), pB = gB + (tail ? sB : 2 * sB);      // first byte of loop step 2's K-slice
  auto pieces_for = [&](int t2, uint64_t& p, uint64_t g0, uint64_t st) {       // source of loop step t2's pieces, then advance
    const uint64_t r = uniform64(t2 < n ? p : g0);
    p += st;
    return r;
  };
  {
    const uint64_t qa = pieces_for(2, pA, gA, sA), qb = pieces_for(2, pB, gB, sB);
    step_slot<0, 0, true>(acc, A0, B0, A1, B1, ad, ldsA0, ldsB0, qa, qb, voffA, voffB);
  }
  if (n >= 2) {
    const uint64_t qa = pieces_for(3, pA, gA, sA), qb = pieces_for(3, pB, gB, sB);
    step_slot<1, 0>(acc, A0, B0, A1, B1, ad, ldsA0 + BUF, ldsB0 + BUF, qa, qb, voffA, voffB);
  }
  int t = 2;
  for (; t + 1 < n; t += 2) {
    const uint64_t qa0 = pieces_for(t + 2, pA, gA, sA), qb0 = pieces_for(t + 2, pB, gB, sB);
    step_slot<0, 0>(acc, A0, B0, A1, B1, ad, ldsA0, ldsB0, qa0, qb0, voffA, voffB);
    const uint64_t qa1 = pieces_for(t + 3, pA, gA, sA), qb1 = pieces_for(t + 3, pB, gB, sB);
    step_slot<1, 0>(acc, A0, B0, A1, B1, ad, ldsA0 + BUF, ldsB0 + BUF, qa1, qb1, voffA, voffB);
  }
  if (t < n) step_slot<0, 0>(acc, A0, B0, A1, B1, ad, ldsA0, ldsB0, gA, gB, voffA, voffB);
  // the surplus pieces have landed, the surplus fragment reads returned — for every wave: the images become the epilogue's scratch
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

  // ---- epilogue: per wave a private 16 × 128 fp32 slab in LDS, then 256 contiguous bytes per wave-instruction (atomics / adds / stores)
  float* slab = reinterpret_cast<float*>(smem) + wave * 2048;
  const int r16 = lane & 15, g = lane >> 4;
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) {
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) slab[(4 * g + r) * 128 + nt * 16 + r16] = acc[mt][nt][r] * it.alpha;
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes have landed
    __builtin_amdgcn_wave_barrier();
    if (unit_mode != 0 && a.wide_out) {
      // plain adds / stores (one contributor): FOUR consecutive columns per lane — 2 rows × 512 B per wave-instruction, 16-byte accesses
      const int c4 = (lane & 31) * 4, nq = n0 + wc * 128 + c4;
      f32x4 v4[8], old[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int row = (lane >> 5) + 2 * j, m = m0 + wr * 128 + mt * 16 + row;
        v4[j] = *reinterpret_cast<const f32x4*>(slab + row * 128 + c4);
        if (unit_mode == 1 && m < it.Mout && nq < it.N) old[j] = *reinterpret_cast<const f32x4*>(it.C + (int64_t)m * it.ldc + nq);
        else old[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int row = (lane >> 5) + 2 * j, m = m0 + wr * 128 + mt * 16 + row;
        if (m < it.Mout && nq < it.N) *reinterpret_cast<f32x4*>(it.C + (int64_t)m * it.ldc + nq) = v4[j] + old[j];
      }
      __builtin_amdgcn_wave_barrier();
      continue;
    }
#pragma unroll
    for (int row = 0; row < 16; ++row) {
      const int m = m0 + wr * 128 + mt * 16 + row;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float v = slab[row * 128 + h * 64 + lane];
        const int nn = n0 + wc * 128 + h * 64 + lane;
#ifdef LAKO_EXPERIMENTS
        if (a.no_atomics) continue;
#endif
        if (m < it.Mout && nn < it.N) {
          float* cp = it.C + (int64_t)m * it.ldc + nn;
          if (unit_mode == 2) *cp = v;
          else if (unit_mode == 1) *cp += v;
          else atomicAdd(cp, v);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}
