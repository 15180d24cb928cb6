// gemm_nt4_kernel: the NT tile on FOUR waves (one per SIMD, the accumulators in AGPRs) with a HAND-PLACED K loop (round 6).  Tiles of
// 256 × 256 (MT = 8: 128 × 128 per wave) and 192 × 256 (MT = 6: 96 × 128 per wave — where 192-row tiles land a launch on whole rounds of
// the chip).  Included by gemm.hip inside its anonymous namespace (NtArgs, xcd_remap, tile_coords, lako_keep4 …).
//
// Why: the eight-wave kernel's K-step costs ≈ 3 900 cycles for 2 048 MFMA cycles per SIMD (profiles/r05c_gemm_stamps.txt) — every wave
// waits `vmcnt(0)` for a K-slice that was requested less than one K-step earlier, and hipcc's schedule issues the fragment reads of a
// K-half as a block.  The schedule below is the one the disassembly of the vendor library's MT256x256x64 kernel shows (its structure, not
// its code: profiles/r06a_vendor_loop_schedule.txt):
//   * TWO K-steps of LDS-DMA in flight with TWO LDS buffers: K-slice t + 2 goes into the buffer of K-slice t, operand by operand, as
//     soon as every wave holds that operand's fragments of step t in registers (barrier 1: B, barrier 2: A);
//   * ONE counted wait per K-step, `vmcnt(13)`, about three quarters into the step: thirteen of the step's own pieces may be
//     outstanding, everything older — K-slice t + 1, requested a full K-step earlier — has landed; barrier 3 makes that true for all
//     waves, and the k-half-0 fragments of step t + 1 are read under the step's last MFMAs;
//   * every LDS read / DMA piece / wait / barrier sits at a FIXED MFMA slot (one inline-asm statement per MFMA slot; hipcc's
//     scheduler cannot move volatile asm statements past each other), never more than one memory instruction between two MFMAs;
//   * across tiles the stream never stops: the last two K-steps of a tile request the first two K-slices of the workgroup's next tile,
//     which land under the epilogue (whose transposition scratch is the 32 KiB behind the two 64-KiB buffers).
// LDS image, fragment layout, accumulator layout and the epilogue arithmetic are the eight-wave kernel's (bit-identical results:
// tests/test_kernels_gpu.py::test_gemm_nt4_*).
#pragma once

namespace nt4 {

constexpr int BN = 256, IMG = 256 * TKB, BUF = 2 * IMG, SCRATCH = 2 * BUF, LDS_BYTES = 2 * BUF + 4 * 8192;
static_assert(LDS_BYTES == 160 * 1024, "LDS");

// side operation of an MFMA slot
enum : int { OP_NONE = 0, OP_RD_B1, OP_RD_A1, OP_RD_B0N, OP_RD_A0N, OP_DMA_B, OP_DMA_A, OP_WAIT_L, OP_BAR, OP_WAIT_V };
struct Slot { int op, idx; };
struct Ev { int slot, op, idx; };

// The schedule of one K-step.  Slot s = 8·a + b of part 1 (k-half 0) is MFMA (m-tile a, n-tile b), part 2 (k-half 1) the same P = 8·MT
// slots later.  B is the INNER operand (all eight k-half fragments are needed at the start of a part), A the outer one (fragment a is
// needed at slot 8a of its part).
template <int MT> struct Sched;
template <> struct Sched<8> {
  static constexpr int N = 55;
  static constexpr Ev ev[N] = {
      {0, OP_RD_B1, 0},    {2, OP_RD_B1, 1},    {4, OP_RD_B1, 2},    {6, OP_RD_B1, 3},    {8, OP_RD_B1, 4},    {10, OP_RD_B1, 5},
      {12, OP_RD_B1, 6},   {14, OP_RD_B1, 7},   {20, OP_WAIT_L, 0},  {21, OP_BAR, 0},     // barrier 1: B of this step is in registers everywhere
      {22, OP_DMA_B, 0},   {24, OP_RD_A1, 0},   {25, OP_DMA_B, 1},   {27, OP_RD_A1, 1},   {28, OP_DMA_B, 2},   {30, OP_RD_A1, 2},
      {31, OP_DMA_B, 3},   {33, OP_RD_A1, 3},   {34, OP_DMA_B, 4},   {36, OP_RD_A1, 4},   {37, OP_RD_A1, 5},   {39, OP_RD_A1, 6},
      {41, OP_RD_A1, 7},   {50, OP_WAIT_L, 0},  {51, OP_BAR, 0},     // barrier 2: A of this step is in registers everywhere
      {52, OP_DMA_B, 5},   {55, OP_DMA_B, 6},   {58, OP_DMA_B, 7},   {61, OP_DMA_A, 0},   {64, OP_DMA_A, 1},   {85, OP_DMA_A, 2},
      {87, OP_DMA_A, 3},   {89, OP_DMA_A, 4},   {91, OP_WAIT_V, 13}, {92, OP_BAR, 0},     // barrier 3: K-slice t + 1 has landed for every wave
      {93, OP_RD_B0N, 0},  {94, OP_RD_B0N, 1},  {95, OP_RD_B0N, 2},  {96, OP_DMA_A, 5},   {97, OP_RD_B0N, 3},  {98, OP_RD_B0N, 4},
      {100, OP_DMA_A, 6},  {101, OP_RD_B0N, 5}, {103, OP_RD_B0N, 6}, {104, OP_RD_B0N, 7}, {105, OP_RD_A0N, 0}, {106, OP_RD_A0N, 1},
      {109, OP_RD_A0N, 2}, {112, OP_RD_A0N, 3}, {114, OP_RD_A0N, 4}, {117, OP_RD_A0N, 5}, {120, OP_RD_A0N, 6}, {124, OP_RD_A0N, 7},
      {125, OP_DMA_A, 7},  {126, OP_WAIT_L, 0},
  };
};
template <> struct Sched<6> {
  static constexpr int N = 49;
  static constexpr Ev ev[N] = {
      {0, OP_RD_B1, 0},   {2, OP_RD_B1, 1},   {4, OP_RD_B1, 2},   {6, OP_RD_B1, 3},   {8, OP_RD_B1, 4},   {10, OP_RD_B1, 5},
      {12, OP_RD_B1, 6},  {14, OP_RD_B1, 7},  {16, OP_WAIT_L, 0}, {17, OP_BAR, 0},
      {18, OP_DMA_B, 0},  {19, OP_RD_A1, 0},  {21, OP_DMA_B, 1},  {22, OP_RD_A1, 1},  {24, OP_DMA_B, 2},  {25, OP_RD_A1, 2},
      {27, OP_DMA_B, 3},  {28, OP_RD_A1, 3},  {30, OP_DMA_B, 4},  {31, OP_RD_A1, 4},  {33, OP_RD_A1, 5},  {41, OP_WAIT_L, 0},
      {42, OP_BAR, 0},    {43, OP_DMA_B, 5},  {45, OP_DMA_B, 6},  {47, OP_DMA_B, 7},  {49, OP_DMA_A, 0},  {51, OP_DMA_A, 1},
      {53, OP_DMA_A, 2},  {55, OP_DMA_A, 3},  {57, OP_DMA_A, 4},  {59, OP_WAIT_V, 13}, {60, OP_BAR, 0},
      {61, OP_RD_B0N, 0}, {62, OP_RD_B0N, 1}, {63, OP_RD_B0N, 2}, {64, OP_DMA_A, 5},  {65, OP_RD_B0N, 3}, {66, OP_RD_B0N, 4},
      {69, OP_RD_B0N, 5}, {71, OP_RD_B0N, 6}, {72, OP_RD_B0N, 7}, {73, OP_RD_A0N, 0}, {74, OP_RD_A0N, 1}, {77, OP_RD_A0N, 2},
      {80, OP_RD_A0N, 3}, {82, OP_RD_A0N, 4}, {85, OP_RD_A0N, 5}, {94, OP_WAIT_L, 0},
  };
};

template <int MT>
constexpr Slot slot_of(int s) {
  for (int i = 0; i < Sched<MT>::N; ++i)
    if (Sched<MT>::ev[i].slot == s) return {Sched<MT>::ev[i].op, Sched<MT>::ev[i].idx};
  return {OP_NONE, 0};
}

// The rules a table must obey (checked at compile time):
//  every fragment read and DMA piece exactly once, at most one event per slot, nothing at the step's last slot (it carries the wait states
//  for the loop exit);
//  B1 reads → lgkmcnt(0) → barrier 1 → the B pieces;  A1 reads → lgkmcnt(0) → barrier 2 → the A pieces, the lgkmcnt(0) before part 2 starts;
//  a fragment register is overwritten only after its last MFMA of the step (B0[j]: slot P − 8 + j, A0[i]: 8i + 7; the k-half-1 registers are
//  rewritten in the NEXT step's part 1, after their last use at the end of this one);
//  the counted wait's operand = the pieces issued before it; barrier 3 behind it; the next step's k-half-0 reads behind barrier 3 and in
//  front of the final lgkmcnt(0).
template <int MT>
constexpr bool valid() {
  constexpr int P = 8 * MT, E = 2 * P;
  int first[10][8] = {}, cnt[10][8] = {};
  int waits_l[3] = {-1, -1, -1}, bars[3] = {-1, -1, -1}, nw = 0, nb = 0, wait_v = -1, wait_v_n = -1, seen = 0;
  for (int s = 0; s < E; ++s) {
    const Slot sl = slot_of<MT>(s);
    if (sl.op == OP_NONE) continue;
    ++seen;
    if (sl.op == OP_WAIT_L) { if (nw >= 3) return false; waits_l[nw++] = s; continue; }
    if (sl.op == OP_BAR) { if (nb >= 3) return false; bars[nb++] = s; continue; }
    if (sl.op == OP_WAIT_V) { if (wait_v >= 0) return false; wait_v = s; wait_v_n = sl.idx; continue; }
    if (sl.idx < 0 || sl.idx >= 8) return false;
    if (cnt[sl.op][sl.idx]++ == 0) first[sl.op][sl.idx] = s;
  }
  if (seen != Sched<MT>::N) return false;      // (two events on one slot: the first of the table would win silently)
  if (slot_of<MT>(E - 1).op != OP_NONE) return false;
  if (nw != 3 || nb != 3 || wait_v < 0) return false;
  for (int j = 0; j < 8; ++j) {
    if (cnt[OP_RD_B1][j] != 1 || cnt[OP_RD_B0N][j] != 1 || cnt[OP_DMA_B][j] != 1) return false;
    if (!(first[OP_RD_B1][j] < waits_l[0])) return false;
    if (!(first[OP_DMA_B][j] > bars[0])) return false;
    if (!(first[OP_RD_B0N][j] > bars[2] && first[OP_RD_B0N][j] < waits_l[2] && first[OP_RD_B0N][j] > P - 8 + j)) return false;
  }
  for (int i = 0; i < 8; ++i) {
    const int want = i < MT ? 1 : 0;
    if (cnt[OP_RD_A1][i] != want || cnt[OP_RD_A0N][i] != want || cnt[OP_DMA_A][i] != want) return false;
    if (!want) continue;
    if (!(first[OP_RD_A1][i] < waits_l[1])) return false;
    if (!(first[OP_DMA_A][i] > bars[1])) return false;
    if (!(first[OP_RD_A0N][i] > bars[2] && first[OP_RD_A0N][i] < waits_l[2] && first[OP_RD_A0N][i] > 8 * i + 7)) return false;
  }
  if (!(waits_l[0] < bars[0] && bars[0] < waits_l[1] && waits_l[1] < bars[1] && waits_l[1] < P && bars[1] < wait_v && wait_v < bars[2])) return false;
  int before = 0;
  for (int j = 0; j < 8; ++j) before += first[OP_DMA_B][j] < wait_v ? 1 : 0;
  for (int i = 0; i < MT; ++i) before += first[OP_DMA_A][i] < wait_v ? 1 : 0;
  return before == wait_v_n;
}
static_assert(valid<8>(), "schedule of the 256-row tile");
static_assert(valid<6>(), "schedule of the 192-row tile");

struct Addr {            // per-lane LDS byte addresses of the fragment reads (m-/n-tile 0; tile i is + i · 2048)
  uint32_t a[2][2];      // [buffer][k-half]
  uint32_t b[2][2];
};

// one LDS-DMA piece outside the K loop (prologue)
__device__ __forceinline__ void dma_piece(uint32_t lds_dst, uint32_t voff, uint64_t sbase) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
}

// (`s_add_u32 m0, …` overwrites SCC: every statement that contains one says so — hipcc keeps scalar compares of its own alive across asm
// statements, and a stale SCC selected the wrong K-slice pointer in the first version of the weight-gradient kernel.)
// One K-step from buffer X (compile-time): acc += A(step)·B(step)ᵀ; requests K-slice t + 2 (pA / pB: its first byte of row 0 of the
// tile, voffA / voffB: the per-lane row · ld + swizzled chunk of piece j) into buffer X; leaves the k-half-0 fragments of step t + 1
// (buffer X ^ 1) in A0 / B0.  (Recursive templates, not lambdas: hipcc rejects asm operands captured by a generic lambda.)
template <int MT, int X, int s, bool FIRST = false>
__device__ __forceinline__ void step_slot(f32x4 (&acc)[8][MT], u32x4 (&A0)[MT], u32x4 (&B0)[8], u32x4 (&A1)[MT], u32x4 (&B1)[8], const Addr& ad,
                                          uint32_t ldsA, uint32_t ldsB, uint64_t pA, uint64_t pB, const uint32_t (&voffA)[MT],
                                          const uint32_t (&voffB)[8]) {
  constexpr int P = 8 * MT;
  if constexpr (s < 2 * P) {
    constexpr int part = s / P, ma = (s % P) / 8, nb = s % 8;
    constexpr Slot sl = slot_of<MT>(s);
    u32x4& af = part ? A1[ma] : A0[ma];
    u32x4& bf = part ? B1[nb] : B0[nb];
    f32x4& c = acc[nb][ma];
    if constexpr (FIRST && s < P) {
      // a tile's first K-step: part 1 STARTS the accumulators (C = 0) — no zeroing pass over 256 registers between the tiles.  The side
      // operation of the slot is its own statement here (the MFMA's output operand is write-only).
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(bf), "v"(af));
      if constexpr (sl.op == OP_RD_B1) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(B1[sl.idx]) : "v"(ad.b[X][1]), "n"(sl.idx * 2048) : "memory");
      } else if constexpr (sl.op == OP_RD_A1) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(A1[sl.idx]) : "v"(ad.a[X][1]), "n"(sl.idx * 2048) : "memory");
      } else if constexpr (sl.op == OP_DMA_B) {
        asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3" : : "s"(ldsB), "n"(sl.idx * 4096), "v"(voffB[sl.idx]), "s"(pB) : "memory", "scc");
      } else if constexpr (sl.op == OP_DMA_A) {
        asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3" : : "s"(ldsA), "n"(sl.idx * 4096), "v"(voffA[sl.idx]), "s"(pA) : "memory", "scc");
      } else if constexpr (sl.op == OP_WAIT_L) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      } else if constexpr (sl.op == OP_BAR) {
        asm volatile("s_barrier" ::: "memory");
      } else {
        static_assert(sl.op == OP_NONE, "part 1 of a step holds k-half-1 reads, DMA pieces, lgkmcnt waits and barriers only");
      }
    } else if constexpr (s == 2 * P - 1) {
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
                   : "+a"(c) : "v"(bf), "v"(af), "s"(ldsB), "n"(sl.idx * 4096), "v"(voffB[sl.idx]), "s"(pB) : "memory", "scc");
    } else if constexpr (sl.op == OP_DMA_A) {
      asm volatile("s_add_u32 m0, %3, %4\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\tglobal_load_lds_dwordx4 %5, %6"
                   : "+a"(c) : "v"(bf), "v"(af), "s"(ldsA), "n"(sl.idx * 4096), "v"(voffA[sl.idx]), "s"(pA) : "memory", "scc");
    } else if constexpr (sl.op == OP_WAIT_L) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_waitcnt lgkmcnt(0)" : "+a"(c) : "v"(bf), "v"(af) : "memory");
    } else if constexpr (sl.op == OP_BAR) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_barrier" : "+a"(c) : "v"(bf), "v"(af) : "memory");
    } else if constexpr (sl.op == OP_WAIT_V) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_waitcnt vmcnt(%3)" : "+a"(c) : "v"(bf), "v"(af), "n"(sl.idx) : "memory");
    }
    step_slot<MT, X, s + 1, FIRST>(acc, A0, B0, A1, B1, ad, ldsA, ldsB, pA, pB, voffA, voffB);
  }
}
template <int N_, int I>
__device__ __forceinline__ void read_frags(u32x4 (&dst)[N_], uint32_t addr) {
  if constexpr (I < N_) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[I]) : "v"(addr), "n"(I * 2048) : "memory");
    read_frags<N_, I + 1>(dst, addr);
  }
}

// per-lane source offsets of a tile's pieces of one operand (piece j of a wave = image rows 32 j + 8 wave … + 7; rows past the edge
// clamped to the last valid row: they feed accumulator rows that are never stored)
template <int NP>
__device__ __forceinline__ void piece_offsets(uint32_t (&voff)[NP], int rows_valid, int64_t ld_bytes, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int row = (wave + 4 * j) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    voff[j] = (uint32_t)(min(row, rows_valid - 1) * ld_bytes + c * 16);
  }
}

}  // namespace nt4

// ---- epilogue of one tile (shared by the bf16 kernels and the MX kernel below: the accumulator layout is the same).
// The MFMA pipes idle while it runs (the accumulators are the next tile's too), so it is built for latency: the
// wave's (16·MT) × 128 fp32 region goes through its 8 KiB of LDS scratch 16 rows at a time — written straight from the accumulator
// registers (`ds_write_b128` takes AGPR data: no register copies, no arithmetic in the accumulator layout) — and comes back
// row-major, a lane owning 8 consecutive columns of a row (two dropout quads): alpha, ReLU, aux mask, dropout, residual in the generic
// epilogue's order on every element, one rounding, ONE 16-byte buffer store per lane and 4 rows (4 × 256 B per wave-instruction;
// rows past the edge fall out of the descriptor's range, lanes past the last column start from an out-of-range offset; the row
// offsets are scalar).  LDS executes a wave's accesses in order: pass p + 1 is written as soon as pass p's reads have been ISSUED,
// and its write latency hides behind pass p's arithmetic and stores.  (Every lane-derived value is rebuilt from an opaque lane id:
// as invariants of the tile loop they would be hoisted above the K loop and spilled.)
template <int MT, bool SIDE, int EPI>
__device__ __forceinline__ void nt4_epilogue(const NtArgs& a, f32x4 (&acc)[8][MT], char* smem, int wave, int wr, int wc, int m0, int n0) {
  using namespace nt4;
  const int le = fresh_lane();
  const int r16e = le & 15, ge = le >> 4, cj = le & 15, rq = le >> 4;
  char* ep = smem + SCRATCH + wave * 8192;
  const int mw = m0 + wr * (MT * 16), nw = n0 + wc * 128;                       // first row / column of the wave's region
  const int rows_v = min(MT * 16, a.M - mw), cols_v = min(128, a.N - nw);
  const int ldc_b = (int)a.ldc * 2;
  const bool in_cols = cj * 8 < cols_v;
  const auto crs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<bf16_t*>(a.C) + (int64_t)mw * a.ldc + nw, 0,
                                                     (rows_v > 0 && cols_v > 0) ? (rows_v - 1) * ldc_b + cols_v * 2 : 0, 0x00020000);
  const int cvo = (in_cols && !NT_DBG(a, 8)) ? rq * ldc_b + cj * 16 : (int)0x80000000;
  // side operand (SIDE instantiation): same addressing on its own descriptor
  const bool has_res = SIDE && (a.flags & LAKO_EPI_RESID);
  const int lds_b = SIDE ? (int)(has_res ? a.ldr : a.ldaux) * 2 : 0;
  const auto srs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(reinterpret_cast<const bf16_t*>(has_res ? a.resid : a.aux)) + (SIDE ? (int64_t)mw * (has_res ? a.ldr : a.ldaux) + nw : 0), 0,
      (SIDE && rows_v > 0 && cols_v > 0) ? (rows_v - 1) * lds_b + cols_v * 2 : 0, 0x00020000);
  const int svo = in_cols ? rq * lds_b + cj * 16 : (int)0x80000000;
  const uint32_t wr_base = (uint32_t)(r16e * 512), rd_base = (uint32_t)(rq * 512);
  // quad of (row mw + rq, column nw + 8 cj); rows 4 apart are N quads apart
  const uint32_t quad0 = (uint32_t)((((uint64_t)(a.row0 + mw + rq)) * (uint64_t)a.N + (uint64_t)(nw + cj * 8)) >> 2);
  auto write_pass = [&](int mt) {
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) *reinterpret_cast<f32x4*>(ep + wr_base + (((nt * 4 + ge) ^ r16e) * 16)) = acc[nt][mt];
  };
  // ALPHA: the product is scaled (only the plain epilogue is built with it: every fused epilogue of the reader has alpha = 1, and with one
  // wave per SIMD the two packed multiplies per four outputs are a third of the plain epilogue's vector work — nt4_ok sends a fused
  // epilogue with alpha != 1 to the eight-wave kernels)
  auto run = [&](auto RELU_, auto DROP_, auto RES_, auto ALPHA_) {
    constexpr bool RELU = decltype(RELU_)::value, DROP = decltype(DROP_)::value, RES = SIDE && decltype(RES_)::value, AUX = SIDE && !RES;
    constexpr bool ALPHA = decltype(ALPHA_)::value;
    // the side operand's 16 bytes per lane and row, requested SD passes ahead (a ring of SD × 4 registers quads: one pass is ≈ 800 cycles of
    // this wave's work, a load from HBM several thousand — one pass ahead left the epilogue waiting for memory in every pass:
    // profiles/r06h_nt4_epilogue_stamps.txt)
#ifndef LAKO_NT4_SD
#define LAKO_NT4_SD 2
#endif
    constexpr int SD = LAKO_NT4_SD;
    u32x4 side[SD][4];
    if constexpr (SIDE) {
#pragma unroll
      for (int p0 = 0; p0 < SD && p0 < MT; ++p0)
#pragma unroll
        for (int it = 0; it < 4; ++it) side[p0][it] = __builtin_amdgcn_raw_buffer_load_b128(srs, svo, (p0 * 16 + it * 4) * lds_b, 0);
    }
    write_pass(0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      // the pass's rows come back in two halves (2 × 4 reads): the second half is requested — and the next pass written — once the first
      // half's arithmetic has been issued (LDS executes a wave's accesses in order: the write cannot overtake the reads)
      f32x4 v[4][2];
      auto read_rows = [&](int it0) {
#pragma unroll
        for (int it = it0; it < it0 + 2; ++it)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int row_l = it * 4 + rq;
            v[it][h] = *reinterpret_cast<const f32x4*>(ep + rd_base + it * 2048 + (((2 * cj + h) ^ row_l) * 16));
          }
      };
      read_rows(0);
      u32x4 sv4[4];
      if constexpr (SIDE) {      // this pass's side data out of the ring, the pass SD ahead into its place
#pragma unroll
        for (int it = 0; it < 4; ++it) sv4[it] = side[mt % SD][it];
        if (mt + SD < MT) {
#pragma unroll
          for (int it = 0; it < 4; ++it) side[mt % SD][it] = __builtin_amdgcn_raw_buffer_load_b128(srs, svo, ((mt + SD) * 16 + it * 4) * lds_b, 0);
        }
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        if (it == 2) {
          read_rows(2);
          if (mt + 1 < MT) write_pass(mt + 1);
        }
        bf16x8 sv = {}, o;
        if constexpr (SIDE) sv = __builtin_bit_cast(bf16x8, sv4[it]);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          f32x4 x = v[it][h];
          if constexpr (ALPHA) x = x * a.alpha;
          if constexpr (RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = fmaxf(x[r], 0.f);
          }
          if constexpr (AUX) {
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = (float)sv[4 * h + r] > 0.f ? x[r] * a.aux_scale : 0.f;
          }
          if constexpr (DROP) {
            // the dropout quad (4 consecutive elements of a row) of these four outputs: element index / 4, below 2^32 (nt4_ok) — 32-bit
            // additions from the lane's first quad instead of a 64-bit multiply per row (the index arithmetic was half of the
            // 32-bit multiplies of this epilogue, and with one wave per SIMD they are wall time: profiles/r06h_nt4_epilogue_stamps.txt)
            bool kp[4];
            lako_keep4(a.drop_key, (uint64_t)(quad0 + (uint32_t)(mt * 4 + it) * (uint32_t)a.N + (uint32_t)h), a.drop_thresh, kp);
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = kp[r] ? x[r] * a.drop_scale : 0.f;
          }
          if constexpr (RES) {
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] += (float)sv[4 * h + r];
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) o[4 * h + r] = (bf16_t)x[r];
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), crs, cvo, (mt * 16 + it * 4) * ldc_b, 0);
      }
    }
  };
  // DIRECT form of the dropout epilogues without a side operand.  The MFMAs run with the operands swapped, so a lane already holds four
  // consecutive COLUMNS of one row — a dropout quad — per 16 × 16 tile: the same arithmetic in the accumulator layout, rounded to bf16, then
  // `v_permlane16_swap` (gfx950: the odd 16-lane rows of one register against the even rows of another) pairs the halves of two
  // neighbouring n-tiles into 16 bytes of a row per lane — no LDS pass; a store covers 16 rows × 64 bytes.  Measured
  // (tools/gemm_nt4_direct.py, profiles/r06m_nt4_direct_epilogue.txt): −1 … −3 % per launch for dropout and ReLU + dropout, +3 … +7 % for
  // plain / ReLU / alpha (the 64-byte runs cost the stores more than the LDS pass cost the wave) — so only the dropout forms exist.
  auto run_direct = [&](auto RELU_, auto DROP_, auto ALPHA_) {
    constexpr bool RELU = decltype(RELU_)::value, DROP = decltype(DROP_)::value, ALPHA = decltype(ALPHA_)::value;
    const int mrow = le & 15;
    const int colb = ((ge & 1) * 16 + (ge >> 1) * 8) * 2;                  // first byte of the lane's 16: g = 0 / 1 / 2 / 3 → 0 / 32 / 16 / 48
    int dvo[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) dvo[p] = (p * 32 + colb / 2 < cols_v && !NT_DBG(a, 8)) ? mrow * ldc_b + colb + p * 64 : (int)0x80000000;
    // quad of (row mw + mrow, column nw + 4 g): rows 16 apart are 4 N quads apart, n-tiles 4 quads
    const uint32_t dq0 = (uint32_t)((((uint64_t)(a.row0 + mw + mrow)) * (uint64_t)a.N + (uint64_t)(nw + ge * 4)) >> 2);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        uint32_t pk[2][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          f32x4 x = acc[2 * p + h][mt];
          if constexpr (ALPHA) x = x * a.alpha;
          if constexpr (RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = fmaxf(x[r], 0.f);
          }
          if constexpr (DROP) {
            bool kp[4];
            lako_keep4(a.drop_key, (uint64_t)(dq0 + (uint32_t)(mt * 4) * (uint32_t)a.N + (uint32_t)((2 * p + h) * 4)), a.drop_thresh, kp);
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = kp[r] ? x[r] * a.drop_scale : 0.f;
          }
          const bf16x4 o4 = {(bf16_t)x[0], (bf16_t)x[1], (bf16_t)x[2], (bf16_t)x[3]};
          const u32x2 w = __builtin_bit_cast(u32x2, o4);
          pk[h][0] = w[0];
          pk[h][1] = w[1];
        }
        // even rows of lanes keep n-tile 2p (their own half + the neighbour's), odd rows n-tile 2p + 1
        const u32x2 s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
        const u32x2 s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{s0[0], s1[0], s0[1], s1[1]}, crs, dvo[p], (mt * 16) * ldc_b, 0);
        __builtin_amdgcn_sched_barrier(0);      // (the scheduler otherwise copies the accumulators of many groups out at once: spills)
      }
    }
  };
  const bool relu = a.flags & LAKO_EPI_RELU, drop = a.drop_thresh != 0;
  using T_ = std::true_type;
  using F_ = std::false_type;
  if constexpr (EPI == 1) run_direct(F_{}, T_{}, F_{});
  else if constexpr (EPI == 2) run_direct(T_{}, T_{}, F_{});
  else if constexpr (SIDE) {
    // (the host sends here: residual with or without dropout, or the aux mask alone, alpha = 1 — nt4_ok)
    if (has_res) {
      if (drop) run(F_{}, T_{}, T_{}, F_{});
      else run(F_{}, F_{}, T_{}, F_{});
    } else {
      run(F_{}, F_{}, F_{}, F_{});
    }
  } else {
    if (drop) {
      if (relu) run(T_{}, T_{}, F_{}, F_{});
      else run(F_{}, T_{}, F_{}, F_{});
    } else if (relu) {
      run(T_{}, F_{}, F_{}, F_{});
    } else if (a.alpha != 1.0f) {
      run(F_{}, F_{}, F_{}, T_{});
    } else {
      run(F_{}, F_{}, F_{}, F_{});
    }
  }
}

// MT: m-tiles per wave (8: 256-row tiles, 6: 192-row tiles).  SIDE = false: alpha, ReLU, dropout; true: also the aux mask and the residual
// (the arithmetic then happens after the transposition, on row-major fp32 values — same operations in the same order on every element).
// EPI (SIDE = false only): 0 = the epilogues through the LDS transposition, chosen at run time from the flags; 1 / 2 = dropout / ReLU + dropout straight
// from the accumulator layout — its own instantiation each: as sibling branches of one kernel the accumulator reads common to them are
// hoisted above the branch (256 live VGPRs: spills), and pinned in place they cost register copies inside the K loop.
template <int MT, bool SIDE, int EPI = 0>
__global__ __launch_bounds__(256) void gemm_nt4_kernel(NtArgs a) {
  using namespace nt4;
  constexpr int BM = MT * 32;
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
      ad.a[x][0] = lds0 + x * BUF + (uint32_t)((wr * MT * 16 + r16) * TKB) + key16;
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
  uint32_t voffA[MT], voffB[8];
  piece_offsets<MT>(voffA, min(BM, a.M - m0), lda_b, wave, lane);
  piece_offsets<8>(voffB, min(BN, a.N - n0), ldb_b, wave, lane);
  // prologue: K-slices 0 and 1 of the first tile, then the k-half-0 fragments of step 0
  {
    const uint64_t pA = reinterpret_cast<uint64_t>(a.A + (int64_t)m0 * lda_b), pB = reinterpret_cast<uint64_t>(a.B + (int64_t)n0 * ldb_b);
#pragma unroll
    for (int x = 0; x < 2; ++x) {
#pragma unroll
      for (int j = 0; j < MT; ++j) dma_piece(ldsA0 + x * BUF + j * 4096, voffA[j], pA + x * TKB);
#pragma unroll
      for (int j = 0; j < 8; ++j) dma_piece(ldsB0 + x * BUF + j * 4096, voffB[j], pB + x * TKB);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

  int kst = 0;      // (stamps only) K-steps so far, over all tiles of the workgroup
  while (true) {
    const int next_tile = tile + gridDim.x;
    const bool has_next = next_tile < nwg;
    tile_coords(has_next ? next_tile : tile, a.tiles_m, a.tiles_n, a.group_m, tm_, tn_);
    const int nm0 = tm_ * BM, nn0 = tn_ * BN;
    const char* Abase = a.A + (int64_t)m0 * lda_b;
    const char* Bbase = a.B + (int64_t)n0 * ldb_b;

    f32x4 acc[8][MT];      // [n-tile][m-tile]; element r of lane (l & 15, g): C[m = mt·16 + (l & 15)][n = nt·16 + 4g + r]
    // the k-half-0 fragments of the tile's first K-step (buffer 0: landed and published by the previous tile's last counted wait + barrier
    // 3, or by the prologue).  The previous tile's last step has read them already — re-reading them here (≈ 300 cycles per tile) instead of
    // carrying 64 registers across the epilogue is what lets the epilogue keep four passes of side-operand loads in flight without spills.
    u32x4 A0[MT], B0[8], A1[MT], B1[8];
    read_frags<8, 0>(B0, ad.b[0][0]);
    read_frags<MT, 0>(A0, ad.a[0][0]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    uint64_t pA = reinterpret_cast<uint64_t>(Abase) + 2 * TKB, pB = reinterpret_cast<uint64_t>(Bbase) + 2 * TKB;
    // the stream moves on to the workgroup's next tile (its K-slices 0 and 1) in the tile's last pair of steps; the last tile requests its
    // own first slices again (the pieces are always issued: the counted wait depends on it) — they land in buffers nobody reads any more
    auto next_tile_stream = [&]() {
      pA = reinterpret_cast<uint64_t>(a.A + (int64_t)nm0 * lda_b);
      pB = reinterpret_cast<uint64_t>(a.B + (int64_t)nn0 * ldb_b);
      piece_offsets<MT>(voffA, min(BM, a.M - nm0), lda_b, wave, fresh_lane());
      piece_offsets<8>(voffB, min(BN, a.N - nn0), ldb_b, wave, fresh_lane());
    };
    if (NT_DBG(a, 64) && blockIdx.x == 0 && (threadIdx.x & 63) == 0 && kst < 128)      // (stamps) 4: the 100-MHz clock at the same point as 1
      reinterpret_cast<unsigned long long*>(a.queue)[((threadIdx.x >> 6) * 128 + kst) * 8 + 4] = __builtin_amdgcn_s_memrealtime();
    NT_STAMP(1);      // (experiments build, debug bit 6: s_memtime of workgroup 0 — 0: top of a K-step, 1: top of a tile's K loop, 2 / 3: begin / end of the epilogue)
    if (nk == 2) next_tile_stream();
    NT_STAMP(0);
    step_slot<MT, 0, 0, true>(acc, A0, B0, A1, B1, ad, ldsA0, ldsB0, pA, pB, voffA, voffB);
    ++kst;
    NT_STAMP(0);
    step_slot<MT, 1, 0>(acc, A0, B0, A1, B1, ad, ldsA0 + BUF, ldsB0 + BUF, pA + TKB, pB + TKB, voffA, voffB);
    ++kst;
    pA += 2 * TKB;
    pB += 2 * TKB;
    for (int t = 2; t < nk; t += 2) {
      if (t + 2 == nk) next_tile_stream();
      NT_STAMP(0);
      step_slot<MT, 0, 0>(acc, A0, B0, A1, B1, ad, ldsA0, ldsB0, pA, pB, voffA, voffB);
      ++kst;
      NT_STAMP(0);
      step_slot<MT, 1, 0>(acc, A0, B0, A1, B1, ad, ldsA0 + BUF, ldsB0 + BUF, pA + TKB, pB + TKB, voffA, voffB);
      ++kst;
      pA += 2 * TKB;
      pB += 2 * TKB;
    }

    // ---- epilogue (nt4_epilogue above) ----
    NT_STAMP(2);
    if (NT_DBG(a, 128)) {      // (experiments build: no epilogue at all — what a tile costs outside its K loop)
      asm volatile("" :: "a"(acc[0][0]), "a"(acc[7][MT - 1]));
    } else {
      nt4_epilogue<MT, SIDE, EPI>(a, acc, smem, wave, wr, wc, m0, n0);
    }
    NT_STAMP(3);
    if (!has_next) break;
    tile = next_tile;
    m0 = nm0;
    n0 = nn0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the last tile's surplus pieces must have landed before the LDS is released
}
