// Fast path of the T5 attention (HF5:144-173,281-369; src/model.py:286-349): bf16, d_head 64, at most 256 keys per sequence,
// no causal mask, no score capture — the ENCODER self-attention of the FiD reader, which is 96 % of the attention work of a
// training step (B·N passages × H heads of ≤ text_maxlength tokens).  Same math, same launch contract and the same dropout
// recipe as the generic kernels of attn.hip (which keep every other case: fp32 parity mode, d_head 32, causal decoder
// self-attention, key-split cross-attention over N·L keys, score capture).
//
// Why a second implementation: the generic kernels issue ≈27 vector instructions per score element in the forward and
// ≈36 / 40 in the two backward passes (rocprofv3 SQ_INSTS_VALU ÷ MFMA count, profiles/r02a_attn_pmc.txt) against 4–8 MFMAs
// per 16×16 tile: they are bound by vector issue, and a third of their LDS cycles are bank conflicts.  Here
//   * the K / V / Q / dO images in LDS are unpadded 128-byte rows whose 32-byte slots are XOR-swizzled by (row >> 1) & 3:
//     conflict-free for BOTH access patterns an image sees — ds_read_b128 row fragments (score products) and
//     ds_read_b64_tr_b16 transposed fragments (P·V, dSᵀ·K, Pᵀ·dO, dSᵀ·Q) — with per-lane offsets that are loop constants;
//   * the relative-position bias is the MFMA accumulator's initial value, read as ONE aligned 16-byte LDS vector per tile
//     from four copies of the [2L−1] table shifted by 0..3 floats (key − query changes by 16 per tile, so the copy a lane
//     needs and its base address are constants of the 16-row block; no index clamps, no per-element gathers);
//   * exp(s − m) is v_fma + v_exp (base-2 domain), the normaliser and the dropout scale are folded into per-row factors;
//   * dropout draws come from the full-rate block hash of attn_shared.h (no 32-bit multiplies);
//   * no key mask: sequences are ragged (packed rows + offsets, the training path) or exactly Lk long, so only the LAST tile
//     pair of a sequence can hold keys past its end — a −inf vector built in registers; launches with a key mask (the padded
//     path: LAKO_UNPAD=0, masks with holes) stay on the generic kernels.
// Occupancy is what these kernels are bound by (rocprofv3: with 8 waves per CU every wave waits half of its cycles and the
// vector pipes are a third busy; the work does not get faster by issuing fewer instructions, it gets faster with more waves):
// the forward and the dK/dV pass run 8 waves per workgroup on one staged image pair — 16 waves per CU, ≤ 128 registers —
// and the dQ pass stages K / V ONCE per (sequence, head) for all its query blocks (attn.hip: once per 4 blocks).
// Structure as in attn.hip: one workgroup (4 waves) per (sequence, head[, block group]); one operand side staged in LDS, the
// other held per wave as MFMA fragments loaded straight from HBM; the score tile leaves the MFMA with the LDS-side index in
// the accumulator registers, so it is directly the B operand of the product that contracts over that index.
#include <algorithm>

#include "attn_shared.h"
#include "det.h"
#include "lds_image.h"

namespace {

LAKO_DET_STORAGE      // LAKO_DETERMINISTIC=1: see det.h

constexpr int EB_PADLO = 16;              // bias table: entries in front of index 0 (queries past Lq in the last block)
constexpr int EB_ST = 592;                // floats per shifted copy: ≥ 511 + 16 + 36, ≡ 16 (mod 64) → the 4 copies sit 16 banks apart
constexpr float LOG2E = 1.4426950408889634f;

// The four shifted copies of the (optionally reversed) bias table of head h:  copy c, float j  =  T'[j + c],
// T'[t] = rel_bias[h][t − PADLO] inside [0, R) and 0 outside (only masked keys / discarded queries ever read the padding);
// REV: T' is read backwards, T'rev[t] = T'[last − t] (the dK/dV pass walks key − query downwards along the registers).
// (Round 5: ALL of a thread's table reads are requested before the first one is waited for.  As a plain loop — read, store to LDS, next —
//  hipcc waits `vmcnt(0)` in front of every store: five serialised L2 round trips at the start of every forward workgroup, the first of
//  them behind the K / V image DMA, ≈ 3 000 – 4 000 of a workgroup's ≈ 27 000 cycles — profiles/r05t_attn_fwd_stamps.txt.)
template <int NT = 256>
__device__ __forceinline__ void estage_bias(float* b4, const float* rel_bias, int h, int R, bool rev) {
  const int last = R - 1 + EB_PADLO + 36;
  constexpr int IT = (4 * EB_ST + NT - 1) / NT;
  float v[IT];
#pragma unroll
  for (int k = 0; k < IT; ++k) {
    const int i = (int)threadIdx.x + k * NT;
    const int c = i / EB_ST, j = i - c * EB_ST;
    int t = j + c;
    if (rev) t = last - t;
    t -= EB_PADLO;
    const bool in = rel_bias != nullptr && i < 4 * EB_ST && t >= 0 && t < R;
    v[k] = in ? rel_bias[(int64_t)h * R + (in ? t : 0)] : 0.f;
  }
#pragma unroll
  for (int k = 0; k < IT; ++k) {
    const int i = (int)threadIdx.x + k * NT;
    if (i < 4 * EB_ST) b4[i] = v[k];
  }
}

__device__ __forceinline__ float egroup_max(float v) {  // across the 4 lane groups (same lane & 15)
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float egroup_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

__device__ __forceinline__ void eload_frags(u32x4 (&f)[2], const char* base, int64_t stride_b, int row, int L, int g) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row < L) v = *reinterpret_cast<const u32x4*>(base + (int64_t)row * stride_b + (i * 4 + g) * 16);
    f[i] = v;
  }
}

// The same through a buffer descriptor (persistent kernels): rows [0, L) of the strided [L, 64] slice at `base`; the lane's 32-bit
// offset row·stride + 16g is a constant of the lane, rows past L fall out of the descriptor's range (zeros) — no 64-bit per-lane
// pointers, no branches.  `voff` = (lane & 15)·stride_b + 16·(lane >> 4), `row0` the (wave-uniform) first row.
// (The first row goes into the descriptor's BASE, not into the scalar offset operand: a raw buffer's range check covers the vector
// offset + the instruction offset only.)
__device__ __forceinline__ void eload_frags_buf(u32x4 (&f)[2], const char* base, uint32_t stride_b, int row0, int L, uint32_t voff) {
  const int left = L - row0;                       // rows of the slice from row0 on
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base) + (int64_t)row0 * stride_b, 0,
                                                     left > 0 ? (int)((uint32_t)(left - 1) * stride_b + 128u) : 0, 0x00020000);
#pragma unroll
  for (int i = 0; i < 2; ++i) f[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff + i * 64, 0, 0));
}

// tile[image row row0 + 4g + r][register side = lane & 15] = init + Σ_d X[row0 + ·][d]·Y[·][d];  roff[i]: the lane's byte
// offset of fragment step i inside the 16-row block (a constant of the lane: the swizzle key repeats every 8 rows)
__device__ __forceinline__ f32x4 escore(const char* img, int row0, const u32x4 (&yf)[2], const uint32_t (&roff)[2], f32x4 acc) {
#pragma unroll
  for (int i = 0; i < 2; ++i) acc = emma(*reinterpret_cast<const u32x4*>(img + row0 * EROW + roff[i]), yf[i], acc);
  return acc;
}

// acc[db][r] (d = 16·db + 4g + r, register side = lane & 15) += Σ_{32 rows from row0} X[row][d]·w[row][lane & 15]: w0 / w1 are
// the two 16-row tiles at row0 and row0 + 16; toff[db]: the lane's byte offset of its transposed 8-byte read in the block
__device__ __forceinline__ void epv(f32x4 (&acc)[4], f32x4 w0, f32x4 w1, const char* img, int row0, const uint32_t (&toff)[4]) {
  bf16x8 bfrag;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    bfrag[e] = (bf16_t)w0[e];
    bfrag[4 + e] = (bf16_t)w1[e];
  }
#pragma unroll
  for (int db = 0; db < 4; ++db) {
    const char* ap = img + row0 * EROW + toff[db];
    s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ap));
    s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ap + 16 * EROW));
    u32x2 u0 = __builtin_bit_cast(u32x2, t0), u1 = __builtin_bit_cast(u32x2, t1);
    u32x4 afrag = {u0[0], u0[1], u1[0], u1[1]};
    acc[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, afrag), bfrag, acc[db], 0, 0, 0);
  }
}

// the same with the two weight tiles already rounded to bf16 (4 values per tile in 2 registers)
__device__ __forceinline__ void epv_packed(f32x4 (&acc)[4], u32x2 w0, u32x2 w1, const char* img, int row0, const uint32_t (&toff)[4]) {
  const u32x4 bw = {w0[0], w0[1], w1[0], w1[1]};
  const bf16x8 bfrag = __builtin_bit_cast(bf16x8, bw);
#pragma unroll
  for (int db = 0; db < 4; ++db) {
    const char* ap = img + row0 * EROW + toff[db];
    s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ap));
    s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ap + 16 * EROW));
    u32x2 u0 = __builtin_bit_cast(u32x2, t0), u1 = __builtin_bit_cast(u32x2, t1);
    u32x4 afrag = {u0[0], u0[1], u1[0], u1[1]};
    acc[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, afrag), bfrag, acc[db], 0, 0, 0);
  }
}

// DPP move of a float inside 16-lane rows (CTRL: 0x100 + n row_shl:n — lane j reads lane j + n; 0x110 + n row_shr:n — lane j
// reads lane j − n); lanes whose source lies outside the row read 0
template <int CTRL>
__device__ __forceinline__ float edpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// The bias-gradient sums of one score tile in six instructions (hipcc emits v_mov_b32_dpp + v_add_f32 pairs for the builtin form: twelve):
// lo = d0 + shl1(d1) + shl2(d2) + shl3(d3), hs = shr15(d1) + shr14(d2) + shr13(d3) (row shifts inside 16-lane rows, lanes shifted in from
// outside a row read 0).  The leading s_nop covers the VALU-write → DPP-read hazard for operands written just before the statement (the
// hazard recogniser does not look inside inline asm).
__device__ __forceinline__ void edpp_diag_sums(float d0, float d1, float d2, float d3, float& lo, float& hs) {
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %3, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_mov_b32_dpp %1, %3 row_shr:15 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %4, %0 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %4, %1 row_shr:14 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %5, %0 row_shl:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %5, %1 row_shr:13 row_mask:0xf bank_mask:0xf bound_ctrl:1"
      : "=&v"(lo), "=&v"(hs)
      : "v"(d0), "v"(d1), "v"(d2), "v"(d3));
}
// acc += x moved down by N lanes inside its 16-lane row (row_shl:N — lane j reads lane j + N; N = 0: a plain add)
template <int N>
__device__ __forceinline__ void eacc_shl(float& acc, float x) {
  if constexpr (N == 0) acc += x;
  else if constexpr (N == 3) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %0 row_shl:3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(x));
  else if constexpr (N == 6) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %0 row_shl:6 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(x));
  else asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %0 row_shl:9 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(x));
}

// −inf for keys past the sequence's end in tile `t` (rows t·16 + 4g + r), 0 elsewhere
__device__ __forceinline__ f32x4 etail(int t, int g, int Lk) {
  f32x4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = (t * 16 + 4 * g + r < Lk) ? 0.f : -INFINITY;
  return v;
}

// Stage rows [0, nrows) of a strided [L, 64] bf16 tensor into a swizzled image by LDS-DMA (nrows a multiple of 8): piece p (8 image rows,
// 1 KiB) is issued by wave p mod NW; lane → (row, physical 16-byte chunk) of the piece, its source the logical chunk the swizzle puts
// there (eswz is its own inverse); rows >= L fall out of the descriptor's range: zero-filled.  Asynchronous: the caller waits
// (s_waitcnt vmcnt(0)) and barriers.
template <int NW>
__device__ __forceinline__ void estage_dma(char* img, const char* base, uint32_t stride_b, int nrows, int L, int wave, int lane) {
  const auto rs = lds_dma_rsrc(base, L > 0 ? (uint32_t)(L - 1) * stride_b + 128u : 0u);
  const int pr = lane >> 3, pc = lane & 7;
  for (int pz = wave; pz < (nrows >> 3); pz += NW) {
    const int row = pz * 8 + pr;
    lds_dma16(img + (row - pr) * EROW, rs, row < L ? (uint32_t)row * stride_b + (uint32_t)eswz(pc, row) * 16u : 0x80000000u);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// forward: 16 queries of one wave against the staged K / V, keys in chunks of at most 3 tile pairs (online softmax across the
// up to three chunks of a sequence of <= 256 keys: the score tiles of a chunk are 24 registers, the kernel fits 128)
// ---------------------------------------------------------------------------------------------------------------------
template <bool DROP, int NPC, typename TAIL>
__device__ __forceinline__ void efwd_chunk(const char* Kimg, const char* Vimg, int pair0, bool last, const TAIL& tail,
                                           const float* bptr, const u32x4 (&qf)[2], const ELane& el, uint32_t pblk, uint32_t dkey,
                                           const DropRow& drc, uint32_t t_hi, float& m_run, float& l_run, f32x4 (&oacc)[4]) {
  f32x4 s[2 * NPC];
#pragma unroll
  for (int t = 0; t < 2 * NPC; ++t) {
    const int row0 = (pair0 * 2 + t) * 16;
    f32x4 init = *reinterpret_cast<const f32x4*>(bptr + row0);
    if (t >= 2 * NPC - 2) {
      if (last) init += tail[t - (2 * NPC - 2)];
    }
    s[t] = escore(Kimg, row0, qf, el.roff, init);
  }
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < 2 * NPC; ++t) m = fmaxf(fmaxf(fmaxf(fmaxf(m, s[t][0]), s[t][1]), s[t][2]), s[t][3]);   // two v_max3_f32 per tile
  m = fmaxf(egroup_max(m), m_run);
  const float alpha = __builtin_amdgcn_exp2f((m_run - m) * LOG2E);       // first chunk: exp2(−inf) = 0
  const float mneg = -m * LOG2E;
  float psum = 0.f;
#pragma unroll
  for (int t = 0; t < 2 * NPC; ++t) {
    f32x4 pr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      pr[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t][r], LOG2E, mneg));
      psum += pr[r];
    }
    if (DROP) {
      bool kp[4];
      drop_keep_row(drop_base(pblk + (uint32_t)((pair0 * 2 + t) * 4), dkey), drc, t_hi, kp);
#pragma unroll
      for (int r = 0; r < 4; ++r) pr[r] = kp[r] ? pr[r] : 0.f;
    }
    s[t] = pr;
  }
  l_run = l_run * alpha + psum;           // per-lane partial (its 4 key columns); summed over the lane groups at the end
  m_run = m;
  if (pair0 > 0) {
#pragma unroll
    for (int db = 0; db < 4; ++db) oacc[db] *= alpha;
  }
#pragma unroll
  for (int p = 0; p < NPC; ++p) epv(oacc, s[2 * p], s[2 * p + 1], Vimg, (pair0 + p) * 32, el.toff);
}

// (experiments build) enc_fwd_c_kernel under the clock: per workgroup (< 4096) and wave eight 64-bit words in a.dbg_buf —
// 0 entry, 1 everything requested (K / V DMA, Q, bias copies), 2 landed + barrier passed, 3 first query block done, 4 exit,
// 5 HW_ID[15:0] | XCC_ID << 16, 6 the scalar-load chain done (DMA issue starts), 7 (Lq << 16 | query blocks of this wave) — read back by tools/attn_fwd_stamps.py (LAKO_ATTN_DEBUG bit 262144)
#ifdef LAKO_EXPERIMENTS
#define FWD_STAMP(P, VAL)                                                                                                    \
  do {                                                                                                                       \
    if (a.dbg_buf && (threadIdx.x & 63) == 0) {                                                                              \
      const int wg_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;                                        \
      if (wg_ < 4096) reinterpret_cast<unsigned long long*>(a.dbg_buf)[(wg_ * 8 + wave) * 8 + (P)] = (VAL);                  \
    }                                                                                                                        \
  } while (0)
#else
#define FWD_STAMP(P, VAL) do {} while (0)
#endif
constexpr int FWD_NW = 8;     // waves per forward workgroup: 2 workgroups (133 KB of LDS) = 16 waves per CU, 4 per SIMD
// LDS: K image | V image | 4 bias copies
__host__ __device__ constexpr int efwd_lds(int rows) { return 2 * rows * EROW + 4 * EB_ST * 4; }

template <bool DROP>
__global__ __launch_bounds__(FWD_NW * 64, FWD_NW / 2) void enc_fwd_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int R = a.chunk_rows;
  char* Kimg = smem;
  char* Vimg = smem + R * EROW;
  float* b4 = reinterpret_cast<float*>(smem + 2 * R * EROW);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  const int b = attn_seq(a, blockIdx.z), h = blockIdx.y;
  const int q0 = a.q_off ? __builtin_amdgcn_readfirstlane(a.q_off[b]) : 0;
  const int k0 = a.k_off ? __builtin_amdgcn_readfirstlane(a.k_off[b]) : 0;
  const int Lq = a.q_off ? __builtin_amdgcn_readfirstlane(a.q_off[b + 1]) - q0 : a.Lq;
  const int Lk = a.k_off ? __builtin_amdgcn_readfirstlane(a.k_off[b + 1]) - k0 : a.Lk;
  const int qb_begin = blockIdx.x * a.blocks_per_wg;
  const int qb_end = min((Lq + 15) >> 4, qb_begin + a.blocks_per_wg);
  if (qb_begin >= qb_end || Lk <= 0) return;            // nothing to write (ragged: a short or empty sequence)
  const int64_t hoff = (int64_t)h * 64;
  const char* qbase = a.q + ((a.q_off ? (int64_t)q0 * a.qst : (int64_t)b * a.qsb) + hoff) * 2;
  const char* kbase = a.k + ((a.k_off ? (int64_t)k0 * a.kst : (int64_t)b * a.ksb) + hoff) * 2;
  const char* vbase = a.v + ((a.k_off ? (int64_t)k0 * a.vst : (int64_t)b * a.vsb) + hoff) * 2;
  char* obase = a.out + ((a.q_off ? (int64_t)q0 * a.ost : (int64_t)b * a.osb) + hoff) * 2;
  const int np = (Lk + 31) >> 5;                        // tile pairs; only the last can hold keys past the end
  const int nk = np << 5;
  estage<FWD_NW * 64>(Kimg, kbase, a.kst * 2, nk, Lk);
  estage<FWD_NW * 64>(Vimg, vbase, a.vst * 2, nk, Lk);
  estage_bias<FWD_NW * 64>(b4, a.rel_bias, h, a.R, false);
  __syncthreads();
  const ELane el = elane(lane);
  const uint32_t t_hi = a.drop_t16 << 16;
  const int KB = (a.Lk + 3) >> 2, QB = (a.Lq + 3) >> 2;
  const float oscale = DROP ? a.drop_scale : 1.0f;
  const f32x4 tail[2] = {etail(2 * np - 2, g, Lk), etail(2 * np - 1, g, Lk)};
  // chunks of at most CH tile pairs: [0, n0), [CH, CH + n1), [2·CH, 2·CH + n2).  CH = 3 with dropout (24 registers of score tiles
  // beside the hash state: 135 µs per config-2 launch against 145 at CH = 4, which spills), CH = 4 without (107 µs against 200:
  // the compiler's schedule of the three-chunk form serialises there)
  constexpr int CH = DROP ? 3 : 4;
  const int n0 = min(np, CH), n1 = min(max(np - CH, 0), CH), n2 = max(np - 2 * CH, 0);

  u32x4 qf_next[2];
  auto request_q = [&](int qbn) { eload_frags(qf_next, qbase, a.qst * 2, qbn < qb_end ? qbn * 16 + l15 : Lq, Lq, g); };
  request_q(qb_begin + wave);
  for (int qb = qb_begin + wave; qb < qb_end; qb += FWD_NW) {
    const int qi = qb * 16 + l15;
    u32x4 qf[2] = {qf_next[0], qf_next[1]};
    request_q(qb + FWD_NW);
    // bias: tile row0, lane (query qi, group g), register r reads T[row0 + 4g + r − qi + rel_off]; the copy and the aligned
    // base are constants of the block (row0 is a multiple of 16)
    const int i0 = 4 * g - qi + a.rel_off + EB_PADLO;
    const float* bptr = b4 + (i0 & 3) * EB_ST + (i0 & ~3);
    const uint32_t pblk = (uint32_t)((b * a.H + h) * QB + (qi >> 2)) * (uint32_t)KB + (uint32_t)g;
    const DropRow drc = drop_row_consts(qi);
    float m = -INFINITY, l = 0.f;
    f32x4 oacc[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) oacc[db] = f32x4{0.f, 0.f, 0.f, 0.f};
#define ECH(N, P0, LAST) case N: if constexpr (N <= CH) efwd_chunk<DROP, N>(Kimg, Vimg, P0, LAST, tail, bptr, qf, el, pblk, a.drop_key, drc, t_hi, m, l, oacc); break;
    switch (n0) { ECH(1, 0, n1 == 0) ECH(2, 0, n1 == 0) ECH(3, 0, n1 == 0) ECH(4, 0, n1 == 0) default: break; }
    switch (n1) { ECH(1, CH, n2 == 0) ECH(2, CH, n2 == 0) ECH(3, CH, n2 == 0) ECH(4, CH, n2 == 0) default: break; }
    if constexpr (CH == 3) {
      switch (n2) { ECH(1, 2 * CH, true) ECH(2, 2 * CH, true) default: break; }
    }
#undef ECH
    l = egroup_sum(l);
    if (qi < Lq) {
      const float inv = 1.0f / l;
      const float f = inv * oscale;
      bf16_t* op = reinterpret_cast<bf16_t*>(obase + (int64_t)qi * a.ost * 2);
#pragma unroll
      for (int db = 0; db < 4; ++db) store4(op + db * 16 + 4 * g, oacc[db] * f);
      if (g == 0 && a.stats) {
        float* st = a.stats + (((int64_t)b * a.H + h) * a.Lq + qi) * 4;
        st[0] = m;
        st[1] = inv;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// forward, per (sequence, head) item as enc_fwd_kernel, with the K / V images arriving by LDS-DMA (round 4).  enc_fwd_kernel stages
// K, then V, through registers and then the bias copies — three dependent memory round trips that nothing in the workgroup overlaps —
// before its first MFMA; here every wave issues its share of the K and V pieces first (1 KiB = 8 image rows per wave-instruction, the
// swizzle applied to the per-lane SOURCE address, rows past the sequence's end zero-filled by the descriptor's range check), then
// requests the bias table and its first Q fragments: everything is in flight together, ONE round trip before the first MFMA.
// (A version that let the first query block start on key chunk 0 while chunks 1 – 2 were still landing needed the Q fragments loaded
//  by inline asm — a compiler-visible load is waited for with vmcnt(0), i.e. for every DMA piece — and the dropout instantiation, at its
//  128-register cap, SPILLED those registers before the load had landed: garbage.  Not kept.)
// Same arithmetic per block (efwd_chunk): results are bit-identical to enc_fwd_kernel.
// ---------------------------------------------------------------------------------------------------------------------
template <bool DROP>
__global__ __launch_bounds__(FWD_NW * 64, FWD_NW / 2) void enc_fwd_c_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int R = a.chunk_rows;
  char* Kimg = smem;
  char* Vimg = smem + R * EROW;
  float* b4 = reinterpret_cast<float*>(smem + 2 * R * EROW);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  FWD_STAMP(0, __builtin_amdgcn_s_memtime());
  const int b = attn_seq(a, blockIdx.z), h = blockIdx.y;
  // (the ragged launch reads its four offsets in ONE batch of scalar loads: one by one behind a null test each, as the general form below
  //  compiles, they were four dependent round trips of ≈ 900 cycles at the start of every workgroup — profiles/r05t_attn_fwd_stamps.txt)
  int q0 = 0, k0 = 0, Lq = a.Lq, Lk = a.Lk;
  if (a.q_off != nullptr && a.k_off != nullptr) {
    const int qa = a.q_off[b], qe = a.q_off[b + 1], ka = a.k_off[b], ke = a.k_off[b + 1];
    q0 = __builtin_amdgcn_readfirstlane(qa);
    k0 = __builtin_amdgcn_readfirstlane(ka);
    Lq = __builtin_amdgcn_readfirstlane(qe) - q0;
    Lk = __builtin_amdgcn_readfirstlane(ke) - k0;
  } else {
    if (a.q_off) {
      q0 = __builtin_amdgcn_readfirstlane(a.q_off[b]);
      Lq = __builtin_amdgcn_readfirstlane(a.q_off[b + 1]) - q0;
    }
    if (a.k_off) {
      k0 = __builtin_amdgcn_readfirstlane(a.k_off[b]);
      Lk = __builtin_amdgcn_readfirstlane(a.k_off[b + 1]) - k0;
    }
  }
  const int qb_begin = blockIdx.x * a.blocks_per_wg;
  const int qb_end = min((Lq + 15) >> 4, qb_begin + a.blocks_per_wg);
  if (qb_begin >= qb_end || Lk <= 0) return;
  const int64_t hoff = (int64_t)h * 64;
  const char* qbase = a.q + ((a.q_off ? (int64_t)q0 * a.qst : (int64_t)b * a.qsb) + hoff) * 2;
  const char* kbase = a.k + ((a.k_off ? (int64_t)k0 * a.kst : (int64_t)b * a.ksb) + hoff) * 2;
  const char* vbase = a.v + ((a.k_off ? (int64_t)k0 * a.vst : (int64_t)b * a.vsb) + hoff) * 2;
  char* obase = a.out + ((a.q_off ? (int64_t)q0 * a.ost : (int64_t)b * a.osb) + hoff) * 2;
  const int np = (Lk + 31) >> 5;
  FWD_STAMP(6, __builtin_amdgcn_s_memtime());        // (the sequence id and its offsets are known: the scalar-load chain is behind us)
  estage_dma<FWD_NW>(Kimg, kbase, (uint32_t)a.kst * 2u, np << 5, Lk, wave, lane);
  estage_dma<FWD_NW>(Vimg, vbase, (uint32_t)a.vst * 2u, np << 5, Lk, wave, lane);
  u32x4 qf_next[2];
  auto request_q = [&](int qbn) { eload_frags(qf_next, qbase, a.qst * 2, qbn < qb_end ? qbn * 16 + l15 : Lq, Lq, g); };
  request_q(qb_begin + wave);
  estage_bias<FWD_NW * 64>(b4, a.rel_bias, h, a.R, false);
  FWD_STAMP(1, __builtin_amdgcn_s_memtime());
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the wave's DMA pieces have landed (the compiler does not count them)
  __syncthreads();
  FWD_STAMP(2, __builtin_amdgcn_s_memtime());
  int nblk_ = 0;
  const ELane el = elane(lane);
  const uint32_t t_hi = a.drop_t16 << 16;
  const int KB = (a.Lk + 3) >> 2, QB = (a.Lq + 3) >> 2;
  const float oscale = DROP ? a.drop_scale : 1.0f;
  const f32x4 tail[2] = {etail(2 * np - 2, g, Lk), etail(2 * np - 1, g, Lk)};
  constexpr int CH = DROP ? 3 : 4;
  const int n0 = min(np, CH), n1 = min(max(np - CH, 0), CH), n2 = max(np - 2 * CH, 0);
  for (int qb = qb_begin + wave; qb < qb_end; qb += FWD_NW) {
    const int qi = qb * 16 + l15;
    u32x4 qf[2] = {qf_next[0], qf_next[1]};
    request_q(qb + FWD_NW);
    const int i0 = 4 * g - qi + a.rel_off + EB_PADLO;
    const float* bptr = b4 + (i0 & 3) * EB_ST + (i0 & ~3);
    const uint32_t pblk = (uint32_t)((b * a.H + h) * QB + (qi >> 2)) * (uint32_t)KB + (uint32_t)g;
    const DropRow drc = drop_row_consts(qi);
    float m = -INFINITY, l = 0.f;
    f32x4 oacc[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) oacc[db] = f32x4{0.f, 0.f, 0.f, 0.f};
#define ECH(N, P0, LAST) case N: if constexpr (N <= CH) efwd_chunk<DROP, N>(Kimg, Vimg, P0, LAST, tail, bptr, qf, el, pblk, a.drop_key, drc, t_hi, m, l, oacc); break;
    switch (n0) { ECH(1, 0, n1 == 0) ECH(2, 0, n1 == 0) ECH(3, 0, n1 == 0) ECH(4, 0, n1 == 0) default: break; }
    switch (n1) { ECH(1, CH, n2 == 0) ECH(2, CH, n2 == 0) ECH(3, CH, n2 == 0) ECH(4, CH, n2 == 0) default: break; }
    if constexpr (CH == 3) {
      switch (n2) { ECH(1, 2 * CH, true) ECH(2, 2 * CH, true) default: break; }
    }
#undef ECH
    l = egroup_sum(l);
    if (nblk_++ == 0) { asm volatile("" :: "v"(oacc[0]), "v"(l)); FWD_STAMP(3, __builtin_amdgcn_s_memtime()); }
    if (qi < Lq) {
      const float inv = 1.0f / l;
      const float f = inv * oscale;
      bf16_t* op = reinterpret_cast<bf16_t*>(obase + (int64_t)qi * a.ost * 2);
#pragma unroll
      for (int db = 0; db < 4; ++db) store4(op + db * 16 + 4 * g, oacc[db] * f);
      if (g == 0 && a.stats) {
        float* st = a.stats + (((int64_t)b * a.H + h) * a.Lq + qi) * 4;
        st[0] = m;
        st[1] = inv;
      }
    }
  }
  FWD_STAMP(4, __builtin_amdgcn_s_memtime());
  FWD_STAMP(5, (unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xFFFFu) | ((unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xFu) << 16));
  FWD_STAMP(7, ((unsigned long long)Lq << 16) | (unsigned long long)nblk_);
}

// ---------------------------------------------------------------------------------------------------------------------
// forward, PERSISTENT form (round 3).  The kernel above starts one workgroup per (sequence, head): 3 840 workgroups per config-2
// launch, each of which stages its K / V images (a global → register → LDS round trip nothing overlaps), then walks ⌈L/16⌉ ≈ 7…13
// query blocks with 8 waves — a second pass over the waves for 9…13 blocks with most of them idle (1.7 passes per item on the
// benchmark's length distribution, 1.0 needed).  Here ONE 16-wave workgroup per CU walks a list of items of ONE head:
//   * wave w owns query block w of every item (≤ 256 queries = 16 blocks: always one pass; waves past the item's last block only
//     help with the staging);
//   * the K / V images of item i + 1 arrive by LDS-DMA (buffer_load … lds, the swizzle applied to the per-lane SOURCE address,
//     rows past the sequence's end zero-filled by the descriptor's range check) into the second image pair while item i is
//     computed; one s_waitcnt + one barrier per item;
//   * the four shifted bias copies are staged once per workgroup (the head never changes), Q fragments are requested an item ahead.
// Same arithmetic per query block as the kernel above (efwd_chunk, the same chunking): results are bit-identical.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int PF_NW = 16;
// LDS: 2 × (K image | V image) | 4 bias copies
__host__ __device__ constexpr int efwdp_lds(int rows) { return 4 * rows * EROW + 4 * EB_ST * 4; }

template <bool DROP>
__global__ __launch_bounds__(PF_NW * 64) void enc_fwd_p_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int R = a.chunk_rows;                       // image rows (multiple of 32, >= every sequence's key count rounded up to 32)
  float* b4 = reinterpret_cast<float*>(smem + 4 * R * EROW);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  const int h = blockIdx.x % a.H, slot = blockIdx.x / a.H, nslots = gridDim.x / a.H;
  if (slot >= a.Bn) return;
  estage_bias<PF_NW * 64>(b4, a.rel_bias, h, a.R, false);
  const ELane el = elane(lane);
  const uint32_t t_hi = a.drop_t16 << 16;
  const int KB = (a.Lk + 3) >> 2, QB = (a.Lq + 3) >> 2;
  const float oscale = DROP ? a.drop_scale : 1.0f;
  const int64_t hoff = (int64_t)h * 64;
  constexpr int CH = DROP ? 3 : 4;
  // DMA pieces (1 KiB = 8 image rows) of an image pair: K rows, then V rows; this wave issues pieces wave, wave + 16, … (always all
  // of them, so that the image rows past a short sequence's end are rewritten with zeros).  Lane → (row, physical chunk) of the
  // piece; its source is the logical chunk that the swizzle puts there.
  const int ppi = R >> 3;                           // pieces per image
  auto item_geom = [&](int b, int& q0, int& k0, int& Lq, int& Lk) {
    q0 = a.q_off ? __builtin_amdgcn_readfirstlane(a.q_off[b]) : 0;
    k0 = a.k_off ? __builtin_amdgcn_readfirstlane(a.k_off[b]) : 0;
    Lq = a.q_off ? __builtin_amdgcn_readfirstlane(a.q_off[b + 1]) - q0 : a.Lq;
    Lk = a.k_off ? __builtin_amdgcn_readfirstlane(a.k_off[b + 1]) - k0 : a.Lk;
  };
  auto issue = [&](int b, char* buf) {
    int q0, k0, Lq, Lk;
    item_geom(b, q0, k0, Lq, Lk);
    const char* kbase = a.k + ((a.k_off ? (int64_t)k0 * a.kst : (int64_t)b * a.ksb) + hoff) * 2;
    const char* vbase = a.v + ((a.k_off ? (int64_t)k0 * a.vst : (int64_t)b * a.vsb) + hoff) * 2;
    const uint32_t kstb = (uint32_t)a.kst * 2u, vstb = (uint32_t)a.vst * 2u;
    const auto krs = lds_dma_rsrc(kbase, Lk > 0 ? (uint32_t)(Lk - 1) * kstb + 128u : 0u);
    const auto vrs = lds_dma_rsrc(vbase, Lk > 0 ? (uint32_t)(Lk - 1) * vstb + 128u : 0u);
    int le = lane;
    asm volatile("" : "+v"(le));                    // rebuilt per call: as loop invariants the offsets would stay live across the items
    const int pr = le >> 3, pc = le & 7;
    for (int p = wave; p < 2 * ppi; p += PF_NW) {
      const bool isv = p >= ppi;
      const int row = (isv ? p - ppi : p) * 8 + pr;
      const uint32_t voff = (uint32_t)row * (isv ? vstb : kstb) + (uint32_t)eswz(pc, row) * 16u;   // eswz is its own inverse
      lds_dma16(buf + (isv ? R * EROW : 0) + (row - pr) * EROW, isv ? vrs : krs, row < Lk ? voff : 0x80000000u);
    }
  };
  u32x4 qf_next[2];
  auto request_q = [&](int b) {
    int q0, k0, Lq, Lk;
    item_geom(b, q0, k0, Lq, Lk);
    const char* qbase = a.q + ((a.q_off ? (int64_t)q0 * a.qst : (int64_t)b * a.qsb) + hoff) * 2;
    eload_frags_buf(qf_next, qbase, (uint32_t)a.qst * 2u, wave * 16, Lq, (uint32_t)l15 * (uint32_t)a.qst * 2u + (uint32_t)g * 16u);
  };
  issue(attn_seq(a, slot), smem);
  request_q(attn_seq(a, slot));
  int cur = 0;
  for (int bi = slot; bi < a.Bn; bi += nslots) {
    const int b = attn_seq(a, bi);
    char* Kimg = smem + cur * 2 * R * EROW;
    char* Vimg = Kimg + R * EROW;
    u32x4 qf[2] = {qf_next[0], qf_next[1]};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of item b (and its Q fragments) have landed
    __syncthreads();                                         // … everybody's; and every wave is done with the other image pair
    if (bi + nslots < a.Bn) {
      issue(attn_seq(a, bi + nslots), smem + (cur ^ 1) * 2 * R * EROW);
      request_q(attn_seq(a, bi + nslots));
    }
    cur ^= 1;
    int q0, k0, Lq, Lk;
    item_geom(b, q0, k0, Lq, Lk);
    if (wave * 16 >= Lq || Lk <= 0) continue;               // (wave-uniform) no query block for this wave / an empty sequence
    char* obase = a.out + ((a.q_off ? (int64_t)q0 * a.ost : (int64_t)b * a.osb) + hoff) * 2;
    const int np = (Lk + 31) >> 5;
    // (−inf for the keys past the sequence's end, built where the last chunk uses it: as two live vectors it costs 8 registers)
    struct { int np, g, Lk; __device__ __forceinline__ f32x4 operator[](int u) const { return etail(2 * np - 2 + u, g, Lk); } } tail{np, g, Lk};
    const int n0 = min(np, CH), n1 = min(max(np - CH, 0), CH), n2 = max(np - 2 * CH, 0);
    const int qi = wave * 16 + l15;
    const int i0 = 4 * g - qi + a.rel_off + EB_PADLO;
    const float* bptr = b4 + (i0 & 3) * EB_ST + (i0 & ~3);
    const uint32_t pblk = (uint32_t)((b * a.H + h) * QB + (qi >> 2)) * (uint32_t)KB + (uint32_t)g;
    const DropRow drc = drop_row_consts(qi);
    float m = -INFINITY, l = 0.f;
    f32x4 oacc[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) oacc[db] = f32x4{0.f, 0.f, 0.f, 0.f};
#define ECH(N, P0, LAST) case N: if constexpr (N <= CH) efwd_chunk<DROP, N>(Kimg, Vimg, P0, LAST, tail, bptr, qf, el, pblk, a.drop_key, drc, t_hi, m, l, oacc); break;
    switch (n0) { ECH(1, 0, n1 == 0) ECH(2, 0, n1 == 0) ECH(3, 0, n1 == 0) ECH(4, 0, n1 == 0) default: break; }
    switch (n1) { ECH(1, CH, n2 == 0) ECH(2, CH, n2 == 0) ECH(3, CH, n2 == 0) ECH(4, CH, n2 == 0) default: break; }
    if constexpr (CH == 3) {
      switch (n2) { ECH(1, 2 * CH, true) ECH(2, 2 * CH, true) default: break; }
    }
#undef ECH
    l = egroup_sum(l);
    if (qi < Lq) {
      const float inv = 1.0f / l;
      const float f = inv * oscale;
      bf16_t* op = reinterpret_cast<bf16_t*>(obase + (int64_t)qi * a.ost * 2);
#pragma unroll
      for (int db = 0; db < 4; ++db) store4(op + db * 16 + 4 * g, oacc[db] * f);
      if (g == 0 && a.stats) {
        float* st = a.stats + (((int64_t)b * a.H + h) * a.Lq + qi) * 4;
        st[0] = m;
        st[1] = inv;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward, dQ pass: K / V in LDS — staged ONCE per (sequence, head) —, a wave owns the query blocks wave, wave + 4, wave + 8,
// wave + 12 of the sequence (static slots); Sᵀ[key][query] and dPᵀ tiles → dSᵀ → dQᵀ += Kᵀ·dSᵀ.
// The bias gradient (Σ dS over batch rows and score diagonals) is accumulated in registers over the workgroup's batch rows:
// tile (slot s, key tile t) of wave w lies on tile diagonal t − (w + 4s), so the accumulators are indexed by t − 4s + 12
// (28 of them — every index static).  Inside a tile the element of lane (query l15, group g), register r sits on diagonal
// key − query = 16·Δ + 4g + (r − l15): a DPP row shift by r brings the four registers of a 16-lane row onto a common
// r − l15, so a tile costs TWO accumulator registers (lo: r − l15 = −j ≤ 0 at lane j; hi: r − l15 = 16 − j > 0 at lanes
// 13..15) instead of four — 56 registers for the 28 tile diagonals, which leaves room to request the next block's
// operands (Q, dO, O fragments, statistics) while the current one is computed.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int DQ_NACC = 28;
// LDS: K | V | 4 bias copies | drel_l[512] | offsets
__host__ __device__ constexpr int ebwd0_lds(int rows) { return 2 * rows * EROW + 4 * EB_ST * 4 + 512 * 4 + (4 + 2 * OFFS_MAX) * 4; }

template <bool DROP>
__global__ __launch_bounds__(256, 2) void enc_bwd_dq_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int R = a.chunk_rows;
  char* Kimg = smem;
  char* Vimg = smem + R * EROW;
  float* b4 = reinterpret_cast<float*>(smem + 2 * R * EROW);
  float* drel_l = b4 + 4 * EB_ST;
  int* offs_l = reinterpret_cast<int*>(drel_l + 512) + 4;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  const int h = blockIdx.y;
  const int b_begin = blockIdx.z * a.bn_per_wg, b_end = min(a.Bn, b_begin + a.bn_per_wg);
  if (threadIdx.x <= b_end - b_begin) {
    if (a.q_off) offs_l[threadIdx.x] = a.q_off[b_begin + threadIdx.x];
    if (a.k_off) offs_l[OFFS_MAX + threadIdx.x] = a.k_off[b_begin + threadIdx.x];
  }
  const bool want_drel = a.drel != nullptr;
  for (int i = threadIdx.x; i < 512; i += 256) drel_l[i] = 0.f;
  estage_bias(b4, a.rel_bias, h, a.R, false);
  __syncthreads();
  const ELane el = elane(lane);
  const uint32_t t_hi = a.drop_t16 << 16;
  const int KB = (a.Lk + 3) >> 2, QB = (a.Lq + 3) >> 2;
  const float dscale = DROP ? a.drop_scale : 1.0f, inv_dscale = 1.0f / dscale;
  float acc_lo[DQ_NACC], acc_hi[DQ_NACC];
#pragma unroll
  for (int t = 0; t < DQ_NACC; ++t) acc_lo[t] = acc_hi[t] = 0.f;

  for (int b = b_begin; b < b_end; ++b) {      // (this kernel walks CONSECUTIVE sequences: a.order is not applied)
    const int q0 = a.q_off ? __builtin_amdgcn_readfirstlane(offs_l[b - b_begin]) : 0;
    const int k0 = a.k_off ? __builtin_amdgcn_readfirstlane(offs_l[OFFS_MAX + b - b_begin]) : 0;
    const int Lq = a.q_off ? __builtin_amdgcn_readfirstlane(offs_l[b - b_begin + 1]) - q0 : a.Lq;
    const int Lk = a.k_off ? __builtin_amdgcn_readfirstlane(offs_l[OFFS_MAX + b - b_begin + 1]) - k0 : a.Lk;
    if (Lq <= 0 || Lk <= 0) continue;                      // (workgroup-uniform) an empty sequence
    const int64_t hoff = (int64_t)h * 64;
    const char* qbase = a.q + ((a.q_off ? (int64_t)q0 * a.qst : (int64_t)b * a.qsb) + hoff) * 2;
    const char* kbase = a.k + ((a.k_off ? (int64_t)k0 * a.kst : (int64_t)b * a.ksb) + hoff) * 2;
    const char* vbase = a.v + ((a.k_off ? (int64_t)k0 * a.vst : (int64_t)b * a.vsb) + hoff) * 2;
    const char* obase = a.o + ((a.q_off ? (int64_t)q0 * a.ost : (int64_t)b * a.osb) + hoff) * 2;
    const char* dobase = a.dout + ((a.q_off ? (int64_t)q0 * a.ost : (int64_t)b * a.osb) + hoff) * 2;
    char* dqbase = a.dq + ((a.q_off ? (int64_t)q0 * a.qst : (int64_t)b * a.qsb) + hoff) * 2;
    float* stats = a.stats + ((int64_t)b * a.H + h) * a.Lq * 4;
    const int np = (Lk + 31) >> 5, nk = np << 5;
    const int nqb = (Lq + 15) >> 4;
    // operands of the wave's next query block: requested one block ahead (the first before the staging barrier)
    u32x4 qf_n[2], dof_n[2], of_n[2];
    float m_n = 0.f, il_n = 0.f;
    auto request = [&](int qbn) {
      const int qn = qbn < nqb ? qbn * 16 + l15 : Lq;
      eload_frags(qf_n, qbase, a.qst * 2, qn, Lq, g);
      eload_frags(dof_n, dobase, a.ost * 2, qn, Lq, g);
      eload_frags(of_n, obase, a.ost * 2, qn, Lq, g);
      m_n = il_n = 0.f;
      if (qn < Lq) {
        m_n = stats[qn * 4];
        il_n = stats[qn * 4 + 1];
      }
    };
    request(wave);
    __syncthreads();                                        // every wave is done with the previous row's images
    if (!ATTN_DBG(a, 16)) {
      estage(Kimg, kbase, a.kst * 2, nk, Lk);
      estage(Vimg, vbase, a.vst * 2, nk, Lk);
    }
    __syncthreads();
    const f32x4 tail[2] = {etail(2 * np - 2, g, Lk), etail(2 * np - 1, g, Lk)};
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
      const int qb = wave + 4 * sl;
      if (qb < nqb) {                                       // (wave-uniform; a guard, not a break: see the pair loop)
      const int qi = qb * 16 + l15;
      u32x4 qf[2] = {qf_n[0], qf_n[1]}, dof[2] = {dof_n[0], dof_n[1]};
      const float st_m = m_n, st_il = il_n;
      // δ = rowsum(dO ∘ O), handed to the dK/dV pass through the statistics
      float part = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        bf16x8 dv = __builtin_bit_cast(bf16x8, dof_n[i]), ov = __builtin_bit_cast(bf16x8, of_n[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) part += (float)dv[e] * (float)ov[e];
      }
      if (sl < 3) request(qb + 4);
      const float delta = egroup_sum(part);
      if (qi < Lq && g == 0) stats[qi * 4 + 2] = delta;
      // p̃ = scale·p = exp2(s·log2e − lse2) with lse2 = m·log2e + log2(l) − log2(scale);  p = 0 for rows past Lq (1/l = 0)
      const float lse2 = st_il > 0.f ? st_m * LOG2E - __builtin_amdgcn_logf(st_il * dscale) : INFINITY;
      const float ndel = -delta * inv_dscale;               // dP′ = dP − δ/scale straight out of the MFMA
      const int i0 = 4 * g - qi + a.rel_off + EB_PADLO;
      int boff = (i0 & 3) * EB_ST + (i0 & ~3);
      asm volatile("" : "+v"(boff));                        // keeps the bias reads at their use (hoisted they cost 64 registers)
      const float* bptr = b4 + boff;
      const uint32_t pblk = (uint32_t)((b * a.H + h) * QB + (qi >> 2)) * (uint32_t)KB + (uint32_t)g;
      const DropRow drc = drop_row_consts(qi);
      f32x4 dq[4];
#pragma unroll
      for (int db = 0; db < 4; ++db) dq[db] = f32x4{0.f, 0.f, 0.f, 0.f};
      // every pair is a guarded block that finishes its own accumulations: with `break`s in the unrolled loop the compiler sinks
      // the bias-gradient adds of ALL pairs into the exit paths and keeps their operands live until then (500 bytes of spills)
#pragma unroll
      for (int tp = 0; tp < 8; ++tp) {
        if (tp < np && !ATTN_DBG(a, 32)) {
        f32x4 ds[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int t = 2 * tp + u, row0 = t * 16;
          f32x4 init = *reinterpret_cast<const f32x4*>(bptr + row0);
          if (tp == np - 1) init += tail[u];
          const f32x4 sv = escore(Kimg, row0, qf, el.roff, init);
          f32x4 dp = escore(Vimg, row0, dof, el.roff, f32x4{ndel, ndel, ndel, ndel});
          f32x4 p;
#pragma unroll
          for (int r = 0; r < 4; ++r) p[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[r], LOG2E, -lse2));
          if (DROP) {
            bool kp[4];
            drop_keep_row(drop_base(pblk + (uint32_t)(t * 4), a.drop_key), drc, t_hi, kp);
#pragma unroll
            for (int r = 0; r < 4; ++r) dp[r] = kp[r] ? dp[r] : ndel;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) ds[u][r] = p[r] * dp[r];      // dS = P ∘ (dropped dP − δ)
          // bias gradient: shift register r left by r lanes inside its 16-lane row (lanes shifted in from outside the row: 0)
          const int ai = t - 4 * sl + 12;
          acc_lo[ai] += ds[u][0] + edpp<0x101>(ds[u][1]) + edpp<0x102>(ds[u][2]) + edpp<0x103>(ds[u][3]);     // row_shl:1..3
          acc_hi[ai] += edpp<0x11F>(ds[u][1]) + edpp<0x11E>(ds[u][2]) + edpp<0x11D>(ds[u][3]);               // row_shr:15..13
        }
        epv(dq, ds[0], ds[1], Kimg, tp * 32, el.toff);              // dQᵀ += Kᵀ·dSᵀ
        __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (qi < Lq) {
        bf16_t* op = reinterpret_cast<bf16_t*>(dqbase + (int64_t)qi * a.qst * 2);
#pragma unroll
        for (int db = 0; db < 4; ++db) store4(op + db * 16 + 4 * g, dq[db]);
      }
      }
    }
  }
  if (want_drel) {
    // accumulator i of wave w, lane (j = lane & 15, g): lo is the sum over key − query = 16·(i − 12 − w) + 4g − j, hi (lanes 13..15)
    // over 16·(i − 12 − w) + 4g + 16 − j.  Once per workgroup: LDS atomics into the head's table, then one global atomic per entry.
#pragma unroll
    for (int i = 0; i < DQ_NACC; ++i) {
      const int d0 = 16 * (i - 12 - wave) + 4 * g - l15 + a.rel_off;
      if (acc_lo[i] != 0.f && d0 >= 0 && d0 < a.R) atomicAdd(&drel_l[d0], acc_lo[i]);
      if (acc_hi[i] != 0.f && d0 + 16 >= 0 && d0 + 16 < a.R) atomicAdd(&drel_l[d0 + 16], acc_hi[i]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < a.R; i += 256) {
      const float v = drel_l[i];
      if (v != 0.f) atomicAdd(a.drel + (int64_t)h * a.R + i, v);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward, dQ pass, PERSISTENT form (round 3) — as enc_fwd_p_kernel: one 16-wave workgroup per CU walks the sequences of ONE
// head, wave w owns query block w of every item, the K / V images of the next item arrive by LDS-DMA under the current item's
// arithmetic.  A wave's tiles (query block w, key tile t) lie on tile diagonal t − w whatever the item, so the bias-gradient
// accumulators are indexed by t alone (16 × 2 registers, summed over all the workgroup's items) and flushed once at the end.
// The Q / dO / O fragments and the row statistics of the item are requested right after the barrier, before the next item's DMA
// is issued (an item ahead they would cost 19 registers across the key loop; ≤ 128 here).  Same arithmetic per query block as
// enc_bwd_dq_kernel; the bias gradient is summed in another order (float atomics anyway).
// ---------------------------------------------------------------------------------------------------------------------
// LDS: 2 × (K image | V image) | 4 bias copies | drel_l[512]
__host__ __device__ constexpr int ebwd0p_lds(int rows) { return 4 * rows * EROW + 4 * EB_ST * 4 + 512 * 4; }

template <bool DROP>
__global__ __launch_bounds__(PF_NW * 64) void enc_bwd_dq_p_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int R = a.chunk_rows;
  float* b4 = reinterpret_cast<float*>(smem + 4 * R * EROW);
  float* drel_l = b4 + 4 * EB_ST;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  const int h = blockIdx.x % a.H, slot = blockIdx.x / a.H, nslots = gridDim.x / a.H;
  if (slot >= a.Bn) return;
  const bool want_drel = a.drel != nullptr;
  for (int i = threadIdx.x; i < 512; i += PF_NW * 64) drel_l[i] = 0.f;
  estage_bias<PF_NW * 64>(b4, a.rel_bias, h, a.R, false);
  const ELane el = elane(lane);
  const uint32_t t_hi = a.drop_t16 << 16;
  const int KB = (a.Lk + 3) >> 2, QB = (a.Lq + 3) >> 2;
  const float dscale = DROP ? a.drop_scale : 1.0f, inv_dscale = 1.0f / dscale;
  const int64_t hoff = (int64_t)h * 64;
  const int ppi = R >> 3;
  // acc_lo[t]: lane j of a 16-lane row sums the tile's elements with register − query = −j; the elements that wrap around the row
  // (register − query = 16 − j, lanes 13..15 only) of the FOUR tiles 4u … 4u + 3 share acc_hi[u]: tile 4u + k's three lanes are
  // moved down by 3k lanes (one more DPP shift; the lanes it leaves are zero), 4 registers instead of 16
  float acc_lo[16], acc_hi[4];
#pragma unroll
  for (int t = 0; t < 16; ++t) acc_lo[t] = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t) acc_hi[t] = 0.f;
  auto item_geom = [&](int b, int& q0, int& k0, int& Lq, int& Lk) {
    q0 = a.q_off ? __builtin_amdgcn_readfirstlane(a.q_off[b]) : 0;
    k0 = a.k_off ? __builtin_amdgcn_readfirstlane(a.k_off[b]) : 0;
    Lq = a.q_off ? __builtin_amdgcn_readfirstlane(a.q_off[b + 1]) - q0 : a.Lq;
    Lk = a.k_off ? __builtin_amdgcn_readfirstlane(a.k_off[b + 1]) - k0 : a.Lk;
  };
  auto issue = [&](int b, char* buf) {
    int q0, k0, Lq, Lk;
    item_geom(b, q0, k0, Lq, Lk);
    const char* kbase = a.k + ((a.k_off ? (int64_t)k0 * a.kst : (int64_t)b * a.ksb) + hoff) * 2;
    const char* vbase = a.v + ((a.k_off ? (int64_t)k0 * a.vst : (int64_t)b * a.vsb) + hoff) * 2;
    const uint32_t kstb = (uint32_t)a.kst * 2u, vstb = (uint32_t)a.vst * 2u;
    const auto krs = lds_dma_rsrc(kbase, Lk > 0 ? (uint32_t)(Lk - 1) * kstb + 128u : 0u);
    const auto vrs = lds_dma_rsrc(vbase, Lk > 0 ? (uint32_t)(Lk - 1) * vstb + 128u : 0u);
    int le = lane;
    asm volatile("" : "+v"(le));
    const int pr = le >> 3, pc = le & 7;
    for (int p = wave; p < 2 * ppi; p += PF_NW) {
      const bool isv = p >= ppi;
      const int row = (isv ? p - ppi : p) * 8 + pr;
      const uint32_t voff = (uint32_t)row * (isv ? vstb : kstb) + (uint32_t)eswz(pc, row) * 16u;
      lds_dma16(buf + (isv ? R * EROW : 0) + (row - pr) * EROW, isv ? vrs : krs, row < Lk ? voff : 0x80000000u);
    }
  };
  issue(attn_seq(a, slot), smem);
  int cur = 0;
  for (int bi = slot; bi < a.Bn; bi += nslots) {
    const int b = attn_seq(a, bi);
    char* Kimg = smem + cur * 2 * R * EROW;
    char* Vimg = Kimg + R * EROW;
    int q0, k0, Lq, Lk;
    item_geom(b, q0, k0, Lq, Lk);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const bool mine = wave * 16 < Lq && Lk > 0;              // (wave-uniform)
    const int qi = wave * 16 + l15;
    u32x4 qf[2], dof[2], of[2];
    float st_m = 0.f, st_il = 0.f;
    float* stats = a.stats + ((int64_t)b * a.H + h) * a.Lq * 4;
    if (mine) {
      const char* qbase = a.q + ((a.q_off ? (int64_t)q0 * a.qst : (int64_t)b * a.qsb) + hoff) * 2;
      const char* obase = a.o + ((a.q_off ? (int64_t)q0 * a.ost : (int64_t)b * a.osb) + hoff) * 2;
      const char* dobase = a.dout + ((a.q_off ? (int64_t)q0 * a.ost : (int64_t)b * a.osb) + hoff) * 2;
      eload_frags_buf(qf, qbase, (uint32_t)a.qst * 2u, wave * 16, Lq, (uint32_t)l15 * (uint32_t)a.qst * 2u + (uint32_t)g * 16u);
      eload_frags_buf(dof, dobase, (uint32_t)a.ost * 2u, wave * 16, Lq, (uint32_t)l15 * (uint32_t)a.ost * 2u + (uint32_t)g * 16u);
      eload_frags_buf(of, obase, (uint32_t)a.ost * 2u, wave * 16, Lq, (uint32_t)l15 * (uint32_t)a.ost * 2u + (uint32_t)g * 16u);
      if (qi < Lq) {
        st_m = stats[qi * 4];
        st_il = stats[qi * 4 + 1];
      }
    }
    if (bi + nslots < a.Bn) issue(attn_seq(a, bi + nslots), smem + (cur ^ 1) * 2 * R * EROW);
    cur ^= 1;
    if (!mine) continue;
    char* dqbase = a.dq + ((a.q_off ? (int64_t)q0 * a.qst : (int64_t)b * a.qsb) + hoff) * 2;
    const int np = (Lk + 31) >> 5;
    // δ = rowsum(dO ∘ O), handed to the dK/dV pass through the statistics
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      bf16x8 dv = __builtin_bit_cast(bf16x8, dof[i]), ov = __builtin_bit_cast(bf16x8, of[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) part += (float)dv[e] * (float)ov[e];
    }
    const float delta = egroup_sum(part);
    if (qi < Lq && g == 0) stats[qi * 4 + 2] = delta;
    const float lse2 = st_il > 0.f ? st_m * LOG2E - __builtin_amdgcn_logf(st_il * dscale) : INFINITY;
    const float ndel = -delta * inv_dscale;
    const int i0 = 4 * g - qi + a.rel_off + EB_PADLO;
    int boff = (i0 & 3) * EB_ST + (i0 & ~3);
    asm volatile("" : "+v"(boff));
    const float* bptr = b4 + boff;
    const uint32_t pblk = (uint32_t)((b * a.H + h) * QB + (qi >> 2)) * (uint32_t)KB + (uint32_t)g;
    const DropRow drc = drop_row_consts(qi);
    f32x4 dq[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) dq[db] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tp = 0; tp < 8; ++tp) {
      if (tp < np) {
        f32x4 ds[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int t = 2 * tp + u, row0 = t * 16;
          f32x4 init = *reinterpret_cast<const f32x4*>(bptr + row0);
          if (tp == np - 1) init += etail(t, g, Lk);
          const f32x4 sv = escore(Kimg, row0, qf, el.roff, init);
          f32x4 dp = escore(Vimg, row0, dof, el.roff, f32x4{ndel, ndel, ndel, ndel});
          f32x4 p;
#pragma unroll
          for (int r = 0; r < 4; ++r) p[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[r], LOG2E, -lse2));
          if (DROP) {
            bool kp[4];
            drop_keep_row(drop_base(pblk + (uint32_t)(t * 4), a.drop_key), drc, t_hi, kp);
#pragma unroll
            for (int r = 0; r < 4; ++r) dp[r] = kp[r] ? dp[r] : ndel;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) ds[u][r] = p[r] * dp[r];
          acc_lo[t] += ds[u][0] + edpp<0x101>(ds[u][1]) + edpp<0x102>(ds[u][2]) + edpp<0x103>(ds[u][3]);
          const float hi = edpp<0x11F>(ds[u][1]) + edpp<0x11E>(ds[u][2]) + edpp<0x11D>(ds[u][3]);    // lanes 13..15, 0 elsewhere
          float hs = hi;                                           // row_shl:3k — lane j reads lane j + 3k (t is static after unrolling)
          switch (t & 3) {
            case 1: hs = edpp<0x103>(hi); break;
            case 2: hs = edpp<0x106>(hi); break;
            case 3: hs = edpp<0x109>(hi); break;
            default: break;
          }
          acc_hi[t >> 2] += hs;
        }
        epv(dq, ds[0], ds[1], Kimg, tp * 32, el.toff);              // dQᵀ += Kᵀ·dSᵀ
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (qi < Lq) {
      bf16_t* op = reinterpret_cast<bf16_t*>(dqbase + (int64_t)qi * a.qst * 2);
#pragma unroll
      for (int db = 0; db < 4; ++db) store4(op + db * 16 + 4 * g, dq[db]);
    }
  }
  if (want_drel) {
    // accumulator t of wave w, lane (j = lane & 15, g): lo sums key − query = 16·(t − w) + 4g − j; the wrapped elements of tile
    // t = 4u + k sit in acc_hi[u] at lane j − 3k (j = 13..15) and belong to key − query = 16·(t − w) + 4g + 16 − j
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int d0 = 16 * (t - wave) + 4 * g - l15 + a.rel_off;
      if (acc_lo[t] != 0.f && d0 >= 0 && d0 < a.R) atomicAdd(&drel_l[d0], acc_lo[t]);
    }
    if (l15 >= 4) {
      const int k = (15 - l15) / 3, j = l15 + 3 * k;               // lanes 13..15 → k 0, 10..12 → 1, 7..9 → 2, 4..6 → 3
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int d1 = 16 * (4 * u + k - wave) + 4 * g + 16 - j + a.rel_off;
        if (acc_hi[u] != 0.f && d1 >= 0 && d1 < a.R) atomicAdd(&drel_l[d1], acc_hi[u]);
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < a.R; i += PF_NW * 64) {
      const float v = drel_l[i];
      if (v != 0.f) atomicAdd(a.drel + (int64_t)h * a.R + i, v);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward, dK/dV pass: Q / dO in LDS, a wave owns 16 keys; S[query][key] and dP tiles → P̃, dS → dKᵀ += Qᵀ·dS, dVᵀ += dOᵀ·P̃
// ---------------------------------------------------------------------------------------------------------------------
constexpr int DKV_NW = 4;     // (8 waves on <= 128 registers measured 10 % slower: no room to request the next key block ahead)
// LDS: Q | dO | 4 reversed bias copies | lse2[R] | −δ/scale[R]
__host__ __device__ constexpr int ebwd1_lds(int rows) { return 2 * rows * EROW + 4 * EB_ST * 4 + 2 * rows * 4; }

template <bool DROP>
__global__ __launch_bounds__(DKV_NW * 64, 2) void enc_bwd_dkv_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int R = a.chunk_rows;
  char* Qimg = smem;
  char* dOimg = smem + R * EROW;
  float* b4 = reinterpret_cast<float*>(smem + 2 * R * EROW);
  float* lse_l = b4 + 4 * EB_ST;
  float* ndel_l = lse_l + R;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  const int b = attn_seq(a, blockIdx.z), h = blockIdx.y;
  const int q0 = a.q_off ? __builtin_amdgcn_readfirstlane(a.q_off[b]) : 0;
  const int k0 = a.k_off ? __builtin_amdgcn_readfirstlane(a.k_off[b]) : 0;
  const int Lq = a.q_off ? __builtin_amdgcn_readfirstlane(a.q_off[b + 1]) - q0 : a.Lq;
  const int Lk = a.k_off ? __builtin_amdgcn_readfirstlane(a.k_off[b + 1]) - k0 : a.Lk;
  const int kb_begin = blockIdx.x * a.blocks_per_wg;
  const int kb_end = min((Lk + 15) >> 4, kb_begin + a.blocks_per_wg);
  if (kb_begin >= kb_end) return;
  const int64_t hoff = (int64_t)h * 64;
  const char* qbase = a.q + ((a.q_off ? (int64_t)q0 * a.qst : (int64_t)b * a.qsb) + hoff) * 2;
  const char* dobase = a.dout + ((a.q_off ? (int64_t)q0 * a.ost : (int64_t)b * a.osb) + hoff) * 2;
  const char* kbase = a.k + ((a.k_off ? (int64_t)k0 * a.kst : (int64_t)b * a.ksb) + hoff) * 2;
  const char* vbase = a.v + ((a.k_off ? (int64_t)k0 * a.vst : (int64_t)b * a.vsb) + hoff) * 2;
  char* dkbase = a.dk + ((a.k_off ? (int64_t)k0 * a.kst : (int64_t)b * a.ksb) + hoff) * 2;
  char* dvbase = a.dv + ((a.k_off ? (int64_t)k0 * a.vst : (int64_t)b * a.vsb) + hoff) * 2;
  const float* stats = a.stats + ((int64_t)b * a.H + h) * a.Lq * 4;
  const float dscale = DROP ? a.drop_scale : 1.0f, inv_dscale = 1.0f / dscale;
  const int npq = (max(Lq, 1) + 31) >> 5, nq = npq << 5;    // query tile pairs, staged query rows
  // (round 4) Q / dO by LDS-DMA, issued first: the images, the bias table and the statistics are then ONE memory round trip instead of four
  if (a.dma_stage) {
    estage_dma<DKV_NW>(Qimg, qbase, (uint32_t)a.qst * 2u, nq, Lq, wave, lane);
    estage_dma<DKV_NW>(dOimg, dobase, (uint32_t)a.ost * 2u, nq, Lq, wave, lane);
  } else {
    estage<DKV_NW * 64>(Qimg, qbase, a.qst * 2, nq, Lq);
    estage<DKV_NW * 64>(dOimg, dobase, a.ost * 2, nq, Lq);
  }
  estage_bias<DKV_NW * 64>(b4, a.rel_bias, h, a.R, true);
  for (int i = threadIdx.x; i < nq; i += DKV_NW * 64) {
    float l2 = INFINITY, nd = 0.f;                          // rows past Lq: p = exp2(−inf) = 0
    if (i < Lq) {
      const f32x4 st4 = *reinterpret_cast<const f32x4*>(stats + i * 4);
      l2 = st4[0] * LOG2E - __builtin_amdgcn_logf(st4[1] * dscale);
      nd = -st4[2] * inv_dscale;
    }
    lse_l[i] = l2;
    ndel_l[i] = nd;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the wave's DMA pieces (the compiler does not count them)
  __syncthreads();
  const ELane el = elane(lane);
  const uint32_t t_hi = a.drop_t16 << 16;
  const int KB = (a.Lk + 3) >> 2, QB = (a.Lq + 3) >> 2;
  const int last = a.R - 1 + EB_PADLO + 36;

  u32x4 kf_next[2], vf_next[2];
  auto request = [&](int kbn) {
    const int kn = kbn < kb_end ? kbn * 16 + l15 : Lk;
    eload_frags(kf_next, kbase, a.kst * 2, kn, Lk, g);
    eload_frags(vf_next, vbase, a.vst * 2, kn, Lk, g);
  };
  request(kb_begin + wave);
  for (int kb = kb_begin + wave; kb < kb_end; kb += DKV_NW) {
    const int ki = kb * 16 + l15;                           // this lane's key
    u32x4 kf[2] = {kf_next[0], kf_next[1]}, vf[2] = {vf_next[0], vf_next[1]};
    request(kb + DKV_NW);
    const float kadd_lane = ki < Lk ? 0.f : -INFINITY;
    f32x4 dkacc[4], dvacc[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      dkacc[db] = f32x4{0.f, 0.f, 0.f, 0.f};
      dvacc[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // bias: tile rows are queries row0 + 4g + r, the lane's key is ki: T[ki − (row0 + 4g + r) + rel_off] = Trev[j0 + row0 + r],
    // j0 = last − (ki − 4g + rel_off + PADLO)
    const int j0 = last - (ki - 4 * g + a.rel_off + EB_PADLO);
    const float* bptr = b4 + (j0 & 3) * EB_ST + (j0 & ~3);
    const DropCol dcc = drop_col_consts(ki);
    const uint32_t cblk = (uint32_t)((b * a.H + h) * QB + g) * (uint32_t)KB + (uint32_t)(ki >> 2);
    const f32x4 kaddv = {kadd_lane, kadd_lane, kadd_lane, kadd_lane};
    uint32_t cblk_tp = cblk;                                // + (row0 >> 2)·KB, advanced by additions (a 32-bit multiply is quarter rate)
    const uint32_t kb4 = 4u * (uint32_t)KB;
    for (int tp = 0; tp < npq; ++tp, cblk_tp += 2 * kb4) {
      f32x4 pt[2], ds[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int row0 = tp * 32 + u * 16;
        const f32x4 init = *reinterpret_cast<const f32x4*>(bptr + row0) + kaddv;
        const f32x4 nd = *reinterpret_cast<const f32x4*>(ndel_l + row0 + 4 * g);
        const f32x4 l2 = *reinterpret_cast<const f32x4*>(lse_l + row0 + 4 * g);
        const f32x4 sv = escore(Qimg, row0, kf, el.roff, init);
        f32x4 dp = escore(dOimg, row0, vf, el.roff, nd);
        f32x4 p;
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[r], LOG2E, -l2[r]));
        pt[u] = p;
        if (DROP) {
          bool kp[4];
          drop_keep_col(drop_base(cblk_tp + (uint32_t)u * kb4, a.drop_key), dcc, t_hi, kp);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            pt[u][r] = kp[r] ? p[r] : 0.f;
            dp[r] = kp[r] ? dp[r] : nd[r];
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) ds[u][r] = p[r] * dp[r];
      }
      epv(dkacc, ds[0], ds[1], Qimg, tp * 32, el.toff);       // dKᵀ += Qᵀ·dS
      epv(dvacc, pt[0], pt[1], dOimg, tp * 32, el.toff);      // dVᵀ += dOᵀ·P̃
    }
    if (ki < Lk) {
      bf16_t* kp_ = reinterpret_cast<bf16_t*>(dkbase + (int64_t)ki * a.kst * 2);
      bf16_t* vp_ = reinterpret_cast<bf16_t*>(dvbase + (int64_t)ki * a.vst * 2);
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        store4(kp_ + db * 16 + 4 * g, dkacc[db]);
        store4(vp_ + db * 16 + 4 * g, dvacc[db]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward, dK/dV pass, PERSISTENT form (round 3) — the structure of enc_fwd_p_kernel: one 16-wave workgroup per CU walks the
// sequences of ONE head, wave w owns key block w of every item (≤ 256 keys: one pass), the Q / dO images and the raw per-query
// statistics of the next item arrive by LDS-DMA into the second image pair under the current item's arithmetic.  At the top of an
// item: wait + barrier (images landed, everybody done with the other pair) → request this wave's K / V fragments → issue the next
// item's DMA → convert the raw statistics to (lse2, −δ/scale) → barrier → arithmetic.  (K / V fragments requested an item ahead
// cost 16 registers across the key loop, and with ≤ 128 registers the spill reloads — vector-memory operations in the same in-order
// counter — made every item wait for the DMA it had just issued.)  The reversed bias copies are staged once per workgroup.
// Same arithmetic per key block as enc_bwd_dkv_kernel: bit-identical results.
// ---------------------------------------------------------------------------------------------------------------------
// LDS: 2 × (Q image | dO image) | 4 reversed bias copies | 2 × raw statistics [R][4] | lse2[R] | −δ/scale[R]
// (the raw statistics arrive in 64-row DMA pieces: each pair holds the row count rounded up to 64, or a piece would run over into the
//  other pair / the converted vectors — found by test_attention_persistent_kernels_equal_per_item_kernels at 224 image rows)
__host__ __device__ constexpr int ebwd1p_lds(int rows) { return 4 * rows * EROW + 4 * EB_ST * 4 + 2 * ((rows + 63) / 64 * 64) * 16 + 2 * rows * 4; }

template <bool DROP>
__global__ __launch_bounds__(PF_NW * 64) void enc_bwd_dkv_p_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int R = a.chunk_rows;                       // image rows: every sequence's query count rounded up to 32
  float* b4 = reinterpret_cast<float*>(smem + 4 * R * EROW);
  const int RS = (R + 63) / 64 * 64;                // statistics rows per pair (whole 64-row DMA pieces)
  float* raw_l = b4 + 4 * EB_ST;                    // [2][RS][4]
  float* lse_l = raw_l + 2 * RS * 4;
  float* ndel_l = lse_l + R;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  const int h = blockIdx.x % a.H, slot = blockIdx.x / a.H, nslots = gridDim.x / a.H;
  if (slot >= a.Bn) return;
  estage_bias<PF_NW * 64>(b4, a.rel_bias, h, a.R, true);
  const ELane el = elane(lane);
  const uint32_t t_hi = a.drop_t16 << 16;
  const int KB = (a.Lk + 3) >> 2, QB = (a.Lq + 3) >> 2;
  const float dscale = DROP ? a.drop_scale : 1.0f, inv_dscale = 1.0f / dscale;
  const int64_t hoff = (int64_t)h * 64;
  const int last = a.R - 1 + EB_PADLO + 36;
  const int ppi = R >> 3;
  auto item_geom = [&](int b, int& q0, int& k0, int& Lq, int& Lk) {
    q0 = a.q_off ? __builtin_amdgcn_readfirstlane(a.q_off[b]) : 0;
    k0 = a.k_off ? __builtin_amdgcn_readfirstlane(a.k_off[b]) : 0;
    Lq = a.q_off ? __builtin_amdgcn_readfirstlane(a.q_off[b + 1]) - q0 : a.Lq;
    Lk = a.k_off ? __builtin_amdgcn_readfirstlane(a.k_off[b + 1]) - k0 : a.Lk;
  };
  auto issue = [&](int b, int pair) {               // Q rows | dO rows | raw statistics of item b → image / statistics pair `pair`
    int q0, k0, Lq, Lk;
    item_geom(b, q0, k0, Lq, Lk);
    char* buf = smem + pair * 2 * R * EROW;
    const char* qbase = a.q + ((a.q_off ? (int64_t)q0 * a.qst : (int64_t)b * a.qsb) + hoff) * 2;
    const char* dobase = a.dout + ((a.q_off ? (int64_t)q0 * a.ost : (int64_t)b * a.osb) + hoff) * 2;
    const uint32_t qstb = (uint32_t)a.qst * 2u, ostb = (uint32_t)a.ost * 2u;
    const auto qrs = lds_dma_rsrc(qbase, Lq > 0 ? (uint32_t)(Lq - 1) * qstb + 128u : 0u);
    const auto drs = lds_dma_rsrc(dobase, Lq > 0 ? (uint32_t)(Lq - 1) * ostb + 128u : 0u);
    int le = lane;
    asm volatile("" : "+v"(le));
    const int pr = le >> 3, pc = le & 7;
    for (int p = wave; p < 2 * ppi; p += PF_NW) {
      const bool isd = p >= ppi;
      const int row = (isd ? p - ppi : p) * 8 + pr;
      const uint32_t voff = (uint32_t)row * (isd ? ostb : qstb) + (uint32_t)eswz(pc, row) * 16u;
      lds_dma16(buf + (isd ? R * EROW : 0) + (row - pr) * EROW, isd ? drs : qrs, row < Lq ? voff : 0x80000000u);
    }
    if (wave * 64 < R) {                            // statistics rows [64·wave, 64·wave + 64): 16 bytes per query, contiguous
      const auto srs = lds_dma_rsrc(a.stats + ((int64_t)b * a.H + h) * a.Lq * 4, (uint32_t)max(Lq, 0) * 16u);
      lds_dma16(raw_l + (pair * RS + wave * 64) * 4, srs, (uint32_t)(wave * 64 + le) * 16u);
    }
  };
  issue(attn_seq(a, slot), 0);
  int cur = 0;
  for (int bi = slot; bi < a.Bn; bi += nslots) {
    const int b = attn_seq(a, bi);
    char* Qimg = smem + cur * 2 * R * EROW;
    char* dOimg = Qimg + R * EROW;
    int q0, k0, Lq, Lk;
    item_geom(b, q0, k0, Lq, Lk);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's pieces of item b have landed (its stores of the last item are out)
    __syncthreads();                                         // … everybody's; and every wave is done with the other pair and with lse / −δ
    const bool mine = wave * 16 < Lk;                        // (wave-uniform) this wave's key block exists
    u32x4 kf[2], vf[2];
    if (mine) {
      const char* kbase = a.k + ((a.k_off ? (int64_t)k0 * a.kst : (int64_t)b * a.ksb) + hoff) * 2;
      const char* vbase = a.v + ((a.k_off ? (int64_t)k0 * a.vst : (int64_t)b * a.vsb) + hoff) * 2;
      eload_frags_buf(kf, kbase, (uint32_t)a.kst * 2u, wave * 16, Lk, (uint32_t)l15 * (uint32_t)a.kst * 2u + (uint32_t)g * 16u);
      eload_frags_buf(vf, vbase, (uint32_t)a.vst * 2u, wave * 16, Lk, (uint32_t)l15 * (uint32_t)a.vst * 2u + (uint32_t)g * 16u);
    }
    if (bi + nslots < a.Bn) issue(attn_seq(a, bi + nslots), cur ^ 1);
    if ((int)threadIdx.x < R) {                              // rows past Lq arrive as zeros: p = exp2(−inf) = 0 there
      const f32x4 st4 = *reinterpret_cast<const f32x4*>(raw_l + (cur * RS + threadIdx.x) * 4);
      lse_l[threadIdx.x] = st4[1] > 0.f ? st4[0] * LOG2E - __builtin_amdgcn_logf(st4[1] * dscale) : INFINITY;
      ndel_l[threadIdx.x] = -st4[2] * inv_dscale;
    }
    __syncthreads();
    cur ^= 1;
    if (!mine) continue;
    char* dkbase = a.dk + ((a.k_off ? (int64_t)k0 * a.kst : (int64_t)b * a.ksb) + hoff) * 2;
    char* dvbase = a.dv + ((a.k_off ? (int64_t)k0 * a.vst : (int64_t)b * a.vsb) + hoff) * 2;
    const int npq = (max(Lq, 1) + 31) >> 5;
    const int ki = wave * 16 + l15;
    const float kadd_lane = ki < Lk ? 0.f : -INFINITY;
    f32x4 dkacc[4], dvacc[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      dkacc[db] = f32x4{0.f, 0.f, 0.f, 0.f};
      dvacc[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int j0 = last - (ki - 4 * g + a.rel_off + EB_PADLO);
    const float* bptr = b4 + (j0 & 3) * EB_ST + (j0 & ~3);
    const DropCol dcc = drop_col_consts(ki);
    const uint32_t cblk = (uint32_t)((b * a.H + h) * QB + g) * (uint32_t)KB + (uint32_t)(ki >> 2);
    const f32x4 kaddv = {kadd_lane, kadd_lane, kadd_lane, kadd_lane};
    uint32_t cblk_tp = cblk;
    const uint32_t kb4 = 4u * (uint32_t)KB;
    for (int tp = 0; tp < npq; ++tp, cblk_tp += 2 * kb4) {
      f32x4 pt[2], ds[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int row0 = tp * 32 + u * 16;
        const f32x4 init = *reinterpret_cast<const f32x4*>(bptr + row0) + kaddv;
        const f32x4 nd = *reinterpret_cast<const f32x4*>(ndel_l + row0 + 4 * g);
        const f32x4 l2 = *reinterpret_cast<const f32x4*>(lse_l + row0 + 4 * g);
        const f32x4 sv = escore(Qimg, row0, kf, el.roff, init);
        f32x4 dp = escore(dOimg, row0, vf, el.roff, nd);
        f32x4 p;
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[r], LOG2E, -l2[r]));
        pt[u] = p;
        if (DROP) {
          bool kp[4];
          drop_keep_col(drop_base(cblk_tp + (uint32_t)u * kb4, a.drop_key), dcc, t_hi, kp);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            pt[u][r] = kp[r] ? p[r] : 0.f;
            dp[r] = kp[r] ? dp[r] : nd[r];
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) ds[u][r] = p[r] * dp[r];
      }
      epv(dkacc, ds[0], ds[1], Qimg, tp * 32, el.toff);       // dKᵀ += Qᵀ·dS
      epv(dvacc, pt[0], pt[1], dOimg, tp * 32, el.toff);      // dVᵀ += dOᵀ·P̃
    }
    if (ki < Lk) {
      bf16_t* kp_ = reinterpret_cast<bf16_t*>(dkbase + (int64_t)ki * a.kst * 2);
      bf16_t* vp_ = reinterpret_cast<bf16_t*>(dvbase + (int64_t)ki * a.vst * 2);
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        store4(kp_ + db * 16 + 4 * g, dkacc[db]);
        store4(vp_ + db * 16 + 4 * g, dvacc[db]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward, ONE pass (round 5): dQ, dK, dV and the bias gradient of a (sequence, head) item from one workgroup that reads
// q, k, v, dO and O ONCE.  The two-pass form above recomputes the score tiles (and their exp / dropout hashes — the bulk
// of the vector work) in both passes and reads q, k, v, dO from HBM twice; round 2's one-pass attempt summed dQ through
// LDS float atomics (≈ 0.3 lane-operations per clock and CU) and was 7× slower.  Here nothing is summed across waves:
//   * one 16-wave workgroup per CU walks the items of ONE head; wave w owns key tile w (16 keys: its V rows stay in
//     registers for the whole item, dKᵀ and dVᵀ of those keys accumulate in registers);
//   * the queries stream through in slabs of 32 rows — the Q, dO and O rows plus the row statistics of a slab are one
//     ring stage filled by LDS-DMA NST − 1 slabs ahead, ACROSS item boundaries: no staging phase per item;
//   * P1 (per slab, every wave with a key tile): S = Q·Kᵀ + bias and dP = dO·Vᵀ − δ for (32 queries × its 16 keys),
//     P̃, dS, dKᵀ += Qᵀ·dS, dVᵀ += dOᵀ·P̃ as in the dK/dV pass above, the bias gradient in diagonal-indexed registers
//     (DPP row shifts; the wave's tile diagonal depends on the slab only), and dS — rounded to bf16 exactly as the MFMA
//     operand is — goes into a [key][32 queries] LDS slab;
//   * P2 (one slab later, waves 14 and 15 — beyond the key tiles of every sequence of ≤ 224 keys, so they are dedicated):
//     dQᵀ[64 d × 16 queries] = Kᵀ·dSᵀ contracted over ALL keys of the item — both operands transposed reads (the K image
//     of the item, staged once by LDS-DMA during the previous item, and the dS slab): every output element has ONE
//     owning wave and a complete sum — no atomics, no cross-wave reduction;
//   * wave 14 also issues the ring's DMA; wave 15 forms δ = rowsum(dO ∘ O) and the (lse₂, −δ/scale) vectors of the slab
//     that has just landed, one slab ahead of its use, and issues the next item's K image.
// One barrier per slab; every role walks the same stage sequence with its own cursor (P1 waves carry no state of the
// other roles).  The dS slabs are double-buffered (P2 of slab n − 1 runs beside P1 of slab n), the K image too (item
// j + 1's arrives while item j computes; every item takes at least two slab iterations so that the image being
// overwritten is dead).  HBM traffic per item: q, k, v, dO, O read once, dq, dk, dv written once.
// Results: dk, dv as the dK/dV pass (same arithmetic per tile); dq sums the same bf16 dS values in fp32 over all keys
// in another order than the dQ pass.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int FB_NW = 16;
constexpr int FB_SLAB = 32 * EROW;                      // one 32-row slab image
constexpr int FB_STAGE = 3 * FB_SLAB + 1024;            // Q | dO | O | raw statistics (32 × 16 B, then 512 B of DMA zero fill)
constexpr int FB_TBL = 64;                              // items per table pass
// LDS: 2 K images | 2 dS slabs [rows][32 q] | ring | 4 reversed bias copies | lse2 / −δ/scale per stage | item table + header | lane constants [64][12] |
// stage table [FB_TBL·8][4]
__host__ __device__ constexpr int efb_lds(int rows, int nst) {
  return 2 * rows * EROW + 2 * rows * 64 + nst * FB_STAGE + 4 * EB_ST * 4 + nst * 2 * 32 * 4 + FB_TBL * 16 + 16 + 64 * 48 + FB_TBL * 8 * 16;
}

// (experiments build) cycle stamps of workgroup 0: [wave][iteration < 256][point < 8] s_memtime values in a.scores_out
#define FB_STAMP(P)                                                                                                          \
  do {                                                                                                                       \
    if (ATTN_DBG(a, 131072) && blockIdx.x == 0 && (threadIdx.x & 63) == 0 && itn < 256)                                     \
      reinterpret_cast<unsigned long long*>(a.scores_out)[(wave * 256 + itn) * 8 + (P)] = __builtin_amdgcn_s_memtime();       \
  } while (0)

#define FB_STAMP_AT(IT, P)                                                                                                   \
  do {                                                                                                                       \
    if (ATTN_DBG(a, 131072) && blockIdx.x == 0 && (threadIdx.x & 63) == 0)                                                  \
      reinterpret_cast<unsigned long long*>(a.scores_out)[(wave * 256 + (IT)) * 8 + (P)] = __builtin_amdgcn_s_memtime();      \
  } while (0)

struct FbItem { int b, q0, k0, Lq, Lk; };
// one stage of the stream = (item j, query slab s), decoded from its 16-byte record in the stage table
struct FbStage { int b, s, kpar, last, qrow, left, Lk, j; };
constexpr int FB_MAXST = FB_TBL * 8;                    // stages per table pass (≤ 8 slabs per item)

// byte offset of the 8-byte slot (q-tile u, lane group g) of key row `row` in a dS slab: 64-byte rows, slot bits swizzled by the
// row so that the transposed reads of P2 (8 rows × 4 slots per half wave) and the writes of P1 (16 rows × 1 slot) spread over the banks
__device__ __forceinline__ uint32_t fb_ds_off(int row, int u, int g) {
  return (uint32_t)(row * 64 + ((((u ^ ((row >> 2) & 1)) << 2) | (g ^ ((row >> 1) & 3))) << 3));
}

// Rows are addressed in ONE form: row r of sequence b of a tensor with token stride st (elements) sits at (off + r)·st + h·64, where
// off = q_off[b] for ragged launches and b·(batch stride ÷ token stride) for padded ones (the launcher checks the divisibility and puts the
// quotients into rq / rk).
template <bool DROP, int NST>
__global__ __launch_bounds__(FB_NW * 64) void enc_bwd_fused_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int R = a.chunk_rows;                       // key image rows: a multiple of 32 ≥ every sequence's key count
  char* Kimg0 = smem;
  char* dST0 = smem + 2 * R * EROW;
  char* ring = dST0 + 2 * R * 64;
  float* b4 = reinterpret_cast<float*>(ring + NST * FB_STAGE);
  float* lse_l = b4 + 4 * EB_ST;
  float* ndel_l = lse_l + NST * 32;
  int* tbl = reinterpret_cast<int*>(ndel_l + NST * 32);                       // items [FB_TBL][4] + header {items, stages, -, -}
  uint32_t* ltab = reinterpret_cast<uint32_t*>(tbl + FB_TBL * 4 + 4);         // P1's lane constants: 3 × 16 bytes per lane (below)
  uint32_t* stab = ltab + 64 * 12;                                            // stages [FB_MAXST][4]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = blockIdx.x % a.H, slot0 = blockIdx.x / a.H, nslots = gridDim.x / a.H;
  if (slot0 >= a.Bn) return;
  const bool want_drel = a.drel != nullptr;
  int itn = 0;                                      // (stamps only) iterations so far
  FB_STAMP_AT(255, 0);
  estage_bias<FB_NW * 64>(b4, a.rel_bias, h, a.R, true);
  for (int i = threadIdx.x * 16; i < 2 * R * 64; i += FB_NW * 64 * 16) *reinterpret_cast<u32x4*>(dST0 + i) = u32x4{0u, 0u, 0u, 0u};     // (P2 reads every row of a slab)
  const int KB = (a.Lk + 3) >> 2, QB = (a.Lq + 3) >> 2;
  const float dscale = DROP ? a.drop_scale : 1.0f, inv_dscale = 1.0f / dscale;
  const int nmine = (a.Bn - slot0 + nslots - 1) / nslots;         // items of this workgroup (empty ones included)
  auto item = [&](int j) {
    const u32x4 e = *reinterpret_cast<const u32x4*>(tbl + j * 4);
    FbItem it;
    it.b = __builtin_amdgcn_readfirstlane((int)e[0]);
    it.q0 = __builtin_amdgcn_readfirstlane((int)e[1]);
    it.k0 = __builtin_amdgcn_readfirstlane((int)e[2]);
    const int w3 = __builtin_amdgcn_readfirstlane((int)e[3]);
    it.Lq = w3 & 0xFFFF;
    it.Lk = w3 >> 16;
    return it;
  };
  auto stage_raw = [&](int n) { return *reinterpret_cast<const u32x4*>(stab + n * 4); };
  auto stage_dec = [&](u32x4 e) {
    FbStage st;
    const int w0 = __builtin_amdgcn_readfirstlane((int)e[0]), w2 = __builtin_amdgcn_readfirstlane((int)e[2]);
    st.b = w0 & 0xFFFF;
    st.s = (w0 >> 16) & 0xFF;
    st.kpar = (w0 >> 24) & 1;
    st.last = (w0 >> 25) & 1;
    st.qrow = __builtin_amdgcn_readfirstlane((int)e[1]);
    st.left = w2 & 0xFFFF;
    st.Lk = w2 >> 16;
    st.j = __builtin_amdgcn_readfirstlane((int)e[3]);
    return st;
  };
  auto bar = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  if (wave >= 14) {
    // ================= waves 14 / 15: the ring's DMA, δ, the K images, P2 — no key tile (the launcher takes sequences of ≤ 224 keys) =================
    // They are the youngest waves of their SIMDs: at equal priority the arbiter serves them last, and every other wave then waits for them at
    // the barrier (stamps: P2 took 4 500 cycles beside three busy P1 waves) — raise their priority once.
    __builtin_amdgcn_s_setprio(3);
    const int part = wave - 14;
    // ---- the ring's DMA: Q and dO rows of a slab (part 0: 8 pieces of 8 rows, wave 14), O rows and the 32 raw statistics records (part 1: 5 pieces)
    auto issue_stage = [&](const FbStage& st, int slot) {
      int le = lane;
      asm volatile("" : "+v"(le));
      char* sp = ring + slot * FB_STAGE;
      const int nrow = min(st.left, 32);                              // rows of this slab that exist
      const uint32_t qstb = (uint32_t)a.qst * 2u, ostb = (uint32_t)a.ost * 2u;
      const int pr = le >> 3, pc = le & 7;
      const uint32_t sw = (uint32_t)eswz(pc, pr) * 16u;               // (the swizzle key repeats every 8 rows)
      if (part == 0) {
        const auto qrs = lds_dma_rsrc(a.q + h * 128 + (int64_t)st.qrow * qstb, nrow > 0 ? (uint32_t)(nrow - 1) * qstb + 128u : 0u);
        const auto drs = lds_dma_rsrc(a.dout + h * 128 + (int64_t)st.qrow * ostb, nrow > 0 ? (uint32_t)(nrow - 1) * ostb + 128u : 0u);
#pragma unroll
        for (int pz = 0; pz < 4; ++pz) {
          const int row = pz * 8 + pr;
          const bool in = row < nrow;
          lds_dma16(sp + pz * 1024, qrs, in ? (uint32_t)row * qstb + sw : 0x80000000u);
          lds_dma16(sp + FB_SLAB + pz * 1024, drs, in ? (uint32_t)row * ostb + sw : 0x80000000u);
        }
      } else {
        const auto ors = lds_dma_rsrc(a.o + h * 128 + (int64_t)st.qrow * ostb, nrow > 0 ? (uint32_t)(nrow - 1) * ostb + 128u : 0u);
        const auto srs = lds_dma_rsrc(a.stats + (((int64_t)st.b * a.H + h) * a.Lq + 32 * st.s) * 4, (uint32_t)nrow * 16u);
#pragma unroll
        for (int pz = 0; pz < 4; ++pz) {
          const int row = pz * 8 + pr;
          lds_dma16(sp + 2 * FB_SLAB + pz * 1024, ors, row < nrow ? (uint32_t)row * ostb + sw : 0x80000000u);
        }
        lds_dma16(sp + 3 * FB_SLAB, srs, le < nrow ? (uint32_t)le * 16u : 0x80000000u);
      }
    };
    constexpr int PIECES0 = 8, PIECES1 = 5;
    auto issue_kimg = [&](const FbItem& it, int j) {                  // K image of item j, all pieces by the calling wave
      const uint32_t kstb = (uint32_t)a.kst * 2u;
      estage_dma<1>(Kimg0 + (j & 1) * R * EROW, a.k + h * 128 + (int64_t)it.k0 * kstb, kstb, R, it.Lk, 0, lane);     // all R rows: zeros past the sequence's end (P2)
    };
    // ---- δ and the row constants of a stage that has landed: 32 queries, 2 lanes per query
    auto cstage = [&](const FbStage& st, int slot) {
      const char* sp = ring + slot * FB_STAGE;
      const int ql = lane >> 1, qh2 = lane & 1;
      float part_ = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int off = ql * EROW + eswz(4 * qh2 + c, ql) * 16;
        const u32x4 dv = *reinterpret_cast<const u32x4*>(sp + FB_SLAB + off), ov = *reinterpret_cast<const u32x4*>(sp + 2 * FB_SLAB + off);
        // v_dot2c_f32_bf16: two exact products and the running fp32 sum per instruction.  (The pairs are taken by shufflevector: hipcc 7.2
        // turns `bit_cast<bf16x2>(u32x4[e])` of an unrolled e into FOUR uses of element 0 — found by the parity tests.)
        const bf16x8 dvb = __builtin_bit_cast(bf16x8, dv), ovb = __builtin_bit_cast(bf16x8, ov);
        part_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(dvb, dvb, 0, 1), __builtin_shufflevector(ovb, ovb, 0, 1), part_, false);
        part_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(dvb, dvb, 2, 3), __builtin_shufflevector(ovb, ovb, 2, 3), part_, false);
        part_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(dvb, dvb, 4, 5), __builtin_shufflevector(ovb, ovb, 4, 5), part_, false);
        part_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(dvb, dvb, 6, 7), __builtin_shufflevector(ovb, ovb, 6, 7), part_, false);
      }
      part_ += __shfl_xor(part_, 1, 64);
      const f32x4 raw = *reinterpret_cast<const f32x4*>(sp + 3 * FB_SLAB + ql * 16);
      if (qh2 == 0) {
        lse_l[slot * 32 + ql] = raw[1] > 0.f ? raw[0] * LOG2E - __builtin_amdgcn_logf(raw[1] * dscale) : INFINITY;
        ndel_l[slot * 32 + ql] = -part_ * inv_dscale;
        if (ql < st.left) a.stats[(((int64_t)st.b * a.H + h) * a.Lq + 32 * st.s + ql) * 4 + 2] = part_;   // (the contract of lako_attn_bwd_t.lse)
      }
    };
    // ---- P2: dQᵀ[d-blocks 2·part, 2·part + 1 × both q-tiles] of a stage whose dS slab is dS[par]: the two waves share the key range and split
    // the d-blocks, so each K fragment (A) and each dS fragment (B) is read once per wave and used twice: 8 transposed reads per 4 MFMAs.
    // The key steps are fully unrolled over ALL R / 32 steps of the image (the K image is staged whole — rows past the sequence's end are
    // zeros — and the dS slabs start as zeros, so the steps past an item's keys add exact zeros): in straight-line code hipcc's wait counts
    // are exact and the fragments of step k + 1 really are in flight under the MFMAs of step k (in a loop it waits lgkmcnt(0) before every
    // step's MFMAs — the stamps showed 2 700 cycles per call for 28 MFMAs).  Addresses are 32-bit LDS offsets: xor on a generic pointer makes
    // hipcc re-derive the LDS address with a null check per read.
    auto p2 = [&](const FbStage& st, int par) {
      int le = lane;
      asm volatile("" : "+v"(le));
      const int trr = 4 * (le >> 4) + ((le & 15) >> 2), trp = le & 3;
      const uint32_t Ka = (uint32_t)(uintptr_t)LDS_PTR(Kimg0) + (uint32_t)(st.kpar * R * EROW + trr * EROW + ((((trr >> 1) & 3) ^ (2 * part)) << 5) + trp * 8);
      const uint32_t Da = (uint32_t)(uintptr_t)LDS_PTR(dST0) + (uint32_t)(par * R * 64) + fb_ds_off(trr, 0, trp);
      const uint32_t Kx = Ka ^ 32u, Dx = Da ^ 32u;                    // the next d-block / q-tile 1
      f32x4 acc[2][2];                                                 // [d-block][q-tile]
#pragma unroll
      for (int x = 0; x < 4; ++x) acc[x >> 1][x & 1] = f32x4{0.f, 0.f, 0.f, 0.f};
      auto trd = [](uint32_t off) {
        return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)off));
      };
      struct Frag { u32x2 a0[2], a1[2], b0[2], b1[2]; };
      auto fetch = [&](Frag& f, int kk) {
        f.a0[0] = trd(Ka + kk * 32 * EROW); f.a1[0] = trd(Ka + kk * 32 * EROW + 16 * EROW);
        f.a0[1] = trd(Kx + kk * 32 * EROW); f.a1[1] = trd(Kx + kk * 32 * EROW + 16 * EROW);
        f.b0[0] = trd(Da + kk * 32 * 64);   f.b1[0] = trd(Da + kk * 32 * 64 + 16 * 64);
        f.b0[1] = trd(Dx + kk * 32 * 64);   f.b1[1] = trd(Dx + kk * 32 * 64 + 16 * 64);
      };
      auto mma = [&](const Frag& f) {
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int q = 0; q < 2; ++q)
            acc[d][q] = emma(u32x4{f.a0[d][0], f.a0[d][1], f.a1[d][0], f.a1[d][1]}, u32x4{f.b0[q][0], f.b0[q][1], f.b1[q][0], f.b1[q][1]}, acc[d][q]);
      };
      auto run = [&](auto NKc) {
        constexpr int NK = decltype(NKc)::value;
        Frag f[2];
        fetch(f[0], 0);
        static_for<NK>([&](auto Kc) {
          constexpr int k = decltype(Kc)::value;
          if constexpr (k + 1 < NK) fetch(f[(k + 1) & 1], k + 1);
          mma(f[k & 1]);
        });
      };
      switch (R >> 5) {
        case 1: run(std::integral_constant<int, 1>{}); break;
        case 2: run(std::integral_constant<int, 2>{}); break;
        case 3: run(std::integral_constant<int, 3>{}); break;
        case 4: run(std::integral_constant<int, 4>{}); break;
        case 5: run(std::integral_constant<int, 5>{}); break;
        case 6: run(std::integral_constant<int, 6>{}); break;
        default: run(std::integral_constant<int, 7>{}); break;
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int ql = 16 * q + (le & 15);
        if (ql < st.left) {
          bf16_t* op = reinterpret_cast<bf16_t*>(a.dq + h * 128 + (int64_t)(st.qrow + ql) * ((int64_t)a.qst * 2)) + 32 * part + 4 * (le >> 4);
          store4(op, acc[0][q]);
          store4(op + 16, acc[1][q]);
        }
      }
    };
    for (int t0 = 0; t0 < nmine; t0 += FB_TBL) {
      __syncthreads();
      __syncthreads();                                              // (wave 0 builds the item and stage tables between these two)
      const int nit = __builtin_amdgcn_readfirstlane(tbl[FB_TBL * 4]);
      const int nstage = __builtin_amdgcn_readfirstlane(tbl[FB_TBL * 4 + 1]);
      if (nit == 0) continue;
      // prologue: item 0's K image, the first NST − 1 stages; δ of stage 0
      if (wave == 15) issue_kimg(item(0), 0);
#pragma unroll
      for (int k = 0; k < NST - 1; ++k)
        if (k < nstage) issue_stage(stage_dec(stage_raw(k)), k);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (wave == 15) cstage(stage_dec(stage_raw(0)), 0);
      // the stage stream: iteration n = P1 of stage n (the other waves), P2 of stage n − 1, δ of stage n + 1, DMA of stage n + NST − 1
      bool issued = false;
      for (int n = 0; n <= nstage; ++n) {
        // the pieces issued in the previous iteration (the wave's newest operations) may stay in flight with a ring of 4.  (The next item's K
        // image needs no wait of its own: wave 15 issues it BEFORE the ring pieces of that iteration, so the counted wait of the following
        // iteration already covers it — a vmcnt(0) at every new item made wave 15 wait for ring pieces it had just issued: ≈ 2 000 cycles per item.)
        if (NST >= 4 && issued) {
          if (wave == 14) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES0) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES1) : "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        FB_STAMP(0);
        bar();
        FB_STAMP(1);
        const int slot = n % NST;
        // the records of this iteration's stages, requested together
        const u32x4 r1 = stage_raw(min(n + 1, FB_MAXST - 1)), r2 = stage_raw(max(n - 1, 0)), rd = stage_raw(min(n + NST - 1, FB_MAXST - 1));
        if (wave == 15 && n + 1 < nstage && !ATTN_DBG(a, 16384)) cstage(stage_dec(r1), (slot + 1) % NST);
        FB_STAMP(2);
        if (n > 0 && !ATTN_DBG(a, 2048)) p2(stage_dec(r2), (n - 1) & 1);
        FB_STAMP(3);
        {
          // the next item's K image when P1 is at slab 1 of an item: the buffer's last reader was P2 of the item before, one iteration ago
          // (issued before the ring pieces: the wave's newest operations stay the ring's)
          if (wave == 15 && n < nstage) {
            const FbStage s0 = stage_dec(stage_raw(n));
            if (s0.s == 1 && s0.j + 1 < nit) issue_kimg(item(s0.j + 1), s0.j + 1);
          }
        }
        issued = n + NST - 1 < nstage;
        if (issued && !ATTN_DBG(a, 8192)) issue_stage(stage_dec(rd), (slot + NST - 1) % NST);
        FB_STAMP(4);
        ++itn;
      }
    }
    if (want_drel) {                                                 // (the barriers of the other waves' bias-gradient flush)
      __syncthreads();
      __syncthreads();
    }
    return;
  }
  // ================= waves 0 – 13: P1 =================
  // Lane constants of P1 as a table in LDS, read where they are used: held in registers across the slabs they would cost the dK / dV
  // accumulators their room (148 registers), rebuilt per slab ≈ 50 vector instructions of the ≈ 200 a slab takes — and P1 is bound by
  // vector issue.  Lane l = (l15, g) of key tile 0; the wave's tile adds multiples of 16 keys in scalar arithmetic.
  //   [0] row-fragment offset (step 1: ^ 64)   [1] byte offset of its reversed bias copy   [2] dS slab offset of q-tile 0 (q-tile 1: ^ 32)
  //   [3] dropout block index, lane part       [4..7] the four multipliers of its key column   [8] 5·((k & 3) >> 1)   [9] 16·(k & 1)
  //   [10] transposed-read offset of d-block 0 (d-block db: ^ (db << 5))
  if (wave == 0) {
    const int g = lane >> 4, l15 = lane & 15;
    const int last = a.R - 1 + EB_PADLO + 36;
    const int j0 = last - (l15 - 4 * g + a.rel_off + EB_PADLO);
    const int tr = 4 * g + (l15 >> 2);
    const DropCol dc = drop_col_consts(l15);
    uint32_t* e = ltab + lane * 12;
    e[0] = (uint32_t)(l15 * EROW + eswz(g, l15) * 16);
    e[1] = (uint32_t)(((j0 & 3) * EB_ST + (j0 & ~3)) * 4);
    e[2] = fb_ds_off(l15, 0, g);
    e[3] = (uint32_t)(g * KB + (l15 >> 2));
    e[4] = dc.m[0]; e[5] = dc.m[1]; e[6] = dc.m[2]; e[7] = dc.m[3];
    e[8] = 5u * (uint32_t)((l15 >> 1) & 1);
    e[9] = dc.sh;
    e[10] = (uint32_t)(tr * EROW + (((tr >> 1) & 3) << 5) + (l15 & 3) * 8);
    e[11] = 0u;
  }
  // bias-gradient accumulators: lane (key j = l15, group g) sums the elements with key − query = 16·(w − qt) + j − 4g of q-tile qt in
  // acc_lo[qt]; the elements that wrap around the 16-lane row (key − query = 16·(w − qt) + j − 16 − 4g, lanes 13..15) of the four
  // q-tiles 4u … 4u + 3 share acc_hi[u], q-tile 4u + k's three lanes moved down by 3k lanes (as in enc_bwd_dq_p_kernel)
  float acc_lo[16], acc_hi[4];
#pragma unroll
  for (int t = 0; t < 16; ++t) acc_lo[t] = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t) acc_hi[t] = 0.f;
  u32x4 vf[2] = {};
  f32x4 dkacc[4], dvacc[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) dkacc[db] = dvacc[db] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto load_v = [&](const FbItem& it) {                             // this wave's V rows of an item → vf
    const uint32_t vstb = (uint32_t)a.vst * 2u;
    eload_frags_buf(vf, a.v + h * 128 + (int64_t)it.k0 * vstb, vstb, wave * 16, it.Lk, (uint32_t)(lane & 15) * vstb + (uint32_t)(lane >> 4) * 16u);
  };

  for (int t0 = 0; t0 < nmine; t0 += FB_TBL) {
    // ---- tables of this pass: the non-empty items in walk order (sequence, q row offset, k row offset, Lq | Lk << 16) and their stages
    __syncthreads();                                                // the previous pass is done with the tables, the images and the slabs
    if (wave == 0) {
      const int t = t0 + lane;
      int b = 0, q0 = 0, k0 = 0, Lq = 0, Lk = 0;
      if (t < nmine) {
        const int bi = slot0 + t * nslots;
        b = a.order ? a.order[bi] : bi;
        q0 = a.q_off ? a.q_off[b] : b * a.rq;
        k0 = a.k_off ? a.k_off[b] : b * a.rk;
        Lq = a.q_off ? a.q_off[b + 1] - q0 : a.Lq;
        Lk = a.k_off ? a.k_off[b + 1] - k0 : a.Lk;
      }
      const bool ok = Lq > 0 && Lk > 0;
      const uint64_t m = __ballot(ok);
      const int pos = __popcll(m & ((1ull << lane) - 1ull));
      if (ok) {
        int* e = tbl + pos * 4;
        e[0] = b; e[1] = q0; e[2] = k0; e[3] = Lq | (Lk << 16);
      }
      const int ns = ok ? max(2, (Lq + 31) >> 5) : 0;                // slab iterations of the item (≥ 2: the K image double buffer)
      int incl = ns;                                                  // inclusive prefix sum over the lanes
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
      }
      const int first = incl - ns;
      for (int sl = 0; sl < ns; ++sl) {
        uint32_t* e = stab + (first + sl) * 4;
        e[0] = (uint32_t)b | ((uint32_t)sl << 16) | ((uint32_t)(pos & 1) << 24) | ((uint32_t)(sl == ns - 1) << 25);
        e[1] = (uint32_t)(q0 + 32 * sl);
        e[2] = (uint32_t)max(Lq - 32 * sl, 0) | ((uint32_t)Lk << 16);
        e[3] = (uint32_t)pos;
      }
      if (lane == 63) tbl[FB_TBL * 4 + 1] = incl;
      if (lane == 0) tbl[FB_TBL * 4] = (int)__popcll(m);
    }
    __syncthreads();
    const int nit = __builtin_amdgcn_readfirstlane(tbl[FB_TBL * 4]);
    const int nstage = __builtin_amdgcn_readfirstlane(tbl[FB_TBL * 4 + 1]);
    if (nit == 0) continue;
    // ---- P1: the wave's key tile against the slab of stage st (ring slot `slot`), dS → dS[par]
    auto p1 = [&](const FbStage& st, int slot, int par) {
      const int s = st.s;
      const int nkt2 = ((st.Lk + 31) >> 5) << 1;                      // key tiles, rounded up to whole pairs (P2 contracts over pairs)
      const char* lrow = reinterpret_cast<const char*>(ltab) + lane * 48;
      const u32x4 c0 = *reinterpret_cast<const u32x4*>(lrow);
      char* dS = dST0 + par * R * 64 + wave * 16 * 64;
      const uint32_t dsw0 = c0[2];                                    // (q-tile 1: ^ 32)
      if (wave * 16 >= st.Lk) {                                       // no key tile for this wave in this item
        if (wave < nkt2) {                                            // … but the odd tile of the last pair: its dS rows are zeros
          *reinterpret_cast<u32x2*>(dS + dsw0) = u32x2{0u, 0u};
          *reinterpret_cast<u32x2*>(dS + (dsw0 ^ 32u)) = u32x2{0u, 0u};
        }
        if (st.last && st.j + 1 < nit) load_v(item(st.j + 1));
        return;
      }
      const char* Qs = ring + slot * FB_STAGE;
      const char* dOs = Qs + FB_SLAB;
      if (ATTN_DBG(a, 4096)) return;                                  // (timing experiment: no P1 arithmetic)
      if (s == 0) {
#pragma unroll
        for (int db = 0; db < 4; ++db) dkacc[db] = dvacc[db] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      const uint32_t roff[2] = {c0[0], c0[0] ^ 64u};
      const char* Kimg = Kimg0 + st.kpar * R * EROW + wave * 16 * EROW;
      const bool tailk = (wave + 1) * 16 > st.Lk;                     // (wave-uniform) the tile holds keys past the sequence's end
      const char* bptr = reinterpret_cast<const char*>(b4) + c0[1] + (32 * s - 16 * wave) * 4;
      const char* ndp = reinterpret_cast<const char*>(ndel_l) + slot * 128 + (lane >> 4) * 16;       // this lane's 4 queries of q-tile 0 (q-tile 1: + 64)
      const char* lsp = reinterpret_cast<const char*>(lse_l) + slot * 128 + (lane >> 4) * 16;
      const uint32_t cblk = (uint32_t)((st.b * a.H + h) * QB + 8 * s) * (uint32_t)KB + 4u * (uint32_t)wave + c0[3];
      const uint32_t kb4 = 4u * (uint32_t)KB;
      const uint32_t t_hi = a.drop_t16 << 16;
      u32x2 ptb[2], dsb[2];                                            // P̃ and dS of the two tiles, rounded to bf16 as the MFMA operands are
      const u32x4 kfa = *reinterpret_cast<const u32x4*>(Kimg + roff[0]), kfb = *reinterpret_cast<const u32x4*>(Kimg + roff[1]);
      const u32x4 kf[2] = {kfa, kfb};
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        f32x4 sv, dp;
        {
          f32x4 init = *reinterpret_cast<const f32x4*>(bptr + 64 * u);
          if (tailk) {
            const float kadd_lane = wave * 16 + (lane & 15) < st.Lk ? 0.f : -INFINITY;
            init += f32x4{kadd_lane, kadd_lane, kadd_lane, kadd_lane};
          }
          const f32x4 nd = *reinterpret_cast<const f32x4*>(ndp + 64 * u);
          sv = escore(Qs, 16 * u, kf, roff, init);
          dp = escore(dOs, 16 * u, vf, roff, nd);
        }
        if (ATTN_DBG(a, 131072)) { asm volatile("" :: "v"(sv), "v"(dp)); if (u == 0) FB_STAMP(2); else FB_STAMP(4); }
        if (u == 1 && st.last && st.j + 1 < nit) {     // the V rows are dead: the next item's arrive under the rest of this slab's work
          __builtin_amdgcn_sched_barrier(0);
          load_v(item(st.j + 1));
          __builtin_amdgcn_sched_barrier(0);
        }
        const f32x4 l2 = *reinterpret_cast<const f32x4*>(lsp + 64 * u);
        f32x4 p, pt;
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[r], LOG2E, -l2[r]));
        pt = p;
        if (DROP && !ATTN_DBG(a, 1024)) {
          // the column form of the block recipe (attn_shared.h::drop_keep_col) with the lane's constants from the table: word (r, (k & 3) >> 1)
          // = mix(h, M[r][·], 6 + 2r + 5·((k & 3) >> 1)), its half k & 1 against the threshold
          const u32x4 c1 = *reinterpret_cast<const u32x4*>(lrow + 16);
          const u32x2 c2 = *reinterpret_cast<const u32x2*>(lrow + 32);
          const uint32_t hb = drop_base(cblk + (uint32_t)u * kb4, a.drop_key);
          const uint32_t hs5 = hb >> c2[0];
          const f32x4 nd = *reinterpret_cast<const f32x4*>(ndp + 64 * u);     // (re-read: 4 registers less across the hashes)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const uint32_t t = __umul24(hb, c1[r]) + (hs5 >> (6u + 2u * (uint32_t)r));
            const bool kp = ((t ^ (t >> 16)) << c2[1]) >= t_hi;
            pt[r] = kp ? p[r] : 0.f;
            dp[r] = kp ? dp[r] : nd[r];
          }
        }
        f32x4 ds;
#pragma unroll
        for (int r = 0; r < 4; ++r) ds[r] = p[r] * dp[r];
        // dS slab: the 4 queries of this lane's key, rounded as the MFMA operand of the dK product is
        const bf16x4 dsv = {(bf16_t)ds[0], (bf16_t)ds[1], (bf16_t)ds[2], (bf16_t)ds[3]};
        const bf16x4 ptv = {(bf16_t)pt[0], (bf16_t)pt[1], (bf16_t)pt[2], (bf16_t)pt[3]};
        dsb[u] = __builtin_bit_cast(u32x2, dsv);
        ptb[u] = __builtin_bit_cast(u32x2, ptv);
        *reinterpret_cast<u32x2*>(dS + (u ? dsw0 ^ 32u : dsw0)) = dsb[u];
        if (want_drel && !ATTN_DBG(a, 32768)) {
          // lane j' collects, by DPP row shifts, the elements (key j' + r, query 4g + r) of the tile: key − query = 16·(w − qt) + j' − 4g;
          // those that wrap around the row land in lanes 13..15 of `hs`
          float lo, hs;
          edpp_diag_sums(ds[0], ds[1], ds[2], ds[3], lo, hs);
          static_for<8>([&](auto S) {
            constexpr int sc = decltype(S)::value;
            if (s == sc) {
              acc_lo[2 * sc + u] += lo;
              // q-tile 4v + k of a group of four: moved down by 3k lanes
              constexpr int k3 = (2 * sc) & 3;
              if (u == 0) eacc_shl<3 * k3>(acc_hi[sc >> 1], hs);
              else eacc_shl<3 * (k3 + 1)>(acc_hi[sc >> 1], hs);
            }
          });
        }
        if (ATTN_DBG(a, 131072)) { if (u == 0) FB_STAMP(3); else FB_STAMP(5); }
        __builtin_amdgcn_sched_barrier(0);
      }
      {
        // transposed-read offsets: d-block db is ^ (db << 5)
        const uint32_t toff0 = *reinterpret_cast<const uint32_t*>(lrow + 40);
        const uint32_t toff[4] = {toff0, toff0 ^ 32u, toff0 ^ 64u, toff0 ^ 96u};
        if (!ATTN_DBG(a, 65536)) {
        epv_packed(dkacc, dsb[0], dsb[1], Qs, 0, toff);            // dKᵀ += Qᵀ·dS
        epv_packed(dvacc, ptb[0], ptb[1], dOs, 0, toff);           // dVᵀ += dOᵀ·P̃
        }
      }
      if (ATTN_DBG(a, 131072)) { asm volatile("" :: "v"(dkacc[0]), "v"(dvacc[3])); FB_STAMP(6); }
      if (st.last) {
        // dK / dV of the item.  A lane holds, per 16-wide d-block, the 4 d-values 4g … 4g + 3 of its key: stored as they are that is sixteen
        // 8-byte stores per wave and item, each touching 16 rows x 32 bytes — 2 000 – 4 000 cycles of store issue at every item boundary
        // (profiles/r05b_attn_fused_stamps.txt).  The lanes g and g ^ 1 of a key (lane ^ 16: one ds_swizzle) swap halves instead — the even
        // one ends up with the full 16 bytes of d-blocks 0 and 1, the odd one with those of d-blocks 2 and 3 — and the item leaves in eight
        // 16-byte stores per wave.  Same values, same bytes.
        const int g = lane >> 4, ki = wave * 16 + (lane & 15);
        const bool odd = g & 1;
        auto pack = [](f32x4 v) {
          const bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
          return __builtin_bit_cast(u32x2, o);
        };
        auto pair16 = [&](const f32x4 (&acc)[4], u32x4 (&out)[2]) {
          const u32x2 pk[4] = {pack(acc[0]), pack(acc[1]), pack(acc[2]), pack(acc[3])};
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const u32x2 send = odd ? pk[i] : pk[2 + i], keep = odd ? pk[2 + i] : pk[i];
            u32x2 recv;
            recv[0] = (uint32_t)__builtin_amdgcn_ds_swizzle((int)send[0], 0x401F);      // BITMASK_PERM: lane ^ 16
            recv[1] = (uint32_t)__builtin_amdgcn_ds_swizzle((int)send[1], 0x401F);
            out[i] = odd ? u32x4{recv[0], recv[1], keep[0], keep[1]} : u32x4{keep[0], keep[1], recv[0], recv[1]};
          }
        };
        u32x4 ko[2], vo[2];
        pair16(dkacc, ko);
        pair16(dvacc, vo);
        if (ki < st.Lk) {
          const int k0 = item(st.j).k0;
          // even lanes: d-blocks 0, 1 at d = 16i + 4g;  odd lanes: d-blocks 2, 3 at d = 16(2 + i) + 4(g − 1)
          const int dcol = odd ? 32 + 4 * (g - 1) : 4 * g;
          bf16_t* kp_ = reinterpret_cast<bf16_t*>(a.dk + h * 128 + (int64_t)(k0 + ki) * ((int64_t)a.kst * 2)) + dcol;
          bf16_t* vp_ = reinterpret_cast<bf16_t*>(a.dv + h * 128 + (int64_t)(k0 + ki) * ((int64_t)a.vst * 2)) + dcol;
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<u32x4*>(kp_ + 16 * i) = ko[i];
            *reinterpret_cast<u32x4*>(vp_ + 16 * i) = vo[i];
          }
        }
      }
    };
    // prologue: item 0's V rows
    load_v(item(0));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    FB_STAMP_AT(255, 1);
    for (int n = 0; n <= nstage; ++n) {
      FB_STAMP(0);
      bar();
      FB_STAMP(1);
      if (n < nstage) p1(stage_dec(stage_raw(n)), n % NST, n & 1);
      FB_STAMP(7);
      ++itn;
    }
  }
  FB_STAMP_AT(255, 2);
  if (want_drel) {
    // Flush without atomics (the two-pass kernels add every accumulator lane to an LDS table by ds_add_f32: ≈ 0.3 lane-operations per clock and
    // CU — 18 000 of them were ≈ 28 µs of a 270 µs launch): every wave lays its 20 accumulator registers out in LDS (the images are dead),
    // then thread d sums the ≤ 112 lanes that hold diagonal d in a FIXED order and adds the sum to the head's table with one global atomic.
    // acc_lo[qt], lane (j, g) of wave w: key − query = 16·(w − qt) + j − 4g; the wrapped elements of q-tile qt sit in acc_hi[qt >> 2] at
    // lane jj − 3·(qt & 3) (jj = 13..15) and belong to key − query = 16·(w − qt) + jj − 16 − 4g.
    float* F = reinterpret_cast<float*>(smem);            // [14 waves][20][64]
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 16; ++t) F[(wave * 20 + t) * 64 + lane] = acc_lo[t];
#pragma unroll
    for (int t = 0; t < 4; ++t) F[(wave * 20 + 16 + t) * 64 + lane] = acc_hi[t];
    __syncthreads();
    for (int d = threadIdx.x; d < a.R; d += FB_NW * 64) {
      const int dd = d - a.rel_off;                         // key − query
      float sum = 0.f;
      for (int w = 0; w < 14; ++w) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int x = dd + 4 * g, j = x & 15;             // 16·(w − qt) + j = x
          const int qt = w - ((x - j) >> 4);
          if (qt >= 0 && qt < 16) sum += F[(w * 20 + qt) * 64 + g * 16 + j];
          if (j >= 13) {                                    // the same lane index in the wrapped part: 16·(w − qt') + j − 16 = x
            const int q2 = qt - 1;
            if (q2 >= 0 && q2 < 16) sum += F[(w * 20 + 16 + (q2 >> 2)) * 64 + g * 16 + j - 3 * (q2 & 3)];
          }
        }
      }
      if (a.det) lako_det::add(&lako_det_slots[h * a.R + d], sum);      // LAKO_DETERMINISTIC: fixed point, converted once below
      else if (sum != 0.f) atomicAdd(a.drel + (int64_t)h * a.R + d, sum);
    }
    if (a.det) {
      // (the flag lives in the dynamic allocation — F[0], dead after the barrier inside last_block: a static __shared__ variable would move
      //  the base of `smem`, and the LDS-DMA destinations of this kernel are absolute LDS addresses)
      if (lako_det::last_block(&lako_det_ticket, gridDim.x, reinterpret_cast<int*>(smem)))
        for (int i = threadIdx.x; i < a.H * a.R; i += 14 * 64) a.drel[i] += lako_det::take(&lako_det_slots[i]);      // (waves 14 / 15 have left)
    }
  }
  FB_STAMP_AT(255, 3);
}

template <typename K>
void eset_lds(K kern, int bytes, lako_lds_cur_t& cur) {
  lako_raise_max_lds(reinterpret_cast<const void*>(kern), bytes, cur);
}

// query / key blocks per workgroup: all of them when there are already plenty of workgroups, otherwise split so that the grid
// reaches ~1024 workgroups (multiples of the workgroup's wave count)
inline int eblocks_per_wg(int nblocks, int64_t bh, int nw) {
  const int64_t want = 1024;
  if (bh >= want || nblocks <= nw) return ((nblocks + nw - 1) / nw) * nw;
  const int groups = (int)((want + bh - 1) / bh);
  int per = (nblocks + groups - 1) / groups;
  per = ((per + nw - 1) / nw) * nw;
  return per < nw ? nw : per;
}

#define EDISPATCH(KERN, NTHR, drop, grid, lds, s, args)                                                                          \
  do {                                                                                                                           \
    if (drop) { static lako_lds_cur_t c; eset_lds(&KERN<true>, lds, c); hipLaunchKernelGGL((KERN<true>), grid, dim3(NTHR), lds, s, args); }    \
    else { static lako_lds_cur_t c; eset_lds(&KERN<false>, lds, c); hipLaunchKernelGGL((KERN<false>), grid, dim3(NTHR), lds, s, args); }       \
  } while (0)

}  // namespace

#ifdef LAKO_EXPERIMENTS
// (experiments build only) cycle stamps of enc_bwd_fused_kernel: one static device buffer, read back by tools/attn_stamps.py
constexpr size_t FB_STAMP_BYTES = 16 * 256 * 8 * 8;
static void* lako_exp_stamp_buffer() {
  static void* buf = nullptr;
  if (!buf) {
    (void)hipMalloc(&buf, FB_STAMP_BYTES);
    (void)hipMemset(buf, 0, FB_STAMP_BYTES);
  }
  return buf;
}
constexpr size_t FWD_STAMP_BYTES = 4096 * 8 * 8 * 8;
static void* lako_exp_fwd_stamp_buffer() {
  static void* buf = nullptr;
  if (!buf) (void)hipMalloc(&buf, FWD_STAMP_BYTES);
  return buf;
}
extern "C" int lako_exp_attn_fwd_stamps(void* host_dst, size_t bytes) {
  (void)hipDeviceSynchronize();
  return (int)hipMemcpy(host_dst, lako_exp_fwd_stamp_buffer(), bytes < FWD_STAMP_BYTES ? bytes : FWD_STAMP_BYTES, hipMemcpyDeviceToHost);
}
extern "C" int lako_exp_attn_stamps(void* host_dst, size_t bytes) {
  (void)hipDeviceSynchronize();
  return (int)hipMemcpy(host_dst, lako_exp_stamp_buffer(), bytes < FB_STAMP_BYTES ? bytes : FB_STAMP_BYTES, hipMemcpyDeviceToHost);
}
#endif

// Shapes the fast path takes (attn.hip asks before falling back to its generic kernels).
bool lako_attn_enc_supported(const AttnArgs& a, int dtype, int d_head) {
  static const bool off = getenv("LAKO_ATTN_FAST") && atoi(getenv("LAKO_ATTN_FAST")) == 0;   // A/B against the generic kernels
  if (off) return false;
  if (dtype != LAKO_BF16 || d_head != 64 || a.causal || a.scores_out || a.key_mask) return false;
  if (a.Lk > 256 || a.Lq > 256 || a.Lk < 1) return false;
  if (a.rel_bias && !(a.R == a.Lq + a.Lk - 1 && a.rel_off == a.Lq - 1)) return false;   // the encoder's table: no index clamping
  if (a.Lq <= 16 && a.Lk > 64) return false;        // one query block over many keys: the generic key-split mode is the better fit
  if ((int64_t)a.Bn * a.H >= 65536) return false;
  return true;
}

// Without a bias table (R = 0, rel_off = 0) the kernels still read their four zero-filled copies at index key − query + rel_off + 16:
// with rel_off = 0 that is NEGATIVE for queries past 16 + 4g, i.e. the bytes in front of the copies — the tail rows of an image
// (zeros by luck while the sequence ends ≥ 4 rows before the image does; an image pair nothing was staged into in the persistent
// kernels: found by tests/test_kernels_gpu.py::test_attention[enc_fast_nobias-bf16_persistent]).  Index like the encoder's table.
static void enorm_nobias(AttnArgs& a) {
  if (!a.rel_bias) {
    a.R = a.Lq + a.Lk - 1;
    a.rel_off = a.Lq - 1;
    a.drel = nullptr;          // no table, no table gradient
  }
}

// Which passes run their persistent kernel: LAKO_ATTN_PERSIST is a bit mask — 1 forward, 2 dQ pass, 4 dK/dV pass, 8 = also for
// launches with fewer than 1024 (sequence, head) items (tests).  Default 2: measured per config-2 launch on MI355X (rocprofv3,
// profiles/r03d_attn_persistent_kernels.txt) the persistent dQ pass takes 194 µs against 249, the dK/dV pass 197 against 189 and the
// forward 146 against 126 — with one 16-wave workgroup per CU every item is a barrier-to-barrier phase whose length is ONE wave's
// dependency chain; the per-item kernels' two independent workgroups per CU hide that better than the next item's DMA does, except
// in the dQ pass, whose per-item kernel re-stages K / V three times per workgroup and keeps 4 query blocks per wave.
static bool epersist(int bit, const AttnArgs& a) {
  const char* pe = getenv("LAKO_ATTN_PERSIST");      // read per call: a test switches it
  const int m = pe ? atoi(pe) : 18;
  return (m & bit) && a.H <= 256 && ((int64_t)a.Bn * a.H >= 1024 || (m & 8));
}

int lako_attn_enc_fwd(AttnArgs& a, hipStream_t s) {
#ifdef LAKO_EXPERIMENTS
  static const int dbg = getenv("LAKO_ATTN_DEBUG") ? atoi(getenv("LAKO_ATTN_DEBUG")) : 0;
  a.dbg_flags = dbg;
  a.dbg_buf = (dbg & 262144) ? lako_exp_fwd_stamp_buffer() : nullptr;
#endif
  enorm_nobias(a);
  a.chunk_rows = ((a.Lk + 31) / 32) * 32;
  const int nqb = (a.Lq + 15) / 16;
  // persistent form: one 16-wave workgroup per CU walking the sequences of one head (see epersist)
  if (epersist(1, a) && (int64_t)a.kst * 2 * 256 < (1ll << 31) && (int64_t)a.vst * 2 * 256 < (1ll << 31)) {
    const int nslots = std::min(a.Bn, std::max(1, 256 / a.H));
    const int lds = efwdp_lds(a.chunk_rows);
    EDISPATCH(enc_fwd_p_kernel, PF_NW * 64, a.drop_t16 != 0, dim3(a.H * nslots), lds, s, a);
    return 0;
  }
  a.blocks_per_wg = eblocks_per_wg(nqb, (int64_t)a.Bn * a.H, FWD_NW);
  const int lds = efwd_lds(a.chunk_rows);
  const dim3 grid((nqb + a.blocks_per_wg - 1) / a.blocks_per_wg, a.H, a.Bn);
  // K / V by LDS-DMA, everything in flight together (LAKO_ATTN_FWD_DMA=0: register staging, three round trips)
  static const bool dma_off = getenv("LAKO_ATTN_FWD_DMA") && atoi(getenv("LAKO_ATTN_FWD_DMA")) == 0;
  if (!dma_off && (int64_t)a.kst * 2 * 256 < (1ll << 31) && (int64_t)a.vst * 2 * 256 < (1ll << 31)) {
    EDISPATCH(enc_fwd_c_kernel, FWD_NW * 64, a.drop_t16 != 0, grid, lds, s, a);
    return 0;
  }
  EDISPATCH(enc_fwd_kernel, FWD_NW * 64, a.drop_t16 != 0, grid, lds, s, a);
  return 0;
}

int lako_attn_enc_bwd(AttnArgs& a, hipStream_t s) {
#ifdef LAKO_EXPERIMENTS
  static const int dbg = getenv("LAKO_ATTN_DEBUG") ? atoi(getenv("LAKO_ATTN_DEBUG")) : 0;
  a.dbg_flags = dbg;
  if (dbg & 256) goto dkv_pass;      // time the passes separately
#endif
  enorm_nobias(a);
  // one-pass backward (round 5, the default where its LDS layout fits): bit 16 of LAKO_ATTN_PERSIST (default 18 = persistent dQ pass as the
  // fallback + the fused kernel); a ring of 4 stages where 160 KiB hold it (<= 256 key rows do), else 3
  {
    const char* pe = getenv("LAKO_ATTN_PERSIST");      // read per call: a test switches it
    const int m = pe ? atoi(pe) : 18;
    const int rows = ((a.Lk + 31) / 32) * 32;
    const bool fits32 = (int64_t)a.kst * 2 * 256 < (1ll << 31) && (int64_t)a.vst * 2 * 256 < (1ll << 31) && (int64_t)a.qst * 2 * 256 < (1ll << 31) &&
                        (int64_t)a.ost * 2 * 256 < (1ll << 31);
    // one addressing form for ragged and padded launches (row offset × token stride): the batch strides must be whole rows, the same number
    // on every tensor of a side; dq / dk / dv share the strides of q / k / v by contract
    const bool ragged = a.q_off && a.k_off;
    const bool padded = !a.q_off && !a.k_off && a.qst > 0 && a.ost > 0 && a.kst > 0 && a.vst > 0 && a.qsb % a.qst == 0 && a.osb % a.ost == 0 &&
                        a.qsb / a.qst == a.osb / a.ost && a.ksb % a.kst == 0 && a.vsb % a.vst == 0 && a.ksb / a.kst == a.vsb / a.vst &&
                        a.qsb / a.qst * a.Bn < (1ll << 31) && a.ksb / a.kst * a.Bn < (1ll << 31);
    if ((m & 16 || a.det) && a.H <= 256 && fits32 && (ragged || padded) && rows <= 224 && a.Lq <= 256 && efb_lds(rows, 3) <= 160 * 1024 &&
        (!a.det || a.H * a.R <= lako_det::SLOTS)) {
      AttnArgs f = a;
      f.chunk_rows = rows;
      f.rq = padded ? (int)(a.qsb / a.qst) : 0;
      f.rk = padded ? (int)(a.ksb / a.kst) : 0;
      const int nslots = std::min(a.Bn, std::max(1, 256 / a.H));
      const bool drop = a.drop_t16 != 0;
      // (read on every call, as LAKO_ATTN_PERSIST above: tests switch the ring depth inside one process — cached in a static the first
      //  fused launch would fix it for the whole run and the three-stage instantiations would never be exercised)
      const char* nst_s = getenv("LAKO_ATTN_FUSED_NST");
      const int nst_env = nst_s ? atoi(nst_s) : 4;
      const int nst = (nst_env >= 4 && efb_lds(rows, 4) <= 160 * 1024) ? 4 : 3;
      const int lds = efb_lds(rows, nst);
      const dim3 grid(a.H * nslots);
#ifdef LAKO_EXPERIMENTS
      if (f.dbg_flags & 131072) f.scores_out = reinterpret_cast<float*>(lako_exp_stamp_buffer());
#endif
#define EFB(D, N)                                                                                          \
  do {                                                                                                     \
    static lako_lds_cur_t c;                                                                               \
    eset_lds(&enc_bwd_fused_kernel<D, N>, lds, c);                                                         \
    hipLaunchKernelGGL((enc_bwd_fused_kernel<D, N>), grid, dim3(FB_NW * 64), lds, s, f);                   \
  } while (0)
      if (nst == 4) { if (drop) EFB(true, 4); else EFB(false, 4); }
      else { if (drop) EFB(true, 3); else EFB(false, 3); }
#undef EFB
      return 0;
    }
  }
  // LAKO_DETERMINISTIC: the two-pass kernels below add into the workgroup's bias-gradient table by LDS float atomics in arrival order —
  // not taken in that mode: the caller runs the generic kernels (attn.hip), whose contributions go through det.h
  if (a.det) return 1;
  {  // dQ pass: every query block of a sequence in one workgroup, several batch rows per workgroup (bias gradient in registers)
    AttnArgs q = a;
    q.chunk_rows = ((a.Lk + 31) / 32) * 32;
    if (epersist(2, a) && (int64_t)a.kst * 2 * 256 < (1ll << 31) && (int64_t)a.vst * 2 * 256 < (1ll << 31)) {
      const int nslots = std::min(a.Bn, std::max(1, 256 / a.H));
      EDISPATCH(enc_bwd_dq_p_kernel, PF_NW * 64, a.drop_t16 != 0, dim3(a.H * nslots), ebwd0p_lds(q.chunk_rows), s, q);
    } else {
      const int64_t wgs = (int64_t)a.Bn * a.H;
      static const int wg_target = getenv("LAKO_ATTN_WGS") ? atoi(getenv("LAKO_ATTN_WGS")) : 1024;
      q.bn_per_wg = (int)(wgs / wg_target > 1 ? wgs / wg_target : 1);
      if (q.bn_per_wg > OFFS_MAX - 1) q.bn_per_wg = OFFS_MAX - 1;
      const int lds = ebwd0_lds(q.chunk_rows);
      const dim3 grid(1, a.H, (a.Bn + q.bn_per_wg - 1) / q.bn_per_wg);
      EDISPATCH(enc_bwd_dq_kernel, 256, a.drop_t16 != 0, grid, lds, s, q);
    }
  }
#ifdef LAKO_EXPERIMENTS
  if (a.dbg_flags & 512) return 0;
dkv_pass:
#endif
  {  // dK/dV pass
    AttnArgs k = a;
    k.chunk_rows = ((a.Lq + 31) / 32) * 32;
    const int nkb = (a.Lk + 15) / 16;
    if (epersist(4, a) && (int64_t)a.qst * 2 * 256 < (1ll << 31) && (int64_t)a.ost * 2 * 256 < (1ll << 31)) {
      const int nslots = std::min(a.Bn, std::max(1, 256 / a.H));
      const int lds = ebwd1p_lds(k.chunk_rows);
      EDISPATCH(enc_bwd_dkv_p_kernel, PF_NW * 64, a.drop_t16 != 0, dim3(a.H * nslots), lds, s, k);
      return 0;
    }
    static const bool dkv_dma_off = getenv("LAKO_ATTN_DKV_DMA") && atoi(getenv("LAKO_ATTN_DKV_DMA")) == 0;
    k.dma_stage = !dkv_dma_off && (int64_t)a.qst * 2 * 256 < (1ll << 31) && (int64_t)a.ost * 2 * 256 < (1ll << 31);
    k.blocks_per_wg = eblocks_per_wg(nkb, (int64_t)a.Bn * a.H, DKV_NW);
    const int lds = ebwd1_lds(k.chunk_rows);
    const dim3 grid((nkb + k.blocks_per_wg - 1) / k.blocks_per_wg, a.H, a.Bn);
    EDISPATCH(enc_bwd_dkv_kernel, DKV_NW * 64, a.drop_t16 != 0, grid, lds, s, k);
  }
  return 0;
}
