// Cross-attention over the concatenated n_passages·L encoder states, computed IN THE ENCODER-STATE SPACE
// (src/model.py:286-349 / HF5:281-369 for the decoder's EncDecAttention; bf16, d_kv 64, ragged keys).
//
// The reference projects every encoder state to a key and a value for every decoder layer (K = E·Wkᵀ, V = E·Wvᵀ: two
// [B·N·L, d] × [d, d] products per layer — 1/7 of the reader's GEMM work, and a [B·N·L, 2·L_dec·d] buffer written, read by the
// attention, and walked twice more in backward) although only T·H ≈ 100 query rows per sample ever look at them.  The same
// numbers come out of
//     scores[t,h,s] = q[t,h,:]·(Wk_h·E[s,:]) = (q[t,h,:]·Wk_h)·E[s,:]          Q'[t,h,:] = q[t,h,:]·Wk_h   ∈ R^d   ("expand")
//     ctx[t,h,:]    = Σ_s P[t,h,s]·(Wv_h·E[s,:]) = Wv_h·(Σ_s P[t,h,s]·E[s,:])  C'[t,h,:] = Σ_s P·E[s,:]    ∈ R^d   ("contract")
// i.e. the per-head projections are applied to the R = T·H query rows of a sample instead of to its N·L keys, and the
// attention itself becomes two thin products against the encoder output E, which every head and every layer shares:
//     S  [R, keys] = Q'·Eᵀ         (xscores_kernel:  a 128-key tile of one sample per workgroup, K = d in 64-wide LDS slices)
//     P  = dropout(softmax(S))      (xsoftmax_fwd_kernel: one row per workgroup; the attention-dropout block recipe of attn_shared.h)
//     C' [R, d]    = P·E           (xcontext_kernel: 64 output columns × a key range per workgroup, E read transposed from LDS)
// Backward is the same three kernels on (dC', dS) plus ONE weight-gradient-style product for the encoder states after the last
// decoder layer,  dE[s,:] = Σ_layers Σ_r P[r,s]·dC'[r,:] + dS[r,s]·Q'[r,:]  (lako_gemm_tn_grouped over the samples, K = layers·2R),
// and three head-batched small products (hb_nt_kernel / hb_tn_kernel) for Q', dC', ctx, dq and the Wk / Wv gradients.
// FLOPs per decoder layer and sample: 4·R·keys·d instead of 4·keys·d² forward (R = 96 vs d = 768 at T5-base), no K/V buffer.
//
// Layouts.  Scores / probabilities are [R, P_total] matrices whose columns are the keys of all samples side by side, each sample's
// segment padded to a multiple of 256 columns (p_off; the padding holds zeros) so that tiles and 16-byte accesses never straddle
// two samples; E is the packed [rows, d] encoder output with k_off[b] the first row of sample b (engine.py::_ragged_batch).
#include "attn_shared.h"
#include "lds_image.h"

namespace {

struct XArgs {
  const char* q;        // bf16 query-side rows: sample b, row r at q + (b·q_sb + r·q_ld)·2
  int64_t q_sb, q_ld;
  const char* e;        // bf16 encoder states [rows, D], row stride e_ld
  int64_t e_ld;
  const int32_t *k_off, *p_off;   // [B + 1] first encoder row / first padded score column of sample b
  float* s;             // fp32 [R, s_ld] scores (xscores) / input rows (softmax)
  int64_t s_ld;
  const char* p;        // bf16 [R, p_ld] probabilities (xcontext input)
  int64_t p_ld;
  float* out;           // fp32 context rows: key split z, sample b, row r at out + z·out_zs + b·out_sb + r·out_ld
  int64_t out_zs, out_sb, out_ld;
  int R, D, B, Z;
};

// --------------------------------------------------------------------------------------------------------------------
// LDS-DMA staging of 128-byte image rows (buffer_load … lds, common.h): one wave instruction moves 8 image rows (1 KiB) straight
// from HBM / L2 into LDS; the image swizzle is applied on the global side (the lane that fills chunk position cp of row ρ fetches
// chunk eswz(cp, ρ) — the swizzle is an involution).  Rows at or past `valid` read as zeros (out-of-range buffer offset).
// PERM: image row ρ of a 32-row block holds source row 8·((ρ & 15) >> 2) + 4·(ρ >> 4) + (ρ & 3) of the block, so that the
// 8 K-values a lane group g contributes to an MFMA over 32 rows (transposed reads: rows 4g … 4g+3 and 16 + 4g … 16 + 4g+3) are the
// 8 CONSECUTIVE source rows 8g … 8g+7 — the other operand is then one aligned 16-byte read of a row-major [·, 32] block.
// --------------------------------------------------------------------------------------------------------------------
template <bool PERM>
__device__ __forceinline__ void xdma8(char* img, const lako_u32x4_t& rsrc, uint32_t stride_b, int inst, int lane, int valid) {
  const int rho = inst * 8 + (lane >> 3), cp = lane & 7;
  int src = rho;
  if (PERM) src = (rho & ~31) + 8 * ((rho & 15) >> 2) + 4 * ((rho >> 4) & 1) + (rho & 3);
  const uint32_t voff = src < valid ? (uint32_t)src * stride_b + (uint32_t)eswz(cp, rho) * 16u : 0xFFFFFFF0u;
  lds_dma16(img + inst * 1024, rsrc, voff);
}
__device__ __forceinline__ lako_u32x4_t xrsrc(const char* base, int valid, uint32_t stride_b) {
  return lds_dma_rsrc(base, valid > 0 ? (uint32_t)(valid - 1) * stride_b + 128u : 0u);
}
template <int N>
__device__ __forceinline__ void xwait_vm() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}

__device__ __forceinline__ int xsample_of(const int32_t* p_off, int B, int col0, int* slot) {
  for (int i = threadIdx.x; i < B; i += blockDim.x)
    if (p_off[i] <= col0 && col0 < p_off[i + 1]) *slot = i;
  __syncthreads();
  return __builtin_amdgcn_readfirstlane(*slot);
}

// --------------------------------------------------------------------------------------------------------------------
// S[r, p_off[b] + s] = Σ_c Q'[b][r][c]·E[k_off[b] + s][c]       grid (P_total / 256, ⌈R / 128⌉), 512 threads
// A workgroup owns 256 keys of one sample × up to 128 query rows; the contraction over D runs in 64-wide slices through a ring of
// three LDS stages (two key images + the query image, 48 KiB each) filled by LDS-DMA two slices ahead.  Wave w: keys 32w … 32w+31.
// --------------------------------------------------------------------------------------------------------------------
constexpr int XS_IMG = 128 * EROW;          // one 128-row image
constexpr int XS_STAGE = 3 * XS_IMG, XS_NSTG = 3;

__global__ __launch_bounds__(512, 1) void xscores_kernel(XArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  __shared__ int slot;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  const int col0 = blockIdx.x * 256;
  if (col0 >= __builtin_amdgcn_readfirstlane(a.p_off[a.B])) return;   // launched over the padded batch's columns (captured decode steps)
  const int b = xsample_of(a.p_off, a.B, col0, &slot);
  const int key0 = col0 - __builtin_amdgcn_readfirstlane(a.p_off[b]);
  const int kfirst = __builtin_amdgcn_readfirstlane(a.k_off[b]) + key0;
  const int nk = min(256, __builtin_amdgcn_readfirstlane(a.k_off[b + 1]) - kfirst);
  const int r0 = blockIdx.y * 128, nr = min(128, a.R - r0), nrt = (nr + 15) >> 4;
  const char* ebase = a.e + (int64_t)kfirst * a.e_ld * 2;
  const char* qbase = a.q + ((int64_t)b * a.q_sb + (int64_t)r0 * a.q_ld) * 2;
  const uint32_t est = (uint32_t)(a.e_ld * 2), qst = (uint32_t)(a.q_ld * 2);
  const ELane el = elane(lane);
  const int nsl = a.D >> 6;

  auto issue = [&](int sl) {
    char* stage = smem + (sl % XS_NSTG) * XS_STAGE;
    const lako_u32x4_t ra = xrsrc(ebase + sl * 128, min(nk, 128), est);
    const lako_u32x4_t rb = xrsrc(ebase + (int64_t)128 * est + sl * 128, nk - 128, est);
    const lako_u32x4_t rq = xrsrc(qbase + sl * 128, nr, qst);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int j = wave * 6 + i;     // 48 instructions per stage: 16 per image
      if (j < 16) xdma8<false>(stage, ra, est, j, lane, nk);
      else if (j < 32) xdma8<false>(stage + XS_IMG, rb, est, j - 16, lane, nk - 128);
      else xdma8<false>(stage + 2 * XS_IMG, rq, qst, j - 32, lane, nr);
    }
  };

  f32x4 acc[2][8];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int rt = 0; rt < 8; ++rt) acc[kb][rt] = f32x4{0.f, 0.f, 0.f, 0.f};

  issue(0);
  if (nsl > 1) issue(1);
  for (int sl = 0; sl < nsl; ++sl) {
    if (sl + 1 < nsl) xwait_vm<6>();
    else xwait_vm<0>();
    __syncthreads();                       // stage sl has landed for every wave; everyone is done with stage sl − 1
    if (sl + 2 < nsl) issue(sl + 2);
    const char* stage = smem + (sl % XS_NSTG) * XS_STAGE;
    const char* eimg = stage + (wave >> 2) * XS_IMG + (wave & 3) * 32 * EROW;
    const char* qimg = stage + 2 * XS_IMG;
    u32x4 ef[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int i = 0; i < 2; ++i) ef[kb][i] = *reinterpret_cast<const u32x4*>(eimg + 16 * kb * EROW + el.roff[i]);
#pragma unroll
    for (int rt = 0; rt < 8; ++rt) {
      if (rt < nrt) {
        const u32x4 q0 = *reinterpret_cast<const u32x4*>(qimg + rt * 16 * EROW + el.roff[0]);
        const u32x4 q1 = *reinterpret_cast<const u32x4*>(qimg + rt * 16 * EROW + el.roff[1]);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          acc[kb][rt] = emma(ef[kb][0], q0, acc[kb][rt]);
          acc[kb][rt] = emma(ef[kb][1], q1, acc[kb][rt]);
        }
      }
    }
  }
  // acc[kb][rt][i] = S[r0 + 16·rt + l15][col0 + 32·wave + 16·kb + 4g + i]; keys past the sample's end were staged as zero rows and
  // land in the segment's padding columns
#pragma unroll
  for (int rt = 0; rt < 8; ++rt) {
    const int row = r0 + rt * 16 + l15;
    if (rt < nrt && row < a.R) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
        *reinterpret_cast<f32x4*>(a.s + (int64_t)row * a.s_ld + col0 + 32 * wave + 16 * kb + 4 * g) = acc[kb][rt];
    }
  }
}

// --------------------------------------------------------------------------------------------------------------------
// out[z][b][r][c0 .. c0+127] = Σ_{keys of key range z of sample b} P[r][p_off[b] + s]·E[k_off[b] + s][c]
// grid (D / 128, B·Z, ⌈R / 128⌉), 512 threads.  The keys run in stages of 64 through a ring of four LDS stages (two 64-key × 64-column
// images of E with permuted rows + the [128 rows × 64 keys] image of P, 32 KiB) filled by LDS-DMA three stages ahead.
// Wave w: column half w >> 2, row tiles 2(w & 3) and 2(w & 3) + 1 — every output element has one owner, no cross-wave reduction.
// --------------------------------------------------------------------------------------------------------------------
constexpr int XC_NSTG = 4, XC_STAGE = 32768, XC_EIMG = 64 * EROW;
constexpr int XC_TST = 68;   // floats per row of a wave's write-out block

__global__ __launch_bounds__(512, 1) void xcontext_kernel(XArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  const int c0 = blockIdx.x * 128;
  const int b = blockIdx.y / a.Z, z = blockIdx.y - b * a.Z;
  const int kfirst = __builtin_amdgcn_readfirstlane(a.k_off[b]);
  const int Sb = __builtin_amdgcn_readfirstlane(a.k_off[b + 1]) - kfirst;
  const int nst_all = (Sb + 63) >> 6;
  const int st_begin = (int)((int64_t)z * nst_all / a.Z), st_end = (int)((int64_t)(z + 1) * nst_all / a.Z);
  const int nst = st_end - st_begin;     // may be 0 (fewer stages than splits): the slab still gets its zeros
  const int r0 = blockIdx.z * 128, nr = min(128, a.R - r0), nrt = (nr + 15) >> 4;
  const char* ebase = a.e + ((int64_t)kfirst * a.e_ld + c0) * 2;
  const uint32_t est = (uint32_t)(a.e_ld * 2), pst = (uint32_t)(a.p_ld * 2);
  const char* pbase = a.p + ((int64_t)r0 * a.p_ld + __builtin_amdgcn_readfirstlane(a.p_off[b])) * 2;
  const ELane el = elane(lane);
  const int ch = wave >> 2, rt0 = 2 * (wave & 3);

  auto issue = [&](int i) {      // stage i of this workgroup = keys (st_begin + i)·64 …
    char* stage = smem + (i % XC_NSTG) * XC_STAGE;
    const int key0 = (st_begin + i) * 64, valid = min(64, Sb - key0);
    if (wave < 4) {              // waves 0,1: columns c0 … c0+63; waves 2,3: c0+64 …: 8 instructions per image
      const int half = wave >> 1;
      const lako_u32x4_t re = xrsrc(ebase + (int64_t)key0 * est + half * 128, valid, est);
#pragma unroll
      for (int k = 0; k < 4; ++k) xdma8<true>(stage + half * XC_EIMG, re, est, (wave & 1) * 4 + k, lane, valid);
    } else {                     // waves 4..7: the 16 instructions of the probability image
      const lako_u32x4_t rp = xrsrc(pbase + key0 * 2, nr, pst);
#pragma unroll
      for (int k = 0; k < 4; ++k) xdma8<false>(stage + 2 * XC_EIMG, rp, pst, (wave - 4) * 4 + k, lane, nr);
    }
  };

  f32x4 acc[2][4];
#pragma unroll
  for (int rtl = 0; rtl < 2; ++rtl)
#pragma unroll
    for (int db = 0; db < 4; ++db) acc[rtl][db] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (nst > 0) issue(0);
  if (nst > 1) issue(1);
  if (nst > 2) issue(2);
  for (int i = 0; i < nst; ++i) {
    const int ahead = min(2, nst - 1 - i);
    if (ahead == 2) xwait_vm<8>();
    else if (ahead == 1) xwait_vm<4>();
    else xwait_vm<0>();
    __syncthreads();
    if (i + 3 < nst) issue(i + 3);
    const char* stage = smem + (i % XC_NSTG) * XC_STAGE;
    const char* eimg = stage + ch * XC_EIMG;
    const char* pimg = stage + 2 * XC_EIMG;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      u32x4 af[4];
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const char* ap = eimg + 32 * kb * EROW + el.toff[db];
        s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ap));
        s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ap + 16 * EROW));
        const u32x2 u0 = __builtin_bit_cast(u32x2, t0), u1 = __builtin_bit_cast(u32x2, t1);
        af[db] = u32x4{u0[0], u0[1], u1[0], u1[1]};
      }
#pragma unroll
      for (int rtl = 0; rtl < 2; ++rtl) {
        if (rt0 + rtl < nrt) {
          const u32x4 pf = *reinterpret_cast<const u32x4*>(pimg + (rt0 + rtl) * 16 * EROW + el.roff[kb]);
#pragma unroll
          for (int db = 0; db < 4; ++db) acc[rtl][db] = emma(af[db], pf, acc[rtl][db]);
        }
      }
    }
  }
  // acc[rtl][db][i] = out[r0 + 16·(rt0 + rtl) + l15][c0 + 64·ch + 16·db + 4g + i] over this workgroup's keys.  Each wave turns its
  // 32 × 64 block around in LDS so that one store instruction writes 64 CONSECUTIVE floats of a row (in the accumulator layout an
  // instruction scatters 16-byte pieces over 16 rows, and the write-out cost more than the product).  Key splits write separate
  // slabs that the consumer adds in a fixed order: no atomics, results do not depend on the order workgroups finish in.
  __syncthreads();                                   // every wave is done with the stages
  float* tr = reinterpret_cast<float*>(smem) + wave * (32 * XC_TST);
#pragma unroll
  for (int rtl = 0; rtl < 2; ++rtl)
#pragma unroll
    for (int db = 0; db < 4; ++db) *reinterpret_cast<f32x4*>(tr + (rtl * 16 + l15) * XC_TST + 16 * db + 4 * g) = acc[rtl][db];
  const int nrows = min(32, nr - rt0 * 16);          // rows of this wave that exist
  float* op = a.out + (int64_t)z * a.out_zs + (int64_t)b * a.out_sb + (int64_t)(r0 + rt0 * 16) * a.out_ld + c0 + 64 * ch + lane;
  for (int k = 0; k < nrows; ++k) op[(int64_t)k * a.out_ld] = tr[k * XC_TST + lane];
}

// --------------------------------------------------------------------------------------------------------------------
// One decode step (T = 1: R = H ≤ 16 query rows per sample) in ONE pass over the encoder states: scores, softmax and context of a
// key range per workgroup, flash-decoding style — the encoder states are read once instead of once by the scores and once by the
// context kernel, and no score / probability matrix exists.  grid (Z key ranges, B), 512 threads.
//   keys run in stages of 32 through two LDS stages ([32 keys × D] as D/64 images with PERMUTED rows, filled by LDS-DMA one stage
//   ahead); wave w owns the D/8 columns [w·D/8, (w+1)·D/8) of E for BOTH products: its slice of the contraction of S = Q′·Eᵀ (the
//   eight partial score tiles meet in LDS) and its columns of C′ = P·E; the summed score tile is already the context product's
//   operand (registers: 4 keys of tile 0 and of tile 1 per lane — the transposed reads' K order), online softmax per row.
//   out: part_ml[z][b][r] = (row max, Σ exp), part_c[z][b][r][:] = Σ_keys exp(s − max)·E[key][:]   (xdecode_combine_kernel merges)
// --------------------------------------------------------------------------------------------------------------------
struct XdArgs {
  const char* q;        // bf16 [B][R][D] expanded queries (row stride q_ld, sample stride q_sb)
  int64_t q_sb, q_ld;
  const char* e;
  int64_t e_ld;
  const int32_t* k_off;
  float* part_ml;       // [Z][B][16][2]
  float* part_c;        // [Z][B][16][D]
  int R, D, B, Z;
};

template <int KS>      // K-steps (of 32 dims) per wave: D = 256·KS
__global__ __launch_bounds__(512, 1) void xdecode_kernel(XdArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];   // 2 stages [32][D] | partial score tiles [8][32][16] fp32
  constexpr int D = 256 * KS, STAGE = 32 * D * 2, NIMG = D / 64, NDB = D / 128;   // NDB: 16-column blocks per wave
  constexpr int NSTG = 3 * STAGE + 8 * 32 * 16 * 4 <= 160 * 1024 ? 3 : 2;        // stages of the ring (one or two in flight)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  const int z = blockIdx.x, b = blockIdx.y;
  const int kfirst = __builtin_amdgcn_readfirstlane(a.k_off[b]);
  const int Sb = __builtin_amdgcn_readfirstlane(a.k_off[b + 1]) - kfirst;
  const int nst_all = (Sb + 31) >> 5;
  const int st_begin = (int)((int64_t)z * nst_all / a.Z), st_end = (int)((int64_t)(z + 1) * nst_all / a.Z);
  const int nst = st_end - st_begin;
  const char* ebase = a.e + (int64_t)kfirst * a.e_ld * 2;
  const uint32_t est = (uint32_t)(a.e_ld * 2);
  float* sred = reinterpret_cast<float*>(smem + NSTG * STAGE);

  // the wave's slice of the expanded queries: row l15, dims (wave·KS + t)·32 + 8g … as MFMA fragments
  u32x4 qf[KS];
#pragma unroll
  for (int t = 0; t < KS; ++t) {
    qf[t] = u32x4{0u, 0u, 0u, 0u};
    if (l15 < a.R) qf[t] = *reinterpret_cast<const u32x4*>(a.q + ((int64_t)b * a.q_sb + (int64_t)l15 * a.q_ld + (wave * KS + t) * 32 + 8 * g) * 2);
  }
  // a USE of the loaded registers that the compiler sees: it waits for these loads HERE, not at their first use inside the loop
  // (an s_waitcnt vmcnt(0) there would also wait for every LDS-DMA stage in flight)
#pragma unroll
  for (int t = 0; t < KS; ++t) asm volatile("" : "+v"(qf[t]));

  auto issue = [&](int i) {      // stage i of this workgroup: keys (st_begin + i)·32 …, NIMG images of 4 instructions
    char* stage = smem + (i % NSTG) * STAGE;
    const int key0 = (st_begin + i) * 32, valid = min(32, Sb - key0);
#pragma unroll
    for (int k = 0; k < NIMG / 2; ++k) {
      const int j = wave * (NIMG / 2) + k;          // instruction j: image j >> 2, rows 8·(j & 3) …
      const lako_u32x4_t re = xrsrc(ebase + (int64_t)key0 * est + (j >> 2) * 128, valid, est);
      xdma8<true>(stage + (j >> 2) * 32 * EROW, re, est, j & 3, lane, valid);
    }
  };

  f32x4 cacc[NDB];
  // byte offset of the wave's column block db inside a stage (image + transposed-read offset), computed — NOT el.toff[runtime index]:
  // at D = 768 the block index depends on the wave, a dynamically indexed register array lands in scratch memory, and every
  // scratch reload is a vector-memory operation whose s_waitcnt vmcnt(0) also waits for the LDS-DMA stages in flight (the ring
  // ran one stage at a time: 31.7 µs instead of 17 for config 2's decode step)
  uint32_t coff[NDB];
#pragma unroll
  for (int db = 0; db < NDB; ++db) {
    cacc[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int col = wave * (D / 8) + db * 16, tr = 4 * g + (l15 >> 2);
    coff[db] = (uint32_t)((col >> 6) * 32 * EROW + tr * EROW + ((((col & 63) >> 4) ^ ((tr >> 1) & 3)) << 5) + (l15 & 3) * 8);
  }
  uint32_t soff[KS];      // 32-dim step wave·KS + t of the contraction: image (step >> 1), row-fragment half (step & 1) — computed too
#pragma unroll
  for (int t = 0; t < KS; ++t) {
    const int sidx = wave * KS + t;
    soff[t] = (uint32_t)((sidx >> 1) * 32 * EROW + l15 * EROW + eswz((sidx & 1) * 4 + g, l15) * 16);
  }
  float m_run = -INFINITY, l_run = 0.f;

  if (nst > 0) issue(0);
  if (NSTG == 3 && nst > 1) issue(1);
  for (int i = 0; i < nst; ++i) {
    if (NSTG == 3 && i + 1 < nst) xwait_vm<NIMG / 2>();      // stage i + 1 (this wave's NIMG / 2 instructions) may still be in flight
    else xwait_vm<0>();
    __syncthreads();                       // stage i has landed; every wave is past stage i − 1 (and past reading its score tiles)
    if (i + NSTG - 1 < nst) issue(i + NSTG - 1);
    const char* stage = smem + (i % NSTG) * STAGE;
    // partial scores over this wave's dims: tile kt, lane (r = l15, g): image rows 16·kt + 4g … + 3
    f32x4 sp[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int t = 0; t < KS; ++t)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
        sp[kt] = emma(*reinterpret_cast<const u32x4*>(stage + soff[t] + 16 * kt * EROW), qf[t], sp[kt]);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) *reinterpret_cast<f32x4*>(sred + ((wave * 2 + kt) * 16 + l15) * 16 + 4 * g) = sp[kt];
    __syncthreads();
    f32x4 sv[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      sv[kt] = *reinterpret_cast<const f32x4*>(sred + (kt * 16 + l15) * 16 + 4 * g);
#pragma unroll
      for (int w = 1; w < 8; ++w) sv[kt] += *reinterpret_cast<const f32x4*>(sred + ((w * 2 + kt) * 16 + l15) * 16 + 4 * g);
    }
    // image row 16·kt + 4g + e holds key 8g + 4·kt + e of the stage (xdma8<PERM>)
    const int kbase = (st_begin + i) * 32 + 8 * g;
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (kbase + 4 * kt + e >= Sb) sv[kt][e] = -INFINITY;
        mx = fmaxf(mx, sv[kt][e]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);                 // finite: every stage of a range holds at least one key
    const float alpha = __expf(m_run - m_new);
    bf16x8 pf;
    float ps = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float pv = __expf(sv[kt][e] - m_new);
        ps += pv;
        pf[kt * 4 + e] = (bf16_t)pv;
      }
    l_run = l_run * alpha + ps;
    m_run = m_new;
    const u32x4 pfrag = __builtin_bit_cast(u32x4, pf);
#pragma unroll
    for (int db = 0; db < NDB; ++db) {
      const char* ap = stage + coff[db];
      s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ap));
      s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ap + 16 * EROW));
      const u32x2 u0 = __builtin_bit_cast(u32x2, t0), u1 = __builtin_bit_cast(u32x2, t1);
      cacc[db] = emma(u32x4{u0[0], u0[1], u1[0], u1[1]}, pfrag, cacc[db] * alpha);
    }
  }
  // lane (r = l15, g): cacc[db][i] = Σ exp(s − m)·E[key][wave·D/8 + 16·db + 4g + i]
  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
  const int64_t pr = ((int64_t)z * a.B + b) * 16 + l15;
  if (wave == 0 && g == 0) {
    a.part_ml[pr * 2] = m_run;
    a.part_ml[pr * 2 + 1] = l_run;
  }
#pragma unroll
  for (int db = 0; db < NDB; ++db) *reinterpret_cast<f32x4*>(a.part_c + pr * D + wave * (D / 8) + db * 16 + 4 * g) = cacc[db];
}

// ctx[b][h][j] = (Σ_z w_z·C′_z[b][h][:] / Σ_z w_z·l_z)·Wv[h·64 + j][:],  w_z = exp(m_z − max_z m_z):  grid (H, B), 256 threads
struct XcArgs {
  const float *part_ml, *part_c;
  const char* wv;       // bf16 [H·64][ldw] rows of the layer's V projection
  int64_t ldw;
  char* ctx;            // bf16 [B][H·64] (row stride ctx_ld)
  int64_t ctx_ld;
  int D, B, Z;
};

template <int SPAN8>     // 16-byte weight loads per thread: D = 32·SPAN8
__global__ __launch_bounds__(256) void xdecode_combine_kernel(XcArgs a) {
  // every global load of a thread — the ranges' statistics, its 4 columns of up to 16 partial contexts, its quarter of a weight
  // row — is issued before the first use: the kernel is a handful of dependent latencies otherwise (12 µs for 0.2 MFLOP)
  __shared__ float cl[1024];
  __shared__ float wz[64], lz[64];
  const int h = blockIdx.x, b = blockIdx.y;
  const int j = threadIdx.x >> 2, qd = threadIdx.x & 3, span = SPAN8 * 8;
  const char* wr = a.wv + ((int64_t)(h * 64 + j) * a.ldw + qd * span) * 2;
  u32x4 w8[SPAN8];
#pragma unroll
  for (int u = 0; u < SPAN8; ++u) w8[u] = *reinterpret_cast<const u32x4*>(wr + u * 16);
  const int c = threadIdx.x * 4;
  f32x4 t[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    t[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (u < a.Z && c < a.D) t[u] = *reinterpret_cast<const f32x4*>(a.part_c + (((int64_t)u * a.B + b) * 16 + h) * a.D + c);
  }
  if (threadIdx.x < 64) {
    float mz = -INFINITY, l = 0.f;
    if ((int)threadIdx.x < a.Z) {
      const float* ml = a.part_ml + (((int64_t)threadIdx.x * a.B + b) * 16 + h) * 2;
      mz = ml[0];
      l = ml[1];
    }
    float M = mz;                                   // max over the ranges: one wave holds them all
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) M = fmaxf(M, __shfl_xor(M, o, 64));
    wz[threadIdx.x] = mz == -INFINITY ? 0.f : __expf(mz - M);
    lz[threadIdx.x] = l;
  }
  __syncthreads();
  float L = 0.f;
  for (int z = 0; z < a.Z; ++z) L += wz[z] * lz[z];
  const float invL = 1.0f / L;
  if (c < a.D) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 16; ++u) v += t[u] * wz[u];
    for (int z0 = 16; z0 < a.Z; z0 += 16) {          // more than 16 key ranges: further batches
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (z0 + u < a.Z) v += *reinterpret_cast<const f32x4*>(a.part_c + (((int64_t)(z0 + u) * a.B + b) * 16 + h) * a.D + c) * wz[z0 + u];
    }
    *reinterpret_cast<f32x4*>(cl + c) = v * invL;
  }
  __syncthreads();
  float acc = 0.f;
#pragma unroll
  for (int u = 0; u < SPAN8; ++u) {
    const bf16x8 wv8 = __builtin_bit_cast(bf16x8, w8[u]);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc = __builtin_fmaf((float)wv8[e], cl[qd * span + u * 8 + e], acc);
  }
  acc += __shfl_xor(acc, 1, 64);
  acc += __shfl_xor(acc, 2, 64);
  if (qd == 0) reinterpret_cast<bf16_t*>(a.ctx)[(int64_t)b * a.ctx_ld + h * 64 + j] = (bf16_t)acc;
}

// --------------------------------------------------------------------------------------------------------------------
// softmax over the keys of one (sample, row): grid (R, B), 256 threads
// --------------------------------------------------------------------------------------------------------------------
struct SmArgs {
  const float* s;       // [R, s_ld] raw scores
  const float* dp;      // [R, s_ld] d(loss)/d(dropped probabilities)     (backward)
  int64_t s_ld;
  float* stats;         // [B, R, 2] = (row max, 1 / row sum)
  char* p;              // bf16 [R, p_ld]: forward dropout(P); backward dS
  int64_t p_ld;
  const int32_t *k_off, *p_off;
  int R, H, T, B, KB, QB;
  uint32_t drop_t16, drop_key;
  float drop_scale;
};

__device__ __forceinline__ float xblock_max(float v, float* sh) {
  v = wave_max(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  v = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
  __syncthreads();
  return v;
}
__device__ __forceinline__ float xblock_sum(float v, float* sh) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  v = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  __syncthreads();
  return v;
}

// keep flags of keys 4c … 4c+3 of row r = t·H + h of sample b: the block recipe of attn_shared.h with bh = b·H + h, q = t, k = s
__device__ __forceinline__ void xkeep4(const SmArgs& a, uint32_t prow, const DropRow& drc, int c, bool (&kp)[4]) {
  drop_keep_row(drop_base(prow + (uint32_t)c, a.drop_key), drc, a.drop_t16 << 16, kp);
}

// Rows of at most 1024·NV keys (config 2: 4 000): the row — and in backward dP — is read ONCE into registers (NV 16-byte vectors per
// thread, all loads in flight together) instead of two / three times from L2 with a reduction between the passes.
template <bool BWD, int NV>
__global__ __launch_bounds__(256) void xsoftmax_reg_kernel(SmArgs a) {
  __shared__ float sh[4];
  const int r = blockIdx.x, b = blockIdx.y;
  const int t = r / a.H, h = r - t * a.H;
  const int pc0 = a.p_off[b], Sb = a.k_off[b + 1] - a.k_off[b], Sp = a.p_off[b + 1] - pc0;
  const float* srow = a.s + (int64_t)r * a.s_ld + pc0;
  const float* drow = BWD ? a.dp + (int64_t)r * a.s_ld + pc0 : nullptr;
  bf16_t* prow_out = reinterpret_cast<bf16_t*>(a.p + ((int64_t)r * a.p_ld + pc0) * 2);
  const uint32_t prow = (uint32_t)(((b * a.H + h) * a.QB + (t >> 2)) * a.KB);
  const DropRow drc = drop_row_consts(t);
  const int n4 = (Sb + 3) >> 2, np4 = Sp >> 2;
  f32x4 v[NV], d[NV];
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    const int c = threadIdx.x + u * 256;
    v[u] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    d[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < n4) {
      v[u] = *reinterpret_cast<const f32x4*>(srow + 4 * c);
      if (BWD) d[u] = *reinterpret_cast<const f32x4*>(drow + 4 * c);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (4 * c + i >= Sb) v[u][i] = -INFINITY;
    }
  }
  float m, inv;
  if (!BWD) {
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < NV; ++u) mx = fmaxf(fmaxf(mx, fmaxf(v[u][0], v[u][1])), fmaxf(v[u][2], v[u][3]));
    m = xblock_max(mx, sh);
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < NV; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[u][i] = __expf(v[u][i] - m);        // exp(−inf) = 0 for keys past the end
        sum += v[u][i];
      }
    sum = xblock_sum(sum, sh);
    inv = 1.0f / sum;
    if (threadIdx.x == 0) {
      a.stats[((int64_t)b * a.R + r) * 2] = m;
      a.stats[((int64_t)b * a.R + r) * 2 + 1] = inv;
    }
  } else {
    m = a.stats[((int64_t)b * a.R + r) * 2];
    inv = a.stats[((int64_t)b * a.R + r) * 2 + 1];
#pragma unroll
    for (int u = 0; u < NV; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) v[u][i] = __expf(v[u][i] - m);
  }
  const float keep_scale = a.drop_t16 ? a.drop_scale : 1.0f;
  float delta = 0.f;
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    const int c = threadIdx.x + u * 256;
    bool kp[4] = {true, true, true, true};
    if (a.drop_t16 && c < n4) xkeep4(a, prow, drc, c, kp);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float p = v[u][i] * inv;
      if (!BWD) v[u][i] = kp[i] ? p * keep_scale : 0.f;
      else {
        d[u][i] = kp[i] ? keep_scale * d[u][i] : 0.f;      // dropout_bwd(dP)
        delta += p * d[u][i];
        v[u][i] = p;
      }
    }
  }
  if (BWD) {
    delta = xblock_sum(delta, sh);
#pragma unroll
    for (int u = 0; u < NV; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) v[u][i] = v[u][i] * (d[u][i] - delta);
  }
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    const int c = threadIdx.x + u * 256;
    if (c < np4) store4(prow_out + 4 * c, c < n4 ? v[u] : f32x4{0.f, 0.f, 0.f, 0.f});
  }
}

template <bool BWD>
__global__ __launch_bounds__(256) void xsoftmax_kernel(SmArgs a) {
  __shared__ float sh[4];
  const int r = blockIdx.x, b = blockIdx.y;
  const int t = r / a.H, h = r - t * a.H;
  const int pc0 = a.p_off[b], Sb = a.k_off[b + 1] - a.k_off[b], Sp = a.p_off[b + 1] - pc0;   // Sp: padded segment length
  const float* srow = a.s + (int64_t)r * a.s_ld + pc0;
  char* prow_out = a.p + ((int64_t)r * a.p_ld + pc0) * 2;
  const uint32_t prow = (uint32_t)(((b * a.H + h) * a.QB + (t >> 2)) * a.KB);
  const DropRow drc = drop_row_consts(t);
  const int n4 = (Sb + 3) >> 2;
  float m, inv;
  if (!BWD) {
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < n4; c += 256) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(srow + 4 * c);
#pragma unroll
      for (int i = 0; i < 4; ++i) mx = fmaxf(mx, 4 * c + i < Sb ? v[i] : -INFINITY);
    }
    m = xblock_max(mx, sh);
    float sum = 0.f;
    for (int c = threadIdx.x; c < n4; c += 256) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(srow + 4 * c);
#pragma unroll
      for (int i = 0; i < 4; ++i) sum += 4 * c + i < Sb ? __expf(v[i] - m) : 0.f;
    }
    sum = xblock_sum(sum, sh);
    inv = 1.0f / sum;
    if (threadIdx.x == 0) {
      a.stats[((int64_t)b * a.R + r) * 2] = m;
      a.stats[((int64_t)b * a.R + r) * 2 + 1] = inv;
    }
  } else {
    m = a.stats[((int64_t)b * a.R + r) * 2];
    inv = a.stats[((int64_t)b * a.R + r) * 2 + 1];
  }
  const float keep_scale = a.drop_t16 ? a.drop_scale : 1.0f;
  float delta = 0.f;
  if (BWD) {
    const float* drow = a.dp + (int64_t)r * a.s_ld + pc0;
    for (int c = threadIdx.x; c < n4; c += 256) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(srow + 4 * c);
      const f32x4 d = *reinterpret_cast<const f32x4*>(drow + 4 * c);
      bool kp[4] = {true, true, true, true};
      if (a.drop_t16) xkeep4(a, prow, drc, c, kp);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (4 * c + i < Sb && kp[i]) delta += __expf(v[i] - m) * inv * keep_scale * d[i];
    }
    delta = xblock_sum(delta, sh);
  }
  for (int c = threadIdx.x; c < (Sp >> 2); c += 256) {
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (c < n4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(srow + 4 * c);
      bool kp[4] = {true, true, true, true};
      if (a.drop_t16) xkeep4(a, prow, drc, c, kp);
      if (!BWD) {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (4 * c + i < Sb && kp[i]) ? __expf(v[i] - m) * inv * keep_scale : 0.f;
      } else {
        const f32x4 d = *reinterpret_cast<const f32x4*>(a.dp + (int64_t)r * a.s_ld + pc0 + 4 * c);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          o[i] = 4 * c + i < Sb ? __expf(v[i] - m) * inv * ((kp[i] ? keep_scale * d[i] : 0.f) - delta) : 0.f;
      }
    }
    store4(reinterpret_cast<bf16_t*>(prow_out) + 4 * c, o);
  }
}

// --------------------------------------------------------------------------------------------------------------------
// head-batched small products (M = B·T decoder rows)
//   hb_nt:  C_h[m, n] = Σ_k A_h[m, k]·B_h[n, k]           one wave per 16 × (16·NT) tile, operands straight from HBM
//   hb_tn:  C_h[j, c] += Σ_m A[m, h·64 + j]·B_h[m, c]     one workgroup per (64-column block, head), both operands transposed in LDS
// rows are addressed in two levels (sample, row in the sample) so that they can live inside per-sample blocks of a bigger buffer
// --------------------------------------------------------------------------------------------------------------------
struct HbArgs {
  const char* A;
  int64_t a_sb, a_st, a_sh;   // element offset of row m = (b, t), head h: b·a_sb + t·a_st + h·a_sh
  const char* B;
  int64_t b_sh, ldb;          // hb_nt: bf16 B_h[n, k] at h·b_sh + n·ldb + k;  hb_tn: rows like A's: b·b_sb + t·b_st + h·b_sh
  int64_t b_sb, b_st;
  char* C;
  int64_t c_sb, c_st, c_sh;   // hb_nt: bf16 rows like A's;  hb_tn: fp32 C_h[j, c] at h·c_sh + j·c_st + c
  int M, T, N, K, H;
  int n_slabs;                // the fp32 operand (hb_nt: A, hb_tn: B) is the sum of n_slabs buffers slab_stride elements apart
  int64_t slab_stride;        // (the key splits of xcontext_kernel), added in slab order
};

__device__ __forceinline__ u32x4 hb_load8(const char* base, int64_t off, bool f32, bool ok, int n_slabs = 1, int64_t slab_stride = 0) {
  u32x4 v = {0u, 0u, 0u, 0u};
  if (!ok) return v;
  if (!f32) return *reinterpret_cast<const u32x4*>(base + off * 2);
  f32x4 x = *reinterpret_cast<const f32x4*>(base + off * 4);
  f32x4 y = *reinterpret_cast<const f32x4*>(base + off * 4 + 16);
  for (int z = 1; z < n_slabs; ++z) {
    x += *reinterpret_cast<const f32x4*>(base + (off + z * slab_stride) * 4);
    y += *reinterpret_cast<const f32x4*>(base + (off + z * slab_stride) * 4 + 16);
  }
  bf16x8 o = {(bf16_t)x[0], (bf16_t)x[1], (bf16_t)x[2], (bf16_t)x[3], (bf16_t)y[0], (bf16_t)y[1], (bf16_t)y[2], (bf16_t)y[3]};
  return __builtin_bit_cast(u32x4, o);
}

template <bool AF32, int NT, int KU>   // KU K-steps (of 32) are loaded together before their MFMAs: K % (32·KU) == 0
__global__ __launch_bounds__(256) void hb_nt_kernel(HbArgs a) {
  const int lane = threadIdx.x & 63, g = lane >> 4, l15 = lane & 15;
  const int tiles_m = (a.M + 15) >> 4, tiles_n = a.N / (16 * NT);
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= tiles_m * tiles_n * a.H) return;
  const int h = wid / (tiles_m * tiles_n), rem = wid - h * tiles_m * tiles_n;
  const int mt = rem / tiles_n, n0 = (rem - mt * tiles_n) * 16 * NT;
  const int m = mt * 16 + l15;
  const bool mok = m < a.M;
  const int mb = m / a.T, mtt = m - mb * a.T;
  const int64_t aoff = (int64_t)mb * a.a_sb + (int64_t)mtt * a.a_st + (int64_t)h * a.a_sh + 8 * g;
  const int64_t boff = (int64_t)h * a.b_sh + (int64_t)(n0 + l15) * a.ldb + 8 * g;
  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < a.K; k0 += 32 * KU) {
    u32x4 af[KU], bf[KU][NT];
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      af[u] = hb_load8(a.A, aoff + k0 + 32 * u, AF32, mok, a.n_slabs, a.slab_stride);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        bf[u][nt] = *reinterpret_cast<const u32x4*>(a.B + (boff + (int64_t)nt * 16 * a.ldb + k0 + 32 * u) * 2);
    }
#pragma unroll
    for (int u = 0; u < KU; ++u)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = emma(bf[u][nt], af[u], acc[nt]);     // acc[nt][i] = C[m][n0 + 16·nt + 4g + i]
  }
  if (mok) {
    bf16_t* crow = reinterpret_cast<bf16_t*>(a.C) + (int64_t)mb * a.c_sb + (int64_t)mtt * a.c_st + (int64_t)h * a.c_sh + n0 + 4 * g;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) store4(crow + nt * 16, acc[nt]);
  }
}

// "contract" shape (N = 64 columns per head, K = d): one workgroup per 16 × 16 output tile, its four waves take a quarter of K each —
// ALL of a wave's operand loads (KS K-steps × (up to four key-split slabs of A + B)) are in flight at once, one latency instead of
// K / 256 dependent batches — and add their partial tiles through LDS.  (The one-wave-per-tile form above took 18 µs for
// 0.3 GFLOP at config 2; with the slabs read one after the other and 64 workgroups, T5-large took 15 µs.)
template <bool AF32, int KS>
__global__ __launch_bounds__(256) void hb_contract_kernel(HbArgs a) {
  __shared__ __attribute__((aligned(16))) float part[4][16][20];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, l15 = lane & 15;
  const int mt = blockIdx.x, h = blockIdx.y, nt = blockIdx.z;        // one 16 × 16 output tile per workgroup
  const int m = mt * 16 + l15;
  const bool mok = m < a.M;
  const int mb = m / a.T, mtt = m - mb * a.T;
  const int kbase = wave * KS * 32;
  const int64_t aoff = (int64_t)mb * a.a_sb + (int64_t)mtt * a.a_st + (int64_t)h * a.a_sh + 8 * g + kbase;
  const int64_t boff = (int64_t)h * a.b_sh + (int64_t)(nt * 16 + l15) * a.ldb + 8 * g + kbase;
  // every load of the wave in flight at once: KS K-steps × (up to 4 slabs of A, two 16-byte halves each when fp32) + KS of B
  constexpr int NSL = AF32 ? 4 : 1;
  f32x4 ax[KS][NSL], ay[KS][NSL];
  u32x4 ab[KS], bf[KS];
#pragma unroll
  for (int u = 0; u < KS; ++u) {
    bf[u] = *reinterpret_cast<const u32x4*>(a.B + (boff + 32 * u) * 2);
    if (AF32) {
#pragma unroll
      for (int z = 0; z < NSL; ++z) {
        ax[u][z] = ay[u][z] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (mok && z < a.n_slabs) {
          const char* ap = a.A + (aoff + 32 * u + z * a.slab_stride) * 4;
          ax[u][z] = *reinterpret_cast<const f32x4*>(ap);
          ay[u][z] = *reinterpret_cast<const f32x4*>(ap + 16);
        }
      }
    } else {
      ab[u] = mok ? *reinterpret_cast<const u32x4*>(a.A + (aoff + 32 * u) * 2) : u32x4{0u, 0u, 0u, 0u};
    }
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < KS; ++u) {
    u32x4 af;
    if (AF32) {
      f32x4 x = ax[u][0], y = ay[u][0];
#pragma unroll
      for (int z = 1; z < NSL; ++z) { x += ax[u][z]; y += ay[u][z]; }
      if (a.n_slabs > NSL) {                      // more than four key-split slabs: the rest in order (rare)
        for (int z = NSL; z < a.n_slabs; ++z) {
          const char* ap = a.A + (aoff + 32 * u + z * a.slab_stride) * 4;
          if (mok) { x += *reinterpret_cast<const f32x4*>(ap); y += *reinterpret_cast<const f32x4*>(ap + 16); }
        }
      }
      bf16x8 o = {(bf16_t)x[0], (bf16_t)x[1], (bf16_t)x[2], (bf16_t)x[3], (bf16_t)y[0], (bf16_t)y[1], (bf16_t)y[2], (bf16_t)y[3]};
      af = __builtin_bit_cast(u32x4, o);
    } else {
      af = ab[u];
    }
    acc = emma(bf[u], af, acc);                   // acc[i] = C[m = l15][16·nt + 4g + i]
  }
  *reinterpret_cast<f32x4*>(&part[wave][l15][4 * g]) = acc;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int row = threadIdx.x >> 2, c4 = (threadIdx.x & 3) * 4;
    const int mr = mt * 16 + row;
    if (mr < a.M) {
      f32x4 v = *reinterpret_cast<const f32x4*>(&part[0][row][c4]);
#pragma unroll
      for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const f32x4*>(&part[w][row][c4]);
      const int rb = mr / a.T, rt = mr - rb * a.T;
      store4(reinterpret_cast<bf16_t*>(a.C) + (int64_t)rb * a.c_sb + (int64_t)rt * a.c_st + (int64_t)h * a.c_sh + nt * 16 + c4, v);
    }
  }
}

template <bool BF32>
__device__ __forceinline__ void hb_tn_body(const HbArgs& a, char* img) {      // img: A image [128 m][64 j] | B image [128 m][64 c]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  const int c0 = blockIdx.x * 64, h = blockIdx.y;
  const ELane el = elane(lane);
  const uint32_t toff_w = wave == 0 ? el.toff[0] : wave == 1 ? el.toff[1] : wave == 2 ? el.toff[2] : el.toff[3];
  f32x4 acc[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) acc[db] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int m0 = 0; m0 < a.M; m0 += 128) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = threadIdx.x + u * 256, row = idx >> 3, c = idx & 7, m = m0 + row;
      const bool ok = m < a.M;
      const int mb = ok ? m / a.T : 0, mtt = ok ? m - mb * a.T : 0;
      const u32x4 va = hb_load8(a.A, (int64_t)mb * a.a_sb + (int64_t)mtt * a.a_st + (int64_t)h * a.a_sh + c * 8, false, ok);
      const u32x4 vb = hb_load8(a.B, (int64_t)mb * a.b_sb + (int64_t)mtt * a.b_st + (int64_t)h * a.b_sh + c0 + c * 8, BF32, ok,
                                a.n_slabs, a.slab_stride);
      *reinterpret_cast<u32x4*>(img + row * EROW + eswz(c, row) * 16) = va;
      *reinterpret_cast<u32x4*>(img + XS_IMG + row * EROW + eswz(c, row) * 16) = vb;
    }
    __syncthreads();
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      if (m0 + blk * 32 < a.M) {
        auto tread = [&](const char* base, uint32_t off) {
          const char* ap = base + blk * 32 * EROW + off;
          s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ap));
          s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ap + 16 * EROW));
          const u32x2 u0 = __builtin_bit_cast(u32x2, t0), u1 = __builtin_bit_cast(u32x2, t1);
          return u32x4{u0[0], u0[1], u1[0], u1[1]};
        };
        const u32x4 bf = tread(img + XS_IMG, toff_w);
#pragma unroll
        for (int db = 0; db < 4; ++db) acc[db] = emma(tread(img, el.toff[db]), bf, acc[db]);
      }
    }
  }
  // acc[db][i] = Σ_m A[m][h·64 + 16·db + 4g + i]·B[m][c0 + 16·wave + l15]
  float* cbase = reinterpret_cast<float*>(a.C) + (int64_t)h * a.c_sh + c0 + 16 * wave + l15;
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      atomicAdd(cbase + (int64_t)(db * 16 + 4 * g + i) * a.c_st, acc[db][i]);   // one contributor per element and launch: no
                                                                                // read-modify-write round trip, still deterministic
    }
}

template <bool BF32>
__global__ __launch_bounds__(256) void hb_tn_kernel(HbArgs a) {
  __shared__ __attribute__((aligned(16))) char img[2 * XS_IMG];
  hb_tn_body<BF32>(a, img);
}

// several problems of one shape in ONE launch (round 4: the Wk / Wv gradients of all decoder layers after the last layer's backward —
// 24 launches of ≈ 8 µs for ≈ 0.1 GFLOP each were nothing but their dispatch): blockIdx.z picks the problem
constexpr int HB_MULTI_MAX = 24;
struct HbMulti { HbArgs it[HB_MULTI_MAX]; };
template <bool BF32>
__global__ __launch_bounds__(256) void hb_tn_multi_kernel(HbMulti m) {
  __shared__ __attribute__((aligned(16))) char img[2 * XS_IMG];
  hb_tn_body<BF32>(m.it[blockIdx.z], img);
}

int xcheck_common(const char* fn, const void* e, int64_t e_ld, const int32_t* k_off, const int32_t* p_off, int R, int D, int B) {
  if (!e || !k_off || !p_off) { lako_set_error("%s: null operand", fn); return LAKO_E_BADARG; }
  if (R <= 0 || B <= 0 || B > 512 || D <= 0 || D % 128 != 0 || e_ld % 8 != 0 || e_ld < D || e_ld >= (1 << 20)) {
    lako_set_error("%s: bad shape (R=%d B=%d (<= 512) D=%d (%% 128) e_ld=%lld (%% 8))", fn, R, B, D, (long long)e_ld);
    return LAKO_E_BADARG;
  }
  if ((reinterpret_cast<uintptr_t>(e) & 15) != 0) { lako_set_error("%s: encoder states not 16-byte aligned", fn); return LAKO_E_ALIGN; }
  return LAKO_OK;
}

void xset_lds(const void* fn, int bytes, lako_lds_cur_t& done) { lako_raise_max_lds(fn, bytes, done); }

}  // namespace

extern "C" int lako_xattn_scores(const void* q, int64_t q_sb, int64_t q_ld, const void* e, int64_t e_ld, const int32_t* k_off,
                                 const int32_t* p_off, int64_t p_total, float* s, int64_t s_ld, int R, int D, int B,
                                 lako_stream_t stream) {
  int rc = xcheck_common("lako_xattn_scores", e, e_ld, k_off, p_off, R, D, B);
  if (rc) return rc;
  LAKO_CHECK_ARG(q && s && p_total > 0 && p_total % 256 == 0 && s_ld >= p_total && s_ld % 4 == 0 && q_ld % 8 == 0 && q_sb % 8 == 0 &&
                     q_ld < (1 << 20),
                 "lako_xattn_scores: p_total %% 256, s_ld %% 4, q_ld / q_sb %% 8 (p_total=%lld s_ld=%lld)", (long long)p_total,
                 (long long)s_ld);
  LAKO_CHECK_ALIGN(q, 16);
  LAKO_CHECK_ALIGN(s, 16);
  XArgs a{};
  a.q = (const char*)q; a.q_sb = q_sb; a.q_ld = q_ld;
  a.e = (const char*)e; a.e_ld = e_ld;
  a.k_off = k_off; a.p_off = p_off;
  a.s = s; a.s_ld = s_ld;
  a.R = R; a.D = D; a.B = B; a.Z = 1;
  static lako_lds_cur_t done;
  xset_lds(reinterpret_cast<const void*>(&xscores_kernel), XS_NSTG * XS_STAGE, done);
  hipLaunchKernelGGL(xscores_kernel, dim3((unsigned)(p_total / 256), (unsigned)((R + 127) / 128)), dim3(512), XS_NSTG * XS_STAGE,
                     (hipStream_t)stream, a);
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

extern "C" int lako_xattn_context(const void* p, int64_t p_ld, const void* e, int64_t e_ld, const int32_t* k_off,
                                  const int32_t* p_off, float* out, int64_t out_zs, int64_t out_sb, int64_t out_ld, int R, int D,
                                  int B, int key_splits, lako_stream_t stream) {
  int rc = xcheck_common("lako_xattn_context", e, e_ld, k_off, p_off, R, D, B);
  if (rc) return rc;
  LAKO_CHECK_ARG(p && out && p_ld % 8 == 0 && p_ld < (1 << 23) && out_ld % 4 == 0 && out_sb % 4 == 0 && out_zs % 4 == 0 &&
                     key_splits >= 1 && key_splits <= 64,
                 "lako_xattn_context: p_ld %% 8, out_ld / out_sb / out_zs %% 4, 1 <= key_splits <= 64");
  LAKO_CHECK_ALIGN(p, 16);
  LAKO_CHECK_ALIGN(out, 16);
  XArgs a{};
  a.p = (const char*)p; a.p_ld = p_ld;
  a.e = (const char*)e; a.e_ld = e_ld;
  a.k_off = k_off; a.p_off = p_off;
  a.out = out; a.out_zs = out_zs; a.out_sb = out_sb; a.out_ld = out_ld;
  a.R = R; a.D = D; a.B = B; a.Z = key_splits;
  const int lds = XC_NSTG * XC_STAGE;
  static lako_lds_cur_t done;
  xset_lds(reinterpret_cast<const void*>(&xcontext_kernel), lds, done);
  hipLaunchKernelGGL(xcontext_kernel, dim3((unsigned)(D / 128), (unsigned)(B * key_splits), (unsigned)((R + 127) / 128)),
                     dim3(512), lds, (hipStream_t)stream, a);
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

static int xsoftmax_args(const char* fn, SmArgs& a, const float* s, int64_t s_ld, float* stats, void* p, int64_t p_ld,
                         const int32_t* k_off, const int32_t* p_off, int B, int T, int H, int max_keys, lako_dropout_t drop) {
  if (!s || !stats || !p || !k_off || !p_off || B <= 0 || T <= 0 || H <= 0 || max_keys <= 0 || s_ld % 4 != 0 || p_ld % 4 != 0) {
    lako_set_error("%s: null operand or bad shape", fn);
    return LAKO_E_BADARG;
  }
  if (!(drop.p >= 0.f && drop.p < 1.f)) { lako_set_error("%s: dropout p out of range", fn); return LAKO_E_BADARG; }
  if (drop.p > 0.f && (int64_t)B * H * ((T + 3) / 4) * ((max_keys + 3) / 4) >= (1ll << 32)) {
    lako_set_error("%s: dropout block index exceeds 32 bits", fn);
    return LAKO_E_UNSUPPORTED;
  }
  a.s = s; a.s_ld = s_ld; a.stats = stats; a.p = (char*)p; a.p_ld = p_ld; a.k_off = k_off; a.p_off = p_off;
  a.R = T * H; a.H = H; a.T = T; a.B = B;
  a.KB = (max_keys + 3) >> 2; a.QB = (T + 3) >> 2;
  a.drop_t16 = 0; a.drop_scale = 1.0f;
  if (drop.p > 0.f) {
    uint32_t t = (uint32_t)((double)drop.p * 65536.0 + 0.5);
    a.drop_t16 = t < 1 ? 1 : (t > 65535 ? 65535 : t);
    a.drop_scale = 1.0f / (1.0f - drop.p);
  }
  a.drop_key = lako_drop_key(drop.seed, drop.site);
  return LAKO_OK;
}

extern "C" int lako_xattn_softmax_fwd(const float* s, int64_t s_ld, float* stats, void* p, int64_t p_ld, const int32_t* k_off,
                                      const int32_t* p_off, int B, int T, int H, int max_keys, lako_dropout_t drop,
                                      lako_stream_t stream) {
  SmArgs a{};
  int rc = xsoftmax_args("lako_xattn_softmax_fwd", a, s, s_ld, stats, p, p_ld, k_off, p_off, B, T, H, max_keys, drop);
  if (rc) return rc;
  LAKO_CHECK_ALIGN(s, 16);
  LAKO_CHECK_ALIGN(p, 8);
  const dim3 grid((unsigned)(T * H), (unsigned)B);
  const int seg = (max_keys + 255) / 256 * 256;     // no sample's padded segment is longer
  if (seg <= 4096) hipLaunchKernelGGL((xsoftmax_reg_kernel<false, 4>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else if (seg <= 8192) hipLaunchKernelGGL((xsoftmax_reg_kernel<false, 8>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((xsoftmax_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, a);
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

extern "C" int lako_xattn_softmax_bwd(const float* s, const float* dp, int64_t s_ld, const float* stats, void* ds, int64_t p_ld,
                                      const int32_t* k_off, const int32_t* p_off, int B, int T, int H, int max_keys,
                                      lako_dropout_t drop, lako_stream_t stream) {
  SmArgs a{};
  int rc = xsoftmax_args("lako_xattn_softmax_bwd", a, s, s_ld, const_cast<float*>(stats), ds, p_ld, k_off, p_off, B, T, H,
                         max_keys, drop);
  if (rc) return rc;
  LAKO_CHECK_ARG(dp != nullptr, "lako_xattn_softmax_bwd: null dp");
  LAKO_CHECK_ALIGN(s, 16);
  LAKO_CHECK_ALIGN(dp, 16);
  LAKO_CHECK_ALIGN(ds, 8);
  a.dp = dp;
  const dim3 grid((unsigned)(T * H), (unsigned)B);
  const int seg = (max_keys + 255) / 256 * 256;
  if (seg <= 4096) hipLaunchKernelGGL((xsoftmax_reg_kernel<true, 4>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else if (seg <= 8192) hipLaunchKernelGGL((xsoftmax_reg_kernel<true, 8>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((xsoftmax_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, a);
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

extern "C" int lako_headbatch_nt(const lako_headbatch_t* p, lako_stream_t stream) {
  LAKO_CHECK_ARG(p && p->a && p->b && p->c, "lako_headbatch_nt: null operand");
  LAKO_CHECK_ARG(p->M > 0 && p->T > 0 && p->H > 0 && p->K > 0 && p->K % 32 == 0 && p->N > 0 && p->N % 16 == 0,
                 "lako_headbatch_nt: M, T, H > 0, K %% 32 == 0, N %% 16 == 0 (N=%d K=%d)", p->N, p->K);
  LAKO_CHECK_ARG((p->a_dtype == LAKO_BF16 || p->a_dtype == LAKO_F32) && p->b_dtype == LAKO_BF16, "lako_headbatch_nt: dtypes");
  LAKO_CHECK_ARG(p->a_sb % 8 == 0 && p->a_st % 8 == 0 && p->a_sh % 8 == 0 && p->b_sh % 8 == 0 && p->ldb % 8 == 0 &&
                     p->c_sb % 4 == 0 && p->c_st % 4 == 0 && p->c_sh % 4 == 0,
                 "lako_headbatch_nt: operand strides must keep 16-byte (A, B) / 8-byte (C) alignment");
  LAKO_CHECK_ALIGN(p->a, 16);
  LAKO_CHECK_ALIGN(p->b, 16);
  LAKO_CHECK_ALIGN(p->c, 8);
  HbArgs a{};
  a.A = (const char*)p->a; a.a_sb = p->a_sb; a.a_st = p->a_st; a.a_sh = p->a_sh;
  a.B = (const char*)p->b; a.b_sh = p->b_sh; a.ldb = p->ldb;
  a.C = (char*)p->c; a.c_sb = p->c_sb; a.c_st = p->c_st; a.c_sh = p->c_sh;
  a.M = p->M; a.T = p->T; a.N = p->N; a.K = p->K; a.H = p->H;
  a.n_slabs = p->n_slabs > 0 ? p->n_slabs : 1; a.slab_stride = p->slab_stride;
  LAKO_CHECK_ARG(a.n_slabs == 1 || (p->a_dtype == LAKO_F32 && p->slab_stride % 4 == 0), "lako_headbatch_nt: slabs need an fp32 A");
  const bool f32 = p->a_dtype == LAKO_F32;
  hipStream_t s = (hipStream_t)stream;
  if (p->N == 64 && p->K % 128 == 0 && (p->K == 512 || p->K == 768 || p->K == 1024)) {
    const dim3 grid((unsigned)((p->M + 15) / 16), (unsigned)p->H, 4);
#define HC_GO(F32, KSV) hipLaunchKernelGGL((hb_contract_kernel<F32, KSV>), grid, dim3(256), 0, s, a)
    if (p->K == 512) { if (f32) HC_GO(true, 4); else HC_GO(false, 4); }
    else if (p->K == 768) { if (f32) HC_GO(true, 6); else HC_GO(false, 6); }
    else { if (f32) HC_GO(true, 8); else HC_GO(false, 8); }
#undef HC_GO
    LAKO_LAUNCH_CHECK();
    return LAKO_OK;
  }
  const int nt = p->N % 64 == 0 && p->N > 64 ? 4 : 1;
  const int ku = nt == 4 ? (p->K % 64 == 0 ? 2 : 1) : (p->K % 256 == 0 ? 8 : 1);
  const int waves = ((p->M + 15) / 16) * (p->N / (16 * nt)) * p->H;
  const dim3 grid((unsigned)((waves + 3) / 4));
#define HB_GO(F32, NTV, KUV) hipLaunchKernelGGL((hb_nt_kernel<F32, NTV, KUV>), grid, dim3(256), 0, s, a)
  if (nt == 4) {
    if (ku == 2) { if (f32) HB_GO(true, 4, 2); else HB_GO(false, 4, 2); }
    else { if (f32) HB_GO(true, 4, 1); else HB_GO(false, 4, 1); }
  } else {
    if (ku == 8) { if (f32) HB_GO(true, 1, 8); else HB_GO(false, 1, 8); }
    else { if (f32) HB_GO(true, 1, 1); else HB_GO(false, 1, 1); }
  }
#undef HB_GO
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

static int hb_tn_args(const lako_headbatch_t* p, HbArgs& a);

extern "C" int lako_headbatch_tn(const lako_headbatch_t* p, lako_stream_t stream) {
  HbArgs a{};
  const int rc = hb_tn_args(p, a);
  if (rc != LAKO_OK) return rc;
  const dim3 grid((unsigned)(p->N / 64), (unsigned)p->H);
  if (p->b_dtype == LAKO_F32) hipLaunchKernelGGL((hb_tn_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((hb_tn_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, a);
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

extern "C" int lako_headbatch_tn_multi(const lako_headbatch_t* items, int n, lako_stream_t stream) {
  LAKO_CHECK_ARG(items && n >= 1 && n <= HB_MULTI_MAX, "lako_headbatch_tn_multi: 1..%d problems", HB_MULTI_MAX);
  HbMulti mm{};
  for (int i = 0; i < n; ++i) {
    const int rc = hb_tn_args(&items[i], mm.it[i]);
    if (rc != LAKO_OK) return rc;
    LAKO_CHECK_ARG(items[i].N == items[0].N && items[i].H == items[0].H && items[i].b_dtype == items[0].b_dtype,
                   "lako_headbatch_tn_multi: problem %d differs in N / H / b_dtype from problem 0", i);
  }
  const dim3 grid((unsigned)(items[0].N / 64), (unsigned)items[0].H, (unsigned)n);
  if (items[0].b_dtype == LAKO_F32) hipLaunchKernelGGL((hb_tn_multi_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, mm);
  else hipLaunchKernelGGL((hb_tn_multi_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, mm);
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

static int hb_tn_args(const lako_headbatch_t* p, HbArgs& a) {
  LAKO_CHECK_ARG(p && p->a && p->b && p->c, "lako_headbatch_tn: null operand");
  LAKO_CHECK_ARG(p->M > 0 && p->T > 0 && p->H > 0 && p->N > 0 && p->N % 64 == 0 && p->K == 64,
                 "lako_headbatch_tn: M, T, H > 0, N %% 64 == 0, K (rows of A per head) == 64");
  LAKO_CHECK_ARG(p->a_dtype == LAKO_BF16 && (p->b_dtype == LAKO_BF16 || p->b_dtype == LAKO_F32), "lako_headbatch_tn: dtypes");
  LAKO_CHECK_ARG(p->a_sb % 8 == 0 && p->a_st % 8 == 0 && p->a_sh % 8 == 0 && p->b_sb % 8 == 0 && p->b_st % 8 == 0 &&
                     p->b_sh % 8 == 0,
                 "lako_headbatch_tn: operand strides must keep 16-byte alignment");
  LAKO_CHECK_ALIGN(p->a, 16);
  LAKO_CHECK_ALIGN(p->b, 16);
  LAKO_CHECK_ALIGN(p->c, 4);
  a = HbArgs{};
  a.A = (const char*)p->a; a.a_sb = p->a_sb; a.a_st = p->a_st; a.a_sh = p->a_sh;
  a.B = (const char*)p->b; a.b_sb = p->b_sb; a.b_st = p->b_st; a.b_sh = p->b_sh;
  a.C = (char*)p->c; a.c_st = p->c_st; a.c_sh = p->c_sh;
  a.M = p->M; a.T = p->T; a.N = p->N; a.K = p->K; a.H = p->H;
  a.n_slabs = p->n_slabs > 0 ? p->n_slabs : 1; a.slab_stride = p->slab_stride;
  LAKO_CHECK_ARG(a.n_slabs == 1 || (p->b_dtype == LAKO_F32 && p->slab_stride % 4 == 0), "lako_headbatch_tn: slabs need an fp32 B");
  return LAKO_OK;
}

extern "C" int lako_xattn_decode(const void* q, int64_t q_sb, int64_t q_ld, const void* e, int64_t e_ld, const int32_t* k_off,
                                 float* part_ml, float* part_c, int R, int D, int B, int key_splits, lako_stream_t stream) {
  LAKO_CHECK_ARG(q && e && k_off && part_ml && part_c, "lako_xattn_decode: null operand");
  LAKO_CHECK_ARG(R >= 1 && R <= 16 && (D == 512 || D == 768 || D == 1024) && B >= 1 && key_splits >= 1 && key_splits <= 64 &&
                     e_ld % 8 == 0 && e_ld >= D && e_ld < (1 << 20) && q_ld % 8 == 0 && q_sb % 8 == 0,
                 "lako_xattn_decode: R <= 16, D in {512, 768, 1024}, 1 <= key_splits <= 64, 16-byte rows (R=%d D=%d)", R, D);
  LAKO_CHECK_ALIGN(q, 16);
  LAKO_CHECK_ALIGN(e, 16);
  LAKO_CHECK_ALIGN(part_c, 16);
  XdArgs a{};
  a.q = (const char*)q; a.q_sb = q_sb; a.q_ld = q_ld;
  a.e = (const char*)e; a.e_ld = e_ld; a.k_off = k_off;
  a.part_ml = part_ml; a.part_c = part_c;
  a.R = R; a.D = D; a.B = B; a.Z = key_splits;
  const int stage_b = 32 * D * 2, red_b = 8 * 32 * 16 * 4;
  const int lds = (3 * stage_b + red_b <= 160 * 1024 ? 3 : 2) * stage_b + red_b;
  const dim3 grid((unsigned)key_splits, (unsigned)B);
  hipStream_t s = (hipStream_t)stream;
#define XD_GO(KSV)                                                                                                     \
  do {                                                                                                                 \
    static lako_lds_cur_t done;                                                                                          \
    xset_lds(reinterpret_cast<const void*>(&xdecode_kernel<KSV>), lds, done);                                          \
    hipLaunchKernelGGL((xdecode_kernel<KSV>), grid, dim3(512), lds, s, a);                                             \
  } while (0)
  if (D == 512) XD_GO(2);
  else if (D == 768) XD_GO(3);
  else XD_GO(4);
#undef XD_GO
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}

extern "C" int lako_xattn_decode_combine(const float* part_ml, const float* part_c, const void* wv, int64_t ldw, void* ctx,
                                         int64_t ctx_ld, int H, int D, int B, int key_splits, lako_stream_t stream) {
  LAKO_CHECK_ARG(part_ml && part_c && wv && ctx, "lako_xattn_decode_combine: null operand");
  LAKO_CHECK_ARG(H >= 1 && H <= 16 && (D == 512 || D == 768 || D == 1024) && B >= 1 && key_splits >= 1 && key_splits <= 64 &&
                     ldw % 8 == 0,
                 "lako_xattn_decode_combine: H <= 16, D in {512, 768, 1024}, 1 <= key_splits <= 64");
  LAKO_CHECK_ALIGN(wv, 16);
  XcArgs a{};
  a.part_ml = part_ml; a.part_c = part_c; a.wv = (const char*)wv; a.ldw = ldw; a.ctx = (char*)ctx; a.ctx_ld = ctx_ld;
  a.D = D; a.B = B; a.Z = key_splits;
  const dim3 grid((unsigned)H, (unsigned)B);
  if (D == 512) hipLaunchKernelGGL((xdecode_combine_kernel<16>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else if (D == 768) hipLaunchKernelGGL((xdecode_combine_kernel<24>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((xdecode_combine_kernel<32>), grid, dim3(256), 0, (hipStream_t)stream, a);
  LAKO_LAUNCH_CHECK();
  return LAKO_OK;
}
