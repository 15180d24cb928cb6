/* lako_hip.h — C-ABI of liblako_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the LaKo
 * Fusion-in-Decoder reader hot path.
 *
 * The reference (hackerchenzhuo/LaKo) has NO native code and no FFI: its hot path is
 * `FiDT5.forward/generate` (src/model.py:39-60,227-234) dispatching stock torch ops from HuggingFace
 * `modeling_t5.py` ("HF5:line" below = transformers 5.15.0, the readable stand-in for the 3.0.2 the
 * reference names).  Each entry point here replaces one torch-op sequence of that path; the comment
 * on each function cites the reference/HF lines whose arithmetic it reproduces.  The Python host side
 * (lako_amd/ops.py) binds these with ctypes — see INTEGRATION.md.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer unless said otherwise
 *  - row-major contiguous tensors; activations [rows, features]; weights in HF layout [out, in]
 *  - dtype: LAKO_F32 or LAKO_BF16 storage; accumulation is always fp32
 *  - all work is enqueued on `stream` (a hipStream_t); no function allocates, frees or synchronises
 *  - return 0 on success, negative LAKO_E_* otherwise; lako_last_error() gives the message
 *  - pointers must be 16-byte aligned; feature dimensions multiples of 8
 */
#ifndef LAKO_HIP_H
#define LAKO_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 4): the struct and argument changes of round 3 (lako_gemm_nt_t.tuning, the tuning / alpha arguments of lako_gemm_tn*,
 * lako_fact_scores, no lako_set_tuning) are incompatible with callers built against version 1 — such a caller must be rebuilt.
 * lako_amd/_lib.py::load() refuses a library whose lako_version() is not the one it was written for. */
#define LAKO_ABI_VERSION 4

enum { LAKO_F32 = 0, LAKO_BF16 = 1, LAKO_FP8_E4M3 = 2 /* MX block-scaled operands of lako_gemm_nt_mx only */ };
enum { LAKO_OK = 0, LAKO_E_BADARG = -1, LAKO_E_ALIGN = -2, LAKO_E_LAUNCH = -3, LAKO_E_UNSUPPORTED = -4 };

typedef void* lako_stream_t; /* hipStream_t */

int lako_version(void);
/* copies the calling thread's last error message (NUL-terminated) into buf; returns its length */
int lako_last_error(char* buf, size_t n);

/* stateless dropout: the four consecutive elements 4q … 4q+3 of site `site` share h = hash32(key(seed, site) ^ q) and
 * w = h*0x9E3779B1; their 16-bit draws are h>>16, h&0xffff, w>>16, w&0xffff and an element is kept iff its draw >=
 * round(p*65536); kept elements are scaled by 1/(1-p) (torch.nn.Dropout semantics, HF5:725,745,83-94,400).  Forward and
 * backward kernels regenerate the mask from (seed, site); p == 0 disables. */
typedef struct {
  float p;
  uint32_t seed;
  uint32_t site;
} lako_dropout_t;

/* ---- kernel-selection knobs (A/B measurements, tests that pin one kernel variant; training uses the defaults) ---------------
 * The library holds NO mutable tuning state (re-entrant, thread-safe, no mutable globals): a caller that wants
 * anything but the defaults OWNS a lako_tuning_t and passes it to the GEMM entry points; NULL there means the process defaults —
 * built once from the LAKO_TUNING environment string and read-only afterwards.  Every knob selects between kernels / tilings that
 * compute the same result; the two `experiment` fields change results or cache policy and are honoured only by a build with
 * -DLAKO_EXPERIMENTS (lako_tuning_set refuses their keys in the release library with LAKO_E_UNSUPPORTED).
 *   lako_tuning_init(t): defaults, then the LAKO_TUNING environment string "key=value,key=value" (error: LAKO_E_BADARG / _UNSUPPORTED)
 *   lako_tuning_set(t, key, value): one field of the CALLER's struct by its key ("gemm_" + field name); unknown key: LAKO_E_BADARG */
typedef struct lako_tuning {
  int32_t nt_variant;    /* -1 heuristics (default); 0 = 128x128 tile / 4 waves, 1 = 256x128 / 8 waves, 2 = 256x256 / 8 waves,
                            4 = 128x128 4-slot ring, 5 = small tiles with K split over the waves (M <= 256),
                            6 = 256x256 / 4 waves with hipcc's schedule (measured slower, kept for A/B), 7 = 192x256 / 8 waves,
                            8 = 288x256 / 8 waves, 9 = 256x256 / 4 waves with the hand-placed K loop (round 6), 3 = the same, 192x256 */
  int32_t nt_tail_split; /* 1: rows beyond the full rounds of 256x256 tiles go to a second launch with small tiles */
  int32_t nt_ring;       /* 1: few-tile problems on the 4-slot ring kernel */
  int32_t nt_skinny;     /* 1: M <= 256 on the split-K kernel; 2 / 3 / 4 force 64² / 32² / 16² tiles */
  int32_t nt_side_lds;   /* 1: residual / aux operand of a 256x256 tile staged through LDS */
  int32_t nt_wide_epi;   /* 1: LDS-transposed epilogue with 16-byte stores for plain bf16 outputs */
  int32_t nt_group_m;    /* tile-rows per band of the banded tile order (default 8; 0 = row-major, < 0 forces |value|) */
  int32_t nt_persistent; /* 1: at most one resident workgroup set, each walking several tiles */
  int32_t nt_stagger;    /* 1: the second half of a workgroup's waves issues its LDS-DMA later in the K-step */
  int32_t nt_dephase;    /* start offset between neighbouring workgroups of a multi-round launch, 10-ns ticks (default 100, 0 = lockstep) */
  int32_t nt_dephase_n;  /* phases of that offset (default 2) */
  int32_t tn_big;        /* 1: 256x256 weight-gradient kernel */
  int32_t tn_split;      /* > 0 forces the K-split count of the 256x256 weight-gradient kernel */
  int32_t nt_debug;      /* experiment */
  int32_t nt_store_aux;  /* experiment */
  int32_t nt_tile192;    /* 1: 192-row tiles (variant 7) where the round count favours them (default 0: measured no faster) */
  int32_t nt_queue;      /* 1: the persistent 256x256 kernel pulls its tiles from per-XCD ticket counters (a launch that shares the chip
                            with RCCL kernels — LAKO_DP_MODE=overlap — ends when the tiles do, not when the last-started workgroup does) */
  int32_t nt_pp;         /* 0 (default): the two-phase main loop of rounds 1-3; 1 (A/B only, measured 3-11 % slower): the 8-phase loop of the
                            256x256 / 8-wave bf16 kernel (K >= 128): quadrant-wise fragment reads, half-tile LDS-DMA stagings retired by
                            counted waits, waves 4-7 one barrier behind waves 0-3.  Same results bit for bit */
  int32_t nt_glds;       /* 1 (default): whole K-slices are staged by global_load_lds (rows past the edge clamped) instead of
                            buffer_load ... lds; partial K-slices (K % 64) keep the buffer form and its zero fill */
  int32_t nt_tile288;    /* 1 (default): 288-row tiles (variant 8) where they save a round or the tail launch — plain epilogues only */
  int32_t nt_four;       /* 1 (default, round 6): problems of at least one round of 256x256 tiles with bf16 operands and output, K % 128 == 0 and
                            one of the epilogues {alpha, ReLU, dropout} / {residual, dropout} / {aux mask} run on the four-wave kernels with
                            the hand-placed K loop (256- or 192-row tiles, whichever needs less time for the launch's rounds of the chip);
                            0: the eight-wave kernels of rounds 1-5.  nt_variant 9 / 3 force the 256- / 192-row four-wave kernel.  Same
                            results bit for bit.  3 (= 1 | 2): as 1, but the dropout epilogues without a side operand go through the LDS
                            transposition like the others instead of leaving straight from the accumulator layout (A/B measurements).
                            lako_gemm_nt_mx follows the same key: non-zero = the four-wave MX kernel where K % 512 == 0 and the launch
                            fills the chip */
  int32_t tn_four;       /* 1 (default, round 6): the 256x256 weight-gradient kernel on four waves with the same hand-placed K loop
                            (K ranges of at least 128 rows, no slab reduction); 0: the eight-wave kernel.  Same results bit for bit up to
                            the order of the float atomics of K-splits */
  int32_t reserved[10];  /* zero */
} lako_tuning_t;
int lako_tuning_init(lako_tuning_t* t);
int lako_tuning_set(lako_tuning_t* t, const char* key, int value);

/* ---- GEMM  (K3/K6/K7/K10/K12 of SURVEY.md §2.3: every nn.Linear(bias=False) on the path,
 *      HF5:304,325-326,367 (q,k,v,o), HF5:83-94 (wi,wo), HF5:1047 (lm_head)) --------------------- */
enum {
  LAKO_EPI_RELU = 1,    /* v = max(v, 0)                                  (HF5:85 ReLU)             */
  LAKO_EPI_RESID = 2,   /* v = resid + dropout(v)                         (HF5:400,141 residual)    */
  LAKO_EPI_AUXMASK = 4, /* v = aux > 0 ? v * aux_scale : 0                (ReLU+dropout backward)   */
  LAKO_EPI_ATOMIC = 8,  /* C (fp32) += v with float atomics instead of a store                     */
  LAKO_EPI_NORM_A = 16  /* (round 4) A is the UN-normalised input of a T5LayerNorm (HF5:236-256): the operand is
                           bf16(norm_w * (A * rstd)), rstd = rsqrt(mean(A^2) + norm_eps), formed in the kernel's fragments exactly
                           as lako_rmsnorm_fwd would write it; optional side outputs norm_out (the normalised rows) and norm_rstd.
                           Only where the M <= 256 kernel runs (bf16, K % 32 == 0, K <= 1024): LAKO_E_UNSUPPORTED otherwise,
                           nothing launched — the caller then runs lako_rmsnorm_fwd + a plain product.                       */
};
typedef struct {
  const void* A; /* [M, K] row-major, lda */
  const void* B; /* [N, K] row-major, ldb  — C = A · Bᵀ ("NT": both operands K-contiguous)        */
  void* C;       /* [M, N] row-major, ldc */
  int64_t M, N, K, lda, ldb, ldc;
  int in_dtype, out_dtype;
  float alpha; /* v = alpha * acc first */
  int flags;   /* LAKO_EPI_* */
  const void* resid; /* [M, N] in out_dtype, ldr   (LAKO_EPI_RESID) */
  int64_t ldr;
  const void* aux; /* [M, N] in in_dtype, ldaux      (LAKO_EPI_AUXMASK) */
  int64_t ldaux;
  float aux_scale;
  lako_dropout_t drop; /* applied after relu / before the residual add; idx = m*N + n */
  const lako_tuning_t* tuning; /* HOST pointer, NULL = process defaults */
  /* LAKO_EPI_NORM_A (round 4; at the END of the struct: a caller that never sets the flag may leave them uninitialised) */
  const float* norm_w; /* [K] fp32 */
  float norm_eps;
  void* norm_out;      /* optional bf16 [M, K], row stride norm_ld elements */
  int64_t norm_ld;
  float* norm_rstd;    /* optional fp32 [M] */
} lako_gemm_nt_t;
int lako_gemm_nt(const lako_gemm_nt_t* p, lako_stream_t stream);

/* ---- MX block-scaled fp8 GEMM (BASELINE config 5 "fp8 MFMA GEMMs"; the same nn.Linear products as lako_gemm_nt) -----------
 * OCP microscaling: operands are e4m3 bytes, every 32 consecutive k of a row share one E8M0 scale byte s (element = q * 2^(s-127));
 * gfx950's v_mfma_scale_f32_16x16x128_f8f6f4 applies the scales inside the matrix core at twice the bf16 rate.
 * lako_mx_quantize: x bf16 [rows, K] (row stride ld) -> q [rows, K] e4m3 (contiguous) and scales [rows, 4, KSP] bytes with
 *   KSP = ceil(K/128) rounded up to a multiple of 4: the scale of block j (k = 32j .. 32j+31) of a row sits at [row][j % 4][j / 4]
 *   (the four K-steps a matrix-core lane needs next are one aligned dword).  s - 127 = floor(log2(max|x| of the block)) - 8,
 *   q = saturate_e4m3(x * 2^(127-s)).  K % 128 == 0.
 * lako_gemm_nt_mx: C [M,N] bf16 = epilogue(alpha * A.B^T), p->A / p->B the e4m3 matrices (in_dtype LAKO_FP8_E4M3, lda / ldb in
 *   bytes), a_scales / b_scales as written by lako_mx_quantize; epilogue flags RELU, RESID, AUXMASK and dropout as lako_gemm_nt. */
int lako_mx_quantize(const void* x, int64_t rows, int64_t K, int64_t ld, void* q, uint8_t* scales, lako_stream_t stream);
/* T5LayerNorm (lako_rmsnorm_fwd without dropout, bf16) with lako_mx_quantize of its output fused in (round 6, ABI 4): y bf16 [rows, d]
 * and rstd as lako_rmsnorm_fwd writes them, q [rows, d] e4m3 and scales [rows, 4, KSP(d)] as lako_mx_quantize(y) would — bit for bit —
 * in one pass over the rows (the fp8 forward products of a layer consume the normalised rows next).  d % 128 == 0. */
int lako_rmsnorm_fwd_mx(const void* x, const float* w, void* y, float* rstd, void* q, uint8_t* scales, int64_t rows, int d, float eps,
                        lako_stream_t stream);
int lako_gemm_nt_mx(const lako_gemm_nt_t* p, const uint8_t* a_scales, const uint8_t* b_scales, lako_stream_t stream);

/* C[M,N] (fp32) += alpha * Aᵀ·B with A [K, M], B [K, N] row-major (weight gradients dW = dYᵀ·X —
 * the autograd of every nn.Linear above).  Split-K over `split_k` workgroups, fp32 atomics. */
int lako_gemm_tn(const void* A, const void* B, float* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                 int64_t ldb, int64_t ldc, int in_dtype, float alpha, int split_k, const lako_tuning_t* tuning,
                 lako_stream_t stream);

/* Several weight gradients in ONE launch: C_i[M_i,N_i] += alpha_i * A_iᵀ·B_i for i < n_items (<= LAKO_TN_GROUP_MAX), all with
 * the same K (= tokens) and input dtype.  The four dW of a transformer layer fill the chip with ~3 K-splits instead of 7-28
 * each, and every split costs one fp32 atomic pass over the output (≈1.3 TB/s chip-wide).  Falls back to one lako_gemm_tn
 * per item for shapes the 256x256 kernel does not take (fp32 inputs, M or N < 256).  `items` is a HOST array.
 * split_k: 0 = choose; 1 = every output element has ONE contributor (into a zeroed C the result is then independent of the
 * order workgroups finish in); round 4, lako_gemm_tn too: -1 = one contributor AND the caller promises that nothing else adds to
 * these outputs while the launch runs: C += v by plain loads and stores (the chip's copy rate instead of ≈1.3 TB/s of float
 * atomics — the decoder's weight gradients, K = batch x answer length); -2 = the same, overwriting: C = v, no zeroed C needed (the
 * encoder-state gradient of the cross-attention); -3 (grouped only) = HYBRID under the same promise as -1: the first
 * 256 * floor(tiles / 256) output tiles — whole rounds of the chip — run their whole K in one workgroup each and add by plain loads
 * and stores, the remaining tiles are cut into up to 16 K-pieces that add by atomics (the encoder's weight gradients: 1 296 tiles =
 * five rounds of full-K tiles + 16 tiles x 16 pieces, instead of 15.2 -> 16 rounds of third-K units with one atomic pass each).
 * With split_k < 0 the items' output ranges must be disjoint: overlapping ones are refused with LAKO_E_BADARG before anything is launched. */
#define LAKO_TN_GROUP_MAX 48
typedef struct {
  const void* a; /* [K, M] row-major, lda */
  const void* b; /* [K, N] row-major, ldb */
  float* c;      /* [M, N] fp32, ldc */
  int64_t M, N, lda, ldb, ldc;
  float alpha;
  int32_t rows_out; /* 0 = M; else only the first rows_out (<= M) rows of C are touched — M must be a multiple of 8 for the operand
                       loads, a problem whose true row count is not (a sample's keys) says so here instead of spilling up to 7 rows
                       into its neighbour's output (round 4: needed once C is overwritten rather than added to) */
} lako_gemm_tn_item_t;
int lako_gemm_tn_grouped(const lako_gemm_tn_item_t* items, int n_items, int64_t K, int in_dtype, int split_k,
                         const lako_tuning_t* tuning, void* workspace, int64_t workspace_bytes, lako_stream_t stream);
/* workspace (round 4; optional — NULL keeps the float atomics): caller-owned scratch of at least
 * lako_gemm_tn_grouped_workspace(...) bytes for the SAME arguments (0 = this launch needs none).  With it the K-splits of a tile meet
 * through fp32 partial tiles and one arrival counter per tile: every split stores its partial, the split that arrives last sums them in
 * split order and adds the tile to C by plain loads / stores — no atomics, and the result no longer depends on the order workgroups
 * finish in.  The scratch may be reused by the next launch on the same stream; it holds nothing between launches. */
int64_t lako_gemm_tn_grouped_workspace(const lako_gemm_tn_item_t* items, int n_items, int64_t K, int in_dtype, int split_k,
                                       const lako_tuning_t* tuning);

/* ---- T5LayerNorm (RMSNorm, HF5:59-72): y = dropout(x * rsqrt(mean(x²) + eps) * w) ----------------
 * rstd [rows] fp32 is written for the backward.  w is fp32 [d]. */
int lako_rmsnorm_fwd(const void* x, const float* w, void* y, float* rstd, int64_t rows, int d, float eps,
                     int dtype, lako_dropout_t drop, lako_stream_t stream);
/* dx = (dres ? dres : 0) + rmsnorm_bwd(dropout_bwd(dy)); dw (fp32 [d]) += Σ_rows dy·x·rstd (atomics).
 * dx_drop (may be NULL): also written, = lako_dropout_apply(dx, drop_out) — the gradient entering the next residual branch
 * through ITS output dropout, produced in the same pass instead of a second read of dx */
int lako_rmsnorm_bwd(const void* dy, const void* x, const float* w, const float* rstd, const void* dres,
                     void* dx, float* dw, int64_t rows, int d, int dtype, lako_dropout_t drop, void* dx_drop,
                     lako_dropout_t drop_out, lako_stream_t stream);

/* ---- embedding (HF5:678 embed_tokens + HF5:725 dropout) --------------------------------------------
 * out[t] = dropout(table[ids[t]]);  bwd: dtable (fp32 [vocab, d]) += scatter of dropout_bwd(dout) */
int lako_embed_fwd(const int64_t* ids, const void* table, void* out, int64_t n_tok, int d, int64_t vocab,
                   int dtype, lako_dropout_t drop, lako_stream_t stream);
int lako_embed_bwd(const int64_t* ids, const void* dout, float* dtable, int64_t n_tok, int d, int64_t vocab,
                   int dtype, lako_dropout_t drop, lako_stream_t stream);
/* The same sum added in a fixed order (LAKO_DETERMINISTIC=1): `perm` [n_tok] sorts the token positions by id, equal
 * ids in position order (a stable argsort: index plumbing of the caller); each row of dtable is written once, by one wave. */
int lako_embed_bwd_ordered(const int64_t* ids, const int64_t* perm, const void* dout, float* dtable, int64_t n_tok, int d,
                           int64_t vocab, int dtype, lako_dropout_t drop, lako_stream_t stream);
/* 1 when the library runs in the bit-reproducible mode (environment LAKO_DETERMINISTIC=1, read once per process): every float
 * accumulation shared between workgroups is then order-independent (fixed-point integer atomics, one-contributor splits). */
int lako_deterministic(void);

/* ---- relative position bias (HF5:217-279): the bias depends only on (key_pos - query_pos), so the
 * [nb, H] embedding is expanded once per stack into rel[H, R] with rel[h][r] = table[lut[r]][h];
 * the attention kernels index it as r = j - i + rel_off.  lut (int32 [R]) holds
 * _relative_position_bucket(r - (R-1)/2 …) computed on the host in float32 exactly as HF5:217-262. */
int lako_relpos_expand(const float* table, const int32_t* lut, float* rel, int H, int R, int nb,
                       lako_stream_t stream);
/* dtable[lut[r]][h] += drel[h][r] */
int lako_relpos_reduce(const float* drel, const int32_t* lut, float* dtable, int H, int R, int nb,
                       lako_stream_t stream);

/* ---- attention (HF5:144-173,281-369; the same math as the reference's own
 * cross_attention_forward, src/model.py:286-349): P = softmax_fp32(Q·Kᵀ + bias + mask) — scores are
 * NOT scaled by 1/sqrt(dk) — then dropout(P)·V.  One call covers every (batch row, head).
 * Used for encoder self-attention (bidirectional bias + key padding mask), decoder self-attention
 * (causal + unidirectional bias) and cross-attention over the concatenated n_passages·L encoder
 * states (key padding mask only).  bf16 / d_head 64 / <= 256 keys / no key mask / no causal mask / no score capture (the ragged
 * encoder self-attention of training) takes the fast path of csrc/attn_enc.hip; everything else the generic kernels. */
typedef struct {
  const void *q, *k, *v; /* element (b, t, h, c) at ptr + b*stride_b + t*stride_t + h*d_head + c */
  void* out;             /* same addressing with o_stride_* */
  float* lse;            /* [Bn, H, Lq, 4] fp32 softmax row statistics saved for backward: (row max, 1/row sum,
                            scratch for delta = rowsum(dO∘O) written by lako_attn_bwd, unused) */
  int64_t q_stride_b, q_stride_t, k_stride_b, k_stride_t, v_stride_b, v_stride_t, o_stride_b, o_stride_t;
  const float* rel_bias; /* [H, R] fp32 or NULL */
  int R, rel_off;        /* bias index = clamp(j - i + rel_off, 0, R - 1): a distance outside the table takes the nearest entry
                            (and, in lako_attn_bwd, adds its gradient there) — a table whose far entries are all one bucket may
                            be cut short: the decoder-side bias of transformers 3.0.2's cross-attention (keys at or after the query:
                            bucket 0) is [H, 2T - 1] for any number of keys */
  const uint8_t* key_mask; /* [Bn, Lk] 1 = attend, 0 = padding; or NULL */
  int causal, causal_off;  /* causal: key j visible iff j <= i + causal_off */
  int Bn, H, Lq, Lk, d_head;
  int dtype;
  lako_dropout_t drop; /* on the probabilities.  Block recipe (csrc/attn_shared.h; mirrored in tests/ref_ops.py): the 4x4 block
                          (queries 4a.., keys 4c..) of head-row bh = b*H+h shares hb = mix(((bh*ceil(Lq/4) + a)*ceil(Lk/4) + c) ^ key,
                          0x5BD1E9, 13); element (q, k) draws 16 bits of W = mix(hb, M[q&3][(k&3)>>1], 6 + 2*(q&3) + 5*((k&3)>>1)):
                          the high half for even k, the low half for odd k; keep iff draw >= round(p*65536);
                          mix(x, c, s) = t ^ (t >> 16), t = (x mod 2^24)*c + (x >> s) mod 2^32 (full-rate instructions only) */
  float* scores_out;   /* optional [Bn, H, Lq, Lk] fp32 raw pre-softmax scores (+bias, masked keys = 0):
                          the quantity src/model.py:316-329 stores for get_crossattention_scores */
  /* RAGGED sequences (optional; NULL = the padded layout above).  q_off / k_off: device int32 [Bn + 1] row offsets into ONE
   * packed [rows, H*d_head] buffer: sequence b owns rows [off[b], off[b+1]) (addressing ptr + (off[b] + t)*stride_t + …, the
   * batch stride is ignored); `out` follows q_off.  Lq / Lk are then the MAXIMUM lengths: the statistics buffer and the dropout
   * draws stay indexed in padded coordinates, so a ragged call reproduces the padded call on the rows that exist.  Keys
   * carry no padding in this form (key_mask must be NULL); score capture is not available. */
  const int32_t* q_off;
  const int32_t* k_off;
  /* optional (round 4): device int32 [Bn], a PERMUTATION of 0 … Bn-1 — the order in which workgroups take the sequences.  Results do
   * not depend on it (each sequence is computed as before, into its own rows); it exists for load balance: sequences sorted by
   * descending length start the long ones first and give every persistent workgroup (which walks positions slot, slot + n, …)
   * one sequence of every length class.  Honoured by the encoder fast path (bf16, d_head 64, <= 256 keys), ignored elsewhere. */
  const int32_t* order;
} lako_attn_fwd_t;
int lako_attn_fwd(const lako_attn_fwd_t* p, lako_stream_t stream);

typedef struct {
  const void *q, *k, *v, *out, *dout;
  float* lse; /* the forward's [Bn, H, Lq, 4] statistics; slot 2 is overwritten with delta = rowsum(dO∘O) */
  void *dq_out, *dk_out, *dv_out; /* same addressing as q / k / v (dK, dV use the k / v strides) */
  int64_t q_stride_b, q_stride_t, k_stride_b, k_stride_t, v_stride_b, v_stride_t, o_stride_b, o_stride_t;
  const float* rel_bias;
  float* drel; /* [H, R] fp32, += (atomics); or NULL */
  int R, rel_off;
  const uint8_t* key_mask;
  int causal, causal_off;
  int Bn, H, Lq, Lk, d_head;
  int dtype;
  lako_dropout_t drop;
  const int32_t* q_off; /* ragged sequences, as in lako_attn_fwd_t (dq follows q_off; dk, dv follow k_off) */
  const int32_t* k_off;
  const int32_t* order; /* as in lako_attn_fwd_t */
} lako_attn_bwd_t;
int lako_attn_bwd(const lako_attn_bwd_t* p, lako_stream_t stream);

/* ---- cross-attention in the encoder-state space (the decoder's EncDecAttention over the concatenated n_passages·L encoder
 * states, src/model.py:286-349 / HF5:281-369; bf16, d_kv 64, ragged keys) --------------------------------------------------
 * Same result as projecting every encoder state to K and V and calling lako_attn_fwd, re-associated so that the per-head
 * projections act on the R = T·H query rows of a sample instead of on its keys (csrc/xattn.hip):
 *   Q'[b][t·H + h][:] = q[b,t,h,:]·Wk_h          lako_headbatch_nt   ("expand": [B·T, 64] x [64, D] per head)
 *   S  = Q'·E_bᵀ                                 lako_xattn_scores   (raw scores, no 1/sqrt(d) as in T5)
 *   P  = dropout(softmax_keys(S))                lako_xattn_softmax_fwd
 *   C' = P·E_b                                   lako_xattn_context  (fp32; the key ranges of a sample write separate slabs)
 *   ctx[b,t,h,:] = C'[b][t·H + h][:]·Wv_hᵀ       lako_headbatch_nt   ("contract": [B·T, D] x [D, 64] per head)
 * backward: dC' = expand(dctx, Wv), dP = lako_xattn_scores(dC'), dS = lako_xattn_softmax_bwd, dQ' = lako_xattn_context(dS),
 * dq = contract(dQ', Wk), dWk_h += q_hᵀ·dQ'_h and dWv_h += dctx_hᵀ·C'_h (lako_headbatch_tn), and after the last layer
 * dE_b = Σ_layers [P; dS]ᵀ·[dC'; Q'] (lako_gemm_tn_grouped, one item per sample).
 * Key layout: E is the packed [rows, D] encoder output, sample b owns rows [k_off[b], k_off[b+1]); score / probability matrices
 * are [R, p_total] with sample b's keys in columns [p_off[b], p_off[b] + n_keys(b)), every p_off[b] a multiple of 256 and the
 * columns up to p_off[b+1] zero padding (written by these functions).  k_off / p_off: device int32 [B + 1], B <= 256. */
int lako_xattn_scores(const void* q, int64_t q_sb, int64_t q_ld, /* bf16 rows: sample b, row r at q + b*q_sb + r*q_ld (elements) */
                      const void* e, int64_t e_ld, const int32_t* k_off, const int32_t* p_off, int64_t p_total,
                      float* s, int64_t s_ld, /* fp32 [R, s_ld] */
                      int R, int D, int B, lako_stream_t stream);
/* p bf16 [R, p_ld] (layout as s); out fp32: key range z (< key_splits) of sample b, row r at out + z*out_zs + b*out_sb + r*out_ld.
 * The keys of a sample are split over key_splits workgroups, each WRITES its own slab (no atomics: the result does not depend
 * on the order workgroups finish in); the consumer adds the slabs in order (lako_headbatch_t.n_slabs). */
int lako_xattn_context(const void* p, int64_t p_ld, const void* e, int64_t e_ld, const int32_t* k_off, const int32_t* p_off,
                       float* out, int64_t out_zs, int64_t out_sb, int64_t out_ld, int R, int D, int B, int key_splits,
                       lako_stream_t stream);
/* row r = t*H + h of sample b: stats[b][r] = (max, 1 / sum exp); p = dropout(softmax) as bf16, zeros in the padding columns.
 * Dropout: the attention block recipe of lako_attn_fwd with bh = b*H + h, q = t, k = key index in the sample, Lq = T,
 * Lk = max_keys (so that both formulations of the cross-attention drop the same probabilities); max_keys must be at least
 * every sample's key count (rows of up to 8 192 keys are processed from registers). */
int lako_xattn_softmax_fwd(const float* s, int64_t s_ld, float* stats, void* p, int64_t p_ld, const int32_t* k_off,
                           const int32_t* p_off, int B, int T, int H, int max_keys, lako_dropout_t drop, lako_stream_t stream);
/* ds = P∘(dropout_bwd(dp) − Σ_keys dropout(P)·dp) as bf16 (P recomputed from s and stats) */
int lako_xattn_softmax_bwd(const float* s, const float* dp, int64_t s_ld, const float* stats, void* ds, int64_t p_ld,
                           const int32_t* k_off, const int32_t* p_off, int B, int T, int H, int max_keys, lako_dropout_t drop,
                           lako_stream_t stream);

/* One decode step (R = H <= 16 expanded query rows per sample, D in {512, 768, 1024}) in one pass over the encoder states:
 * key range z (< key_splits) of sample b -> part_ml[z][b][r] = (max, sum exp), part_c[z][b][r][:] = sum exp(s - max) * E[key][:]
 * (fp32, 16 rows per (z, b)); lako_xattn_decode_combine merges the ranges and applies the V projection of head h = r:
 * ctx[b][h*64 + j] = softmax-weighted mean of E . Wv[h*64 + j][:]  (bf16, row stride ctx_ld). */
int lako_xattn_decode(const void* q, int64_t q_sb, int64_t q_ld, const void* e, int64_t e_ld, const int32_t* k_off,
                      float* part_ml, float* part_c, int R, int D, int B, int key_splits, lako_stream_t stream);
int lako_xattn_decode_combine(const float* part_ml, const float* part_c, const void* wv, int64_t ldw, void* ctx, int64_t ctx_ld,
                              int H, int D, int B, int key_splits, lako_stream_t stream);

/* Head-batched small products over the M = B·T decoder rows.  A row m = (b, t) of head h starts at element offset
 * b*sb + t*st + h*sh of its buffer (two-level rows: the rows of a sample may sit inside a per-sample block of a bigger buffer).
 *   lako_headbatch_nt:  C_h[m, n] = Σ_k A_h[m, k]·B_h[n, k];  A bf16 or fp32 (a_dtype), B_h[n, k] bf16 at b + h*b_sh + n*ldb + k,
 *                       C bf16 rows addressed like A's.  K % 32 == 0, N % 16 == 0.
 *   lako_headbatch_tn:  C_h[j, c] += Σ_m A[m][h*a_sh + j]·B_h[m][c], j < 64 (K), c < N;  A bf16, B rows bf16 or fp32 (b_dtype)
 *                       addressed like A's, C fp32 at c + h*c_sh + j*c_st + c (one atomic add per element and launch). */
typedef struct {
  const void *a, *b;
  void* c;
  int64_t a_sb, a_st, a_sh;
  int64_t b_sb, b_st, b_sh, ldb;
  int64_t c_sb, c_st, c_sh;
  int M, T, H, N, K;
  int a_dtype, b_dtype;
  int n_slabs;          /* 0 / 1: plain.  > 1: the fp32 operand (nt: A, tn: B) is the sum of n_slabs buffers */
  int64_t slab_stride;  /* slab_stride elements apart, added in slab order */
} lako_headbatch_t;
int lako_headbatch_nt(const lako_headbatch_t* p, lako_stream_t stream);
int lako_headbatch_tn(const lako_headbatch_t* p, lako_stream_t stream);
/* (round 4) n <= 24 lako_headbatch_tn problems of one N, H and b_dtype in ONE launch — the Wk / Wv gradients of all decoder layers,
 * deferred to the end of the decoder's backward (each alone is nothing but its dispatch). */
int lako_headbatch_tn_multi(const lako_headbatch_t* items, int n, lako_stream_t stream);

/* ---- LM-head loss (HF5:1051-1054): CrossEntropyLoss(ignore_index=-100), mean over valid labels ----
 * logits fp32 [M, V]; loss_out[0] = mean loss, loss_out[1] = number of valid labels;
 * dlogits (dtype, [M, V], optional) = upstream * (softmax - onehot) / n_valid, 0 on ignored rows;
 * upstream = device pointer to the scalar d(objective)/d(loss) (NULL = 1). */
int lako_ce_fwd_bwd(const float* logits, const int64_t* labels, float* loss_out, void* dlogits,
                    const float* upstream, int64_t M, int64_t V, int dtype, lako_stream_t stream);

/* ---- optimizer (train_reader.py:76-79; src/util.py:185-227): global-norm clip + HF AdamW with
 * correct_bias=False + decoupled weight decay, one fused pass over the flat parameter buffer ------ */
/* out[0] += Σ g²  (caller zeroes out[0]) */
int lako_sumsq(const float* g, int64_t n, float* out, lako_stream_t stream);
/* coef = min(1, max_norm / (sqrt(gnorm_sq[0]) * grad_scale + 1e-6)) as torch clip_grad_norm_;
 * g' = g * grad_scale * coef;  m,v,p updated as A.6 of SURVEY.md;  shadow (dtype) = p if not NULL. */
int lako_adamw_step(float* p, const float* g, float* m, float* v, void* shadow, int64_t n, float lr, float beta1,
                    float beta2, float eps, float weight_decay, const float* gnorm_sq, float max_norm,
                    float grad_scale, int shadow_dtype, lako_stream_t stream);
/* dst[c][r] = (dtype) src[r][c]  — transposed low-precision weight copies used by the dX GEMMs */
int lako_transpose_cast(const float* src, void* dst, int64_t rows, int64_t cols, int dtype, lako_stream_t stream);
/* the same for a whole table of matrices in one launch (all weights after an optimizer step): matrix i is
 * src_base + desc[4i] ([rows = desc[4i+2]][cols = desc[4i+3]], src_dtype) → dst_base + desc[4i+1] ([cols][rows], dtype);
 * tile_prefix[i] = number of 64×64 tiles of matrices 0..i-1, total_tiles = their sum over all n matrices.
 * desc / tile_prefix are DEVICE arrays; offsets in elements, multiples of 4.  src_dtype LAKO_F32 (the master weights) or, for bf16
 * copies, LAKO_BF16 (the bf16 shadow the optimizer step has just written: the same values from half the bytes — round 4). */
int lako_transpose_cast_batched(const void* src_base, int src_dtype, void* dst_base, const int64_t* desc, const int32_t* tile_prefix,
                                int n, int total_tiles, int dtype, lako_stream_t stream);
int lako_cast(const float* src, void* dst, int64_t n, int dtype, lako_stream_t stream);

/* ---- small helpers ------------------------------------------------------------------------------ */
/* y = dropout_scale_mask(x) (used for the backward of residual-branch dropout, HF5:400,141) */
int lako_dropout_apply(const void* x, void* y, int64_t n, int dtype, lako_dropout_t drop, lako_stream_t stream);
/* _shift_right (HF5:618-637): dec[b][0] = 0; dec[b][t] = labels[b][t-1] (−100 → 0) */
int lako_shift_right(const int64_t* labels, int64_t* dec_ids, int B, int T, lako_stream_t stream);
/* unpadded encoder input (round 4): the token ids of the valid positions of ids [BN, L], packed in (passage, position) order —
 * out[off[j] + t] = ids[j*L + t] for t < off[j+1] - off[j]; off: device int32 [BN + 1] (the row offsets the attention kernels use). */
int lako_pack_ids(const int64_t* ids, const int32_t* off, int64_t* out, int BN, int L, lako_stream_t stream);
/* greedy step (HF generate, num_beams=1, do_sample=False): next = argmax(logits[b]); rows already
 * done emit pad(0); done |= next == eos.  seq[b*seq_ld + pos] = next; next_ids[b] = next.
 * n_done[0] = number of finished rows after the step. */
int lako_greedy_step(const float* logits, int64_t V, int B, int64_t* seq, int64_t seq_ld, int pos, int64_t* next_ids,
                     uint8_t* done, int32_t* n_done, int64_t eos_id, int64_t pad_id, lako_stream_t stream);

/* ---- per-fact aggregation of the captured step-0 cross-attention scores (src/model.py:100-115,143-204,
 * FiDT5.get_crossattention_scores; replaces its per-sample Python loops) ----------------------------------------------------
 * scores fp32 [B, H, n_layers, N·L] (raw pre-softmax scores as lako_attn_fwd's scores_out lays them out per layer),
 * mask uint8 [B, N, L], ids int64 [B, N, L], out fp64 [B, n_context].  Per sample: vals[l] = Σ over heads and the layers
 * [layer0, layer0 + layers_used) of the scores of passage `passage` (masked positions 0); the passage is cut into spans that end
 * at token id 5 in the token ids of passage `ids_passage` (the reference takes the scores of passage ceil(N/2) and the ids of
 * passage 1, src/model.py:164-174: the same passage for N = 2), starting at index 2; span j -> style 0 mean | 1 max | 2 "21mean" (mean of the larger half); an unterminated
 * last span counts when the passage's last token is not pad(0); missing spans are -5; all divided by layers_used * H. */
int lako_fact_scores(const float* scores, const uint8_t* mask, const int64_t* ids, double* out, int B, int H, int n_layers,
                     int layer0, int layers_used, int N, int L, int passage, int ids_passage, int n_context, int style, lako_stream_t stream);

/* ---- exact inner-product search (src/index.py:19-50 faiss.IndexFlatIP.search; the scores are one
 * lako_gemm_nt of the queries against the stored embeddings) -----------------------------------------------------------
 * out_vals / out_idx [rows, k]: the k largest entries of each row of scores [rows, n] (row stride ld), descending; equal
 * scores in ascending index order.  1 <= k <= min(n, 1024). */
int lako_topk(const float* scores, int64_t rows, int64_t n, int64_t ld, int k, float* out_vals, int64_t* out_idx,
              lako_stream_t stream);

/* ---- product-quantised index: faiss.IndexPQ(d, M, nbits, METRIC_INNER_PRODUCT) behind src/index.py:21-23 (Indexer with
 * n_subquantizers > 0; round 3).  A vector is M sub-vectors of dsub floats; centroids [M][ksub][dsub] fp32, ksub = 2^nbits <= 256.
 * lako_pq_assign: per (vector, m) the nearest centroid in L2, lowest index on ties -> codes [n][M] (may be NULL); with sums
 *   [M][ksub][dsub] / counts [M][ksub] (both or neither; caller zeroes them) the sub-vector is also added to its centroid's running
 *   sum (the k-means update), and *err (may be NULL; caller zeroes) receives the summed squared distances.  x [n][ldx], ldx >= M*dsub.
 * lako_pq_lut:  lut [nq][M][ksub] = <q_m, centroid[m][c]>.      q [nq][ldq].
 * lako_pq_scan: scores[q][i] = sum over m (ascending, fp32) of lut[q][m][codes[i][m]]; scores [nq][ld], ld >= n; top-k: lako_topk. */
int lako_pq_assign(const float* x, int64_t n, int64_t ldx, const float* centroids, int M, int ksub, int dsub, uint8_t* codes,
                   float* sums, int32_t* counts, float* err, lako_stream_t stream);
int lako_pq_lut(const float* q, int64_t nq, int64_t ldq, const float* centroids, int M, int ksub, int dsub, float* lut,
                lako_stream_t stream);
int lako_pq_scan(const float* lut, const uint8_t* codes, int64_t n, int64_t nq, int M, int ksub, float* scores, int64_t ld,
                 lako_stream_t stream);

/* ---- retriever bi-encoder, forward (the backward follows below; SURVEY.md §8 f4: src/model.py:375-483 over HF BertModel); the matrix products and
 * the attention are lako_gemm_nt / lako_attn_fwd --------------------------------------------------------------------------
 * y = LayerNorm(x + lin_bias + resid)·gamma + beta (torch.nn.LayerNorm); lin_bias (fp32 [d]) and resid may be NULL */
int lako_layernorm_fwd(const void* x, const float* lin_bias, const void* resid, const float* gamma, const float* beta, void* y,
                       int64_t rows, int d, float eps, int dtype, lako_stream_t stream);
/* BertEmbeddings: out[t] = LayerNorm(word[ids[t]] + pos[t mod L] + type0) (token type 0, absolute positions); tables fp32 */
int lako_bert_embed(const int64_t* ids, const float* word, const float* pos, const float* type0, const float* gamma,
                    const float* beta, void* out, int64_t n_tok, int L, int d, int64_t vocab, float eps, int dtype,
                    lako_stream_t stream);
/* y = act(x + bias) over [rows, n]: act 0 identity, 1 exact (erf) GELU */
int lako_bias_act(const void* x, const float* bias, void* y, int64_t rows, int n, int act, int dtype, lako_stream_t stream);
/* out[b] (fp32 [d]) = mean of the L rows of x[b] ([B, L, d]); with mask ([B, L] uint8) only rows with mask != 0
 * (src/model.py:471-478) */
int lako_seq_mean(const void* x, const uint8_t* mask, float* out, int B, int L, int d, int dtype, lako_stream_t stream);
/* score[b][i] = scale * <q[b], p[b][i]> (fp32; src/model.py:443-448 einsum('bd,bid->bi') / sqrt(d)) */
int lako_bi_score(const float* q, const float* p, float* out, int B, int n, int d, float scale, lako_stream_t stream);

/* loss[0] = torch.nn.KLDivLoss()(log_softmax(score, -1), gold) over fp32 [B, n] (src/model.py:480-483); value only */
int lako_kldiv_fwd(const float* score, const float* gold, float* loss, int B, int n, lako_stream_t stream);

/* ---- retriever TRAINING (the autograd of src/model.py:413-483 under train_retriever.py:37-71 — KL distillation
 * of the reader's per-fact cross-attention scores into the bi-encoder).  The matrix products of the backward are lako_gemm_nt
 * (dX, with its residual epilogue for the skip connections) and lako_gemm_tn (dW), the attention backward lako_attn_bwd; these are
 * the row-wise pieces between them (csrc/bertbwd.hip).  Parameter gradients (fp32) are ACCUMULATED (+=).
 * lako_layernorm_bwd: backward of lako_layernorm_fwd, y = LayerNorm(z)*gamma + beta with z = x + lin_bias + resid (z and its
 *   statistics are recomputed from the forward's inputs): dz [rows, d] = the gradient of x and of resid; dgamma += sum_rows dy*zhat,
 *   dbeta += sum_rows dy, dbias (may be NULL) += sum_rows dz.
 * lako_bias_act_bwd: backward of y = act(x + bias) on a [rows, n] column block with row stride ld: dx (may be NULL, may alias dy)
 *   = dy * act'(x + bias), dbias += dbias_scale * sum_rows dx.  act 0 identity (x unused), 1 exact GELU.
 * lako_seq_mean_bwd: dx[b, l, :] = dout[b, :] / count_b on the rows that took part in lako_seq_mean, 0 elsewhere.
 * lako_bi_score_bwd: dp[b][i] = scale*dscore[b][i]*q[b], dq[b] = scale*sum_i dscore[b][i]*p[b][i] (overwritten).
 * lako_kldiv_bwd: dscore = upstream/(B*n) * (softmax(score)*sum_j gold - gold)  (upstream: device scalar or NULL = 1).
 * lako_bert_embed_bwd: backward of lako_bert_embed: dword / dpos rows by float atomics, dtype0, dgamma, dbeta column sums. */
int lako_layernorm_bwd(const void* dy, const void* x, const float* lin_bias, const void* resid, const float* gamma, void* dz,
                       float* dgamma, float* dbeta, float* dbias, int64_t rows, int d, float eps, int dtype, lako_stream_t stream);
int lako_bias_act_bwd(const void* dy, const void* x, const float* bias, void* dx, float* dbias, int64_t rows, int n, int64_t ld,
                      int act, float dbias_scale, int dtype, lako_stream_t stream);
int lako_seq_mean_bwd(const float* dout, const uint8_t* mask, void* dx, int B, int L, int d, int dtype, lako_stream_t stream);
int lako_bi_score_bwd(const float* dscore, const float* q, const float* p, float* dq, float* dp, int B, int n, int d, float scale,
                      lako_stream_t stream);
int lako_kldiv_bwd(const float* score, const float* gold, float* dscore, const float* upstream, int B, int n, lako_stream_t stream);
int lako_bert_embed_bwd(const int64_t* ids, const float* word, const float* pos, const float* type0, const float* gamma,
                        const void* dy, float* dword, float* dpos, float* dtype0, float* dgamma, float* dbeta, int64_t n_tok, int L,
                        int d, int64_t vocab, float eps, int dtype, lako_stream_t stream);


/* ---- data-parallel communication (round 5; SURVEY.md §8 b2 / e): the gradient all-reduce of the data-parallel reader over RCCL / xGMI, for
 * hosts that do not go through torch.distributed (which lako_amd/dist.py uses — backend "nccl" is the same RCCL).  Replaces what
 * /root/reference gets from torch's process group: src/slurm.py:157-160 (init_process_group), src/util.py:248-275 (the reductions).
 * One process per GPU; rank 0 makes the rendezvous id and hands its 128 bytes to the other ranks out of band (a file, a socket, MPI);
 * every rank then calls lako_comm_init on ITS device (hipSetDevice first).  The communicator is a caller-owned handle: no process state.
 * lako_allreduce is an in-place SUM on the caller's stream (asynchronous like every entry point here; order it against the kernels that
 * produce / consume the buffer by using the same stream or events).  Divide by the world size in the consumer: lako_adamw_step's grad_scale.
 * RCCL is looked up when first needed (the library already in the process, else the system's librccl.so.1); without it these return
 * LAKO_E_UNSUPPORTED and everything else in this header still works. */
#define LAKO_COMM_ID_BYTES 128
typedef struct lako_comm lako_comm_t;
int lako_comm_unique_id(uint8_t id[LAKO_COMM_ID_BYTES]);
int lako_comm_init(lako_comm_t** comm, int rank, int world, const uint8_t id[LAKO_COMM_ID_BYTES]);
int lako_comm_world_size(const lako_comm_t* comm);   /* as RCCL reports it; negative on error */
int lako_allreduce(lako_comm_t* comm, void* buf, int64_t count, int dtype /* LAKO_F32 | LAKO_BF16 */, lako_stream_t stream);
int lako_comm_destroy(lako_comm_t* comm);

/* ---- caller scratch: bytes of workspace an entry point wants for the given arguments; 0 = none.  Every kernel works in
 * the caller's buffers; the one optional scratch is lako_gemm_tn_grouped's slab reduction (its own query above, forwarded here). */
#define LAKO_WS_GEMM_TN_GROUPED 1
typedef struct {
  const lako_gemm_tn_item_t* items;
  int n_items;
  int64_t K;
  int in_dtype, split_k;
  const lako_tuning_t* tuning;
} lako_ws_gemm_tn_grouped_t;
int64_t lako_workspace_bytes(int op, const void* args);

#ifdef __cplusplus
}
#endif
#endif /* LAKO_HIP_H */
