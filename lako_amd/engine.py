"""Host-side engine of the FiD reader: parameter layout in HBM, workspaces, and the hand-orchestrated
forward / backward / greedy-decode schedules over the op vocabulary of `lako_amd.ops.HipOps`.

What it reproduces (reference file:line; HF5 = transformers 5.15.0 modeling_t5.py):
  * FiDT5.forward + EncoderWrapper.forward reshapes            src/model.py:39-51, 227-234
  * T5Stack / T5Block / attention / FFN / final norm            HF5:663-750, 448-509, 281-369, 83-94, 59-72
  * shift_right, tied LM head × d_model^-0.5, CE(ignore=-100)   HF5:618-637, 1044-1054
  * autograd of all of the above (train_reader.py:73)          written out by hand below
  * greedy generate with KV cache                              src/model.py:54-60, train_reader.py:142-146

Memory plan (sized for 288 GB HBM3E — nothing is recomputed unless `use_checkpoint`):
  P   fp32 master parameters, one flat buffer               G   fp32 gradients, same layout
  W   compute-dtype shadow of P (aliases P in fp32 mode)      WT  per-matrix transposed shadows ([in,out])
so that every GEMM of the step is either NT (X·Wᵀ forward, dY·(Wᵀ)ᵀ for dX) or TN (dYᵀ·X for dW).
q,k,v weights of a layer are adjacent rows of one [3·inner, d] block (one fused QKV GEMM), and the
cross-attention k,v weights of ALL decoder layers form one [L_dec·2·inner, d] block, so the encoder
output [B·N·L, d] is read once for all layers' K/V projections.
Gradient buffer order = the order gradients complete in backward (decoder, cross-K/V, encoder layers
last→first, embedding), so data-parallel buckets can be all-reduced while backward is still running.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np
import torch

from .config import FiDConfig
from .relpos import bucket_lut

ALIGN = 64  # elements; keeps every block 256-B aligned in fp32 and 128-B aligned in bf16

# dropout site ids (unique per place a torch.nn.Dropout sits in the reference graph)
S_ENC_EMBED, S_ENC_FINAL, S_DEC_EMBED, S_DEC_FINAL = 1, 2, 3, 4


def _enc_site(i, k):   # k: 0 attn probs, 1 attn out, 2 ffn act, 3 ffn out
    return 100 + i * 8 + k


def _dec_site(i, k):   # k: 0 self probs, 1 self out, 2 cross probs, 3 cross out, 4 ffn act, 5 ffn out
    return 4000 + i * 8 + k


@dataclass
class Block:
    name: str
    shape: tuple
    members: list            # [(plain HF name, row0, nrows)]
    transpose: bool = False  # keep a transposed low-precision copy (GEMM weights)
    off: int = 0
    toff: int = -1

    @property
    def numel(self):
        return int(np.prod(self.shape))


class Mat:
    """One GEMM weight block: p fp32 master [N,K], g fp32 grad, w compute-dtype [N,K], wt compute-dtype [K,N]."""
    __slots__ = ("p", "g", "w", "wt")


class Vec:
    __slots__ = ("p", "g")


def build_layout(cfg: FiDConfig) -> list[Block]:
    d, f, inner, H, nb = cfg.d_model, cfg.d_ff, cfg.inner_dim, cfg.num_heads, cfg.relative_attention_num_buckets
    blocks: list[Block] = []

    def add(name, shape, members, transpose=False):
        blocks.append(Block(name, tuple(shape), members, transpose))

    Ld, Le = cfg.num_decoder_layers, cfg.num_layers
    add("dec.final_ln", (d,), [("decoder.final_layer_norm.weight", 0, d)])
    for i in range(Ld):
        p = f"decoder.block.{i}.layer."
        add(f"dec.{i}.qkv", (3 * inner, d), [(p + f"0.SelfAttention.{n}.weight", j * inner, inner)
                                             for j, n in enumerate("qkv")], True)
        add(f"dec.{i}.o", (d, inner), [(p + "0.SelfAttention.o.weight", 0, d)], True)
        add(f"dec.{i}.ln1", (d,), [(p + "0.layer_norm.weight", 0, d)])
        add(f"dec.{i}.cq", (inner, d), [(p + "1.EncDecAttention.q.weight", 0, inner)], True)
        add(f"dec.{i}.co", (d, inner), [(p + "1.EncDecAttention.o.weight", 0, d)], True)
        add(f"dec.{i}.ln2", (d,), [(p + "1.layer_norm.weight", 0, d)])
        add(f"dec.{i}.wi", (f, d), [(p + "2.DenseReluDense.wi.weight", 0, f)], True)
        add(f"dec.{i}.wo", (d, f), [(p + "2.DenseReluDense.wo.weight", 0, d)], True)
        add(f"dec.{i}.ln3", (d,), [(p + "2.layer_norm.weight", 0, d)])
    add("dec.rel", (nb, H), [("decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight", 0, nb)])
    if cfg.legacy_cross_bias:     # transformers 3.0.2: the first cross-attention layer owns a table, every cross-attention adds its bias
        add("dec.xrel", (nb, H), [("decoder.block.0.layer.1.EncDecAttention.relative_attention_bias.weight", 0, nb)])
    kv = []
    for i in range(Ld):
        p = f"decoder.block.{i}.layer.1.EncDecAttention."
        kv += [(p + "k.weight", 2 * i * inner, inner), (p + "v.weight", (2 * i + 1) * inner, inner)]
    add("dec.kv_all", (Ld * 2 * inner, d), kv, True)
    add("enc.final_ln", (d,), [("encoder.final_layer_norm.weight", 0, d)])
    for i in reversed(range(Le)):
        p = f"encoder.block.{i}.layer."
        add(f"enc.{i}.qkv", (3 * inner, d), [(p + f"0.SelfAttention.{n}.weight", j * inner, inner)
                                             for j, n in enumerate("qkv")], True)
        add(f"enc.{i}.o", (d, inner), [(p + "0.SelfAttention.o.weight", 0, d)], True)
        add(f"enc.{i}.ln1", (d,), [(p + "0.layer_norm.weight", 0, d)])
        add(f"enc.{i}.wi", (f, d), [(p + "1.DenseReluDense.wi.weight", 0, f)], True)
        add(f"enc.{i}.wo", (d, f), [(p + "1.DenseReluDense.wo.weight", 0, d)], True)
        add(f"enc.{i}.ln2", (d,), [(p + "1.layer_norm.weight", 0, d)])
    add("enc.rel", (nb, H), [("encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight", 0, nb)])
    add("shared", (cfg.vocab_size, d), [("shared.weight", 0, cfg.vocab_size)], True)
    off = toff = 0
    for b in blocks:
        b.off = off
        off += -(-b.numel // ALIGN) * ALIGN
        if b.transpose:
            b.toff = toff
            toff += -(-b.numel // ALIGN) * ALIGN
    return blocks


def layout_sizes(blocks):
    last = blocks[-1]
    n = last.off + -(-last.numel // ALIGN) * ALIGN
    tb = [b for b in blocks if b.transpose]
    nt = tb[-1].toff + -(-tb[-1].numel // ALIGN) * ALIGN
    return n, nt


@dataclass
class _Ragged:
    """Unpadded encoder batch: only the valid tokens of the [B·N, L] passages, packed in (passage, position) order."""
    M: int                  # number of valid tokens (rows of every encoder buffer)
    off: torch.Tensor       # int32 [B·N + 1] row offset of each passage
    soff: torch.Tensor      # int32 [B + 1]   row offset of each sample (its N passages are consecutive)
    idx: torch.Tensor       # int64 [M] position of each packed token in the flat [B·N·L] input (None: use pack(), lengths from the host)
    soff_h: list = None     # the sample offsets on the host
    order: torch.Tensor = None   # int32 [B·N] passages by descending length: the processing order of the attention kernels
    first_min: int = 0      # the shortest FIRST passage of a sample (legacy cross-attention bias: see Engine._legacy_layout)


@dataclass
class _XPlan:
    """Key layout of the encoder-state-space cross-attention (csrc/xattn.hip): sample b owns the encoder rows
    [k_h[b], k_h[b+1]) and the score columns [p_h[b], p_h[b] + n_keys(b)) of a [·, ptot] matrix (segments padded to 256)."""
    k_off: torch.Tensor     # int32 [B + 1] device
    p_off: torch.Tensor     # int32 [B + 1] device
    k_h: list = None
    p_h: list = None
    ptot: int = 0
    splits: int = 1         # workgroups sharing the keys of a sample in xattn_context
    p0: int = 0             # first score column of this plan inside the batch's [·, cap] matrices (a chain of samples: its window)
    sub: dict = None        # {n chains: [per-chain _XPlan]} (column offsets relative to the chain's window)


@dataclass
class _Ctx:
    """What forward leaves behind for backward."""
    B: int = 0
    N: int = 0
    L: int = 0
    T: int = 0
    p: float = 0.0
    seed: int = 0
    ids: torch.Tensor = None
    mask_u8: torch.Tensor = None
    enc_mask: torch.Tensor = None      # the ENCODER's key mask: mask_u8, or None when the batch has no padding at all (Engine._enc_mask)
    labels: torch.Tensor = None
    dec_ids: torch.Tensor = None
    ckpt: bool = False
    rag: object = None      # _Ragged: the encoder ran on the valid tokens only
    xa: object = None       # _XPlan: cross-attention ran in the encoder-state space
    chains: list = None     # the decoder's sample chains (_dec_chains)
    enc_ids: torch.Tensor = None   # the encoder's token ids: all [B·N·L], or the valid ones when rag is set
    ws: dict = field(default_factory=dict)


class Engine:
    def __init__(self, cfg: FiDConfig, ops, device, dtype=torch.bfloat16, seed: int = 0, fp8: bool | None = None):
        if cfg.d_kv not in (32, 64):
            raise ValueError(f"d_kv={cfg.d_kv} unsupported by the attention kernels (32 or 64)")
        if cfg.d_model % 8 or cfg.d_ff % 8 or cfg.vocab_size % 8:
            raise ValueError("d_model, d_ff and vocab_size must be multiples of 8")
        self.cfg, self.ops, self.device, self.dtype = cfg, ops, torch.device(device), dtype
        # LAKO_DETERMINISTIC=1 (read once by the library; RefOps has no such mode — its torch sums are what they are): bit-identical steps
        self.det = bool(getattr(ops, "det", False))
        self.blocks = build_layout(cfg)
        self.n_param, self.n_trans = layout_sizes(self.blocks)
        dev = self.device
        self.P = torch.zeros(self.n_param, dtype=torch.float32, device=dev)
        self.G = torch.zeros(self.n_param, dtype=torch.float32, device=dev)
        self._g_fresh = True         # G is all zeros (construction / zero_grad) and no backward has added to it yet
        self.W = self.P if dtype == torch.float32 else torch.zeros(self.n_param, dtype=dtype, device=dev)
        self.WT = torch.zeros(self.n_trans, dtype=dtype, device=dev)
        self.opt_m = None   # AdamW moments, allocated by the optimizer on first use
        self.opt_v = None
        self.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.by_name = {b.name: b for b in self.blocks}
        self._bind()
        self.seed_base = int(seed)
        self.step_count = 0
        self.use_checkpoint = False
        self._ws_cache: dict = {}
        self._tr_table = None
        self._rag_cache: dict = {}
        self._xplan_cache: dict = {}
        self.xattn_active = False
        self._all_valid = False
        self._row_cap: dict = {}     # {rows of the current unpadded batch: padded row count}
        self._lut_cache: dict = {}
        self.ctx: _Ctx | None = None
        self.grad_hook = None    # callable(lo, hi): gradients G[lo:hi] are final (data-parallel overlap)
        self.grad_ranges_early = True   # the hook wants every range as early as possible (overlap mode); False: only that all are done
        self.shadows_stale = True
        # MX block-scaled fp8 forward GEMMs (BASELINE config 5): the encoder's QKV and FFN-in projections and the cross-K/V
        # projection — the GEMMs whose A operand is a norm output — multiply e4m3 operands with E8M0 block scales on
        # v_mfma_scale_f32_16x16x128_f8f6f4; everything else (attention, norms, the other GEMMs, the whole backward) stays bf16
        self.fp8 = bool(int(os.environ.get("LAKO_FP8", "0"))) if fp8 is None else bool(fp8)
        if self.fp8 and (dtype != torch.bfloat16 or not hasattr(ops, "gemm_nt_mx")):
            raise ValueError("fp8 GEMMs need the bf16 engine on the HIP op set")
        self._w8: dict = {}      # block name → (e4m3 bytes [N, K], scales) of the weight shadow

    # ------------------------------------------------------------------------------------------
    # parameter views
    # ------------------------------------------------------------------------------------------
    def _view(self, flat, b: Block, transposed=False):
        if transposed:
            return flat[b.toff:b.toff + b.numel].view(b.shape[1], b.shape[0])
        return flat[b.off:b.off + b.numel].view(b.shape)

    def _mat(self, name) -> Mat:
        b = self.by_name[name]
        m = Mat()
        m.p, m.g, m.w = self._view(self.P, b), self._view(self.G, b), self._view(self.W, b)
        m.wt = self._view(self.WT, b, True)
        return m

    def _vec(self, name) -> Vec:
        b = self.by_name[name]
        v = Vec()
        v.p, v.g = self._view(self.P, b), self._view(self.G, b)
        return v

    def _bind(self):
        cfg = self.cfg
        self.shared = self._mat("shared")
        self.kv_all = self._mat("dec.kv_all")
        self.enc_final, self.dec_final = self._vec("enc.final_ln"), self._vec("dec.final_ln")
        self.enc_rel, self.dec_rel = self._vec("enc.rel"), self._vec("dec.rel")
        self.dec_xrel = self._vec("dec.xrel") if cfg.legacy_cross_bias else None
        self.enc, self.dec = [], []
        for i in range(cfg.num_layers):
            self.enc.append(dict(qkv=self._mat(f"enc.{i}.qkv"), o=self._mat(f"enc.{i}.o"), ln1=self._vec(f"enc.{i}.ln1"),
                                 wi=self._mat(f"enc.{i}.wi"), wo=self._mat(f"enc.{i}.wo"), ln2=self._vec(f"enc.{i}.ln2")))
        for i in range(cfg.num_decoder_layers):
            self.dec.append(dict(qkv=self._mat(f"dec.{i}.qkv"), o=self._mat(f"dec.{i}.o"), ln1=self._vec(f"dec.{i}.ln1"),
                                 cq=self._mat(f"dec.{i}.cq"), co=self._mat(f"dec.{i}.co"), ln2=self._vec(f"dec.{i}.ln2"),
                                 wi=self._mat(f"dec.{i}.wi"), wo=self._mat(f"dec.{i}.wo"), ln3=self._vec(f"dec.{i}.ln3")))

    def named_param_views(self, flat=None):
        """plain HF-T5 name → view into `flat` (default: the fp32 master)."""
        flat = self.P if flat is None else flat
        out = {}
        for b in self.blocks:
            v = self._view(flat, b)
            for name, r0, nr in b.members:
                out[name] = v[r0:r0 + nr] if len(b.shape) == 2 else v
        return out

    def refresh_transposed(self, from_shadow=False):
        """WT ← transposed compute-dtype copies of every GEMM weight in P, one launch over a static table.  from_shadow: the
        compute-dtype shadow W is current (the fused optimizer step has just written it) — transpose IT: identical values
        (W = bf16(P) element for element) from half the bytes (446 instead of 892 MB at T5-base)."""
        if self._tr_table is None:
            mats = [b for b in self.blocks if b.transpose]
            desc, prefix, total = [], [], 0
            for b in mats:
                desc += [b.off, b.toff, b.shape[0], b.shape[1]]
                prefix.append(total)
                total += -(-b.shape[0] // 64) * -(-b.shape[1] // 64)
            self._tr_table = (torch.tensor(desc, dtype=torch.int64, device=self.device),
                              torch.tensor(prefix, dtype=torch.int32, device=self.device), len(mats), total)
        src = self.W if (from_shadow and self.W is not self.P and self.W.dtype == self.WT.dtype) else self.P
        self.ops.transpose_cast_batched(src, self.WT, *self._tr_table)

    def refresh_shadows(self):
        """Re-derive the compute-dtype copies (W, WT) from the fp32 master after it changed."""
        if self.W is not self.P:
            self.ops.cast(self.P, self.W)
        self.refresh_transposed()
        if self.fp8:
            self.refresh_fp8()
        self.shadows_stale = False

    def refresh_after_step(self):
        """After the fused optimizer step (which wrote P and the compute-dtype shadow W in one pass): every DERIVED copy of the
        weights — the transposed shadows and, in fp8 mode, the e4m3 + block-scale shadows the forward GEMMs multiply with."""
        self.refresh_transposed(from_shadow=True)
        if self.fp8:
            self.refresh_fp8()
        self.shadows_stale = False

    def _fp8_blocks(self):
        names = [f"enc.{i}.{m}" for i in range(self.cfg.num_layers) for m in ("qkv", "wi")] + ["dec.kv_all"]
        return [self.by_name[n] for n in names if self.by_name[n].shape[1] % 128 == 0]

    def refresh_fp8(self):
        """e4m3 + block-scale shadows of the weights that take part in fp8 GEMMs, from the bf16 shadow W"""
        for b in self._fp8_blocks():
            if b.name not in self._w8:
                n, k = b.shape
                self._w8[b.name] = (torch.empty(n, k, dtype=torch.uint8, device=self.device),      # (scale padding bytes stay 0)
                                    torch.zeros(n, self.ops.mx_scale_cols(k), dtype=torch.uint8, device=self.device))
            q, sc = self._w8[b.name]
            self.ops.mx_quantize(self._view(self.W, b), q, sc)

    def _norm_gemm_w(self, ws, h, lnp, xn, rs, eps, name, w, out, **epi):
        """xn = RMSNorm(h); out = epilogue(xn · wᵀ).  With an fp8 shadow of this weight the norm also emits the e4m3 bytes and block scales of xn
        (lako_rmsnorm_fwd_mx: one pass instead of norm + quantiser — the same bytes), which the MX product consumes."""
        w8 = self._w8.get(name) if self.fp8 else None
        M, K = h.shape
        if w8 is None or K % 128 != 0 or not hasattr(self.ops, "rmsnorm_fwd_mx") or os.environ.get("LAKO_FP8_FUSED_NORM", "1") == "0":
            self.ops.rmsnorm_fwd(h, lnp, xn, rs, eps)
            return self._gemm_w(ws, xn, name, w, out, **epi)
        xq, xs = self._q8_buffers(ws, M, K)
        self.ops.rmsnorm_fwd_mx(h, lnp, xn, rs, eps, xq, xs)
        self.ops.gemm_nt_mx(xq, xs, w8[0], w8[1], out, **epi)

    def _q8_buffers(self, ws, M, K):
        xq = self._buf(ws, f"q8.{K}", (M, K), torch.uint8)
        fresh = ws.get(f"^s8.{K}")
        xs = self._buf(ws, f"s8.{K}", (M, self.ops.mx_scale_cols(K)), torch.uint8)
        if ws[f"^s8.{K}"] is not fresh:
            self.ops.zero_(ws[f"^s8.{K}"])      # the quantiser writes ⌈K/128⌉ of the KSP scale columns: the padding stays 0
        return xq, xs

    def _gemm_w(self, ws, x, name, w, out, **epi):
        """out = epilogue(x · wᵀ): on the block-scaled fp8 matrix cores when this weight has an fp8 shadow (x is quantised into
        workspace scratch first), else the bf16 GEMM"""
        w8 = self._w8.get(name) if self.fp8 else None
        if w8 is None:
            return self.ops.gemm_nt(x, w, out, **epi)
        M, K = x.shape
        xq, xs = self._q8_buffers(ws, M, K)
        self.ops.mx_quantize(x, xq, xs)
        self.ops.gemm_nt_mx(xq, xs, w8[0], w8[1], out, **epi)

    def zero_grad(self):
        self.ops.zero_(self.G)
        self._g_fresh = True

    # ------------------------------------------------------------------------------------------
    # helpers
    # ------------------------------------------------------------------------------------------
    def _lut(self, qlen, klen, bidirectional):
        key = (qlen, klen, bidirectional)
        if key not in self._lut_cache:
            lut = bucket_lut(qlen, klen, bidirectional, self.cfg.relative_attention_num_buckets,
                             self.cfg.relative_attention_max_distance)
            self._lut_cache[key] = torch.from_numpy(lut).to(self.device)
        return self._lut_cache[key]

    def _legacy_layout(self, rag, T):
        """`legacy_cross_bias` (transformers 3.0.2, src/model.py:323-329): every cross-attention adds
        `bias[h, t, s] = table[bucket(s − t)][h]` with the DECODER's one-sided buckets (`bidirectional = not is_decoder`), s the position of
        the key in the concatenated [n_passages · text_maxlength] sequence.  Keys at or after the query position all share bucket 0, so
        the bias differs from a per-head constant only for the first t keys of the FIRST passage — which is why it is an ordinary
        relative table for the projected kernels (index key − query + T − 1 into `[H, 2T − 1]`, indices past the end clamped onto
        "distance >= 0") as long as a key's index IS its position there: always on the padded layout; on the packed layout when every
        sample's first passage has at least T − 1 valid tokens (the packed index of a later passage's token is smaller than its
        position).  Returns the layout to use: `rag`, or None (padded) when a first passage is too short."""
        if rag is not None and self.cfg.legacy_cross_bias and rag.first_min < T - 1:
            return None
        return rag

    def _buf(self, ws, name, shape, dtype=None):
        """Workspace tensor `name` of this shape.  The backing allocation only grows: a request with fewer ROWS (the
        unpadded encoder: the number of valid tokens changes from batch to batch) is a prefix view of it."""
        shape = tuple(shape)
        t = ws.get(name)
        if t is not None and t.shape == shape and t.dtype == (dtype or self.dtype):      # the same request as last time (the common case)
            return t
        base = ws.get("^" + name)
        if base is None or tuple(base.shape[1:]) != shape[1:] or base.shape[0] < shape[0] or base.dtype != (dtype or self.dtype):
            # token-row buffers of an unpadded batch are allocated at the padded row count once, so that a later batch with
            # more valid tokens never reallocates the workspace in the middle of training
            rows = max(shape[0], self._row_cap.get(shape[0], 0))
            base = torch.empty((rows,) + shape[1:], dtype=dtype or self.dtype, device=self.device)
            ws["^" + name] = base
        t = base if base.shape[0] == shape[0] else base[:shape[0]]
        ws[name] = t
        return t

    def _tn_scratch(self, nbytes):
        """grow-only device scratch for the weight-gradient launches whose K-splits meet through partial tiles (lako_gemm_tn_grouped).
        OFF by default (LAKO_TN_SLABS=1 switches it on): measured +0.6 ms per step against the float atomics (round 4,
        profiles/r04s_tn_slab_reduction.txt) — the atomics are fire-and-forget, the last arriver's slab reads and read-modify-write are
        a serial tail.  What it buys is bit-reproducible encoder weight gradients."""
        if os.environ.get("LAKO_TN_SLABS", "0") != "1" and not self.det:      # (LAKO_DETERMINISTIC=1 takes it: DESIGN.md §4)
            return None
        t = getattr(self, "_tn_ws", None)
        if t is None or t.numel() < nbytes:
            t = self._tn_ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        return t

    def _embed_bwd(self, ids, dout, drop):
        """shared-embedding gradient += scatter of dropout_bwd(dout).  LAKO_DETERMINISTIC=1: rows of one id are added in position order by
        one wave (the sort is index plumbing, torch's stable argsort on the same stream) instead of by float atomics in arrival order."""
        if self.det:
            ids = ids.reshape(-1)
            self.ops.embed_bwd_ordered(ids, torch.argsort(ids, stable=True), dout, self.shared.g, drop)
        else:
            self.ops.embed_bwd(ids, dout, self.shared.g, drop)

    def _workspace(self, key):
        return self._ws_cache.setdefault(key, {})

    def _heads(self, t2d, rows_b, rows_t, col0):
        """[rows_b*rows_t, ld] buffer → [B, T, H, dk] view of columns [col0, col0 + inner)."""
        H, dk = self.cfg.num_heads, self.cfg.d_kv
        if t2d.shape[0] != rows_b * rows_t or t2d.stride(1) != 1 or col0 + H * dk > t2d.shape[1]:
            raise ValueError(f"_heads: {tuple(t2d.shape)} is not [{rows_b}*{rows_t}, >= {col0 + H * dk}] row-major")
        s0 = t2d.stride(0)          # (one view op instead of view → slice → unflatten: this runs ≈290 times per training step)
        return t2d.as_strided((rows_b, rows_t, H, dk), (rows_t * s0, s0, dk, 1), t2d.storage_offset() + col0)

    # ------------------------------------------------------------------------------------------
    # forward
    # ------------------------------------------------------------------------------------------
    def _enc_layer_fwd(self, ws, i, j, h, h_out, BN, L, mask_u8, rel, dr, rag=None):
        """One encoder block (HF5:448-509): RMSNorm → fused QKV → attention → O+residual → RMSNorm → FFN+residual.
        Intermediates are written to the workspace slot `j` (per layer when everything is kept, slot 0 when the
        layer is recomputed in backward or nothing is kept)."""
        cfg, ops = self.cfg, self.ops
        d, f, inner, H, eps = cfg.d_model, cfg.d_ff, cfg.inner_dim, cfg.num_heads, cfg.layer_norm_epsilon
        Me, lw = (rag.M if rag is not None else BN * L), self.enc[i]
        xn1 = self._buf(ws, f"e.xn1.{j}", (Me, d))
        qkv = self._buf(ws, f"e.qkv.{j}", (Me, 3 * inner))
        self._norm_gemm_w(ws, h, lw["ln1"].p, xn1, self._buf(ws, f"e.rs1.{j}", (Me,), torch.float32), eps, f"enc.{i}.qkv", lw["qkv"].w, qkv)
        ctx = self._buf(ws, f"e.ctx.{j}", (Me, inner))
        hb, ht, akw = self._enc_attn_layout(rag, BN, L, mask_u8)
        ops.attn_fwd(self._heads(qkv, hb, ht, 0), self._heads(qkv, hb, ht, inner), self._heads(qkv, hb, ht, 2 * inner),
                     self._heads(ctx, hb, ht, 0), self._buf(ws, f"e.st.{j}", (BN, H, L, 4), torch.float32),
                     rel_bias=rel, rel_off=L - 1, drop=dr(_enc_site(i, 0)), **akw)
        h1 = self._buf(ws, f"e.h1.{j}", (Me, d))
        ops.gemm_nt(ctx, lw["o"].w, h1, resid=h, drop=dr(_enc_site(i, 1)))
        xn2 = self._buf(ws, f"e.xn2.{j}", (Me, d))
        a1 = self._buf(ws, f"e.a1.{j}", (Me, f))
        self._norm_gemm_w(ws, h1, lw["ln2"].p, xn2, self._buf(ws, f"e.rs2.{j}", (Me,), torch.float32), eps, f"enc.{i}.wi", lw["wi"].w, a1, relu=True,
                          drop=dr(_enc_site(i, 2)))
        if h_out is not None:
            ops.gemm_nt(a1, lw["wo"].w, h_out, resid=h1, drop=dr(_enc_site(i, 3)))

    @staticmethod
    def _enc_attn_layout(rag, BN, L, mask_u8):
        """(batch, time) of the [·, ·, H, dk] head views of the token-major buffers + the attention kwargs: padded
        [BN, L] with a key mask, or ONE packed run of rows with per-passage offsets."""
        if rag is None:      # (mask_u8 None: a batch without any padding — no mask to apply, and the encoder's fast attention kernels take it)
            return BN, L, (dict(key_mask=mask_u8) if mask_u8 is not None else {})
        return 1, rag.M, dict(q_off=rag.off, k_off=rag.off, max_q=L, max_k=L, order=rag.order)

    def _enc_mask(self, rag, mask_u8):
        """The key mask of the encoder's self-attention on the padded layout — None when the batch has no padded position at all (what
        _ragged_batch found: every passage at text_maxlength, e.g. `bench.py --all-valid`): a mask of ones adds nothing, and without one
        the padded [B·N, L] batch is a set of exact-length sequences for the fast encoder kernels (csrc/attn_enc.hip) instead of the generic
        masked ones.  LAKO_ENC_ALLVALID_MASK=1 keeps the mask (A/B)."""
        if rag is None and self._all_valid and os.environ.get("LAKO_ENC_ALLVALID_MASK", "0") != "1":
            return None
        return mask_u8

    def _encode(self, ws, ids_flat, mask_u8, BN, L, p, seed, save, rag=None, want_kv=True):
        """save: True  — keep every intermediate of every layer (training; 288 GB of HBM make this the default);
                 "ckpt" — keep only each block's input and recompute the block in backward (`set_checkpoint(True)`,
                          the reference's CheckpointWrapper, src/model.py:237-283);
                 False — keep nothing (generate)."""
        cfg, ops = self.cfg, self.ops
        d, H = cfg.d_model, cfg.num_heads
        Me, Le, eps = (rag.M if rag is not None else BN * L), cfg.num_layers, cfg.layer_norm_epsilon
        dr = (lambda site: (p, seed, site)) if p > 0 else (lambda site: None)
        ix = (lambda i: i) if save is True else (lambda i: 0)
        hx = (lambda i: i) if save else (lambda i: i % 2)
        ops.embed_fwd(ids_flat, self.shared.w, self._buf(ws, f"e.h{hx(0)}", (Me, d)), dr(S_ENC_EMBED))
        rel = self._buf(ws, "e.rel", (H, 2 * L - 1), torch.float32)
        ops.relpos_expand(self.enc_rel.p, self._lut(L, L, True), rel)
        for i in range(Le):
            self._enc_layer_fwd(ws, i, ix(i), ws[f"e.h{hx(i)}"], self._buf(ws, f"e.h{hx(i + 1)}", (Me, d)), BN, L,
                                mask_u8, rel, dr, rag)
        enc_out = self._buf(ws, "e.out", (Me, d))
        ops.rmsnorm_fwd(ws[f"e.h{hx(Le)}"], self.enc_final.p, enc_out, self._buf(ws, "e.rsf", (Me,), torch.float32),
                        eps, dr(S_ENC_FINAL))
        if not want_kv:      # cross-attention in the encoder-state space: the keys and values ARE enc_out
            return enc_out, None
        # K/V projections of every decoder layer's cross-attention in ONE GEMM (reads enc_out once)
        kv = self._buf(ws, "e.kv", (Me, self.kv_all.w.shape[0]))
        self._gemm_w(ws, enc_out, "dec.kv_all", self.kv_all.w, kv)
        return enc_out, kv

    def _ragged_batch(self, attention_mask, B, N, L, lengths=None):
        """Padded positions never influence the loss, the gradients or the decoded tokens: they are masked as keys in the
        encoder's self-attention and in the decoder's cross-attention, and nothing else reads them.  When the mask has the
        tokenizer's form (valid tokens first, then padding) the encoder therefore runs on the valid tokens only — a quarter
        fewer rows in every GEMM / norm / attention at the OKVQA-like lengths of the benchmark.  Returns None (padded path)
        for LAKO_UNPAD=0, masks with holes, a sample without any valid token, or when nothing is padded.

        `lengths` (host int tensor [B, N] or [B·N]): the number of valid tokens of every passage as the collator that
        built the mask knows it (the mask is born on the host: src/data.py:88-104 pads to text_maxlength) — asserts the
        prefix form, so no device→host sync is needed: offsets are built on the host, uploaded from pinned memory, and the
        packed-row index is expanded on the device.  Without it the lengths are read back from the mask (one host sync per
        new mask tensor; repeated batches hit a cache keyed by the tensor's identity and version)."""
        self._all_valid = False
        if os.environ.get("LAKO_UNPAD", "1") == "0":
            return None
        dev = attention_mask.device
        if lengths is not None:
            lens_h = torch.as_tensor(lengths, dtype=torch.int32, device="cpu").reshape(-1)
            if lens_h.numel() != B * N:
                raise ValueError(f"passage_lengths has {lens_h.numel()} entries for {B * N} passages")
            if int(lens_h.min()) < 0 or int(lens_h.max()) > L:
                raise ValueError("passage_lengths out of range [0, L]")
            M = int(lens_h.sum())
            self._all_valid = M == B * N * L
            self._check_lengths(attention_mask, lens_h, B, N, L)
            if int(lens_h.view(B, N).sum(1).min()) <= 0 or M >= B * N * L:
                return None
            off = torch.zeros(B * N + 1, dtype=torch.int32)
            torch.cumsum(lens_h, 0, out=off[1:])
            pin = (lambda t: t.pin_memory()) if dev.type == "cuda" else (lambda t: t)
            off_d = pin(off).to(dev, non_blocking=True)
            idx = None      # (the packed token ids come from ONE lako_pack_ids launch over the offsets: no packed-row index on this path)
            # passages sorted by descending length: the order the encoder-attention workgroups take them in (load balance only:
            # long items first, every persistent workgroup one passage of each length class — measured with pre-sorted lengths:
            # attention −0.24 ms per step)
            order = torch.argsort(lens_h, descending=True, stable=True).to(torch.int32)
            order_d = pin(order).to(dev, non_blocking=True) if os.environ.get("LAKO_ATTN_ORDER", "1") != "0" else None
            return _Ragged(M=M, off=off_d, soff=off_d[::N].contiguous(), idx=idx, soff_h=off[::N].tolist(), order=order_d,
                           first_min=int(lens_h.view(B, N)[:, 0].min()))
        key = (attention_mask.data_ptr(), attention_mask._version, B, N, L)
        hit = self._rag_cache.get(key)
        if hit is not None:
            self._all_valid = hit[2]
            return hit[0]
        m = attention_mask.reshape(B * N, L).bool()
        lens = m.sum(1, dtype=torch.int32)
        prefix = (m == (torch.arange(L, device=m.device)[None, :] < lens[:, None])).all()
        host = torch.cat([lens, prefix.to(torch.int32)[None]]).cpu()
        lens_h, ok = host[:-1], bool(host[-1])
        per_sample = lens_h.view(B, N).sum(1)
        rag = None
        if ok and int(per_sample.min()) > 0 and int(lens_h.sum()) < B * N * L:
            off = torch.zeros(B * N + 1, dtype=torch.int32)
            off[1:] = torch.cumsum(lens_h, 0)
            rag = _Ragged(M=int(off[-1]), off=off.to(dev), soff=off[::N].contiguous().to(dev),
                          idx=m.reshape(-1).nonzero().reshape(-1), soff_h=off[::N].tolist(),
                          order=torch.argsort(lens_h, descending=True, stable=True).to(torch.int32).to(dev)
                          if os.environ.get("LAKO_ATTN_ORDER", "1") != "0" else None,
                          first_min=int(lens_h.view(B, N)[:, 0].min()))
        self._all_valid = ok and int(lens_h.sum()) == B * N * L
        if len(self._rag_cache) >= 16:
            self._rag_cache.clear()
        self._rag_cache[key] = (rag, attention_mask, self._all_valid)   # holding the tensor keeps (data_ptr, version) unambiguous
        return rag

    def _packed_ids(self, ws, ids_flat, rag, L):
        """token ids of the valid positions in packed order"""
        if rag.idx is not None:
            return ids_flat[rag.idx]
        out = self._buf(ws, "e.ids", (rag.M,), torch.int64)
        self.ops.pack_ids(ids_flat, rag.off, out, L)
        return out

    def _check_lengths(self, attention_mask, lens_h, B, N, L):
        """`passage_lengths` must describe the mask (valid tokens first, exactly lengths[b, n] of them): anything else would pack the
        wrong tokens without a sign.  Every batch's mask is therefore compared with the uploaded lengths ON THE DEVICE (round 6: on by
        default; LAKO_CHECK_LENGTHS=0 switches it off), and the one-byte verdict travels to pinned host memory behind an event — the
        training path still never waits for the device: a verdict is looked at only once its event has completed (`query()`), i.e. a
        mismatch raises a step or two late instead of never; `check_lengths_now()` waits for everything outstanding (end of an epoch,
        tests).  Cost: five small launches on [B·N, L] booleans, ≈ 0.03 ms per step at config 2 (profiles/r06j_length_check_cost.txt)."""
        if os.environ.get("LAKO_CHECK_LENGTHS", "1") == "0":
            return
        self._poll_lengths(block=False)
        dev = attention_mask.device
        m = attention_mask.reshape(B * N, L).bool()
        if dev.type != "cuda":                # the CPU test double: nothing to overlap with — compare at once
            want = torch.arange(L)[None, :] < lens_h[:, None]
            if bool((m != want).any()):
                self._lengths_error()
            return
        ar = self._len_arange.get(L) if hasattr(self, "_len_arange") else None
        if ar is None or ar.device != dev:
            if not hasattr(self, "_len_arange"):
                self._len_arange, self._len_pending = {}, []
            ar = self._len_arange[L] = torch.arange(L, device=dev, dtype=torch.int32)[None, :]
        want = ar < lens_h.pin_memory().to(dev, non_blocking=True)[:, None]
        flag = torch.empty(1, dtype=torch.bool).pin_memory()
        flag.copy_((m != want).any().reshape(1), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._len_pending.append((ev, flag))

    @staticmethod
    def _lengths_error():
        raise ValueError("passage_lengths do not describe the attention mask of an earlier batch: the mask must be valid-tokens-first with "
                         "exactly lengths[b, n] valid tokens per passage (LAKO_CHECK_LENGTHS=0 switches the check off)")

    def _poll_lengths(self, block):
        pend = getattr(self, "_len_pending", None)
        while pend and (block or pend[0][0].query()):
            ev, flag = pend.pop(0)
            if block:
                ev.synchronize()
            if bool(flag[0]):                 # (pinned host memory: no device access)
                pend.clear()
                self._lengths_error()

    def check_lengths_now(self):
        """wait for every outstanding mask-vs-lengths verdict and raise on a mismatch"""
        self._poll_lengths(block=True)

    def _xattn_plan(self, rag, B, N, L, rows_q):
        """The cross-attention runs in the encoder-state space (csrc/xattn.hip: no K/V projection of the N·L encoder states, the
        per-head projections act on the T·H query rows instead) whenever every sample's keys are a contiguous run of encoder rows:
        the unpadded batch, or a batch without any padding.  Masks with holes / LAKO_UNPAD=0 / LAKO_XATTN=0 / shapes the kernels
        do not take (fp32 engine, d_kv != 64, d_model % 128) keep the projected formulation (lako_attn_fwd with a key mask)."""
        cfg = self.cfg
        ok = getattr(self.ops, "xattn_ok", None)
        if cfg.legacy_cross_bias:      # the bias is an argument of the projected kernels only (_legacy_layout)
            return None
        if os.environ.get("LAKO_XATTN", "1") == "0" or ok is None or not ok(self.dtype, cfg.d_kv, cfg.d_model) or B > 512:
            return None
        if rag is not None:
            k_h = list(rag.soff_h)
        elif self._all_valid:
            k_h = [b * N * L for b in range(B + 1)]
        else:
            return None
        key = (tuple(k_h), rows_q)
        hit = self._xplan_cache.get(key)
        if hit is not None:
            return hit
        p_h = [0]
        for b in range(B):
            p_h.append(p_h[-1] + -(-(k_h[b + 1] - k_h[b]) // 256) * 256)
        host = torch.tensor([k_h, p_h], dtype=torch.int32)
        if self.device.type == "cuda":
            host = host.pin_memory()
        dev = host.to(self.device, non_blocking=True)
        # one 512-thread workgroup per CU: split a sample's keys until ≈256 workgroups are in flight
        per = max(1, cfg.d_model // 128) * B * -(-rows_q // 128)
        splits = int(os.environ.get("LAKO_XATTN_SPLITS", "0")) or max(1, min(8, 256 // per))
        plan = _XPlan(k_off=dev[0], p_off=dev[1], k_h=k_h, p_h=p_h, ptot=p_h[-1], splits=splits)
        if len(self._xplan_cache) >= 8:
            self._xplan_cache.clear()
        self._xplan_cache[key] = plan
        return plan

    def _xattn_subplans(self, xa, n, B):
        """The plan cut into n chains of samples: chain c owns the score columns [p_h[b0], p_h[b1]) — its kernels get views of that
        window and column offsets relative to it (the kernels launch one workgroup per 256 columns from column 0 of what they are
        given) — and the absolute key offsets of its samples.  Cached with the plan."""
        if xa.sub is None:
            xa.sub = {}
        if n not in xa.sub:
            bounds = [(c * B // n, (c + 1) * B // n) for c in range(n)]
            rel = [[p - xa.p_h[b0] for p in xa.p_h[b0:b1 + 1]] for b0, b1 in bounds]
            host = torch.tensor([v for r in rel for v in r], dtype=torch.int32)
            if self.device.type == "cuda":
                host = host.pin_memory()
            dev = host.to(self.device, non_blocking=True)
            plans, at = [], 0
            for (b0, b1), r in zip(bounds, rel):
                plans.append(_XPlan(k_off=xa.k_off[b0:b1 + 1], p_off=dev[at:at + len(r)], k_h=xa.k_h[b0:b1 + 1], p_h=r, ptot=r[-1],
                                    splits=xa.splits, p0=xa.p_h[b0]))
                at += len(r)
            xa.sub[n] = plans
        return xa.sub[n]

    def _xw(self, i, which, transposed=False):
        """Per-head views of decoder layer i's cross-attention K ("k") or V ("v") projection inside the kv_all block:
        [H, dk, d] of the weight, or (transposed) [H, d, dk] of its transposed shadow — the B operands of headbatch_nt."""
        inner, H, dk = self.cfg.inner_dim, self.cfg.num_heads, self.cfg.d_kv
        r0 = (2 * i + (which == "v")) * inner
        if transposed:
            return self.kv_all.wt[:, r0:r0 + inner].unflatten(1, (H, dk)).permute(1, 0, 2)
        return self.kv_all.w[r0:r0 + inner].unflatten(0, (H, dk))

    def _xg(self, i, which):
        inner, H, dk = self.cfg.inner_dim, self.cfg.num_heads, self.cfg.d_kv
        r0 = (2 * i + (which == "v")) * inner
        return self.kv_all.g[r0:r0 + inner].unflatten(0, (H, dk))

    def _xattn_buffers(self, ws, B, N, L, T, Ld, Z):
        """Workspace of the encoder-state-space cross-attention, sized for the padded batch (the column count of a batch's score
        matrices, xa.ptot, is a prefix of the allocation).  Per decoder layer l and R = T·H query rows per sample:
          dq [B, Ld·2R, d]   rows l·2R … : dC' (written in backward), rows l·2R + R … : Q'          — B operand of the dE product
          ps [Ld·2R, cap]    rows l·2R … : dropout(P),               rows l·2R + R … : dS (backward) — A operand of the dE product
          s  [Ld, R, cap] fp32 raw scores, st [Ld, B, R, 2] softmax statistics, c [Ld, Z, B, R, d] fp32 C' = P·E in Z key-split slabs"""
        cfg = self.cfg
        R, d = T * cfg.num_heads, cfg.d_model
        cap = B * (-(-(N * L) // 256) * 256)
        return dict(R=R, cap=cap, rows=B * N * L,
                    dq=self._buf(ws, "x.dq", (B, Ld * 2 * R, d)),
                    ps=self._buf(ws, "x.ps", (Ld * 2 * R, cap)),
                    s=self._buf(ws, "x.s", (Ld, R, cap), torch.float32),
                    st=self._buf(ws, "x.st", (Ld, B, R, 2), torch.float32),
                    c=self._buf(ws, "x.c", (Ld, Z, B, R, d), torch.float32))

    def _xattn_fwd(self, xb, xa, i, qc, enc_out, c2, B, T, max_keys, drop):
        """Cross-attention of decoder layer i in the encoder-state space (see csrc/xattn.hip): qc [B·T, inner] → c2 [B·T, inner]"""
        ops, H, dk, R = self.ops, self.cfg.num_heads, self.cfg.d_kv, xb["R"]
        qp = xb["dq"][:, (2 * i + 1) * R:(2 * i + 2) * R]
        ops.headbatch_nt(qc.view(B, T, H, dk), self._xw(i, "k", transposed=True), qp.unflatten(1, (T, H)))
        s = xb["s"][i][:, xa.p0:xa.p0 + xa.ptot]
        ops.xattn_scores(qp, enc_out, xa.k_off, xa.p_off, xa.ptot, s)
        pr = xb["ps"][2 * i * R:(2 * i + 1) * R, xa.p0:xa.p0 + xa.ptot]
        ops.xattn_softmax_fwd(s, xb["st"][i], pr, xa.k_off, xa.p_off, T, H, max_keys, drop)
        ops.xattn_context(pr, enc_out, xa.k_off, xa.p_off, xb["c"][i])
        ops.headbatch_nt(xb["c"][i].unflatten(2, (T, H)), self._xw(i, "v"), c2.view(B, T, H, dk))

    def _xattn_bwd(self, xb, xa, tmp, i, qc, enc_out, dctx, dqc, B, T, max_keys, drop, b0=0, B_all=None, hb=None):
        """Backward of _xattn_fwd: dctx [B·T, inner] → dqc [B·T, inner], the layer's Wk / Wv gradients, and the rows (dC', dS) the
        encoder-state gradient is assembled from after the last layer (_xattn_denc)."""
        ops, H, dk, R, d = self.ops, self.cfg.num_heads, self.cfg.d_kv, xb["R"], self.cfg.d_model
        dcp = xb["dq"][:, 2 * i * R:(2 * i + 1) * R]
        d4 = dctx.view(B, T, H, dk)
        ops.headbatch_nt(d4, self._xw(i, "v", transposed=True), dcp.unflatten(1, (T, H)))
        # (`hb`: list collecting the layers' Wk / Wv gradient products for ONE launch after the last layer — their operands then live in
        # per-layer buffers: the caller's dctx, and dqp below)
        if hb is not None:
            hb.append((d4, xb["c"][i].unflatten(2, (T, H)), self._xg(i, "v")))
        else:
            ops.headbatch_tn(d4, xb["c"][i].unflatten(2, (T, H)), self._xg(i, "v"))
        dp = self._buf(tmp, "x.dp", (R, xb["cap"]), torch.float32)[:, xa.p0:xa.p0 + xa.ptot]
        ops.xattn_scores(dcp, enc_out, xa.k_off, xa.p_off, xa.ptot, dp)
        ds = xb["ps"][(2 * i + 1) * R:(2 * i + 2) * R, xa.p0:xa.p0 + xa.ptot]
        ops.xattn_softmax_bwd(xb["s"][i][:, xa.p0:xa.p0 + xa.ptot], dp, xb["st"][i], ds, xa.k_off, xa.p_off, T, H, max_keys, drop)
        dqp = self._buf(tmp, f"x.dqp.{i}" if hb is not None else "x.dqp", (xa.splits, B_all or B, R, d), torch.float32)[:, b0:b0 + B]      # (a chain: its samples' slabs)
        ops.xattn_context(ds, enc_out, xa.k_off, xa.p_off, dqp)
        ops.headbatch_nt(dqp.unflatten(2, (T, H)), self._xw(i, "k"), dqc.view(B, T, H, dk))
        if hb is not None:
            hb.append((qc.view(B, T, H, dk), dqp.unflatten(2, (T, H)), self._xg(i, "k")))
        else:
            ops.headbatch_tn(qc.view(B, T, H, dk), dqp.unflatten(2, (T, H)), self._xg(i, "k"))

    def _xattn_denc(self, xb, xa, tmp, Me, dxe):
        """dE[s, :] = Σ_layers Σ_r P[r, s]·dC'[r, :] + dS[r, s]·Q'[r, :] — per sample one [keys, d] = psᵀ·dq product with
        K = layers·2R (weight-gradient form: both operands K-major), fp32, then rounded once to the compute dtype."""
        ops, d = self.ops, self.cfg.d_model
        de = self._buf(tmp, "x.de", (xb["rows"] + 8, d), torch.float32)[:Me + 8]    # sized for the padded batch
        items = []
        for b in range(len(xa.k_h) - 1):
            nk = xa.k_h[b + 1] - xa.k_h[b]
            n8 = -(-nk // 8) * 8      # the operand loads want a multiple of 8 columns (the score matrices' zero padding); only the
            items.append((xb["ps"][:, xa.p_h[b]:xa.p_h[b] + n8], xb["dq"][b], de[xa.k_h[b]:xa.k_h[b] + n8], 1.0, nk))   # sample's nk rows are written
        # one contributor per element, every element of dE written exactly once by plain stores (split_k −2: no zero fill, no atomics)
        # — dE feeds the encoder's bf16 backward: independent of the order workgroups finish in
        ops.gemm_tn_grouped(items, split_k=-2)
        if dxe.dtype == torch.float32:
            dxe.copy_(de[:Me])
        else:
            ops.cast(de[:Me].view(-1), dxe.view(-1))

    # ------------------------------------------------------------------------------------------
    # the decoder as independent chains of samples
    # ------------------------------------------------------------------------------------------
    def _dec_chains(self, B, T, p, seed, xa, xb, rag, kb, kt, ckw):
        """The decoder of a training step is ≈400 launches on B·T ≈ 128 rows: each kernel is a few µs of launch ramp, dependent
        round trips and drain, and on one stream they run strictly one after the other.  Nothing in the decoder couples two samples
        (self- and cross-attention are per sample, everything else per row), so the batch can be cut into `LAKO_DEC_CHAINS` runs of
        samples whose layer-by-layer chains are enqueued on separate HIP streams; the weight gradients — the one place where the samples
        meet — are summed over ALL rows in grouped launches after the last layer.  MEASURED NEGATIVE (round 4, profiles/r04e_decoder_chains.txt):
        1 chain 40.95 ms per step, 2 chains 42.0, 4 chains 47.8 — kernels of two queues do overlap (each takes 1.5 × as long) but the
        pair is no faster than the two in sequence, and every extra launch costs its boundary.  Default 1 = the single-stream schedule;
        the split stays selectable and tested (results: forward values unchanged bit for bit — row-wise kernels; each chain draws its
        dropout masks from its own seed)."""
        n = max(1, min(int(os.environ.get("LAKO_DEC_CHAINS", "1")), B // 2)) if B >= 4 else 1
        chains = []
        for c in range(n):
            b0, b1 = c * B // n, (c + 1) * B // n
            seed_c = (seed ^ (0x9E3779B9 * c)) & 0xFFFFFFFF
            ch = dict(c=c, b0=b0, b1=b1, B=b1 - b0, r0=b0 * T, r1=b1 * T, xa=None, xb=None, kv_b=kb, kv_t=kt, kv_r=None, ckw=ckw,
                      dr=(lambda site, sd=seed_c: (p, sd, site)) if p > 0 else (lambda site: None))
            if xa is not None:
                ch["xa"] = self._xattn_subplans(xa, n, B)[c] if n > 1 else xa
                ch["xb"] = dict(xb, dq=xb["dq"][b0:b1], st=xb["st"][:, b0:b1], c=xb["c"][:, :, b0:b1]) if n > 1 else xb
            elif n > 1:
                if rag is None:        # padded keys: the chain's samples are rows [b0·S, b1·S) of the projected K / V
                    ch.update(kv_b=b1 - b0, kv_r=(b0 * kt, b1 * kt), ckw=dict(ckw, key_mask=ckw["key_mask"][b0:b1]))
                else:                  # packed keys: the whole buffer, the chain's sample offsets
                    ch["ckw"] = dict(ckw, k_off=ckw["k_off"][b0:b1 + 1])
            chains.append(ch)
        return chains

    def _run_chains(self, chains, n_layers, layer_fn, reverse=False):
        """layer_fn(i, chain) for every layer and chain — on one stream when there is one chain (or no GPU: the CPU test double), else
        chain c on its own side stream: forked from the current stream before the first layer, joined after the last."""
        order = range(n_layers - 1, -1, -1) if reverse else range(n_layers)
        if len(chains) == 1 or self.device.type != "cuda":
            for i in order:
                for ch in chains:
                    layer_fn(i, ch)
            return
        if getattr(self, "_chain_streams", None) is None or len(self._chain_streams) < len(chains):
            self._chain_streams = [torch.cuda.Stream(self.device) for _ in chains]
            self._chain_events = [torch.cuda.Event() for _ in range(len(chains) + 1)]
        main = torch.cuda.current_stream(self.device)
        fork = self._chain_events[-1]
        fork.record(main)
        for st in self._chain_streams[:len(chains)]:
            st.wait_event(fork)
        for i in order:
            for ch in chains:
                with torch.cuda.stream(self._chain_streams[ch["c"]]):
                    layer_fn(i, ch)
        for ch in chains:
            ev = self._chain_events[ch["c"]]
            ev.record(self._chain_streams[ch["c"]])
            main.wait_event(ev)

    def _rows(self, ws, name, ch, shape, dtype=None):
        """rows [r0, r1) of the [B·T, …] workspace tensor `name` (allocated at full size)"""
        return self._buf(ws, name, shape, dtype)[ch["r0"]:ch["r1"]]

    def _cross_kv(self, kv, ch, col0):
        """[kb, kt, H, dk] head view of the projected cross-attention K or V (columns col0 …) for the chain's samples"""
        if ch["kv_r"] is not None:
            kv = kv[ch["kv_r"][0]:ch["kv_r"][1]]
        return self._heads(kv, ch["kv_b"], ch["kv_t"] if ch["kv_r"] is None else kv.shape[0] // ch["kv_b"], col0)

    def _dec_layer_fwd(self, ws, i, ch, enc_out, kv, rel, T, S):
        """One decoder block (HF5:448-509) on the rows of one chain: RMSNorm → QKV → causal self-attention → O + residual → RMSNorm →
        cross-attention query → cross-attention → O + residual → RMSNorm → FFN + residual."""
        cfg, ops, lw, dr = self.cfg, self.ops, self.dec[i], ch["dr"]
        d, f, inner, H, eps = cfg.d_model, cfg.d_ff, cfg.inner_dim, cfg.num_heads, cfg.layer_norm_epsilon
        Bc, Ld = ch["B"], cfg.num_decoder_layers
        Md = ws[f"d.h{i}"].shape[0]
        R = lambda name, cols, dt=None: self._rows(ws, name, ch, (Md, cols) if cols else (Md,), dt)       # noqa: E731
        h = ws[f"d.h{i}"][ch["r0"]:ch["r1"]]
        if i == 0:
            ops.embed_fwd(ws["d.ids"].view(-1)[ch["r0"]:ch["r1"]], self.shared.w, h, dr(S_DEC_EMBED))
        xn1 = R(f"d.xn1.{i}", d)
        # (the norm in front of a projection is formed inside the projection's kernel where the library can — LAKO_EPI_NORM_A, the decoder's
        # B·T <= 256 rows: one launch instead of two; xn / rstd are still written, the backward reads them.  LAKO_DEC_FUSE_NORM=0: two launches)
        fuse_norm = os.environ.get("LAKO_DEC_FUSE_NORM", "1") != "0"
        qkv = R(f"d.qkv.{i}", 3 * inner)
        if fuse_norm:
            ops.gemm_nt(h, lw["qkv"].w, qkv, norm=(lw["ln1"].p, eps, xn1, R(f"d.rs1.{i}", 0, torch.float32)))
        else:
            ops.rmsnorm_fwd(h, lw["ln1"].p, xn1, R(f"d.rs1.{i}", 0, torch.float32), eps)
            ops.gemm_nt(xn1, lw["qkv"].w, qkv)
        c1 = R(f"d.ctx.{i}", inner)
        st = self._buf(ws, f"d.st.{i}", (Md // T, H, T, 4), torch.float32)[ch["b0"]:ch["b1"]]
        ops.attn_fwd(self._heads(qkv, Bc, T, 0), self._heads(qkv, Bc, T, inner), self._heads(qkv, Bc, T, 2 * inner),
                     self._heads(c1, Bc, T, 0), st, rel_bias=rel, rel_off=T - 1, causal=True, drop=dr(_dec_site(i, 0)))
        h1 = R(f"d.h1.{i}", d)
        ops.gemm_nt(c1, lw["o"].w, h1, resid=h, drop=dr(_dec_site(i, 1)))
        xn2 = R(f"d.xn2.{i}", d)
        qc = R(f"d.qc.{i}", inner)
        if fuse_norm:
            ops.gemm_nt(h1, lw["cq"].w, qc, norm=(lw["ln2"].p, eps, xn2, R(f"d.rs2.{i}", 0, torch.float32)))
        else:
            ops.rmsnorm_fwd(h1, lw["ln2"].p, xn2, R(f"d.rs2.{i}", 0, torch.float32), eps)
            ops.gemm_nt(xn2, lw["cq"].w, qc)
        c2 = R(f"d.cctx.{i}", inner)
        if ch["xa"] is not None:
            self._xattn_fwd(ch["xb"], ch["xa"], i, qc, enc_out, c2, Bc, T, S, dr(_dec_site(i, 2)))
        else:
            cst = self._buf(ws, f"d.cst.{i}", (Md // T, H, T, 4), torch.float32)[ch["b0"]:ch["b1"]]
            ops.attn_fwd(self._heads(qc, Bc, T, 0), self._cross_kv(kv, ch, 2 * i * inner), self._cross_kv(kv, ch, (2 * i + 1) * inner),
                         self._heads(c2, Bc, T, 0), cst, drop=dr(_dec_site(i, 2)), **ch["ckw"])
        h2 = R(f"d.h2.{i}", d)
        ops.gemm_nt(c2, lw["co"].w, h2, resid=h1, drop=dr(_dec_site(i, 3)))
        xn3 = R(f"d.xn3.{i}", d)
        a1 = R(f"d.a1.{i}", f)
        if fuse_norm:
            ops.gemm_nt(h2, lw["wi"].w, a1, relu=True, drop=dr(_dec_site(i, 4)), norm=(lw["ln3"].p, eps, xn3, R(f"d.rs3.{i}", 0, torch.float32)))
        else:
            ops.rmsnorm_fwd(h2, lw["ln3"].p, xn3, R(f"d.rs3.{i}", 0, torch.float32), eps)
            ops.gemm_nt(xn3, lw["wi"].w, a1, relu=True, drop=dr(_dec_site(i, 4)))
        h_out = ws[f"d.h{i + 1}"][ch["r0"]:ch["r1"]]
        ops.gemm_nt(a1, lw["wo"].w, h_out, resid=h2, drop=dr(_dec_site(i, 5)))
        if i == Ld - 1:
            ops.rmsnorm_fwd(h_out, self.dec_final.p, ws["d.out"][ch["r0"]:ch["r1"]], ws["d.rsf"][ch["r0"]:ch["r1"]], eps, dr(S_DEC_FINAL))

    def _dec_layer_bwd(self, ws, tmp, i, ch, dh_all, drel, kv, dkv, fused, T, S, dw_all):
        """Backward of _dec_layer_fwd on the rows of one chain; dh_all [B·T, d] holds the gradient wrt the block's output on entry and
        wrt its input on exit.  With dropout on, the six weight-gradient problems of the block are NOT launched here: their operands
        (the dropped residual-branch gradients, dpre, dqc, dqkv) live in per-layer scratch, and the chain that handles the batch's last
        rows appends the FULL-batch problems to `dw_all` — one grouped launch over all rows and layers after the chains have joined."""
        cfg, ops, lw, dr = self.cfg, self.ops, self.dec[i], ch["dr"]
        d, f, inner, H = cfg.d_model, cfg.d_ff, cfg.inner_dim, cfg.num_heads
        Bc, r0, r1 = ch["B"], ch["r0"], ch["r1"]
        Md, p, Ld = dh_all.shape[0], (0.0 if dr(0) is None else dr(0)[0]), cfg.num_decoder_layers
        sl = lambda t: t[r0:r1]                                                                   # noqa: E731
        R = lambda name, cols: self._rows(tmp, name, ch, (Md, cols))                              # noqa: E731
        dh = sl(dh_all)
        defer = p > 0 and os.environ.get("LAKO_DEC_DEFER_DW", "1") != "0"      # 0: launch them inside the chain, on its rows (A/B)
        last = r1 == Md          # (this chain's call is the one that registers the full-batch weight-gradient problems)
        if i == Ld - 1:          # final norm of the decoder: dout (gradient wrt the normed, dropped output) → dh
            ops.rmsnorm_bwd(sl(tmp["d.dxn"]), sl(ws[f"d.h{Ld}"]), self.dec_final.p, sl(ws["d.rsf"]), None, dh, self.dec_final.g, dr(S_DEC_FINAL),
                            **self._nxt((R(f"d.dy.ffn.{i}", d), dr(_dec_site(i, 5))) if fused else None))

        def wgrad(name, dy_name, dy_c, x_all, g):
            """dW += dyᵀ·x: deferred to the grouped launch (full-batch operands) or launched on the chain's rows"""
            if defer:
                if last:
                    dw_all.append((tmp[dy_name], x_all, g, 1.0))
            else:
                ops.gemm_tn(dy_c, sl(x_all), g)

        # ---- FFN ------------------------------------------------------------------------------
        if p > 0:
            dy = R(f"d.dy.ffn.{i}", d)
            if not fused:
                ops.dropout_apply(dh, dy, dr(_dec_site(i, 5)))
        else:
            dy = dh
        wgrad("wo", f"d.dy.ffn.{i}", dy, ws[f"d.a1.{i}"], lw["wo"].g)
        dpre = R(f"d.dpre.{i}" if defer else "d.dpre", f)
        ops.gemm_nt(dy, lw["wo"].wt, dpre, aux=sl(ws[f"d.a1.{i}"]), aux_scale=1.0 / (1.0 - p))
        wgrad("wi", f"d.dpre.{i}", dpre, ws[f"d.xn3.{i}"], lw["wi"].g)
        dxn = R("d.dxn", d)
        ops.gemm_nt(dpre, lw["wi"].wt, dxn)
        dy_c = R(f"d.dy.c.{i}", d) if p > 0 else None
        ops.rmsnorm_bwd(dxn, sl(ws[f"d.h2.{i}"]), lw["ln3"].p, sl(ws[f"d.rs3.{i}"]), dh, dh, lw["ln3"].g,
                        **self._nxt((dy_c, dr(_dec_site(i, 3))) if fused else None))
        # ---- cross-attention ---------------------------------------------------------------------
        dy = dh
        if p > 0:
            dy = dy_c
            if not fused:
                ops.dropout_apply(dh, dy, dr(_dec_site(i, 3)))
        wgrad("co", f"d.dy.c.{i}", dy, ws[f"d.cctx.{i}"], lw["co"].g)
        hb = ch.get("hb") if ch["xa"] is not None else None      # the Wk / Wv gradient products of all layers in one launch (backward())
        dctx = R(f"d.xdctx.{i}" if hb is not None else "d.dctx", inner)
        ops.gemm_nt(dy, lw["co"].wt, dctx)
        dqc = R(f"d.dqc.{i}" if defer else "d.dqc", inner)
        qc = sl(ws[f"d.qc.{i}"])
        if ch["xa"] is not None:
            self._xattn_bwd(ch["xb"], ch["xa"], tmp, i, qc, ws["e.out"], dctx, dqc, Bc, T, S, dr(_dec_site(i, 2)), b0=ch["b0"],
                            B_all=Md // T, hb=hb)
        else:
            c2 = sl(ws[f"d.cctx.{i}"])
            ops.attn_bwd(self._heads(qc, Bc, T, 0), self._cross_kv(kv, ch, 2 * i * inner), self._cross_kv(kv, ch, (2 * i + 1) * inner),
                         self._heads(c2, Bc, T, 0), self._heads(dctx, Bc, T, 0), ws[f"d.cst.{i}"][ch["b0"]:ch["b1"]],
                         self._heads(dqc, Bc, T, 0), self._cross_kv(dkv, ch, 2 * i * inner), self._cross_kv(dkv, ch, (2 * i + 1) * inner),
                         drop=dr(_dec_site(i, 2)), **ch["ckw"])
        wgrad("cq", f"d.dqc.{i}", dqc, ws[f"d.xn2.{i}"], lw["cq"].g)
        ops.gemm_nt(dqc, lw["cq"].wt, dxn)
        dy_s = R(f"d.dy.{i}", d) if p > 0 else None
        ops.rmsnorm_bwd(dxn, sl(ws[f"d.h1.{i}"]), lw["ln2"].p, sl(ws[f"d.rs2.{i}"]), dh, dh, lw["ln2"].g,
                        **self._nxt((dy_s, dr(_dec_site(i, 1))) if fused else None))
        # ---- causal self-attention ---------------------------------------------------------------
        dy = dh
        if p > 0:
            dy = dy_s
            if not fused:
                ops.dropout_apply(dh, dy, dr(_dec_site(i, 1)))
        wgrad("o", f"d.dy.{i}", dy, ws[f"d.ctx.{i}"], lw["o"].g)
        dctx = R("d.dctx", inner)      # (not the cross-attention's buffer: that one may be an operand of the deferred Wv-gradient launch)
        ops.gemm_nt(dy, lw["o"].wt, dctx)
        qkv, c1 = sl(ws[f"d.qkv.{i}"]), sl(ws[f"d.ctx.{i}"])
        dqkv = R(f"d.dqkv.{i}" if defer else "d.dqkv", 3 * inner)
        ops.attn_bwd(self._heads(qkv, Bc, T, 0), self._heads(qkv, Bc, T, inner), self._heads(qkv, Bc, T, 2 * inner),
                     self._heads(c1, Bc, T, 0), self._heads(dctx, Bc, T, 0), ws[f"d.st.{i}"][ch["b0"]:ch["b1"]],
                     self._heads(dqkv, Bc, T, 0), self._heads(dqkv, Bc, T, inner), self._heads(dqkv, Bc, T, 2 * inner),
                     rel_bias=ws["d.rel"], drel=drel, rel_off=T - 1, causal=True, drop=dr(_dec_site(i, 0)))
        wgrad("qkv", f"d.dqkv.{i}", dqkv, ws[f"d.xn1.{i}"], lw["qkv"].g)
        ops.gemm_nt(dqkv, lw["qkv"].wt, dxn)
        nxt = None
        if fused and i > 0:
            nxt = (R(f"d.dy.ffn.{i - 1}", d), dr(_dec_site(i - 1, 5)))
        ops.rmsnorm_bwd(dxn, sl(ws[f"d.h{i}"]), lw["ln1"].p, sl(ws[f"d.rs1.{i}"]), dh, dh, lw["ln1"].g, **self._nxt(nxt))
        if i == 0:
            self._embed_bwd(ws["d.ids"].view(-1)[r0:r1], dh, dr(S_DEC_EMBED))

    def forward_loss(self, input_ids, attention_mask, labels, training: bool, lengths=None):
        """input_ids/attention_mask [B,N,L], labels [B,T] (−100 = ignore) → (loss 0-d fp32, logits [B,T,V] fp32).
        `lengths`: optional host-side valid lengths per passage (see _ragged_batch)."""
        if self.shadows_stale:
            self.refresh_shadows()
        cfg, ops = self.cfg, self.ops
        B, N, L = input_ids.shape
        T = labels.shape[1]
        d, f, inner, H, V = cfg.d_model, cfg.d_ff, cfg.inner_dim, cfg.num_heads, cfg.vocab_size
        p = float(cfg.dropout_rate) if training else 0.0
        seed = (self.seed_base * 1000003 + self.step_count) & 0xFFFFFFFF
        ws = self._workspace(("train", B, N, L, T))
        ctx = _Ctx(B=B, N=N, L=L, T=T, p=p, seed=seed, ws=ws)
        ctx.ids = input_ids.reshape(-1).contiguous()
        ctx.mask_u8 = attention_mask.reshape(B * N, L).to(torch.uint8).contiguous()
        ctx.labels = labels.reshape(-1).contiguous()
        dr = (lambda site: (p, seed, site)) if p > 0 else (lambda site: None)
        eps = cfg.layer_norm_epsilon
        ctx.ckpt = bool(self.use_checkpoint and training)
        rag = ctx.rag = self._legacy_layout(self._ragged_batch(attention_mask, B, N, L, lengths), T)
        self._row_cap = {rag.M: B * N * L} if rag is not None else {}
        enc_ids = ctx.ids if rag is None else self._packed_ids(ws, ctx.ids, rag, L)
        ctx.enc_ids = enc_ids
        xa = ctx.xa = self._xattn_plan(rag, B, N, L, T * H)
        self.xattn_active = xa is not None         # (bench.py / tests: which formulation the last forward ran)
        ctx.enc_mask = self._enc_mask(rag, ctx.mask_u8)
        enc_out, kv = self._encode(ws, enc_ids, ctx.enc_mask, B * N, L, p, seed, save="ckpt" if ctx.ckpt else True, rag=rag,
                                   want_kv=xa is None)
        # ---- decoder ------------------------------------------------------------------------
        Md, S, Ld = B * T, N * L, cfg.num_decoder_layers
        if rag is None:      # cross-attention keys: [B, N·L] with the padding mask, or each sample's packed valid tokens
            kb, kt, ckw = B, S, dict(key_mask=ctx.mask_u8.view(B, S))
        else:
            kb, kt, ckw = 1, rag.M, dict(k_off=rag.soff, max_k=S)
        dec_ids = self._buf(ws, "d.ids", (B, T), torch.int64)
        ops.shift_right(labels.contiguous(), dec_ids)
        ctx.dec_ids = dec_ids
        rel = self._buf(ws, "d.rel", (H, 2 * T - 1), torch.float32)
        ops.relpos_expand(self.dec_rel.p, self._lut(T, T, False), rel)
        if self.dec_xrel is not None:      # (_legacy_layout)
            xrel = self._buf(ws, "d.xrel", (H, 2 * T - 1), torch.float32)
            ops.relpos_expand(self.dec_xrel.p, self._lut(T, T, False), xrel)
            ckw = dict(ckw, rel_bias=xrel, rel_off=T - 1)
        xb = self._xattn_buffers(ws, B, N, L, T, Ld, xa.splits) if xa is not None else None
        for i in range(Ld + 1):
            self._buf(ws, f"d.h{i}", (Md, d))
        dec_out = self._buf(ws, "d.out", (Md, d))
        self._buf(ws, "d.rsf", (Md,), torch.float32)
        chains = ctx.chains = self._dec_chains(B, T, p, seed, xa, xb, rag, kb, kt, ckw)
        # (embedding → blocks → final norm per chain of samples; the LM head and the loss see all rows again)
        self._run_chains(chains, Ld, lambda i, ch: self._dec_layer_fwd(ws, i, ch, enc_out, kv, rel, T, S))
        logits = self._buf(ws, "d.logits", (Md, V), torch.float32)
        ops.gemm_nt(dec_out, self.shared.w, logits, alpha=d ** -0.5)       # tied LM head × d_model^-0.5
        loss_buf = self._buf(ws, "d.loss", (2,), torch.float32)
        ops.ce_fwd_bwd(logits, ctx.labels, loss_buf, None)
        self.ctx = ctx
        return loss_buf[0], logits.view(B, T, V)

    # ------------------------------------------------------------------------------------------
    # backward
    # ------------------------------------------------------------------------------------------
    def _ready(self, first_block: str, last_block: str):
        if self.grad_hook is not None:
            a, b = self.by_name[first_block], self.by_name[last_block]
            self.grad_hook(a.off, b.off + -(-b.numel // ALIGN) * ALIGN)

    def _ffn_bwd(self, lw, dh, a1, xn, h_in, rs, ln, p, drop_out, tmp, dw=None, dy_pre=None, nxt=None, role="d", slot=""):
        """residual FFN sublayer backward; dh is updated in place to the gradient wrt the sublayer input.
        `dw`: list collecting the weight-gradient problems for one grouped launch (encoder layers with dropout: their
        operands live in per-layer scratch that stays untouched until the layer's backward is complete).
        `dy_pre`: dropout_bwd(dh) already written by the RMSNorm backward that produced dh; `nxt` = (buffer, dropout) asks
        this sublayer's RMSNorm backward to do the same for the next consumer (saves a pass over the residual gradient).
        `role` ("d" decoder / "e" encoder) names the scratch: by ROLE, never by row count — the unpadded encoder has a
        different row count for almost every batch, and a size in the name would leave one dead buffer set per batch."""
        ops = self.ops
        M, d = dh.shape
        f = a1.shape[1]
        dy = dh
        if p > 0 and dy_pre is not None:
            dy = dy_pre
        elif p > 0:
            dy = self._buf(tmp, f"{role}.dy.ffn{slot}" if dw is not None else f"{role}.dy", (M, d))
            ops.dropout_apply(dh, dy, drop_out)
        if dw is not None and p > 0:
            dw.append((dy, a1, lw["wo"].g, 1.0))
        else:
            ops.gemm_tn(dy, a1, lw["wo"].g)
        dpre = self._buf(tmp, f"{role}.dpre{slot}", (M, f))
        ops.gemm_nt(dy, lw["wo"].wt, dpre, aux=a1, aux_scale=1.0 / (1.0 - p))   # ∘ [relu'>0] ∘ dropout
        if dw is not None:
            dw.append((dpre, xn, lw["wi"].g, 1.0))
        else:
            ops.gemm_tn(dpre, xn, lw["wi"].g)
        dxn = self._buf(tmp, f"{role}.dxn", (M, d))
        ops.gemm_nt(dpre, lw["wi"].wt, dxn)
        ops.rmsnorm_bwd(dxn, h_in, ln.p, rs, dh, dh, ln.g, **self._nxt(nxt))

    @staticmethod
    def _nxt(nxt):
        return {} if nxt is None else dict(dx_drop=nxt[0], drop_out=nxt[1])

    def backward(self, upstream=None):
        """Accumulate d(loss)/d(params) into G.  `upstream`: optional device scalar tensor multiplying the loss."""
        ctx, cfg, ops = self.ctx, self.cfg, self.ops
        if ctx is None:
            raise RuntimeError("backward() without a preceding forward_loss()")
        ws, p, seed = ctx.ws, ctx.p, ctx.seed
        B, N, L, T = ctx.B, ctx.N, ctx.L, ctx.T
        d, f, inner, H, V = cfg.d_model, cfg.d_ff, cfg.inner_dim, cfg.num_heads, cfg.vocab_size
        rag = ctx.rag
        self._row_cap = {rag.M: B * N * L} if rag is not None else {}
        Md, Me, S, Ld, Le = B * T, (rag.M if rag is not None else B * N * L), N * L, cfg.num_decoder_layers, cfg.num_layers
        dr = (lambda site: (p, seed, site)) if p > 0 else (lambda site: None)
        tmp = self._workspace(("bwd", B, N, L, T))
        if rag is None:
            kb, kt, ckw = B, S, dict(key_mask=ctx.mask_u8.view(B, S))
        else:
            kb, kt, ckw = 1, rag.M, dict(k_off=rag.soff, max_k=S)
        # ---- loss + LM head ------------------------------------------------------------------
        dlog = self._buf(tmp, "dlogits", (Md, V))
        ops.ce_fwd_bwd(ws["d.logits"], ctx.labels, self._buf(tmp, "loss2", (2,), torch.float32), dlog, upstream)
        alpha = d ** -0.5
        dh = self._buf(tmp, "d.dh", (Md, d))
        dout = self._buf(tmp, "d.dxn", (Md, d))
        # K = vocabulary (32128) for an [Md, d] output of a handful of tiles: accumulate in fp32 with atomics so that the
        # library may split K over the idle CUs (gemm_nt_ring_kernel), then round once to the compute dtype
        dout32 = dout if dout.dtype == torch.float32 else self._buf(tmp, "d.dout32", (Md, d), torch.float32)
        if self.det:      # LAKO_DETERMINISTIC=1: one workgroup per output tile walks the whole vocabulary (no K-split, no atomics)
            ops.gemm_nt(dlog, self.shared.wt, dout, alpha=alpha)
        else:
            ops.zero_(dout32)
            ops.gemm_nt(dlog, self.shared.wt, dout32, alpha=alpha, atomic=True)
            if dout32 is not dout:
                ops.cast(dout32.view(-1), dout.view(-1))
        # (K = B·T rows, 378 tiles: one K-split; nothing else touches the embedding gradient until the embedding backward kernels later on
        # this stream: plain read-modify-write instead of 99 MB of float atomics)
        # (and the FIRST writer of a freshly zeroed gradient buffer overwrites: no read at all.  A second backward without zero_grad —
        # gradient accumulation — adds)
        one = -2 if self._g_fresh else -1
        self._g_fresh = False
        ops.gemm_tn(dlog, ws["d.out"], self.shared.g, alpha=alpha, split_k=one)
        # every RMSNorm backward also writes dropout_bwd(dx) for the residual branch that consumes dx next (fused=True)
        fused = p > 0 and os.environ.get("LAKO_FUSE_DROP", "1") != "0"   # 0: separate dropout_apply launches (A/B)
        xa = ctx.xa
        xb = self._xattn_buffers(ws, B, N, L, T, Ld, xa.splits) if xa is not None else None
        dkv = kv = None
        if xa is None:
            dkv = self._buf(tmp, "dkv", (Me, self.kv_all.w.shape[0]))
            kv = ws["e.kv"]
        drel = self._buf(tmp, "d.drel", (H, 2 * T - 1), torch.float32)
        ops.zero_(drel)
        dxrel = None
        if self.dec_xrel is not None:      # (the chains' key arguments were built in forward_loss without the gradient table)
            dxrel = self._buf(tmp, "d.dxrel", (H, 2 * T - 1), torch.float32)
            ops.zero_(dxrel)
            for ch in ctx.chains:
                ch["ckw"] = dict(ch["ckw"], drel=dxrel)
        dw_all = []      # every decoder weight gradient of the step (K = B·T rows): grouped launches after the last layer
        # the 2 · Ld head-batched Wk / Wv gradient products likewise: ONE launch after the last layer (each alone is its dispatch: 24 × 8 µs);
        # with one chain only (LAKO_DEC_HB_DEFER=0: inside the layers)
        hb_all = [] if (xa is not None and len(ctx.chains) == 1 and os.environ.get("LAKO_DEC_HB_DEFER", "1") != "0") else None
        for ch in ctx.chains:
            ch["hb"] = hb_all
        self._run_chains(ctx.chains, Ld, lambda i, ch: self._dec_layer_bwd(ws, tmp, i, ch, dh, drel, kv, dkv, fused, T, S, dw_all),
                         reverse=True)
        if hb_all:
            ops.headbatch_tn_multi(hb_all)
        if dw_all:      # (K = B·T rows: one K-split; the chains have joined and nothing else touches these gradients: plain read-modify-write)
            ops.gemm_tn_grouped(dw_all, split_k=one)
        ops.relpos_reduce(drel, self._lut(T, T, False), self.dec_rel.g)
        if dxrel is not None:
            ops.relpos_reduce(dxrel, self._lut(T, T, False), self.dec_xrel.g)
            for ch in ctx.chains:
                ch["ckw"] = {k: v for k, v in ch["ckw"].items() if k != "drel"}
        # ---- cross K/V projection of all decoder layers -----------------------------------------
        deh = self._buf(tmp, "e.dh", (Me, d))
        dxe = self._buf(tmp, "e.dxn", (Me, d))
        if xa is not None:
            self._xattn_denc(xb, xa, tmp, Me, dxe)
        else:
            ops.gemm_tn(dkv, ws["e.out"], self.kv_all.g)
            ops.gemm_nt(dkv, self.kv_all.wt, dxe)
        # WEIGHT GRADIENTS OF SEVERAL ENCODER LAYERS IN ONE LAUNCH (round 4).  A layer's four dW are 108 tiles of 256²; with two K-splits
        # that is 216 workgroups on 256 CUs — every layer's launch leaves 16 % of the chip idle, and nothing can fill it (one workgroup
        # per CU: 128 KiB of LDS).  The products do not have to run where their operands are produced: with the operands of `grp`
        # layers kept in per-slot scratch (dy, dpre, dqkv: 0.66 GB per layer at config 2) their 108·grp tiles × splits go out as ONE grid
        # that the dispatcher deals over the CUs as they free up — 12 layers × 2 splits = 2 592 units = 10.1 rounds of half-K units
        # instead of 12 rounds.  Not with set_checkpoint (the forward operands of a layer are recomputed into ONE slot), not when a
        # data-parallel hook wants every layer's gradients as soon as they exist (overlap mode; the deferred all-reduce does not), not
        # without dropout (dy aliases the residual gradient then).
        grp = 1
        if fused and not ctx.ckpt and not (self.grad_hook is not None and self.grad_ranges_early):
            grp = max(1, min(int(os.environ.get("LAKO_ENC_DW_GROUP", "12")), Le, 12))
        # K-splits of the grouped launch: 3 (1 296 tiles × 3 = 15.2 rounds of third-K units; measured per step, rocprofv3, 14 steps:
        # one launch per layer 8.70 ms, 12 layers × 2 / 3 / 4 splits 8.35 / 8.32 / 8.30 ms, 6 layers × 3 splits 8.35 — the library's
        # own cost model, built for one-round launches, would pick 1 split = 6 whole rounds: the slowest, 41.3 ms per step)
        # −3: the hybrid schedule of lako_gemm_tn_grouped — whole rounds of full-K tiles + a finely cut remainder (round 4: 39.8 → 39.1 ms per step
        # against 3 uniform K-splits; a positive value forces that many uniform splits)
        enc_split = int(os.environ.get("LAKO_ENC_DW_SPLIT", "-3"))
        slot_of = (lambda i: f".{i % grp}") if grp > 1 else (lambda i: "")
        dy_f = self._buf(tmp, "e.dy.ffn" + slot_of(Le - 1), (Me, d)) if fused else None
        ops.rmsnorm_bwd(dxe, ws[f"e.h{Le}"], self.enc_final.p, ws["e.rsf"], None, deh, self.enc_final.g, dr(S_ENC_FINAL),
                        **self._nxt((dy_f, dr(_enc_site(Le - 1, 3))) if fused else None))
        self._ready("dec.final_ln", "enc.final_ln")
        # ---- encoder ---------------------------------------------------------------------------
        BN = B * N
        drel_e = self._buf(tmp, "e.drel", (H, 2 * L - 1), torch.float32)
        ops.zero_(drel_e)
        dw, dw_layers = [], []
        for i in reversed(range(Le)):
            lw = self.enc[i]
            j, sl = i, slot_of(i)
            if ctx.ckpt:   # recompute this block's intermediates from its saved input (same seeds → same dropout masks)
                j = 0
                self._enc_layer_fwd(ws, i, 0, ws[f"e.h{i}"], None, BN, L, ctx.enc_mask, ws["e.rel"], dr, rag)
            # the layer's four weight gradients (K = all tokens, small M×N) go out in ONE grouped launch (with those of the other
            # layers of its group): 108 tiles per layer fill the chip with ~2 K-splits, where four separate launches need 7–28 splits
            # each and pay one fp32 atomic pass over the output per split
            dy_s = self._buf(tmp, "e.dy" + sl, (Me, d)) if fused else None
            self._ffn_bwd(lw, deh, ws[f"e.a1.{j}"], ws[f"e.xn2.{j}"], ws[f"e.h1.{j}"], ws[f"e.rs2.{j}"], lw["ln2"], p,
                          dr(_enc_site(i, 3)), tmp, dw, dy_pre=dy_f, nxt=(dy_s, dr(_enc_site(i, 1))) if fused else None, role="e", slot=sl)
            dy = deh
            if p > 0:
                dy = dy_s if fused else self._buf(tmp, "e.dy", (Me, d))
                if not fused:
                    ops.dropout_apply(deh, dy, dr(_enc_site(i, 1)))
                dw.append((dy, ws[f"e.ctx.{j}"], lw["o"].g, 1.0))
            else:
                ops.gemm_tn(dy, ws[f"e.ctx.{j}"], lw["o"].g)     # dy aliases deh, which the next ops rewrite
            dctx = self._buf(tmp, "e.dctx", (Me, inner))
            ops.gemm_nt(dy, lw["o"].wt, dctx)
            qkv = ws[f"e.qkv.{j}"]
            dqkv = self._buf(tmp, "e.dqkv" + sl, (Me, 3 * inner))
            hb, ht, akw = self._enc_attn_layout(rag, BN, L, ctx.enc_mask)
            ops.attn_bwd(self._heads(qkv, hb, ht, 0), self._heads(qkv, hb, ht, inner), self._heads(qkv, hb, ht, 2 * inner),
                         self._heads(ws[f"e.ctx.{j}"], hb, ht, 0), self._heads(dctx, hb, ht, 0), ws[f"e.st.{j}"],
                         self._heads(dqkv, hb, ht, 0), self._heads(dqkv, hb, ht, inner),
                         self._heads(dqkv, hb, ht, 2 * inner), rel_bias=ws["e.rel"], drel=drel_e, rel_off=L - 1,
                         drop=dr(_enc_site(i, 0)), **akw)
            dw.append((dqkv, ws[f"e.xn1.{j}"], lw["qkv"].g, 1.0))
            ops.gemm_nt(dqkv, lw["qkv"].wt, dxe)
            dw_layers.append(i)
            if i % grp == 0:      # the group is complete: its launch goes out before the norm backward below rewrites the slot of layer i − 1
                ops.gemm_tn_grouped(dw, split_k=enc_split if grp > 1 else 0, workspace=self._tn_scratch)
                dw = []
            dy_f = self._buf(tmp, "e.dy.ffn" + slot_of(i - 1), (Me, d)) if fused and i > 0 else None
            ops.rmsnorm_bwd(dxe, ws[f"e.h{i}"], lw["ln1"].p, ws[f"e.rs1.{j}"], deh, deh, lw["ln1"].g,
                            **self._nxt((dy_f, dr(_enc_site(i - 1, 3))) if fused and i > 0 else None))
            if i % grp == 0:
                for li in dw_layers:
                    self._ready(f"enc.{li}.qkv", f"enc.{li}.ln2")
                dw_layers = []
        self._embed_bwd(ctx.enc_ids, deh, dr(S_ENC_EMBED))
        ops.relpos_reduce(drel_e, self._lut(L, L, True), self.enc_rel.g)
        self._ready("enc.rel", "shared")
        self.step_count += 1   # next forward draws fresh dropout masks
        self.ctx = None

    # ------------------------------------------------------------------------------------------
    # greedy decode  (HF generate: num_beams=1, do_sample=False, use_cache=True)
    # ------------------------------------------------------------------------------------------
    def generate(self, input_ids, attention_mask, max_length: int, capture_scores: bool = False, lengths=None):
        if self.shadows_stale:
            self.refresh_shadows()
        cfg, ops = self.cfg, self.ops
        B, N, L = input_ids.shape
        d, f, inner, H, V = cfg.d_model, cfg.d_ff, cfg.inner_dim, cfg.num_heads, cfg.vocab_size
        S, Ld, eps, dk = N * L, cfg.num_decoder_layers, cfg.layer_norm_epsilon, cfg.d_kv
        ws = self._workspace(("gen", B, N, L, max_length))
        ids = input_ids.reshape(-1).contiguous()
        mask_u8 = attention_mask.reshape(B * N, L).to(torch.uint8).contiguous()
        # valid tokens only (as in forward_loss) unless the raw scores are captured: those are laid out per padded position
        rag = None if capture_scores else self._legacy_layout(self._ragged_batch(attention_mask, B, N, L, lengths), max_length)
        if capture_scores:
            self._all_valid = False      # (_ragged_batch did not look at this mask: nothing is known about its padding — _enc_mask keeps it)
        self._row_cap = {rag.M: B * N * L} if rag is not None else {}
        # what changes from call to call is COPIED into workspace buffers (stable addresses: the decode steps below are captured
        # into HIP graphs): the samples' key offsets of the unpadded batch, or the key mask of the padded one
        if rag is None:
            kmask = self._buf(ws, "g.kmask", (B, S), torch.uint8)
            kmask.copy_(mask_u8.view(B, S))
            kb, kt, ckw, mode = B, S, dict(key_mask=kmask), "p"
        else:
            koff = self._buf(ws, "g.koff", (B + 1,), torch.int32)
            koff.copy_(rag.soff)
            ids, kb, kt, ckw, mode = self._packed_ids(ws, ids, rag, L), 1, rag.M, dict(k_off=koff, max_k=S), "r"
        # cross-attention in the encoder-state space (R = H query rows per sample and step; no K/V projection of the N·L encoder
        # states): five launches per layer and step instead of one — slower while the host's launch rate bounds the loop (eager:
        # 113 ms against 93 ms for 50 tokens at config 2), faster once a step is one graph launch.  Never when the raw scores are
        # captured: those come out of the projected kernel in the reference's layout.  LAKO_XATTN_GENERATE=0/1 overrides.
        use_graph = self.device.type == "cuda" and not capture_scores and getattr(ops, "probe", None) is None and \
            os.environ.get("LAKO_GEN_GRAPH", "1") != "0"
        use_x = os.environ.get("LAKO_XATTN_GENERATE", "1" if use_graph else "0") == "1" and not capture_scores
        xa = self._xattn_plan(rag, B, N, L, H) if use_x else None
        self.xattn_active = xa is not None
        enc_out, kv = self._encode(ws, ids, self._enc_mask(rag, mask_u8), B * N, L, 0.0, 0, save=False, rag=rag, want_kv=xa is None)
        if xa is not None:
            mode += "x"
            cap = B * (-(-S // 256) * 256)
            xoff = self._buf(ws, "g.xoff", (2, B + 1), torch.int32)
            xoff[0].copy_(xa.k_off)
            xoff[1].copy_(xa.p_off)
            xq = self._buf(ws, "g.xq", (B, H, d))
            xs = self._buf(ws, "g.xs", (H, cap), torch.float32)      # launched over all `cap` columns: tiles past the batch's
            xp = self._buf(ws, "g.xp", (H, cap))                      # last key column exit at once
            xst = self._buf(ws, "g.xst", (B, H, 2), torch.float32)
            xc = self._buf(ws, "g.xc", (xa.splits, B, H, d), torch.float32)
            # scores, softmax and context of a decode step in one pass over the encoder states when the shapes allow
            fused = dk == 64 and getattr(ops, "xattn_decode_ok", lambda *_: False)(H, d) and os.environ.get("LAKO_XATTN_DECODE", "1") != "0"
            if fused:
                zd = max(1, min(64, 256 // B))
                pml = self._buf(ws, "g.xpml", (zd, B, 16, 2), torch.float32)
                pc = self._buf(ws, "g.xpc", (zd, B, 16, d), torch.float32)
                mode += "f"
        ML = max_length
        seq = self._buf(ws, "g.seq", (B, ML), torch.int64)
        nxt = self._buf(ws, "g.next", (B,), torch.int64)
        done = self._buf(ws, "g.done", (B,), torch.uint8)
        # finished-row count after every step, mirrored to pinned host memory: the host looks at the count of step t − LAG
        # (an event wait on that step only), so it keeps enqueueing LAG steps ahead of the GPU instead of draining the queue
        # after every token; the LAG extra steps a finished batch may run emit pad for every row and are cut off below
        ndone = self._buf(ws, "g.ndone", (ML,), torch.int32)
        if "g.ndone_host" not in ws or ws["g.ndone_host"].numel() < ML:
            ws["g.ndone_host"] = torch.zeros(ML, dtype=torch.int32).pin_memory() if torch.cuda.is_available() else \
                torch.zeros(ML, dtype=torch.int32)
        ndone_host, LAG, events = ws["g.ndone_host"], 2, []
        for t_ in (seq, nxt, done, ndone):
            ops.zero_(t_)                                  # decoder_start_token_id = pad = 0
        rel = self._buf(ws, "g.rel", (H, 2 * ML - 1), torch.float32)
        ops.relpos_expand(self.dec_rel.p, self._lut(ML, ML, False), rel)
        xrel = None
        if self.dec_xrel is not None:
            # legacy_cross_bias (_legacy_layout).  Step t's query sits at position t: table index key − t + ML − 1.  With score capture the
            # reference runs ITS cross-attention forward (src/model.py:286-349), which sizes the bias by the step's own query length —
            # 1 under the key / value cache — so every step sees the bias of position 0: one constant per head (src/model.py:323-325)
            xrel = self._buf(ws, "g.xrel", (H, 2 * ML - 1), torch.float32)
            ops.relpos_expand(self.dec_xrel.p, self._lut(ML, ML, False), xrel)
        # per layer [B, max_length, (q | k | v)]: ONE projection GEMM per step writes the step's q, k and v into row t; q is read
        # back from there, k and v stay as the cache
        cache = [self._buf(ws, f"g.cache.{i}", (B, ML, 3 * inner)) for i in range(Ld)]
        scores = self._buf(ws, "g.scores", (Ld, B, H, 1, S), torch.float32) if capture_scores else None
        h, h1, h2, xn = (self._buf(ws, f"g.{n}", (B, d)) for n in ("h", "h1", "h2", "xn"))
        rs = self._buf(ws, "g.rs", (B,), torch.float32)
        c1, qc, c2 = (self._buf(ws, f"g.{n}", (B, inner)) for n in ("c1", "qc", "c2"))
        st = self._buf(ws, "g.st", (B, H, 1, 4), torch.float32)
        a1 = self._buf(ws, "g.a1", (B, f))
        logits = self._buf(ws, "g.logits", (B, V), torch.float32)
        def step(t):
            """enqueue decode step t (reads nxt, writes seq[:, t + 1], nxt, done, ndone[t])"""
            ops.embed_fwd(nxt, self.shared.w, h)
            fz = os.environ.get("LAKO_DEC_FUSE_NORM", "1") != "0"      # the norm inside the projection's kernel (LAKO_EPI_NORM_A), as in training
            nrm = (lambda w_: dict(norm=(w_, eps, xn, rs))) if fz else None
            xkw = ckw if xrel is None else dict(ckw, rel_bias=xrel, rel_off=ML - 1 - (0 if capture_scores else t))
            for i in range(Ld):
                lw = self.dec[i]
                if fz:
                    ops.gemm_nt(h, lw["qkv"].w, cache[i].view(B, ML * 3 * inner)[:, t * 3 * inner:(t + 1) * 3 * inner], **nrm(lw["ln1"].p))
                else:
                    ops.rmsnorm_fwd(h, lw["ln1"].p, xn, rs, eps)
                    ops.gemm_nt(xn, lw["qkv"].w, cache[i].view(B, ML * 3 * inner)[:, t * 3 * inner:(t + 1) * 3 * inner])
                qs = cache[i][:, t:t + 1, :inner].unflatten(2, (H, dk))
                kc = cache[i][:, :t + 1, inner:2 * inner].unflatten(2, (H, dk))
                vc = cache[i][:, :t + 1, 2 * inner:].unflatten(2, (H, dk))
                ops.attn_fwd(qs, kc, vc, c1.view(B, 1, H, dk), st, rel_bias=rel, rel_off=ML - 1 - t)
                ops.gemm_nt(c1, lw["o"].w, h1, resid=h)
                if fz:
                    ops.gemm_nt(h1, lw["cq"].w, qc, **nrm(lw["ln2"].p))
                else:
                    ops.rmsnorm_fwd(h1, lw["ln2"].p, xn, rs, eps)
                    ops.gemm_nt(xn, lw["cq"].w, qc)
                if xa is None:
                    ops.attn_fwd(qc.view(B, 1, H, dk), self._heads(kv, kb, kt, 2 * i * inner),
                                 self._heads(kv, kb, kt, (2 * i + 1) * inner), c2.view(B, 1, H, dk), st,
                                 scores_out=scores[i] if (capture_scores and t == 0) else None, **xkw)
                else:
                    ops.headbatch_nt(qc.view(B, 1, H, dk), self._xw(i, "k", transposed=True), xq.view(B, 1, H, d))
                    if fused:
                        ops.xattn_decode(xq, enc_out, xoff[0], pml, pc)
                        r0 = (2 * i + 1) * inner
                        ops.xattn_decode_combine(pml, pc, self.kv_all.w[r0:r0 + inner], c2, H)
                    else:
                        ops.xattn_scores(xq, enc_out, xoff[0], xoff[1], cap, xs)
                        ops.xattn_softmax_fwd(xs, xst, xp, xoff[0], xoff[1], 1, H, S)
                        ops.xattn_context(xp, enc_out, xoff[0], xoff[1], xc)
                        ops.headbatch_nt(xc.view(xa.splits, B, 1, H, d), self._xw(i, "v"), c2.view(B, 1, H, dk))
                ops.gemm_nt(c2, lw["co"].w, h2, resid=h1)
                if fz:
                    ops.gemm_nt(h2, lw["wi"].w, a1, relu=True, **nrm(lw["ln3"].p))
                else:
                    ops.rmsnorm_fwd(h2, lw["ln3"].p, xn, rs, eps)
                    ops.gemm_nt(xn, lw["wi"].w, a1, relu=True)
                ops.gemm_nt(a1, lw["wo"].w, h, resid=h2)
            ops.rmsnorm_fwd(h, self.dec_final.p, xn, rs, eps)
            ops.gemm_nt(xn, self.shared.w, logits, alpha=d ** -0.5)
            ops.greedy_step(logits, seq, t + 1, nxt, done, ndone[t:t + 1], cfg.eos_token_id, cfg.pad_token_id)

        # A decode step is ≈130 dependent launches of a few µs each: the host's launch rate bounds the loop.  After one eager call
        # per (shape, mode) — the kernels' one-time setup must not happen under capture — every step position is captured ONCE into a
        # HIP graph (kernel arguments that depend on t are baked in; everything that changes from call to call lives in workspace
        # buffers whose addresses do not: the offsets / key mask are COPIED into them above) and replayed: one launch per token.
        graphs = None
        if use_graph:
            gs = ws.setdefault("g.graphs", {}).get(mode)
            ptrs = {k_: v.data_ptr() for k_, v in ws.items() if k_.startswith("^")}
            if gs is None or any(ptrs.get(k_) != p_ for k_, p_ in gs["ptrs"].items()):      # a workspace buffer moved: start over
                gs = ws["g.graphs"][mode] = dict(ptrs=ptrs, warm=False, steps={}, pool=None)
            if gs["warm"]:
                graphs = gs
            gs["warm"] = True
        n_out = 1
        for t in range(ML - 1):
            if graphs is None:
                step(t)
            else:
                g = graphs["steps"].get(t)
                if g is None:
                    g = torch.cuda.CUDAGraph()
                    if graphs["pool"] is None:
                        graphs["pool"] = torch.cuda.graph_pool_handle()
                    # thread_local: a data-loader / pin-memory thread may touch the runtime while this thread captures
                    with torch.cuda.graph(g, pool=graphs["pool"], capture_error_mode="thread_local"):
                        step(t)
                    graphs["steps"][t] = g
                g.replay()
            n_out = t + 2
            ndone_host[t:t + 1].copy_(ndone[t:t + 1], non_blocking=True)
            if ndone.is_cuda:
                events.append(torch.cuda.Event())
                events[-1].record()
            look = t - LAG if (ndone.is_cuda and t < ML - 2) else t     # the last step is checked at once
            if look >= 0:
                if ndone.is_cuda:
                    events[look].synchronize()
                first = next((u for u in range(max(look - LAG, 0), look + 1) if int(ndone_host[u]) == B), None)
                if first is not None:        # HF stops right after the step in which the last row emitted EOS
                    n_out = first + 2
                    break
        out = seq[:, :n_out].clone()
        if capture_scores:
            # [B, H, n_layers, S] — the layout get_crossattention_scores builds with torch.cat(dim=2), src/model.py:152-160
            return out, scores[:, :, :, 0, :].permute(1, 2, 0, 3).contiguous()
        return out
