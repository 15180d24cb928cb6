"""Tensor-level bindings of the C-ABI (include/lako_hip.h) for torch device tensors.

`HipOps` is the only implementation the product uses.  Every method enqueues one (or two) HIP kernels
on torch's current stream and returns nothing; outputs are caller-allocated tensors.  PyTorch is used
here for device memory and streams only — no torch math runs on this path.

The method set is the engine's whole vocabulary (`lako_amd/engine.py`); `tests/ref_ops.py` implements
the same methods in plain fp32 torch on CPU as a *test double*, used (a) to check each HIP kernel on
the GPU and (b) to check the engine's orchestration against the oracle without a GPU.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import (EPI_ATOMIC, EPI_AUXMASK, EPI_NORM_A, EPI_RELU, EPI_RESID, LAKO_BF16, LAKO_F32, LAKO_FP8_E4M3, AttnBwd, AttnFwd, Dropout,
                   GemmNT, LakoError, NO_DROP, check)


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return LAKO_F32
    if t.dtype == torch.bfloat16:
        return LAKO_BF16
    raise LakoError(f"unsupported dtype {t.dtype}")


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _drop(d) -> Dropout:
    if d is None or d[0] <= 0.0:
        return NO_DROP
    return Dropout(float(d[0]), int(d[1]) & 0xFFFFFFFF, int(d[2]) & 0xFFFFFFFF)


def _rowmajor2d(t: torch.Tensor, name: str):
    if t.dim() != 2 or t.stride(1) != 1:
        raise LakoError(f"{name}: expected a 2-D tensor with unit inner stride, got shape {tuple(t.shape)} "
                        f"strides {t.stride()}")
    return t.shape[0], t.shape[1], t.stride(0)


def _bthd(t: torch.Tensor, name: str):
    """[B, T, H, dk] view with strides (sb, st, dk, 1)."""
    if t.dim() != 4 or t.stride(3) != 1 or t.stride(2) != t.shape[3]:
        raise LakoError(f"{name}: expected [B,T,H,dk] with head-contiguous layout, got {tuple(t.shape)} {t.stride()}")
    return t.stride(0), t.stride(1)


try:
    _raw_stream, _cur_device = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice
except AttributeError:        # a torch build without the private accessors: the public (slow) path
    _raw_stream, _cur_device = (lambda dev: torch.cuda.current_stream(dev).cuda_stream), torch.cuda.current_device


class HipOps:
    """The product's op set: hand-written gfx950 kernels behind the C-ABI."""

    name = "hip"

    def __init__(self):
        if not torch.cuda.is_available():
            raise LakoError("HipOps needs a ROCm device (torch.cuda.is_available() is False); no CPU fallback exists")
        self.lib = _lib.load()
        assert self.lib.lako_version() == _lib.ABI_VERSION
        self.probe = None   # list → every op records (name, algorithmic flops, start event, end event)
        # kernel-selection knobs are THIS object's (the library keeps no tuning state): defaults + the LAKO_TUNING environment
        # string ("key=value,key=value": A/B measurements), changed by set_tuning, handed to every GEMM call
        self.tuning = _lib.Tuning()
        check(self.lib.lako_tuning_init(C.byref(self.tuning)), "lako_tuning_init")
        self._tuning_p = C.addressof(self.tuning)
        # LAKO_DETERMINISTIC=1: the library's shared float sums are order-independent; here the weight-gradient
        # products take one contributor per output element unless the caller names a split (the engine's slab path)
        self.det = bool(self.lib.lako_deterministic())

    @staticmethod
    def _stream():
        # the raw handle of the current stream of the current device in two C calls: torch.cuda.current_stream() builds a Stream
        # object through several Python layers — measured 8.7 µs per launch, ≈6 ms of the ≈12.5 ms it took the host to enqueue one
        # training step of ≈700 launches (tools/host_profile.py)
        return C.c_void_p(_raw_stream(_cur_device()))

    def _timed(self, name, flops, launch):
        """Launch through `launch()`; when a probe list is installed, bracket it with HIP events recorded on the
        stream the kernel runs on (torch's current stream — the one `_stream()` hands to the C-ABI)."""
        if self.probe is None:
            return launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        self.probe.append((name, flops, e0, e1))

    def probe_summary(self):
        """{name: (launches, total ms, total flops)} of the installed probe (call after a device sync)."""
        out = {}
        for name, fl, e0, e1 in self.probe or []:
            n, ms, f = out.get(name, (0, 0.0, 0.0))
            out[name] = (n + 1, ms + e0.elapsed_time(e1), f + fl)
        return out

    # ---- plumbing ---------------------------------------------------------------------------
    def zero_(self, t: torch.Tensor):
        t.zero_()

    def set_tuning(self, key: str, value: int):
        check(self.lib.lako_tuning_set(C.byref(self.tuning), key.encode(), int(value)), "lako_tuning_set")

    # ---- MX block-scaled fp8 GEMM (BASELINE config 5) -------------------------------------------------------------
    @staticmethod
    def mx_scale_cols(K):
        """bytes of scales per operand row: 4 (k-blocks of a K-step) × KSP, KSP = ⌈K/128⌉ rounded up to 4"""
        return 4 * (((K // 128) + 3) // 4 * 4)

    def mx_quantize(self, x, q, scales):
        """x bf16 [rows, K] (unit inner stride) → q uint8 [rows, K] (e4m3 bytes), scales uint8 [rows, mx_scale_cols(K)]"""
        rows, K, ld = _rowmajor2d(x, "mx_quantize x")
        if x.dtype != torch.bfloat16 or q.dtype != torch.uint8 or scales.dtype != torch.uint8 or tuple(q.shape) != (rows, K) or \
                not q.is_contiguous() or not scales.is_contiguous() or tuple(scales.shape) != (rows, self.mx_scale_cols(K)):
            raise LakoError("mx_quantize: x bf16 [rows, K], q uint8 [rows, K], scales uint8 [rows, mx_scale_cols(K)], all contiguous rows")
        self._timed("mx_quantize", 0.0, lambda: check(self.lib.lako_mx_quantize(_p(x), rows, K, ld, _p(q), _p(scales), self._stream()), "lako_mx_quantize"))

    def gemm_nt_mx(self, Aq, As, Bq, Bs, Cm, *, alpha=1.0, relu=False, resid=None, aux=None, aux_scale=1.0, drop=None):
        """C bf16 [M, N] = epilogue(alpha · dequant(Aq, As) · dequant(Bq, Bs)ᵀ) on the block-scaled fp8 matrix cores"""
        M, K, lda = _rowmajor2d(Aq, "gemm_nt_mx A")
        N, K2, ldb = _rowmajor2d(Bq, "gemm_nt_mx B")
        M2, N2, ldc = _rowmajor2d(Cm, "gemm_nt_mx C")
        if K != K2 or M != M2 or N != N2 or Aq.dtype != torch.uint8 or Bq.dtype != torch.uint8 or Cm.dtype != torch.bfloat16:
            raise LakoError(f"gemm_nt_mx: shape/dtype mismatch A{tuple(Aq.shape)} B{tuple(Bq.shape)} C{tuple(Cm.shape)}")
        sc = self.mx_scale_cols(K)
        if tuple(As.shape) != (M, sc) or tuple(Bs.shape) != (N, sc) or not As.is_contiguous() or not Bs.is_contiguous():
            raise LakoError("gemm_nt_mx: scales must be contiguous [rows, mx_scale_cols(K)] uint8")
        p = GemmNT()
        p.A, p.B, p.C = Aq.data_ptr(), Bq.data_ptr(), Cm.data_ptr()
        p.M, p.N, p.K, p.lda, p.ldb, p.ldc = M, N, K, lda, ldb, ldc
        p.in_dtype, p.out_dtype = LAKO_FP8_E4M3, LAKO_BF16
        p.alpha = float(alpha)
        p.flags = (EPI_RELU if relu else 0) | (EPI_RESID if resid is not None else 0) | (EPI_AUXMASK if aux is not None else 0)
        if resid is not None:
            if resid.dtype != Cm.dtype or resid.shape != Cm.shape:
                raise LakoError("gemm_nt_mx: resid must match C")
            p.resid, p.ldr = resid.data_ptr(), resid.stride(0)
        if aux is not None:
            if aux.dtype != torch.bfloat16 or aux.shape != Cm.shape:
                raise LakoError("gemm_nt_mx: aux must be bf16 [M, N]")
            p.aux, p.ldaux = aux.data_ptr(), aux.stride(0)
        p.aux_scale = float(aux_scale)
        p.drop = _drop(drop)
        p.tuning = self._tuning_p
        self._timed("gemm_nt_mx", 2.0 * M * N * K, lambda: check(self.lib.lako_gemm_nt_mx(C.byref(p), _p(As), _p(Bs), self._stream()), "lako_gemm_nt_mx"))

    # ---- GEMMs -----------------------------------------------------------------------------
    def gemm_nt(self, A, B, Cm, *, alpha=1.0, relu=False, resid=None, aux=None, aux_scale=1.0, drop=None,
                atomic=False, norm=None):
        """`norm` = (w fp32 [K], eps, xn bf16 [M, K], rstd fp32 [M]): A is the un-normalised input of a T5LayerNorm — the product runs on
        bf16(w·(A·rstd)) formed inside the kernel (LAKO_EPI_NORM_A: one launch instead of two), xn / rstd receive what lako_rmsnorm_fwd would
        have written.  Where the library cannot do that (rows > 256, K > 1024, fp32 …) the norm runs as its own launch first."""
        if norm is not None:
            w, eps, xn, rstd = norm
            if self._gemm_nt_call(A, B, Cm, alpha, relu, resid, aux, aux_scale, drop, atomic, norm) == 0:
                return
            self.rmsnorm_fwd(A, w, xn, rstd, eps)         # (LAKO_E_UNSUPPORTED: nothing was launched)
            A = xn
        rc = self._gemm_nt_call(A, B, Cm, alpha, relu, resid, aux, aux_scale, drop, atomic, None)
        if rc != 0:
            check(rc, "lako_gemm_nt")

    def _gemm_nt_call(self, A, B, Cm, alpha, relu, resid, aux, aux_scale, drop, atomic, norm):
        M, K, lda = _rowmajor2d(A, "gemm_nt A")
        N, K2, ldb = _rowmajor2d(B, "gemm_nt B")
        M2, N2, ldc = _rowmajor2d(Cm, "gemm_nt C")
        if K != K2 or M != M2 or N != N2 or A.dtype != B.dtype:
            raise LakoError(f"gemm_nt: shape/dtype mismatch A{tuple(A.shape)} B{tuple(B.shape)} C{tuple(Cm.shape)}")
        flags = (EPI_RELU if relu else 0) | (EPI_RESID if resid is not None else 0) | \
                (EPI_AUXMASK if aux is not None else 0) | (EPI_ATOMIC if atomic else 0) | (EPI_NORM_A if norm is not None else 0)
        p = GemmNT()
        p.A, p.B, p.C = A.data_ptr(), B.data_ptr(), Cm.data_ptr()
        p.M, p.N, p.K, p.lda, p.ldb, p.ldc = M, N, K, lda, ldb, ldc
        p.in_dtype, p.out_dtype = _dt(A), _dt(Cm)
        p.alpha, p.flags = float(alpha), flags
        if resid is not None:
            if resid.dtype != Cm.dtype or resid.shape != Cm.shape:
                raise LakoError("gemm_nt: resid must match C")
            p.resid, p.ldr = resid.data_ptr(), resid.stride(0)
        if aux is not None:
            if aux.dtype != A.dtype or aux.shape != Cm.shape:
                raise LakoError("gemm_nt: aux must be [M,N] in the input dtype")
            p.aux, p.ldaux = aux.data_ptr(), aux.stride(0)
        p.aux_scale = float(aux_scale)
        p.drop = _drop(drop)
        p.tuning = self._tuning_p
        if norm is not None:
            w, eps, xn, rstd = norm
            if w.dtype != torch.float32 or w.numel() != K or xn.shape != A.shape or xn.dtype != A.dtype or xn.stride(1) != 1 or \
                    rstd.dtype != torch.float32 or rstd.numel() != M:
                raise LakoError("gemm_nt: norm = (w fp32 [K], eps, xn like A, rstd fp32 [M])")
            p.norm_w, p.norm_eps, p.norm_out, p.norm_ld, p.norm_rstd = _p(w), float(eps), _p(xn), xn.stride(0), _p(rstd)
        # probe classes follow the kernel the library picks: M <= 256 rows (the decoder) runs on the split-K / ring kernels
        rc = [0]

        def go():
            rc[0] = self.lib.lako_gemm_nt(C.byref(p), self._stream())
            if rc[0] != 0 and not (norm is not None and rc[0] == -4):        # -4 = LAKO_E_UNSUPPORTED: the caller falls back
                check(rc[0], "lako_gemm_nt")
        self._timed(f"gemm_nt{'_skinny' if M <= 256 else ''}.{p.in_dtype}{p.out_dtype}", 2.0 * M * N * K, go)
        return rc[0]

    def gemm_tn(self, A, B, Cm, *, alpha=1.0, split_k=0):
        K, M, lda = _rowmajor2d(A, "gemm_tn A")
        K2, N, ldb = _rowmajor2d(B, "gemm_tn B")
        M2, N2, ldc = _rowmajor2d(Cm, "gemm_tn C")
        if K != K2 or M != M2 or N != N2 or A.dtype != B.dtype or Cm.dtype != torch.float32:
            raise LakoError(f"gemm_tn: shape/dtype mismatch A{tuple(A.shape)} B{tuple(B.shape)} C{tuple(Cm.shape)}")
        if self.det and split_k == 0:
            split_k = 1
        self._timed("gemm_tn", 2.0 * M * N * K, lambda: check(self.lib.lako_gemm_tn(_p(A), _p(B), _p(Cm), M, N, K, lda, ldb, ldc, _dt(A), float(alpha), int(split_k),
                                    self._tuning_p, self._stream()), "lako_gemm_tn"))

    def gemm_tn_grouped(self, problems, split_k=0, workspace=None):
        """`workspace`: callable(nbytes) → a device uint8 tensor of at least that size (the engine's grow-only scratch), or None.
        [(A [K, M], B [K, N], C [M, N] fp32, alpha[, rows_out]), …] with one K: every C += alpha·Aᵀ·B, one launch per
        TN_GROUP_MAX problems (see lako_gemm_tn_grouped in include/lako_hip.h).  split_k=1: one contributor per output element;
        −1: the same and nothing else adds to C meanwhile (plain read-modify-write); −2: C is overwritten."""
        from ._lib import TN_GROUP_MAX, GemmTNItem
        for g0 in range(0, len(problems), TN_GROUP_MAX):
            grp = problems[g0:g0 + TN_GROUP_MAX]
            arr = (GemmTNItem * len(grp))()
            K0, flops = None, 0.0
            for it, prob in zip(arr, grp):
                A, B, Cm, alpha = prob[:4]
                it.rows_out = int(prob[4]) if len(prob) > 4 else 0       # (A, B, C, alpha[, rows of C actually written])
                K, M, lda = _rowmajor2d(A, "gemm_tn_grouped A")
                K2, N, ldb = _rowmajor2d(B, "gemm_tn_grouped B")
                M2, N2, ldc = _rowmajor2d(Cm, "gemm_tn_grouped C")
                if K != K2 or M != M2 or N != N2 or A.dtype != B.dtype or Cm.dtype != torch.float32 or A.dtype != grp[0][0].dtype:
                    raise LakoError(f"gemm_tn_grouped: shape/dtype mismatch A{tuple(A.shape)} B{tuple(B.shape)} C{tuple(Cm.shape)}")
                if K0 is None:
                    K0 = K
                elif K != K0:
                    raise LakoError("gemm_tn_grouped: all problems must share K")
                it.a, it.b, it.c = _p(A), _p(B), _p(Cm)
                it.M, it.N, it.lda, it.ldb, it.ldc, it.alpha = M, N, lda, ldb, ldc, float(alpha)
                flops += 2.0 * M * N * K
            ws_p, ws_n = None, 0
            if workspace is not None:      # K-splits through partial tiles in caller scratch instead of float atomics (see the header)
                need = int(self.lib.lako_gemm_tn_grouped_workspace(arr, len(grp), K0, _dt(grp[0][0]), int(split_k), self._tuning_p))
                if need > 0:
                    ws_p, ws_n = _p(workspace(need)), need
            self._timed("gemm_tn", flops, lambda: check(self.lib.lako_gemm_tn_grouped(arr, len(grp), K0, _dt(grp[0][0]), int(split_k), self._tuning_p,
                                                                                      ws_p, ws_n, self._stream()), "lako_gemm_tn_grouped"))

    # ---- norm / embedding / dropout ---------------------------------------------------------------
    def rmsnorm_fwd(self, x, w, y, rstd, eps, drop=None):
        rows, d = x.shape
        self._timed("rmsnorm_fwd", 0.0, lambda: check(self.lib.lako_rmsnorm_fwd(_p(x), _p(w), _p(y), _p(rstd), rows, d, float(eps), _dt(x), _drop(drop),
                                        self._stream()), "lako_rmsnorm_fwd"))

    def rmsnorm_fwd_mx(self, x, w, y, rstd, eps, q, scales):
        """rmsnorm_fwd (no dropout, bf16) + mx_quantize(y) in one pass: q / scales as mx_quantize would write them"""
        rows, d = x.shape
        if x.dtype != torch.bfloat16 or q.dtype != torch.uint8 or tuple(q.shape) != (rows, d) or not q.is_contiguous() or \
                tuple(scales.shape) != (rows, self.mx_scale_cols(d)) or not scales.is_contiguous() or not x.is_contiguous() or not y.is_contiguous():
            raise LakoError("rmsnorm_fwd_mx: bf16 contiguous x / y [rows, d], q uint8 [rows, d], scales uint8 [rows, mx_scale_cols(d)]")
        self._timed("rmsnorm_fwd", 0.0, lambda: check(self.lib.lako_rmsnorm_fwd_mx(_p(x), _p(w), _p(y), _p(rstd), _p(q), _p(scales), rows, d, float(eps), self._stream()),
                                                      "lako_rmsnorm_fwd_mx"))

    def rmsnorm_bwd(self, dy, x, w, rstd, dres, dx, dw, drop=None, dx_drop=None, drop_out=None):
        """dx_drop (optional): also receives dropout_apply(dx, drop_out) — the next residual branch's incoming gradient"""
        rows, d = x.shape
        if dx_drop is not None and (not dx_drop.is_contiguous() or dx_drop.shape != dx.shape or dx_drop.dtype != dx.dtype):
            raise LakoError("rmsnorm_bwd: dx_drop must match dx")
        self._timed("rmsnorm_bwd", 0.0, lambda: check(self.lib.lako_rmsnorm_bwd(_p(dy), _p(x), _p(w), _p(rstd), _p(dres), _p(dx), _p(dw), rows, d, _dt(x),
                                        _drop(drop), _p(dx_drop), _drop(drop_out), self._stream()), "lako_rmsnorm_bwd"))

    def embed_fwd(self, ids, table, out, drop=None):
        self._timed("embed_fwd", 0.0, lambda: check(self.lib.lako_embed_fwd(_p(ids), _p(table), _p(out), ids.numel(), table.shape[1], table.shape[0],
                                      _dt(table), _drop(drop), self._stream()), "lako_embed_fwd"))

    def embed_bwd(self, ids, dout, dtable, drop=None):
        self._timed("embed_bwd", 0.0, lambda: check(self.lib.lako_embed_bwd(_p(ids), _p(dout), _p(dtable), ids.numel(), dtable.shape[1], dtable.shape[0],
                                      _dt(dout), _drop(drop), self._stream()), "lako_embed_bwd"))

    def embed_bwd_ordered(self, ids, perm, dout, dtable, drop=None):
        """embed_bwd with the rows of one id added in position order by one wave (`perm`: stable argsort of ids)."""
        self._timed("embed_bwd", 0.0, lambda: check(self.lib.lako_embed_bwd_ordered(_p(ids), _p(perm), _p(dout), _p(dtable), ids.numel(), dtable.shape[1],
                                      dtable.shape[0], _dt(dout), _drop(drop), self._stream()), "lako_embed_bwd_ordered"))

    def dropout_apply(self, x, y, drop):
        self._timed("dropout_apply", 0.0, lambda: check(self.lib.lako_dropout_apply(_p(x), _p(y), x.numel(), _dt(x), _drop(drop), self._stream()), "lako_dropout_apply"))

    # ---- relative position bias ---------------------------------------------------------------
    def relpos_expand(self, table, lut, rel):
        nb, H = table.shape
        self._timed("relpos_expand", 0.0, lambda: check(self.lib.lako_relpos_expand(_p(table), _p(lut), _p(rel), H, rel.shape[1], nb, self._stream()), "lako_relpos_expand"))

    def relpos_reduce(self, drel, lut, dtable):
        nb, H = dtable.shape
        self._timed("relpos_reduce", 0.0, lambda: check(self.lib.lako_relpos_reduce(_p(drel), _p(lut), _p(dtable), H, drel.shape[1], nb, self._stream()), "lako_relpos_reduce"))

    # ---- attention ------------------------------------------------------------------------------
    @staticmethod
    def _ragged(p, q, k, q_off, k_off, max_q, max_k):
        """ragged sequences (lako_attn_fwd_t.q_off / k_off): q / k arrive as ONE packed [1, rows, H, dk] view each."""
        Bn, Lq, H, dk = q.shape
        Lk = k.shape[1]
        for off, t, mx, nm in ((q_off, q, max_q, "q"), (k_off, k, max_k, "k")):
            if off is not None:
                if off.dtype != torch.int32 or not off.is_contiguous() or t.shape[0] != 1 or mx is None:
                    raise LakoError(f"attention: {nm}_off must be contiguous int32 [Bn + 1], {nm} a packed [1, rows, H, dk] view, max_{nm} given")
        if q_off is not None:
            Bn, Lq, p.q_off = q_off.numel() - 1, int(max_q), q_off.data_ptr()
        if k_off is not None:
            Lk, p.k_off = int(max_k), k_off.data_ptr()
            if q_off is None and k_off.numel() - 1 != Bn:
                raise LakoError("attention: k_off must have Bn + 1 entries")
        return Bn, Lq, Lk, H, dk

    @staticmethod
    def _order(order, Bn):
        """lako_attn_*_t.order: a device int32 permutation of the Bn sequences (processing order; results do not depend on it)"""
        if order is None:
            return None
        if order.dtype != torch.int32 or not order.is_contiguous() or order.numel() != Bn:
            raise LakoError(f"attention: order must be a contiguous int32 permutation of the {Bn} sequences")
        return order.data_ptr()

    def attn_fwd(self, q, k, v, out, stats, *, rel_bias=None, rel_off=0, key_mask=None, causal=False, causal_off=0,
                 drop=None, scores_out=None, q_off=None, k_off=None, max_q=None, max_k=None, order=None):
        p = AttnFwd()
        Bn, Lq, Lk, H, dk = self._ragged(p, q, k, q_off, k_off, max_q, max_k)
        p.order = self._order(order, Bn)
        p.q, p.k, p.v, p.out, p.lse = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), stats.data_ptr()
        p.q_stride_b, p.q_stride_t = _bthd(q, "attn q")
        p.k_stride_b, p.k_stride_t = _bthd(k, "attn k")
        p.v_stride_b, p.v_stride_t = _bthd(v, "attn v")
        p.o_stride_b, p.o_stride_t = _bthd(out, "attn out")
        if rel_bias is not None:
            p.rel_bias, p.R = rel_bias.data_ptr(), rel_bias.shape[1]
        p.rel_off = int(rel_off)
        if key_mask is not None:
            if key_mask.dtype not in (torch.uint8, torch.bool) or tuple(key_mask.shape) != (Bn, Lk):
                raise LakoError("attn_fwd: key_mask must be uint8/bool [Bn, Lk]")
            p.key_mask = key_mask.data_ptr()
        p.causal, p.causal_off = int(causal), int(causal_off)
        p.Bn, p.H, p.Lq, p.Lk, p.d_head = Bn, H, Lq, Lk, dk
        p.dtype = _dt(q)
        p.drop = _drop(drop)
        if scores_out is not None:
            p.scores_out = scores_out.data_ptr()
        self._timed("attn_fwd", 4.0 * Bn * H * Lq * Lk * dk, lambda: check(self.lib.lako_attn_fwd(C.byref(p), self._stream()), "lako_attn_fwd"))

    def attn_bwd(self, q, k, v, out, dout, stats, dq, dk_, dv, *, rel_bias=None, drel=None, rel_off=0, key_mask=None,
                 causal=False, causal_off=0, drop=None, q_off=None, k_off=None, max_q=None, max_k=None, order=None):
        p = AttnBwd()
        Bn, Lq, Lk, H, dk = self._ragged(p, q, k, q_off, k_off, max_q, max_k)
        p.order = self._order(order, Bn)
        p.q, p.k, p.v, p.out, p.dout, p.lse = (q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(),
                                               dout.data_ptr(), stats.data_ptr())
        p.dq_out, p.dk_out, p.dv_out = dq.data_ptr(), dk_.data_ptr(), dv.data_ptr()
        p.q_stride_b, p.q_stride_t = _bthd(q, "attn q")
        p.k_stride_b, p.k_stride_t = _bthd(k, "attn k")
        p.v_stride_b, p.v_stride_t = _bthd(v, "attn v")
        p.o_stride_b, p.o_stride_t = _bthd(out, "attn out")
        for a, b_, n in ((dq, q, "dq"), (dk_, k, "dk"), (dv, v, "dv"), (dout, out, "dout")):
            if a.stride() != b_.stride() or a.shape != b_.shape:
                raise LakoError(f"attn_bwd: {n} must share the layout of its forward tensor")
        if rel_bias is not None:
            p.rel_bias, p.R = rel_bias.data_ptr(), rel_bias.shape[1]
        if drel is not None:
            p.drel = drel.data_ptr()
        p.rel_off = int(rel_off)
        if key_mask is not None:
            p.key_mask = key_mask.data_ptr()
        p.causal, p.causal_off = int(causal), int(causal_off)
        p.Bn, p.H, p.Lq, p.Lk, p.d_head = Bn, H, Lq, Lk, dk
        p.dtype = _dt(q)
        p.drop = _drop(drop)
        self._timed("attn_bwd", 8.0 * Bn * H * Lq * Lk * dk, lambda: check(self.lib.lako_attn_bwd(C.byref(p), self._stream()), "lako_attn_bwd"))

    # ---- loss / optimizer -----------------------------------------------------------------------
    # ---- cross-attention in the encoder-state space (csrc/xattn.hip) ------------------------------------
    @staticmethod
    def xattn_ok(dtype, d_kv, d_model):
        """shapes the kernels take (everything else stays on lako_attn_fwd over projected K / V)"""
        return dtype == torch.bfloat16 and d_kv == 64 and d_model % 128 == 0

    @staticmethod
    def _bthx(t, name):
        """[B, T, H, X] view, last dim contiguous → (element strides of b, t, h)"""
        if t.dim() != 4 or t.stride(3) != 1:
            raise LakoError(f"{name}: expected a [B, T, H, X] view with a contiguous last dim, got {tuple(t.shape)} / {t.stride()}")
        return t.stride(0), t.stride(1), t.stride(2)

    def headbatch_nt(self, A, Bw, Cm):
        """Cm[b,t,h,n] = Σ_k A[b,t,h,k]·Bw[h,n,k]:  A [B,T,H,K] bf16|fp32, Bw [H,N,K] bf16, Cm [B,T,H,N] bf16 (strided views)"""
        from ._lib import HeadBatch
        slabs, slab_stride = 1, 0
        if A.dim() == 5:      # [Z, B, T, H, K] fp32: the key-split slabs of xattn_context, added in order
            slabs, slab_stride, A = A.shape[0], A.stride(0), A[0]
        Bz, T, H, K = A.shape
        N = Bw.shape[1]
        if Bw.shape != (H, N, K) or Cm.shape != (Bz, T, H, N) or Bw.stride(2) != 1:
            raise LakoError(f"headbatch_nt: shapes A{tuple(A.shape)} B{tuple(Bw.shape)} C{tuple(Cm.shape)}")
        hb = HeadBatch()
        hb.a, hb.b, hb.c = _p(A), _p(Bw), _p(Cm)
        hb.a_sb, hb.a_st, hb.a_sh = self._bthx(A, "headbatch_nt A")
        hb.b_sh, hb.ldb = Bw.stride(0), Bw.stride(1)
        hb.c_sb, hb.c_st, hb.c_sh = self._bthx(Cm, "headbatch_nt C")
        hb.M, hb.T, hb.H, hb.N, hb.K = Bz * T, T, H, N, K
        hb.a_dtype, hb.b_dtype = _dt(A), _dt(Bw)
        hb.n_slabs, hb.slab_stride = slabs, slab_stride
        self._timed("headbatch", 2.0 * Bz * T * H * N * K, lambda: check(self.lib.lako_headbatch_nt(hb, self._stream()), "lako_headbatch_nt"))

    def headbatch_tn(self, A, Bm, Cw):
        """Cw[h,j,c] += Σ_{b,t} A[b,t,h,j]·Bm[b,t,h,c]:  A [B,T,H,64] bf16, Bm [B,T,H,N] bf16|fp32, Cw [H,64,N] fp32 (views)"""
        hb, flops = self._hb_tn_item(A, Bm, Cw)
        self._timed("headbatch", flops, lambda: check(self.lib.lako_headbatch_tn(hb, self._stream()), "lako_headbatch_tn"))

    def headbatch_tn_multi(self, problems):
        """[(A, Bm, Cw), …] as headbatch_tn, problems of one shape, HB_MULTI_MAX per launch (lako_headbatch_tn_multi)"""
        from ._lib import HB_MULTI_MAX, HeadBatch
        for g0 in range(0, len(problems), HB_MULTI_MAX):
            grp = problems[g0:g0 + HB_MULTI_MAX]
            arr = (HeadBatch * len(grp))()
            flops = 0.0
            for i, (A, Bm, Cw) in enumerate(grp):
                hb, fl = self._hb_tn_item(A, Bm, Cw)
                C.memmove(C.byref(arr, i * C.sizeof(HeadBatch)), C.byref(hb), C.sizeof(HeadBatch))
                flops += fl
            self._timed("headbatch", flops, lambda: check(self.lib.lako_headbatch_tn_multi(arr, len(grp), self._stream()), "lako_headbatch_tn_multi"))

    def _hb_tn_item(self, A, Bm, Cw):
        from ._lib import HeadBatch
        slabs, slab_stride = 1, 0
        if Bm.dim() == 5:
            slabs, slab_stride, Bm = Bm.shape[0], Bm.stride(0), Bm[0]
        Bz, T, H, K = A.shape
        N = Bm.shape[3]
        if Bm.shape != (Bz, T, H, N) or Cw.shape != (H, K, N) or Cw.stride(2) != 1 or Cw.dtype != torch.float32:
            raise LakoError(f"headbatch_tn: shapes A{tuple(A.shape)} B{tuple(Bm.shape)} C{tuple(Cw.shape)}")
        hb = HeadBatch()
        hb.a, hb.b, hb.c = _p(A), _p(Bm), _p(Cw)
        hb.a_sb, hb.a_st, hb.a_sh = self._bthx(A, "headbatch_tn A")
        hb.b_sb, hb.b_st, hb.b_sh = self._bthx(Bm, "headbatch_tn B")
        hb.c_sh, hb.c_st = Cw.stride(0), Cw.stride(1)
        hb.M, hb.T, hb.H, hb.N, hb.K = Bz * T, T, H, N, K
        hb.a_dtype, hb.b_dtype = _dt(A), _dt(Bm)
        hb.n_slabs, hb.slab_stride = slabs, slab_stride
        return hb, 2.0 * Bz * T * H * N * K

    def xattn_scores(self, Q, E, k_off, p_off, p_total, S):
        """S[r, p_off[b] + s] = Q[b, r, :]·E[k_off[b] + s, :]:  Q [B, R, D] bf16 view, E [rows, D] bf16, S [R, >= p_total] fp32"""
        Bz, R, D = Q.shape
        if Q.stride(2) != 1 or E.stride(1) != 1 or S.stride(1) != 1 or S.shape[0] != R or E.shape[1] != D:
            raise LakoError(f"xattn_scores: shapes Q{tuple(Q.shape)} E{tuple(E.shape)} S{tuple(S.shape)}")
        self._timed("xattn", 2.0 * R * D * E.shape[0], lambda: check(self.lib.lako_xattn_scores(
            _p(Q), Q.stride(0), Q.stride(1), _p(E), E.stride(0), _p(k_off), _p(p_off), int(p_total), _p(S), S.stride(0), R, D, Bz,
            self._stream()), "lako_xattn_scores"))

    def xattn_context(self, P, E, k_off, p_off, out):
        """Σ_z out[z, b, r, :] = Σ_s P[r, p_off[b] + s]·E[k_off[b] + s, :]:  P [R, ld] bf16, out [Z, B, R, D] fp32 view — slab z gets
        the z-th key range of every sample (plain stores; the consumers add the slabs)"""
        Z, Bz, R, D = out.shape
        if P.stride(1) != 1 or P.shape[0] != R or out.stride(3) != 1 or out.dtype != torch.float32 or E.shape[1] != D:
            raise LakoError(f"xattn_context: shapes P{tuple(P.shape)} E{tuple(E.shape)} out{tuple(out.shape)}")
        self._timed("xattn", 2.0 * R * D * E.shape[0], lambda: check(self.lib.lako_xattn_context(
            _p(P), P.stride(0), _p(E), E.stride(0), _p(k_off), _p(p_off), _p(out), out.stride(0), out.stride(1), out.stride(2),
            R, D, Bz, Z, self._stream()), "lako_xattn_context"))

    @staticmethod
    def xattn_decode_ok(H, d_model):
        return H <= 16 and d_model in (512, 768, 1024)

    def xattn_decode(self, Q, E, k_off, part_ml, part_c):
        """one decode step, scores + softmax + context of a key range per workgroup: Q [B, R <= 16, D] bf16 view,
        part_ml [Z, B, 16, 2] / part_c [Z, B, 16, D] fp32 (contiguous) receive every range's (max, Σ exp) and Σ exp·E"""
        Bz, R, D = Q.shape
        Z = part_c.shape[0]
        if part_c.shape != (Z, Bz, 16, D) or part_ml.shape != (Z, Bz, 16, 2) or not part_c.is_contiguous() or not part_ml.is_contiguous() \
                or Q.stride(2) != 1 or E.stride(1) != 1 or E.shape[1] != D:
            raise LakoError(f"xattn_decode: shapes Q{tuple(Q.shape)} E{tuple(E.shape)} part{tuple(part_c.shape)}")
        self._timed("xattn", 4.0 * R * D * E.shape[0], lambda: check(self.lib.lako_xattn_decode(
            _p(Q), Q.stride(0), Q.stride(1), _p(E), E.stride(0), _p(k_off), _p(part_ml), _p(part_c), R, D, Bz, Z, self._stream()),
            "lako_xattn_decode"))

    def xattn_decode_combine(self, part_ml, part_c, Wv, ctx, H):
        """ctx [B, H·64] bf16 = (merged ranges of xattn_decode)·Wvᵀ per head; Wv [H·64, D] bf16 rows of the V projection"""
        Z, Bz, _, D = part_c.shape
        if Wv.shape != (H * 64, D) or Wv.stride(1) != 1 or ctx.shape != (Bz, H * 64) or ctx.stride(1) != 1:
            raise LakoError(f"xattn_decode_combine: shapes Wv{tuple(Wv.shape)} ctx{tuple(ctx.shape)}")
        self._timed("xattn", 2.0 * Bz * H * 64 * D, lambda: check(self.lib.lako_xattn_decode_combine(
            _p(part_ml), _p(part_c), _p(Wv), Wv.stride(0), _p(ctx), ctx.stride(0), H, D, Bz, Z, self._stream()),
            "lako_xattn_decode_combine"))

    def xattn_softmax_fwd(self, S, stats, P, k_off, p_off, T, H, max_keys, drop=None):
        """stats [B, T·H, 2] fp32, P [T·H, ld] bf16 = dropout(softmax over each sample's keys of S)"""
        Bz = stats.shape[0]
        self._timed("xattn_softmax", 0.0, lambda: check(self.lib.lako_xattn_softmax_fwd(
            _p(S), S.stride(0), _p(stats), _p(P), P.stride(0), _p(k_off), _p(p_off), Bz, T, H, int(max_keys), _drop(drop),
            self._stream()), "lako_xattn_softmax_fwd"))

    def xattn_softmax_bwd(self, S, dP, stats, dS, k_off, p_off, T, H, max_keys, drop=None):
        Bz = stats.shape[0]
        if dP.stride(0) != S.stride(0):
            raise LakoError("xattn_softmax_bwd: S and dP must share their row stride")
        self._timed("xattn_softmax", 0.0, lambda: check(self.lib.lako_xattn_softmax_bwd(
            _p(S), _p(dP), S.stride(0), _p(stats), _p(dS), dS.stride(0), _p(k_off), _p(p_off), Bz, T, H, int(max_keys),
            _drop(drop), self._stream()), "lako_xattn_softmax_bwd"))

    # ---- loss / optimizer ---------------------------------------------------------------------------
    def ce_fwd_bwd(self, logits, labels, loss_out, dlogits, upstream=None):
        M, V = logits.shape
        self._timed("ce_fwd_bwd", 0.0, lambda: check(self.lib.lako_ce_fwd_bwd(_p(logits), _p(labels), _p(loss_out), _p(dlogits), _p(upstream), M, V,
                                       _dt(dlogits) if dlogits is not None else LAKO_F32, self._stream()), "lako_ce_fwd_bwd"))

    def sumsq(self, g, out):
        self._timed("sumsq", 0.0, lambda: check(self.lib.lako_sumsq(_p(g), g.numel(), _p(out), self._stream()), "lako_sumsq"))

    def adamw_step(self, p, g, m, v, shadow, *, lr, beta1, beta2, eps, weight_decay, gnorm_sq, max_norm, grad_scale):
        self._timed("adamw_step", 0.0, lambda: check(self.lib.lako_adamw_step(_p(p), _p(g), _p(m), _p(v), _p(shadow), p.numel(), float(lr), float(beta1),
                                       float(beta2), float(eps), float(weight_decay), _p(gnorm_sq), float(max_norm),
                                       float(grad_scale), _dt(shadow) if shadow is not None else LAKO_F32,
                                       self._stream()), "lako_adamw_step"))

    def transpose_cast(self, src, dst):
        rows, cols = src.shape
        self._timed("transpose_cast", 0.0, lambda: check(self.lib.lako_transpose_cast(_p(src), _p(dst), rows, cols, _dt(dst), self._stream()), "lako_transpose_cast"))

    def transpose_cast_batched(self, src_flat, dst_flat, desc, tile_prefix, n, total_tiles):
        """every [rows, cols] matrix of the table (fp32, or bf16 for bf16 copies) → its [cols, rows] compute-dtype copy, one launch
        (desc and tile_prefix are device tensors: see lako_transpose_cast_batched in include/lako_hip.h)"""
        if src_flat.dtype not in (torch.float32, torch.bfloat16) or (src_flat.dtype == torch.bfloat16 and dst_flat.dtype != torch.bfloat16):
            raise LakoError("transpose_cast_batched: fp32 source, or bf16 source with bf16 copies")
        self._timed("transpose_cast", 0.0, lambda: check(self.lib.lako_transpose_cast_batched(
            _p(src_flat), _dt(src_flat), _p(dst_flat), _p(desc), _p(tile_prefix), int(n), int(total_tiles), _dt(dst_flat), self._stream()),
            "lako_transpose_cast_batched"))

    def cast(self, src, dst):
        self._timed("cast", 0.0, lambda: check(self.lib.lako_cast(_p(src), _p(dst), src.numel(), _dt(dst), self._stream()), "lako_cast"))

    # ---- per-fact aggregation of captured cross-attention scores ---------------------------
    FACT_STYLES = {"mean": 0, "max": 1, "21mean": 2}

    def fact_scores(self, scores, mask, ids, out, *, layer0, layers_used, passage, style, ids_passage=None):
        """scores fp32 [B, H, nl, N·L], mask uint8/bool [B, N, L], ids int64 [B, N, L] → out fp64 [B, n_context]
        (see lako_fact_scores in include/lako_hip.h; src/model.py:143-204)"""
        B, H, nl, S = scores.shape
        _, N, L = ids.shape
        if scores.dtype != torch.float32 or not scores.is_contiguous() or S != N * L or out.dtype != torch.float64 or \
                ids.dtype != torch.int64 or not ids.is_contiguous() or not mask.is_contiguous() or \
                mask.dtype not in (torch.uint8, torch.bool) or tuple(mask.shape) != (B, N, L) or not out.is_contiguous():
            raise LakoError("fact_scores: scores fp32 [B,H,nl,N·L], mask uint8 [B,N,L], ids int64 [B,N,L], out fp64 [B,n_context]")
        if style not in self.FACT_STYLES:
            raise LakoError(f"fact_scores: attention_score_style {style!r} (mean | max | 21mean)")
        self._timed("fact_scores", 0.0, lambda: check(self.lib.lako_fact_scores(
            _p(scores), _p(mask), _p(ids), _p(out), B, H, nl, int(layer0), int(layers_used), N, L, int(passage),
            int(passage if ids_passage is None else ids_passage), out.shape[1],
            self.FACT_STYLES[style], self._stream()), "lako_fact_scores"))

    # ---- exact inner-product search ---------------------------------------------------
    def topk(self, scores, k, out_vals, out_idx):
        """the k largest entries of every row of fp32 scores [rows, n], descending, ties in ascending index order"""
        rows, n = scores.shape
        if scores.dtype != torch.float32 or scores.stride(1) != 1 or out_idx.dtype != torch.int64 or out_vals.dtype != torch.float32:
            raise LakoError("topk: scores fp32 row-major, out_vals fp32, out_idx int64")
        if tuple(out_vals.shape) != (rows, k) or tuple(out_idx.shape) != (rows, k) or not out_vals.is_contiguous() or not out_idx.is_contiguous():
            raise LakoError("topk: outputs must be contiguous [rows, k]")
        self._timed("topk", 0.0, lambda: check(self.lib.lako_topk(_p(scores), rows, n, scores.stride(0), int(k), _p(out_vals), _p(out_idx),
                                                                 self._stream()), "lako_topk"))

    # ---- retriever bi-encoder forward ---------------------------------------------------

    # ---- product quantiser (faiss.IndexPQ, src/index.py:21-23) ------------------------------------------------------------
    def pq_assign(self, x, centroids, codes=None, sums=None, counts=None, err=None):
        """x [n, >= M·dsub] fp32 (row stride free), centroids [M, ksub, dsub] fp32 → codes [n, M] uint8 (nearest centroid per
        sub-quantiser) and / or the k-means accumulators sums [M, ksub, dsub] / counts [M, ksub] int32 (+ err [1])."""
        M, ksub, dsub = centroids.shape
        if x.dtype != torch.float32 or centroids.dtype != torch.float32 or x.stride(1) != 1 or not centroids.is_contiguous():
            raise LakoError("pq_assign: x / centroids fp32, x rows contiguous, centroids contiguous")
        if x.dim() != 2 or x.shape[1] < M * dsub:
            raise LakoError(f"pq_assign: x has {tuple(x.shape)} columns, the centroids split M·dsub = {M * dsub}")
        if codes is not None and (codes.dtype != torch.uint8 or not codes.is_contiguous() or tuple(codes.shape) != (x.shape[0], M)):
            raise LakoError("pq_assign: codes must be contiguous uint8 [n, M]")
        if (sums is None) != (counts is None) or (sums is not None and (tuple(sums.shape) != (M, ksub, dsub) or counts.dtype != torch.int32 or
                                                                    sums.dtype != torch.float32 or not sums.is_contiguous() or
                                                                    tuple(counts.shape) != (M, ksub) or not counts.is_contiguous())):
            raise LakoError("pq_assign: sums [M, ksub, dsub] fp32 and counts [M, ksub] int32 go together")
        self._timed("pq_assign", 0.0, lambda: check(self.lib.lako_pq_assign(_p(x), x.shape[0], x.stride(0), _p(centroids), M, ksub, dsub, _p(codes),
                                                                            _p(sums), _p(counts), _p(err), self._stream()), "lako_pq_assign"))

    def pq_lut(self, q, centroids, lut):
        M, ksub, dsub = centroids.shape
        if q.dtype != torch.float32 or q.stride(1) != 1 or not lut.is_contiguous() or tuple(lut.shape) != (q.shape[0], M, ksub) or \
                lut.dtype != torch.float32 or centroids.dtype != torch.float32 or not centroids.is_contiguous():
            raise LakoError("pq_lut: q fp32 rows contiguous, lut contiguous fp32 [nq, M, ksub], centroids contiguous fp32")
        if q.dim() != 2 or q.shape[1] < M * dsub:
            raise LakoError(f"pq_lut: q has {tuple(q.shape)} columns, the centroids split M·dsub = {M * dsub}")
        self._timed("pq_lut", 0.0, lambda: check(self.lib.lako_pq_lut(_p(q), q.shape[0], q.stride(0), _p(centroids), M, ksub, dsub, _p(lut),
                                                                      self._stream()), "lako_pq_lut"))

    def pq_scan(self, lut, codes, scores):
        nq, M, ksub = lut.shape
        n = codes.shape[0]
        if not lut.is_contiguous() or codes.dtype != torch.uint8 or not codes.is_contiguous() or codes.shape[1] != M or \
                scores.dtype != torch.float32 or scores.stride(1) != 1 or scores.shape[0] != nq or scores.shape[1] != n:
            raise LakoError("pq_scan: lut [nq, M, ksub] fp32, codes [n, M] uint8, scores [nq, n] fp32 (row stride free)")
        self._timed("pq_scan", 0.0, lambda: check(self.lib.lako_pq_scan(_p(lut), _p(codes), n, nq, M, ksub, _p(scores), scores.stride(0),
                                                                        self._stream()), "lako_pq_scan"))

    def layernorm_fwd(self, x, gamma, beta, y, *, lin_bias=None, resid=None, eps=1e-12):
        """y = LayerNorm(x + lin_bias + resid)·gamma + beta over the rows of [rows, d]"""
        rows, d = x.shape
        if not x.is_contiguous() or not y.is_contiguous() or (resid is not None and (not resid.is_contiguous() or resid.dtype != x.dtype)):
            raise LakoError("layernorm_fwd: contiguous [rows, d] tensors of one dtype")
        self._timed("layernorm_fwd", 0.0, lambda: check(self.lib.lako_layernorm_fwd(_p(x), _p(lin_bias), _p(resid), _p(gamma), _p(beta), _p(y), rows, d,
                                                                                float(eps), _dt(x), self._stream()), "lako_layernorm_fwd"))

    def bert_embed(self, ids, word, pos, type0, gamma, beta, out, L, eps=1e-12):
        n_tok, d = out.shape
        if L > pos.shape[0]:
            raise LakoError(f"bert_embed: sequence length {L} exceeds the {pos.shape[0]} learned positions")
        self._timed("bert_embed", 0.0, lambda: check(self.lib.lako_bert_embed(_p(ids), _p(word), _p(pos), _p(type0), _p(gamma), _p(beta), _p(out), n_tok,
                                                                             int(L), d, word.shape[0], float(eps), _dt(out), self._stream()), "lako_bert_embed"))

    def bias_act(self, x, bias, y, gelu=False):
        rows, n = x.shape
        if not x.is_contiguous() or not y.is_contiguous():
            raise LakoError("bias_act: contiguous [rows, n] tensors")
        self._timed("bias_act", 0.0, lambda: check(self.lib.lako_bias_act(_p(x), _p(bias), _p(y), rows, n, 1 if gelu else 0, _dt(x), self._stream()), "lako_bias_act"))

    def seq_mean(self, x, mask, out):
        B, L, d = x.shape
        if not x.is_contiguous() or out.dtype != torch.float32 or (mask is not None and mask.dtype not in (torch.uint8, torch.bool)):
            raise LakoError("seq_mean: x contiguous [B, L, d], mask uint8/bool [B, L], out fp32 [B, d]")
        self._timed("seq_mean", 0.0, lambda: check(self.lib.lako_seq_mean(_p(x), _p(mask), _p(out), B, L, d, _dt(x), self._stream()), "lako_seq_mean"))

    def bi_score(self, q, p, out, scale):
        B, n, d = p.shape
        if q.dtype != torch.float32 or p.dtype != torch.float32 or not q.is_contiguous() or not p.is_contiguous():
            raise LakoError("bi_score: fp32 contiguous q [B, d], p [B, n, d]")
        self._timed("bi_score", 0.0, lambda: check(self.lib.lako_bi_score(_p(q), _p(p), _p(out), B, n, d, float(scale), self._stream()), "lako_bi_score"))

    def kldiv_fwd(self, score, gold, loss):
        B, n = score.shape
        if score.dtype != torch.float32 or gold.dtype != torch.float32 or not score.is_contiguous() or not gold.is_contiguous() or gold.shape != score.shape:
            raise LakoError("kldiv_fwd: fp32 contiguous [B, n] score and gold")
        self._timed("kldiv_fwd", 0.0, lambda: check(self.lib.lako_kldiv_fwd(_p(score), _p(gold), _p(loss), B, n, self._stream()), "lako_kldiv_fwd"))

    # ---- retriever training (csrc/bertbwd.hip) ------------------------------------------------
    def layernorm_bwd(self, dy, x, gamma, dz, dgamma, dbeta, *, lin_bias=None, resid=None, dbias=None, eps=1e-12):
        """backward of layernorm_fwd (z = x + lin_bias + resid recomputed): dz; dgamma / dbeta / dbias (fp32) accumulated"""
        rows, d = x.shape
        if not (dy.is_contiguous() and x.is_contiguous() and dz.is_contiguous()) or (resid is not None and not resid.is_contiguous()):
            raise LakoError("layernorm_bwd: contiguous [rows, d] tensors")
        self._timed("layernorm_bwd", 0.0, lambda: check(self.lib.lako_layernorm_bwd(
            _p(dy), _p(x), _p(lin_bias), _p(resid), _p(gamma), _p(dz), _p(dgamma), _p(dbeta), _p(dbias), rows, d, float(eps), _dt(x),
            self._stream()), "lako_layernorm_bwd"))

    def bias_act_bwd(self, dy, x, bias, dx, dbias, gelu=False, dbias_scale=1.0):
        """dx (None: skip) = dy·act'(x + bias) on a [rows, n] column block (unit inner stride, shared row stride); dbias += scale·Σ_rows dx"""
        rows, n = dy.shape
        ld = dy.stride(0)
        if dy.stride(1) != 1 or (x is not None and (x.stride() != dy.stride() or x.shape != dy.shape)) or \
                (dx is not None and (dx.stride() != dy.stride() or dx.shape != dy.shape)):
            raise LakoError("bias_act_bwd: dy, x and dx must share shape and strides (unit inner stride)")
        self._timed("bias_act_bwd", 0.0, lambda: check(self.lib.lako_bias_act_bwd(
            _p(dy), _p(x), _p(bias), _p(dx), _p(dbias), rows, n, ld, 1 if gelu else 0, float(dbias_scale), _dt(dy), self._stream()),
            "lako_bias_act_bwd"))

    def seq_mean_bwd(self, dout, mask, dx):
        B, L, d = dx.shape
        if dout.dtype != torch.float32 or not dout.is_contiguous() or not dx.is_contiguous():
            raise LakoError("seq_mean_bwd: dout fp32 [B, d], dx contiguous [B, L, d]")
        self._timed("seq_mean_bwd", 0.0, lambda: check(self.lib.lako_seq_mean_bwd(_p(dout), _p(mask), _p(dx), B, L, d, _dt(dx), self._stream()),
                                                       "lako_seq_mean_bwd"))

    def bi_score_bwd(self, dscore, q, p, dq, dp, scale):
        B, n, d = p.shape
        for t in (dscore, q, p, dq, dp):
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise LakoError("bi_score_bwd: fp32 contiguous tensors")
        self._timed("bi_score_bwd", 0.0, lambda: check(self.lib.lako_bi_score_bwd(_p(dscore), _p(q), _p(p), _p(dq), _p(dp), B, n, d, float(scale),
                                                                                 self._stream()), "lako_bi_score_bwd"))

    def kldiv_bwd(self, score, gold, dscore, upstream=None):
        B, n = score.shape
        self._timed("kldiv_bwd", 0.0, lambda: check(self.lib.lako_kldiv_bwd(_p(score), _p(gold), _p(dscore), _p(upstream), B, n, self._stream()),
                                                    "lako_kldiv_bwd"))

    def bert_embed_bwd(self, ids, word, pos, type0, gamma, dy, dword, dpos, dtype0, dgamma, dbeta, L, eps=1e-12):
        n_tok, d = dy.shape
        self._timed("bert_embed_bwd", 0.0, lambda: check(self.lib.lako_bert_embed_bwd(
            _p(ids), _p(word), _p(pos), _p(type0), _p(gamma), _p(dy), _p(dword), _p(dpos), _p(dtype0), _p(dgamma), _p(dbeta), n_tok, int(L), d,
            word.shape[0], float(eps), _dt(dy), self._stream()), "lako_bert_embed_bwd"))

    # ---- integer helpers ------------------------------------------------------------------------
    def shift_right(self, labels, dec_ids):
        B, T = labels.shape
        self._timed("shift_right", 0.0, lambda: check(self.lib.lako_shift_right(_p(labels), _p(dec_ids), B, T, self._stream()), "lako_shift_right"))

    def pack_ids(self, ids, off, out, L):
        """out[off[j] + t] = ids[j·L + t] for the valid positions t of every passage j (ids int64 [BN·L], off int32 [BN + 1], out int64 [M])"""
        if ids.dtype != torch.int64 or out.dtype != torch.int64 or off.dtype != torch.int32 or not ids.is_contiguous() or \
                not out.is_contiguous() or not off.is_contiguous() or ids.numel() != (off.numel() - 1) * L:
            raise LakoError("pack_ids: ids int64 [BN·L], off int32 [BN + 1], out int64 [M], all contiguous")
        self._timed("pack_ids", 0.0, lambda: check(self.lib.lako_pack_ids(_p(ids), _p(off), _p(out), off.numel() - 1, int(L), self._stream()), "lako_pack_ids"))

    def greedy_step(self, logits, seq, pos, next_ids, done, n_done, eos_id=1, pad_id=0):
        B, V = logits.shape
        self._timed("greedy_step", 0.0, lambda: check(self.lib.lako_greedy_step(_p(logits), V, B, _p(seq), seq.stride(0), int(pos), _p(next_ids), _p(done),
                                        _p(n_done), int(eos_id), int(pad_id), self._stream()), "lako_greedy_step"))
