"""TEST DOUBLE of lako_amd.ops.HipOps — plain fp32 torch restatement of every C-ABI op's contract.

Lives under tests/ on purpose: it is never imported by the product.  Two uses:
  * `-m gpu` tests run each HIP kernel and this reference on the same inputs and compare;
  * `-m "not gpu"` tests inject it into lako_amd.engine to check the host-side orchestration
    (manual forward/backward, parameter layout, optimizer plumbing) against the oracle on CPU.
Works on any device; low-precision tensors are upcast to fp32, computed, and rounded on store.
"""
from __future__ import annotations

import numpy as np
import torch

M32 = 0xFFFFFFFF
FLT_MAX = float(np.finfo(np.float32).max)


def _mul32(x: torch.Tensor, c: int) -> torch.Tensor:
    """(x * c) mod 2^32 for int64 tensors holding uint32 values (no int64 overflow)."""
    lo, hi = x & 0xFFFF, x >> 16
    return (lo * c + (((hi * c) & M32) << 16)) & M32


def hash32(x: torch.Tensor) -> torch.Tensor:
    x = x ^ (x >> 16)
    x = _mul32(x, 0x7FEB352D)
    x = x ^ (x >> 15)
    x = _mul32(x, 0x846CA68B)
    return x ^ (x >> 16)


def _hash32_int(x: int) -> int:
    x &= M32
    x ^= x >> 16
    x = (x * 0x7FEB352D) & M32
    x ^= x >> 15
    x = (x * 0x846CA68B) & M32
    return x ^ (x >> 16)


def drop_key(seed: int, site: int) -> int:
    return _hash32_int((seed * 0x9E3779B9 + site * 0x85EBCA6B + 0x1234567) & M32)


def keep_mask(drop, idx: torch.Tensor):
    """(keep bool tensor, scale) for element indices `idx` (int64) — the integer recipe of csrc/common.h (lako_keep4, round 6):
    elements 4q … 4q+3 share x = lo(q) ^ key ^ hi(q)·0x27d4eb2f, h = mix(x, 0x5BD1E9, 13), w0 = mix(h, 0x6C8E95, 6),
    w1 = mix(h, 0x1B873B, 11) with mix(a, c, s) = t ^ (t >> 16), t = (a mod 2^24)·c + (a >> s) mod 2^32; 16-bit draws w0>>16, w0&0xffff,
    w1>>16, w1&0xffff; keep iff draw >= round(p·65536)."""
    p, seed, site = drop
    t16 = min(max(int(float(np.float32(p)) * 65536.0 + 0.5), 1), 65535)
    key = drop_key(int(seed) & M32, int(site) & M32)
    q, fld = idx >> 2, idx & 3
    lo, hi = q & M32, q >> 32
    h = _drop_mix(lo ^ key ^ ((hi * 0x27D4EB2F) & M32), 0x5BD1E9, 13)
    w0, w1 = _drop_mix(h, 0x6C8E95, 6), _drop_mix(h, 0x1B873B, 11)
    word = torch.where(fld >= 2, w1, w0)
    draw = torch.where((fld & 1).bool(), word & 0xFFFF, word >> 16)
    scale = float(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))
    return draw >= t16, scale


def _drop_mix(a, c, s):
    t = ((a & 0xFFFFFF) * c + (a >> s)) & M32
    return t ^ (t >> 16)


DROP_C0 = 0x5BD1E9
DROP_MUL = [0x6C8E95, 0x1B873B, 0x4F1BBD, 0x7A3C6F, 0x35D2A7, 0x59E4C1, 0x2545F5, 0x63D9AB]      # [i·2 + j]


def attn_keep_mask(BH, Lq, Lk, key, p, dev="cpu"):
    """keep[bh, q, k] of the attention-probability dropout — the integer recipe of csrc/attn_shared.h (drop_base / drop_mix)."""
    QB, KB = (Lq + 3) // 4, (Lk + 3) // 4
    bh = torch.arange(BH, device=dev, dtype=torch.int64).view(BH, 1, 1)
    q = torch.arange(Lq, device=dev, dtype=torch.int64).view(1, Lq, 1)
    k = torch.arange(Lk, device=dev, dtype=torch.int64).view(1, 1, Lk)
    blk = ((bh * QB + (q >> 2)) * KB + (k >> 2)) & M32
    h = _drop_mix(blk ^ key, DROP_C0, 13)
    i, j = (q & 3).expand(BH, Lq, Lk), ((k & 3) >> 1).expand(BH, Lq, Lk)
    mul = torch.tensor(DROP_MUL, device=dev, dtype=torch.int64)[i * 2 + j]
    w = ((h & 0xFFFFFF) * mul + (h >> (6 + 2 * i + 5 * j))) & M32
    w = w ^ (w >> 16)
    draw = torch.where((k & 1).bool(), w & 0xFFFF, w >> 16)
    t16 = min(max(int(float(np.float32(p)) * 65536.0 + 0.5), 1), 65535)
    return draw >= t16


def _on(d):
    return d is not None and d[0] > 0.0


def _apply_drop(x: torch.Tensor, drop, idx=None):
    if not _on(drop):
        return x
    if idx is None:
        idx = torch.arange(x.numel(), device=x.device, dtype=torch.int64).view(x.shape)
    keep, scale = keep_mask(drop, idx)
    return torch.where(keep, x * scale, torch.zeros_like(x))


def f(t):
    return t.float()


class RefOps:
    name = "ref"

    def zero_(self, t):
        t.zero_()

    # ---- GEMMs -----------------------------------------------------------------------------
    def gemm_nt(self, A, B, Cm, *, alpha=1.0, relu=False, resid=None, aux=None, aux_scale=1.0, drop=None,
                atomic=False, norm=None):
        if norm is not None:          # A is the un-normalised input: the product runs on the RMSNorm'd rows, which are also handed back
            w, eps, xn, rstd = norm
            self.rmsnorm_fwd(A, w, xn, rstd, eps)
            A = xn
        v = (f(A) @ f(B).t()) * alpha
        if relu:
            v = torch.relu(v)
        if aux is not None:
            v = torch.where(f(aux) > 0, v * aux_scale, torch.zeros_like(v))
        v = _apply_drop(v, drop)
        if resid is not None:
            v = v + f(resid)
        if atomic:
            Cm += v
        else:
            Cm.copy_(v)

    def gemm_tn(self, A, B, Cm, *, alpha=1.0, split_k=0, rows_out=0):
        v = (f(A).t() @ f(B)) * alpha
        r = rows_out or Cm.shape[0]
        if split_k == -2:       # overwrite
            Cm[:r].copy_(v[:r])
        else:
            Cm[:r] += v[:r]

    def gemm_tn_grouped(self, problems, split_k=0, workspace=None):
        for prob in problems:
            A, B, Cm, alpha = prob[:4]
            self.gemm_tn(A, B, Cm, alpha=alpha, split_k=split_k, rows_out=prob[4] if len(prob) > 4 else 0)

    # ---- norm / embedding / dropout ------------------------------------------------------------
    def rmsnorm_fwd(self, x, w, y, rstd, eps, drop=None):
        xf = f(x)
        rs = torch.rsqrt(xf.pow(2).mean(-1) + eps)
        rstd.copy_(rs)
        y.copy_(_apply_drop(w * (xf * rs[:, None]), drop))

    def rmsnorm_bwd(self, dy, x, w, rstd, dres, dx, dw, drop=None, dx_drop=None, drop_out=None):
        g = _apply_drop(f(dy), drop)
        xf, rs = f(x), rstd[:, None]
        d = xf.shape[1]
        s = (w * g * xf).sum(-1, keepdim=True)
        out = rs * w * g - xf * (rs ** 3) * s / d
        if dres is not None:
            out = out + f(dres)
        dx.copy_(out)
        dw += (g * xf * rs).sum(0)
        if dx_drop is not None:
            self.dropout_apply(dx, dx_drop, drop_out)

    def embed_fwd(self, ids, table, out, drop=None):
        out.copy_(_apply_drop(f(table)[ids.reshape(-1)], drop).view(out.shape))

    def embed_bwd(self, ids, dout, dtable, drop=None):
        g = _apply_drop(f(dout).reshape(-1, dtable.shape[1]), drop)
        dtable.index_add_(0, ids.reshape(-1), g)

    def dropout_apply(self, x, y, drop):
        y.copy_(_apply_drop(f(x).reshape(-1), drop).view(y.shape))

    # ---- relative position bias --------------------------------------------------------------
    def relpos_expand(self, table, lut, rel):
        rel.copy_(table[lut.long()].t())

    def relpos_reduce(self, drel, lut, dtable):
        dtable.index_add_(0, lut.long(), drel.t().contiguous())

    # ---- attention ------------------------------------------------------------------------------
    @staticmethod
    def _attn_core(q, k, v, rel_bias, rel_off, key_mask, causal, causal_off, drop):
        """q,k,v fp32 [B,L,H,dk] → (out [B,Lq,H,dk], m, inv_l, raw scores, masked bool)."""
        B, Lq, H, dk = q.shape
        Lk = k.shape[1]
        dev = q.device
        s = torch.einsum("bqhd,bkhd->bhqk", q, k)
        i = torch.arange(Lq, device=dev)[:, None]
        j = torch.arange(Lk, device=dev)[None, :]
        if rel_bias is not None:
            bi = j - i + rel_off            # an index outside the table takes its nearest entry (include/lako_hip.h: a one-sided table —
            bias = rel_bias[:, bi.clamp(0, rel_bias.shape[1] - 1)]      # the legacy cross-attention bias — relies on it)
            s = s + bias[None]
        masked = torch.zeros(B, 1, Lq, Lk, dtype=torch.bool, device=dev)
        if key_mask is not None:
            masked = masked | ~key_mask.bool()[:, None, None, :]
        if causal:
            masked = masked | (j > i + causal_off)[None, None]
        masked = masked.expand(B, H, Lq, Lk)
        raw = torch.where(masked, torch.zeros_like(s), s)
        # HF ADDS finfo.min (HF5:163-164): s + min rounds to min, so a fully masked row becomes uniform AND
        # the gradient still flows to the scores of masked keys (torch.where would block it)
        sm = s + masked.float() * (-FLT_MAX)
        m = sm.max(-1).values
        p = torch.exp(sm - m[..., None])
        l = p.sum(-1)
        pn = p / l[..., None]
        if _on(drop):
            # attention-probability dropout (csrc/attn_shared.h): the 4×4 block (queries 4a…, keys 4c…) of head-row bh shares
            # h = mix(((bh·QB + a)·KB + c) ^ key, C0, 13); element (q, k) draws 16 bits of W = mix(h, M[q&3][(k&3)>>1], S[…]):
            # high half for even k, low half for odd k; mix(a, c, s) = t ^ (t >> 16), t = (a mod 2^24)·c + (a >> s) mod 2^32
            pr, seed, site = drop
            keep = attn_keep_mask(B * H, Lq, Lk, drop_key(int(seed) & M32, int(site) & M32), pr, dev).view(B, H, Lq, Lk)
            scale = float(np.float32(1.0) / (np.float32(1.0) - np.float32(pr)))
            pn = torch.where(keep, pn * scale, torch.zeros_like(pn))
        out = torch.einsum("bhqk,bkhd->bqhd", pn, v)
        return out, m, 1.0 / l, raw

    # Ragged sequences (q_off / k_off, see include/lako_hip.h): by definition the ragged call equals the PADDED call
    # on the rows that exist, so the double pads the packed [1, rows, H, dk] tensors to [Bn, Lmax, H, dk], derives the
    # key mask from the lengths, runs the padded math and copies the existing rows back.
    @staticmethod
    def _pad(t, off, Lmax):
        n = off.numel() - 1
        o = off.tolist()
        out = torch.zeros(n, Lmax, *t.shape[2:], dtype=torch.float32, device=t.device)
        for b in range(n):
            out[b, :o[b + 1] - o[b]] = f(t[0, o[b]:o[b + 1]])
        return out

    @staticmethod
    def _unpad_into(dst, src, off):
        o = off.tolist()
        for b in range(len(o) - 1):
            dst[0, o[b]:o[b + 1]] = src[b, :o[b + 1] - o[b]].to(dst.dtype)

    @staticmethod
    def _len_mask(off, Lmax):
        lens = (off[1:] - off[:-1]).long()
        return torch.arange(Lmax, device=off.device)[None, :] < lens[:, None]

    def attn_fwd(self, q, k, v, out, stats, *, rel_bias=None, rel_off=0, key_mask=None, causal=False, causal_off=0,
                 drop=None, scores_out=None, q_off=None, k_off=None, max_q=None, max_k=None, order=None):
        qf, kf, vf = f(q), f(k), f(v)
        if q_off is not None:
            qf = self._pad(q, q_off, max_q)
        if k_off is not None:
            assert key_mask is None
            kf, vf, key_mask = self._pad(k, k_off, max_k), self._pad(v, k_off, max_k), self._len_mask(k_off, max_k)
        o, m, il, raw = self._attn_core(qf, kf, vf, rel_bias, rel_off, key_mask, causal, causal_off, drop)
        if q_off is not None:
            self._unpad_into(out, o, q_off)
        else:
            out.copy_(o)
        stats[..., 0] = m
        stats[..., 1] = il
        if scores_out is not None:
            scores_out.copy_(raw)

    def attn_bwd(self, q, k, v, out, dout, stats, dq, dk_, dv, *, rel_bias=None, drel=None, rel_off=0, key_mask=None,
                 causal=False, causal_off=0, drop=None, q_off=None, k_off=None, max_q=None, max_k=None, order=None):
        qp, kp, vp, dop = f(q), f(k), f(v), f(dout)
        if q_off is not None:
            qp, dop = self._pad(q, q_off, max_q), self._pad(dout, q_off, max_q)
        if k_off is not None:
            assert key_mask is None
            kp, vp, key_mask = self._pad(k, k_off, max_k), self._pad(v, k_off, max_k), self._len_mask(k_off, max_k)
        with torch.enable_grad():
            qf, kf, vf = (t.detach().clone().requires_grad_(True) for t in (qp, kp, vp))
            rb = rel_bias.detach().clone().requires_grad_(True) if rel_bias is not None else None
            o, _, _, _ = self._attn_core(qf, kf, vf, rb, rel_off, key_mask, causal, causal_off, drop)
            ins = [qf, kf, vf] + ([rb] if rb is not None else [])
            gs = torch.autograd.grad(o, ins, dop)
        if q_off is not None:
            self._unpad_into(dq, gs[0], q_off)
        else:
            dq.copy_(gs[0])
        if k_off is not None:
            self._unpad_into(dk_, gs[1], k_off)
            self._unpad_into(dv, gs[2], k_off)
        else:
            dk_.copy_(gs[1])
            dv.copy_(gs[2])
        if drel is not None:
            drel += gs[3]

    # ---- cross-attention in the encoder-state space (csrc/xattn.hip) --------------------------------
    @staticmethod
    def xattn_ok(dtype, d_kv, d_model):
        return True

    def headbatch_nt(self, A, Bw, Cm):
        if A.dim() == 5:           # key-split slabs, added in order
            A = A.sum(0)
        Cm.copy_(torch.einsum("bthk,hnk->bthn", f(A).to(Bw.dtype).float(), f(Bw)))

    def headbatch_tn(self, A, Bm, Cw):
        if Bm.dim() == 5:
            Bm = Bm.sum(0)
        Cw += torch.einsum("bthj,bthc->hjc", f(A), f(Bm).to(A.dtype).float())

    def headbatch_tn_multi(self, problems):
        for A, Bm, Cw in problems:
            self.headbatch_tn(A, Bm, Cw)

    def xattn_scores(self, Q, E, k_off, p_off, p_total, S):
        ko, po = k_off.tolist(), p_off.tolist()
        for b in range(Q.shape[0]):
            n = ko[b + 1] - ko[b]
            S[:, po[b]:po[b + 1]] = 0
            S[:, po[b]:po[b] + n] = f(Q[b]) @ f(E[ko[b]:ko[b + 1]]).T

    def xattn_context(self, P, E, k_off, p_off, out):
        ko, po = k_off.tolist(), p_off.tolist()
        out.zero_()                # [Z, B, R, D]: the double puts everything into slab 0
        for b in range(out.shape[1]):
            n = ko[b + 1] - ko[b]
            out[0, b] = f(P[:, po[b]:po[b] + n]) @ f(E[ko[b]:ko[b + 1]])

    @staticmethod
    def xattn_decode_ok(H, d_model):
        return H <= 16

    def xattn_decode(self, Q, E, k_off, part_ml, part_c):
        ko = k_off.tolist()
        Z, Bz, _, D = part_c.shape
        R = Q.shape[1]
        part_ml[..., 0] = -float("inf")
        part_ml[..., 1] = 0
        part_c.zero_()
        for b in range(Bz):
            n = ko[b + 1] - ko[b]
            nst = -(-n // 32)
            for z in range(Z):
                lo, hi = min(n, (z * nst // Z) * 32), min(n, ((z + 1) * nst // Z) * 32)
                if hi <= lo:
                    continue
                e = f(E[ko[b] + lo:ko[b] + hi])
                s = f(Q[b]) @ e.T
                m = s.max(-1).values
                p = torch.exp(s - m[:, None])
                part_ml[z, b, :R, 0] = m
                part_ml[z, b, :R, 1] = p.sum(-1)
                part_c[z, b, :R] = p.to(Q.dtype).float() @ e

    def xattn_decode_combine(self, part_ml, part_c, Wv, ctx, H):
        Z, Bz, _, D = part_c.shape
        m = part_ml[:, :, :H, 0]
        M = m.max(0).values
        w = torch.where(torch.isinf(m), torch.zeros_like(m), torch.exp(m - M[None]))
        L = (w * part_ml[:, :, :H, 1]).sum(0)
        c = (w[..., None] * part_c[:, :, :H]).sum(0) / L[..., None]              # [B, H, D]
        ctx.copy_(torch.einsum("bhc,hjc->bhj", c, f(Wv).view(H, 64, D)).reshape(Bz, H * 64))

    @staticmethod
    def _xkeep(b, T, H, n, max_keys, drop, dev):
        """keep[r = t·H + h, s] and the scale of sample b (the attention recipe with bh = b·H + h, q = t, k = s, Lk = max_keys)"""
        pr, seed, site = drop
        BH = (b + 1) * H
        keep = attn_keep_mask(BH, T, max_keys, drop_key(int(seed) & M32, int(site) & M32), pr, dev)[b * H:, :, :n]   # [H, T, n]
        return keep.permute(1, 0, 2).reshape(T * H, n), float(np.float32(1.0) / (np.float32(1.0) - np.float32(pr)))

    def xattn_softmax_fwd(self, S, stats, P, k_off, p_off, T, H, max_keys, drop=None):
        ko, po = k_off.tolist(), p_off.tolist()
        for b in range(stats.shape[0]):
            n = ko[b + 1] - ko[b]
            s = f(S[:, po[b]:po[b] + n])
            m = s.max(-1).values
            e = torch.exp(s - m[:, None])
            l = e.sum(-1)
            pn = e / l[:, None]
            if _on(drop):
                keep, scale = self._xkeep(b, T, H, n, max_keys, drop, S.device)
                pn = torch.where(keep, pn * scale, torch.zeros_like(pn))
            P[:, po[b]:po[b + 1]] = 0
            P[:, po[b]:po[b] + n] = pn.to(P.dtype)
            stats[b, :, 0] = m
            stats[b, :, 1] = 1.0 / l

    def xattn_softmax_bwd(self, S, dP, stats, dS, k_off, p_off, T, H, max_keys, drop=None):
        ko, po = k_off.tolist(), p_off.tolist()
        for b in range(stats.shape[0]):
            n = ko[b + 1] - ko[b]
            pn = torch.exp(f(S[:, po[b]:po[b] + n]) - stats[b, :, 0:1]) * stats[b, :, 1:2]
            d = f(dP[:, po[b]:po[b] + n])
            if _on(drop):
                keep, scale = self._xkeep(b, T, H, n, max_keys, drop, S.device)
                d = torch.where(keep, d * scale, torch.zeros_like(d))
            delta = (pn * d).sum(-1, keepdim=True)
            dS[:, po[b]:po[b + 1]] = 0
            dS[:, po[b]:po[b] + n] = (pn * (d - delta)).to(dS.dtype)

    # ---- loss / optimizer -----------------------------------------------------------------------
    def ce_fwd_bwd(self, logits, labels, loss_out, dlogits, upstream=None):
        valid = labels != -100
        n = valid.sum().float()
        lse = torch.logsumexp(logits, -1)
        tgt = logits.gather(1, labels.clamp(min=0)[:, None])[:, 0]
        loss_out[0] = ((lse - tgt) * valid).sum() / n
        loss_out[1] = n
        if dlogits is not None:
            p = torch.softmax(logits, -1)
            p[torch.arange(len(labels), device=labels.device), labels.clamp(min=0)] -= 1.0
            up = upstream[0] if upstream is not None else 1.0
            dlogits.copy_(p * (valid[:, None] / n) * up)

    def sumsq(self, g, out):
        out += (g.double() ** 2).sum().float()

    def adamw_step(self, p, g, m, v, shadow, *, lr, beta1, beta2, eps, weight_decay, gnorm_sq, max_norm, grad_scale):
        coef = grad_scale
        if gnorm_sq is not None:
            total = torch.sqrt(gnorm_sq[0]) * grad_scale
            coef = coef * torch.clamp(max_norm / (total + 1e-6), max=1.0)
        gg = g * coef
        m.mul_(beta1).add_(gg, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(gg, gg, value=1 - beta2)
        p.sub_(lr * (m / (v.sqrt() + eps)))
        if weight_decay > 0:
            p.sub_(lr * weight_decay * p)
        if shadow is not None:
            shadow.copy_(p)

    def transpose_cast(self, src, dst):
        dst.copy_(src.t())

    def transpose_cast_batched(self, src_flat, dst_flat, desc, tile_prefix, n, total_tiles):
        d = desc.view(-1, 4).tolist()
        assert len(d) == n
        for so, do, rows, cols in d:
            dst_flat[do:do + rows * cols].view(cols, rows).copy_(src_flat[so:so + rows * cols].view(rows, cols).t())

    def cast(self, src, dst):
        dst.copy_(src.view(dst.shape))

    def topk(self, scores, k, out_vals, out_idx):
        # value descending, index ascending among equal values (a stable sort of the negated scores)
        order = torch.sort(-scores, dim=1, stable=True).indices[:, :k]
        out_idx.copy_(order)
        out_vals.copy_(torch.gather(scores, 1, order))

    # ---- product quantiser (csrc/pq.hip) ------------------------------------------------------------------
    def pq_assign(self, x, centroids, codes=None, sums=None, counts=None, err=None):
        M, ksub, dsub = centroids.shape
        xs = x[:, :M * dsub].reshape(x.shape[0], M, dsub)
        d2 = ((xs[:, :, None, :] - centroids[None]) ** 2).sum(-1)             # [n, M, ksub]
        best = d2.argmin(-1)                                                  # lowest index on ties
        if codes is not None:
            codes.copy_(best.to(torch.uint8))
        if sums is not None:
            for m in range(M):
                sums[m].index_add_(0, best[:, m], xs[:, m])
                counts[m] += torch.bincount(best[:, m], minlength=ksub).to(torch.int32)
        if err is not None:
            err += d2.gather(-1, best[..., None]).sum()

    def pq_lut(self, q, centroids, lut):
        M, ksub, dsub = centroids.shape
        lut.copy_(torch.einsum("qmj,mcj->qmc", q[:, :M * dsub].reshape(q.shape[0], M, dsub), centroids))

    def pq_scan(self, lut, codes, scores):
        acc = torch.zeros(lut.shape[0], codes.shape[0], dtype=torch.float32)
        for m in range(lut.shape[1]):                                         # ascending m, fp32: the kernel's order
            acc += lut[:, m, :][:, codes[:, m].long()]
        scores.copy_(acc)

    # ---- retriever bi-encoder forward ------------------------------------------------------------------
    def layernorm_fwd(self, x, gamma, beta, y, *, lin_bias=None, resid=None, eps=1e-12):
        v = x.float()
        if lin_bias is not None:
            v = v + lin_bias.float()
        if resid is not None:
            v = v + resid.float()
        y.copy_(torch.nn.functional.layer_norm(v, (v.shape[-1],), gamma.float(), beta.float(), eps))

    def bert_embed(self, ids, word, pos, type0, gamma, beta, out, L, eps=1e-12):
        n_tok, d = out.shape
        ids = ids.reshape(-1)
        ids = torch.where((ids < 0) | (ids >= word.shape[0]), torch.zeros_like(ids), ids)
        v = word[ids].float() + pos[torch.arange(n_tok, device=ids.device) % L].float() + type0.float()
        out.copy_(torch.nn.functional.layer_norm(v, (d,), gamma.float(), beta.float(), eps))

    def bias_act(self, x, bias, y, gelu=False):
        v = x.float() + bias.float()
        y.copy_(torch.nn.functional.gelu(v) if gelu else v)

    def seq_mean(self, x, mask, out):
        v = x.float()
        if mask is None:
            out.copy_(v.mean(1))
        else:
            m = mask.bool()
            out.copy_(v.masked_fill(~m[:, :, None], 0.0).sum(1) / m.sum(1, keepdim=True).float())

    def bi_score(self, q, p, out, scale):
        out.copy_(torch.einsum("bd,bid->bi", q, p) * scale)

    def kldiv_fwd(self, score, gold, loss):
        loss[0] = torch.nn.KLDivLoss()(torch.log_softmax(score, -1), gold)

    # ---- retriever training (test doubles of csrc/bertbwd.hip: torch autograd of the forward doubles above) -------------------
    @torch.enable_grad()
    def layernorm_bwd(self, dy, x, gamma, dz, dgamma, dbeta, *, lin_bias=None, resid=None, dbias=None, eps=1e-12):
        z = x.float()
        if lin_bias is not None:
            z = z + lin_bias.float()
        if resid is not None:
            z = z + resid.float()
        z = z.detach().requires_grad_(True)
        g = gamma.float().detach().requires_grad_(True)
        b = torch.zeros_like(g).requires_grad_(True)
        y = torch.nn.functional.layer_norm(z, (z.shape[-1],), g, b, eps)
        gz, gg, gb = torch.autograd.grad(y, [z, g, b], dy.float())
        dz.copy_(gz)
        dgamma += gg
        dbeta += gb
        if dbias is not None:
            dbias += gz.sum(0)

    @torch.enable_grad()
    def bias_act_bwd(self, dy, x, bias, dx, dbias, gelu=False, dbias_scale=1.0):
        g = dy.float()
        if gelu:
            v = (x.float() + bias.float()).detach().requires_grad_(True)
            g, = torch.autograd.grad(torch.nn.functional.gelu(v), v, g)
        if dx is not None:
            dx.copy_(g)
        dbias += g.sum(0) * dbias_scale

    def seq_mean_bwd(self, dout, mask, dx):
        B, L, d = dx.shape
        if mask is None:
            dx.copy_((dout[:, None, :] / L).expand(B, L, d))
        else:
            m = mask.bool()
            dx.copy_(torch.where(m[:, :, None], dout[:, None, :] / m.sum(1).float()[:, None, None], torch.zeros((), dtype=dout.dtype)))

    def bi_score_bwd(self, dscore, q, p, dq, dp, scale):
        dp.copy_(dscore[:, :, None] * q[:, None, :] * scale)
        dq.copy_(torch.einsum("bi,bid->bd", dscore, p) * scale)

    @torch.enable_grad()
    def kldiv_bwd(self, score, gold, dscore, upstream=None):
        sc = score.detach().clone().requires_grad_(True)
        loss = torch.nn.KLDivLoss()(torch.log_softmax(sc, -1), gold)
        g, = torch.autograd.grad(loss, sc)
        dscore.copy_(g * (upstream[0] if upstream is not None else 1.0))

    @torch.enable_grad()
    def bert_embed_bwd(self, ids, word, pos, type0, gamma, dy, dword, dpos, dtype0, dgamma, dbeta, L, eps=1e-12):
        n_tok, d = dy.shape
        ids = ids.reshape(-1)
        ids = torch.where((ids < 0) | (ids >= word.shape[0]), torch.zeros_like(ids), ids)
        p_idx = torch.arange(n_tok, device=ids.device) % L
        z = (word[ids].float() + pos[p_idx].float() + type0.float()).detach().requires_grad_(True)
        g = gamma.float().detach().requires_grad_(True)
        b = torch.zeros_like(g).requires_grad_(True)
        y = torch.nn.functional.layer_norm(z, (d,), g, b, eps)
        gz, gg, gb = torch.autograd.grad(y, [z, g, b], dy.float())
        dword.index_add_(0, ids, gz)
        dpos.index_add_(0, p_idx, gz)
        dtype0 += gz.sum(0)
        dgamma += gg
        dbeta += gb

    # ---- integer helpers ------------------------------------------------------------------------
    def shift_right(self, labels, dec_ids):
        dec_ids[:, 0] = 0
        dec_ids[:, 1:] = labels[:, :-1]
        dec_ids.masked_fill_(dec_ids == -100, 0)

    def pack_ids(self, ids, off, out, L):
        o = off.tolist()
        for j in range(len(o) - 1):
            out[o[j]:o[j + 1]] = ids[j * L:j * L + (o[j + 1] - o[j])]

    def greedy_step(self, logits, seq, pos, next_ids, done, n_done, eos_id=1, pad_id=0):
        nxt = logits.argmax(-1)
        nxt = torch.where(done.bool(), torch.full_like(nxt, pad_id), nxt)
        seq[:, pos] = nxt
        next_ids.copy_(nxt)
        done |= (nxt == eos_id).to(done.dtype)
        n_done[0] = int(done.sum())

    # ---- per-fact aggregation (test double of lako_fact_scores: the reference's own loops, src/model.py:100-115,170-199) ----
    def fact_scores(self, scores, mask, ids, out, *, layer0, layers_used, passage, style, ids_passage=None):
        import heapq
        ids_passage = passage if ids_passage is None else ids_passage
        B, H, nl, S = scores.shape
        _, N, L = ids.shape
        s = scores.view(B, H, nl, N, L)[:, :, layer0:layer0 + layers_used].masked_fill(~mask.bool()[:, None, None], 0.0)
        fact = s[:, :, :, passage].sum(dim=[1, 2])

        def span(vals, a, b):
            if style == "mean":
                return sum(vals[a:b]) / (b - a)
            if style == "max":
                return max(vals[a:b])
            num = max(int((b - a + 1) / 2), 1)
            return sum(heapq.nlargest(num, vals[a:b])) / num
        n_ctx = out.shape[1]
        for b in range(B):
            toks, vals = ids[b][ids_passage].tolist(), fact[b].tolist()
            res, start = [], 2
            for _ in range(n_ctx):
                try:
                    end = toks.index(5, start) + 1
                except ValueError:
                    break
                res.append(span(vals, start, end))
                start = end
            if len(res) < n_ctx and toks[-1] != 0 and len(toks) > start:
                res.append(span(vals, start, len(toks)))
            res += [-5] * (n_ctx - len(res))
            out[b] = torch.tensor(res, dtype=torch.float64) / (layers_used * H)


# ---- MX (OCP microscaling) e4m3: the integer recipe of csrc/gemm.hip::mx_quantize_kernel, for the kernel tests ---------------
def mx_quantize_ref(x: torch.Tensor):
    """x float [rows, K] → (q float [rows, K] = the e4m3-rounded scaled elements, e int [rows, K/32] = E8M0 scale bytes)."""
    rows, K = x.shape
    xb = x.float().view(rows, K // 32, 32)
    amax = xb.abs().amax(-1)
    ex = ((amax.view(torch.int32) >> 23) & 0xFF) - 8
    ex = ex.clamp(0, 254)
    inv = ((254 - ex) << 23).to(torch.int32).view(torch.float32)
    t = (xb * inv[..., None]).clamp(-448.0, 448.0)
    q = t.to(torch.float8_e4m3fn).float()                 # OCP e4m3, round to nearest even
    return q.view(rows, K), ex


def mx_dequant_ref(q: torch.Tensor, ex: torch.Tensor):
    rows, K = q.shape
    scale = torch.ldexp(torch.ones_like(ex, dtype=torch.float32), ex - 127)
    return (q.view(rows, K // 32, 32) * scale[..., None]).view(rows, K)


def mx_scales_layout(ex: torch.Tensor, K: int):
    """[rows, K/32] block exponents → the library's [rows, 4, KSP] byte layout"""
    rows = ex.shape[0]
    ksp = ((K // 128) + 3) // 4 * 4
    out = torch.zeros(rows, 4, ksp, dtype=torch.uint8)
    blk = torch.arange(K // 32)
    out[:, blk % 4, blk // 4] = ex.to(torch.uint8)
    return out.view(rows, 4 * ksp)
