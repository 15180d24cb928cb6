"""TRAINING of the BERT bi-encoder retriever — SURVEY.md §8 f4, the half round 2 left out: the reference's
`train_retriever.py:37-71` step (`model(..., gold_score=…)[3].backward()`, clip, AdamW, scheduler) on `src.model.Retriever`
(src/model.py:375-483: KLDivLoss of log_softmax(q·p/√d) against the reader's per-fact cross-attention scores), written out by
hand over the C-ABI like the reader's backward (engine.py):

  forward (training)   the inference schedule of retriever.py plus what the backward needs: per layer the block input, the fused
                       q|k|v, the attention output and softmax statistics, the two pre-LayerNorm sums, the FFN pre-activation and
                       activation; dropout (`hidden_dropout_prob` after the embeddings and after both dense outputs,
                       `attention_probs_dropout_prob` on the attention probabilities — what `src.util.set_dropout` sets on the
                       reference's nn.Dropout modules) by the library's stateless recipes: the backward regenerates the masks
  backward             KLDiv → bi-encoder scores → pooling → (projection + LayerNorm) → BERT blocks last → first → embeddings,
                       once for the passage tower and once for the question tower (shared weights: gradients accumulate);
                       matrix products = lako_gemm_nt (dX = dY·W through transposed weight copies, residual gradients added by the
                       RESID epilogue) and lako_gemm_tn (dW), attention = lako_attn_bwd, everything row-wise = csrc/bertbwd.hip
  parameters           one flat fp32 master buffer P and gradient buffer G (views as nn.Parameters / .grad, like FiDT5), so that
                       `util.set_optim` / `util.clip_grad_norm_` — the fused AdamW(no bias correction) of the reader — apply unchanged

No CPU path: like the forward, this needs the HIP library (tests drive the host logic through the fp32 test double).
"""
from __future__ import annotations

import math

import torch

S_EMB = 1


def _site(layer: int, k: int, tower: int) -> int:      # k: 0 attention probs, 1 attention output, 2 FFN output
    return 100 + (layer * 4 + k) * 2 + tower


class _Tower:
    """what one encoder pass (question tower 0 / passage tower 1) leaves behind for the backward"""

    def __init__(self):
        self.layers = []


class RetrieverEngine:
    """The part of `lako_amd.engine.Engine`'s surface the fused optimizer and the clip use (util.AdamW / clip_grad_norm_):
    flat fp32 P / G / moments, the op set, and a hook after the optimizer step (the packed compute-dtype weights are stale)."""

    def __init__(self, model, ops, device):
        self.model, self.ops, self.device = model, ops, torch.device(device)
        self.G_all = torch.zeros_like(model._master)
        self.P = model._master[:model._n_train]      # the trained range (everything but the unused pooler)
        self.G = self.G_all[:model._n_train]
        self.W = self.P                      # no flat low-precision shadow: the pack casts per matrix
        self.opt_m = self.opt_v = None
        self.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.shadows_stale = False
        self.step_count = 0
        self.fp8 = False
        self.grad_hook = None

    def refresh_after_step(self):
        self.model._pack = None              # weights changed under the packed copies
        self.model._master_version += 1

    def zero_grad(self):
        self.ops.zero_(self.G)


class _KLLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, trainer, loss):
        ctx.trainer = trainer
        return loss.clone()

    @staticmethod
    def backward(ctx, gloss):
        ctx.trainer.backward(gloss.detach().float().reshape(1).contiguous())
        return None, None, None


class RetrieverTrainer:
    def __init__(self, model):
        self.m = model
        self.ctx = None

    # ------------------------------------------------------------------------------------------------------------------
    def _encode(self, tower: int, text_ids, text_mask, text_type, apply_mask, extract_cls, p_h, p_a, seed):
        m = self.m
        ops, pk, c, cd = m._get_ops(), m._packed(train=True), m.config, m.compute_dtype
        B, L = text_ids.shape
        dev = pk["word"].device
        d, H, f = c.hidden_size, c.num_attention_heads, c.intermediate_size
        dk, M = d // H, B * L
        T = _Tower()
        T.B, T.L, T.tower, T.text_type = B, L, tower, text_type
        T.ids = text_ids.to(dev, torch.int64).contiguous()
        T.km = text_mask.to(dev).ne(0).to(torch.uint8).contiguous() if apply_mask else None
        new = lambda *s, dt=cd: torch.empty(*s, dtype=dt, device=dev)       # noqa: E731
        x = new(M, d)
        ops.bert_embed(T.ids.view(-1), pk["word"], pk["pos"], pk["type0"], pk["emb_g"], pk["emb_b"], x, L, c.layer_norm_eps)
        if p_h > 0:
            ops.dropout_apply(x, x, (p_h, seed, S_EMB * 2 + tower))
        for i, ly in enumerate(pk["layers"]):
            A = dict(x=x)
            qkv = new(M, 3 * d)
            ops.gemm_nt(x, ly["wqkv"], qkv)
            ops.bias_act(qkv, ly["bqkv"], qkv)
            q4 = qkv.view(B, L, 3, H, dk)
            ctx = new(M, d)
            stats = new(B, H, L, 4, dt=torch.float32)
            ops.attn_fwd(q4[:, :, 0], q4[:, :, 1], q4[:, :, 2], ctx.view(B, L, H, dk), stats, key_mask=T.km,
                         drop=(p_a, seed, _site(i, 0, tower)) if p_a > 0 else None)
            u1 = new(M, d)
            ops.gemm_nt(ctx, ly["wo"], u1)
            ops.bias_act(u1, ly["bo"], u1)
            if p_h > 0:
                ops.dropout_apply(u1, u1, (p_h, seed, _site(i, 1, tower)))
            x1 = new(M, d)
            ops.layernorm_fwd(u1, ly["g1"], ly["b1"], x1, resid=x, eps=c.layer_norm_eps)
            hraw, h = new(M, f), new(M, f)
            ops.gemm_nt(x1, ly["wi"], hraw)
            ops.bias_act(hraw, ly["bi"], h, gelu=True)
            u2 = new(M, d)
            ops.gemm_nt(h, ly["wo2"], u2)
            ops.bias_act(u2, ly["bo2"], u2)
            if p_h > 0:
                ops.dropout_apply(u2, u2, (p_h, seed, _site(i, 2, tower)))
            x2 = new(M, d)
            ops.layernorm_fwd(u2, ly["g2"], ly["b2"], x2, resid=x1, eps=c.layer_norm_eps)
            A.update(qkv=qkv, ctx=ctx, stats=stats, u1=u1, x1=x1, hraw=hraw, h=h, u2=u2)
            T.layers.append(A)
            x = x2
        T.x_final = x
        head = None
        if c.projection:
            head = "proj"
        elif c.asymmetric_retri == "yes":
            head = "proj_iq" if text_type == "q" else "proj_fact"
        T.head = head
        if head is not None:
            wp, bp, g, b = pk[head]
            t = new(M, wp.shape[0])
            ops.gemm_nt(x, wp, t)
            xh = new(M, wp.shape[0])
            ops.layernorm_fwd(t, g, b, xh, lin_bias=bp, eps=1e-5)
            T.head_t = t
            x = xh
        dim = x.shape[1]
        if extract_cls:
            sel = torch.zeros(B, L, dtype=torch.uint8, device=dev)
            sel[:, 0] = 1
            T.pool_mask = sel
        elif apply_mask:
            T.pool_mask = text_mask.to(dev).ne(0).to(torch.uint8).contiguous()
        else:
            T.pool_mask = None
        out = torch.empty(B, dim, dtype=torch.float32, device=dev)
        ops.seq_mean(x.view(B, L, dim), T.pool_mask, out)
        T.dim = dim
        return out, T

    def forward(self, question_ids, question_mask, passage_ids, passage_mask, gold_score):
        m = self.m
        c, ops = m.config, m._get_ops()
        eng = m._get_engine()
        p_h = float(c.hidden_dropout_prob) if m.training else 0.0
        p_a = float(c.attention_probs_dropout_prob) if m.training else 0.0
        seed = (m._seed * 1000003 + eng.step_count) & 0xFFFFFFFF
        bsz, n_passages, plen = passage_ids.shape
        q, Tq = self._encode(0, question_ids, question_mask, "q", c.apply_question_mask, c.extract_cls, p_h, p_a, seed)
        p, Tp = self._encode(1, passage_ids.reshape(bsz * n_passages, plen), passage_mask.reshape(bsz * n_passages, plen), "f",
                             c.apply_passage_mask, c.extract_cls, p_h, p_a, seed)
        dev = q.device
        scale = 1.0 / math.sqrt(q.shape[-1])
        score = torch.empty(bsz, n_passages, dtype=torch.float32, device=dev)
        ops.bi_score(q, p.view(bsz, n_passages, -1), score, scale)
        gold = gold_score.to(dev, torch.float32).contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        ops.kldiv_fwd(score, gold, loss)
        self.ctx = dict(Tq=Tq, Tp=Tp, q=q, p=p, score=score, gold=gold, scale=scale, p_h=p_h, p_a=p_a, seed=seed, n=n_passages)
        anchor = m._by_name["model.embeddings.word_embeddings.weight"]
        return q, p, score, _KLLossFn.apply(anchor, self, loss[0])

    # ------------------------------------------------------------------------------------------------------------------
    def _backward_tower(self, T, dout, p_h, p_a, seed):
        m = self.m
        ops, pk, c, cd = m._get_ops(), m._packed(train=True), m.config, m.compute_dtype
        G = m._grad_views()
        B, L, tower = T.B, T.L, T.tower
        d, H, f = c.hidden_size, c.num_attention_heads, c.intermediate_size
        dk, M = d // H, B * L
        dev = dout.device
        qs = 1.0 / math.sqrt(dk)
        new = lambda *s, dt=cd: torch.empty(*s, dtype=dt, device=dev)       # noqa: E731
        dxh = new(B, L, T.dim)
        ops.seq_mean_bwd(dout, T.pool_mask, dxh)
        dx = dxh.view(M, T.dim)
        if T.head is not None:
            pj, nm = T.head, {"proj": "norm", "proj_iq": "norm_iq", "proj_fact": "norm_fact"}[T.head]
            wp, bp, g, b = pk[T.head]
            dz = new(M, T.dim)
            ops.layernorm_bwd(dx, T.head_t, g, dz, G[nm + ".weight"], G[nm + ".bias"], lin_bias=bp, dbias=G[pj + ".bias"], eps=1e-5)
            ops.gemm_tn(dz, T.x_final, G[pj + ".weight"])
            dx = new(M, d)
            ops.gemm_nt(dz, pk[T.head + "_t"], dx)
        for i in reversed(range(len(pk["layers"]))):
            ly, A = pk["layers"][i], T.layers[i]
            pre = f"model.encoder.layer.{i}."
            # ---- FFN sublayer: x2 = LN(dropout(h·Wo2ᵀ + bo2) + x1)
            dz2 = new(M, d)
            ops.layernorm_bwd(dx, A["u2"], ly["g2"], dz2, G[pre + "output.LayerNorm.weight"], G[pre + "output.LayerNorm.bias"],
                              resid=A["x1"], eps=c.layer_norm_eps)
            dt2 = dz2
            if p_h > 0:
                dt2 = new(M, d)
                ops.dropout_apply(dz2, dt2, (p_h, seed, _site(i, 2, tower)))
            ops.bias_act_bwd(dt2, None, ly["bo2"], None, G[pre + "output.dense.bias"])
            ops.gemm_tn(dt2, A["h"], G[pre + "output.dense.weight"])
            dh = new(M, f)
            ops.gemm_nt(dt2, ly["wo2_t"], dh)
            ops.bias_act_bwd(dh, A["hraw"], ly["bi"], dh, G[pre + "intermediate.dense.bias"], gelu=True)
            ops.gemm_tn(dh, A["x1"], G[pre + "intermediate.dense.weight"])
            dx1 = new(M, d)
            ops.gemm_nt(dh, ly["wi_t"], dx1, resid=dz2)
            # ---- attention sublayer: x1 = LN(dropout(ctx·Woᵀ + bo) + x)
            dz1 = new(M, d)
            ops.layernorm_bwd(dx1, A["u1"], ly["g1"], dz1, G[pre + "attention.output.LayerNorm.weight"],
                              G[pre + "attention.output.LayerNorm.bias"], resid=A["x"], eps=c.layer_norm_eps)
            dt1 = dz1
            if p_h > 0:
                dt1 = new(M, d)
                ops.dropout_apply(dz1, dt1, (p_h, seed, _site(i, 1, tower)))
            ops.bias_act_bwd(dt1, None, ly["bo"], None, G[pre + "attention.output.dense.bias"])
            ops.gemm_tn(dt1, A["ctx"], G[pre + "attention.output.dense.weight"])
            dctx = new(M, d)
            ops.gemm_nt(dt1, ly["wo_t"], dctx)
            q4 = A["qkv"].view(B, L, 3, H, dk)
            dqkv = torch.zeros(M, 3 * d, dtype=cd, device=dev)
            dq4 = dqkv.view(B, L, 3, H, dk)
            ops.attn_bwd(q4[:, :, 0], q4[:, :, 1], q4[:, :, 2], A["ctx"].view(B, L, H, dk), dctx.view(B, L, H, dk), A["stats"],
                         dq4[:, :, 0], dq4[:, :, 1], dq4[:, :, 2], key_mask=T.km,
                         drop=(p_a, seed, _site(i, 0, tower)) if p_a > 0 else None)
            # the packed query weights / bias carry the 1/sqrt(d_head) score scale: the gradients of the parameters take it back
            for j, (nm, sc) in enumerate((("query", qs), ("key", 1.0), ("value", 1.0))):
                blk = dqkv[:, j * d:(j + 1) * d]
                ops.bias_act_bwd(blk, None, ly["bqkv"][j * d:(j + 1) * d], None, G[pre + f"attention.self.{nm}.bias"], dbias_scale=sc)
                ops.gemm_tn(blk, A["x"], G[pre + f"attention.self.{nm}.weight"], alpha=sc)
            dx = new(M, d)
            ops.gemm_nt(dqkv, ly["wqkv_t"], dx, resid=dz1)
        if p_h > 0:
            ops.dropout_apply(dx, dx, (p_h, seed, S_EMB * 2 + tower))
        e = "model.embeddings."
        ops.bert_embed_bwd(T.ids.view(-1), pk["word"], pk["pos"], pk["type0"], pk["emb_g"], dx, G[e + "word_embeddings.weight"],
                           G[e + "position_embeddings.weight"], G[e + "token_type_embeddings.weight"][0], G[e + "LayerNorm.weight"],
                           G[e + "LayerNorm.bias"], L, c.layer_norm_eps)

    def backward(self, upstream=None):
        cx = self.ctx
        if cx is None:
            raise RuntimeError("Retriever: backward() without a preceding training forward")
        m = self.m
        ops = m._get_ops()
        B, n = cx["score"].shape
        dscore = torch.empty_like(cx["score"])
        ops.kldiv_bwd(cx["score"], cx["gold"], dscore, upstream)
        dq, dp = torch.empty_like(cx["q"]), torch.empty_like(cx["p"])
        ops.bi_score_bwd(dscore, cx["q"], cx["p"].view(B, n, -1), dq, dp.view(B, n, -1), cx["scale"])
        self._backward_tower(cx["Tp"], dp, cx["p_h"], cx["p_a"], cx["seed"])
        self._backward_tower(cx["Tq"], dq, cx["p_h"], cx["p_a"], cx["seed"])
        # HF BertEmbeddings builds word_embeddings with padding_idx = pad_token_id (0): torch never gives that row a gradient
        ops.zero_(m._grad_views()["model.embeddings.word_embeddings.weight"][m.config.pad_token_id])
        m._get_engine().step_count += 1
        self.ctx = None
