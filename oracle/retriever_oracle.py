"""TEST INFRASTRUCTURE — CPU restatement (plain torch, float64 by default) of the reference's BERT bi-encoder forward, for
checking `lako_amd/retriever.py`; never imported by the product path.

Pinned: `tests/golden/retriever.npz` holds outputs of the reference's own `src.model.Retriever` (on transformers'
`BertModel`, eval mode) for the weights `init_weights(...)` generates, written by `oracle/make_fixtures.py::make_retriever`;
`tests/test_retriever.py` checks this restatement against them.

  bert_encode      transformers BertModel.forward (modeling_bert.py): BertEmbeddings (word + absolute position + token type 0
                   → LayerNorm), BertLayer × n (self-attention with additive finfo.min key mask, scores / sqrt(d_head);
                   BertSelfOutput / BertOutput: dense + residual → LayerNorm; BertIntermediate: dense → exact GELU)
  embed_text       src/model.py:451-478
  forward          src/model.py:413-449 (einsum('bd,bid->bi') / sqrt(d)); kldivloss :480-483
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def param_shapes(c) -> dict:
    from lako_amd.retriever import _param_shapes       # names/shapes only (host metadata, no compute)
    return _param_shapes(c)


def init_weights(c, seed: int = 0) -> dict[str, torch.Tensor]:
    """Deterministic in `seed`; larger than BERT's 0.02 init so that attention and LayerNorm are not near-trivial."""
    g = torch.Generator().manual_seed(seed)
    w = {}
    for name, shape in param_shapes(c).items():
        if name.endswith("LayerNorm.weight") or (name.startswith("norm") and name.endswith(".weight")):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(".bias"):
            t = 0.1 * torch.randn(shape, generator=g)
        elif "embeddings" in name:
            t = 0.5 * torch.randn(shape, generator=g)
        else:
            t = torch.randn(shape, generator=g) * (2.0 / math.sqrt(shape[1]))
        w[name] = t.float()
    return w


def bert_encode(c, w, ids, mask=None, dtype=torch.float64):
    W = {k: v.to(dtype) for k, v in w.items()}
    B, L = ids.shape
    d, H = c.hidden_size, c.num_attention_heads
    dk = d // H
    # (nn.Embedding(vocab, hidden, padding_idx=pad_token_id = 0) in HF BertEmbeddings: the pad row receives NO gradient)
    x = F.embedding(ids, W["model.embeddings.word_embeddings.weight"], padding_idx=0) + W["model.embeddings.position_embeddings.weight"][:L][None] \
        + W["model.embeddings.token_type_embeddings.weight"][0]
    x = F.layer_norm(x, (d,), W["model.embeddings.LayerNorm.weight"], W["model.embeddings.LayerNorm.bias"], c.layer_norm_eps)
    add = None
    if mask is not None:
        add = (1.0 - mask.to(dtype))[:, None, None, :] * torch.finfo(dtype).min
    for i in range(c.num_hidden_layers):
        p = f"model.encoder.layer.{i}."
        lin = lambda t, n: t @ W[p + n + ".weight"].T + W[p + n + ".bias"]          # noqa: E731
        heads = lambda t: t.view(B, L, H, dk).transpose(1, 2)                       # noqa: E731
        q, k, v = heads(lin(x, "attention.self.query")), heads(lin(x, "attention.self.key")), heads(lin(x, "attention.self.value"))
        s = q @ k.transpose(-1, -2) / math.sqrt(dk)
        if add is not None:
            s = s + add
        ctx = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, L, d)
        x = F.layer_norm(lin(ctx, "attention.output.dense") + x, (d,), W[p + "attention.output.LayerNorm.weight"],
                         W[p + "attention.output.LayerNorm.bias"], c.layer_norm_eps)
        h = F.gelu(lin(x, "intermediate.dense"))
        x = F.layer_norm(lin(h, "output.dense") + x, (d,), W[p + "output.LayerNorm.weight"], W[p + "output.LayerNorm.bias"],
                         c.layer_norm_eps)
    return x


def embed_text(c, w, ids, mask, text_type="q", apply_mask=False, extract_cls=False, dtype=torch.float64):
    x = bert_encode(c, w, ids, mask if apply_mask else None, dtype)
    head = None
    if c.projection:
        head = ("proj", "norm")
    elif c.asymmetric_retri == "yes":
        head = ("proj_iq", "norm_iq") if text_type == "q" else ("proj_fact", "norm_fact")
    if head is not None:
        pj, nm = head
        x = x @ w[pj + ".weight"].to(dtype).T + w[pj + ".bias"].to(dtype)
        x = F.layer_norm(x, (x.shape[-1],), w[nm + ".weight"].to(dtype), w[nm + ".bias"].to(dtype), 1e-5)
    if extract_cls:
        return x[:, 0]
    if apply_mask:
        m = mask.bool()
        return x.masked_fill(~m[:, :, None], 0.0).sum(1) / m.sum(1)[:, None]
    return x.mean(1)


def forward(c, w, question_ids, question_mask, passage_ids, passage_mask, gold_score=None, dtype=torch.float64):
    bsz, n, plen = passage_ids.shape
    q = embed_text(c, w, question_ids, question_mask, "q", c.apply_question_mask, c.extract_cls, dtype)
    p = embed_text(c, w, passage_ids.view(bsz * n, plen), passage_mask.view(bsz * n, plen), "f", c.apply_passage_mask,
                   c.extract_cls, dtype)
    score = torch.einsum("bd,bid->bi", q, p.view(bsz, n, -1)) / math.sqrt(q.shape[-1])
    loss = None
    if gold_score is not None:
        loss = torch.nn.KLDivLoss()(torch.log_softmax(score, -1), gold_score.to(dtype))
    return q, p, score, loss


def synthetic_batch(c, bsz, n, qlen, plen, seed=0):
    """ids in [1, vocab), valid-first bool masks with ragged lengths (pad id 0), gold scores = a softmax."""
    g = torch.Generator().manual_seed(seed)
    def one(rows, L):
        ids = torch.randint(1, c.vocab_size, (rows, L), generator=g)
        lens = torch.randint(max(1, L // 3), L + 1, (rows,), generator=g)
        lens[0] = L
        m = torch.arange(L)[None, :] < lens[:, None]
        return ids * m, m
    qi, qm = one(bsz, qlen)
    pi, pm = one(bsz * n, plen)
    gold = torch.softmax(torch.randn(bsz, n, generator=g) * 2.0, -1)
    return qi, qm, pi.view(bsz, n, plen), pm.view(bsz, n, plen), gold
