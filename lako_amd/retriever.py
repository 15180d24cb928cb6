"""BERT bi-encoder retriever, FORWARD (inference) path — SURVEY.md §8 f4, the drop-in for `src.model.Retriever`
(src/model.py:352-483) where it is used to embed questions / facts and score them (`forward(..., gold_score=None)`,
`embed_text`; passage_retrieval.py / generate_passage_embeddings.py feed `lako_amd.index.Indexer` with its outputs).

What runs: HF `BertModel` (absolute positions, token type 0, post-LN blocks, biased linears, exact GELU, eval mode — no
dropout) → optional projection + LayerNorm (`proj`/`norm`, or `proj_iq`/`proj_fact` for the asymmetric variant) → CLS
row or (masked) mean → `einsum('bd,bid->bi') / sqrt(d)`.  Every FLOP is a gfx950 kernel behind the C-ABI: the reader's
`lako_gemm_nt` and `lako_attn_fwd` (the 1/sqrt(d_head) scale is folded into the query weights here) plus the row-wise
kernels of `csrc/bertops.hip`.  Training the retriever (KL distillation backward, train_retriever.py) is NOT built:
`forward(gold_score=…)` returns the KLDivLoss *value* only, and `.backward()` does not exist on it.

Parameters carry the reference's state-dict key names (`model.embeddings.…`, `model.encoder.layer.i.…`, `proj.weight`, …),
so a checkpoint written by the reference's `Retriever.save_pretrained` loads with `load_state_dict`.
"""
from __future__ import annotations

import json
import math
import os
from dataclasses import asdict, dataclass

import torch
from torch import nn

from ._lib import LakoError


@dataclass
class RetrieverConfig:
    """The fields of src/model.py:352-372 plus the BertConfig fields the forward needs."""
    vocab_size: int = 30522
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    max_position_embeddings: int = 512
    type_vocab_size: int = 2
    layer_norm_eps: float = 1e-12
    hidden_act: str = "gelu"
    indexing_dimension: int = 256
    apply_question_mask: bool = False
    apply_passage_mask: bool = False
    extract_cls: bool = False
    passage_maxlength: int = 130
    question_maxlength: int = 130
    projection: bool = True
    asymmetric_retri: object = False       # the reference compares with the string "yes" (src/model.py:395,462)

    @classmethod
    def from_hf(cls, cfg):
        get = (lambda k, d: cfg.get(k, d)) if isinstance(cfg, dict) else (lambda k, d: getattr(cfg, k, d))
        kw = {f: get(f, getattr(cls, f)) for f in cls.__dataclass_fields__}
        if get("position_embedding_type", "absolute") not in ("absolute", None):
            raise LakoError("Retriever: only absolute position embeddings are built")
        return cls(**kw)

    def to_json(self):
        return json.dumps(asdict(self), indent=2)


def _param_shapes(c: RetrieverConfig):
    d, f = c.hidden_size, c.intermediate_size
    s = {"model.embeddings.word_embeddings.weight": (c.vocab_size, d),
         "model.embeddings.position_embeddings.weight": (c.max_position_embeddings, d),
         "model.embeddings.token_type_embeddings.weight": (c.type_vocab_size, d),
         "model.embeddings.LayerNorm.weight": (d,), "model.embeddings.LayerNorm.bias": (d,)}
    for i in range(c.num_hidden_layers):
        p = f"model.encoder.layer.{i}."
        for n in ("query", "key", "value"):
            s[p + f"attention.self.{n}.weight"] = (d, d)
            s[p + f"attention.self.{n}.bias"] = (d,)
        s[p + "attention.output.dense.weight"] = (d, d)
        s[p + "attention.output.dense.bias"] = (d,)
        s[p + "attention.output.LayerNorm.weight"] = (d,)
        s[p + "attention.output.LayerNorm.bias"] = (d,)
        s[p + "intermediate.dense.weight"] = (f, d)
        s[p + "intermediate.dense.bias"] = (f,)
        s[p + "output.dense.weight"] = (d, f)
        s[p + "output.dense.bias"] = (d,)
        s[p + "output.LayerNorm.weight"] = (d,)
        s[p + "output.LayerNorm.bias"] = (d,)
    s["model.pooler.dense.weight"] = (d, d)          # present in the checkpoints, unused by embed_text (output [0])
    s["model.pooler.dense.bias"] = (d,)
    heads = []
    if c.projection:
        heads = [("proj", "norm")]
    elif c.asymmetric_retri == "yes":
        heads = [("proj_iq", "norm_iq"), ("proj_fact", "norm_fact")]
    for pj, nm in heads:
        s[pj + ".weight"] = (c.indexing_dimension, d)
        s[pj + ".bias"] = (c.indexing_dimension,)
        s[nm + ".weight"] = (c.indexing_dimension,)
        s[nm + ".bias"] = (c.indexing_dimension,)
    return s


IGNORED_KEYS = ("model.embeddings.position_ids", "model.embeddings.token_type_ids")    # HF buffers in older checkpoints


class Retriever(nn.Module):
    """`Retriever(config)`; `load_state_dict(reference_state_dict)`; `.cuda()`; `forward(...)` / `embed_text(...)`."""

    def __init__(self, config, dtype: torch.dtype | None = None, seed: int = 0, _ops=None):
        super().__init__()
        self.config = config if isinstance(config, RetrieverConfig) else RetrieverConfig.from_hf(config)
        c = self.config
        if c.hidden_act not in ("gelu",):
            raise LakoError(f"Retriever: hidden_act {c.hidden_act!r} is not built (exact GELU only)")
        if c.hidden_size % c.num_attention_heads or c.hidden_size // c.num_attention_heads not in (32, 64):
            raise LakoError("Retriever: head size must be 32 or 64 (the attention kernel's tiles)")
        env = os.environ.get("LAKO_DTYPE", "bf16").lower()
        self.compute_dtype = dtype or (torch.float32 if env in ("f32", "fp32", "float32") else torch.bfloat16)
        self._ops = _ops
        self._pack = None
        self._pack_key = None
        g = torch.Generator().manual_seed(seed)
        self._by_name = {}
        for name, shape in _param_shapes(c).items():
            if name.endswith("LayerNorm.weight") or name.startswith("norm") and name.endswith(".weight"):
                t = torch.ones(shape)
            elif name.endswith(".bias"):
                t = torch.zeros(shape)
            else:
                t = torch.randn(shape, generator=g) * 0.02      # BertPreTrainedModel._init_weights std
            mod = self
            parts = name.split(".")
            for part in parts[:-1]:
                if part not in mod._modules:
                    mod.add_module(part, nn.Module())
                mod = mod._modules[part]
            prm = nn.Parameter(t, requires_grad=False)
            mod.register_parameter(parts[-1], prm)
            self._by_name[name] = prm
        self.eval()

    # ------------------------------------------------------------------------------------------
    def load_state_dict(self, state_dict, strict=True, **kw):
        sd = {k: v for k, v in state_dict.items() if k not in IGNORED_KEYS}
        missing = set(self._by_name) - set(sd) - {"model.pooler.dense.weight", "model.pooler.dense.bias"}
        unexpected = set(sd) - set(self._by_name)
        if strict and (missing or unexpected):
            raise KeyError(f"Retriever.load_state_dict: missing {sorted(missing)[:4]} unexpected {sorted(unexpected)[:4]}")
        with torch.no_grad():
            for k, v in sd.items():
                if k in self._by_name:
                    p = self._by_name[k]
                    if tuple(v.shape) != tuple(p.shape):
                        raise ValueError(f"{k}: shape {tuple(v.shape)} != {tuple(p.shape)}")
                    p.copy_(torch.as_tensor(v).to(p.device, torch.float32))
        self._pack = None
        return missing

    def save_pretrained(self, path):
        from safetensors.torch import save_file
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            f.write(self.config.to_json())
        save_file({k: v.detach().cpu().contiguous().clone() for k, v in self.state_dict().items()},
                  os.path.join(path, "model.safetensors"), metadata={"format": "pt"})

    @classmethod
    def from_pretrained(cls, path, **kw):
        with open(os.path.join(path, "config.json")) as f:
            cfg = RetrieverConfig.from_hf(json.load(f))
        model = cls(cfg, **kw)
        st = os.path.join(path, "model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu")
        model.load_state_dict(sd)
        return model

    # ------------------------------------------------------------------------------------------
    def _get_ops(self):
        if self._ops is None:
            from .ops import HipOps
            self._ops = HipOps()            # raises without a GPU / the HIP library: there is no CPU fallback
        return self._ops

    def _packed(self):
        """Compute-layout weights: fused [3d, d] QKV with the 1/sqrt(d_head) score scale folded into the query rows,
        matrices in the compute dtype, biases / LayerNorm / embedding tables in fp32."""
        dev = self._by_name["model.embeddings.word_embeddings.weight"].device
        key = (str(dev), self.compute_dtype, tuple(p._version for p in self._by_name.values()))
        if self._pack is not None and self._pack_key == key:
            return self._pack
        c, P, cd = self.config, self._by_name, self.compute_dtype
        qs = 1.0 / math.sqrt(c.hidden_size // c.num_attention_heads)
        f32 = lambda n: P[n].detach().float().contiguous()        # noqa: E731
        w = lambda n: P[n].detach().to(cd).contiguous()           # noqa: E731
        pk = {"word": f32("model.embeddings.word_embeddings.weight"), "pos": f32("model.embeddings.position_embeddings.weight"),
              "type0": f32("model.embeddings.token_type_embeddings.weight")[0].contiguous(),
              "emb_g": f32("model.embeddings.LayerNorm.weight"), "emb_b": f32("model.embeddings.LayerNorm.bias"), "layers": []}
        for i in range(c.num_hidden_layers):
            p = f"model.encoder.layer.{i}."
            a = p + "attention.self."
            pk["layers"].append({
                "wqkv": torch.cat([P[a + "query.weight"].detach() * qs, P[a + "key.weight"].detach(), P[a + "value.weight"].detach()]).to(cd).contiguous(),
                "bqkv": torch.cat([P[a + "query.bias"].detach() * qs, P[a + "key.bias"].detach(), P[a + "value.bias"].detach()]).float().contiguous(),
                "wo": w(p + "attention.output.dense.weight"), "bo": f32(p + "attention.output.dense.bias"),
                "g1": f32(p + "attention.output.LayerNorm.weight"), "b1": f32(p + "attention.output.LayerNorm.bias"),
                "wi": w(p + "intermediate.dense.weight"), "bi": f32(p + "intermediate.dense.bias"),
                "wo2": w(p + "output.dense.weight"), "bo2": f32(p + "output.dense.bias"),
                "g2": f32(p + "output.LayerNorm.weight"), "b2": f32(p + "output.LayerNorm.bias")})
        for pj, nm in (("proj", "norm"), ("proj_iq", "norm_iq"), ("proj_fact", "norm_fact")):
            if pj + ".weight" in P:
                pk[pj] = (w(pj + ".weight"), f32(pj + ".bias"), f32(nm + ".weight"), f32(nm + ".bias"))
        self._pack, self._pack_key = pk, key
        return pk

    # ------------------------------------------------------------------------------------------
    def encode(self, text_ids, text_mask=None):
        """BertModel(input_ids, attention_mask)[0] as a [B·L, hidden] tensor in the compute dtype."""
        ops, pk, c, cd = self._get_ops(), self._packed(), self.config, self.compute_dtype
        B, L = text_ids.shape
        dev = pk["word"].device
        d, H, f = c.hidden_size, c.num_attention_heads, c.intermediate_size
        dk = d // H
        M = B * L
        ids = text_ids.to(dev, torch.int64).contiguous()
        km = None
        if text_mask is not None:
            km = text_mask.to(dev).ne(0).to(torch.uint8).contiguous()
        new = lambda *s, dt=cd: torch.empty(*s, dtype=dt, device=dev)     # noqa: E731
        x, x1, qkv, ctx, t = new(M, d), new(M, d), new(M, 3 * d), new(M, d), new(M, d)
        h = new(M, f)
        stats = new(B, H, L, 4, dt=torch.float32)     # softmax row statistics (written, unused: forward only)
        ops.bert_embed(ids.view(-1), pk["word"], pk["pos"], pk["type0"], pk["emb_g"], pk["emb_b"], x, L, c.layer_norm_eps)
        q4 = qkv.view(B, L, 3, H, dk)
        for ly in pk["layers"]:
            ops.gemm_nt(x, ly["wqkv"], qkv)
            ops.bias_act(qkv, ly["bqkv"], qkv)
            ops.attn_fwd(q4[:, :, 0], q4[:, :, 1], q4[:, :, 2], ctx.view(B, L, H, dk), stats, key_mask=km)
            ops.gemm_nt(ctx, ly["wo"], t)
            ops.layernorm_fwd(t, ly["g1"], ly["b1"], x1, lin_bias=ly["bo"], resid=x, eps=c.layer_norm_eps)
            ops.gemm_nt(x1, ly["wi"], h)
            ops.bias_act(h, ly["bi"], h, gelu=True)
            ops.gemm_nt(h, ly["wo2"], t)
            ops.layernorm_fwd(t, ly["g2"], ly["b2"], x, lin_bias=ly["bo2"], resid=x1, eps=c.layer_norm_eps)
        return x

    def embed_text(self, text_ids, text_mask, text_type="q", apply_mask=False, extract_cls=False):
        """src/model.py:451-478 → fp32 [B, dim]."""
        ops, pk, c = self._get_ops(), self._packed(), self.config
        B, L = text_ids.shape
        x = self.encode(text_ids, text_mask if apply_mask else None)
        head = None
        if c.projection:
            head = pk["proj"]
        elif c.asymmetric_retri == "yes":
            head = pk["proj_iq"] if text_type == "q" else pk["proj_fact"]
        if head is not None:
            wp, bp, g, b = head
            t = torch.empty(B * L, wp.shape[0], dtype=x.dtype, device=x.device)
            ops.gemm_nt(x, wp, t)
            x = torch.empty_like(t)
            ops.layernorm_fwd(t, g, b, x, lin_bias=bp, eps=1e-5)            # nn.LayerNorm default eps
        dim = x.shape[1]
        out = torch.empty(B, dim, dtype=torch.float32, device=x.device)
        if extract_cls:
            sel = torch.zeros(B, L, dtype=torch.uint8, device=x.device)
            sel[:, 0] = 1
            ops.seq_mean(x.view(B, L, dim), sel, out)                        # the mean of one row is that row
        elif apply_mask:
            ops.seq_mean(x.view(B, L, dim), text_mask.to(x.device).ne(0).to(torch.uint8).contiguous(), out)
        else:
            ops.seq_mean(x.view(B, L, dim), None, out)
        return out

    def forward(self, question_ids, question_mask, passage_ids, passage_mask, gold_score=None):
        """src/model.py:413-449 → (question_output, passage_output, score, loss)."""
        c = self.config
        bsz, n_passages, plen = passage_ids.shape
        q = self.embed_text(question_ids, question_mask, "q", c.apply_question_mask, c.extract_cls)
        p = self.embed_text(passage_ids.reshape(bsz * n_passages, plen), passage_mask.reshape(bsz * n_passages, plen), "f",
                            c.apply_passage_mask, c.extract_cls)
        score = torch.empty(bsz, n_passages, dtype=torch.float32, device=q.device)
        self._get_ops().bi_score(q, p.view(bsz, n_passages, -1), score, 1.0 / math.sqrt(q.shape[-1]))
        loss = None
        if gold_score is not None:
            loss = torch.empty(1, dtype=torch.float32, device=q.device)
            self._get_ops().kldiv_fwd(score, gold_score.to(q.device, torch.float32).contiguous(), loss)
            loss = loss[0]
        return q, p, score, loss

    def kldivloss(self, score, gold_score):
        """src/model.py:480-483 (value only)."""
        loss = torch.empty(1, dtype=torch.float32, device=score.device)
        self._get_ops().kldiv_fwd(score.contiguous(), gold_score.to(score.device, torch.float32).contiguous(), loss)
        return loss[0]
