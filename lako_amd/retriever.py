"""BERT bi-encoder retriever, FORWARD (inference) path — SURVEY.md §8 f4, the drop-in for `src.model.Retriever`
(src/model.py:352-483) where it is used to embed questions / facts and score them (`forward(..., gold_score=None)`,
`embed_text`; passage_retrieval.py / generate_passage_embeddings.py feed `lako_amd.index.Indexer` with its outputs).

What runs: HF `BertModel` (absolute positions, token type 0, post-LN blocks, biased linears, exact GELU, eval mode — no
dropout) → optional projection + LayerNorm (`proj`/`norm`, or `proj_iq`/`proj_fact` for the asymmetric variant) → CLS
row or (masked) mean → `einsum('bd,bid->bi') / sqrt(d)`.  Every FLOP is a gfx950 kernel behind the C-ABI: the reader's
`lako_gemm_nt` and `lako_attn_fwd` (the 1/sqrt(d_head) scale is folded into the query weights here) plus the row-wise
kernels of `csrc/bertops.hip`.  TRAINING (round 3): with autograd enabled, `forward(..., gold_score=…)` runs the training
schedule of `lako_amd/retriever_train.py` (activations kept, dropout in `.train()` mode) and returns a loss whose `.backward()`
runs the hand-written backward (csrc/bertbwd.hip + the reader's GEMM / attention-backward kernels) into `.grad`; the parameters
are views of one flat fp32 buffer, so `util.set_optim` / `util.clip_grad_norm_` (the reader's fused AdamW) apply unchanged —
train_retriever.py:37-71.  Under `torch.no_grad()` the forward is the inference schedule below.

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
    pad_token_id: int = 0                  # BertConfig default: the word-embedding row that never receives a gradient
    hidden_dropout_prob: float = 0.1       # BertConfig defaults; train_retriever.py:237 overrides both through
    attention_probs_dropout_prob: float = 0.1   # src.util.set_dropout(model, opt.dropout) → Retriever.set_dropout

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
        self._seed = int(seed)
        self._engine = None
        self._trainer = None
        self._master_version = 0
        g = torch.Generator().manual_seed(seed)
        # one flat fp32 master buffer (64-element aligned blocks); every nn.Parameter is a view of it — the fused optimizer and the
        # gradient-norm kernel walk P / G as flat arrays (util.AdamW, util.clip_grad_norm_), like the reader's engine
        shapes = _param_shapes(c)
        # (the pooler — in the checkpoints, never on the path: embed_text takes BertModel's output [0] — sits behind the trained range:
        #  the reference's optimizer skips parameters without a gradient, decay included, so the fused step must not see it)
        order = [k for k in shapes if not k.startswith("model.pooler.")] + [k for k in shapes if k.startswith("model.pooler.")]
        self._layout, off, self._n_train = {}, 0, None
        for name in order:
            shape = shapes[name]
            if name.startswith("model.pooler.") and self._n_train is None:
                self._n_train = off
            n = 1
            for k in shape:
                n *= k
            self._layout[name] = (off, n, tuple(shape))
            off += -(-n // 64) * 64
        self._n_train = off if self._n_train is None else self._n_train
        self._master = torch.zeros(off, dtype=torch.float32)
        self._by_name = {}
        for name in shapes:                       # (draw in the state-dict order: the initialisation does not depend on the layout)
            o, n, shape = self._layout[name]
            v = self._master[o:o + n].view(shape)
            if name.endswith("LayerNorm.weight") or name.startswith("norm") and name.endswith(".weight"):
                v.fill_(1.0)
            elif not name.endswith(".bias"):
                v.copy_(torch.randn(shape, generator=g) * 0.02)      # BertPreTrainedModel._init_weights std
            mod = self
            parts = name.split(".")
            for part in parts[:-1]:
                if part not in mod._modules:
                    mod.add_module(part, nn.Module())
                mod = mod._modules[part]
            prm = nn.Parameter(v, requires_grad=True)
            mod.register_parameter(parts[-1], prm)
            self._by_name[name] = prm
        self.eval()

    # ------------------------------------------------------------------------------------------
    # flat storage: moving the model moves ONE buffer and re-points the parameter views (nn.Module._apply would give every
    # parameter a storage of its own)
    def _views(self, flat):
        return {name: flat[o:o + n].view(shape) for name, (o, n, shape) in self._layout.items()}

    def _rebind(self):
        views = self._views(self._master)
        gviews = self._views(self._engine.G_all) if self._engine is not None else None
        for name, prm in self._by_name.items():
            prm.data = views[name]
            prm.grad = gviews[name] if gviews is not None else None

    def _apply(self, fn, recurse=True):
        new = fn(self._master)
        if self._engine is not None and new.device == self._master.device:
            if new.data_ptr() != self._master.data_ptr():
                self._master.copy_(new)
                self._pack = None
            return self
        self._master = new if new.dtype == torch.float32 else new.float()
        self._engine = None
        self._pack = None
        self._rebind()
        return self

    def _get_engine(self):
        """flat P / G / AdamW moments for util.set_optim / util.clip_grad_norm_ (retriever_train.RetrieverEngine)"""
        if self._engine is None:
            from .retriever_train import RetrieverEngine
            self._engine = RetrieverEngine(self, self._get_ops(), self._master.device)
            self._rebind()
        return self._engine

    def _grad_views(self):
        eng = self._get_engine()
        if getattr(self, "_gv", None) is None or self._gv[0] is not eng.G_all:
            self._gv = (eng.G_all, self._views(eng.G_all))
        return self._gv[1]

    def zero_grad(self, set_to_none: bool = False):
        if self._engine is not None:
            self._engine.zero_grad()

    def set_dropout(self, p: float):
        """src/util.py set_dropout(model, p) (train_retriever.py:237): every nn.Dropout of the BERT tower gets probability p"""
        self.config.hidden_dropout_prob = float(p)
        self.config.attention_probs_dropout_prob = float(p)

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
        self._master_version += 1
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
            sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu", weights_only=True)      # a state dict: tensors only
        model.load_state_dict(sd)
        return model

    # ------------------------------------------------------------------------------------------
    def _get_ops(self):
        if self._ops is None:
            from .ops import HipOps
            self._ops = HipOps()            # raises without a GPU / the HIP library: there is no CPU fallback
        return self._ops

    def _packed(self, train: bool = False):
        """Compute-layout weights: fused [3d, d] QKV with the 1/sqrt(d_head) score scale folded into the query rows,
        matrices in the compute dtype, biases / LayerNorm / embedding tables in fp32.  train: also the transposed copies the
        backward's dX products read (dX = dY·W as an NT product against Wᵀ)."""
        dev = self._by_name["model.embeddings.word_embeddings.weight"].device
        key = (str(dev), self.compute_dtype, self._master_version, tuple(p._version for p in self._by_name.values()))
        if self._pack is not None and self._pack_key == key and (not train or self._pack.get("train")):
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
                if train:
                    pk[pj + "_t"] = pk[pj][0].t().contiguous()
        if train:
            for ly in pk["layers"]:
                for n in ("wqkv", "wo", "wi", "wo2"):
                    ly[n + "_t"] = ly[n].t().contiguous()
            pk["train"] = True
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
        """src/model.py:413-449 → (question_output, passage_output, score, loss).  With a gold score and autograd enabled the
        loss carries the hand-written backward (train_retriever.py:57-66); otherwise this is the inference schedule."""
        c = self.config
        if gold_score is not None and torch.is_grad_enabled():
            if self._trainer is None:
                from .retriever_train import RetrieverTrainer
                self._trainer = RetrieverTrainer(self)
            return self._trainer.forward(question_ids, question_mask, passage_ids, passage_mask, gold_score)
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
