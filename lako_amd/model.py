"""`FiDT5` — the drop-in boundary: same surface as the reference's `src.model.FiDT5`
(src/model.py:20-213) as used by train_reader.py / test_reader.py, with every FLOP executed by the
hand-written gfx950 kernels of liblako_hip.so through `lako_amd.engine.Engine`.

Surface kept:  FiDT5(config) · load_t5(state_dict) · from_pretrained / save_pretrained
(wrapped key names `encoder.encoder.block.i.module.…`) · cuda/train/eval/parameters/zero_grad ·
set_checkpoint · model(input_ids=[B,N,L], attention_mask=[B,N,L], labels=[B,T])[0].backward() ·
generate(input_ids, attention_mask, max_length) · overwrite_forward_crossattention /
reset_score_storage / get_crossattention_scores.

There is no CPU execution path: forward/generate raise unless the model lives on a ROCm device (tests
inject a CPU test double through the private `_ops` argument — never the product).
"""
from __future__ import annotations

import json
import os
import weakref

import torch
from torch import nn

from .config import FiDConfig
from .engine import Engine, build_layout, layout_sizes

# the relative-position table transformers 3.0.2 gives the decoder's first cross-attention layer (dropped on load by every later
# transformers: HF5:899-901).  Ignored unless the config says `legacy_cross_bias` — then it is a parameter (config.py, engine.py)
LEGACY_IGNORED = ("decoder.block.0.layer.1.EncDecAttention.relative_attention_bias.weight",)
ALIASES = ("lm_head.weight", "encoder.embed_tokens.weight", "decoder.embed_tokens.weight")


def wrapped_name(plain: str) -> str:
    """plain T5 key → key of the wrapped FiD model (src/model.py:62-66,216-225,274-283)."""
    if plain.startswith("encoder."):
        rest = plain[len("encoder."):]
        if rest.startswith("block."):
            parts = rest.split(".")
            rest = ".".join(parts[:2] + ["module"] + parts[2:])
        return "encoder.encoder." + rest
    return plain


def plain_name(name: str) -> str:
    if name.startswith("encoder.encoder."):
        name = "encoder." + name[len("encoder.encoder."):]
    return name.replace(".module.layer.", ".layer.")


class FiDOutput(tuple):
    """(loss, logits) — indexable like HF's Seq2SeqLMOutput (`model(...)[0]` is the loss)."""

    def __new__(cls, loss, logits):
        return super().__new__(cls, (loss, logits))

    loss = property(lambda self: self[0])
    logits = property(lambda self: self[1])


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model, ids, mask, labels, lengths):
        loss, logits = model._engine.forward_loss(ids, mask, labels, training=model.training, lengths=lengths)
        ctx.model = model
        ctx.mark_non_differentiable(logits)
        return loss.clone(), logits

    @staticmethod
    def backward(ctx, gloss, _glogits):
        ctx.model._engine.backward(upstream=gloss.detach().float().reshape(1).contiguous())
        return None, None, None, None, None, None


class FiDT5(nn.Module):
    def __init__(self, config, dtype: torch.dtype | None = None, seed: int = 0, _ops=None, fp8: bool | None = None):
        super().__init__()
        self.config = config if isinstance(config, FiDConfig) else FiDConfig.from_hf(config)
        env = os.environ.get("LAKO_DTYPE", "bf16").lower()
        self.compute_dtype = dtype or {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "f32": torch.float32,
                                       "fp32": torch.float32, "float32": torch.float32}[env]
        self._ops, self._seed, self._fp8 = _ops, seed, fp8
        self._blocks = build_layout(self.config)
        n, _ = layout_sizes(self._blocks)
        self._master = torch.zeros(n, dtype=torch.float32)
        self._engine: Engine | None = None
        self._use_checkpoint = False
        self._capture_scores = False
        self._score_storage = None
        self.n_passages = None
        self._wrapped = True
        self._build_tree()
        self.reset_parameters()

    # ------------------------------------------------------------------------------------------
    # parameter tree (names as the reference's state_dict), all views into one flat fp32 buffer
    # ------------------------------------------------------------------------------------------
    def _plain_views(self, flat):
        out = {}
        for b in self._blocks:
            v = flat[b.off:b.off + b.numel].view(b.shape)
            for name, r0, nr in b.members:
                out[name] = v[r0:r0 + nr] if len(b.shape) == 2 else v
        return out

    def _build_tree(self):
        for k in list(self._modules):
            del self._modules[k]
        self._params_by_plain = {}
        for plain, view in self._plain_views(self._master).items():
            name = wrapped_name(plain) if self._wrapped else plain
            parts = name.split(".")
            mod = self
            for part in parts[:-1]:
                if part not in mod._modules:
                    mod.add_module(part, nn.Module())
                mod = mod._modules[part]
            prm = nn.Parameter(view)
            prm._lako_model = weakref.ref(self)
            mod.register_parameter(parts[-1], prm)
            self._params_by_plain[plain] = prm

    def _rebind(self):
        views = self._plain_views(self._master)
        gviews = self._plain_views(self._engine.G) if self._engine is not None else None
        for plain, prm in self._params_by_plain.items():
            prm.data = views[plain]
            prm.grad = gviews[plain] if gviews is not None else None

    def _apply(self, fn, recurse=True):
        new = fn(self._master)
        if self._engine is not None and new.device == self._master.device:
            # .cuda() / .to(same device) / .float() on a model that already has its engine: nothing moves.  The engine
            # holds the AdamW moments of a resumed run and the data-parallel gradient hook — rebuilding it would drop both.
            if new.data_ptr() != self._master.data_ptr():
                self._master.copy_(new)          # fn produced a copy (e.g. a dtype round trip): keep P as the storage
                self._mark_stale()
            return self
        self._master = new if new.dtype == torch.float32 else new.float()
        self._engine = None
        self._rebind()
        return self

    def reset_parameters(self):
        """HF T5 `_init_weights` (HF5:563-616) with factor 1.0."""
        cfg = self.config
        d, dk, H, f = cfg.d_model, cfg.d_kv, cfg.num_heads, cfg.d_ff
        with torch.no_grad():
            for plain, prm in self._params_by_plain.items():
                if plain.endswith("layer_norm.weight"):
                    prm.fill_(1.0)
                elif plain == "shared.weight":
                    prm.normal_(0.0, 1.0)
                elif plain.endswith(".q.weight"):
                    prm.normal_(0.0, (d * dk) ** -0.5)
                elif plain.endswith(".k.weight") or plain.endswith(".v.weight") or plain.endswith("wi.weight"):
                    prm.normal_(0.0, d ** -0.5)
                elif plain.endswith(".o.weight"):
                    prm.normal_(0.0, (H * dk) ** -0.5)
                elif plain.endswith("relative_attention_bias.weight"):
                    prm.normal_(0.0, d ** -0.5)
                elif plain.endswith("wo.weight"):
                    prm.normal_(0.0, f ** -0.5)
        self._mark_stale()

    def _mark_stale(self):
        if self._engine is not None:
            self._engine.shadows_stale = True

    # ------------------------------------------------------------------------------------------
    # engine
    # ------------------------------------------------------------------------------------------
    def _get_engine(self) -> Engine:
        if self._engine is None:
            dev = self._master.device
            ops = self._ops
            if ops is None:
                if dev.type != "cuda":
                    raise RuntimeError("FiDT5 runs only on a ROCm device (call .cuda() first); there is no CPU path")
                from .ops import HipOps
                ops = HipOps()
            eng = Engine.__new__(Engine)
            Engine.__init__(eng, self.config, ops, dev, self.compute_dtype, seed=self._seed, fp8=self._fp8)
            eng.P.copy_(self._master)
            self._master = eng.P
            eng.use_checkpoint = self._use_checkpoint
            self._engine = eng
            self._rebind()
        return self._engine

    # ------------------------------------------------------------------------------------------
    # reference API
    # ------------------------------------------------------------------------------------------
    def forward(self, input_ids=None, attention_mask=None, labels=None, passage_lengths=None, **kwargs):
        """src/model.py:39-51: accepts [B,N,L] or already-flattened [B,N·L] (n_passages remembered).
        `passage_lengths` (extension, optional): HOST int tensor [B, N] of valid tokens per passage, for masks of the
        collator's form (valid tokens first) — lets the unpadded encoder skip the device→host read-back of the mask.  The
        lengths must describe `attention_mask` exactly (lengths[b, n] valid tokens, then padding) — anything else would pack the wrong
        tokens.  Every batch is verified on the device without a host sync; a mismatch raises ValueError a step or two later
        (engine.Engine._check_lengths; `model._engine.check_lengths_now()` waits for the outstanding verdicts; LAKO_CHECK_LENGTHS=0
        switches the check off)."""
        if input_ids is None or labels is None:
            raise ValueError("FiDT5.forward needs input_ids and labels (train_reader.py:67-71)")
        if input_ids.dim() == 3:
            self.n_passages = input_ids.size(1)
        elif self.n_passages is None:
            raise ValueError("2-D input_ids before any 3-D call: n_passages unknown")
        B = input_ids.size(0)
        ids = input_ids.reshape(B, self.n_passages, -1)
        if attention_mask is None:
            attention_mask = torch.ones_like(ids, dtype=torch.bool)
        mask = attention_mask.reshape(B, self.n_passages, -1)
        self._get_engine()
        loss, logits = _LossFn.apply(self._params_by_plain["shared.weight"], self, ids, mask, labels, passage_lengths)
        return FiDOutput(loss, logits)

    @torch.no_grad()
    def generate(self, input_ids, attention_mask, max_length, passage_lengths=None):
        """src/model.py:54-60 → greedy decode, int64 [B, ≤max_length] with the leading start token."""
        self.n_passages = input_ids.size(1)
        eng = self._get_engine()
        if self._capture_scores and self._score_storage is None:
            out, scores = eng.generate(input_ids, attention_mask, max_length, capture_scores=True)
            self._score_storage = scores
            return out
        return eng.generate(input_ids, attention_mask, max_length, lengths=passage_lengths)

    def set_checkpoint(self, use_checkpoint):
        """src/model.py:84-90: recompute each encoder block in backward from its saved input instead of keeping its
        intermediates (the reference's CheckpointWrapper + torch.utils.checkpoint, :237-283).  With 288 GB of HBM per
        GPU this is never required — it trades one extra encoder forward for ~13 GB at config 2."""
        self._use_checkpoint = bool(use_checkpoint)
        if self._engine is not None:
            self._engine.use_checkpoint = self._use_checkpoint

    def wrap_encoder(self, use_checkpoint=False):
        self._wrapped = True
        self._build_tree()
        self._rebind()

    def unwrap_encoder(self):
        self._wrapped = False
        self._build_tree()
        self._rebind()

    def load_t5(self, state_dict):
        """src/model.py:79-82: plain T5 key names (what T5ForConditionalGeneration.state_dict() has)."""
        self.load_state_dict(state_dict)

    def state_dict(self, *args, **kwargs):
        sd = super().state_dict(*args, **kwargs)
        shared = sd["shared.weight"]
        sd["lm_head.weight"] = shared
        sd[("encoder.encoder." if self._wrapped else "encoder.") + "embed_tokens.weight"] = shared
        sd["decoder.embed_tokens.weight"] = shared
        return sd

    def load_state_dict(self, state_dict, strict=True, **kw):
        seen = set()
        with torch.no_grad():
            for k, v in state_dict.items():
                pk = plain_name(k)
                if pk in LEGACY_IGNORED and pk not in self._params_by_plain:
                    # a checkpoint of the reference's transformers 3.0.2 carries a TRAINED relative-attention table of the decoder's
                    # first cross-attention; this model (HF >= 4 semantics) has none: without --legacy_cross_bias the table is dropped and
                    # losses / tokens / cross-attention scores differ from the reference's — say so instead of loading silently
                    if bool(torch.as_tensor(v).ne(0).any()):
                        import warnings
                        warnings.warn(f"{k}: the checkpoint carries a non-zero cross-attention relative-position table (transformers 3.0.2 "
                                      "semantics) that this model ignores; load it with legacy_cross_bias=True (from_pretrained / "
                                      "--legacy_cross_bias) to reproduce the reference's numbers", stacklevel=2)
                    continue
                if pk in ALIASES:
                    continue
                if pk not in self._params_by_plain:
                    if strict:
                        raise KeyError(f"unexpected key {k}")
                    continue
                prm = self._params_by_plain[pk]
                if tuple(v.shape) != tuple(prm.shape):
                    raise ValueError(f"{k}: shape {tuple(v.shape)} != {tuple(prm.shape)}")
                prm.copy_(torch.as_tensor(v).to(prm.device, torch.float32))
                seen.add(pk)
        missing = set(self._params_by_plain) - seen
        if strict and missing:
            raise KeyError(f"missing keys: {sorted(missing)[:5]}…")
        self._mark_stale()
        return missing

    def save_pretrained(self, path):
        """Directory layout of HF save_pretrained (src/util.py:110): config.json + weights under the
        wrapped key names the reference's checkpoints carry."""
        from safetensors.torch import save_file
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            f.write(self.config.to_json())
        sd = {k: v.detach().cpu().contiguous().clone() for k, v in self.state_dict().items()}
        save_file(sd, os.path.join(path, "model.safetensors"), metadata={"format": "pt"})

    @classmethod
    def from_pretrained(cls, path, legacy_cross_bias=None, **kw):
        """`legacy_cross_bias=True`: read the checkpoint as transformers 3.0.2 wrote and ran it — the reference's own checkpoints
        (README.md:21) carry the first cross-attention layer's relative-position table, and their config.json knows nothing of the
        flag (config.py).  None: what config.json says (False when it says nothing: the table is ignored)."""
        with open(os.path.join(path, "config.json")) as f:
            cfg = FiDConfig.from_hf(json.load(f))
        if legacy_cross_bias is not None:
            cfg.legacy_cross_bias = bool(legacy_cross_bias)
        model = cls(cfg, **kw)
        st = os.path.join(path, "model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu", weights_only=True)      # a state dict: tensors only
        model.load_state_dict(sd)
        return model

    def zero_grad(self, set_to_none: bool = False):
        if self._engine is not None:
            self._engine.zero_grad()

    # ---- cross-attention score capture (src/model.py:92-98,143-213) --------------
    def overwrite_forward_crossattention(self):
        """The reference rebinds every decoder block's EncDecAttention.forward to a score-storing copy; here the
        cross-attention kernel itself emits the step-0 pre-softmax scores when capture is on."""
        self._capture_scores = True

    def reset_score_storage(self):
        self._score_storage = None

    def get_crossattention_scores(self, opt, context_ids, tokenizer, context_mask):
        """src/model.py:143-204 (stream == 2): per-fact aggregation of the stored step-0 scores → float64 [B, n_context].
        Facts are the spans of the fact passage separated by token id 5 ('.'), starting at index 2.  One launch of
        `lako_fact_scores` (a workgroup per sample: masked head/layer sum, span cut, mean | max | 21mean) replaces the
        reference's per-sample Python loops; the result comes back as a CPU tensor like the reference's."""
        assert opt.stream == 2
        scores = self._score_storage
        if scores is None:
            raise RuntimeError("no scores stored: call overwrite_forward_crossattention() and generate() first")
        eng = self._get_engine()
        B, H, nl, S = scores.shape
        N, L = context_mask.size(1), context_mask.size(2)
        if S != N * L:
            raise ValueError(f"stored scores cover {S} keys, the batch has {N}x{L}")
        # torch.chunk(·, 2): the second chunk starts at ceil(n / 2) — the later half of the layers, the fact passage(s)
        layer0 = -(-nl // 2) if opt.use_last_half_layer_attention == "yes" else 0
        passage = -(-N // 2)       # scores: first passage of the second chunk (src/model.py:164,174) …
        ids_passage = 1            # … token ids: context_ids[b][1] (:173) — the same passage when N = 2
        if N < 2:
            raise ValueError("get_crossattention_scores needs the stream-2 layout: at least 2 passages")
        dev = scores.device
        out = torch.empty(B, opt.n_context, dtype=torch.float64, device=dev)
        eng.ops.fact_scores(scores.contiguous(), context_mask.to(dev).to(torch.uint8).contiguous(),
                            context_ids.to(dev).contiguous(), out, layer0=layer0, layers_used=nl - layer0, passage=passage,
                            ids_passage=ids_passage, style=opt.attention_score_style)
        return out.cpu()
