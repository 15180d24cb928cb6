"""Model dimensions of the FiD reader (what `FiDT5(config)` needs from an HF `T5Config`, train_reader.py:243-244)."""
from __future__ import annotations

import json
from dataclasses import asdict, dataclass


@dataclass
class FiDConfig:
    vocab_size: int = 32128
    d_model: int = 512
    d_kv: int = 64
    d_ff: int = 2048
    num_layers: int = 6
    num_decoder_layers: int = 6
    num_heads: int = 8
    relative_attention_num_buckets: int = 32
    relative_attention_max_distance: int = 128
    dropout_rate: float = 0.1
    layer_norm_epsilon: float = 1e-6
    decoder_start_token_id: int = 0
    pad_token_id: int = 0
    eos_token_id: int = 1
    # transformers 3.0.2 — the version the reference pins (README.md:21) — gives the decoder's FIRST cross-attention layer a
    # relative-position table and adds its bias in every cross-attention (src/model.py:301-303,323-329); from transformers 4 on
    # the table does not exist and the bias is zero.  False: the table of a checkpoint is ignored (the semantics of every current
    # transformers and of the golden fixtures); True: the table is a parameter and the bias is applied (engine.py "legacy").
    legacy_cross_bias: bool = False

    @property
    def inner_dim(self) -> int:
        return self.num_heads * self.d_kv

    @staticmethod
    def named(size: str, **over) -> "FiDConfig":
        """t5-small / t5-base / t5-large dimensions (the `--model_size` flag, src/options.py:57)."""
        table = {
            "tiny": dict(vocab_size=64, d_model=32, d_kv=32, d_ff=64, num_layers=2, num_decoder_layers=2, num_heads=2),
            "small": dict(),
            "base": dict(d_model=768, d_ff=3072, num_layers=12, num_decoder_layers=12, num_heads=12),
            "large": dict(d_model=1024, d_ff=4096, num_layers=24, num_decoder_layers=24, num_heads=16),
        }
        if size not in table:
            raise ValueError(f"unknown model size {size!r}")
        kw = dict(table[size])
        kw.update(over)
        return FiDConfig(**kw)

    @staticmethod
    def from_hf(cfg) -> "FiDConfig":
        """Accept an HF `T5Config`-like object or a dict (duck-typed on the fields we use)."""
        get = (lambda k, d=None: cfg.get(k, d)) if isinstance(cfg, dict) else (lambda k, d=None: getattr(cfg, k, d))
        ff = get("feed_forward_proj", "relu")
        if ff not in (None, "relu"):
            raise ValueError(f"only the T5 v1.0 ReLU feed-forward is supported, got {ff!r}")
        return FiDConfig(
            vocab_size=get("vocab_size"), d_model=get("d_model"), d_kv=get("d_kv"), d_ff=get("d_ff"),
            num_layers=get("num_layers"), num_decoder_layers=get("num_decoder_layers") or get("num_layers"),
            num_heads=get("num_heads"), relative_attention_num_buckets=get("relative_attention_num_buckets", 32),
            relative_attention_max_distance=get("relative_attention_max_distance", 128),
            dropout_rate=get("dropout_rate", 0.1), layer_norm_epsilon=get("layer_norm_epsilon", 1e-6),
            decoder_start_token_id=get("decoder_start_token_id", 0) or 0, pad_token_id=get("pad_token_id", 0) or 0,
            eos_token_id=get("eos_token_id", 1) if get("eos_token_id", 1) is not None else 1,
            legacy_cross_bias=bool(get("legacy_cross_bias", False)))

    def to_dict(self) -> dict:
        d = asdict(self)
        d.update(model_type="t5", architectures=["FiDT5"], feed_forward_proj="relu", is_encoder_decoder=True,
                 tie_word_embeddings=True)
        return d

    def to_json(self) -> str:
        return json.dumps(self.to_dict(), indent=2, sort_keys=True)
