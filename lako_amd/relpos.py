"""Host-side relative-position bucket table (T5Attention._relative_position_bucket, HF5:217-262).

The bias of T5 attention depends only on rel = key_pos - query_pos, so the host computes one int32
LUT over rel ∈ [-(Lq-1), Lk-1] and the device expands the [num_buckets, H] embedding through it
(lako_relpos_expand).  The log-spaced boundaries are evaluated in float32, step by step as the
reference does, and are pinned by tests/golden/tables.npz (generated from transformers itself)."""
from __future__ import annotations

import math

import numpy as np


def relative_position_bucket(rel: np.ndarray, bidirectional: bool, num_buckets: int = 32,
                             max_distance: int = 128) -> np.ndarray:
    rel = rel.astype(np.int64)
    buckets = np.zeros_like(rel)
    if bidirectional:
        num_buckets //= 2
        buckets = buckets + (rel > 0).astype(np.int64) * num_buckets
        rel = np.abs(rel)
    else:
        rel = -np.minimum(rel, 0)
    max_exact = num_buckets // 2
    is_small = rel < max_exact
    with np.errstate(divide="ignore"):
        ratio = np.log(rel.astype(np.float32) / np.float32(max_exact))
    val = ratio / np.float32(math.log(max_distance / max_exact)) * np.float32(num_buckets - max_exact)
    val = np.where(np.isfinite(val), val, np.float32(0))
    large = max_exact + val.astype(np.float32).astype(np.int64)      # .to(torch.long) truncates toward zero
    large = np.minimum(large, num_buckets - 1)
    return buckets + np.where(is_small, rel, large)


def bucket_lut(q_len: int, k_len: int, bidirectional: bool, num_buckets: int, max_distance: int) -> np.ndarray:
    """int32 [q_len + k_len - 1]; entry r is the bucket of rel = r - (q_len - 1)."""
    rel = np.arange(-(q_len - 1), k_len, dtype=np.int64)
    return relative_position_bucket(rel, bidirectional, num_buckets, max_distance).astype(np.int32)
