"""CPU oracle for the LaKo Fusion-in-Decoder reader hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32) restatement of the algorithm the reference executes for
`FiDT5.forward` / `FiDT5.generate` / one `train_reader.py` step.  It is *not* part of the product:
only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it, and
there only as the checker (or the timed CPU baseline), never as the thing shipped.

Parity pin: the functions below are checked against golden vectors produced by running the
reference's own `src/model.py` (with three harness-side shims, see `oracle/make_fixtures.py`) on
top of the installed `transformers` 5.15.0 `modeling_t5.py` — the third-party file that really holds
the arithmetic (the reference pins "transformers 3.0.2" in prose only, README.md:21; 3.0.2 is not
installable offline).  Fixtures live in `tests/golden/`; `tests/test_oracle_golden.py` does the check.

Citations: `path:line` is relative to /root/reference; `HF5:line` is
transformers/models/t5/modeling_t5.py of transformers 5.15.0.

Weights are a dict keyed by plain HF-T5 names (what `FiDT5.load_t5` consumes, src/model.py:79-82):
  shared.weight, {encoder,decoder}.block.{i}.layer.{j}.{SelfAttention,EncDecAttention}.{q,k,v,o}.weight,
  ...SelfAttention.relative_attention_bias.weight (block 0 only), ...layer_norm.weight,
  ...DenseReluDense.{wi,wo}.weight, {encoder,decoder}.final_layer_norm.weight.
"""
from __future__ import annotations

import heapq
import math
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class T5Dims:
    vocab_size: int = 32128
    d_model: int = 512
    d_kv: int = 64
    d_ff: int = 2048
    num_layers: int = 6
    num_decoder_layers: int = 6
    num_heads: int = 8
    num_buckets: int = 32
    max_distance: int = 128
    eps: float = 1e-6
    dropout: float = 0.1
    # transformers 3.0.2 (the version the reference pins, README.md:21): the decoder's first cross-attention layer owns a
    # relative-position table and every cross-attention adds its bias (src/model.py:301-303,323-329).  False = transformers >= 4.
    legacy_cross_bias: bool = False

    @staticmethod
    def named(size: str) -> "T5Dims":
        if size == "small":
            return T5Dims()
        if size == "base":
            return T5Dims(d_model=768, d_ff=3072, num_layers=12, num_decoder_layers=12, num_heads=12)
        if size == "large":
            return T5Dims(d_model=1024, d_ff=4096, num_layers=24, num_decoder_layers=24, num_heads=16)
        if size == "tiny":  # the fixture config
            return T5Dims(vocab_size=64, d_model=32, d_kv=32, d_ff=64, num_layers=2, num_decoder_layers=2,
                          num_heads=2)
        raise ValueError(size)


# ----------------------------------------------------------------------------------------------
# parameter naming / init
# ----------------------------------------------------------------------------------------------
XREL = "decoder.block.0.layer.1.EncDecAttention.relative_attention_bias.weight"


def param_shapes(dims: T5Dims) -> dict[str, tuple[int, ...]]:
    """Plain-T5 state-dict layout (HF5:640-655, 176-215, 75-82, 50-57)."""
    inner = dims.num_heads * dims.d_kv
    out: dict[str, tuple[int, ...]] = {"shared.weight": (dims.vocab_size, dims.d_model)}
    for stack, nl in (("encoder", dims.num_layers), ("decoder", dims.num_decoder_layers)):
        for i in range(nl):
            p = f"{stack}.block.{i}.layer."
            for n in "qkv":
                out[p + f"0.SelfAttention.{n}.weight"] = (inner, dims.d_model)
            out[p + "0.SelfAttention.o.weight"] = (dims.d_model, inner)
            if i == 0:
                out[p + "0.SelfAttention.relative_attention_bias.weight"] = (dims.num_buckets, dims.num_heads)
            out[p + "0.layer_norm.weight"] = (dims.d_model,)
            ff = 1
            if stack == "decoder":
                for n in "qkv":
                    out[p + f"1.EncDecAttention.{n}.weight"] = (inner, dims.d_model)
                out[p + "1.EncDecAttention.o.weight"] = (dims.d_model, inner)
                out[p + "1.layer_norm.weight"] = (dims.d_model,)
                ff = 2
            out[p + f"{ff}.DenseReluDense.wi.weight"] = (dims.d_ff, dims.d_model)
            out[p + f"{ff}.DenseReluDense.wo.weight"] = (dims.d_model, dims.d_ff)
            out[p + f"{ff}.layer_norm.weight"] = (dims.d_model,)
        out[f"{stack}.final_layer_norm.weight"] = (dims.d_model,)
    if dims.legacy_cross_bias:      # (last: the other weights of a seed do not depend on the flag)
        out[XREL] = (dims.num_buckets, dims.num_heads)
    return out


def init_weights(dims: T5Dims, seed: int = 0, shared_std: float | None = None) -> dict[str, torch.Tensor]:
    """HF-style init (HF5:563-616) with `shared` std scaled down so greedy decode is not degenerate
    (SURVEY.md A.5).  Deterministic in `seed`."""
    g = torch.Generator().manual_seed(seed)
    d, dk, H, f = dims.d_model, dims.d_kv, dims.num_heads, dims.d_ff
    if shared_std is None:
        shared_std = d ** -0.5 * 4.0
    w: dict[str, torch.Tensor] = {}
    for name, shape in param_shapes(dims).items():
        if name.endswith("layer_norm.weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name == "shared.weight":
            t = torch.randn(shape, generator=g) * shared_std
        elif name.endswith(".q.weight"):
            t = torch.randn(shape, generator=g) * (d * dk) ** -0.5
        elif name.endswith(".k.weight") or name.endswith(".v.weight"):
            t = torch.randn(shape, generator=g) * d ** -0.5
        elif name.endswith(".o.weight"):
            t = torch.randn(shape, generator=g) * (H * dk) ** -0.5
        elif name.endswith("relative_attention_bias.weight"):
            t = torch.randn(shape, generator=g) * 0.5
        elif name.endswith("wi.weight"):
            t = torch.randn(shape, generator=g) * d ** -0.5
        elif name.endswith("wo.weight"):
            t = torch.randn(shape, generator=g) * f ** -0.5
        else:
            raise KeyError(name)
        w[name] = t.float()
    return w


# ----------------------------------------------------------------------------------------------
# T5 pieces
# ----------------------------------------------------------------------------------------------
def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """HF5:59-72 — scale only, fp32 variance, no mean subtraction, no bias."""
    var = x.float().pow(2).mean(-1, keepdim=True)
    return weight * (x * torch.rsqrt(var + eps))


def relative_position_bucket(rel: torch.Tensor, bidirectional: bool, num_buckets: int = 32,
                             max_distance: int = 128) -> torch.Tensor:
    """HF5:217-262.  `rel = memory_position - query_position` (int64).  The log term is evaluated in
    float32 exactly as the reference does (boundaries listed in SURVEY.md A.1)."""
    buckets = torch.zeros_like(rel)
    if bidirectional:
        num_buckets //= 2
        buckets = buckets + (rel > 0).to(torch.long) * num_buckets
        rel = torch.abs(rel)
    else:
        rel = -torch.min(rel, torch.zeros_like(rel))
    max_exact = num_buckets // 2
    is_small = rel < max_exact
    large = max_exact + (
        torch.log(rel.float() / max_exact) / math.log(max_distance / max_exact) * (num_buckets - max_exact)
    ).to(torch.long)
    large = torch.min(large, torch.full_like(large, num_buckets - 1))
    return buckets + torch.where(is_small, rel, large)


def compute_bias(table: torch.Tensor, qlen: int, klen: int, bidirectional: bool, num_buckets: int,
                 max_distance: int, past: int = 0) -> torch.Tensor:
    """HF5:264-279 → [1, H, qlen, klen]."""
    ctx = torch.arange(qlen, dtype=torch.long)[:, None] + past
    mem = torch.arange(klen, dtype=torch.long)[None, :]
    b = relative_position_bucket(mem - ctx, bidirectional, num_buckets, max_distance)
    return table[b].permute(2, 0, 1).unsqueeze(0)


def _drop(x: torch.Tensor, p: float, training: bool) -> torch.Tensor:
    return F.dropout(x, p=p, training=training) if (training and p > 0) else x


def attention(x_q: torch.Tensor, x_kv: torch.Tensor, wq, wk, wv, wo, H: int, dk: int,
              add_mask: torch.Tensor | None, bias: torch.Tensor | None, p: float, training: bool,
              capture: dict | None = None) -> torch.Tensor:
    """HF5:281-369 + eager_attention_forward HF5:144-173 (same math as the reference's own
    `cross_attention_forward`, src/model.py:286-349): scores are NOT scaled by 1/sqrt(dk); the
    additive position bias and mask are added before an fp32 softmax; dropout on the probabilities."""
    B, Lq, _ = x_q.shape
    Lk = x_kv.shape[1]
    q = (x_q @ wq.t()).view(B, Lq, H, dk).transpose(1, 2)
    k = (x_kv @ wk.t()).view(B, Lk, H, dk).transpose(1, 2)
    v = (x_kv @ wv.t()).view(B, Lk, H, dk).transpose(1, 2)
    scores = torch.matmul(q, k.transpose(2, 3))
    if bias is not None:
        scores = scores + bias
    if add_mask is not None:
        scores = scores + add_mask
    if capture is not None:
        capture.update(q=q, k=k, v=v, scores=scores)
    probs = F.softmax(scores.float(), dim=-1)
    if capture is not None:
        capture.update(probs=probs)
    probs = _drop(probs, p, training)
    ctx = torch.matmul(probs, v).transpose(1, 2).reshape(B, Lq, H * dk)
    if capture is not None:
        capture.update(ctx=ctx)
    return ctx @ wo.t()


NEG = torch.finfo(torch.float32).min  # HF5 masks with finfo.min (3.0.2 used -10000)


def encoder_stack(w, dims: T5Dims, ids: torch.Tensor, mask: torch.Tensor, training: bool,
                  capture: dict | None = None) -> torch.Tensor:
    """T5Stack.forward for the encoder, HF5:663-750.  ids/mask are [B*N, L]."""
    p = dims.dropout
    h = _drop(w["shared.weight"][ids], p, training)                      # HF5:678,725
    L = ids.shape[1]
    add_mask = torch.zeros(mask.shape, dtype=torch.float32).masked_fill(~mask.bool(), NEG)[:, None, None, :]
    bias = compute_bias(w["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"], L, L, True,
                        dims.num_buckets, dims.max_distance)           # block 0 owns it, all share (HF5:739-742)
    for i in range(dims.num_layers):
        pre = f"encoder.block.{i}.layer."
        cap = {} if (capture is not None and i == 0) else None
        xn = rmsnorm(h, w[pre + "0.layer_norm.weight"], dims.eps)
        a = attention(xn, xn, w[pre + "0.SelfAttention.q.weight"], w[pre + "0.SelfAttention.k.weight"],
                      w[pre + "0.SelfAttention.v.weight"], w[pre + "0.SelfAttention.o.weight"],
                      dims.num_heads, dims.d_kv, add_mask, bias, p, training, cap)
        h = h + _drop(a, p, training)                                    # HF5:384-401
        xn2 = rmsnorm(h, w[pre + "1.layer_norm.weight"], dims.eps)
        ff = _drop(F.relu(xn2 @ w[pre + "1.DenseReluDense.wi.weight"].t()), p, training)
        ff = ff @ w[pre + "1.DenseReluDense.wo.weight"].t()              # HF5:83-94
        h = h + _drop(ff, p, training)                                   # HF5:137-141
        if cap is not None:
            capture.update({"enc0_" + k: v for k, v in cap.items()})
            capture.update(enc0_xn=xn, enc0_bias=bias, enc0_ffn_out=ff, enc0_out=h)
    h = rmsnorm(h, w["encoder.final_layer_norm.weight"], dims.eps)
    return _drop(h, p, training)                                         # HF5:744-745


def shift_right(labels: torch.Tensor) -> torch.Tensor:
    """HF5:618-637: prepend decoder_start_token_id (=0), drop last, -100 → pad (=0)."""
    out = labels.new_zeros(labels.shape)
    out[..., 1:] = labels[..., :-1]
    out[..., 0] = 0
    return out.masked_fill(out == -100, 0)


def decoder_stack(w, dims: T5Dims, dec_ids: torch.Tensor, enc_out: torch.Tensor, enc_mask: torch.Tensor,
                  training: bool, capture: dict | None = None, legacy_pos0: bool = False) -> torch.Tensor:
    """T5Stack.forward for the decoder, HF5:663-750 (+ cross-attention HF5:404-432).  No
    decoder_attention_mask is passed by the reference (train_reader.py:67-71), so the self-attention
    mask is purely causal; cross-attention has no positional term (HF >= 4: zeros, HF5:337-342).

    `dims.legacy_cross_bias` (transformers 3.0.2, what src/model.py:301-303,323-329 is written against): block 0's
    cross-attention computes `compute_bias(qlen, klen)` from ITS table with the decoder's one-sided buckets
    (`bidirectional = not is_decoder`) and every block adds it to the scores before the softmax; the stored score
    includes it.  `legacy_pos0`: the bias of a decode step under the reference's own cross-attention forward with the
    key / value cache — sized by the step's query length 1 (src/model.py:305,323-324), i.e. the row of position 0 for
    every step (stock 3.0.2 slices the row of the step's true position instead; that is `legacy_pos0 = False` on the
    recomputed prefix)."""
    p = dims.dropout
    B, T = dec_ids.shape
    h = _drop(w["shared.weight"][dec_ids], p, training)
    causal = torch.zeros(T, T).masked_fill(torch.triu(torch.ones(T, T, dtype=torch.bool), 1), NEG)[None, None]
    bias = compute_bias(w["decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"], T, T, False,
                        dims.num_buckets, dims.max_distance)
    xmask = torch.zeros(enc_mask.shape, dtype=torch.float32).masked_fill(~enc_mask.bool(), NEG)[:, None, None, :]
    xbias = None
    if dims.legacy_cross_bias:
        xbias = compute_bias(w[XREL], 1 if legacy_pos0 else T, enc_out.shape[1], False, dims.num_buckets, dims.max_distance)
    for i in range(dims.num_decoder_layers):
        pre = f"decoder.block.{i}.layer."
        xn = rmsnorm(h, w[pre + "0.layer_norm.weight"], dims.eps)
        a = attention(xn, xn, w[pre + "0.SelfAttention.q.weight"], w[pre + "0.SelfAttention.k.weight"],
                      w[pre + "0.SelfAttention.v.weight"], w[pre + "0.SelfAttention.o.weight"],
                      dims.num_heads, dims.d_kv, causal, bias, p, training)
        h = h + _drop(a, p, training)
        xn = rmsnorm(h, w[pre + "1.layer_norm.weight"], dims.eps)
        cap = {} if capture is not None else None
        a = attention(xn, enc_out, w[pre + "1.EncDecAttention.q.weight"], w[pre + "1.EncDecAttention.k.weight"],
                      w[pre + "1.EncDecAttention.v.weight"], w[pre + "1.EncDecAttention.o.weight"],
                      dims.num_heads, dims.d_kv, xmask, xbias, p, training, cap)
        if cap is not None:
            capture[f"cross_scores_{i}"] = cap["scores"]
        h = h + _drop(a, p, training)
        xn = rmsnorm(h, w[pre + "2.layer_norm.weight"], dims.eps)
        ff = _drop(F.relu(xn @ w[pre + "2.DenseReluDense.wi.weight"].t()), p, training)
        h = h + _drop(ff @ w[pre + "2.DenseReluDense.wo.weight"].t(), p, training)
    h = rmsnorm(h, w["decoder.final_layer_norm.weight"], dims.eps)
    return _drop(h, p, training)


def fid_encode(w, dims: T5Dims, input_ids: torch.Tensor, attention_mask: torch.Tensor, training: bool,
               capture: dict | None = None):
    """FiDT5.forward + EncoderWrapper.forward reshapes: src/model.py:39-51, 227-234.
    [B,N,L] → encoder on [B*N, L] → view [B, N*L, d]; mask → [B, N*L]."""
    B, N, L = input_ids.shape
    enc = encoder_stack(w, dims, input_ids.reshape(B * N, L), attention_mask.reshape(B * N, L), training, capture)
    return enc.view(B, N * L, -1), attention_mask.reshape(B, N * L)


def fid_forward(w, dims: T5Dims, input_ids, attention_mask, labels, training: bool = False,
                capture: dict | None = None):
    """Loss + logits exactly as T5ForConditionalGeneration.forward, HF5:1008-1066:
    encoder → shift_right(labels) → decoder → × d_model^-0.5 → tied LM head → CE(ignore_index=-100), mean."""
    enc, enc_mask = fid_encode(w, dims, input_ids, attention_mask, training, capture)
    dec_ids = shift_right(labels)
    h = decoder_stack(w, dims, dec_ids, enc, enc_mask, training, capture)
    h = h * dims.d_model ** -0.5                                         # HF5:1044-1045
    logits = h @ w["shared.weight"].t()                                 # tied lm_head, HF5:902-906,1047
    loss = F.cross_entropy(logits.view(-1, logits.size(-1)), labels.view(-1), ignore_index=-100)
    if capture is not None:
        capture.update(enc_out=enc, logits=logits)
    return loss, logits


def fid_generate(w, dims: T5Dims, input_ids, attention_mask, max_length: int,
                 capture: dict | None = None) -> torch.Tensor:
    """Greedy decode as FiDT5.generate (src/model.py:54-60) → HF greedy search with defaults
    num_beams=1, do_sample=False: start token 0; each step argmax of the last position; rows that
    already produced EOS(1) emit pad(0); stop when every row has produced EOS or at max_length
    (which counts the start token).  SURVEY.md A.9.  (Recomputes the decoder prefix each step —
    mathematically identical to the KV-cached loop.)"""
    enc, enc_mask = fid_encode(w, dims, input_ids, attention_mask, False)
    B = input_ids.shape[0]
    seq = torch.zeros(B, 1, dtype=torch.long)
    done = torch.zeros(B, dtype=torch.bool)
    for step in range(max_length - 1):
        cap = {} if (capture is not None and step == 0) else None
        h = decoder_stack(w, dims, seq, enc, enc_mask, False, cap, legacy_pos0=capture is not None)
        if cap is not None:
            capture.update(cap)
        logits = (h[:, -1] * dims.d_model ** -0.5) @ w["shared.weight"].t()
        nxt = logits.argmax(-1)
        nxt = torch.where(done, torch.zeros_like(nxt), nxt)
        seq = torch.cat([seq, nxt[:, None]], 1)
        done = done | (nxt == 1)
        if bool(done.all()):
            break
    return seq


# ----------------------------------------------------------------------------------------------
# optimizer / schedule / train step  (src/util.py:149-168,185-245; train_reader.py:62-84)
# ----------------------------------------------------------------------------------------------
def lr_lambda(step: int, warmup_steps: int, scheduler_steps: int, min_ratio: float = 0.0,
              fixed_lr: bool = False) -> float:
    """WarmupLinearScheduler.lr_lambda, src/util.py:159-168."""
    if step < warmup_steps:
        return (1 - min_ratio) * step / float(max(1, warmup_steps)) + min_ratio
    if fixed_lr:
        return 1.0
    return max(0.0, 1.0 + (min_ratio - 1) * (step - warmup_steps) / float(max(1.0, scheduler_steps - warmup_steps)))


def clip_grad_norm(grads: dict[str, torch.Tensor], max_norm: float) -> float:
    """torch.nn.utils.clip_grad_norm_ (train_reader.py:76): total L2 norm; scale by
    max_norm/(norm+1e-6) clamped to 1."""
    total = torch.sqrt(sum((g.float() ** 2).sum() for g in grads.values()))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads.values():
        g.mul_(coef)
    return float(total)


def adamw_nobc_step(w, grads, state, lr: float, wd: float, b1=0.9, b2=0.999, eps=1e-6):
    """HF <= 4 `AdamW(correct_bias=False)` built at src/util.py:225 (class removed from HF 5; semantics
    restated from SURVEY.md A.6): m ← b1 m + (1-b1) g; v ← b2 v + (1-b2) g²; p ← p − lr·m/(√v+eps);
    then decoupled decay p ← p − lr·wd·p.  Both param groups carry the same wd (src/util.py:189-194)."""
    for k, p in w.items():
        g = grads[k]
        st = state.setdefault(k, {"m": torch.zeros_like(p), "v": torch.zeros_like(p)})
        st["m"].mul_(b1).add_(g, alpha=1 - b1)
        st["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
        p.addcdiv_(st["m"], st["v"].sqrt().add_(eps), value=-lr)
        if wd > 0:
            p.add_(p, alpha=-lr * wd)


def train_step(w, dims, state, input_ids, attention_mask, labels, step_index: int, base_lr: float, wd: float,
               clip: float, warmup_steps: int, total_steps: int, training: bool = True):
    """One iteration of train_reader.py:67-82: fwd → bwd → clip_grad_norm_(clip) → optimizer.step()
    → scheduler.step() → zero_grad.  The lr used at the k-th optimizer step (k = 0, 1, …) is
    base_lr·lr_lambda(k) (LambdaLR applies lambda(0) at construction, then steps after each optimizer step)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in w.items()}
    loss, _ = fid_forward(leaves, dims, input_ids, attention_mask, labels, training=training)
    loss.backward()
    grads = {k: v.grad for k, v in leaves.items()}
    gnorm = clip_grad_norm(grads, clip)
    lr = base_lr * lr_lambda(step_index, warmup_steps, total_steps)
    with torch.no_grad():
        adamw_nobc_step(w, grads, state, lr, wd)
    return float(loss), gnorm


# ----------------------------------------------------------------------------------------------
# cross-attention score aggregation  (src/model.py:100-115,143-204)  — SURVEY.md §8 f1
# ----------------------------------------------------------------------------------------------
def _span_score(vals, start, end, style):
    """get_attention_score, src/model.py:107-115."""
    if style == "mean":
        return sum(vals[start:end]) / (end - start)
    if style == "max":
        return max(vals[start:end])
    if style == "21mean":
        num = max(int((end - start + 1) / 2), 1)
        return sum(heapq.nlargest(num, vals[start:end])) / num
    raise ValueError(style)


def crossattention_fact_scores(scores: torch.Tensor, context_ids: torch.Tensor, context_mask: torch.Tensor,
                               n_context: int, style: str = "mean", last_half: bool = False) -> torch.Tensor:
    """get_crossattention_scores for stream == 2 (src/model.py:143-204).
    `scores`: step-0 pre-softmax cross-attention scores concatenated over decoder layers,
    [B, H, n_layers, N*L];  returns float64 [B, n_context]."""
    if last_half:
        scores = torch.chunk(scores, 2, dim=2)[1]
    B, H, nl, _ = scores.shape
    N = context_mask.size(1)
    s = scores.view(B, H, nl, N, -1).masked_fill(~context_mask[:, None, None].bool(), 0.0)
    fact = torch.chunk(s, 2, dim=3)[1].sum(dim=[1, 2])           # passage 1 = the fact passage
    out = []
    for b in range(B):
        ids = context_ids[b][1].tolist()
        vals = fact[b][0].tolist()
        res, start = [], 2
        for _ in range(n_context):
            try:
                end = ids.index(5, start) + 1
            except ValueError:
                break
            res.append(_span_score(vals, start, end, style))
            start = end
        if len(res) < n_context and ids[-1] != 0:
            end = len(ids)
            if end > start:
                res.append(_span_score(vals, start, end, style))
        while len(res) < n_context:
            res.append(-5)
        out.append(np.array(res[:n_context] if len(res) > n_context else res))
    return torch.from_numpy(np.array(out)) / (nl * H)


# ----------------------------------------------------------------------------------------------
# synthetic batches — shared by tests, bench.py (GPU leg builds its own on device
# from the same recipe) and the cpu_baseline leg.
# ----------------------------------------------------------------------------------------------
def synthetic_batch(B: int, N: int, L: int, T: int, vocab: int, seed: int, ragged: bool = True):
    g = torch.Generator().manual_seed(seed)
    hi = min(vocab, 32100)
    ids = torch.randint(2, hi, (B, N, L), generator=g)
    if ragged:
        lens = torch.randint((L + 1) // 2, L + 1, (B, N), generator=g)
    else:
        lens = torch.full((B, N), L)
    mask = torch.arange(L)[None, None, :] < lens[..., None]
    ids = ids.masked_fill(~mask, 0)
    labels = torch.randint(2, hi, (B, T), generator=g)
    tl = torch.randint(2, T + 1, (B,), generator=g) if T >= 2 else torch.ones(B, dtype=torch.long)
    pos = torch.arange(T)[None]
    labels = torch.where(pos == (tl - 1)[:, None], torch.ones_like(labels), labels)
    labels = labels.masked_fill(pos >= tl[:, None], -100)
    return ids, mask, labels
