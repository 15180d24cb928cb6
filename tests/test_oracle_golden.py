"""Pin the oracle (oracle/fid_t5_oracle.py) to golden vectors produced by the reference itself."""
import numpy as np
import pytest
import torch

from oracle import fid_t5_oracle as O
from tests.util_golden import group, load_case

CASES = ["tiny_a", "tiny_fact", "mid_a", "tiny_eos"]


def test_bucket_tables(golden_dir):
    z = np.load(golden_dir + "/tables.npz")
    rel = torch.from_numpy(z["rel"])
    assert torch.equal(O.relative_position_bucket(rel, True), torch.from_numpy(z["enc_bucket"]))
    assert torch.equal(O.relative_position_bucket(rel, False), torch.from_numpy(z["dec_bucket"]))
    # spot values from SURVEY.md A.1
    enc = dict(zip(z["rel"].tolist(), z["enc_bucket"].tolist()))
    assert enc[-91] == 15 and enc[-90] == 14 and enc[0] == 0 and enc[1] == 17 and enc[91] == 31 and enc[8] == 24
    dec = dict(zip(z["rel"].tolist(), z["dec_bucket"].tolist()))
    assert dec[5] == 0 and dec[-15] == 15 and dec[-16] == 16 and dec[-113] == 31


def test_lr_lambda(golden_dir):
    z = np.load(golden_dir + "/tables.npz")
    mine = np.array([O.lr_lambda(s, 6, 100) for s in range(111)])
    np.testing.assert_allclose(mine, z["lr_lambda_w6_t100"], rtol=0, atol=0)


@pytest.mark.parametrize("name", CASES)
def test_forward_backward(name):
    z, dims, w = load_case(name)
    ids, mask, labels = (torch.from_numpy(z[k]) for k in ("input_ids", "attention_mask", "labels"))
    leaves = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    cap = {}
    loss, logits = O.fid_forward(leaves, dims, ids, mask, labels, training=False, capture=cap)
    assert abs(loss.item() - float(z["loss"])) < 1e-5
    torch.testing.assert_close(logits, torch.from_numpy(z["logits"]), atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(cap["enc_out"], torch.from_numpy(z["enc_out"]), atol=2e-5, rtol=1e-5)
    acts = group(z, "act/")
    B, N, L = ids.shape
    H, dk = dims.num_heads, dims.d_kv
    torch.testing.assert_close(cap["enc0_xn"], acts["enc0_xn"], atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(cap["enc0_bias"], acts["enc0_bias"], atol=0, rtol=0)
    for n in "qkv":
        ref = acts["enc0_" + n].view(B * N, L, H, dk).transpose(1, 2)
        torch.testing.assert_close(cap["enc0_" + n], ref, atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(cap["enc0_ctx"], acts["enc0_ctx"], atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(cap["enc0_ffn_out"], acts["enc0_ffn_out"], atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(cap["enc0_out"], acts["enc0_out"], atol=2e-5, rtol=1e-5)
    loss.backward()
    g = group(z, "g/")
    assert set(g) == set(leaves)
    for k, v in leaves.items():
        torch.testing.assert_close(v.grad, g[k], atol=3e-5, rtol=1e-4, msg=lambda m, k=k: f"{k}: {m}")
    gn = torch.sqrt(sum((v.grad.double() ** 2).sum() for v in leaves.values())).item()
    assert abs(gn - float(z["grad_norm"])) < 1e-4


@pytest.mark.parametrize("name", CASES)
def test_train_steps(name):
    z, dims, w = load_case(name)
    B, N, L = z["input_ids"].shape
    T = z["labels"].shape[1]
    lr, wd, clip, warm, total = z["train_hparams"].tolist()
    seed = {"tiny_a": 1, "tiny_fact": 2, "mid_a": 3, "tiny_eos": 5}[name]
    state = {}
    for k in range(3):
        bi, bm, bl = O.synthetic_batch(B, N, L, T, dims.vocab_size, seed=seed + 200 + k)
        loss, gn = O.train_step(w, dims, state, bi, bm, bl, k, lr, wd, clip, int(warm), int(total), training=False)
        assert abs(loss - z["train_losses"][k]) < 2e-5
        assert abs(gn - z["train_gnorms"][k]) < 1e-4
        if k in (0, 2):
            ref = group(z, f"w_step{k + 1}/")
            for n, v in w.items():
                torch.testing.assert_close(v, ref[n], atol=2e-5, rtol=1e-4, msg=lambda m, n=n: f"{n}: {m}")


@pytest.mark.parametrize("name", CASES)
def test_generate_and_scores(name):
    z, dims, w = load_case(name)
    ids, mask = torch.from_numpy(z["input_ids"]), torch.from_numpy(z["attention_mask"])
    for ml in (4, 12):
        cap = {}
        toks = O.fid_generate(w, dims, ids, mask, ml, capture=cap)
        assert toks.tolist() == z[f"gen_{ml}"].tolist()
    scores = torch.cat([cap[f"cross_scores_{i}"] for i in range(dims.num_decoder_layers)], dim=2)
    ref = torch.from_numpy(z["cross_scores"])
    keep = mask.view(mask.shape[0], 1, 1, -1).expand_as(ref)
    torch.testing.assert_close(scores[keep], ref[keep], atol=2e-5, rtol=1e-5)
    if ids.shape[1] == 2:
        for style in ("mean", "max", "21mean"):
            for half in ("no", "yes"):
                mine = O.crossattention_fact_scores(ref, ids, mask, 5, style, half == "yes")
                np.testing.assert_allclose(mine.numpy(), z[f"fact_scores_{style}_{half}"], rtol=1e-12, atol=1e-12)


def test_chunked_oracle_evaluation_equals_direct():
    """tests/test_real_size_gpu.py evaluates the oracle at BASELINE config 5 (T5-large, 100 passages) with the encoder walked 20
    passages at a time to bound host memory; that evaluation must give the direct fid_forward + backward numbers."""
    from tests.test_real_size_gpu import oracle_fwd_bwd_chunked
    z, dims, w = load_case("mid_a")
    ids, mask, labels = O.synthetic_batch(1, 6, 16, 5, dims.vocab_size, seed=8)
    leaves = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    loss, logits = O.fid_forward(leaves, dims, ids, mask, labels, training=False)
    loss.backward()
    loss_c, logits_c, grads_c = oracle_fwd_bwd_chunked(w, dims, ids, mask, labels, chunk=4)      # chunks of 4 + 2 passages
    assert abs(loss_c - loss.item()) < 1e-6
    torch.testing.assert_close(logits_c, logits.detach(), atol=1e-5, rtol=1e-5)
    for k, v in leaves.items():
        torch.testing.assert_close(grads_c[k], v.grad, atol=1e-6, rtol=1e-4, msg=lambda m, k=k: f"{k}: {m}")


def test_per_sample_oracle_evaluation_equals_batch():
    """tests/test_real_size_gpu.py evaluates the oracle on the benchmark's 16-sample batch one sample at a time; that must give the
    batch's own loss / logits / gradients (mean over all valid label tokens of the batch)."""
    from tests.test_real_size_gpu import oracle_fwd_bwd_per_sample
    z, dims, w = load_case("mid_a")
    ids, mask, labels = O.synthetic_batch(3, 4, 16, 5, dims.vocab_size, seed=18)
    leaves = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    loss, logits = O.fid_forward(leaves, dims, ids, mask, labels, training=False)
    loss.backward()
    loss_s, logits_s, grads_s = oracle_fwd_bwd_per_sample(w, dims, ids, mask, labels)
    assert abs(loss_s - loss.item()) < 1e-6
    torch.testing.assert_close(logits_s, logits.detach(), atol=1e-5, rtol=1e-5)
    for k, v in leaves.items():
        torch.testing.assert_close(grads_s[k], v.grad, atol=1e-6, rtol=1e-4, msg=lambda m, k=k: f"{k}: {m}")


def test_generate_fixture_is_the_oracles_decode():
    """tests/golden/gen_d512.npz (oracle/make_generate_fixture.py): the expected tokens of the bf16 decode test ARE the oracle's fp32 greedy
    decode of the committed (bf16-representable) weights, and that decode reproduces the training answers — the GPU test compares the HIP
    path with these tokens without running the oracle again."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gen_d512.npz"))
    names = ("vocab_size", "d_model", "d_kv", "d_ff", "num_layers", "num_decoder_layers", "num_heads")
    dims = O.T5Dims(dropout=0.0, **{k: int(v) for k, v in zip(names, z["dims"])})
    w = {k[2:]: torch.from_numpy(z[k]).view(torch.bfloat16).float() for k in z.files if k.startswith("w.")}
    ids, mask, labels = (torch.from_numpy(z[k]) for k in ("ids", "mask", "labels"))
    tok = O.fid_generate(w, dims, ids, mask, int(z["max_length"]))
    assert tok.tolist() == z["tokens"].tolist()
    for b in range(ids.shape[0]):
        n = int((labels[b] != -100).sum())
        assert tok[b, 1:1 + n].tolist() == labels[b, :n].tolist()
    assert float(z["min_gap"]) > 1.0


def test_legacy_cross_bias_is_the_references_cross_attention(golden_dir):
    """`legacy_cross_bias` (transformers 3.0.2 semantics, src/model.py:301-303,323-329): tests/golden/legacy_cross.npz holds what the
    reference's OWN cross_attention_forward returns and stores when it is called the way it was written to be called (oracle/make_fixtures.py
    ::make_legacy_cross).  The oracle's attention with `compute_bias(table, qlen, klen, bidirectional=False)` reproduces the output, the
    stored score (bias included, masked keys as the reference leaves them) and the bias itself — for the training call (qlen = T) and for a
    cached decode step (qlen = 1: the row of position 0)."""
    z = np.load(golden_dir + "/legacy_cross.npz")
    w = {k: torch.from_numpy(z[k]) for k in ("q", "k", "v", "o", "table")}
    x, kv, keep = torch.from_numpy(z["x"]), torch.from_numpy(z["kv"]), torch.from_numpy(z["keep"])
    H, dk = w["table"].shape[1], w["q"].shape[0] // w["table"].shape[1]
    add_mask = torch.zeros(keep.shape).masked_fill(~keep, -1e9)[:, None, None, :]
    for tag, xin in (("full", x), ("step", x[:, 3:4])):
        bias = O.compute_bias(w["table"], xin.shape[1], kv.shape[1], False, 32, 128)
        assert torch.equal(bias, torch.from_numpy(z[f"bias_{tag}"]))
        cap = {}
        out = O.attention(xin, kv, w["q"], w["k"], w["v"], w["o"], H, dk, add_mask, bias, 0.0, False, cap)
        torch.testing.assert_close(out, torch.from_numpy(z[f"out_{tag}"]), atol=2e-6, rtol=1e-5)
        torch.testing.assert_close(cap["scores"], torch.from_numpy(z[f"scores_{tag}"]), atol=1e-5, rtol=1e-6)
    # the bias of a key at or after the query's position is ONE value per head; before it, one per distance (the decoder's buckets)
    b = torch.from_numpy(z["bias_full"])[0]
    for t in range(b.shape[1]):
        assert torch.equal(b[:, t, t:], w["table"][0][:, None].expand(-1, b.shape[2] - t))
        for n in range(1, t + 1):
            assert torch.equal(b[:, t, t - n], w["table"][n])


def test_legacy_cross_bias_oracle_model_level():
    """The flag at model level: a zero table is the model without the flag (same seed → same other weights), a non-zero one changes the
    loss; the bias reaches the gradient of the table through every decoder layer; greedy decode with the recomputed prefix uses each
    row's own position (stock 3.0.2), the capture path the row of position 0 (the reference's override: src/model.py:305,323)."""
    import dataclasses
    d0 = O.T5Dims.named("tiny")
    d1 = dataclasses.replace(d0, legacy_cross_bias=True)
    w0, w1 = O.init_weights(d0, 3), O.init_weights(d1, 3)
    assert set(w1) - set(w0) == {O.XREL} and all(torch.equal(w0[k], w1[k]) for k in w0)
    ids, mask, labels = O.synthetic_batch(2, 3, 16, 6, d0.vocab_size, 0)[:3]
    l0, _ = O.fid_forward(w0, d0, ids, mask, labels)
    wz = dict(w1)
    wz[O.XREL] = torch.zeros_like(w1[O.XREL])
    lz, _ = O.fid_forward(wz, d1, ids, mask, labels)
    assert torch.equal(l0, lz)
    leaves = {k: v.clone().requires_grad_(True) for k, v in w1.items()}
    l1, _ = O.fid_forward(leaves, d1, ids, mask, labels)
    assert abs(l1.item() - l0.item()) > 1e-4
    l1.backward()
    g = leaves[O.XREL].grad
    T = labels.shape[1]
    assert g[:T].abs().sum() > 0 and g[T:].abs().sum() == 0        # distances 0 … T − 1 only
    # rows of the score matrix sum to zero in the softmax backward, so the table's gradient sums to zero over the buckets in use
    assert abs(float(g.sum())) < 1e-5 * float(g.abs().sum())
    cap = {}
    O.fid_generate(w1, d1, ids, mask, 5, capture=cap)
    capz = {}
    O.fid_generate(wz, d1, ids, mask, 5, capture=capz)
    shift = cap["cross_scores_0"] - capz["cross_scores_0"]          # [B, H, 1, S]: the row of position 0 = bucket 0 for every key
    valid = mask.view(2, -1).bool()
    for h in range(d0.num_heads):
        assert torch.allclose(shift[:, h, 0][valid], w1[O.XREL][0, h].expand(int(valid.sum())), atol=1e-5)
