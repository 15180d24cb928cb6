"""-m gpu: end-to-end parity of the HIP path (lako_amd.FiDT5 on cuda:0 → C-ABI → gfx950 kernels).

  * against the golden vectors the REFERENCE produced (tests/golden/*.npz): loss, logits, encoder output,
    every parameter gradient, 3 optimizer steps, greedy tokens, step-0 cross-attention scores, per-fact
    scores.  fp32 mode: within 1e-3 (the tolerance BASELINE.json's north_star states); bf16 mode: loss within
    3 %, gradient direction cosine > 0.98 per tensor.
  * against the oracle (CPU restatement) at BASELINE config 1 (T5-small, n_passages=5, L=64, batch 2), fp32.
  * at BASELINE config 2 shapes (T5-base, n_passages=20, L=200) through size-independent properties:
    bf16 vs fp32 agreement of the same HIP path, dropout determinism, passage-permutation invariance of
    the loss (cross-attention has no positional term), padding invariance.
"""
import os
import types

import numpy as np
import pytest
import torch

from lako_amd import FiDConfig, FiDT5
from lako_amd import util as U
from lako_amd.model import plain_name
from oracle import fid_t5_oracle as O
from tests.util_golden import group, load_case

pytestmark = pytest.mark.gpu
CASES = ["tiny_a", "tiny_fact", "mid_a", "tiny_eos"]
SEEDS = {"tiny_a": 1, "tiny_fact": 2, "mid_a": 3, "tiny_eos": 5}
DEV = "cuda:0"


def cfg_of(dims, dropout=0.0):
    return FiDConfig(vocab_size=dims.vocab_size, d_model=dims.d_model, d_kv=dims.d_kv, d_ff=dims.d_ff,
                     num_layers=dims.num_layers, num_decoder_layers=dims.num_decoder_layers, num_heads=dims.num_heads,
                     relative_attention_num_buckets=dims.num_buckets,
                     relative_attention_max_distance=dims.max_distance, dropout_rate=dropout)


def build(name, dtype, dropout=0.0):
    z, dims, w = load_case(name)
    model = FiDT5(cfg_of(dims, dropout), dtype=dtype)
    model.load_t5(w)
    model = model.cuda()
    return z, dims, w, model


def dev(*ts):
    return tuple(torch.from_numpy(t).to(DEV) if isinstance(t, np.ndarray) else t.to(DEV) for t in ts)


def test_hip_library_is_what_runs():
    from lako_amd import _lib
    from lako_amd.ops import HipOps
    ops = HipOps()
    assert ops.lib.lako_version() == 4
    import os
    with open(f"/proc/{os.getpid()}/maps") as f:
        assert "liblako_hip.so" in f.read()
    assert os.path.samefile(_lib.LIB_PATH, os.path.join(os.path.dirname(_lib.__file__), "liblako_hip.so"))


@pytest.mark.parametrize("name", CASES)
def test_fp32_forward_backward_vs_reference(name):
    z, dims, w, model = build(name, torch.float32)
    ids, mask, labels = dev(z["input_ids"], z["attention_mask"], z["labels"])
    model.train()
    out = model(input_ids=ids, attention_mask=mask, labels=labels)
    assert abs(out[0].item() - float(z["loss"])) < 1e-4
    torch.testing.assert_close(out.logits.cpu(), torch.from_numpy(z["logits"]), atol=1e-3, rtol=1e-3)
    # encoder output: every position on the padded path; the valid tokens (all the encoder computes) when unpadded
    eng, want = model._engine, torch.from_numpy(z["enc_out"])
    if eng.ctx.rag is None:
        torch.testing.assert_close(eng.ctx.ws["e.out"].view(ids.shape[0], -1, dims.d_model).cpu(), want, atol=1e-3, rtol=1e-3)
    else:
        torch.testing.assert_close(eng.ctx.ws["e.out"].cpu(), want.reshape(-1, dims.d_model)[mask.reshape(-1).cpu()], atol=1e-3,
                                   rtol=1e-3)
    out[0].backward()
    g = group(z, "g/")
    for n, p in model.named_parameters():
        ref = g[plain_name(n)]
        torch.testing.assert_close(p.grad.cpu(), ref, atol=1e-3 * max(1.0, ref.abs().max().item()), rtol=1e-3,
                                   msg=lambda m, n=n: f"{n}: {m}")


@pytest.mark.parametrize("name", CASES)
def test_fp32_train_steps_vs_reference(name):
    z, dims, w, model = build(name, torch.float32)
    B, N, L = z["input_ids"].shape
    T = z["labels"].shape[1]
    lr, wd, clip, warm, total = z["train_hparams"].tolist()
    opt = types.SimpleNamespace(optim="adamw", lr=lr, weight_decay=wd, scheduler="linear", scheduler_steps=None,
                                total_steps=int(total), warmup_steps=int(warm), fixed_lr=False)
    optimizer, scheduler = U.set_optim(opt, model)
    model.train()
    for k in range(3):
        bi, bm, bl = dev(*O.synthetic_batch(B, N, L, T, dims.vocab_size, seed=SEEDS[name] + 200 + k))
        loss = model(input_ids=bi, attention_mask=bm, labels=bl)[0]
        loss.backward()
        gn = U.clip_grad_norm_(model, clip)
        optimizer.step()
        scheduler.step()
        model.zero_grad()
        assert abs(loss.item() - z["train_losses"][k]) < 1e-3
        assert abs(gn.item() - z["train_gnorms"][k]) < 2e-3 * max(1.0, z["train_gnorms"][k])
        if k == 0 or (k == 2 and name != "tiny_eos"):   # tiny_eos step 3 is ill-conditioned (see test_engine_cpu)
            ref = group(z, f"w_step{k + 1}/")
            for n, p in model.named_parameters():
                torch.testing.assert_close(p.detach().cpu(), ref[plain_name(n)], atol=1e-3, rtol=1e-3,
                                           msg=lambda m, n=n: f"{n}: {m}")


@pytest.mark.parametrize("name", CASES)
def test_fp32_generate_and_scores_vs_reference(name):
    z, dims, w, model = build(name, torch.float32)
    ids, mask = dev(z["input_ids"], z["attention_mask"])
    model.eval()
    for ml in (4, 12):
        toks = model.generate(input_ids=ids, attention_mask=mask, max_length=ml)
        assert toks.dtype == torch.int64 and toks.cpu().tolist() == z[f"gen_{ml}"].tolist()
    model.overwrite_forward_crossattention()
    model.reset_score_storage()
    model.generate(input_ids=ids, attention_mask=mask, max_length=4)
    ref = torch.from_numpy(z["cross_scores"])
    keep = mask.cpu().view(mask.shape[0], 1, 1, -1).expand_as(ref)
    torch.testing.assert_close(model._score_storage.cpu()[keep], ref[keep], atol=1e-3, rtol=1e-3)
    if ids.shape[1] == 2:
        for style in ("mean", "max", "21mean"):
            for half in ("no", "yes"):
                o = types.SimpleNamespace(stream=2, n_context=5, use_last_half_layer_attention=half,
                                          attention_score_style=style)
                mine = model.get_crossattention_scores(o, ids, None, mask)
                np.testing.assert_allclose(mine.numpy(), z[f"fact_scores_{style}_{half}"], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("name", CASES)
def test_bf16_vs_reference(name):
    z, dims, w, model = build(name, torch.bfloat16)
    ids, mask, labels = dev(z["input_ids"], z["attention_mask"], z["labels"])
    model.train()
    out = model(input_ids=ids, attention_mask=mask, labels=labels)
    ref_loss = float(z["loss"])
    assert abs(out[0].item() - ref_loss) < 0.03 * max(1.0, abs(ref_loss)), (out[0].item(), ref_loss)
    out[0].backward()
    g = group(z, "g/")
    for n, p in model.named_parameters():
        ref = g[plain_name(n)].flatten().double()
        got = p.grad.cpu().flatten().double()
        assert torch.isfinite(got).all(), n
        if ref.norm() > 1e-3 * np.sqrt(ref.numel()):      # direction only where the gradient is not ~0
            cos = float(torch.dot(ref, got) / (ref.norm() * got.norm() + 1e-30))
            assert cos > 0.98, f"{n}: cosine {cos:.4f}"
    if name == "tiny_eos":                                # trained margins are wide: tokens must match exactly
        model.eval()
        toks = model.generate(input_ids=ids, attention_mask=mask, max_length=12)
        assert toks.cpu().tolist() == z["gen_12"].tolist()


def test_fp32_config1_vs_oracle():
    """BASELINE config 1: T5-small, n_passages=5, text_maxlength=64, batch 2 (fp32), against the oracle."""
    dims = O.T5Dims.named("small")
    dims.dropout = 0.0
    w = O.init_weights(dims, seed=11)
    ids, mask, labels = O.synthetic_batch(2, 5, 64, 8, dims.vocab_size, seed=12)
    leaves = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    loss, logits = O.fid_forward(leaves, dims, ids, mask, labels, training=False)
    loss.backward()
    model = FiDT5(cfg_of(dims), dtype=torch.float32)
    model.load_t5(w)
    model = model.cuda().train()
    out = model(input_ids=ids.to(DEV), attention_mask=mask.to(DEV), labels=labels.to(DEV))
    assert abs(out[0].item() - loss.item()) < 1e-3
    torch.testing.assert_close(out.logits.cpu(), logits.detach(), atol=1e-3, rtol=1e-3)
    out[0].backward()
    for n, p in model.named_parameters():
        ref = leaves[plain_name(n)].grad
        torch.testing.assert_close(p.grad.cpu(), ref, atol=1e-3 * max(1.0, ref.abs().max().item()), rtol=1e-3,
                                   msg=lambda m, n=n: f"{n}: {m}")
    toks = model.eval().generate(input_ids=ids.to(DEV), attention_mask=mask.to(DEV), max_length=10)
    assert toks.cpu().tolist() == O.fid_generate(w, dims, ids, mask, 10).tolist()


def _base_model(dtype, dropout, seed=0):
    torch.manual_seed(0)
    cfg = FiDConfig.named("base", dropout_rate=dropout)
    m = FiDT5(cfg, dtype=dtype, seed=seed)
    with torch.no_grad():
        m._params_by_plain["shared.weight"].mul_(0.05)   # keep logits in a trainable range for random init
    return m.cuda()


def test_config2_shapes_properties():
    """T5-base, n_passages=20, L=200 (BASELINE config 2 shapes; batch 2 to keep the test short)."""
    B, N, L, T = 2, 20, 200, 8
    ids, mask, labels = dev(*O.synthetic_batch(B, N, L, T, 32128, seed=5))
    m32 = _base_model(torch.float32, 0.0).train()
    l32 = m32(input_ids=ids, attention_mask=mask, labels=labels)[0]
    l32.backward()
    g32 = m32._engine.G.clone()
    assert torch.isfinite(g32).all()
    # (1) passage-permutation invariance: the decoder sees an unordered set of passages
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(1)).to(DEV)
    lp = m32(input_ids=ids[:, perm], attention_mask=mask[:, perm], labels=labels)[0]
    assert abs(lp.item() - l32.item()) < 1e-3
    # (2) padding invariance: what sits under attention_mask == 0 cannot matter
    ids2 = torch.where(mask, ids, torch.full_like(ids, 7))
    lq = m32(input_ids=ids2, attention_mask=mask, labels=labels)[0]
    assert abs(lq.item() - l32.item()) < 1e-3
    # (3) bf16 path agrees with the fp32 path of the same kernels
    sd = {k: v.detach().clone() for k, v in m32.state_dict().items()}
    del m32
    torch.cuda.empty_cache()
    m16 = FiDT5(FiDConfig.named("base", dropout_rate=0.0), dtype=torch.bfloat16)
    m16.load_state_dict(sd)
    m16 = m16.cuda().train()
    l16 = m16(input_ids=ids, attention_mask=mask, labels=labels)[0]
    l16.backward()
    g16 = m16._engine.G
    assert abs(l16.item() - l32.item()) < 0.02 * abs(l32.item()), (l16.item(), l32.item())
    cos = float(torch.dot(g16.double(), g32.double()) / (g16.double().norm() * g32.double().norm()))
    assert cos > 0.98, cos
    assert abs(float(g16.norm() / g32.norm()) - 1.0) < 0.05


def test_config2_dropout_determinism():
    B, N, L, T = 2, 20, 200, 8
    ids, mask, labels = dev(*O.synthetic_batch(B, N, L, T, 32128, seed=6))
    res = []
    for _ in range(2):
        m = _base_model(torch.bfloat16, 0.1, seed=3).train()
        loss = m(input_ids=ids, attention_mask=mask, labels=labels)[0]
        loss.backward()
        res.append((loss.item(), m._engine.G.double().norm().item()))
        del m
        torch.cuda.empty_cache()
    # same seed → same dropout masks.  Activations are bit-deterministic; the loss scalar sums the rows with a float
    # atomic and the gradients accumulate with float atomics, so both may differ by summation-order rounding only
    assert abs(res[0][0] - res[1][0]) < 1e-6 * abs(res[0][0])
    assert abs(res[0][1] - res[1][1]) < 1e-3 * res[0][1]


def test_checkpoint_recompute_on_gpu():
    """set_checkpoint(True) on the HIP path, bf16 + dropout: identical loss, gradients equal up to atomic-order noise."""
    z, dims, w = load_case("mid_a")
    ids, mask, labels = dev(z["input_ids"], z["attention_mask"], z["labels"])
    res = []
    for ck in (False, True):
        m = FiDT5(cfg_of(dims, 0.1), dtype=torch.bfloat16, seed=5)
        m.load_t5(w)
        m = m.cuda().train()
        m.set_checkpoint(ck)
        loss = m(input_ids=ids, attention_mask=mask, labels=labels)[0]
        loss.backward()
        res.append((loss.item(), m._engine.G.clone()))
    assert abs(res[0][0] - res[1][0]) < 1e-6 * max(1.0, abs(res[0][0]))
    torch.testing.assert_close(res[0][1], res[1][1], atol=1e-4, rtol=1e-3)


def test_bf16_training_overfits_a_fixed_batch():
    """End-to-end sanity of the bf16 train step (fused AdamW, bf16 + transposed weight shadows, dropout on):
    T5-small on one fixed synthetic batch must drive the loss far below its starting value."""
    cfg = FiDConfig.named("small", dropout_rate=0.1)
    torch.manual_seed(0)
    m = FiDT5(cfg, dtype=torch.bfloat16, seed=1)
    with torch.no_grad():
        m._params_by_plain["shared.weight"].mul_(0.05)
    m = m.cuda().train()
    ids, mask, labels = dev(*O.synthetic_batch(4, 3, 64, 6, cfg.vocab_size, seed=77))
    opt = types.SimpleNamespace(optim="adamw", lr=1e-3, weight_decay=0.0, scheduler="fixed", fixed_lr=True,
                                scheduler_steps=None, total_steps=100, warmup_steps=0)
    optimizer, scheduler = U.set_optim(opt, m)
    first = None
    for k in range(80):
        loss = m(input_ids=ids, attention_mask=mask, labels=labels)[0]
        loss.backward()
        U.clip_grad_norm_(m, 1.0)
        optimizer.step()
        scheduler.step()
        m.zero_grad()
        first = first if first is not None else loss.item()
    last = loss.item()
    assert first > 8.0 and last < 0.25 * first, (first, last)     # ln(32128) = 10.4 at random init


@pytest.mark.gpu
def test_train_reader_driver_end_to_end(tmp_path):
    """train_reader.py (the reference driver's flags) on synthetic batches with T5-small: 2 epochs × 3 steps, greedy
    evaluation, checkpoint written in the reference's directory format, then a second run resumes from it."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(root, "train_reader.py"), "--model_size", "small", "--per_gpu_batch_size", "2",
            "--n_context", "3", "--text_maxlength", "32", "--optim", "adamw", "--scheduler", "linear", "--weight_decay", "1e-4",
            "--lr", "1e-3", "--epochs", "2", "--synthetic", "2,3,32,4", "--steps", "3", "--checkpoint_dir", str(tmp_path),
            "--name", "run"]
    r = subprocess.run(base, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "epoch 1" in r.stderr and "samples/s" in r.stderr
    ck = tmp_path / "run" / "checkpoint"
    if (ck / "latest").exists():       # written only when the dev score improved over 0
        assert (ck / "best_dev" / "optimizer.pth.tar").exists() and (ck / "best_dev" / "model.safetensors").exists()
    # save explicitly through the library and resume the driver from that directory
    from lako_amd import FiDConfig, FiDT5, util as U
    model = FiDT5(FiDConfig.named("small", dropout_rate=0.1), dtype=torch.bfloat16, seed=0).cuda()
    opt = types.SimpleNamespace(optim="adamw", lr=1e-3, weight_decay=1e-4, scheduler="linear", scheduler_steps=None,
                                total_steps=6, warmup_steps=0, fixed_lr=False)
    optimizer, scheduler = U.set_optim(opt, model)
    U.save(model, optimizer, scheduler, 5, 0.5, opt, str(tmp_path / "manual"), "step-5")
    r2 = subprocess.run(base + ["--model_path", str(tmp_path / "manual" / "checkpoint" / "latest")], capture_output=True,
                        text=True, timeout=600)
    assert r2.returncode == 0, r2.stderr[-2000:]
    assert "model loaded from" in r2.stderr and "step 5" in r2.stderr


@pytest.mark.gpu
def test_deterministic_mode_repeats_bit_for_bit():
    """LAKO_DETERMINISTIC=1: two processes running the same three seeded bf16 training steps (dropout on, ragged
    passages, AdamW + clipping) end with bit-identical losses and weights — every float sum shared between workgroups (norm-weight and
    position-bias gradients, gradient norm, embedding scatter, the K-splits of the weight-gradient products, the LM-head input
    gradient) is order-independent in that mode."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LAKO_DETERMINISTIC="1")
    outs = []
    for _ in range(2):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "det_run.py")], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert outs[0]["deterministic"] is True
    assert outs[0] == outs[1], outs
    first = float.fromhex(outs[0]["losses"][0])
    assert 8.0 < first < 13.0                              # a real step: ln(32128) = 10.4 at random init


@pytest.mark.gpu
def test_deterministic_mode_kernels_match_the_reference():
    """The kernels that change their accumulation under LAKO_DETERMINISTIC=1 (attention bias gradient, norm-weight gradient, gradient
    norm, embedding scatter, weight-gradient K-splits) against the same fp32 doubles as in the default mode: the kernel tests of those
    ops once more in a child pytest with the variable set (the library reads it once per process)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_kernels_gpu.py"), "-q", "-x", "-m", "gpu", "-p",
                        "no:cacheprovider", "-k", "attention and not race or rmsnorm or embed or relpos or optimizer or gemm_tn"],
                       env=dict(os.environ, LAKO_DETERMINISTIC="1"), capture_output=True, text=True, timeout=1200, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1500:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-500:]


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_unpadded_path_equals_padded_path_gpu(dtype, monkeypatch):
    """HIP path, T5-small shapes with ragged passages (one of them empty): unpadded (default) vs LAKO_UNPAD=0 — same loss,
    logits and gradients (dropout off: the element-wise dropout draws are indexed by packed rows and differ by design)."""
    cfg = FiDConfig.named("small", dropout_rate=0.0)
    model = FiDT5(cfg, dtype=dtype, seed=4).cuda()
    with torch.no_grad():
        model._params_by_plain["shared.weight"].mul_(0.05)
    B, N, L, T = 3, 4, 72, 6
    ids, mask, labels = dev(*O.synthetic_batch(B, N, L, T, cfg.vocab_size, seed=77))
    mask[1, 2] = False                                   # a fully padded passage
    ids = ids.masked_fill(~mask, 0)
    model.train()
    res = []
    for flag in ("0", "1"):
        monkeypatch.setenv("LAKO_UNPAD", flag)
        model.zero_grad()
        out = model(input_ids=ids, attention_mask=mask, labels=labels)
        assert (model._engine.ctx.rag is not None) == (flag == "1")
        out[0].backward()
        res.append((out[0].item(), out.logits.float().clone(), model._engine.G.clone()))
    if dtype == torch.float32:          # greedy decoding: same tokens on both paths (fp32: no rounding-order ties)
        model.eval()
        toks = []
        for flag in ("0", "1"):
            monkeypatch.setenv("LAKO_UNPAD", flag)
            toks.append(model.generate(input_ids=ids, attention_mask=mask, max_length=12))
        assert torch.equal(toks[0], toks[1])
    tol = dict(atol=2e-5, rtol=1e-4) if dtype == torch.float32 else dict(atol=2e-2, rtol=2e-2)
    assert abs(res[0][0] - res[1][0]) < (1e-5 if dtype == torch.float32 else 2e-3)
    torch.testing.assert_close(res[1][1], res[0][1], **tol)
    gscale = res[0][2].abs().max().item()
    # bf16: the packed keys of the cross-attention are chunked differently (other rounding of the running softmax)
    assert (res[1][2] - res[0][2]).abs().max().item() < (1e-4 if dtype == torch.float32 else 5e-2) * gscale


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,L,T", [(1, 1, 5, 1), (2, 3, 17, 3), (1, 7, 33, 2), (3, 1, 250, 9)])
def test_fp32_edge_shapes_vs_oracle(B, N, L, T):
    """Shapes at the edges of the kernels' tilings — a single passage, sequences shorter than one 16-row block, one
    decoder position, one sample, more than one 256-key chunk per passage — against the oracle (fp32): loss, logits, every
    gradient and the greedy tokens.  Valid lengths are ragged, so the unpadded path is the one exercised."""
    dims = O.T5Dims(vocab_size=128, d_model=64, d_kv=32, d_ff=128, num_layers=2, num_decoder_layers=2, num_heads=2)
    dims.dropout = 0.0
    w = O.init_weights(dims, seed=21)
    ids, mask, labels = O.synthetic_batch(B, N, L, T, dims.vocab_size, seed=31 + L)
    leaves = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    loss, logits = O.fid_forward(leaves, dims, ids, mask, labels, training=False)
    loss.backward()
    model = FiDT5(cfg_of(dims), dtype=torch.float32)
    model.load_t5(w)
    model = model.cuda().train()
    out = model(input_ids=ids.to(DEV), attention_mask=mask.to(DEV), labels=labels.to(DEV))
    assert (model._engine.ctx.rag is not None) == bool((~mask).any())
    assert abs(out[0].item() - loss.item()) < 1e-4
    torch.testing.assert_close(out.logits.cpu(), logits.detach(), atol=1e-4, rtol=1e-4)
    out[0].backward()
    for n, p in model.named_parameters():
        ref = leaves[plain_name(n)].grad
        torch.testing.assert_close(p.grad.cpu(), ref, atol=1e-4 * max(1.0, ref.abs().max().item()), rtol=1e-3,
                                   msg=lambda m, n=n: f"{n}: {m}")
    toks = model.eval().generate(input_ids=ids.to(DEV), attention_mask=mask.to(DEV), max_length=6)
    assert toks.cpu().tolist() == O.fid_generate(w, dims, ids, mask, 6).tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_generate_graph_replay_equals_eager(dtype, monkeypatch):
    """Greedy decode through captured HIP graphs (engine.generate: one graph per step position, replayed from the second call
    of a shape on) against the eager launch sequence of the same kernels: same tokens for batches that differ in content, valid
    lengths (unpadded offsets copied into the workspace) and in where the rows finish.  In bf16 the decode steps run the
    cross-attention in the encoder-state space (bf16 roundings differ from the projected formulation: near-tie argmaxes of a
    random-init model may flip, so that pair is compared on the fp32 CPU double, tests/test_engine_cpu.py)."""
    cfg = FiDConfig.named("small", dropout_rate=0.0)
    model = FiDT5(cfg, dtype=dtype, seed=11).cuda().eval()
    with torch.no_grad():
        model._params_by_plain["shared.weight"].mul_(0.05)
    B, N, L, ML = 3, 4, 72, 9
    batches = [dev(*O.synthetic_batch(B, N, L, 4, cfg.vocab_size, seed=s))[:2] for s in (1, 2, 3)]
    full = torch.ones_like(batches[0][1])
    batches.append((batches[0][0], full))                  # no padding at all: the padded layout, its own graphs
    monkeypatch.setenv("LAKO_GEN_GRAPH", "0")
    monkeypatch.setenv("LAKO_XATTN_GENERATE", "1")         # (the fp32 engine keeps the projected formulation regardless)
    eager = [model.generate(input_ids=i, attention_mask=m, max_length=ML).clone() for i, m in batches]
    monkeypatch.delenv("LAKO_XATTN_GENERATE")
    monkeypatch.setenv("LAKO_GEN_GRAPH", "1")
    for rnd in range(3):                                    # round 0: eager warm-up of each mode, 1: capture, 2: pure replay
        for (i, m), want in zip(batches, eager):
            got = model.generate(input_ids=i, attention_mask=m, max_length=ML)
            assert torch.equal(got, want), (rnd, got.tolist(), want.tolist())
    ws = model._engine._workspace(("gen", B, N, L, ML))
    assert sorted(ws["g.graphs"]) == (["p", "r"] if dtype == torch.float32 else ["pxf", "rxf"])
    assert all(len(g["steps"]) >= 1 for g in ws["g.graphs"].values())
    assert model._engine.xattn_active == (dtype == torch.bfloat16)


@pytest.mark.gpu
def test_bf16_generate_encoder_space_decode_vs_oracle_tokens():
    """Greedy decode in bf16 on the DEFAULT decode path — HIP graphs, cross-attention in the encoder-state space, the one-pass
    `xdecode_kernel` + combine (src/model.py:54-60, train_reader.py:142-146) — against the ORACLE's tokens on the same weights.
    The fixture (tests/golden/gen_d512.npz, made by oracle/make_generate_fixture.py on a GPU box; round 4 trained it inside this test with up
    to four restarts) is a reader at the smallest dimensions the decode kernel takes — d_model 512, 8 heads, 1 + 2 layers — trained on one
    batch of 4 samples x 10 passages x 200 tokens (up to 2 000 keys per sample: every sample spans all key ranges of the decode kernel) until
    the answers are reproduced with a top-1 / top-2 logit gap of 6.4, far above bf16 noise; its weights are bf16-representable, so the HIP
    path and the oracle compute on the same numbers.  The oracle's fp32 decode of those weights is pinned on the CPU
    (tests/test_oracle_golden.py::test_generate_fixture_is_the_oracles_decode); here the bf16 HIP path must emit the same tokens, ragged EOS rows
    included."""
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gen_d512.npz"))
    names = ("vocab_size", "d_model", "d_kv", "d_ff", "num_layers", "num_decoder_layers", "num_heads")
    cfg = FiDConfig(dropout_rate=0.0, **{k: int(v) for k, v in zip(names, z["dims"])})
    w = {k[2:]: torch.from_numpy(z[k]).view(torch.bfloat16).float() for k in z.files if k.startswith("w.")}
    ids_c, mask_c = torch.from_numpy(z["ids"]), torch.from_numpy(z["mask"])
    want, ML = torch.from_numpy(z["tokens"]), int(z["max_length"])
    B, N, L = ids_c.shape
    m = FiDT5(cfg, dtype=torch.bfloat16)
    m.load_t5(w)
    m = m.cuda().eval()
    ids, mask = dev(ids_c, mask_c)
    for rnd in range(3):                                    # eager warm-up of the mode, graph capture, pure replay
        got = m.generate(input_ids=ids, attention_mask=mask, max_length=ML)
        assert got.cpu().tolist() == want.tolist(), (rnd, got.cpu().tolist(), want.tolist())
    assert m._engine.xattn_active
    ws = m._engine._workspace(("gen", B, N, L, ML))
    assert "rxf" in ws["g.graphs"] and len(ws["g.graphs"]["rxf"]["steps"]) >= 1      # ragged + encoder-space + one-pass decode


@pytest.mark.gpu
def test_encoder_space_cross_attention_long_answers_and_small_model(monkeypatch):
    """T5-small (d_model 512, 8 heads) with 20 answer positions: R = T·H = 160 query rows per sample — two row chunks of the
    scores / context kernels — and samples of very different key counts (one passage of a sample empty, one sample short).
    Both cross-attention formulations on the HIP path, bf16, dropout on (the same masks): agreement to bf16 noise."""
    cfg = FiDConfig.named("small", dropout_rate=0.1)
    model = FiDT5(cfg, dtype=torch.bfloat16, seed=21).cuda().train()
    with torch.no_grad():
        model._params_by_plain["shared.weight"].mul_(0.05)
    B, N, L, T = 3, 5, 64, 20
    ids, mask, labels = dev(*O.synthetic_batch(B, N, L, T, cfg.vocab_size, seed=31))
    mask[1, 2] = False
    mask[2, 1:, 8:] = False
    ids = ids.masked_fill(~mask, 0)
    res = []
    for x in ("0", "1"):
        monkeypatch.setenv("LAKO_XATTN", x)
        model.zero_grad()
        model._get_engine().step_count = 0
        out = model(input_ids=ids, attention_mask=mask, labels=labels)
        assert model._engine.xattn_active == (x == "1")
        out[0].backward()
        res.append((out[0].item(), out.logits.float().clone(), model._engine.G.clone()))
    assert abs(res[0][0] - res[1][0]) < 2e-3 * abs(res[0][0]), (res[0][0], res[1][0])
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    assert rel(res[1][1], res[0][1]) < 0.02, rel(res[1][1], res[0][1])
    assert rel(res[1][2], res[0][2]) < 0.06, rel(res[1][2], res[0][2])


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,short_first", [(torch.float32, False), (torch.float32, True), (torch.bfloat16, False)])
def test_legacy_cross_bias_vs_oracle(dtype, short_first):
    """`legacy_cross_bias` — transformers 3.0.2's relative-position bias in every cross-attention (src/model.py:301-303,323-329; the oracle's
    version is pinned to the reference's own cross_attention_forward by tests/golden/legacy_cross.npz) — through the HIP kernels: T5-small
    dimensions, 6 passages of 48, answers of 9 positions, dropout off.  fp32: loss / logits / every gradient (the table's included)
    within 1e-3 of the oracle, greedy tokens and the captured step-0 scores equal; bf16: loss within 3 %, gradient direction of the table.
    `short_first`: one sample's first passage is shorter than the answer → the batch runs on the padded layout (Engine._legacy_layout)."""
    import dataclasses
    dims = dataclasses.replace(O.T5Dims.named("small"), legacy_cross_bias=True, dropout=0.0, num_layers=2, num_decoder_layers=3,
                               vocab_size=512)
    w = O.init_weights(dims, seed=21)
    w[O.XREL] = w[O.XREL] * 4.0
    B, N, L, T = 3, 6, 48, 9
    ids, mask, labels = O.synthetic_batch(B, N, L, T, dims.vocab_size, seed=22)[:3]
    mask = mask.clone()
    mask[:, 0, :12] = True
    if short_first:
        mask[2, 0, 5:] = False
    ids = ids * mask
    leaves = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    loss, logits = O.fid_forward(leaves, dims, ids, mask, labels, training=False)
    loss.backward()
    cfg = dataclasses.replace(cfg_of(dims), legacy_cross_bias=True)
    model = FiDT5(cfg, dtype=dtype)
    model.load_t5(w)
    model = model.cuda().train()
    out = model(input_ids=ids.to(DEV), attention_mask=mask.to(DEV), labels=labels.to(DEV))
    assert (model._engine.ctx.rag is None) == short_first
    assert not model._engine.xattn_active           # the bias is an argument of the projected kernels
    out[0].backward()
    grads = {plain_name(n): p.grad.cpu() for n, p in model.named_parameters()}
    gx, rx = grads[O.XREL].flatten().double(), leaves[O.XREL].grad.flatten().double()
    assert rx.abs().max() > 1e-4
    if dtype == torch.float32:
        assert abs(out[0].item() - loss.item()) < 1e-3
        torch.testing.assert_close(out.logits.cpu(), logits.detach(), atol=1e-3, rtol=1e-3)
        for k, g in grads.items():
            ref = leaves[k].grad
            torch.testing.assert_close(g, ref, atol=1e-3 * max(1.0, ref.abs().max().item()), rtol=1e-3, msg=lambda m, k=k: f"{k}: {m}")
        # without the table the same weights give another loss: the bias is really applied
        plain = FiDT5(cfg_of(dims), dtype=dtype)
        plain.load_t5(w)
        plain = plain.cuda().train()
        assert abs(plain(input_ids=ids.to(DEV), attention_mask=mask.to(DEV), labels=labels.to(DEV))[0].item() - loss.item()) > 1e-3
        model.eval()
        toks = model.generate(input_ids=ids.to(DEV), attention_mask=mask.to(DEV), max_length=8)
        assert toks.cpu().tolist() == O.fid_generate(w, dims, ids, mask, 8).tolist()
        cap = {}
        O.fid_generate(w, dims, ids, mask, 4, capture=cap)
        model.overwrite_forward_crossattention()
        model.reset_score_storage()
        model.generate(input_ids=ids.to(DEV), attention_mask=mask.to(DEV), max_length=4)
        ref = torch.cat([cap[f"cross_scores_{i}"] for i in range(dims.num_decoder_layers)], dim=2)
        keep = mask.view(B, 1, 1, -1).expand_as(ref)
        torch.testing.assert_close(model._score_storage.cpu()[keep], ref[keep], atol=1e-3, rtol=1e-3)
    else:
        assert abs(out[0].item() - loss.item()) < 0.03 * max(1.0, abs(loss.item()))
        cos = float(torch.dot(gx, rx) / (gx.norm() * rx.norm() + 1e-30))
        assert cos > 0.98, cos


@pytest.mark.gpu
def test_all_valid_batch_takes_the_fast_encoder_kernels_and_matches_the_masked_ones(monkeypatch):
    """A batch without any padding (every passage at text_maxlength; `bench.py`'s all-valid leg) runs the encoder's self-attention
    WITHOUT a key mask, i.e. on the fast encoder kernels (one-pass backward included) instead of the generic masked ones
    (Engine._enc_mask).  Same inputs with the mask kept (LAKO_ENC_ALLVALID_MASK=1 → generic kernels): bf16, dropout off — loss within
    1e-3 relative, every gradient's direction cosine > 0.995 (two kernel families: different summation orders, both bf16)."""
    cfg = FiDConfig.named("small", dropout_rate=0.0, vocab_size=512, num_layers=3, num_decoder_layers=2)
    torch.manual_seed(5)
    B, N, L, T = 4, 8, 96, 6
    ids = torch.randint(2, cfg.vocab_size, (B, N, L), device=DEV)
    mask = torch.ones(B, N, L, dtype=torch.bool, device=DEV)
    labels = torch.randint(2, cfg.vocab_size, (B, T), device=DEV)
    res = {}
    for keep in ("0", "1"):
        monkeypatch.setenv("LAKO_ENC_ALLVALID_MASK", keep)
        torch.manual_seed(6)            # the same initial weights for both runs
        m = FiDT5(cfg, dtype=torch.bfloat16, seed=3)
        with torch.no_grad():
            m._params_by_plain["shared.weight"].mul_(0.2)
        m = m.cuda().train()
        out = m(input_ids=ids, attention_mask=mask, labels=labels)
        assert (m._engine.ctx.enc_mask is None) == (keep == "0")
        out[0].backward()
        res[keep] = (out[0].item(), {n: p.grad.float().cpu().flatten() for n, p in m.named_parameters()})
    l0, l1 = res["0"][0], res["1"][0]
    assert abs(l0 - l1) < 1e-3 * max(1.0, abs(l1)), (l0, l1)
    for n, g0 in res["0"][1].items():
        g1 = res["1"][1][n]
        if g1.norm() > 1e-6 * g1.numel() ** 0.5:
            cos = float(torch.dot(g0.double(), g1.double()) / (g0.double().norm() * g1.double().norm() + 1e-30))
            assert cos > 0.995, f"{n}: {cos:.5f}"
