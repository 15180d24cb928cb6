"""Driver-level GPU tests: (1) the RCCL gradient path on device buffers in a fresh child process (world size 1 — the boxes
have one GPU — both data-parallel modes), (2) the reader's REAL input pipeline on the GPU: JSON examples → Dataset → Collator
(tokenised, padded to text_maxlength, −100 labels) → HIP FiDT5 train step / greedy decode / score dump through
test_reader.evaluate, checked against the oracle on the same collated tensors."""
import json
import os
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

from lako_amd import FiDConfig, FiDT5
from lako_amd import util as U
from lako_amd.data import Collator, Dataset
from lako_amd.model import plain_name
from oracle import fid_t5_oracle as O
from tests.stub_tokenizer import StubTokenizer

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda"


@pytest.mark.parametrize("mode", ["deferred", "overlap"])
def test_rccl_gradient_path_in_child_process(mode):
    """bench.py with LAKO_FORCE_DIST=1: init_process_group('nccl') = RCCL, parameter broadcast, the gradient all-reduce on the
    flat device buffer (or per finished gradient range, LAKO_DP_MODE=overlap), the 1/world factor inside the fused AdamW.
    A fresh process: RCCL initialisation must not meet an already-initialised communicator of this test process."""
    env = dict(os.environ, LAKO_FORCE_DIST="1", LAKO_DP_MODE=mode, MASTER_ADDR="127.0.0.1", MASTER_PORT="29671",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--model", "small", "--batch", "2", "--n-passages", "4",
                        "--seq-len", "64", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--all-valid-steps", "0"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.lstrip().startswith("{")][-1])
    assert out["config"]["rccl_world_size"] == 1 and out["config"]["dp_mode"] == mode
    assert out["value"] > 0 and np.isfinite(out["config"]["final_mean_loss"])
    # same seeds without the collective: the all-reduce over one rank is the identity, so the loss trajectory is the same
    env2 = {k: v for k, v in env.items() if k not in ("LAKO_FORCE_DIST", "LAKO_DP_MODE")}
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--model", "small", "--batch", "2", "--n-passages", "4",
                         "--seq-len", "64", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--all-valid-steps", "0"],
                        capture_output=True, text=True, timeout=900, env=env2)
    assert r2.returncode == 0, r2.stderr[-3000:]
    out2 = json.loads([ln for ln in r2.stdout.splitlines() if ln.lstrip().startswith("{")][-1])
    assert out2["config"]["rccl_world_size"] is None
    assert abs(out["config"]["final_mean_loss"] - out2["config"]["final_mean_loss"]) < 2e-3


_COMM_CHILD = r"""
import ctypes as C, sys, torch
sys.path.insert(0, {root!r})
from lako_amd import _lib
lib = _lib.load()
torch.cuda.set_device(0)
idb = (C.c_uint8 * 128)()
assert lib.lako_comm_unique_id(idb) == 0
h = C.c_void_p()
assert lib.lako_comm_init(C.byref(h), 0, 1, idb) == 0 and h.value
assert lib.lako_comm_world_size(h) == 1
s = torch.cuda.Stream()
for dt, code in ((torch.float32, 0), (torch.bfloat16, 1)):
    g = torch.randn(1 << 20, device="cuda").to(dt)
    want = g.clone()
    with torch.cuda.stream(s):
        s.wait_stream(torch.cuda.current_stream())
        assert lib.lako_allreduce(h, C.c_void_p(g.data_ptr()), g.numel(), code, C.c_void_p(s.cuda_stream)) == 0
    s.synchronize()
    assert torch.equal(g, want), dt                 # SUM over one rank: the identity, in place
assert lib.lako_allreduce(h, C.c_void_p(g.data_ptr()), g.numel(), 2, None) == -1       # fp8 gradients are not a transport
assert lib.lako_comm_destroy(h) == 0
print("comm ok")
"""


def test_comm_c_abi_world_of_one_in_child_process():
    """(round 5) lako_comm_unique_id → lako_comm_init → lako_allreduce (fp32 and bf16, in place, on the caller's stream)
    → lako_comm_destroy through the C-ABI alone, RCCL resolved at run time: over one rank the SUM is the identity.  A fresh process, as for
    the torch.distributed path above."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", _COMM_CHILD.format(root=ROOT)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "comm ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_wrong_passage_lengths_raise_without_a_host_sync_on_the_training_path():
    """(round 6) The mask-vs-lengths check is on by default and asynchronous: a batch whose lengths do not describe its mask raises
    ValueError — at the latest when the outstanding verdicts are collected (`check_lengths_now`), typically one or two batches later —
    and consistent batches under torch's sync debug mode stay pure enqueueing (the test below runs with the check on)."""
    import bench
    cfg = FiDConfig.named("small", dropout_rate=0.0)
    model = FiDT5(cfg, dtype=torch.bfloat16).cuda().train()
    ids, mask, labels, lens = bench.synthetic_batch(2, 4, 64, 6, cfg.vocab_size, seed=77, device="cuda", with_lengths=True)
    model(input_ids=ids, attention_mask=mask, labels=labels, passage_lengths=lens)
    model._engine.check_lengths_now()
    wrong = lens.clone()
    wrong[0, 0] = max(int(wrong[0, 0]) - 1, 0) if int(wrong[0, 0]) > 1 else int(wrong[0, 0]) + 1
    with pytest.raises(ValueError, match="passage_lengths"):
        model(input_ids=ids, attention_mask=mask, labels=labels, passage_lengths=wrong)
        for _ in range(3):                       # the verdict arrives behind an event: a later batch (or the explicit wait) raises
            torch.cuda.synchronize()
            model(input_ids=ids, attention_mask=mask, labels=labels, passage_lengths=lens)
        model._engine.check_lengths_now()
    model._engine.check_lengths_now()            # consumed


def test_training_step_issues_no_host_sync_when_the_collator_hands_over_lengths():
    """(round 5; SURVEY.md §5.8 names host-side launch jitter as the data-parallel risk) With the collator's host-side passage lengths the
    whole step — forward, backward, clip, optimizer — must be pure enqueueing: no device→host read-back, no synchronise.  torch's sync
    debug mode turns any synchronising torch call into an error; the library itself never synchronises (include/lako_hip.h).  The
    reference's bare call (no lengths: one read-back of the mask per new batch) is the documented exception and is not exercised here."""
    import bench
    cfg = FiDConfig.named("small", dropout_rate=0.1)
    model = FiDT5(cfg, dtype=torch.bfloat16).cuda().train()
    opt = types.SimpleNamespace(lr=1e-4, optim="adamw", scheduler="linear", weight_decay=0.01, warmup_steps=2, total_steps=10,
                                scheduler_steps=None, fixed_lr=False)
    optimizer, scheduler = U.set_optim(opt, model)
    batches = [bench.synthetic_batch(2, 4, 64, 6, cfg.vocab_size, seed=90 + i, device="cuda", with_lengths=True) for i in range(3)]

    def step(i):
        ids, mask, labels, lens = batches[i]
        loss = model(input_ids=ids, attention_mask=mask, labels=labels, passage_lengths=lens)[0]
        loss.backward()
        U.clip_grad_norm_(model, 1.0)
        optimizer.step()
        scheduler.step()
        model.zero_grad()
        return loss

    step(0)                                      # first call: allocations, workspace sizing, LDS attributes
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        losses = [step(1), step(2)]
    finally:
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(x)) for x in losses)


def _examples(n=6):
    rng = np.random.RandomState(3)
    words = "red blue dog cat rain tennis racket sofa street sign man woman ball tree car bus".split()
    ex = []
    for i in range(n):
        facts = [{"sentence": " ".join(rng.choice(words, rng.randint(3, 7))) + " .", "id": int(rng.randint(100))}
                 for _ in range(rng.randint(1, 5))]
        tgt = str(rng.choice(words)) + (" " + str(rng.choice(words)) if i % 2 else "")      # 1- and 2-word answers: ragged targets
        ex.append({"question": " ".join(rng.choice(words, 5)), "target": tgt, "answer": {tgt: 1.0, str(rng.choice(words)): 0.3},
                   "img_id": i, "caption": " ".join(rng.choice(words, 6)), "fact": facts})
    return ex


def _dims():
    return O.T5Dims(vocab_size=64, d_model=64, d_kv=32, d_ff=128, num_layers=2, num_decoder_layers=2, num_heads=2)


def test_collated_batches_through_the_hip_reader_vs_oracle(tmp_path):
    """f2 on the GPU: the tensors the reader's own Collator produces (stub tokenizer, stream 2: [question+caption, facts],
    pad-to-text_maxlength, −100 labels) go through the HIP FiDT5 — one train step, greedy decode, per-fact score dump —
    and every result is compared with the oracle run on the same collated tensors."""
    opt = types.SimpleNamespace(n_context=3, fact_use_way="concate", use_fact="yes", stream=2, attention_score_style="21mean",
                                use_last_half_layer_attention="yes", write_crossattention_scores=True, write_results=True,
                                ans_attention="no", dataset="okvqa", model_size="tiny", per_gpu_batch_size=3, text_maxlength=40,
                                is_distributed=False, world_size=1, global_rank=0, is_main=True, device=torch.device(DEV))
    examples = _examples()
    ds = Dataset(examples, opt)
    tok = StubTokenizer()
    col = Collator(opt.text_maxlength, tok, answer_maxlength=-1, stream=2)
    dims = _dims()
    dims.dropout = 0.0
    w = O.init_weights(dims, seed=5)
    model = FiDT5(FiDConfig(vocab_size=dims.vocab_size, d_model=dims.d_model, d_kv=dims.d_kv, d_ff=dims.d_ff,
                            num_layers=dims.num_layers, num_decoder_layers=dims.num_decoder_layers, num_heads=dims.num_heads,
                            dropout_rate=0.0), dtype=torch.float32)
    model.load_t5(w)
    model = model.cuda().train()
    idx, labels, _, ids, mask = col([ds[i] for i in range(3)])
    assert ids.shape == (3, 2, 40) and (~mask).any() and (labels == -100).any()        # padded passages and targets
    # ---- train step on collated tensors (host-side lengths handed over like train_reader.py does) --------------------
    lens = mask.sum(-1).to(torch.int32)
    leaves = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    ref_loss, ref_logits = O.fid_forward(leaves, dims, ids, mask, labels, training=False)
    ref_loss.backward()
    out = model(input_ids=ids.to(DEV), attention_mask=mask.to(DEV), labels=labels.to(DEV), passage_lengths=lens)
    assert model._engine.ctx.rag is not None
    assert abs(out[0].item() - ref_loss.item()) < 1e-4
    torch.testing.assert_close(out.logits.cpu(), ref_logits.detach(), atol=1e-4, rtol=1e-4)
    out[0].backward()
    for n, p in model.named_parameters():
        r = leaves[plain_name(n)].grad
        torch.testing.assert_close(p.grad.cpu(), r, atol=1e-4 * max(1.0, r.abs().max().item()), rtol=1e-3,
                                   msg=lambda m, n=n: f"{n}: {m}")
    model.zero_grad()
    # ---- the evaluation driver: greedy decode + EM + per-fact scores written back into the examples --------------------
    import test_reader as TR
    from torch.utils.data import DataLoader, SequentialSampler
    loader = DataLoader(ds, sampler=SequentialSampler(ds), batch_size=3, collate_fn=col)
    em, stem_em, inc_em, total = TR.evaluate(model, ds, loader, tok, opt, str(tmp_path))
    assert total == len(examples) and 0.0 <= em <= inc_em <= 1.0
    files = os.listdir(tmp_path / "test_results")
    assert len(files) == 1 and len(json.load(open(tmp_path / "test_results" / files[0]))) == len(examples)
    # oracle: tokens and fact scores on the same collated tensors
    for batch in loader:
        bidx, _, _, bids, bmask = batch
        cap = {}
        ref_tok = O.fid_generate(w, dims, bids, bmask, 50, capture=cap)
        got_tok = model.generate(input_ids=bids.to(DEV), attention_mask=bmask.to(DEV), max_length=50)
        assert got_tok.cpu().tolist() == ref_tok.tolist()
        raw = torch.cat([cap[f"cross_scores_{i}"] for i in range(dims.num_decoder_layers)], dim=2)      # [B, H, layers, N·L]
        ref_sc = O.crossattention_fact_scores(raw, bids, bmask, opt.n_context, style="21mean", last_half=True)
        ref_p = torch.softmax(ref_sc, dim=-1)
        for k in range(len(bidx)):
            facts = ds.data[int(bidx[k])]["fact"]
            for j in range(min(opt.n_context, len(facts))):
                assert abs(facts[j]["score"] - ref_p[k, j].item()) < 1e-4, (int(bidx[k]), j)


@pytest.mark.gpu
def test_train_retriever_driver_end_to_end(tmp_path):
    """train_retriever.py (the reference driver's flags: src/options.py base + retriever + optim) on synthetic batches: BERT-base
    width, 2 layers, 6 optimizer steps with dropout on, evaluation (KL loss, inversions, top-k overlap), checkpoint in the
    reference's directory format; the loss must fall on the (fixed) training batches, and a second run resumes from the checkpoint."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(root, "train_retriever.py"), "--synthetic", "4,6,24,40", "--retriever_layers", "2", "--optim", "adamw",
            "--scheduler", "linear", "--lr", "1e-3", "--weight_decay", "0.01", "--dropout", "0.1", "--epochs", "3", "--checkpoint_dir",
            str(tmp_path), "--name", "ret", "--per_gpu_batch_size", "4"]
    r = subprocess.run(base + ["--steps", "40"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    evals = [float(ln.split("eval: ")[1].split(",")[0]) for ln in r.stderr.splitlines() if "eval: " in ln]
    trains = [float(ln.split("train: ")[1].split(",")[0]) for ln in r.stderr.splitlines() if "train: " in ln]
    # the training loss falls over the epochs (with dropout on, one epoch of 16 noisy steps may sit above its predecessor — which one depends
    # on the dropout masks, i.e. on the recipe of csrc/common.h — the last one must be well below the first)
    assert len(evals) >= 2 and all(np.isfinite(evals)) and len(trains) >= 3 and trains[-1] < 0.8 * trains[0], (trains, evals)
    assert "avg top1" in r.stderr and "questions/s" in r.stderr
    ck = tmp_path / "ret" / "checkpoint" / "best_dev"
    assert (ck / "optimizer.pth.tar").exists() and (ck / "model.safetensors").exists()
    r2 = subprocess.run(base + ["--steps", "4", "--model_path", str(tmp_path / "ret" / "checkpoint" / "latest")], capture_output=True,
                        text=True, timeout=900)
    assert r2.returncode == 0, r2.stderr[-3000:]
    assert "model loaded from" in r2.stderr
