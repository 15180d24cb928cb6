#!/usr/bin/env python
"""Test infrastructure (runs on a GPU box: `python oracle/make_generate_fixture.py` → gpurun_out/gen_d512.npz, then copied to
tests/golden/gen_d512.npz).  Builds the fixture of tests/test_parity_gpu.py::test_bf16_generate_encoder_space_decode_vs_oracle_tokens:

  a reader at the SMALLEST dimensions the one-pass decode kernel takes (d_model 512, 8 heads of 64) — one encoder layer, two decoder
  layers, d_ff 64, vocabulary 256 — is trained on one batch (4 samples × 10 passages × 200 tokens, the reference's loop:
  train_reader.py:62-84) on the HIP path until the answers are reproduced with wide margins; the weights are ROUNDED TO bf16 (both sides
  then compute on the same representable numbers), and the ORACLE (fp32, CPU, oracle/fid_t5_oracle.py::fid_generate) decodes them:
  its tokens are the fixture's expected output.  Round 4 trained this inside the test (up to 4 × 600 steps, run-to-run different
  because the weight gradients use float atomics); the committed weights make the test a pure decode comparison.
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lako_amd import FiDConfig, FiDT5  # noqa: E402
from lako_amd import util as U  # noqa: E402
from lako_amd.model import plain_name  # noqa: E402
from oracle import fid_t5_oracle as O  # noqa: E402

DIMS = dict(vocab_size=256, d_model=512, d_kv=64, d_ff=64, num_layers=1, num_decoder_layers=2, num_heads=8)
B, N, L, T, ML = 4, 10, 200, 6, 9


def main():
    cfg = FiDConfig(dropout_rate=0.0, **DIMS)
    dims = O.T5Dims(dropout=0.0, **DIMS)
    ids_c, mask_c, labels_c = O.synthetic_batch(B, N, L, T, cfg.vocab_size, seed=91)
    ids, mask, labels = ids_c.cuda(), mask_c.cuda(), labels_c.cuda()
    opt = types.SimpleNamespace(optim="adamw", lr=1e-3, weight_decay=0.0, scheduler="fixed", fixed_lr=True, scheduler_steps=None,
                                total_steps=4000, warmup_steps=0)
    for attempt in range(6):
        torch.manual_seed(attempt)
        m = FiDT5(cfg, dtype=torch.bfloat16, seed=1 + attempt).cuda().train()
        optimizer, scheduler = U.set_optim(opt, m)
        last = None
        for k in range(800):
            loss = m(input_ids=ids, attention_mask=mask, labels=labels)[0]
            loss.backward()
            U.clip_grad_norm_(m, 1.0)
            optimizer.step()
            scheduler.step()
            m.zero_grad()
            if k % 20 == 19:
                last = loss.item()
                if last < 0.003:
                    break
        print(f"attempt {attempt}: loss {last:.5f} after {k + 1} steps", flush=True)
        if last >= 0.02:
            continue
        w = {plain_name(n): p.detach().float().cpu().to(torch.bfloat16).float() for n, p in m.named_parameters()}     # bf16-representable
        want = O.fid_generate(w, dims, ids_c, mask_c, ML)
        nlab = [int((labels_c[b] != -100).sum()) for b in range(B)]
        ok = all(want[b, 1:1 + nlab[b]].tolist() == labels_c[b, :nlab[b]].tolist() for b in range(B))
        # margins of the oracle's teacher-forced logits: the top-1 / top-2 gap at every answer position must dwarf bf16 noise
        _, logits = O.fid_forward(w, dims, ids_c, mask_c, labels_c, training=False)
        top2 = logits.float().topk(2, dim=-1).values
        gap = (top2[..., 0] - top2[..., 1])[labels_c != -100].min().item()
        print(f"  oracle reproduces the answers: {ok}; smallest top-1 / top-2 logit gap {gap:.3f}", flush=True)
        if ok and gap > 1.0:
            out = {"w." + k_: v.to(torch.bfloat16).view(torch.int16).numpy() for k_, v in w.items()}
            out.update(ids=ids_c.numpy(), mask=mask_c.numpy(), labels=labels_c.numpy(), tokens=want.numpy(),
                       dims=np.array([DIMS[k_] for k_ in ("vocab_size", "d_model", "d_kv", "d_ff", "num_layers", "num_decoder_layers", "num_heads")]),
                       max_length=np.array(ML), min_gap=np.array(gap))
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            path = os.path.join(ROOT, "gpurun_out", "gen_d512.npz")
            np.savez_compressed(path, **out)
            print(f"wrote {path}: {os.path.getsize(path) / 1e6:.1f} MB, tokens {want.tolist()}")
            return 0
    print("no usable fit")
    return 1


if __name__ == "__main__":
    sys.exit(main())
