#!/usr/bin/env python
"""Greedy decode timing at config 2 (T5-base, B=16, N=20, L=200, max_length 50): wall time per generate() call, the host's
share (time until the Python calls return on an idle GPU) and the per-step cost.  `python tools/generate_probe.py`"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from lako_amd import FiDConfig, FiDT5  # noqa: E402

dev = torch.device("cuda:0")
cfg = FiDConfig.named("base")
model = FiDT5(cfg, dtype=torch.bfloat16).cuda()
model.eval()
B, N, L = 16, 20, 200
ids, mask, _ = bench.synthetic_batch(B, N, L, 8, cfg.vocab_size, seed=1, device=dev)
for ml in (2, 50):
    for _ in range(2):
        model.generate(input_ids=ids, attention_mask=mask, max_length=ml)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = model.generate(input_ids=ids, attention_mask=mask, max_length=ml)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"max_length {ml:3d}: generate() returned after {1e3 * (t1 - t0):7.2f} ms, GPU idle after {1e3 * (t2 - t0):7.2f} ms, "
          f"output {tuple(out.shape)}", flush=True)
