#!/usr/bin/env python
"""Is the training step host-bound anywhere?  For a few un-synchronised steps, the HOST time at which forward / backward / optimizer calls RETURN
(everything enqueued) against the GPU time at which the work enqueued up to that point COMPLETES (events recorded at the same places): the host's lead.
A lead near zero at some point means the GPU waits for the launch path there.   python tools/host_lead_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from lako_amd import FiDConfig, FiDT5  # noqa: E402
from lako_amd import util as U  # noqa: E402

dev = torch.device("cuda:0")
cfg = FiDConfig.named("base")
cfg.dropout_rate = 0.1
model = FiDT5(cfg, dtype=torch.bfloat16)
import types  # noqa: E402
with torch.no_grad():
    model._params_by_plain["shared.weight"].mul_(0.05)
opt = types.SimpleNamespace(optim="adamw", lr=1e-4, weight_decay=1e-4, scheduler="linear", scheduler_steps=None, total_steps=400, warmup_steps=24, fixed_lr=False)
model = model.cuda()
model.train()
optimizer, scheduler = U.set_optim(opt, model)
B, N, L, T = 16, 20, 200, 8
batches = [bench.synthetic_batch(B, N, L, T, cfg.vocab_size, seed=i, device=dev, all_valid=False, with_lengths=True) for i in range(4)]


def step(i, marks=None):
    ids, mask, labels, lens = batches[i % 4]

    def mark(name):
        if marks is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append((name, time.perf_counter(), ev))
    mark("start")
    loss = model(input_ids=ids, attention_mask=mask, labels=labels, passage_lengths=lens)[0]
    mark("forward enqueued")
    loss.backward()
    mark("backward enqueued")
    U.clip_grad_norm_(model, 1.0)
    optimizer.step()
    scheduler.step()
    model.zero_grad()
    mark("optimizer enqueued")


for i in range(6):
    step(i)
torch.cuda.synchronize()
all_marks = []
t0 = time.perf_counter()
e0 = torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(4):
    m = []
    step(i, m)
    all_marks.append(m)
torch.cuda.synchronize()
for s, m in enumerate(all_marks):
    for name, th, ev in m:
        host = (th - t0) * 1e3
        gpu = e0.elapsed_time(ev)
        print(f"step {s} {name:20s} host {host:8.2f} ms   GPU reaches it at {gpu:8.2f} ms   host lead {gpu - host:7.2f} ms")
