#!/usr/bin/env python
"""Where the host time of ENQUEUEING one training step goes (bench.py reports its total as config.host_enqueue_ms_per_step):
cProfile over a few config-2 steps, each started on an idle GPU and not synchronised inside.  python tools/host_profile.py [steps]"""
import cProfile
import os
import pstats
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_batch  # noqa: E402
from lako_amd import FiDConfig, FiDT5  # noqa: E402
from lako_amd import util as U  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")
cfg = FiDConfig.named("base", dropout_rate=0.1)
torch.manual_seed(0)
model = FiDT5(cfg, dtype=torch.bfloat16, seed=0)
with torch.no_grad():
    model._params_by_plain["shared.weight"].mul_(0.05)
model = model.cuda().train()
opt = types.SimpleNamespace(optim="adamw", lr=1e-4, weight_decay=1e-4, scheduler="linear", scheduler_steps=None, total_steps=100,
                            warmup_steps=6, fixed_lr=False)
optimizer, scheduler = U.set_optim(opt, model)
batches = [synthetic_batch(16, 20, 200, 8, cfg.vocab_size, seed=i, device=dev, with_lengths=True) for i in range(4)]


def step(i):
    ids, mask, labels, lens = batches[i % 4]
    loss = model(input_ids=ids, attention_mask=mask, labels=labels, passage_lengths=lens)[0]
    loss.backward()
    U.clip_grad_norm_(model, 1.0)
    optimizer.step()
    scheduler.step()
    model.zero_grad()


for i in range(4):
    step(i)
torch.cuda.synchronize()
ts = []
for i in range(steps):
    t0 = time.perf_counter()
    step(i)
    ts.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
print("host ms per step without the profiler:", " ".join(f"{t:.2f}" for t in ts))
torch.autograd.set_multithreading_enabled(False)      # the hand-written backward runs in THIS thread: the profile sees it
pr = cProfile.Profile()
for i in range(steps):
    pr.enable()
    step(i)
    pr.disable()
    torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(22)
