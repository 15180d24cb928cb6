#!/usr/bin/env python
"""Median time of the encoder attention forward on the benchmark's ragged batch (320 passages x 12 heads, lengths U{100..200}, bias + dropout),
with the caches flushed before every launch (in the training step the rows are cold).  Environment switches of the kernel are read once per
process: run one process per setting.    python tools/attn_fwd_time.py [label]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lako_amd.ops import HipOps  # noqa: E402

ops = HipOps()
dev = torch.device("cuda:0")
T = torch.bfloat16
H, dk, L, BN = 12, 64, 200, 320
inner = H * dk
g = torch.Generator().manual_seed(1)
lens = torch.randint(L // 2, L + 1, (BN,), generator=g)
off = torch.zeros(BN + 1, dtype=torch.int32)
off[1:] = torch.cumsum(lens, 0)
M = int(off[-1])
off = off.to(dev)
qkv = (torch.randn(1, M, 3 * inner, device=dev) * 0.5).to(T)
out = torch.zeros(1, M, inner, dtype=T, device=dev)
st = torch.zeros(BN, H, L, 4, device=dev)
rel = torch.randn(H, 2 * L - 1, device=dev)
order = torch.argsort(lens, descending=True, stable=True).to(torch.int32).to(dev)
heads = lambda t, c0: t[:, :, c0:c0 + inner].unflatten(2, (H, dk))      # noqa: E731
kw = dict(rel_bias=rel, rel_off=L - 1, drop=(0.1, 5, 6), q_off=off, k_off=off, max_q=L, max_k=L, order=order)
big = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
ts = []
for r in range(14):
    big.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.attn_fwd(heads(qkv, 0), heads(qkv, inner), heads(qkv, 2 * inner), out.unflatten(2, (H, dk)), st, **kw)
    e1.record()
    torch.cuda.synchronize()
    if r >= 2:
        ts.append(e0.elapsed_time(e1) * 1000.0)
print(f"{sys.argv[1] if len(sys.argv) > 1 else ''}: attn_fwd median {sorted(ts)[len(ts) // 2]:.1f} us  min {min(ts):.1f}  checksum {float(out.float().abs().sum()):.6e}", flush=True)
