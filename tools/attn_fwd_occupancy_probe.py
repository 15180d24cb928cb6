#!/usr/bin/env python
"""Does a third / fourth resident workgroup per CU help the encoder attention FORWARD on short passages?  The kernel's LDS is sized by the
launch's max_k (K + V images of round32(max_k) rows): 200 → 65 KiB → 2 workgroups per CU, 160 → 48 KiB → 3, 128 → 40 KiB → 4.  Same ragged
items (lengths U{lo..hi}), timed with max_k = hi rounded up to 32 and with max_k = 200.      python tools/attn_fwd_occupancy_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lako_amd.ops import HipOps  # noqa: E402

ops = HipOps()
dev = torch.device("cuda:0")
T = torch.bfloat16
H, dk = 12, 64
inner = H * dk
for lo, hi, Bn in ((100, 160, 320), (64, 128, 400), (100, 200, 320)):
    g = torch.Generator().manual_seed(3)
    lens = torch.randint(lo, hi + 1, (Bn,), generator=g).tolist()
    off = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=dev)
    rows = int(off[-1])
    qkv = (torch.randn(1, rows, 3 * inner, device=dev) * 0.5).to(T)
    heads = lambda t, c0: t[:, :, c0:c0 + inner].unflatten(2, (H, dk))      # noqa: E731
    order = torch.argsort(torch.tensor(lens), descending=True, stable=True).to(torch.int32).to(dev)
    res = {}
    for mk in sorted({-(-hi // 32) * 32, 200, 224}):
        if mk < hi:
            continue
        rel = torch.randn(H, 2 * mk - 1, device=dev)
        out = torch.zeros(1, rows, inner, dtype=T, device=dev)
        st = torch.zeros(Bn, H, mk, 4, device=dev)
        kw = dict(rel_bias=rel, rel_off=mk - 1, drop=(0.1, 5, 6), q_off=off, k_off=off, max_q=mk, max_k=mk, order=order)
        ts = []
        for r in range(6):
            for _ in range(2):
                ops.attn_fwd(heads(qkv, 0), heads(qkv, inner), heads(qkv, 2 * inner), out.unflatten(2, (H, dk)), st, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.attn_fwd(heads(qkv, 0), heads(qkv, inner), heads(qkv, 2 * inner), out.unflatten(2, (H, dk)), st, **kw)
            e1.record()
            torch.cuda.synchronize()
            if r:
                ts.append(e0.elapsed_time(e1) * 100.0)
        res[mk] = sorted(ts)[len(ts) // 2]
    print(f"lengths U{{{lo}..{hi}}} x {Bn} passages x {H} heads ({rows} tokens): " + "   ".join(f"max_k {k}: {v:6.1f} us" for k, v in res.items()), flush=True)
