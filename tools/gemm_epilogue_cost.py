#!/usr/bin/env python
"""What each part of an NT epilogue costs on the step's shapes (47 757 rows): plain / relu / relu + dropout / residual / residual + dropout /
aux mask, interleaved in one process, median of R rounds of 10 launches.    python tools/gemm_epilogue_cost.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lako_amd.ops import HipOps  # noqa: E402

ops = HipOps()
dev = torch.device("cuda:0")
T = torch.bfloat16
M = int(os.environ.get("ROWS", "47757"))
drop = (0.1, 1, 2)
R = 5
for N, K in ((3072, 768), (768, 3072), (768, 768)):
    A, B = torch.randn(M, K, device=dev).to(T), torch.randn(N, K, device=dev).to(T)
    C = torch.empty(M, N, dtype=T, device=dev)
    side = torch.randn(M, N, device=dev).to(T)
    variants = [("plain", {}), ("relu", dict(relu=True)), ("relu+drop", dict(relu=True, drop=drop)), ("drop", dict(drop=drop)),
                ("resid", dict(resid=side)), ("resid+drop", dict(resid=side, drop=drop)), ("auxmask", dict(aux=side, aux_scale=1.1))]
    res = {n: [] for n, _ in variants}
    for r in range(R + 1):
        for n, kw in variants:
            for _ in range(2):
                ops.gemm_nt(A, B, C, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.gemm_nt(A, B, C, **kw)
            e1.record()
            torch.cuda.synchronize()
            if r:
                res[n].append(e0.elapsed_time(e1) * 100.0)
    print(f"[{M},{K}]x[{N},{K}]  " + "  ".join(f"{n} {sorted(v)[len(v) // 2]:6.1f}" for n, v in res.items()) + "  us", flush=True)
