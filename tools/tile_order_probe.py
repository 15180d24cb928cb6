#!/usr/bin/env python
"""Timing experiment: banded tile order (lako_set_tuning "gemm_nt_group_m", negative = forced) on the K = 768 shapes whose
outputs are 9 / 12 / 3 tiles wide (row-major by default)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lako_amd.ops import HipOps  # noqa: E402

ops = HipOps(); dev = torch.device("cuda:0"); T = torch.bfloat16
Me = int(os.environ.get("LAKO_PROBE_TOKENS", "64000"))


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for N, K in ((2304, 768), (3072, 768)):
    A, B = torch.randn(Me, K, device=dev).to(T), torch.randn(N, K, device=dev).to(T)
    C = torch.empty(Me, N, dtype=T, device=dev)
    row = []
    for gm in (8, -8, -12, -16, -24, -32, -64, 8, -16, 8, -32):
        ops.set_tuning("gemm_nt_group_m", gm)
        row.append(f"{gm}:{timeit(lambda: ops.gemm_nt(A, B, C)):7.1f}")
    print(f"[{Me},{K}]x[{N},{K}]  " + "  ".join(row), flush=True)
