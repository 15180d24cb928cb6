#!/usr/bin/env python
"""Timing experiment (experiments build: LAKO_LIB=…/liblako_hip_exp.so): the 256x256 NT kernel with a third of its LDS fragment reads
removed (gemm_nt_debug bit 5: wrong results), with the K-loop DMA removed (bit 0), and both — does the fragment-read traffic slow the
LDS-DMA's writes?   python tools/lds_contention_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lako_amd.ops import HipOps  # noqa: E402

ops = HipOps()
dev = torch.device("cuda:0")
ops.set_tuning("gemm_nt_variant", 2)


def timeit(fn, iters=20):
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


for (M, N, K) in [(8192, 8192, 8192), (64000, 2304, 768), (64000, 768, 3072)]:
    A = (torch.randn(M, K, device=dev) * 0.1).bfloat16()
    B = (torch.randn(N, K, device=dev) * 0.1).bfloat16()
    Cm = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    row = []
    # bit 2 (4): the K-loop never waits for its DMA — nor, through the in-order counter, for the previous tile's stores; bit 3 (8): no stores
    for dbg, what in [(0, "as shipped"), (32, "1/3 fewer LDS fragment reads"), (1, "no K-loop DMA"), (33, "no DMA + fewer reads"),
                      (4, "no vmcnt waits in the K-loop"), (8, "no stores"), (12, "no waits, no stores"), (0, "as shipped")]:
        ops.set_tuning("gemm_nt_debug", dbg)
        us = timeit(lambda: ops.gemm_nt(A, B, Cm))
        row.append(f"{what}: {us:8.1f} us ({2.0 * M * N * K / us / 1e6:7.1f} TF/s)")
    print(f"[{M},{K}]x[{N},{K}]  " + " | ".join(row), flush=True)
