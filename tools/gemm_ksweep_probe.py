#!/usr/bin/env python
"""Per-tile cost of the NT kernel outside its K loop: [ROWS, K] x [N, K] at several K for one N — time(K) per round = intercept + slope·(K / 64).
With LAKO_LIB=…/liblako_hip_exp.so the debug bits of `gemm_nt_debug` can be set (DEBUG=8: no epilogue stores; 1: no K-loop DMA; …).
Prints µs per call, rounds, µs per round, and the fit."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lako_amd.ops import HipOps  # noqa: E402

ops = HipOps()
dev = torch.device("cuda:0")
ROWS = int(os.environ.get("ROWS", "47757"))
N = int(os.environ.get("N", "2304"))
KS = [int(k) for k in os.environ.get("KS", "128,256,512,768,1024,1536,2304").split(",")]
VARIANT = int(os.environ.get("VARIANT", "-1"))
DEBUGS = [int(d) for d in os.environ.get("DEBUG", "0").split(",")]
ops.set_tuning("gemm_nt_variant", VARIANT)
bm = 288 if VARIANT in (-1, 8) else 256
for dbg in DEBUGS:
    if dbg or os.environ.get("LAKO_LIB"):
        ops.set_tuning("gemm_nt_debug", dbg)
    pts = []
    for K in KS:
        A = torch.randn(ROWS, K, device=dev).to(torch.bfloat16)
        B = torch.randn(N, K, device=dev).to(torch.bfloat16)
        C = torch.empty(ROWS, N, dtype=torch.bfloat16, device=dev)
        ts = []
        for rep in range(6):
            ops.gemm_nt(A, B, C)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.gemm_nt(A, B, C)
            e1.record()
            torch.cuda.synchronize()
            if rep:
                ts.append(e0.elapsed_time(e1) * 100.0)
        us = statistics.median(ts)
        tiles = -(-ROWS // bm) * -(-N // 256)
        rounds = -(-tiles // 256)
        pts.append((K / 64.0, us / rounds))
        print(f"debug {dbg:3d}  K {K:5d}: {us:8.1f} us per call, {tiles} tiles = {rounds} rounds, {us / rounds:6.2f} us per round", flush=True)
        del A, B, C
    n = len(pts)
    sx, sy = sum(p[0] for p in pts), sum(p[1] for p in pts)
    sxx, sxy = sum(p[0] ** 2 for p in pts), sum(p[0] * p[1] for p in pts)
    slope = (n * sxy - sx * sy) / (n * sxx - sx * sx)
    icpt = (sy - slope * sx) / n
    print(f"debug {dbg:3d}  fit: {icpt:5.2f} us per tile outside the K loop + {slope:5.3f} us per K-step", flush=True)
