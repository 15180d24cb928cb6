#!/usr/bin/env python
"""Timing experiment: cache-policy bits on the output stores of the 256x256 NT kernel (lako_set_tuning "gemm_nt_store_aux").
Needs the experiments build: LAKO_EXPERIMENTS=1 bash lako_amd/csrc/build.sh; LAKO_LIB=lako_amd/liblako_hip_exp.so python tools/store_policy_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lako_amd.ops import HipOps  # noqa: E402

ops = HipOps(); dev = torch.device("cuda:0"); T = torch.bfloat16
Me = 64000


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


for N, K in ((2304, 768), (3072, 768), (768, 768), (18432, 768)):
    A, B = torch.randn(Me, K, device=dev).to(T), torch.randn(N, K, device=dev).to(T)
    C = torch.empty(Me, N, dtype=T, device=dev)
    row = []
    for aux in (0, 1, 2, 3, 16, 17, 18, 19, 0):
        ops.set_tuning("gemm_nt_store_aux", aux)
        row.append(f"{aux}:{timeit(lambda: ops.gemm_nt(A, B, C)):7.1f}")
    print(f"[{Me},{K}]x[{N},{K}]  " + "  ".join(row), flush=True)
