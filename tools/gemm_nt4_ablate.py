#!/usr/bin/env python
"""Experiments build only (LAKO_LIB=lako_amd/liblako_hip_exp.so): what a tile of the four-wave kernel costs outside its K loop.
debug 0 = product; 8 = epilogue without its global stores; 128 = no epilogue at all (results are garbage in both)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lako_amd.ops import HipOps
dev = torch.device("cuda:0"); ops = HipOps(); Me = 47757
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for nm, (M, N, K) in [("qkv", (Me, 2304, 768)), ("o", (Me, 768, 768)), ("wi", (Me, 3072, 768)), ("wo", (Me, 768, 3072)), ("8192^3", (8192,) * 3)]:
    A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16(); C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    line = f"{nm:8s}"
    for v in (9, 3):
        ops.set_tuning("gemm_nt_variant", v)
        for dbg in (0, 8, 128):
            ops.set_tuning("gemm_nt_debug", dbg)
            us = sorted(t(lambda: ops.gemm_nt(A, B, C)) for _ in range(3))[1]
            line += f" | v{v} dbg{dbg}: {us:7.1f}"
    print(line, flush=True)
