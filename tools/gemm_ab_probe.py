#!/usr/bin/env python
"""Interleaved A/B of one tuning key on the encoder's NT shapes at the benchmark's 47 757 valid rows (one process, R rounds):
    python tools/gemm_ab_probe.py gemm_nt_stagger 1 2 0        (key, then the values to compare)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lako_amd.ops import HipOps  # noqa: E402

key, vals = sys.argv[1], [int(v) for v in sys.argv[2:]]
ops = HipOps()
dev = torch.device("cuda:0")
T = torch.bfloat16
M = int(os.environ.get("ROWS", "47757"))
drop = (0.1, 1, 2)
cases = [("qkv  K768 N2304 plain", 2304, 768, {}), ("wi   K768 N3072 relu+drop", 3072, 768, dict(relu=True, drop=drop)),
         ("wo   K3072 N768 res+drop", 768, 3072, dict(resid=True, drop=drop)), ("o    K768 N768 res+drop", 768, 768, dict(resid=True, drop=drop)),
         ("dxqkv K2304 N768 plain", 768, 2304, {}), ("dpre K768 N3072 auxmask", 3072, 768, dict(aux=True)), ("dx   K3072 N768 plain", 768, 3072, {}),
         ("sq 8192^3", 8192, 8192, {})]
R = 5
for nm, N, K, kw in cases:
    Mm = 8192 if nm.startswith("sq") else M
    A, B = torch.randn(Mm, K, device=dev).to(T), torch.randn(N, K, device=dev).to(T)
    C = torch.empty(Mm, N, dtype=T, device=dev)
    k2 = {}
    if kw.get("resid"):
        k2["resid"] = torch.randn(Mm, N, device=dev).to(T)
    if kw.get("aux"):
        k2["aux"], k2["aux_scale"] = torch.randn(Mm, N, device=dev).to(T), 1.1
    if kw.get("relu"):
        k2["relu"] = True
    if kw.get("drop"):
        k2["drop"] = kw["drop"]
    res = {v: [] for v in vals}
    outs = {}
    for v in vals:       # the values must not change the result: compare the outputs bit for bit with the first value's
        ops.set_tuning(key, v)
        ops.gemm_nt(A, B, C, **k2)
        torch.cuda.synchronize()
        outs[v] = C.clone()
    same = " ".join("same" if torch.equal(outs[v], outs[vals[0]]) else f"DIFF({(outs[v].float() - outs[vals[0]].float()).abs().max().item():.3g})" for v in vals[1:])
    for r in range(R + 1):
        for v in vals:
            ops.set_tuning(key, v)
            for _ in range(2):
                ops.gemm_nt(A, B, C, **k2)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.gemm_nt(A, B, C, **k2)
            e1.record()
            torch.cuda.synchronize()
            if r:
                res[v].append(e0.elapsed_time(e1) * 100.0)
    print(f"{nm:28s} " + "   ".join(f"{key}={v}: median {sorted(res[v])[len(res[v]) // 2]:7.1f} min {min(res[v]):7.1f} us" for v in vals) + "   " + same, flush=True)
