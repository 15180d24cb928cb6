#!/usr/bin/env python
"""Start offsets between the workgroups of the four-wave NT kernel: (ticks of 10 ns, phases)."""
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
cfgs = [(0, 2), (100, 2), (300, 2), (600, 2), (100, 4), (200, 4), (400, 4), (100, 8), (200, 8)]
for nm, (M, N, K), v in [("qkv", (Me, 2304, 768), 9), ("wi", (Me, 3072, 768), 9), ("o", (Me, 768, 768), 3), ("wo", (Me, 768, 3072), 3), ("dqkv", (Me, 768, 2304), 3)]:
    A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16(); C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ops.set_tuning("gemm_nt_variant", v)
    res = {c: [] for c in cfgs}
    for _ in range(3):
        for c in cfgs:
            ops.set_tuning("gemm_nt_dephase", c[0]); ops.set_tuning("gemm_nt_dephase_n", c[1])
            res[c].append(t(lambda: ops.gemm_nt(A, B, C)))
    print(f"{nm:5s} v{v} " + " | ".join(f"{c[0]}x{c[1]}: {sorted(r)[1]:6.1f}" for c, r in res.items()), flush=True)
