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
ops.set_tuning("gemm_nt_debug", 256)      # (launch_nt4 honours the de-phase knobs under this bit only: the shipped launches run in lockstep)
for nm, (M, N, K), v, epi in [("qkv", (Me, 2304, 768), 9, "plain"), ("wi", (Me, 3072, 768), 9, "relu+drop"), ("o", (Me, 768, 768), 3, "res+drop"), ("wo", (Me, 768, 3072), 3, "res+drop"),
                              ("dpre", (Me, 3072, 768), 9, "aux"), ("dqkv", (Me, 768, 2304), 3, "plain")]:
    A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16(); C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    S = torch.randn(M, N, device=dev).bfloat16()
    kw = {"plain": {}, "relu+drop": dict(relu=True, drop=(0.1, 1, 2)), "res+drop": dict(resid=S, drop=(0.1, 1, 2)), "aux": dict(aux=S, aux_scale=1.1)}[epi]
    ops.set_tuning("gemm_nt_variant", v)
    res = {c: [] for c in cfgs}
    for _ in range(3):
        for c in cfgs:
            ops.set_tuning("gemm_nt_dephase", c[0]); ops.set_tuning("gemm_nt_dephase_n", c[1])
            res[c].append(t(lambda: ops.gemm_nt(A, B, C, **kw)))
    print(f"{nm:5s} v{v} {epi:9s} " + " | ".join(f"{c[0]}x{c[1]}: {sorted(r)[1]:6.1f}" for c, r in res.items()), flush=True)
