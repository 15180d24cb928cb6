#!/usr/bin/env python
"""Time per ROUND of the chip against the number of K-steps: rounds x (a + b nk) — a = what a tile costs outside its K loop, b = one K-step.
LAKO_LIB=…exp.so adds the column without any epilogue (gemm_nt_debug 128)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lako_amd.ops import HipOps
dev = torch.device("cuda:0"); ops = HipOps()
exp = bool(os.environ.get("LAKO_LIB"))
M, N = int(sys.argv[1]) if len(sys.argv) > 1 else 47757, int(sys.argv[2]) if len(sys.argv) > 2 else 2304
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
tiles = -(-M // 256) * -(-N // 256); rounds = -(-tiles // 256)
print(f"M {M} N {N}: {tiles} tiles of 256^2 = {tiles / 256:.2f} rounds")
for K in (256, 512, 768, 1024, 1536, 2304, 3072, 6144):
    A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16(); C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    line = f"K {K:5d} nk {K // 64:3d}"
    for v in (9, 2):
        ops.set_tuning("gemm_nt_variant", v)
        for dbg in ((0, 128) if exp and v == 9 else (0,)):
            if exp: ops.set_tuning("gemm_nt_debug", dbg)
            us = sorted(t(lambda: ops.gemm_nt(A, B, C)) for _ in range(3))[1]
            line += f" | v{v}{' noepi' if dbg else ''}: {us:7.1f} us = {us / rounds:6.2f}/round"
    print(line, flush=True)
    del A, B, C
