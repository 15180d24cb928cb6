#!/usr/bin/env python
"""Round 6: the four-wave weight-gradient kernel (csrc/gemm_tn4.h, `gemm_tn_four` 1) against the eight-wave one (0): the encoder's grouped
launch of all twelve layers (48 problems, hybrid schedule split_k = -3: 1 296 tiles) and single products, interleaved in one process.
Full-K units have one contributor per element: bit-equal results there; the 16 remainder tiles add K-pieces by float atomics (order)."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lako_amd.ops import HipOps
ap = argparse.ArgumentParser(); ap.add_argument("--rows", type=int, default=47757); ap.add_argument("--iters", type=int, default=5); args = ap.parse_args()
dev = torch.device("cuda:0"); ops = HipOps(); K = args.rows
def rnd(*s): return (torch.randn(*s, device=dev) * 0.5).bfloat16()
def t(fn, n):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
dqkv, xn, dctx, ctx, dpre, dy, a1 = rnd(K, 2304), rnd(K, 768), rnd(K, 768), rnd(K, 768), rnd(K, 3072), rnd(K, 768), rnd(K, 3072)
shapes = [(dqkv, xn), (dctx, ctx), (dpre, xn), (dy, a1)]
def problems(Cs): return [(A, B, C, 1.0) for l in range(12) for (A, B), C in zip(shapes, Cs[l])]
Cs = {f: [[torch.zeros(A.shape[1], B.shape[1], device=dev) for A, B in shapes] for _ in range(12)] for f in (0, 1)}
fl = sum(2.0 * A.shape[1] * B.shape[1] * K for A, B in shapes) * 12
for f in (0, 1):
    ops.set_tuning("gemm_tn_four", f)
    ops.gemm_tn_grouped(problems(Cs[f]), split_k=-3)
torch.cuda.synchronize()
worst = 0.0; nbits = 0
for l in range(12):
    for c0, c1 in zip(Cs[0][l], Cs[1][l]):
        worst = max(worst, float((c0 - c1).abs().max() / c0.abs().max())); nbits += int((c0 != c1).sum())
print(f"grouped hybrid: max |diff| / max |C| = {worst:.2e}, elements that differ {nbits} (float atomics of the 16 remainder tiles)")
res = {0: [], 1: []}
for _ in range(3):
    for f in (0, 1):
        ops.set_tuning("gemm_tn_four", f)
        res[f].append(t(lambda: ops.gemm_tn_grouped(problems(Cs[f]), split_k=-3), args.iters))
for f in (0, 1):
    us = sorted(res[f])[1]
    print(f"grouped 48 problems, tn_four {f}: {us:9.1f} us = {fl / us / 1e6:7.1f} TFLOP/s")
for nm, (A, B) in zip(("dWqkv", "dWo", "dWi", "dWo2"), shapes):
    C0, C1 = torch.zeros(A.shape[1], B.shape[1], device=dev), torch.zeros(A.shape[1], B.shape[1], device=dev)
    ops.set_tuning("gemm_tn_four", 0); ops.gemm_tn(A, B, C0, split_k=-2)
    ops.set_tuning("gemm_tn_four", 1); ops.gemm_tn(A, B, C1, split_k=-2)
    torch.cuda.synchronize()
    ref = A.float().t() @ B.float()
    e0, e1 = float((C0 - ref).abs().max() / ref.abs().max()), float((C1 - ref).abs().max() / ref.abs().max())
    line = f"{nm:6s} one split (overwrite): equal {torch.equal(C0, C1)} rel err vs fp32 {e0:.1e} / {e1:.1e} |"
    for f in (0, 1):
        ops.set_tuning("gemm_tn_four", f)
        us = sorted(t(lambda: ops.gemm_tn(A, B, C0), 10) for _ in range(3))[1]
        line += f" auto split, tn_four {f}: {us:8.1f} us {2.0 * A.shape[1] * B.shape[1] * K / us / 1e6:7.1f} TF |"
    print(line, flush=True)
