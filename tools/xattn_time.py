#!/usr/bin/env python
"""Times the kernels of the encoder-state-space cross-attention (csrc/xattn.hip) at one decoder layer of a configuration
(default config 2: 16 samples × ≈3000 valid keys, T = 8, T5-base) with HIP events, next to the projected formulation's
pieces (the K/V projection GEMM share of one layer and the generic cross-attention kernels)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lako_amd.ops import HipOps  # noqa: E402

ops = HipOps()
dev = torch.device("cuda:0")
B = int(os.environ.get("XB", 16))
NL = int(os.environ.get("XKEYS", 4000))
T = int(os.environ.get("XT", 8))
H = int(os.environ.get("XH", 12))
D, inner, R = H * 64, H * 64, T * H
BF = torch.bfloat16
g = torch.Generator().manual_seed(1)
lens = torch.randint(int(NL * 0.55), NL + 1, (B,), generator=g).tolist()
k = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
p = np.concatenate([[0], np.cumsum([(n + 255) // 256 * 256 for n in lens])]).astype(np.int32)
k_off, p_off, ptot, M = torch.from_numpy(k).to(dev), torch.from_numpy(p).to(dev), int(p[-1]), int(k[-1])
E = (torch.randn(M, D, device=dev) * 0.5).to(BF)
q = (torch.randn(B * T, inner, device=dev) * 0.3).to(BF)
W = (torch.randn(2 * inner, D, device=dev) * 0.05).to(BF)
Wt = W.t().contiguous()
DQ = torch.zeros(B, 2 * R, D, dtype=BF, device=dev)
S = torch.zeros(R, ptot, device=dev)
dP = torch.zeros(R, ptot, device=dev)
PS = torch.zeros(2 * R, ptot, dtype=BF, device=dev)
st = torch.zeros(B, R, 2, device=dev)
Z = int(os.environ.get("XZ", 2))
Cp = torch.zeros(Z, B, R, D, device=dev)
ctx = torch.zeros(B * T, inner, dtype=BF, device=dev)
G = torch.zeros(2 * inner, D, device=dev)
dE = torch.zeros(M + 8, D, device=dev)
drop = (0.1, 1, 2) if os.environ.get("NODROP") != "1" else None


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


items = []
for b, n in enumerate(lens):
    n8 = (n + 7) // 8 * 8
    items.append((PS[:, int(p[b]):int(p[b]) + n8], DQ[b], dE[int(k[b]):int(k[b]) + n8], 1.0))
steps = {
    "expand (Q')": lambda: ops.headbatch_nt(q.view(B, T, H, 64), Wt[:, :inner].unflatten(1, (H, 64)).permute(1, 0, 2), DQ[:, R:].unflatten(1, (T, H))),
    "scores": lambda: ops.xattn_scores(DQ[:, R:], E, k_off, p_off, ptot, S),
    "softmax fwd": lambda: ops.xattn_softmax_fwd(S, st, PS[:R], k_off, p_off, T, H, NL, drop),
    f"context (Z={Z})": lambda: ops.xattn_context(PS[:R], E, k_off, p_off, Cp),
    "contract (ctx)": lambda: ops.headbatch_nt(Cp.unflatten(2, (T, H)), W[inner:].unflatten(0, (H, 64)), ctx.view(B, T, H, 64)),
    "wgrad": lambda: ops.headbatch_tn(q.view(B, T, H, 64), Cp.unflatten(2, (T, H)), G[:inner].unflatten(0, (H, 64))),
    "softmax bwd": lambda: ops.xattn_softmax_bwd(S, dP, st, PS[R:], k_off, p_off, T, H, NL, drop),
    "dE (1 layer's K)": lambda: ops.gemm_tn_grouped(items, split_k=1),
}
tot = 0.0
print(f"B {B}  keys {M} (padded {ptot})  R {R}  D {D}")
for name, fn in steps.items():
    us = timeit(fn)
    tot += us
    print(f"  {name:22s} {us:8.1f} us")
fwd = sum(timeit(steps[k_]) for k_ in ["expand (Q')", "scores", "softmax fwd", f"context (Z={Z})", "contract (ctx)"])
print(f"forward chain (5 launches) ≈ {fwd:.1f} us per layer")

# the encoder-state gradient at its real depth (12 layers: K = 24·R), for two row strides of the probability matrix
for ld in (ptot, 65536, 65536 + 64):
    LD = 12
    PSb = torch.zeros(LD * 2 * R, ld, dtype=BF, device=dev)
    DQb = torch.zeros(B, LD * 2 * R, D, dtype=BF, device=dev)
    its = []
    for b, n in enumerate(lens):
        n8 = (n + 7) // 8 * 8
        its.append((PSb[:, int(p[b]):int(p[b]) + n8], DQb[b], dE[int(k[b]):int(k[b]) + n8], 1.0))
    print(f"  dE K={LD * 2 * R} row stride {ld:6d}: {timeit(lambda: ops.gemm_tn_grouped(its, split_k=1)):8.1f} us (two launches of 8 samples)")
