#!/usr/bin/env python
"""Round 6: the four-wave hand-scheduled 256² NT kernel (csrc/gemm_nt4.h, tuning variant 9) against the shipped heuristics (variant -1), the
forced eight-wave 256² kernel (variant 2) and — a yardstick only — torch.matmul (hipBLASLt), interleaved in ONE process on the same random
operands.  Checks bit-equality of variant 9 with variant 2 first (same accumulation order per element).
    python tools/gemm_nt4_probe.py [--rows 47757] [--iters 20] [--rounds 3] [--no-vendor]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lako_amd.ops import HipOps  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=47757)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--no-vendor", action="store_true")
ap.add_argument("--variants", default="-1,2,9")
ap.add_argument("--shapes", default="all")
args = ap.parse_args()
dev = torch.device("cuda:0")
ops = HipOps()
Me = args.rows
shapes = [("qkv  [Me,768]x[2304,768]", (Me, 2304, 768)), ("o    [Me,768]x[768,768]", (Me, 768, 768)), ("wi   [Me,768]x[3072,768]", (Me, 3072, 768)),
          ("wo   [Me,3072]x[768,3072]", (Me, 768, 3072)), ("dqkv [Me,2304]x[768,2304]", (Me, 768, 2304)), ("4096^3", (4096,) * 3), ("8192^3", (8192,) * 3)]
if args.shapes != "all":
    keep = set(args.shapes.split(","))
    shapes = [s for s in shapes if s[0].split()[0] in keep]
variants = [int(v) for v in args.variants.split(",")]


def run(v, A, B, C, **kw):
    ops.set_tuning("gemm_nt_variant", v)
    ops.gemm_nt(A, B, C, **kw)


def time_fn(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / args.iters


for nm, (M, N, K) in shapes:
    A = torch.randn(M, K, device=dev).bfloat16()
    B = torch.randn(N, K, device=dev).bfloat16()
    C2, C9 = torch.empty(M, N, dtype=torch.bfloat16, device=dev), torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    run(2, A, B, C2)
    C9.fill_(float("nan"))
    run(9, A, B, C9)
    torch.cuda.synchronize()
    same = torch.equal(C2.view(torch.int16), C9.view(torch.int16))
    nbad = 0 if same else int((C2.view(torch.int16) != C9.view(torch.int16)).sum())
    # relu + dropout epilogue too
    run(2, A, B, C2, relu=True, drop=(0.1, 1, 2))
    run(9, A, B, C9, relu=True, drop=(0.1, 1, 2))
    torch.cuda.synchronize()
    same_rd = torch.equal(C2.view(torch.int16), C9.view(torch.int16))
    fl = 2.0 * M * N * K
    best = {}
    for _ in range(args.rounds):
        for v in variants:
            us = time_fn(lambda: run(v, A, B, C9))
            best.setdefault(v, []).append(us)
        if not args.no_vendor:
            us = time_fn(lambda: torch.matmul(A, B.t(), out=C2))
            best.setdefault("vendor", []).append(us)
    line = f"{nm:28s} equal={same}({nbad}) relu+drop equal={same_rd} |"
    for k, v in best.items():
        med = sorted(v)[len(v) // 2]
        line += f" v{k}: {med:8.1f} us {fl / med / 1e6:7.1f} TF (min {min(v):.1f}) |"
    print(line, flush=True)
    del A, B, C2, C9
