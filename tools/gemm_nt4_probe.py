#!/usr/bin/env python
"""Round 6: the four-wave hand-scheduled 256² NT kernel (csrc/gemm_nt4.h, tuning variant 9: 256-row tiles, 3: 192-row tiles) against the shipped heuristics (variant -1), the
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
ap.add_argument("--variants", default="-1,2,9,3")
ap.add_argument("--epis", default="plain,relu+drop,res+drop,aux")
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


EPIS = {"plain": {}, "relu+drop": dict(relu=True, drop=(0.1, 1, 2)), "res+drop": dict(resid=True, drop=(0.1, 1, 2)), "aux": dict(aux=True)}
for nm, (M, N, K) in shapes:
    A = torch.randn(M, K, device=dev).bfloat16()
    B = torch.randn(N, K, device=dev).bfloat16()
    Cr, Ct = torch.empty(M, N, dtype=torch.bfloat16, device=dev), torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    side = torch.randn(M, N, device=dev).bfloat16()
    fl = 2.0 * M * N * K
    for epi in args.epis.split(","):
        kw = dict(EPIS[epi])
        if kw.pop("resid", False):
            kw["resid"] = side
        if kw.pop("aux", False):
            kw["aux"], kw["aux_scale"] = side, 1.1
        run(2, A, B, Cr, **kw)
        eq = {}
        for v in variants:
            if v in (3, 9):
                Ct.fill_(float("nan"))
                run(v, A, B, Ct, **kw)
                torch.cuda.synchronize()
                eq[v] = torch.equal(Cr.view(torch.int16), Ct.view(torch.int16))
        best = {}
        for _ in range(args.rounds):
            for v in variants:
                best.setdefault(v, []).append(time_fn(lambda: run(v, A, B, Ct, **kw)))
            if not args.no_vendor and epi == "plain":
                best.setdefault("vendor", []).append(time_fn(lambda: torch.matmul(A, B.t(), out=Ct)))
        line = f"{nm:26s} {epi:9s} bit-equal to v2: {eq} |"
        for k, v in best.items():
            med = sorted(v)[len(v) // 2]
            line += f" v{k}: {med:7.1f} us {fl / med / 1e6:6.0f} TF |"
        print(line, flush=True)
    del A, B, Cr, Ct, side
