#!/usr/bin/env python
"""Round 6: the epilogues of the four-wave NT kernel straight from the accumulator layout (gemm_nt4_kernel<MT, false, EPI > 0>: v_permlane16_swap pairs two
n-tiles into 16-byte row segments, no LDS pass) against the same epilogues through the LDS transposition — bit-equality with the eight-wave
kernel first, then µs per launch, interleaved in one process.  gemm_nt_four = 3 / 1: LDS form / direct (the build's default; only the dropout forms have a direct instantiation — the first version of this
tool measured all five: plain / ReLU / alpha are 3 - 7 % SLOWER direct, profiles/r06m_nt4_direct_epilogue.txt).
    python tools/gemm_nt4_direct.py [--rows 47757] [--iters 20] [--rounds 3]"""
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
args = ap.parse_args()
dev = torch.device("cuda:0")
ops = HipOps()
Me = args.rows
shapes = [("qkv  [Me,768]x[2304,768]", (Me, 2304, 768), 9), ("o    [Me,768]x[768,768]", (Me, 768, 768), 3), ("wi   [Me,768]x[3072,768]", (Me, 3072, 768), 9),
          ("dwi  [Me,3072]x[768,3072]", (Me, 768, 3072), 3), ("dqkv [Me,2304]x[768,2304]", (Me, 768, 2304), 3), ("edge [70001,768]x[520,768]", (70001, 520, 768), 9),
          ("edge [70001,768]x[520,768]", (70001, 520, 768), 3)]
EPIS = {"drop": dict(drop=(0.1, 1, 2)), "relu+drop": dict(relu=True, drop=(0.1, 1, 2))}


def run(v, four, A, B, C, **kw):
    ops.set_tuning("gemm_nt_variant", v)
    ops.set_tuning("gemm_nt_four", four)
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


for nm, (M, N, K), v in shapes:
    A = torch.randn(M, K, device=dev).bfloat16()
    B = torch.randn(N, K, device=dev).bfloat16()
    Cr, Ct = torch.empty(M, N, dtype=torch.bfloat16, device=dev), torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for epi, kw in EPIS.items():
        run(2, 1, A, B, Cr, **kw)
        eq = {}
        for four in (3, 1):
            Ct.fill_(float("nan"))
            run(v, four, A, B, Ct, **kw)
            torch.cuda.synchronize()
            eq[four] = torch.equal(Cr.view(torch.int16), Ct.view(torch.int16))
        t = {}
        for _ in range(args.rounds):
            for four in (3, 1):
                t.setdefault(four, []).append(time_fn(lambda: run(v, four, A, B, Ct, **kw)))
        med = {k: sorted(x)[len(x) // 2] for k, x in t.items()}
        print(f"{nm:28s} v{v} {epi:9s} bit-equal to the eight-wave kernel {eq} | LDS {med[3]:7.1f} us | direct {med[1]:7.1f} us | {100 * (med[1] / med[3] - 1):+5.1f} %", flush=True)
