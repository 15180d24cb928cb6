#!/usr/bin/env python
"""Round 6: the MX fp8 product on the four-wave tile (csrc/gemm_nt4_mx.h; `gemm_nt_four` 1, variants 9 / 3 force 256- / 192-row tiles) against the
eight-wave MX kernel (`gemm_nt_four` 0) and the bf16 four-wave kernel on the same logical shape — config 5's forward products (T5-large:
d_model 1024, d_ff 4096, 100 passages x 200 tokens x batch 8 at ~75 % valid = 120 000 rows) — µs per launch, interleaved in one process.
    python tools/gemm_mx4_probe.py [--rows 120000] [--iters 10] [--rounds 3]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lako_amd.ops import HipOps  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=120000)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--rounds", type=int, default=3)
args = ap.parse_args()
dev = torch.device("cuda:0")
ops = HipOps()
Me = args.rows
shapes = [("qkv  [Me,1024]x[3072,1024]", (Me, 3072, 1024), {}), ("wi   [Me,1024]x[4096,1024] relu+drop", (Me, 4096, 1024), dict(relu=True, drop=(0.1, 1, 2))),
          ("kv   [Me,1024]x[8192,1024]", (Me, 8192, 1024), {}), ("base qkv [48000,768]x[2304,768]", (48000, 2304, 768), {}), ("wo   [Me,4096]x[1024,4096]", (Me, 1024, 4096), {})]


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


for nm, (M, N, K), kw in shapes:
    A = torch.randn(M, K, device=dev).bfloat16()
    B = (torch.randn(N, K, device=dev) * 0.5).bfloat16()
    Aq, Bq = (torch.zeros(t.shape, dtype=torch.uint8, device=dev) for t in (A, B))
    As, Bs = (torch.zeros(t.shape[0], ops.mx_scale_cols(K), dtype=torch.uint8, device=dev) for t in (A, B))
    ops.mx_quantize(A, Aq, As)
    ops.mx_quantize(B, Bq, Bs)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)

    def mx(four, variant):
        ops.set_tuning("gemm_nt_four", four)
        ops.set_tuning("gemm_nt_variant", variant)
        ops.gemm_nt_mx(Aq, As, Bq, Bs, C, **kw)

    def bf16():
        ops.set_tuning("gemm_nt_four", 1)
        ops.set_tuning("gemm_nt_variant", -1)
        ops.gemm_nt(A, B, C, **kw)

    legs = {"mx eight-wave": lambda: mx(0, -1), "mx four-wave (plan)": lambda: mx(1, -1), "mx four-wave 256": lambda: mx(1, 9), "mx four-wave 192": lambda: mx(1, 3), "bf16 four-wave": bf16,
            "quantise A": lambda: ops.mx_quantize(A, Aq, As)}
    t = {}
    for _ in range(args.rounds):
        for k, fn in legs.items():
            t.setdefault(k, []).append(time_fn(fn))
    fl = 2.0 * M * N * K
    print(f"{nm:40s} " + " | ".join(f"{k}: {sorted(v)[len(v) // 2]:8.1f} us {fl / sorted(v)[len(v) // 2] / 1e6:6.0f} TF" if not k.startswith("quant") else f"{k}: {sorted(v)[len(v) // 2]:7.1f} us"
                                    for k, v in t.items()), flush=True)
ops.set_tuning("gemm_nt_four", 1)
ops.set_tuning("gemm_nt_variant", -1)
