#!/usr/bin/env python
"""Tile HEIGHT of the 256-column NT kernel on the step's plain-epilogue shapes at an unpadded batch's row count (ROWS, default 47757):
variants auto (-1: 256-row tiles + tail split), 2 (256-row tiles, no plan), 7 (192-row), 8 (288-row), interleaved in one process,
median of REPS rounds of ITERS launches each.  Prints µs per call and TFLOP/s."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lako_amd.ops import HipOps  # noqa: E402

ops = HipOps()
dev = torch.device("cuda:0")
ROWS = int(os.environ.get("ROWS", "47757"))
ITERS, REPS = int(os.environ.get("ITERS", "10")), int(os.environ.get("REPS", "5"))
VARIANTS = [int(v) for v in os.environ.get("VARIANTS", "-1,2,7,8").split(",")]
drop = (0.1, 1, 2)
shapes = [("qkv   [M,768]x[2304,768]", 2304, 768, {}),
          ("wi    [M,768]x[3072,768] relu+drop", 3072, 768, dict(relu=True, drop=drop)),
          ("dctx  [M,768]x[768,768]", 768, 768, {}),
          ("dxn   [M,2304]x[768,2304]", 768, 2304, {}),
          ("dxn   [M,3072]x[768,3072]", 768, 3072, {}),
          ("o+res [M,768]x[768,768] res+drop", 768, 768, dict(resid=True, drop=drop)),
          ("wo+res[M,3072]x[768,3072] res+drop", 768, 3072, dict(resid=True, drop=drop)),
          ("dpre  [M,768]x[3072,768] auxmask", 3072, 768, dict(aux=True)),
          ("8192^3", 8192, 8192, {})]
if os.environ.get("ONLY"):
    shapes = [sh for sh in shapes if any(k in sh[0] for k in os.environ["ONLY"].split(","))]
for nm, N, K, kw in shapes:
    M = 8192 if nm.startswith("8192") else ROWS
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    B = torch.randn(N, K, device=dev).to(torch.bfloat16)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    kw = dict(kw)
    if kw.pop("resid", False):
        kw["resid"] = torch.randn(M, N, device=dev).to(torch.bfloat16)
    if kw.pop("aux", False):
        kw["aux"], kw["aux_scale"] = torch.randn(M, N, device=dev).to(torch.bfloat16), 1.1
    res = {v: [] for v in VARIANTS}
    for rep in range(REPS + 1):
        for v in VARIANTS:
            ops.set_tuning("gemm_nt_variant", v)
            ops.gemm_nt(A, B, C, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(ITERS):
                ops.gemm_nt(A, B, C, **kw)
            e1.record()
            torch.cuda.synchronize()
            if rep:
                res[v].append(e0.elapsed_time(e1) * 1e3 / ITERS)
    line = "  ".join(f"v{v}: {statistics.median(res[v]):7.1f} us {2.0 * M * N * K / statistics.median(res[v]) / 1e6:6.0f} TF" for v in VARIANTS)
    print(f"{nm:38s} {line}", flush=True)
    del A, B, C
ops.set_tuning("gemm_nt_variant", -1)
