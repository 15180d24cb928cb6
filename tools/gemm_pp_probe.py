"""A/B of the 256² NT kernel's main-loop schedules on the training step's shapes: gemm_nt_pp = 0 (waves of a SIMD in phase), 1 (ping-pong:
waves 4–7 half a K-step behind), 2 / 3 (every wave runs the early / the late role: the roles' code alone, without the phase shift)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lako_amd.ops import HipOps  # noqa: E402

ops = HipOps()
dev = torch.device("cuda:0")
M = int(os.environ.get("ROWS", "47757"))
shapes = [("qkv", M, 2304, 768, {}), ("wi relu", M, 3072, 768, dict(relu=True, drop=(0.1, 1, 2))), ("dxn K=3072", M, 768, 3072, {}),
          ("8192^3", 8192, 8192, 8192, {})]
modes = [int(x) for x in os.environ.get("MODES", "0,1,2,3").split(",")]
for name, m, n, k, kw in shapes:
    A = torch.randn(m, k, device=dev).bfloat16()
    B = torch.randn(n, k, device=dev).bfloat16()
    C = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    for mode in modes:
        ops.set_tuning("gemm_nt_pp", mode)
        for _ in range(3):
            ops.gemm_nt(A, B, C, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.gemm_nt(A, B, C, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        print(f"{name:12s} [{m},{k}]x[{n},{k}] pp={mode}: {us:9.1f} us  {2.0 * m * n * k / us / 1e6:8.1f} TFLOP/s", flush=True)
