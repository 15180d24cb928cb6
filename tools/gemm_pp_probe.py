"""A/B of the 256² NT kernel's main loops on the training step's shapes: gemm_nt_pp = 0 (the two-phase loop of rounds 1–3) and 1 (the 8-phase loop)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lako_amd.ops import HipOps  # noqa: E402

ops = HipOps()
dev = torch.device("cuda:0")
M = int(os.environ.get("ROWS", "47757"))
shapes = [("qkv", M, 2304, 768, {}), ("wi relu", M, 3072, 768, dict(relu=True, drop=(0.1, 1, 2))), ("dxn K=3072", M, 768, 3072, {}),
          ("dctx K=768", M, 768, 768, {}), ("o+res K=768", M, 768, 768, dict(resid=True, drop=(0.1, 3, 4))),
          ("dpre aux", M, 3072, 768, dict(aux=True)), ("4096^3", 4096, 4096, 4096, {}),
          ("8192^3", 8192, 8192, 8192, {})]
modes = [int(x) for x in os.environ.get("MODES", "0,1,2,3").split(",")]
for name, m, n, k, kw in shapes:
    A = torch.randn(m, k, device=dev).bfloat16()
    B = torch.randn(n, k, device=dev).bfloat16()
    C = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    side = torch.randn(m, n, device=dev).bfloat16() if (kw.get("resid") is True or kw.get("aux") is True) else None
    kw = {k: (side if v is True and k in ("resid", "aux") else v) for k, v in kw.items()}
    for mode in modes:                     # mode = pp + 2·glds
        ops.set_tuning("gemm_nt_pp", mode & 1)
        ops.set_tuning("gemm_nt_glds", mode >> 1)
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
        print(f"{name:12s} [{m},{k}]x[{n},{k}] pp={mode & 1} glds={mode >> 1}: {us:9.1f} us  {2.0 * m * n * k / us / 1e6:8.1f} TFLOP/s", flush=True)
