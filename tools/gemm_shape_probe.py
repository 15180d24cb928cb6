#!/usr/bin/env python
"""One NT GEMM shape, a few launches (for rocprofv3 --pmc / --kernel-trace A/B runs under LAKO_TUNING): SHAPE=M,N,K."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lako_amd.ops import HipOps  # noqa: E402

ops = HipOps()
dev = torch.device("cuda:0")
M, N, K = (int(v) for v in os.environ.get("SHAPE", "48000,3072,768").split(","))
A = torch.randn(M, K, device=dev).to(torch.bfloat16)
B = torch.randn(N, K, device=dev).to(torch.bfloat16)
C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
for _ in range(6):
    ops.gemm_nt(A, B, C)
torch.cuda.synchronize()
