#!/usr/bin/env python
"""Tiny driver for rocprofv3 --pmc runs: a few launches of the config-2 GEMM shapes (NT and TN)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lako_amd.ops import HipOps  # noqa: E402

ops = HipOps()
dev = torch.device("cuda:0")
Me = int(os.environ.get("LAKO_PROBE_TOKENS", "64000"))     # 64000 = padded config 2; ≈48000 = its valid tokens (unpadded path)
d, f, inner = 768, 3072, 768
T = torch.bfloat16
for (M, N, K) in [(Me, 3 * inner, d), (Me, f, d), (Me, d, f), (Me, d, inner)]:
    A = torch.randn(M, K, device=dev).to(T)
    B = torch.randn(N, K, device=dev).to(T)
    C = torch.empty(M, N, dtype=T, device=dev)
    for _ in range(3):
        ops.gemm_nt(A, B, C)
    torch.cuda.synchronize()
    del A, B, C
for (K, M, N) in [(Me, 3 * inner, d), (Me, f, d)]:
    A = torch.randn(K, M, device=dev).to(T)
    B = torch.randn(K, N, device=dev).to(T)
    C = torch.zeros(M, N, device=dev)
    for _ in range(3):
        ops.gemm_tn(A, B, C)
    torch.cuda.synchronize()
