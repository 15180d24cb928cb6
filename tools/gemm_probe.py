#!/usr/bin/env python
"""Tiny driver for rocprofv3 --pmc runs: three calls of each encoder GEMM shape of a config-2 training step, in a fixed order —
the five NT shapes (forward + dX products of an encoder layer, plain epilogues) and the layer's four weight gradients as the ONE
grouped TN launch the engine issues.  tools/gemm_traffic.py attributes the dispatches of a pass to these calls by order.

    LAKO_PROBE_TOKENS=48000 python tools/gemm_probe.py      rows: 64000 = padded config 2, ≈48000 = its valid tokens"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lako_amd.ops import HipOps  # noqa: E402

CALLS = 3
d, f, inner = 768, 3072, 768
# (N, K, launches of this shape per encoder layer and training step)
NT_SHAPES = [(3 * inner, d, 1), (f, d, 2), (d, f, 2), (d, inner, 2), (d, 3 * inner, 1)]
TN_GROUP = [(d, f), (f, d), (d, inner), (3 * inner, d)]      # (M, N) of dWo2, dWi, dWo, dWqkv: K = tokens


def main():
    ops = HipOps()
    dev = torch.device("cuda:0")
    Me = int(os.environ.get("LAKO_PROBE_TOKENS", "64000"))
    T = torch.bfloat16
    for (N, K, _) in NT_SHAPES:
        A = torch.randn(Me, K, device=dev).to(T)
        B = torch.randn(N, K, device=dev).to(T)
        C = torch.empty(Me, N, dtype=T, device=dev)
        for _ in range(CALLS):
            ops.gemm_nt(A, B, C)
        torch.cuda.synchronize()
        del A, B, C
    items = []
    for (M, N) in TN_GROUP:
        A = torch.randn(Me, M, device=dev).to(T)
        B = torch.randn(Me, N, device=dev).to(T)
        items.append((A, B, torch.zeros(M, N, device=dev), 1.0))
    for _ in range(CALLS):
        ops.gemm_tn_grouped(items)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
