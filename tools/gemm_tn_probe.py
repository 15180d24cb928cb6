"""Weight-gradient (TN) launches of one encoder layer at K = a multiple of 64 (so that the global_load_lds instantiation is eligible):
gemm_nt_glds = 0 / 1."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lako_amd.ops import HipOps  # noqa: E402

ops = HipOps()
dev = torch.device("cuda:0")
K = int(os.environ.get("ROWS", "47744"))
d, f = 768, 3072
dy = {n: torch.randn(K, n, device=dev).bfloat16() for n in (768, 2304, 3072)}
x = {n: torch.randn(K, n, device=dev).bfloat16() for n in (768, 3072)}
G = [torch.zeros(768, 3072, device=dev), torch.zeros(3072, 768, device=dev), torch.zeros(768, 768, device=dev), torch.zeros(2304, 768, device=dev)]
items = [(dy[768], x[3072], G[0], 1.0), (dy[3072], x[768], G[1], 1.0), (dy[768], x[768], G[2], 1.0), (dy[2304], x[768], G[3], 1.0)]
fl = 2.0 * K * (768 * 3072 * 2 + 768 * 768 + 2304 * 768)
for glds in (0, 1, 0, 1):
    ops.set_tuning("gemm_nt_glds", glds)
    for _ in range(3):
        ops.gemm_tn_grouped(items)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.gemm_tn_grouped(items)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"grouped dW of one encoder layer, K={K}, glds={glds}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s", flush=True)
