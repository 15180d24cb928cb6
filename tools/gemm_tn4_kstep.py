import os, sys, torch
sys.path.insert(0, "/root/repo")
from lako_amd.ops import HipOps
dev = torch.device("cuda:0"); ops = HipOps()
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for K in (2048, 8192, 32768):
    A = (torch.randn(K, 8192, device=dev) * 0.3).bfloat16(); B = (torch.randn(K, 8192, device=dev) * 0.3).bfloat16(); C = torch.zeros(8192, 8192, device=dev)
    An = (torch.randn(8192, K, device=dev) * 0.3).bfloat16(); Bn = (torch.randn(8192, K, device=dev) * 0.3).bfloat16(); Cn = torch.empty(8192, 8192, dtype=torch.bfloat16, device=dev)
    for four in (1, 0):
        ops.set_tuning("gemm_tn_four", four)
        us = sorted(t(lambda: ops.gemm_tn(A, B, C, split_k=-2)) for _ in range(3))[1]
        print(f"TN K {K} four {four}: {us:9.1f} us = {us / 4 / (K // 64):.3f} us per K-step and round, {2.0 * 8192 * 8192 * K / us / 1e6:.0f} TF")
    ops.set_tuning("gemm_nt_variant", 9)
    us = sorted(t(lambda: ops.gemm_nt(An, Bn, Cn)) for _ in range(3))[1]
    print(f"NT K {K}       : {us:9.1f} us = {us / 4 / (K // 64):.3f} us per K-step and round, {2.0 * 8192 * 8192 * K / us / 1e6:.0f} TF")
