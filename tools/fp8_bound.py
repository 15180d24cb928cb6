#!/usr/bin/env python
"""Upper bound of what MX-fp8 GEMMs can buy at BASELINE config 5 (T5-large, 100 passages x 200 tokens, batch 8: ≈119 k valid
tokens): every NT product of an encoder layer — the four forward GEMMs and the four dX GEMMs — timed in bf16 (lako_gemm_nt) and in
block-scaled fp8 (lako_gemm_nt_mx on operands quantised BEFOREHAND, i.e. with a free quantiser), plus the quantiser itself.
Σ (bf16 − fp8) x 24 layers against the measured config-5 step is the most a complete fp8 forward + dX path could gain.

    python tools/fp8_bound.py [rows]        default rows = 119360"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lako_amd.ops import HipOps  # noqa: E402

ops = HipOps()
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 119360
d, f, inner = 1024, 4096, 1024
BF = torch.bfloat16


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


shapes = [("fwd qkv", 3 * inner, d), ("fwd o", d, inner), ("fwd wi", f, d), ("fwd wo", d, f),
          ("dX qkv", d, 3 * inner), ("dX o", inner, d), ("dX wi (dxn)", d, f), ("dX wo (dpre)", f, d)]
tot_bf, tot_f8, tot_q = 0.0, 0.0, 0.0
for name, N, K in shapes:
    A = (torch.randn(M, K, device=dev) * 0.5).to(BF)
    B = (torch.randn(N, K, device=dev) * 0.05).to(BF)
    C = torch.empty(M, N, dtype=BF, device=dev)
    Aq, Bq = torch.empty(M, K, dtype=torch.uint8, device=dev), torch.empty(N, K, dtype=torch.uint8, device=dev)
    As = torch.zeros(M, ops.mx_scale_cols(K), dtype=torch.uint8, device=dev)
    Bs = torch.zeros(N, ops.mx_scale_cols(K), dtype=torch.uint8, device=dev)
    ops.mx_quantize(A, Aq, As)
    ops.mx_quantize(B, Bq, Bs)
    t_bf = timeit(lambda: ops.gemm_nt(A, B, C))
    t_f8 = timeit(lambda: ops.gemm_nt_mx(Aq, As, Bq, Bs, C))
    t_q = timeit(lambda: ops.mx_quantize(A, Aq, As))
    fl = 2.0 * M * N * K
    print(f"{name:14s} [{M},{K}]x[{N},{K}]  bf16 {t_bf:8.1f} us ({fl / t_bf / 1e6:7.1f} TF/s)   fp8 {t_f8:8.1f} us ({fl / t_f8 / 1e6:7.1f} TF/s)"
          f"   quantise A {t_q:7.1f} us", flush=True)
    tot_bf += t_bf
    tot_f8 += t_f8
    tot_q += t_q
    del A, B, C, Aq, Bq, As, Bs
print(f"per layer: bf16 {tot_bf:.0f} us, fp8 {tot_f8:.0f} us (free quantiser), quantiser passes {tot_q:.0f} us")
print(f"24 layers: fp8 saves at most {(tot_bf - tot_f8) * 24 / 1e3:.1f} ms per step with a free quantiser, "
      f"{(tot_bf - tot_f8 - tot_q) * 24 / 1e3:.1f} ms with one separate quantiser pass per GEMM")
