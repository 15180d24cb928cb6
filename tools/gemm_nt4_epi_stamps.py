#!/usr/bin/env python
"""Epilogue duration (s_memtime cycles, workgroup 0, wave 0) of the four-wave NT kernel per epilogue kind — experiments build."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("LAKO_LIB", os.path.join(ROOT, "lako_amd", "liblako_hip_exp.so"))
from lako_amd.ops import HipOps
ops = HipOps(); dev = torch.device("cuda:0")
M, N, K = 47757, 3072, 768
A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16(); C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
side = torch.randn(M, N, device=dev).bfloat16()
ops.set_tuning("gemm_nt_variant", 9)
for nm, kw in [("plain", {}), ("alpha 0.5", dict(alpha=0.5)), ("relu", dict(relu=True)), ("relu+drop", dict(relu=True, drop=(0.1, 1, 2))), ("drop", dict(drop=(0.1, 1, 2))),
               ("resid", dict(resid=side)), ("res+drop", dict(resid=side, drop=(0.1, 1, 2))), ("aux", dict(aux=side, aux_scale=1.1))]:
    ops.set_tuning("gemm_nt_debug", 0)
    for _ in range(2): ops.gemm_nt(A, B, C, **kw)
    ops.set_tuning("gemm_nt_debug", 64)
    ops.gemm_nt(A, B, C, **kw); torch.cuda.synchronize()
    buf = np.zeros((8, 128, 8), dtype=np.uint64)
    assert ops.lib.lako_exp_nt_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes)) == 0
    s = buf.astype(np.int64); nk = K // 64
    eps = [int(s[0, k, 3] - s[0, k, 2]) for k in range(nk, 9 * nk, nk)]
    steps = [int(s[0, k + 1, 0] - s[0, k, 0]) for k in range(2, 10)]
    print(f"{nm:10s} epilogue cycles per tile {eps}   K-steps {steps[:4]}")
