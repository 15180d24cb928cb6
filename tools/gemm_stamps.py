#!/usr/bin/env python
"""Where one workgroup of the dominant NT GEMM spends its K-steps (experiments build: gemm_nt_debug bit 6 = s_memtime stamps of
workgroup 0, all 8 waves).  VERDICT round 4, item 2: "the s_memtime breakdown of one workgroup's K-loop (barrier wait vs vmcnt wait vs
MFMA issue)".
    LAKO_LIB=lako_amd/liblako_hip_exp.so python tools/gemm_stamps.py [M N K]      (default: the five encoder shapes at 47 757 rows)
Stamp points of a K-step (64 k): 0 top · 1 after this wave's LDS-DMA issue (waves 0-3; waves 4-7 issue after the first MFMA rows) ·
2 end of K-half 0 (its fragment reads + MFMAs) · 3 before the DMA wait of K-half 1 · 4 after `s_waitcnt vmcnt(0)` · 5 after the barrier ·
6 end of the step.  Segments printed per wave: issue = 0→1, half0 = 1→2, half1a = 2→3, vmwait = 3→4, barrier = 4→5, tail = 5→6."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("LAKO_LIB", os.path.join(ROOT, "lako_amd", "liblako_hip_exp.so"))
from lako_amd.ops import HipOps  # noqa: E402

ops = HipOps()
dev = torch.device("cuda:0")
T = torch.bfloat16
shapes = [(47757, 2304, 768), (47757, 3072, 768), (47757, 768, 3072), (47757, 768, 768), (47757, 768, 2304)]
if len(sys.argv) == 4:
    shapes = [tuple(int(x) for x in sys.argv[1:4])]
for M, N, K in shapes:
    A = (torch.randn(M, K, device=dev)).to(T)
    B = (torch.randn(N, K, device=dev)).to(T)
    C = torch.empty(M, N, dtype=T, device=dev)
    ops.set_tuning("gemm_nt_debug", 0)
    for _ in range(3):
        ops.gemm_nt(A, B, C)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.gemm_nt(A, B, C)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    ops.set_tuning("gemm_nt_debug", 64)
    ops.gemm_nt(A, B, C)
    torch.cuda.synchronize()
    buf = np.zeros((8, 128, 8), dtype=np.uint64)
    rc = ops.lib.lako_exp_nt_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
    assert rc == 0
    ops.set_tuning("gemm_nt_debug", 0)
    s = buf.astype(np.int64)
    nst = int((s[0, :, 0] > 0).sum())
    nk = (K * 2 + 127) // 128
    print(f"=== [{M},{K}] x [{N},{K}]: {us:.1f} us = {2.0 * M * N * K / us / 1e6:.0f} TFLOP/s; {nk} K-steps per tile, {nst} stamped K-steps of workgroup 0")
    names = ["issue", "half0", "half1a", "vmwait", "barrier", "tail"]
    # steady state: K-steps 2 .. nk-2 of the first two tiles
    sel = [k for k in range(nst) if 1 <= (k % nk) < nk - 1][: 4 * nk]
    for w in range(8):
        seg = np.array([[s[w, k, p + 1] - s[w, k, p] for p in range(6)] for k in sel])
        step = np.array([s[w, k + 1, 0] - s[w, k, 0] for k in sel if k + 1 < nst and (k + 1) % nk != 0])
        print(f"  wave {w}: " + "  ".join(f"{n} {int(np.median(seg[:, i])):5d}" for i, n in enumerate(names)) + f"   | K-step {int(np.median(step)):5d} cycles")
    # tile boundaries: last stamp of a tile's final K-step to the first stamp of the next tile
    gaps = [int(s[0, k + 1, 0] - s[0, k, 6]) for k in range(nst - 1) if (k + 1) % nk == 0]
    first = [int(s[0, k, 6] - s[0, k, 0]) for k in range(nst) if k % nk == 0]
    last = [int(s[0, k, 6] - s[0, k, 0]) for k in range(nst) if k % nk == nk - 1]
    print(f"  wave 0: epilogue + tile switch (end of a tile's last K-step -> top of the next tile's first) {gaps[:6]} cycles; first K-step of a tile {first[:6]}; last {last[:6]}")
