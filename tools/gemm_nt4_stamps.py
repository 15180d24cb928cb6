#!/usr/bin/env python
"""s_memtime stamps of workgroup 0 of the four-wave NT kernel (experiments build, gemm_nt_debug bit 6): cycles per K-step along the
workgroup's tiles, the epilogue and the gaps between them.   LAKO_LIB=lako_amd/liblako_hip_exp.so python tools/gemm_nt4_stamps.py M N K [variant]"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("LAKO_LIB", os.path.join(ROOT, "lako_amd", "liblako_hip_exp.so"))
from lako_amd.ops import HipOps
ops = HipOps(); dev = torch.device("cuda:0")
M, N, K = (int(x) for x in sys.argv[1:4]); v = int(sys.argv[4]) if len(sys.argv) > 4 else 9
extra = int(sys.argv[5]) if len(sys.argv) > 5 else 0
A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16(); C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
ops.set_tuning("gemm_nt_variant", v)
for _ in range(3): ops.gemm_nt(A, B, C)
ops.set_tuning("gemm_nt_debug", 64 | extra)
ops.gemm_nt(A, B, C); torch.cuda.synchronize()
buf = np.zeros((8, 128, 8), dtype=np.uint64)
assert ops.lib.lako_exp_nt_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes)) == 0
s = buf.astype(np.int64); nk = K // 64
for w in range(4):
    top = s[w, :, 0]; n = int((top > 0).sum())
    d = [int(top[k + 1] - top[k]) for k in range(n - 1)]
    print(f"wave {w}: {n} stamped K-steps, nk {nk}")
    for t0 in range(0, n - 1, nk):
        row = d[t0:t0 + nk]
        ep = int(s[w, min(t0 + nk, 127), 3] - s[w, min(t0 + nk, 127), 2]) if t0 + nk < 128 else -1
        print(f"   tile {t0 // nk}: K-steps " + " ".join(f"{x:5d}" for x in row[:nk - 1]) + f" | last step + epilogue + switch {row[nk - 1] if len(row) == nk else -1:6d} (epilogue {ep})")
    tiles = [k for k in range(0, n, nk)]
    if len(tiles) > 1:
        dc = s[w, tiles[-1], 1] - s[w, tiles[0], 1]; dr = s[w, tiles[-1], 4] - s[w, tiles[0], 4]
        print(f"   clock over tiles 0..{len(tiles) - 1}: {dc} cycles in {dr * 10} ns = {dc / (dr * 10):.3f} GHz; {dr * 10 / 1000 / (len(tiles) - 1):.2f} us per tile")
