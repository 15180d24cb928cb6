#!/usr/bin/env python
"""Where one workgroup of the one-pass attention backward spends its cycles (experiments build: s_memtime stamps of workgroup 0).
    LAKO_LIB=lako_amd/liblako_hip_exp.so LAKO_ATTN_DEBUG=131072 python tools/attn_stamps.py
Stamp points per iteration — P1 waves (0-13): 0 before the barrier, 1 after it, 2 / 4 after the score MFMAs of q-tile 0 / 1,
3 / 5 after the tile's vector work, 6 after the dK / dV products, 7 end; waves 14 / 15: 0, 1, 2 after the delta stage (15), 3 after P2,
4 after the DMA issue (14)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LAKO_ATTN_DEBUG", "131072")
os.environ.setdefault("LAKO_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lako_amd", "liblako_hip_exp.so"))
from lako_amd.ops import HipOps  # noqa: E402

ops = HipOps()
dev = torch.device("cuda:0")
BN, L, H, dk = 320, 200, 12, 64
inner = H * dk
T = torch.bfloat16
g = torch.Generator().manual_seed(1)
lens = torch.randint(L // 2, L + 1, (BN,), generator=g)
off = torch.zeros(BN + 1, dtype=torch.int32)
off[1:] = torch.cumsum(lens, 0)
M = int(off[-1])
off = off.to(dev)
torch.manual_seed(0)
qkv = (torch.randn(M, 3 * inner, device=dev) * 0.5).to(T)
ctx = torch.empty(M, inner, dtype=T, device=dev)
dctx = torch.randn(M, inner, device=dev).to(T)
dqkv = torch.empty(M, 3 * inner, dtype=T, device=dev)
st = torch.empty(BN, H, L, 4, device=dev)
rel = torch.randn(H, 2 * L - 1, device=dev)
drel = torch.zeros_like(rel)
order = torch.argsort(lens, descending=True, stable=True).to(torch.int32).to(dev)
kw = dict(rel_bias=rel, rel_off=L - 1, drop=(0.1, 1, 2), q_off=off, k_off=off, max_q=L, max_k=L, order=order)


def hd(t, c0):
    return t.view(1, M, t.shape[1])[:, :, c0:c0 + inner].unflatten(2, (H, dk))


ops.attn_fwd(hd(qkv, 0), hd(qkv, inner), hd(qkv, 2 * inner), hd(ctx, 0), st, **kw)
for _ in range(3):
    ops.attn_bwd(hd(qkv, 0), hd(qkv, inner), hd(qkv, 2 * inner), hd(ctx, 0), hd(dctx, 0), st,
                 hd(dqkv, 0), hd(dqkv, inner), hd(dqkv, 2 * inner), drel=drel, **kw)
torch.cuda.synchronize()
buf = np.zeros((16, 256, 8), dtype=np.uint64)
rc = ops.lib.lako_exp_attn_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
assert rc == 0, rc
s = buf.astype(np.int64)
nit = int((s[0, :, 0] > 0).sum())
print(f"iterations stamped: {nit}; items of workgroup 0 by length: {[int(lens[int(order[i])]) for i in range(0, BN, 21)]}")
t0 = s[0, 0, 0]
print("per-iteration spans in cycles (wave 0 / wave 5 / wave 13 | wave 14 | wave 15); columns: barrier wait, then the segments between stamp points")
for n in range(min(nit, 40)):
    row = []
    for w in (0, 5, 13):
        v = s[w, n]
        seg = [v[1] - v[0]] + [v[k + 1] - v[k] if v[k + 1] > 0 and v[k] > 0 else -1 for k in range(1, 7)]
        row.append(" ".join(f"{int(x):5d}" for x in seg))
    for w in (14, 15):
        v = s[w, n]
        seg = [v[1] - v[0]] + [v[k + 1] - v[k] if v[k + 1] > 0 and v[k] > 0 else -1 for k in range(1, 4)]
        row.append(" ".join(f"{int(x):5d}" for x in seg))
    nxt = s[0, n + 1, 0] - s[0, n, 0] if n + 1 < nit else 0
    print(f"n={n:3d} iter {int(nxt):6d} | " + " | ".join(row))
tot = s[0, nit - 1, 7] - t0
print(f"total {int(tot)} cycles over {nit} iterations = {tot / max(nit, 1):.0f} per iteration")
for n in (3, 4, 10, 40):
    if n < nit:
        w_ = [int(s[w, n, 7] - s[w, n, 1]) if w < 14 else int(s[w, n, 4] - s[w, n, 1]) for w in range(16)]
        b_ = [int(s[w, n, 1] - s[w, n, 0]) for w in range(16)]
        print(f"n={n}: work per wave after the barrier {w_}")
        print(f"      wait at the barrier       {b_}")
        print(f"      barrier release skew (stamp 1 minus wave 0's) {[int(s[w, n, 1] - s[0, n, 1]) for w in range(16)]}")
k = s[0, 255]
print(f"wave 0: kernel entry -> loop {int(k[1] - k[0])} cycles, loop {int(k[2] - k[1])}, bias-gradient flush {int(k[3] - k[2])}; whole kernel {int(k[3] - k[0])}")
