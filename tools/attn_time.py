#!/usr/bin/env python
"""Times the encoder self-attention launches of config 2 (ragged, bias, dropout) with HIP events: forward, backward."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lako_amd.ops import HipOps  # noqa: E402

ops = HipOps()
dev = torch.device("cuda:0")
BN, L, H, dk = 320, 200, 12, 64
inner = H * dk
T = torch.bfloat16
torch.manual_seed(0)
g = torch.Generator().manual_seed(1)
lens = torch.randint(L // 2, L + 1, (BN,), generator=g)
if os.environ.get("ALL_VALID") == "1":
    lens[:] = L
off = torch.zeros(BN + 1, dtype=torch.int32)
off[1:] = torch.cumsum(lens, 0)
M = int(off[-1])
off = off.to(dev)
qkv = (torch.randn(M, 3 * inner, device=dev) * 0.5).to(T)
ctx = torch.empty(M, inner, dtype=T, device=dev)
dctx = torch.randn(M, inner, device=dev).to(T)
dqkv = torch.empty(M, 3 * inner, dtype=T, device=dev)
st = torch.empty(BN, H, L, 4, device=dev)
rel = torch.randn(H, 2 * L - 1, device=dev)
drel = torch.zeros_like(rel)
drop = (0.1, 1, 2) if os.environ.get("NODROP") != "1" else None
kw = dict(rel_bias=rel, rel_off=L - 1, drop=drop, q_off=off, k_off=off, max_q=L, max_k=L)


def hd(t, c0):
    return t.view(1, M, t.shape[1])[:, :, c0:c0 + inner].unflatten(2, (H, dk))


def fwd():
    ops.attn_fwd(hd(qkv, 0), hd(qkv, inner), hd(qkv, 2 * inner), hd(ctx, 0), st, **kw)


def bwd():
    ops.attn_bwd(hd(qkv, 0), hd(qkv, inner), hd(qkv, 2 * inner), hd(ctx, 0), hd(dctx, 0), st,
                 hd(dqkv, 0), hd(dqkv, inner), hd(dqkv, 2 * inner), drel=drel, **kw)


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


pairs = float(((lens + 15) // 16 * ((lens + 31) // 32)).sum()) * H        # (16 queries × 32 keys) units
print(f"tokens {M}  fwd {timeit(fwd):8.1f} us   bwd {timeit(bwd):8.1f} us   ({pairs:.0f} 16x32 units, tag {os.environ.get('TAG', '')})")
