#!/usr/bin/env python
"""Tiny driver for rocprofv3 --pmc runs: a few launches of the config-2 encoder self-attention (forward, backward with
bias + mask + dropout + bias gradient)."""
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
qkv = (torch.randn(BN * L, 3 * inner, device=dev) * 0.5).to(T)
ctx = torch.empty(BN * L, inner, dtype=T, device=dev)


def heads(t, c0):
    return t.view(BN, L, t.shape[1])[:, :, c0:c0 + inner].unflatten(2, (H, dk))


st = torch.empty(BN, H, L, 4, device=dev)
rel = torch.randn(H, 2 * L - 1, device=dev)
lens = torch.randint(L // 2, L + 1, (BN,), device=dev)
dctx = torch.randn(BN * L, inner, device=dev).to(T)
dqkv = torch.empty(BN * L, 3 * inner, dtype=T, device=dev)
drel = torch.zeros_like(rel)
if os.environ.get("LAKO_PROBE_PADDED") == "1":     # the padded [BN, L] layout with a key mask (LAKO_UNPAD=0)
    km = (torch.arange(L, device=dev)[None] < lens[:, None]).to(torch.uint8)
    kw = dict(rel_bias=rel, rel_off=L - 1, key_mask=km, drop=(0.1, 1, 2))
    hd = heads
else:                                              # what the training step runs: packed valid tokens + passage offsets
    off = torch.zeros(BN + 1, dtype=torch.int32, device=dev)
    off[1:] = torch.cumsum(lens, 0)
    M = int(off[-1])
    kw = dict(rel_bias=rel, rel_off=L - 1, drop=(0.1, 1, 2), q_off=off, k_off=off, max_q=L, max_k=L)

    def hd(t, c0):
        return t[:M].view(1, M, t.shape[1])[:, :, c0:c0 + inner].unflatten(2, (H, dk))
for _ in range(3):
    ops.attn_fwd(hd(qkv, 0), hd(qkv, inner), hd(qkv, 2 * inner), hd(ctx, 0), st, **kw)
    ops.attn_bwd(hd(qkv, 0), hd(qkv, inner), hd(qkv, 2 * inner), hd(ctx, 0), hd(dctx, 0), st,
                 hd(dqkv, 0), hd(dqkv, inner), hd(dqkv, 2 * inner), drel=drel, **kw)
torch.cuda.synchronize()
