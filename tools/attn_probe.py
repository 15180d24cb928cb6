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
km = (torch.arange(L, device=dev)[None] < lens[:, None]).to(torch.uint8)
kw = dict(rel_bias=rel, rel_off=L - 1, key_mask=km, drop=(0.1, 1, 2))
dctx = torch.randn(BN * L, inner, device=dev).to(T)
dqkv = torch.empty(BN * L, 3 * inner, dtype=T, device=dev)
drel = torch.zeros_like(rel)
for _ in range(3):
    ops.attn_fwd(heads(qkv, 0), heads(qkv, inner), heads(qkv, 2 * inner), heads(ctx, 0), st, **kw)
    ops.attn_bwd(heads(qkv, 0), heads(qkv, inner), heads(qkv, 2 * inner), heads(ctx, 0), heads(dctx, 0), st,
                 heads(dqkv, 0), heads(dqkv, inner), heads(dqkv, 2 * inner), drel=drel, **kw)
torch.cuda.synchronize()
