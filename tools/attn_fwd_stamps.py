#!/usr/bin/env python
"""The encoder attention FORWARD (enc_fwd_c_kernel: one 8-wave workgroup per (passage, head)) under the clock — experiments build:
every wave of every workgroup stamps s_memtime at entry / everything requested / landed + barrier / first query block done / exit and
records its HW_ID + XCC_ID, so the launch can be laid out per CU: how many workgroups a CU holds at a time, how long a workgroup waits for
its K / V images, how long the dispatcher takes to put the next workgroup on a CU.
    LAKO_LIB=lako_amd/liblako_hip_exp.so LAKO_ATTN_DEBUG=262144 python tools/attn_fwd_stamps.py"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LAKO_ATTN_DEBUG", "262144")
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
st = torch.empty(BN, H, L, 4, device=dev)
rel = torch.randn(H, 2 * L - 1, device=dev)
order = torch.argsort(lens, descending=True, stable=True).to(torch.int32).to(dev)
kw = dict(rel_bias=rel, rel_off=L - 1, drop=(0.1, 1, 2), q_off=off, k_off=off, max_q=L, max_k=L, order=order)


def hd(t, c0):
    return t.view(1, M, t.shape[1])[:, :, c0:c0 + inner].unflatten(2, (H, dk))


for _ in range(4):
    ops.attn_fwd(hd(qkv, 0), hd(qkv, inner), hd(qkv, 2 * inner), hd(ctx, 0), st, **kw)
torch.cuda.synchronize()
buf = np.zeros((4096, 8, 8), dtype=np.uint64)
ops.lib.lako_exp_attn_fwd_stamps.restype = ctypes.c_int
rc = ops.lib.lako_exp_attn_fwd_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
assert rc == 0, rc
s = buf[:BN * H].astype(np.int64)                     # [workgroup = item z * H + head][wave][point]
t0 = s[:, :, 0].min()
start, req, land, first, end = (s[:, :, i] - t0 for i in range(5))
hw, xcc = s[:, 0, 5] & 0xFFFF, (s[:, 0, 5] >> 16) & 0xF
chain = s[:, :, 6] - t0
cu = ((xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15))      # (xcc, se, sh, cu)
nblk = s[:, :, 7] & 0xFFFF
Lq = s[:, 0, 7] >> 16
span = end.max() - start.min()
print(f"launch: {BN * H} workgroups, span {span} cycles = {span / 2400:.1f} us at 2.4 GHz; distinct CUs seen: {len(np.unique(cu))}")
wg_start, wg_end = start.min(1), end.max(1)
print("per workgroup (cycles, median / p90):")
for nm, v in (("entry -> sequence id and offsets known (the scalar-load chain)", (chain - start).max(1)),
              ("offsets known -> everything requested AND the bias copies written (DMA + Q + bias issue, one wait for all of it)", (req - chain).max(1)),
              ("requested -> landed + barrier (the global round trip nothing overlaps inside the workgroup)", (land - req).max(1)),
              ("first query block of a wave (compute)", (first - land).max(1)),
              ("barrier -> exit of the LAST wave (compute phase of the workgroup)", wg_end - land.max(1)),
              ("whole workgroup", wg_end - wg_start)):
    print(f"  {nm}: {int(np.median(v))} / {int(np.percentile(v, 90))}")
busy = (end - land)                                     # per wave compute time
print(f"waves: blocks per wave median {np.median(nblk):.0f}, max {nblk.max()}; wave compute time median {int(np.median(busy))}, "
      f"share of the workgroup's compute phase a wave is busy: {float((busy.sum(1) / (8.0 * (wg_end - land.max(1)))).mean()):.2f}")
# per CU: concurrency and hand-over gaps
gaps, conc, cover = [], [], []
for c in np.unique(cu):
    idx = np.where(cu == c)[0]
    o = idx[np.argsort(wg_start[idx])]
    ev = sorted([(wg_start[i], 1) for i in o] + [(wg_end[i], -1) for i in o])
    n, last, acc, idle = 0, ev[0][0], 0.0, 0
    for t, d in ev:
        acc += n * (t - last)
        if n == 0:
            idle += t - last
        last = t
        n += d
    life = ev[-1][0] - ev[0][0]
    conc.append(acc / max(life, 1))
    cover.append((ev[0][0], ev[-1][0], idle, len(o)))
    # hand-over: a workgroup ends -> the next one that starts after it on this CU
    ends = np.sort(wg_end[o])
    starts = np.sort(wg_start[o])
    for e in ends[:-2]:
        nxt = starts[np.searchsorted(starts, e, side="right"):]
        if len(nxt):
            gaps.append(nxt[0] - e)
conc, cover, gaps = np.array(conc), np.array(cover), np.array(gaps)
print(f"per CU: workgroups {cover[:, 3].mean():.1f}, mean concurrency {conc.mean():.2f} workgroups, first start {int(np.median(cover[:, 0]))} (p90 {int(np.percentile(cover[:, 0], 90))}), "
      f"last end {int(np.median(cover[:, 1]))}; cycles with NO workgroup between first start and last end: median {int(np.median(cover[:, 2]))}")
print(f"hand-over on a CU (a workgroup exits -> the next workgroup's entry stamp): median {int(np.median(gaps))}, p90 {int(np.percentile(gaps, 90))} cycles")
h = np.bincount(np.clip(Lq // 16, 0, 13))
print("workgroup duration by query length (cycles): " + "  ".join(f"L~{16 * k}: {int(np.median((wg_end - wg_start)[Lq // 16 == k]))}" for k in range(6, 13) if h[k] > 0))
