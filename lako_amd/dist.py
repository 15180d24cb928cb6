"""Data-parallel gradient synchronisation (absent in the reference, which trains on one GPU —
train_reader.py never wraps the model in DDP).

One process per GPU; `torch.distributed` backend "nccl" is RCCL on ROCm and rides xGMI inside a node.
The engine lays gradients out in the order they complete during backward and reports finished ranges
through `Engine.grad_hook`; each range becomes one asynchronous SUM all-reduce issued while the rest of
backward is still running (decoder + cross-K/V bucket first, then one bucket per encoder layer, the
tied embedding last).  `finish()` makes the compute stream wait for all of them; the 1/world factor is
folded into the fused optimizer step (lako_adamw_step grad_scale), so gradients are never rescaled in
a separate pass.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, model, group=None, bucket_bytes: int = 64 << 20, force: bool = False):
        self.force = force          # issue the collectives even at world size 1 (single-GPU test of the RCCL path)
        self.model = model
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.bucket_bytes = bucket_bytes
        self.handles = []
        self._pending = None
        eng = model._get_engine()
        eng.grad_hook = self._on_ready
        model._grad_sync = self

    def _flush(self):
        if self._pending is None:
            return
        lo, hi = self._pending
        self._pending = None
        g = self.model._engine.G[lo:hi]
        self.handles.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _on_ready(self, lo: int, hi: int):
        if self.world_size == 1 and not self.force:
            return
        if self._pending is not None and self._pending[1] == lo:
            self._pending = (self._pending[0], hi)          # contiguous with the previous range: coalesce
        else:
            self._flush()
            self._pending = (lo, hi)
        if (self._pending[1] - self._pending[0]) * 4 >= self.bucket_bytes:
            self._flush()

    def finish(self):
        self._flush()
        for h in self.handles:
            h.wait()
        self.handles = []


def broadcast_parameters(model, src: int = 0, group=None):
    """Make every rank start from rank `src`'s weights."""
    eng = model._get_engine()
    dist.broadcast(eng.P, src, group=group)
    eng.shadows_stale = True
