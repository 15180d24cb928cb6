"""Data-parallel gradient synchronisation (absent in the reference, which trains on one GPU —
train_reader.py never wraps the model in DDP).

One process per GPU; `torch.distributed` backend "nccl" is RCCL on ROCm and rides xGMI inside a node.
The engine lays gradients out in ONE flat fp32 buffer, in the order they complete during backward, and
reports finished ranges through `Engine.grad_hook`.  Two modes:

  "deferred" (default)  one SUM all-reduce of the whole flat buffer after backward.  892 MB at T5-base:
                        ≈4.5 ms on 8 GPUs (≈14 ms on 2: one xGMI link per pair) against a ≈48 ms step.  Chosen as the default because the
                        big GEMM kernels fill the chip with exactly one workgroup per CU (128 KiB LDS, ≈240
                        VGPRs): RCCL kernels running concurrently on their own stream would take CUs away and
                        stretch every such kernel by a whole scheduling round.
  "overlap"             one asynchronous all-reduce per finished range (decoder + cross-K/V first, then per
                        encoder layer, coalesced to `bucket_bytes`), issued while backward is still running;
                        `LAKO_DP_MODE=overlap`.  The persistent GEMM launches then draw their tiles from per-XCD ticket
                        counters (`gemm_nt_queue`), so the CUs the RCCL kernels hold cost a share of the tiles, not a whole
                        extra pass.  Correct (2-rank gloo test, world-1 RCCL test) — to be measured on 8 GPUs (tools/scale.sh).

`LAKO_DP_GRAD_DTYPE=bf16` (or `grad_dtype=torch.bfloat16`): the gradients travel as bf16 — half the bytes over the xGMI links (446 MB
instead of 892 MB at T5-base; it is the 2- and 4-GPU runs, with one link per peer, that pay most for the collective) — through a
bf16 staging buffer: cast → all-reduce → cast back, two extra passes over the flat buffer (≈0.5 ms).  The sum of world-size bf16
gradients carries one more rounding than the single-GPU path has; the default stays fp32.

`finish()` makes the compute stream wait for the collectives; the 1/world factor is folded into the fused
optimizer step (lako_adamw_step grad_scale) and into the clip norm, so gradients are never rescaled in a
separate pass.
"""
from __future__ import annotations

import os

import torch.distributed as dist


class GradSync:
    def __init__(self, model, group=None, bucket_bytes: int = 64 << 20, force: bool = False, mode: str | None = None,
                 grad_dtype=None):
        self.force = force          # issue the collectives even at world size 1 (single-GPU test of the RCCL path)
        self.model = model
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.bucket_bytes = bucket_bytes
        self.mode = mode or os.environ.get("LAKO_DP_MODE", "deferred")
        if self.mode not in ("deferred", "overlap"):
            raise ValueError(f"unknown DP mode {self.mode!r}")
        self.handles = []
        self._pending = None
        self._dirty = False
        if grad_dtype is None:
            grad_dtype = {"bf16": __import__("torch").bfloat16, "fp32": None, "f32": None}[os.environ.get("LAKO_DP_GRAD_DTYPE", "fp32")]
        self.grad_dtype = grad_dtype      # None: all-reduce the fp32 buffer in place
        self._stage = None                # low-precision staging buffer, same layout as G
        eng = model._get_engine()
        eng.grad_hook = self._on_ready
        model._grad_sync = self
        if self.mode == "overlap" and hasattr(eng.ops, "set_tuning"):
            # the collectives' kernels hold CUs while backward's GEMMs run: the persistent GEMM launches pull their tiles from
            # per-XCD ticket counters instead of striding by the grid size (include/lako_hip.h: lako_tuning_t.nt_queue), so a
            # workgroup that becomes resident late finds the queue drained instead of a whole list of tiles to do
            eng.ops.set_tuning("gemm_nt_queue", 1)

    @property
    def active(self):
        return self.world_size > 1 or self.force

    def _flush(self):
        if self._pending is None:
            return
        lo, hi = self._pending
        self._pending = None
        g = self.model._engine.G[lo:hi]
        if self.grad_dtype is None:
            self.handles.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            st = self._staging()[lo:hi]
            st.copy_(g)
            self.handles.append((dist.all_reduce(st, op=dist.ReduceOp.SUM, group=self.group, async_op=True), g, st))

    def _on_ready(self, lo: int, hi: int):
        if not self.active:
            return
        self._dirty = True
        if self.mode == "deferred":
            return
        if self._pending is not None and self._pending[1] == lo:
            self._pending = (self._pending[0], hi)          # contiguous with the previous range: coalesce
        else:
            self._flush()
            self._pending = (lo, hi)
        if (self._pending[1] - self._pending[0]) * 4 >= self.bucket_bytes:
            self._flush()

    def finish(self):
        if not self.active or not self._dirty:
            return
        self._dirty = False
        if self.mode == "deferred":
            G = self.model._engine.G
            if self.grad_dtype is None:
                dist.all_reduce(G, op=dist.ReduceOp.SUM, group=self.group)
            else:
                st = self._staging()
                st.copy_(G)
                dist.all_reduce(st, op=dist.ReduceOp.SUM, group=self.group)
                G.copy_(st)
            return
        self._flush()
        for h in self.handles:
            if isinstance(h, tuple):
                h[0].wait()
                h[1].copy_(h[2])
            else:
                h.wait()
        self.handles = []

    def _staging(self):
        G = self.model._engine.G
        if self._stage is None or self._stage.numel() != G.numel():
            self._stage = G.new_empty(G.shape, dtype=self.grad_dtype)
        return self._stage


def broadcast_parameters(model, src: int = 0, group=None):
    """Make every rank start from rank `src`'s weights."""
    eng = model._get_engine()
    dist.broadcast(eng.P, src, group=group)
    eng.shadows_stale = True
