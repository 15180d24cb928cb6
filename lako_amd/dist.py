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

The default transport is fp32 in place, as the reference's DDP (round 5; round 4 let an unmeasured cost table pick bf16 at N = 2 / 4).
`LAKO_DP_GRAD_DTYPE=auto` opts into that table (`estimate_dp_transport`), `=bf16` forces bf16; `LAKO_DP_MODE` and the constructor
arguments override the mode.

`LAKO_DP_GRAD_DTYPE=bf16` (or `grad_dtype=torch.bfloat16`): the gradients travel as bf16 — half the bytes over the xGMI links (446 MB
instead of 892 MB at T5-base; it is the 2- and 4-GPU runs, with one link per peer, that pay most for the collective) — through a
bf16 staging buffer: cast → all-reduce → cast back, two extra passes over the flat buffer (≈0.5 ms).  The sum of world-size bf16
gradients carries one more rounding than the single-GPU path has.

`finish()` makes the compute stream wait for the collectives; the 1/world factor is folded into the fused
optimizer step (lako_adamw_step grad_scale) and into the clip norm, so gradients are never rescaled in a
separate pass.
"""
from __future__ import annotations

import os

import torch.distributed as dist


# ---- the default per world size, by arithmetic (UNMEASURED ON HARDWARE: no multi-GPU node was available to any round) ------------
# One MI355X has 7 xGMI links, one per peer of an 8-GPU node (≈153 GB/s each, both directions together; DESIGN.md §7 prices a
# direction at ≈64 GB/s after protocol overhead).  xGMI is point to point: an all-reduce among N GPUs can use the N − 1 links a GPU
# has to the other participants and nothing else, so with reduce-scatter + all-gather over the full mesh every GPU sends and
# receives bytes/N per peer link, twice.  The fewer the ranks, the fewer the links: it is the 2- and 4-GPU runs that pay most.
XGMI_GBPS_PER_DIRECTION = 64.0
STAGING_TBPS = 5.0            # the two cast passes of the bf16 transport stream the flat buffer at about the chip's copy rate
BF16_MIN_GAIN_MS = 2.0        # the bf16 transport adds a rounding the single-GPU path does not have: only for a gain worth having


def allreduce_ms(nbytes: int, world: int) -> float:
    """Estimated time of one SUM all-reduce of `nbytes` per rank among `world` GPUs of one xGMI mesh."""
    if world <= 1:
        return 0.0
    return 2.0 * (nbytes / world) / (XGMI_GBPS_PER_DIRECTION * 1e9) * 1e3


def dp_cost_table(n_grad: int, world: int) -> dict:
    """ms per step each gradient transport adds, for `n_grad` fp32 gradient elements: fp32 in place, or bf16 through a staging buffer
    (cast → all-reduce → cast back: 12 B of HBM traffic per element on top of half the link bytes)."""
    stage = n_grad * 12 / (STAGING_TBPS * 1e12) * 1e3
    return {"fp32": allreduce_ms(4 * n_grad, world), "bf16": allreduce_ms(2 * n_grad, world) + stage}


def choose_dp(n_grad: int, world: int) -> tuple:
    """(mode, gradient transport) GradSync uses when neither the caller nor LAKO_DP_MODE / LAKO_DP_GRAD_DTYPE say otherwise: ALWAYS
    ("deferred", "fp32") — the reference's DDP all-reduces fp32 gradients (`src/util.py:248-275`), and the default multi-GPU numerics
    must not depend on an unmeasured cost table (round 5, ADVICE).  Mode "deferred" is the one mode that cannot slow the step's kernels
    down (the persistent GEMMs hold every CU; "overlap" stays an opt-in until it has been measured on a node)."""
    return "deferred", "fp32"


def estimate_dp_transport(n_grad: int, world: int) -> str:
    """What the cost table WOULD pick (`LAKO_DP_GRAD_DTYPE=auto` opts into it; the table it reads is what a multi-GPU bench line prints as `dp_estimated_allreduce_ms`): bf16 where
    it saves ≥ BF16_MIN_GAIN_MS — at T5-base (892 MB of fp32 gradients) N = 2 (13.9 → 7.5 ms) and N = 4 (7.0 → 4.0 ms), not N = 8."""
    t = dp_cost_table(n_grad, world)
    return "bf16" if t["fp32"] - t["bf16"] >= BF16_MIN_GAIN_MS else "fp32"


class GradSync:
    def __init__(self, model, group=None, bucket_bytes: int = 64 << 20, force: bool = False, mode: str | None = None,
                 grad_dtype=None):
        import torch
        self.force = force          # issue the collectives even at world size 1 (single-GPU test of the RCCL path)
        self.model = model
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.bucket_bytes = bucket_bytes
        eng = model._get_engine()
        auto_mode, auto_dtype = choose_dp(eng.G.numel(), self.world_size)
        self.cost_table_ms = dp_cost_table(eng.G.numel(), self.world_size)
        self.mode = mode or os.environ.get("LAKO_DP_MODE") or auto_mode
        if self.mode not in ("deferred", "overlap"):
            raise ValueError(f"unknown DP mode {self.mode!r}")
        self.handles = []
        self._pending = None
        self._dirty = False
        if grad_dtype is None:      # the caller did not say: the environment, else the cost table
            name = os.environ.get("LAKO_DP_GRAD_DTYPE") or auto_dtype
            if name == "auto":      # explicit opt-in to the (unmeasured) cost table
                name = estimate_dp_transport(eng.G.numel(), self.world_size)
            if name not in ("bf16", "fp32", "f32"):
                raise ValueError(f"unknown LAKO_DP_GRAD_DTYPE {name!r}")
            grad_dtype = torch.bfloat16 if name == "bf16" else torch.float32
        if grad_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError(f"gradient transport dtype {grad_dtype} unsupported (fp32 or bf16)")
        self.grad_dtype = None if grad_dtype == torch.float32 else grad_dtype      # None: all-reduce the fp32 buffer in place
        self._stage = None                # low-precision staging buffer, same layout as G
        eng.grad_hook = self._on_ready
        eng.grad_ranges_early = self.mode == "overlap"     # deferred: one all-reduce after backward — the engine may group weight gradients across layers
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
