"""Optimizer / scheduler / clip helpers mirroring the reference's `src/util.py` names and semantics.

  set_optim(opt, model)            src/util.py:230-245
  WarmupLinearScheduler            src/util.py:149-168
  FixedScheduler                   src/util.py:171-176
  AdamW                            the HF<=4 `transformers.AdamW(correct_bias=False)` src/util.py:225 builds
  clip_grad_norm_(model, max_norm) train_reader.py:76 (torch.nn.utils.clip_grad_norm_)
  average_main / weighted_average  src/util.py:248-275 (scalar reduces to rank 0)
  save / load / symlink_force      src/util.py:63-71,105-146 (checkpoint directory: save_pretrained files +
                                   optimizer.pth.tar + `latest` symlink)

The optimizer is fused: one kernel launch updates the whole flat fp32 parameter buffer (clip coefficient,
Adam moments without bias correction, decoupled weight decay) and writes the low-precision shadow.
"""
from __future__ import annotations

import errno
import logging
import os

import torch
import torch.distributed as dist

logger = logging.getLogger(__name__)


def _model_of(params):
    for p in params:
        m = getattr(p, "_lako_model", None)
        if m is not None:
            return m()
    raise ValueError("these parameters do not belong to a lako_amd.FiDT5 (use model.parameters())")


class AdamW(torch.optim.Optimizer):
    """HF<=4 AdamW semantics: m ← β1 m + (1−β1) g; v ← β2 v + (1−β2) g²;
    p ← p − lr·m/(√v+eps); p ← p − lr·wd·p — one fused launch over the flat parameter buffer.
    `correct_bias=True` (HF: step size lr·√(1−β2ᵗ)/(1−β1ᵗ), decay with the plain lr) and `torch_adam=True`
    (torch.optim.Adam as the reference's `--optim adam` builds it, src/util.py:231-232: bias-corrected, eps inside the
    corrected denominator, no decay) run on the same kernel through an equivalent (lr, eps, wd) per step:
        torch Adam:  p −= lr/bc1 · m / (√v/√bc2 + eps)  =  lr·√bc2/bc1 · m / (√v + eps·√bc2).
    All parameter groups must share lr / weight_decay (they do: src/util.py:189-194)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=False,
                 model=None, torch_adam=False):
        params = list(params)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                      correct_bias=bool(correct_bias)))
        if model is None:
            model = _model_of(p for g in self.param_groups for p in g["params"])
        self._model = model
        self._torch_adam = bool(torch_adam)
        self._t = 0

    def _find_model(self):
        return self._model

    @torch.no_grad()
    def step(self, closure=None):
        model = self._find_model()
        g0 = self.param_groups[0]
        for g in self.param_groups[1:]:
            if g["lr"] != g0["lr"] or g["weight_decay"] != g0["weight_decay"]:
                raise ValueError("fused AdamW needs identical lr / weight_decay in all groups")
        eng = model._get_engine()
        sync = getattr(model, "_grad_sync", None)
        scale = 1.0
        if sync is not None:
            sync.finish()
            scale = 1.0 / sync.world_size
        if eng.opt_m is None:
            eng.opt_m = torch.zeros_like(eng.P)
            eng.opt_v = torch.zeros_like(eng.P)
        clip = getattr(model, "_pending_clip", None)
        b1, b2 = g0["betas"]
        self._t += 1
        lr, eps, wd = g0["lr"], g0["eps"], g0["weight_decay"]
        if self._torch_adam or g0["correct_bias"]:
            bc1, bc2 = 1.0 - b1 ** self._t, 1.0 - b2 ** self._t
            step_lr = lr * (bc2 ** 0.5) / bc1
            if self._torch_adam:
                eps, wd = eps * (bc2 ** 0.5), 0.0
            elif step_lr > 0:
                wd = wd * lr / step_lr          # the kernel decays by (step size)·wd: keep HF's lr·wd
            lr = step_lr
        eng.ops.adamw_step(eng.P, eng.G, eng.opt_m, eng.opt_v, None if eng.W is eng.P else eng.W,
                           lr=lr, beta1=b1, beta2=b2, eps=eps, weight_decay=wd,
                           gnorm_sq=eng.gnorm_sq if clip is not None else None,
                           max_norm=clip if clip is not None else 0.0, grad_scale=scale)
        model._pending_clip = None
        eng.refresh_after_step()

    def state_dict(self):
        sd = super().state_dict()
        sd["lako_step"] = self._t
        eng = self._find_model()._engine
        if eng is not None and eng.opt_m is not None:
            sd["lako_exp_avg"], sd["lako_exp_avg_sq"] = eng.opt_m.detach().cpu(), eng.opt_v.detach().cpu()
        return sd

    def load_state_dict(self, sd):
        sd = dict(sd)
        m, v = sd.pop("lako_exp_avg", None), sd.pop("lako_exp_avg_sq", None)
        self._t = int(sd.pop("lako_step", 0))
        super().load_state_dict(sd)
        if m is not None:
            eng = self._find_model()._get_engine()
            eng.opt_m, eng.opt_v = m.to(eng.device), v.to(eng.device)


def clip_grad_norm_(model, max_norm: float):
    """Global L2 norm of all gradients (after the data-parallel all-reduce) computed on the device; the
    clip coefficient min(1, max_norm/(norm+1e-6)) is applied inside the fused optimizer step, so no host
    sync happens here.  Returns the norm as a 0-d device tensor."""
    eng = model._get_engine()
    sync = getattr(model, "_grad_sync", None)
    scale = 1.0
    if sync is not None:
        sync.finish()
        scale = 1.0 / sync.world_size
    eng.ops.zero_(eng.gnorm_sq)
    eng.ops.sumsq(eng.G, eng.gnorm_sq)
    model._pending_clip = float(max_norm)
    return eng.gnorm_sq[0].sqrt() * scale


class WarmupLinearScheduler(torch.optim.lr_scheduler.LambdaLR):
    def __init__(self, optimizer, warmup_steps, scheduler_steps, min_ratio, fixed_lr, last_epoch=-1):
        self.warmup_steps = warmup_steps
        self.scheduler_steps = scheduler_steps
        self.min_ratio = min_ratio
        self.fixed_lr = fixed_lr
        super().__init__(optimizer, self.lr_lambda, last_epoch=last_epoch)

    def lr_lambda(self, step):
        if step < self.warmup_steps:
            return (1 - self.min_ratio) * step / float(max(1, self.warmup_steps)) + self.min_ratio
        if self.fixed_lr:
            return 1.0
        return max(0.0, 1.0 + (self.min_ratio - 1) * (step - self.warmup_steps)
                   / float(max(1.0, self.scheduler_steps - self.warmup_steps)))


class FixedScheduler(torch.optim.lr_scheduler.LambdaLR):
    def __init__(self, optimizer, last_epoch=-1):
        super().__init__(optimizer, self.lr_lambda, last_epoch=last_epoch)

    def lr_lambda(self, step):
        return 1.0


def set_optim(opt, model):
    """src/util.py:230-245 with the fused AdamW in place of HF's."""
    if opt.optim == "adamw":
        no_decay = ["bias", "LayerNorm.bias", "LayerNorm.weight"]
        named = list(model.named_parameters())
        groups = [
            {"params": [p for n, p in named if not any(nd in n for nd in no_decay)], "weight_decay": opt.weight_decay},
            {"params": [p for n, p in named if any(nd in n for nd in no_decay)], "weight_decay": opt.weight_decay},
        ]
        optimizer = AdamW(groups, lr=opt.lr, correct_bias=False, model=model)
    elif opt.optim == "adam":        # src/util.py:231-232: torch.optim.Adam(model.parameters(), lr=opt.lr)
        optimizer = AdamW(model.parameters(), lr=opt.lr, eps=1e-8, weight_decay=0.0, model=model, torch_adam=True)
    else:
        raise ValueError(f"--optim {opt.optim}: adam or adamw (src/options.py)")
    if opt.scheduler == "fixed":
        scheduler = FixedScheduler(optimizer)
    elif opt.scheduler == "linear":
        steps = opt.total_steps if getattr(opt, "scheduler_steps", None) is None else opt.scheduler_steps
        scheduler = WarmupLinearScheduler(optimizer, warmup_steps=opt.warmup_steps, scheduler_steps=steps,
                                          min_ratio=0.0, fixed_lr=opt.fixed_lr)
    else:
        raise ValueError(opt.scheduler)
    return optimizer, scheduler


def average_main(x, opt):
    """src/util.py:248-255: sum-reduce a scalar to rank 0 and divide there."""
    if not getattr(opt, "is_distributed", False):
        return x
    if opt.world_size > 1:
        dist.reduce(x, 0, op=dist.ReduceOp.SUM)
        if opt.is_main:
            x = x / opt.world_size
    return x


def weighted_average(x, count, opt):
    """src/util.py:266-275."""
    if not getattr(opt, "is_distributed", False):
        return x, count
    dev = opt.device if hasattr(opt, "device") else "cpu"
    t_loss = torch.tensor([x * count], device=dev)
    t_total = torch.tensor([count], device=dev)
    dist.reduce(t_loss, 0, op=dist.ReduceOp.SUM)
    dist.reduce(t_total, 0, op=dist.ReduceOp.SUM)
    return (t_loss / t_total).item(), t_total.item()


# ---- checkpoint directory (src/util.py:63-71, 105-146) -------------------------------------------------------
def symlink_force(target, link_name):
    """`ln -sf`: replace an existing link (src/util.py:63-71)."""
    try:
        os.symlink(target, link_name)
    except OSError as e:
        if e.errno != errno.EEXIST:
            raise
        os.remove(link_name)
        os.symlink(target, link_name)


def save(model, optimizer, scheduler, step, best_eval_metric, opt, dir_path, name):
    """<dir_path>/checkpoint/<name>/ = save_pretrained files + optimizer.pth.tar {step, optimizer, scheduler, opt,
    best_eval_metric}; <dir_path>/checkpoint/latest → that directory (src/util.py:105-121).  The optimizer entry
    carries the fused AdamW's flat moment buffers (AdamW.state_dict)."""
    model_to_save = model.module if hasattr(model, "module") else model
    path = os.path.join(dir_path, "checkpoint")
    epoch_path = os.path.join(path, name)
    os.makedirs(epoch_path, exist_ok=True)
    model_to_save.save_pretrained(epoch_path)
    checkpoint = {"step": step, "optimizer": optimizer.state_dict(), "scheduler": scheduler.state_dict(), "opt": opt,
                  "best_eval_metric": best_eval_metric}
    torch.save(checkpoint, os.path.join(epoch_path, "optimizer.pth.tar"))
    symlink_force(epoch_path, os.path.join(path, "latest"))


def load(model_class, dir_path, opt, reset_params=False, **model_kw):
    """Inverse of save (src/util.py:124-146): returns (model, optimizer, scheduler, opt_checkpoint, step,
    best_eval_metric).  `reset_params=True` keeps the weights but builds a fresh optimizer / scheduler from `opt`
    (what train_reader.py:255 does when --model_path is given).  Accepts the older key `best_dev_em`."""
    epoch_path = os.path.realpath(dir_path)
    optimizer_path = os.path.join(epoch_path, "optimizer.pth.tar")
    logger.info("Loading %s", epoch_path)
    model = model_class.from_pretrained(epoch_path, **model_kw)
    device = getattr(opt, "device", None)
    if device is not None and torch.device(device).type == "cuda":
        model = model.to(device)
    logger.info("loading checkpoint %s", optimizer_path)
    # (the reference's format: `opt` is the pickled argparse Namespace, src/util.py:113-118.  Round 5: the file is read with
    # weights_only=True — tensors, containers and plain numbers, plus the two allow-listed attribute bags an `opt` can be (argparse.Namespace,
    # types.SimpleNamespace); anything else in the pickle is refused instead of executed)
    import argparse
    import types
    with torch.serialization.safe_globals([argparse.Namespace, types.SimpleNamespace]):
        checkpoint = torch.load(optimizer_path, map_location="cpu", weights_only=True)
    opt_checkpoint = checkpoint["opt"]
    step = checkpoint["step"]
    best_eval_metric = checkpoint["best_eval_metric"] if "best_eval_metric" in checkpoint else checkpoint["best_dev_em"]
    if not reset_params:
        optimizer, scheduler = set_optim(opt_checkpoint, model)
        scheduler.load_state_dict(checkpoint["scheduler"])
        optimizer.load_state_dict(checkpoint["optimizer"])
    else:
        optimizer, scheduler = set_optim(opt, model)
    return model, optimizer, scheduler, opt_checkpoint, step, best_eval_metric
