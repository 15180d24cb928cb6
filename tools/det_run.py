"""A few seeded training steps of the reader; prints one line: the losses and a digest of every weight afterwards.
Two runs of this script under LAKO_DETERMINISTIC=1 must print the same line (tests/test_parity_gpu.py); without the variable the
digest differs from run to run (float atomics add in arrival order).  The variable is read once per process, hence a script.

  python tools/det_run.py [--size small|base] [--steps 3] [--batch 4] [--passages 8] [--length 160] [--time]
"""
import argparse
import hashlib
import json
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                  # noqa: E402
from lako_amd import FiDConfig, FiDT5, util as U              # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="small")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--passages", type=int, default=8)
    ap.add_argument("--length", type=int, default=160)
    ap.add_argument("--answer", type=int, default=8)
    ap.add_argument("--time", action="store_true", help="also report ms per step over the steps after the first")
    a = ap.parse_args()
    torch.manual_seed(0)                                      # (the dropout key of the engine is drawn from torch's generator)
    cfg = FiDConfig.named(a.size, dropout_rate=0.1)
    m = FiDT5(cfg, dtype=torch.bfloat16, seed=3)
    with torch.no_grad():
        m._params_by_plain["shared.weight"].mul_(0.05)
    m = m.cuda().train()
    g = torch.Generator().manual_seed(11)
    B, N, L, T = a.batch, a.passages, a.length, a.answer
    ids = torch.randint(2, cfg.vocab_size, (B, N, L), generator=g)
    lens = torch.randint(L // 3, L + 1, (B, N), generator=g)       # ragged passages: the unpadded path, partial attention tiles
    mask = torch.arange(L)[None, None, :] < lens[:, :, None]
    ids = ids.masked_fill(~mask, 0)
    labels = torch.randint(2, cfg.vocab_size, (B, T), generator=g)
    labels[0, T - 2:] = -100
    ids, mask, labels = ids.cuda(), mask.cuda(), labels.cuda()
    opt = types.SimpleNamespace(optim="adamw", lr=1e-3, weight_decay=0.01, scheduler="fixed", fixed_lr=True, scheduler_steps=None,
                                total_steps=100, warmup_steps=0)
    optimizer, scheduler = U.set_optim(opt, m)
    losses, t0 = [], None
    for k in range(a.steps):
        if k == 1:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        loss = m(input_ids=ids, attention_mask=mask, labels=labels)[0]
        loss.backward()
        if os.environ.get("DET_RUN_DEBUG"):
            eng = m._engine
            bad = [n for n, p in m._params_by_plain.items() if p.grad is not None and not torch.isfinite(p.grad).all()]
            print(f"step {k}: loss {loss.item():.4f} non-finite grads: {bad[:12]} ({len(bad)})", file=sys.stderr)
        gn = U.clip_grad_norm_(m, 1.0)
        if os.environ.get("DET_RUN_DEBUG"):
            print(f"  grad norm {float(gn) if gn is not None else None}", file=sys.stderr)
        optimizer.step()
        scheduler.step()
        m.zero_grad()
        losses.append(loss.clone())
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / max(1, a.steps - 1) if t0 is not None else None
    h = hashlib.sha256()
    for name in sorted(m._params_by_plain):
        h.update(m._params_by_plain[name].detach().float().cpu().numpy().tobytes())
    out = {"deterministic": bool(m._engine.det), "losses": [float(x.item()).hex() for x in losses], "weights_sha256": h.hexdigest()}
    if a.time:
        out["ms_per_step"] = round(ms, 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
