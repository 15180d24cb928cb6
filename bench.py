#!/usr/bin/env python
"""Headline benchmark: train samples/sec of the FiD reader (BASELINE.json metric) on N MI355X GPUs.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one iteration of train_reader.py:67-82 on a synthetic OKVQA-shaped batch already resident in
HBM: forward (dropout 0.1 on) → backward → global-norm clip(1.0) → fused AdamW(no bias correction) →
linear-warmup scheduler step → zero_grad; with N > 1 the gradient all-reduce (RCCL over xGMI) is inside
the step, overlapped with backward.  Workload = BASELINE config 2: T5-base, per-GPU batch 16
(run_okvqa_train.sh:25-27), n_passages 20, text_maxlength 200, answer length 8, bf16 compute with fp32
master weights / optimizer.  Weak scaling: per-GPU batch fixed, one sample = question + 20 passages.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the bf16 MFMA GEMM): algorithmic
FLOPs of its launches ÷ their HIP-event time measured live on the last timed step; `step_mfma_frac` is
the whole step against the same peak.  `cpu_baseline` times the CPU oracle (same algorithm, torch fp32 on
the host cores) on a bounded sample of the same workload — a reported baseline, not the target.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

def gemm_traffic_bytes(args):
    """HBM-side bytes per launch of the dominant kernel from the committed PMC profile (separate rocprofv3 --pmc passes
    over tools/gemm_probe.py, FETCH_SIZE doubled as the gfx950 guide prescribes): the mean over the probe's four config-2
    encoder GEMM shapes, measured at the row count this run executes (profiles/r01c_* ≈ 48 k valid tokens when padding is
    skipped, profiles/r01b_* = 64 k rows on the padded path / --all-valid).  null for any other workload or when the file is absent —
    counters cannot be collected from inside a timed run."""
    if (args.model, args.batch, args.n_passages, args.seq_len, args.dtype) != ("base", 16, 20, 200, "bf16"):
        return None
    try:
        padded = os.environ.get("LAKO_UNPAD", "1") == "0" or args.all_valid
        with open(os.path.join(ROOT, "profiles", "r01b_gemm_traffic.json" if padded else "r01c_gemm_traffic.json")) as f:
            return json.load(f)["nt_mean_traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


PEAK_BF16_TFLOPS = 2500.0      # dense MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md (no 2:1 sparsity)
PEAK_F32_TFLOPS = 157.3


def train_flops_per_sample(cfg, N, L, T):
    """SURVEY.md §8a formula: 3 × forward FLOPs (fwd + 2× for bwd)."""
    d, inner, f, V = cfg.d_model, cfg.inner_dim, cfg.d_ff, cfg.vocab_size
    Le, Ld, S = cfg.num_layers, cfg.num_decoder_layers, N * L
    fwd = Le * S * (8 * d * inner + 4 * d * f + 4 * L * inner) \
        + Ld * (T * (8 * d * inner + 4 * T * inner) + 4 * S * d * inner + 4 * T * d * inner + 4 * T * S * inner
                + 4 * T * d * f) + 2 * T * d * V
    return 3.0 * fwd


def executed_train_flops(cfg, lens, T):
    """The same formula evaluated on the tokens that exist (lens: [B, N] valid lengths): what an implementation that
    skips padded positions executes — linear terms ∝ Σ len, encoder self-attention ∝ Σ len², cross-attention ∝ Σ len."""
    d, inner, f, V = cfg.d_model, cfg.inner_dim, cfg.d_ff, cfg.vocab_size
    Le, Ld = cfg.num_layers, cfg.num_decoder_layers
    lens = lens.double()
    tok, sq, B = float(lens.sum()), float((lens * lens).sum()), lens.shape[0]
    fwd = Le * (tok * (8 * d * inner + 4 * d * f) + 4 * sq * inner) \
        + Ld * (B * T * (8 * d * inner + 4 * T * inner) + 4 * tok * d * inner + B * 4 * T * d * inner + 4 * T * tok * inner
                + B * 4 * T * d * f) + B * 2 * T * d * V
    return 3.0 * fwd


def synthetic_batch(B, N, L, T, vocab, seed, device, all_valid=False):
    """ids ~ U{2..32099}, per-passage valid length ~ U{ceil(L/2)..L} (or all L: the pure-roofline variant),
    labels end in EOS, -100 pad."""
    g = torch.Generator().manual_seed(seed)
    hi = min(vocab, 32100)
    ids = torch.randint(2, hi, (B, N, L), generator=g)
    lens = torch.randint((L + 1) // 2, L + 1, (B, N), generator=g)
    if all_valid:
        lens = torch.full_like(lens, L)
    mask = torch.arange(L)[None, None, :] < lens[..., None]
    ids = ids.masked_fill(~mask, 0)
    labels = torch.randint(2, hi, (B, T), generator=g)
    tl = torch.randint(2, T + 1, (B,), generator=g)
    pos = torch.arange(T)[None]
    labels = torch.where(pos == (tl - 1)[:, None], torch.ones_like(labels), labels)
    labels = labels.masked_fill(pos >= tl[:, None], -100)
    return ids.to(device), mask.to(device), labels.to(device)


def effective_cpus() -> int:
    """Cores this process may really use: the affinity mask capped by the cgroup CPU quota (os.cpu_count()
    over-reports inside a container and makes the BLAS thread pool thrash)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                        n = min(n, max(1, q // int(f2.read())))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n)


def cpu_worker(args):
    """Child process: the oracle's train step (oracle/fid_t5_oracle.py — the same algorithm in torch CPU fp32,
    dropout on) on the host cores; prints one line per finished step so the parent can stop it at a deadline."""
    from lako_amd import FiDConfig
    from oracle import fid_t5_oracle as O
    cfg = FiDConfig.named(args.model, dropout_rate=args.dropout)
    dims = O.T5Dims(vocab_size=cfg.vocab_size, d_model=cfg.d_model, d_kv=cfg.d_kv, d_ff=cfg.d_ff,
                    num_layers=cfg.num_layers, num_decoder_layers=cfg.num_decoder_layers, num_heads=cfg.num_heads,
                    dropout=cfg.dropout_rate)
    cores = effective_cpus()
    torch.set_num_threads(cores)
    w = O.init_weights(dims, seed=0, shared_std=0.05)
    state = {}
    print(f"CPUINFO {cores}", flush=True)
    for k in range(1 + args.cpu_steps):
        ids, mask, labels = O.synthetic_batch(args.cpu_batch, args.n_passages, args.seq_len, args.target_len,
                                              dims.vocab_size, seed=1000 + k)
        t0 = time.time()
        O.train_step(w, dims, state, ids, mask, labels, k, 1e-4, 1e-4, 1.0, 2, 100, training=True)
        print(f"CPUSTEP {time.time() - t0:.4f}", flush=True)


def cpu_baseline(args):
    """Run cpu_worker in a subprocess with a hard deadline (bench.py must finish within minutes)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", "--model", args.model, "--cpu-batch",
           str(args.cpu_batch), "--cpu-steps", str(args.cpu_steps), "--n-passages", str(args.n_passages),
           "--seq-len", str(args.seq_len), "--target-len", str(args.target_len), "--dropout", str(args.dropout)]
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env)
    try:
        out, _ = proc.communicate(timeout=args.cpu_seconds)
    except subprocess.TimeoutExpired:
        proc.kill()
        out, _ = proc.communicate()
    cores, times = 0, []
    for line in (out or "").splitlines():
        if line.startswith("CPUINFO"):
            cores = int(line.split()[1])
        elif line.startswith("CPUSTEP"):
            times.append(float(line.split()[1]))
    timed = times[1:] if len(times) > 1 else times
    B = args.cpu_batch
    if not timed:
        return {"value": None, "unit": "samples/s", "cores": cores, "kind": "port",
                "sample": f"oracle train step did not finish one step within {args.cpu_seconds:.0f} s"}
    med = sorted(timed)[len(timed) // 2]
    return {"value": round(B / med, 4), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"oracle train step (torch CPU fp32, dropout on), same model/N/L/T, batch {B} instead of "
                      f"{args.batch}; median of {len(timed)} step(s) after {1 if len(times) > 1 else 0} warm-up, "
                      f"{cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="base")
    ap.add_argument("--batch", type=int, default=16, help="per-GPU batch (run_okvqa_train.sh:25-27: 16 for base)")
    ap.add_argument("--n-passages", type=int, default=20)
    ap.add_argument("--seq-len", type=int, default=200)
    ap.add_argument("--target-len", type=int, default=8)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--all-valid", action="store_true", help="every passage has the full text_maxlength tokens (no padding to skip)")
    ap.add_argument("--cpu-batch", type=int, default=1)
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--cpu-seconds", type=float, default=150.0, help="hard deadline for the CPU baseline leg")
    ap.add_argument("--cpu-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--breakdown", action="store_true", help="print the per-op HIP-event breakdown to stderr")
    args = ap.parse_args()
    if args.cpu_worker:
        return cpu_worker(args)

    import torch.distributed as dist
    from lako_amd import FiDConfig, FiDT5
    from lako_amd import util as U

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    force_dist = os.environ.get("LAKO_FORCE_DIST") == "1"   # exercise the RCCL path even at world size 1
    use_dist = world > 1 or force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=device)

    cfg = FiDConfig.named(args.model, dropout_rate=args.dropout)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(0)                                   # identical initial weights on every rank
    model = FiDT5(cfg, dtype=dtype, seed=rank)
    with torch.no_grad():
        model._params_by_plain["shared.weight"].mul_(0.05)  # random-init stand-in for the t5-* checkpoint
    model = model.cuda(local_rank)
    model.train()
    import types
    total = args.warmup + args.steps
    opt = types.SimpleNamespace(optim="adamw", lr=1e-4, weight_decay=1e-4, scheduler="linear", scheduler_steps=None,
                                total_steps=max(total * 4, 100), warmup_steps=max(int(total * 4 * 0.06), 1),
                                fixed_lr=False)
    optimizer, scheduler = U.set_optim(opt, model)
    if use_dist:
        from lako_amd.dist import GradSync, broadcast_parameters
        broadcast_parameters(model)
        GradSync(model, force=force_dist)
    ops = model._get_engine().ops

    B, N, L, T = args.batch, args.n_passages, args.seq_len, args.target_len
    batches = [synthetic_batch(B, N, L, T, cfg.vocab_size, seed=rank * 7919 + i, device=device, all_valid=args.all_valid)
               for i in range(4)]
    lens_all = torch.stack([b[1].sum(-1).cpu() for b in batches])            # [4, B, N] valid lengths
    unpadded = os.environ.get("LAKO_UNPAD", "1") != "0"
    if unpadded:     # input preparation, like building the batches: passage offsets of the resident batches (cached by mask)
        for _, m_, _ in batches:
            model._get_engine()._ragged_batch(m_, B, N, L)
    loss_acc = torch.zeros((), device=device)

    def step(i):
        ids, mask, labels = batches[i % len(batches)]
        loss = model(input_ids=ids, attention_mask=mask, labels=labels)[0]
        loss.backward()
        U.clip_grad_norm_(model, 1.0)
        optimizer.step()
        scheduler.step()
        model.zero_grad()
        loss_acc.add_(loss.detach())

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    host_ms = None
    for i in range(args.warmup):
        if i == args.warmup - 1 and i > 0:
            # host cost of ENQUEUEING one step: the last warm-up step starts on an idle GPU and is not synchronised inside, so
            # the wall time of its Python calls is the launch path alone (if it approaches ms_per_step the run is host-bound)
            fence()
            th = time.perf_counter()
            step(i)
            host_ms = (time.perf_counter() - th) * 1e3
        else:
            step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i == args.steps - 1:
            ops.probe = []                                  # HIP events around every launch of the LAST timed step
        step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    probe = ops.probe_summary()
    ops.probe = None
    if use_dist:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    final_loss = float(loss_acc.item()) / total

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        fl = train_flops_per_sample(cfg, N, L, T)                                 # nominal: every position of [B, N, L]
        # what this implementation executes: padded positions are skipped on the unpadded path (exact: DESIGN.md §4)
        fl_exec = (sum(executed_train_flops(cfg, lens_all[i % 4], T) for i in range(args.warmup, args.warmup + args.steps))
                   / (args.steps * B)) if unpadded else fl
        peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
        dom = "gemm_nt.11" if args.dtype == "bf16" else "gemm_nt.00"
        n_l, t_ms, f_tot = probe.get(dom, (0, 0.0, 0.0))
        achieved = f_tot / (t_ms * 1e-3) / 1e12 if t_ms > 0 else 0.0
        out = {
            "metric": "train samples/sec (question+n_passages) T5-base OKVQA, 1/2/4/8 GPU",
            "value": round(value, 3), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"C2: FiD reader train step (fwd+bwd+clip+AdamW), T5-{args.model} random-init, "
                                   f"synthetic OKVQA-shaped batches resident in HBM",
                       "per_gpu_batch": B, "global_batch": B * world, "n_passages": N, "text_maxlength": L,
                       "answer_len": T, "dropout": args.dropout, "parallelism": f"dp{world}",
                       "host_enqueue_ms_per_step": None if host_ms is None else round(host_ms, 2),
                       "master_weights": "fp32", "final_mean_loss": round(final_loss, 4),
                       "passage_lengths": "all text_maxlength" if args.all_valid else "U{L/2..L} (SURVEY.md §8d)",
                       "valid_token_frac": round(float(lens_all.double().mean()) / L, 4),
                       "padding": "skipped: encoder runs on valid tokens only (results identical)" if unpadded
                                  else "computed like the reference (LAKO_UNPAD=0)"},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": gemm_traffic_bytes(args),
                         "kernel": ("gemm_nt_kernel<bf16,bf16,2,4,8,4> (256x256 tile, both epilogue instantiations; calls with M > 256"
                                    " rows: encoder + cross-K/V GEMMs incl. their small-tile row tails)") if args.dtype == "bf16"
                                   else "gemm_nt_kernel<f32,f32>",
                         "launches_per_step": n_l, "avg_launch_us": round(t_ms * 1e3 / max(n_l, 1), 2),
                         "step_mfma_frac": round(world * B / (elapsed / args.steps) * fl_exec / 1e12 / (peak * world), 4),
                         "train_gflop_per_sample": round(fl / 1e9, 1),
                         "executed_gflop_per_sample": round(fl_exec / 1e9, 1)},
        }
        if args.breakdown:
            tot = sum(v[1] for v in probe.values())
            for k, (n, t, f) in sorted(probe.items(), key=lambda kv: -kv[1][1]):
                tf = f / (t * 1e-3) / 1e12 if t > 0 and f > 0 else 0.0
                print(f"  {k:16s} launches {n:4d}  {t:8.3f} ms  {100 * t / tot:5.1f}%  {tf:8.1f} TFLOP/s", file=sys.stderr)
            print(f"  sum of kernels {tot:.3f} ms vs step {ms:.3f} ms", file=sys.stderr)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line is the LAST thing on stdout: RCCL writes a banner through C stdio, which (on a pipe) would
        # otherwise surface from libc's buffer only at exit, i.e. after this line
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
