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

Every timed step is a NEW batch (8 distinct resident batches, cycled) handed over as the data loader would: device
tensors plus the collator's host-side passage lengths; the unpadded encoder's input preparation (row offsets, packed-row
index) is part of the timed step — nothing about a batch is cached across steps.

Rank 0 prints ONE JSON line.  `value` = samples of all ranks ÷ the wall time of the K timed steps (barrier +
synchronize on both sides, max over ranks); `median_step_ms` is the median of the per-step HIP-event times.  `roofline` is
for the dominant kernel (the bf16 MFMA GEMM): algorithmic FLOPs of its launches ÷ their HIP-event time measured live on
one extra step after the timed region; `step_mfma_frac` is the whole step against the same peak.  `all_valid` is a second,
shorter measurement of the same step with every passage at full length (nothing to skip: the pure-roofline variant).
`cpu_baseline` times the CPU oracle (same algorithm, torch fp32 on the host cores) on a bounded sample of the same workload —
a reported baseline, not the target.
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
    over tools/gemm_probe.py, FETCH_SIZE doubled as the gfx950 guide prescribes; tools/gemm_traffic.py): the mean over the encoder's
    NT GEMM shapes weighted by their launches per layer, measured at the row count this run executes (profiles/r06_gemm_traffic.json
    = 48 k valid tokens when padding is skipped, …_padded.json = 64 k rows on the padded path / --all-valid — re-measured in round 6 on
    the four-wave kernels; the r05 / r04 / r03 files are the fallback).  null for any other workload or when the file is absent — counters cannot be collected from inside a timed run."""
    if (args.model, args.batch, args.n_passages, args.seq_len, args.dtype) != ("base", 16, 20, 200, "bf16"):
        return None, None, None
    padded = os.environ.get("LAKO_UNPAD", "1") == "0" or args.all_valid
    for name in (("r06_gemm_traffic_padded.json", "r05_gemm_traffic_padded.json", "r04_gemm_traffic_padded.json", "r03_gemm_traffic_padded.json") if padded else
                 ("r06_gemm_traffic.json", "r05_gemm_traffic.json", "r04_gemm_traffic.json", "r03_gemm_traffic.json")):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                j = json.load(f)
                return j["nt_mean_traffic_bytes_per_launch"], \
                    f"static: profiles/{name} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/gemm_probe.py, not this run)", \
                    j.get("nt_mean_algorithmic_bytes_per_launch")
        except (OSError, KeyError, ValueError):
            continue
    return None, None, None


PEAK_BF16_TFLOPS = 2500.0      # dense MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md (no 2:1 sparsity)
PEAK_F32_TFLOPS = 157.3
PEAK_FP8_TFLOPS = 5000.0       # dense block-scaled fp8 (v_mfma_scale_f32_16x16x128_f8f6f4), same guide


def train_flops_per_sample(cfg, N, L, T):
    """SURVEY.md §8a formula: 3 × forward FLOPs (fwd + 2× for bwd)."""
    d, inner, f, V = cfg.d_model, cfg.inner_dim, cfg.d_ff, cfg.vocab_size
    Le, Ld, S = cfg.num_layers, cfg.num_decoder_layers, N * L
    fwd = Le * S * (8 * d * inner + 4 * d * f + 4 * L * inner) \
        + Ld * (T * (8 * d * inner + 4 * T * inner) + 4 * S * d * inner + 4 * T * d * inner + 4 * T * S * inner
                + 4 * T * d * f) + 2 * T * d * V
    return 3.0 * fwd


def executed_train_flops(cfg, lens, T, xattn=False):
    """The same formula evaluated on the tokens that exist (lens: [B, N] valid lengths): what an implementation that
    skips padded positions executes — linear terms ∝ Σ len, encoder self-attention ∝ Σ len², cross-attention ∝ Σ len.
    xattn: the cross-attention in the encoder-state space (csrc/xattn.hip) — no K/V projection of the tokens (4·tok·d·inner per
    decoder layer); instead the per-head projections on the T·H query rows (4·T·d·inner per sample) and scores / context products
    with K = d instead of d_kv (4·T·H·tok·d instead of 4·T·tok·inner)."""
    d, inner, f, V, H = cfg.d_model, cfg.inner_dim, cfg.d_ff, cfg.vocab_size, cfg.num_heads
    Le, Ld = cfg.num_layers, cfg.num_decoder_layers
    lens = lens.double()
    tok, sq, B = float(lens.sum()), float((lens * lens).sum()), lens.shape[0]
    cross = (B * 4 * T * d * inner + 4 * T * H * tok * d) if xattn else (4 * tok * d * inner + 4 * T * tok * inner)
    fwd = Le * (tok * (8 * d * inner + 4 * d * f) + 4 * sq * inner) \
        + Ld * (B * T * (8 * d * inner + 4 * T * inner) + cross + B * 4 * T * d * inner + B * 4 * T * d * f) + B * 2 * T * d * V
    return 3.0 * fwd


def synthetic_batch(B, N, L, T, vocab, seed, device, all_valid=False, with_lengths=False):
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
    out = (ids.to(device), mask.to(device), labels.to(device))
    return out + (lens.to(torch.int32),) if with_lengths else out     # lens stays on the host (the collator's knowledge)


def workload_tag(args) -> str:
    """BASELINE.json config this run corresponds to (SURVEY.md §8 shorthand), or 'custom'."""
    key = (args.model, args.n_passages, args.seq_len)
    return {("small", 5, 64): "C1", ("base", 20, 200): "C2", ("large", 40, 200): "C4", ("large", 100, 200): "C5"}.get(key, "custom")


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
    for k in range(args.cpu_warmup + args.cpu_steps):
        ids, mask, labels = O.synthetic_batch(args.cpu_batch, args.n_passages, args.seq_len, args.target_len,
                                              dims.vocab_size, seed=1000 + k)
        t0 = time.time()
        O.train_step(w, dims, state, ids, mask, labels, k, 1e-4, 1e-4, 1.0, 2, 100, training=True)
        print(f"CPUSTEP {time.time() - t0:.4f}", flush=True)


def cpu_baseline(args):
    """Run cpu_worker in a subprocess with a hard deadline (bench.py must finish within minutes)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", "--model", args.model, "--cpu-batch",
           str(args.cpu_batch), "--cpu-steps", str(args.cpu_steps), "--cpu-warmup", str(args.cpu_warmup), "--n-passages",
           str(args.n_passages),
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
    nw = min(args.cpu_warmup, max(len(times) - 1, 0))
    timed = times[nw:]
    B = args.cpu_batch
    if not timed:
        return {"value": None, "unit": "samples/s", "cores": cores, "kind": "port",
                "sample": f"oracle train step did not finish one step within {args.cpu_seconds:.0f} s"}
    med = sorted(timed)[len(timed) // 2]
    return {"value": round(B / med, 4), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"oracle train step (torch CPU fp32, dropout on), same model/N/L/T, batch {B} instead of "
                      f"{args.batch}; median of {len(timed)} step(s) after {nw} warm-up(s) (SURVEY §8d asks for >= 10 after 2; "
                      f"the leg is cut off after {args.cpu_seconds:.0f} s), {cores} threads"}


def visible_gpus() -> int:
    """Devices this process could use, counted WITHOUT creating a HIP context in this process: the KFD topology under /sys
    (GPU nodes carry a non-zero simd_count), narrowed by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when they are set.  Only when /sys has no
    topology (a container without it) does it fall back to torch.cuda.device_count(), which may initialise the runtime here — harmless,
    because the ranks are always started as fresh child processes (subprocess, never exec), but then this parent holds a context meanwhile."""
    n = None
    try:
        root = "/sys/class/kfd/kfd/topology/nodes"
        n = 0
        for d in os.listdir(root):
            with open(os.path.join(root, d, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) <= 0:
                continue
            # a container may see the host's /sys but only some /dev/dri/renderD* nodes (cgroup / --device): count a GPU only when its
            # render node is accessible to this process
            minor = int(props.get("drm_render_minor", "0"))
            n += (minor <= 0) or os.access(f"/dev/dri/renderD{minor}", os.R_OK | os.W_OK)
    except OSError:
        n = None
    if not n:
        return int(torch.cuda.device_count())
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def rank_launch_cmd(n: int, argv: list, port: int | None = None) -> list:
    """The command the driver itself uses for N > 1: one process per GPU started by torch.distributed.run, RCCL rendezvous on
    127.0.0.1, bench.py's own arguments relayed unchanged."""
    port = port or int(os.environ.get("LAKO_BENCH_PORT", 29500 + os.getpid() % 2000))
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(args, argv) -> int:
    """`python bench.py --gpus N` (N > 1) WITHOUT torchrun around it: start the N rank processes as a CHILD (subprocess, never exec; this
    parent counts the devices from /sys, see visible_gpus), relay its stdout (rank 0's JSON line) and return its exit
    code.  Fewer than N visible devices is an error, never a silent 1-GPU line (the reference's scaffolding for this: src/slurm.py:157-160,
    src/util.py:248-275)."""
    import subprocess
    have = visible_gpus()
    if have < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible; refusing to report a {have}-GPU run as {args.gpus}", file=sys.stderr)
        return 3
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))   # dmabuf IPC (RCCL across processes)
    proc = subprocess.run(rank_launch_cmd(args.gpus, argv), env=env)
    return proc.returncode


def check_world(args, env=os.environ):
    """(world, rank, local_rank) from the launcher's environment; a world size that is not --gpus ends the run (exit code 4)."""
    world = int(env.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to print a line for the wrong GPU count")
    return world, int(env.get("RANK", "0")), int(env.get("LOCAL_RANK", "0"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=150, help="timed steps (default 150 ≈ 5.5 s of GPU work at config 2: long enough for a 5-s device-utilisation sampler to see)")
    ap.add_argument("--warmup", type=int, default=10)
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
    ap.add_argument("--cpu-steps", type=int, default=10)
    ap.add_argument("--cpu-warmup", type=int, default=2)
    ap.add_argument("--all-valid-steps", type=int, default=10, help="timed steps of the second, all-passages-full measurement (0 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=150.0, help="hard deadline for the CPU baseline leg")
    ap.add_argument("--cpu-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--fp8", action="store_true", help="MX block-scaled fp8 (e4m3 + E8M0 scales) forward GEMMs for the encoder's QKV / FFN-in "
                    "projections and the cross-K/V projection (BASELINE config 5); roofline then describes that kernel against the fp8 peak")
    ap.add_argument("--breakdown", action="store_true", help="print the per-op HIP-event breakdown to stderr")
    args = ap.parse_args()
    if args.cpu_worker:
        return cpu_worker(args)
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: become one (BEFORE anything here touches a GPU) instead of measuring one GPU under an N-GPU label
        raise SystemExit(self_launch(args, sys.argv[1:]))

    import torch.distributed as dist
    from lako_amd import FiDConfig, FiDT5
    from lako_amd import util as U

    world, rank, local_rank = check_world(args)
    if visible_gpus() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} wants device {local_rank} but only {visible_gpus()} GPU(s) are visible")
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
        if dist.get_world_size() != args.gpus and not (force_dist and args.gpus == 1):
            raise SystemExit(f"bench.py: RCCL reports world size {dist.get_world_size()}, --gpus is {args.gpus}")

    cfg = FiDConfig.named(args.model, dropout_rate=args.dropout)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(0)                                   # identical initial weights on every rank
    model = FiDT5(cfg, dtype=dtype, seed=rank, fp8=True if args.fp8 else None)
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
        sync = GradSync(model, force=force_dist)
    ops = model._get_engine().ops

    B, N, L, T = args.batch, args.n_passages, args.seq_len, args.target_len
    NB = 8                                                   # distinct resident batches, cycled
    unpadded = os.environ.get("LAKO_UNPAD", "1") != "0"
    loss_acc = torch.zeros((), device=device)
    peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
    dom = "gemm_nt.11" if args.dtype == "bf16" else "gemm_nt.00"
    if args.fp8:       # the line describes the fp8 kernel (the bf16 GEMMs of the same run are in `bf16_gemm`)
        peak, dom = PEAK_FP8_TFLOPS, "gemm_nt_mx"
    fl = train_flops_per_sample(cfg, N, L, T)                # nominal: every position of [B, N, L]

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(all_valid, warmup, steps):
        """warm-up, then EXACTLY `steps` timed steps between two fences; one more step afterwards carries the per-launch
        HIP-event probe (its event records would perturb the step they sit in)."""
        batches = [synthetic_batch(B, N, L, T, cfg.vocab_size, seed=rank * 7919 + i, device=device, all_valid=all_valid,
                                   with_lengths=True) for i in range(NB)]

        def step(i):
            ids, mask, labels, lens = batches[i % NB]
            # the batch arrives as from the data loader: device tensors + the collator's host-side lengths; offsets and the
            # packed-row index of the unpadded encoder are rebuilt from them inside the step (no per-batch cache)
            loss = model(input_ids=ids, attention_mask=mask, labels=labels, passage_lengths=lens)[0]
            loss.backward()
            U.clip_grad_norm_(model, 1.0)
            optimizer.step()
            scheduler.step()
            model.zero_grad()
            loss_acc.add_(loss.detach())

        host_ms = None
        for i in range(warmup):
            if i == warmup - 1 and i > 0:
                # host cost of ENQUEUEING one step: the last warm-up step starts on an idle GPU and is not synchronised inside,
                # so the wall time of its Python calls is the launch path alone (if it approaches ms_per_step the run is host-bound)
                fence()
                th = time.perf_counter()
                step(i)
                host_ms = (time.perf_counter() - th) * 1e3
            else:
                step(i)
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        fence()
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(steps):
            step(warmup + i)
            marks[i + 1].record()
        fence()
        elapsed = time.perf_counter() - t0
        per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
        ops.probe = []
        step(warmup + steps)
        torch.cuda.synchronize()
        probe = ops.probe_summary()
        ops.probe = None
        if use_dist:
            t = torch.tensor([elapsed], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        lens_all = torch.stack([b[3] for b in batches])                       # [NB, B, N]
        # what this implementation executes: padded positions are skipped on the unpadded path (exact: DESIGN.md §4)
        xattn = bool(getattr(model._engine, "xattn_active", False))
        full = torch.full_like(lens_all[0], L)
        fl_exec = sum(executed_train_flops(cfg, lens_all[i % NB] if unpadded else full, T, xattn)
                      for i in range(warmup, warmup + steps)) / (steps * B)
        fl_proj = sum(executed_train_flops(cfg, lens_all[i % NB] if unpadded else full, T, False)
                      for i in range(warmup, warmup + steps)) / (steps * B)
        n_l, t_ms, f_tot = probe.get(dom, (0, 0.0, 0.0))
        achieved = f_tot / (t_ms * 1e-3) / 1e12 if t_ms > 0 else 0.0
        nb, tb, fb = probe.get("gemm_nt.11", (0, 0.0, 0.0))
        bf16_gemm = {"launches_per_step": nb, "achieved_tflops": round(fb / (tb * 1e-3) / 1e12, 1) if tb > 0 else 0.0,
                     "ms_per_step": round(tb, 3)}
        mxq = probe.get("mx_quantize", (0, 0.0, 0.0))
        return dict(bf16_gemm=bf16_gemm, mxq_ms=mxq[1], elapsed=elapsed, ms=elapsed / steps * 1e3, median_ms=per_step[len(per_step) // 2], host_ms=host_ms,
                    probe=probe, fl_exec=fl_exec, fl_proj=fl_proj, xattn=xattn, valid_frac=float(lens_all.double().mean()) / L, n_l=n_l, t_ms=t_ms,
                    achieved=achieved, value=world * B * steps / elapsed,
                    step_frac=world * B / (elapsed / steps) * fl_exec / 1e12 / (PEAK_BF16_TFLOPS * world if args.fp8 else peak * world))

    main_run = measure(args.all_valid, args.warmup, args.steps)
    av_run = None
    if not args.all_valid and args.all_valid_steps > 0:
        av_run = measure(True, 3, args.all_valid_steps)
    total = args.warmup + args.steps + 1 + (3 + args.all_valid_steps + 1 if av_run else 0)
    final_loss = float(loss_acc.item()) / total

    if rank == 0:
        r = main_run
        traffic, traffic_src, traffic_alg = gemm_traffic_bytes(args)
        out = {
            "metric": "train samples/sec (question+n_passages) T5-base OKVQA, 1/2/4/8 GPU",
            "value": round(r["value"], 3), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(r["ms"], 3), "median_step_ms": round(r["median_ms"], 3),
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "fp8-mx (e4m3 operands + E8M0 block scales in the fwd QKV / FFN-in / cross-K/V GEMMs; bf16 elsewhere)"
            if args.fp8 else args.dtype, "data": "synthetic",
            "config": {"workload": f"{workload_tag(args)}: FiD reader train step "
                                   f"(fwd+bwd+clip+AdamW), T5-{args.model} random-init, synthetic OKVQA-shaped batches resident in HBM",
                       "per_gpu_batch": B, "global_batch": B * world, "n_passages": N, "text_maxlength": L,
                       "answer_len": T, "dropout": args.dropout, "parallelism": f"dp{world}",
                       "rccl_world_size": dist.get_world_size() if use_dist else None,      # None: one GPU, RCCL not initialised
                       "dp_mode": sync.mode if use_dist else "none (one GPU: no collective)",
                       "dp_grad_dtype": ("bf16" if sync.grad_dtype is not None else "fp32") if use_dist else None,
                       "dp_estimated_allreduce_ms": {k: round(v, 2) for k, v in sync.cost_table_ms.items()} if use_dist else None,
                       "gemm_dephase": {"ticks_10ns": int(ops.tuning.nt_dephase), "phases": int(ops.tuning.nt_dephase_n),
                                        "note": "eight-wave kernels only (row tails, fp32, odd K): every other workgroup of an XCD starts late in multi-round "
                                                "persistent launches; the four-wave kernels of round 6 run in lockstep (tools/gemm_nt4_dephase.py)"},
                       "host_enqueue_ms_per_step": None if r["host_ms"] is None else round(r["host_ms"], 2),
                       "master_weights": "fp32", "final_mean_loss": round(final_loss, 4),
                       "passage_lengths": "all text_maxlength" if args.all_valid else "U{L/2..L} (SURVEY.md §8d)",
                       "valid_token_frac": round(r["valid_frac"], 4),
                       "distinct_batches": NB,
                       "ragged_prep": "in timed region (offsets + packed-row index rebuilt every step from the collator's host-side "
                                      "lengths; no mask read-back, no per-batch cache)" if unpadded else "n/a (padded path)",
                       "padding": "skipped: encoder runs on valid tokens only (results identical)" if unpadded
                                  else "computed like the reference (LAKO_UNPAD=0)",
                       "cross_attention": "encoder-state space (Q' = q.Wk per head, S = Q'.E^T, ctx = (P.E).Wv^T: same results, no K/V "
                                          "projection of the n_passages*L encoder states; LAKO_XATTN=0 = projected K/V)" if r["xattn"]
                                          else "projected K/V (reference formulation)"},
            "roofline": {"bound": "mfma", "achieved": round(r["achieved"], 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(r["achieved"] / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_over_algorithmic_bytes": round(traffic / traffic_alg, 3) if traffic and traffic_alg else None,
                         "kernel": ("gemm_nt4_mx_kernel<MT,SIDE> (round 6: 256x256 / 192x256 tiles on four waves, two K-slices of LDS-DMA in flight, "
                                    "v_mfma_scale_f32_16x16x128_f8f6f4, e4m3 x e4m3 with E8M0 block scales; gemm_nt_mx_kernel where K % 512 != 0: "
                                    "forward QKV / FFN-in / cross-K/V projections)") if args.fp8 else
                                   ("gemm_nt4_kernel<MT,SIDE,EPI> (round 6: 256x256 / 192x256 tiles on four waves, hand-placed K loop with two K-slices of "
                                    "LDS-DMA in flight; all instantiations; calls with M > 256 rows: the encoder's GEMMs; + the cross-K/V "
                                    "projection under LAKO_XATTN=0)") if args.dtype == "bf16"
                                   else "gemm_nt_kernel<f32,f32>",
                         "launches_per_step": r["n_l"], "avg_launch_us": round(r["t_ms"] * 1e3 / max(r["n_l"], 1), 2),
                         "step_mfma_frac": round(r["step_frac"], 4),
                         "nominal_step_frac": round(r["value"] / world * fl / 1e12 / peak, 4),
                         "train_gflop_per_sample": round(fl / 1e9, 1),
                         "executed_gflop_per_sample": round(r["fl_exec"] / 1e9, 1),
                         "flop_note": "train_gflop_per_sample: SURVEY.md §8a formula on every position of [B, N, L]; executed_…: the FLOPs this "
                                      "implementation issues (valid tokens only" + (", cross-attention in the encoder-state space: no K/V "
                                      "projection of the encoder states; the reference formulation on the same valid tokens would be "
                                      f"{r['fl_proj'] / 1e9:.1f} GFLOP/sample" if r["xattn"] else "") + "); step_mfma_frac prices the executed FLOPs, nominal_step_frac = samples/s x train_gflop_per_sample / peak (SURVEY.md §8d's formula, "
                                      "which counts padded positions and the K/V projection)"},
        }
        if args.fp8:
            out["roofline"]["bf16_gemm"] = r["bf16_gemm"]          # the GEMMs that stay bf16 in the same run (backward, o / wo projections)
            out["roofline"]["mx_quantize_ms_per_step"] = round(r["mxq_ms"], 3)
            out["roofline"]["step_mfma_frac_note"] = "whole step priced against the bf16 peak (most FLOPs of the step still run in bf16)"
        if av_run is not None:
            out["all_valid"] = {"value": round(av_run["value"], 3), "unit": "samples/s", "steps": args.all_valid_steps, "warmup": 3,
                                "ms_per_step": round(av_run["ms"], 3), "median_step_ms": round(av_run["median_ms"], 3),
                                "passage_lengths": "all text_maxlength (no padding to skip)",
                                "roofline_frac": round(av_run["achieved"] / peak, 4),
                                "step_mfma_frac": round(av_run["step_frac"], 4)}
        if args.breakdown:
            probe = r["probe"]
            tot = sum(v[1] for v in probe.values())
            for k, (n, t, f) in sorted(probe.items(), key=lambda kv: -kv[1][1]):
                tf = f / (t * 1e-3) / 1e12 if t > 0 and f > 0 else 0.0
                print(f"  {k:16s} launches {n:4d}  {t:8.3f} ms  {100 * t / tot:5.1f}%  {tf:8.1f} TFLOP/s", file=sys.stderr)
            print(f"  sum of kernels {tot:.3f} ms vs step {r['ms']:.3f} ms", file=sys.stderr)
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
