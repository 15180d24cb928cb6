"""bench.py: the pieces the driver depends on.  CPU: the FLOP formulas (SURVEY.md §8a numbers), the synthetic batch
recipe (§8d), that the executed-FLOP count equals the nominal one when nothing is padded.  GPU: one short run must print
exactly one JSON line, last on stdout, with the contract's keys."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from lako_amd import FiDConfig  # noqa: E402


def test_flop_formula_matches_survey_table():
    base, small, large = (FiDConfig.named(n) for n in ("base", "small", "large"))
    assert abs(bench.train_flops_per_sample(base, 20, 200, 8) / 1e9 - 2476) < 1.0          # C2 / C3
    assert abs(bench.train_flops_per_sample(small, 5, 64, 8) / 1e9 - 44.98) < 0.05         # C1
    assert abs(bench.train_flops_per_sample(large, 40, 200, 8) / 1e9 - 17421) < 5.0        # C4


def test_executed_flops_equal_nominal_without_padding_and_shrink_with_it():
    cfg = FiDConfig.named("base")
    B, N, L, T = 4, 20, 200, 8
    full = torch.full((B, N), L)
    assert abs(bench.executed_train_flops(cfg, full, T) / B - bench.train_flops_per_sample(cfg, N, L, T)) < 1e-3 * 2476e9
    _, mask, _ = bench.synthetic_batch(B, N, L, T, cfg.vocab_size, seed=3, device="cpu")
    lens = mask.sum(-1)
    frac = float(lens.double().mean()) / L
    ex = bench.executed_train_flops(cfg, lens, T) / B / bench.train_flops_per_sample(cfg, N, L, T)
    assert 0.6 < frac < 0.9 and frac - 0.03 < ex < frac + 0.01        # ≈ the valid-token share (attention is ∝ len²)


def test_synthetic_batch_recipe():
    ids, mask, labels = bench.synthetic_batch(3, 5, 64, 8, 32128, seed=1, device="cpu")
    assert ids.shape == mask.shape == (3, 5, 64) and labels.shape == (3, 8)
    lens = mask.sum(-1)
    assert int(lens.min()) >= 32 and int(lens.max()) <= 64
    assert torch.equal(mask, torch.arange(64)[None, None] < lens[..., None])        # valid tokens first, then padding
    assert (ids[~mask] == 0).all() and (ids[mask] >= 2).all()
    for row in labels:
        valid = row[row != -100]
        assert 2 <= len(valid) <= 8 and valid[-1] == 1 and (row[len(valid):] == -100).all()
    _, m2, _ = bench.synthetic_batch(3, 5, 64, 8, 32128, seed=1, device="cpu", all_valid=True)
    assert m2.all()
    i4, m4, l4, lens4 = bench.synthetic_batch(3, 5, 64, 8, 32128, seed=1, device="cpu", with_lengths=True)
    assert torch.equal(i4, ids) and torch.equal(m4, mask) and torch.equal(lens4.long(), mask.sum(-1)) and not lens4.is_cuda


def _run_bench(argv, env_extra=None, timeout=120):
    env = dict(os.environ, **(env_extra or {}))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        if env_extra is None or k not in env_extra:
            env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, timeout=timeout, env=env)


def test_bench_gpus_n_cannot_report_the_wrong_gpu_count():
    """`python bench.py --gpus N` the way the driver calls `--gpus 1` (no torchrun around it) must start N ranks itself or fail —
    never print a 1-GPU line under an N-GPU request.  Here: fewer than N devices visible ⇒ non-zero exit, no JSON line."""
    hide = {"HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": "", "ROCR_VISIBLE_DEVICES": ""}
    r = _run_bench(["--gpus", "8", "--steps", "1", "--warmup", "0"], hide)
    assert r.returncode != 0 and "only 0 GPU(s) visible" in r.stderr
    assert not any(ln.lstrip().startswith("{") for ln in r.stdout.splitlines())
    # a launcher environment whose world size is not --gpus (torchrun --nproc-per-node 2 … --gpus 4, or --gpus 1 under a 2-rank launch)
    for gpus, world in (("4", "2"), ("1", "2"), ("2", "1")):
        r = _run_bench(["--gpus", gpus, "--steps", "1", "--warmup", "0"], dict(hide, WORLD_SIZE=world, RANK="0", LOCAL_RANK="0"))
        assert r.returncode != 0 and f"WORLD_SIZE={world}" in r.stderr, (gpus, world, r.stderr[-300:])
        assert not any(ln.lstrip().startswith("{") for ln in r.stdout.splitlines())
    r = _run_bench(["--gpus", "0"], hide)
    assert r.returncode != 0


def test_bench_self_launch_relays_its_arguments():
    cmd = bench.rank_launch_cmd(4, ["--gpus", "4", "--steps", "7", "--warmup", "2", "--no-cpu-baseline"], port=29999)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2", "--no-cpu-baseline"]

    class A:
        gpus = 2
    assert bench.check_world(A, {"WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1"}) == (2, 1, 1)
    with pytest.raises(SystemExit):
        bench.check_world(A, {})                      # not under a launcher and not self-launched: refuse
    with pytest.raises(SystemExit):
        bench.check_world(A, {"WORLD_SIZE": "8"})


@pytest.mark.gpu
def test_bench_prints_one_json_line_last(tmp_path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--model", "small", "--batch", "2", "--n-passages", "3",
                        "--seq-len", "64", "--steps", "2", "--warmup", "1", "--cpu-seconds", "20"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    out = json.loads(lines[-1])
    assert sum(1 for ln in lines if ln.lstrip().startswith("{")) == 1
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1 and out["vs_baseline"] is None
    assert out["unit"] == "samples/s" and out["scaling"] == "weak" and out["data"] == "synthetic" and out["dtype"] == "bf16"
    assert "workload" in out["config"] and "model" not in out["config"]
    for k in ("dp_mode", "dp_grad_dtype", "rccl_world_size", "gemm_dephase"):      # on every line, also at one GPU
        assert k in out["config"], k
    assert out["config"]["rccl_world_size"] is None and out["config"]["dp_mode"].startswith("none")
    rf = out["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert {"value", "unit", "cores", "kind", "sample"} <= set(out["cpu_baseline"]) and out["cpu_baseline"]["kind"] == "port"
    # round 2: the unpadded encoder's input preparation is inside the timed step, per-step median and the all-valid variant are on record
    assert "in timed region" in out["config"]["ragged_prep"] and out["config"]["distinct_batches"] >= 4
    assert out["median_step_ms"] > 0 and "traffic_source" in rf
    av = out["all_valid"]
    assert av["value"] > 0 and av["steps"] >= 1 and "no padding" in av["passage_lengths"]
