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
    rf = out["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert {"value", "unit", "cores", "kind", "sample"} <= set(out["cpu_baseline"]) and out["cpu_baseline"]["kind"] == "port"
    # round 2: the unpadded encoder's input preparation is inside the timed step, per-step median and the all-valid variant are on record
    assert "in timed region" in out["config"]["ragged_prep"] and out["config"]["distinct_batches"] >= 4
    assert out["median_step_ms"] > 0 and "traffic_source" in rf
    av = out["all_valid"]
    assert av["value"] > 0 and av["steps"] >= 1 and "no padding" in av["passage_lengths"]
