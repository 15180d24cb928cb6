"""§8 f3: checkpoint directory format (src/util.py:105-146) and the answer metrics (src/evaluation.py:130-167).

The metric golden values were produced by the reference's own functions (oracle/make_fixtures.py::make_evaluation).
The checkpoint tests run the host logic on the fp32 test double (tests/ref_ops.py): train 2 steps → save → load →
the resumed run must continue exactly like the uninterrupted one (weights, AdamW moments, scheduler position)."""
import json
import os
import types

import pytest
import torch

from lako_amd import FiDT5
from lako_amd import evaluation as E
from lako_amd import util as U
from tests.ref_ops import RefOps
from tests.test_engine_cpu import build


def test_answer_metrics_match_reference():
    with open(os.path.join(os.path.dirname(__file__), "golden", "evaluation.json")) as f:
        rows = json.load(f)
    assert len(rows) >= 10
    for r in rows:
        if r.get("stem"):
            continue
        assert E.normalize_answer(r["prediction"]) == r["normalized"]
        for k, v in r["golds_normalized"].items():
            assert E.normalize_answer(k) == v
        assert float(E.ems(r["prediction"], r["golds"])) == r["ems"]
        assert float(E.includ_ems(r["prediction"], r["golds"])) == r["includ_ems"]


def test_stem_ems_and_stopword_table_match_reference():
    """stem_ems with and without the stop-word mode, and normalize_answer(dele_sw=True) on the shipped table, against
    values the reference's own functions produced (oracle/make_fixtures.py::make_evaluation, same stand-in stemmer)."""
    from tests.util_golden import FixStem as _FixStem, FixTok as _FixTok
    with open(os.path.join(os.path.dirname(__file__), "golden", "evaluation.json")) as f:
        rows = [r for r in json.load(f) if r.get("stem")]
    assert len(rows) >= 20 and len(E.STOP_WORDS) > 300
    hit = 0
    for r in rows:
        assert E.normalize_answer(r["prediction"], dele_sw=True) == r["normalized_sw"]
        assert float(E.stem_ems(r["prediction"], r["golds"], _FixTok(), _FixStem())) == r["stem_ems"]
        assert float(E.stem_ems(r["prediction"], r["golds"], _FixTok(), _FixStem(), dele_sw=True)) == r["stem_ems_sw"]
        hit += r["stem_ems"] != r["stem_ems_sw"] or r["normalized_sw"] != E.normalize_answer(r["prediction"])
    assert hit >= 3          # the stop-word mode changes something on these strings


def test_stem_ems_and_stopword_mode():
    class Tok:
        def tokenize(self, s):
            return s.split()

    class Stem:
        def stem(self, w):
            return w[:-3] if w.endswith("ing") else w
    golds = {"skiing": 1.0, "snowboarding": 0.6, "sledding": 0.3}
    assert E.stem_ems("ski", golds, Tok(), Stem()) == 1.0            # best-scored gold sharing a stem wins
    assert E.stem_ems("people snowboard", golds, Tok(), Stem()) == 0.6
    assert E.stem_ems("swimming", golds, Tok(), Stem()) == 0
    assert E.normalize_answer("what is the dog", dele_sw=True, stop_words=["what", "is"]) == "dog"


def _opt():
    return types.SimpleNamespace(optim="adamw", lr=3e-3, weight_decay=1e-2, scheduler="linear", scheduler_steps=None,
                                 total_steps=8, warmup_steps=2, fixed_lr=False, device="cpu")


def _steps(model, optimizer, scheduler, batches):
    model.train()
    for ids, mask, labels in batches:
        model(input_ids=ids, attention_mask=mask, labels=labels)[0].backward()
        U.clip_grad_norm_(model, 1.0)
        optimizer.step()
        scheduler.step()
        model.zero_grad()


def test_save_load_resume_is_exact(tmp_path):
    from oracle import fid_t5_oracle as O
    z, dims, w, model = build("tiny_a")
    B, N, L = z["input_ids"].shape
    T = z["labels"].shape[1]
    batches = [O.synthetic_batch(B, N, L, T, dims.vocab_size, seed=700 + k) for k in range(4)]
    optimizer, scheduler = U.set_optim(_opt(), model)
    _steps(model, optimizer, scheduler, batches[:2])
    U.save(model, optimizer, scheduler, 2, 0.25, _opt(), str(tmp_path), "step-2")
    ck = tmp_path / "checkpoint"
    assert (ck / "step-2" / "optimizer.pth.tar").exists() and (ck / "step-2" / "config.json").exists()
    assert os.path.islink(ck / "latest") and os.path.realpath(ck / "latest") == os.path.realpath(ck / "step-2")
    _steps(model, optimizer, scheduler, batches[2:])                      # the uninterrupted run

    m2, o2, s2, opt_ck, step, best = U.load(FiDT5, str(ck / "latest"), _opt(), dtype=torch.float32, _ops=RefOps())
    assert step == 2 and best == 0.25 and opt_ck.lr == 3e-3
    assert s2.last_epoch == 2 and abs(s2.get_last_lr()[0] - scheduler.get_last_lr()[0]) > 0   # position restored, not final
    _steps(m2, o2, s2, batches[2:])
    assert torch.equal(m2._engine.P, model._engine.P)
    assert torch.equal(m2._engine.opt_m, model._engine.opt_m) and torch.equal(m2._engine.opt_v, model._engine.opt_v)
    assert s2.get_last_lr() == scheduler.get_last_lr()

    # a second save under the same link name replaces the symlink (symlink_force)
    U.save(m2, o2, s2, 4, 0.5, _opt(), str(tmp_path), "step-4")
    assert os.path.realpath(ck / "latest") == os.path.realpath(ck / "step-4")

    # reset_params: weights kept, optimizer and scheduler fresh (train_reader.py:255)
    m3, o3, s3, _, step3, best3 = U.load(FiDT5, str(ck / "step-2"), _opt(), reset_params=True, dtype=torch.float32,
                                         _ops=RefOps())
    assert step3 == 2 and s3.last_epoch == 0 and m3._get_engine().opt_m is None


class _Evil:
    def __reduce__(self):                       # what an attacker's pickle would run on load
        return (os.system, ("true",))


def test_load_refuses_a_checkpoint_that_carries_code(tmp_path):
    """(round 5) optimizer.pth.tar is read with torch.load(weights_only=True) + an allow-list of the two attribute-bag classes an `opt` can
    be: a pickle that names any other global — here os.system through __reduce__ — is refused, not executed."""
    import pickle
    _, _, _, model = build("tiny_a")
    optimizer, scheduler = U.set_optim(_opt(), model)
    U.save(model, optimizer, scheduler, 7, 0.1, _opt(), str(tmp_path), "bad")
    fp = tmp_path / "checkpoint" / "bad" / "optimizer.pth.tar"
    ck = torch.load(fp, weights_only=False)
    ck["opt"] = _Evil()
    torch.save(ck, fp)
    with pytest.raises(pickle.UnpicklingError):
        U.load(FiDT5, str(tmp_path / "checkpoint" / "bad"), _opt(), dtype=torch.float32, _ops=RefOps())


def test_load_accepts_legacy_metric_key(tmp_path):
    _, _, _, model = build("tiny_a")
    optimizer, scheduler = U.set_optim(_opt(), model)
    U.save(model, optimizer, scheduler, 7, 0.1, _opt(), str(tmp_path), "best_dev")
    fp = tmp_path / "checkpoint" / "best_dev" / "optimizer.pth.tar"
    ck = torch.load(fp, weights_only=False)
    ck["best_dev_em"] = ck.pop("best_eval_metric")
    torch.save(ck, fp)
    *_, step, best = U.load(FiDT5, str(tmp_path / "checkpoint" / "best_dev"), _opt(), dtype=torch.float32, _ops=RefOps())
    assert (step, best) == (7, 0.1)


def test_test_reader_evaluate_writes_scores_and_results(tmp_path):
    """test_reader.py::evaluate on the fp32 test double: greedy decode, the three metrics, results JSON, and the
    per-fact cross-attention scores written back into the examples (test_reader.py:62-76,107-122)."""
    import copy
    import importlib.util
    from torch.utils.data import DataLoader, SequentialSampler

    from lako_amd import FiDConfig
    from lako_amd.data import Collator, Dataset
    from tests.stub_tokenizer import StubTokenizer
    from tests.test_data_pipeline import EXAMPLES
    spec = importlib.util.spec_from_file_location("reader_eval", os.path.join(os.path.dirname(__file__), "..", "test_reader.py"))
    tr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tr)

    cfg = FiDConfig(vocab_size=64, d_model=32, d_kv=32, d_ff=64, num_layers=2, num_decoder_layers=2, num_heads=2, dropout_rate=0.0)
    model = FiDT5(cfg, dtype=torch.float32, seed=3, _ops=RefOps())
    for style in ("mean", "max"):
        for ans_attention in ("no", "yes"):
            opt = types.SimpleNamespace(n_context=2, fact_use_way="concate", use_fact="yes", stream=2, write_results=True,
                                        write_crossattention_scores=True, ans_attention=ans_attention,
                                        attention_score_style=style, use_last_half_layer_attention="no", dataset="okvqa",
                                        model_size="tiny", per_gpu_batch_size=2, text_maxlength=24, is_distributed=False,
                                        eval_data="dev.json", version="v1", device="cpu")
            ds = Dataset(copy.deepcopy(EXAMPLES), opt)
            tok = StubTokenizer()
            dl = DataLoader(ds, sampler=SequentialSampler(ds), batch_size=2, collate_fn=Collator(24, tok, stream=2))
            em, stem_em, inc_em, total = tr.evaluate(model, ds, dl, tok, opt, str(tmp_path), stop_words=["is", "a"])
            assert total == 3 and 0.0 <= em <= inc_em <= 1.0 and 0.0 <= stem_em <= 1.0
            for ex in ds.data:
                n = min(opt.n_context, len(ex["fact"]))
                got = [f["score"] for f in ex["fact"][:n]]
                assert all(0.0 <= s <= 1.0 for s in got) and sum(got) <= 1.0 + 1e-9
                if ans_attention == "yes" or n == opt.n_context:
                    assert abs(sum(got) - 1.0) < 1e-9           # softmax over exactly the scored facts
                assert all("score" not in f for f in ex["fact"][n:])
    res = os.listdir(tmp_path / "test_results")
    assert len(res) == 1 and res[0].startswith("okvqa_tiny_batch_2_maxLen_24_stream_2_content_2_")
    rows = json.load(open(tmp_path / "test_results" / res[0]))
    assert len(rows) == 3 and {"question", "answer", "real answers", "score", "include_score", "stem_score"} <= set(rows[0])
    assert tr.scores_file_name(opt) == "dev_full_attention_of_tiny_with_max_v1.json"


def test_rank_metrics_match_reference():
    """the retriever evaluation's ranking metrics (src/evaluation.py:200-232) against values of the reference's own functions"""
    import json
    from lako_amd.evaluation import eval_batch
    rows = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "rank_metrics.json")))
    for r in rows:
        ks = [k for k in (1, 2, 5) if k <= len(r["scores"])]
        inv, avg, idx = [], {k: [] for k in ks}, {k: [] for k in ks}
        eval_batch([r["scores"]], inv, avg, idx)
        assert inv == r["inversions"]
        assert {str(k): v for k, v in avg.items()} == r["avg_topk"]
        assert {str(k): v for k, v in idx.items()} == r["idx_topk"]
