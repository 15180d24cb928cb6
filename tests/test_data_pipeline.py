"""Reader input pipeline (lako_amd/data.py) against golden outputs of the reference's own Dataset + Collator
(src/data.py) run with the same deterministic stub tokenizer (tests/golden/collate.npz)."""
import types

import numpy as np
import pytest
import torch

from lako_amd.data import Collator, Dataset
from tests.stub_tokenizer import StubTokenizer

EXAMPLES = [
    {"question": "what is the man holding", "target": "umbrella", "answer": {"umbrella": 1.0}, "img_id": 1,
     "caption": "a man in the rain . street sign", "fact": [{"sentence": "umbrella is used for rain .", "id": 3},
                                                            {"sentence": "rain is wet .", "id": 9},
                                                            {"sentence": "a man is a person .", "id": 4}]},
    {"question": "which sport is this", "target": "tennis", "answer": {"tennis": 1.0, "badminton": 0.3}, "img_id": 2,
     "caption": "two people with rackets", "fact": [{"sentence": "racket is used for tennis .", "id": 5},
                                                    {"sentence": "tennis is a sport .", "id": 6}]},
    {"question": "what animal is shown", "answers": ["cat"], "answer": {"cat": 1.0}, "img_id": 3,
     "caption": "a cat on a sofa", "fact": [{"sentence": "cat is a pet .", "id": 7}]},
]


@pytest.mark.parametrize("stream", [1, 2])
@pytest.mark.parametrize("use_fact", ["yes", "no"])
@pytest.mark.parametrize("ans_len", [-1, 3])
def test_collator_matches_reference(golden_dir, stream, use_fact, ans_len):
    z = np.load(golden_dir + "/collate.npz")
    opt = types.SimpleNamespace(n_context=2, fact_use_way="concate", use_fact=use_fact)
    ds = Dataset(EXAMPLES, opt)
    col = Collator(12, StubTokenizer(), answer_maxlength=ans_len, stream=stream)
    index, tid, tmask, pid, pmask = col([ds[i] for i in range(len(ds))])
    key = f"s{stream}_{use_fact}_{ans_len}"
    assert index.tolist() == z[key + "/index"].tolist()
    assert tid.dtype == torch.int64 and tid.tolist() == z[key + "/target_ids"].tolist()
    assert tmask.dtype == torch.bool and tmask.tolist() == z[key + "/target_mask"].tolist()
    assert pid.shape == z[key + "/passage_ids"].shape and pid.tolist() == z[key + "/passage_ids"].tolist()
    assert pmask.dtype == torch.bool and pmask.tolist() == z[key + "/passage_masks"].tolist()
    # conventions the score aggregation relies on (src/model.py:102,127,178): EOS appended to targets, −100 padding
    assert (tid[:, :].eq(1).sum(1) >= (1 if ans_len < 0 else 0)).all() and (tid[~tmask] == -100).all()


def test_separate_facts_give_one_passage_per_fact():
    """`--fact_use_way separate` (a TODO in the reference, src/data.py:139-141): FiD layout, N = 1 + facts."""
    opt = types.SimpleNamespace(n_context=3, fact_use_way="separate", use_fact="yes")
    ds = Dataset(EXAMPLES, opt)
    col = Collator(16, StubTokenizer(), answer_maxlength=-1, stream=2)
    _, _, _, pid, pmask = col([ds[i] for i in range(3)])
    assert pid.shape == (3, 4, 16)                      # question+caption passage + up to 3 fact passages
    assert pmask[0].any(1).all() and not pmask[2, 2:].any()   # example 2 has one fact → trailing empty passages


def test_modern_tokenizer_call_path():
    class Modern(StubTokenizer):
        legacy_api = False
    opt = types.SimpleNamespace(n_context=2, fact_use_way="concate", use_fact="yes")
    ds = Dataset(EXAMPLES, opt)
    a = Collator(12, StubTokenizer(), answer_maxlength=-1, stream=2)([ds[i] for i in range(3)])
    b = Collator(12, Modern(), answer_maxlength=-1, stream=2)([ds[i] for i in range(3)])
    for x, y in zip(a, b):
        assert torch.equal(x, y)
