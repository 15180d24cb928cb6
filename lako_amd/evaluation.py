"""Answer metrics of the reader drivers (reference: src/evaluation.py:130-194, itself the SQuAD / DPR answer
normalisation).  They run on the host over detokenised strings after greedy decoding (train_reader.py:147-160) and
are not on the throughput path; restated here so that `evaluate()` reports the reference's number.

  normalize_answer(s)            lower → strip punctuation → drop the articles a/an/the → squeeze whitespace
  exact_match_score(p, g, v)     v if the normalised strings are equal else 0
  includ_match_score(p, g, v)    v if either normalised string contains the other else 0
  ems / includ_ems(p, golds)     max over the gold dict {answer: soft score}  (OKVQA: score = min(1, #annotators/3))
  stem_ems(p, golds, tok, stem)  first gold (by descending score) that shares a stemmed token with the prediction

`normalize_answer(..., dele_sw=True)` deletes, as SUBSTRINGS, every word of the text that is in the reference's stop-word table
(src/evaluation.py:21-28 — the values ship as data in stop_words.json, written by oracle/make_fixtures.py; `stop_words=`
overrides it).  test_reader.py uses that mode for the stem metric and the answer-aware fact prior (test_reader.py:84-117).
"""
from __future__ import annotations

import json
import os
import re
import string

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "stop_words.json"), encoding="utf-8") as _f:
    STOP_WORDS = frozenset(json.load(_f))

_ARTICLES = re.compile(r"\b(a|an|the)\b")
_PUNCT = set(string.punctuation)


def normalize_answer(s: str, dele_sw: bool = False, stop_words=None) -> str:
    text = s.lower()
    text = "".join(ch for ch in text if ch not in _PUNCT)
    text = _ARTICLES.sub(" ", text)
    if dele_sw:
        stop_words = STOP_WORDS if stop_words is None else stop_words
        for word in text.split():        # the reference deletes every occurrence of a stop word as a SUBSTRING
            if word in stop_words:
                text = text.replace(word, "")
    return " ".join(text.split())


def exact_match_score(prediction, ground_truth, value):
    return (normalize_answer(prediction) == normalize_answer(ground_truth)) * value


def includ_match_score(prediction, ground_truth, value):
    p, g = normalize_answer(prediction), normalize_answer(ground_truth)
    return ((p in g) or (g in p)) * value


def ems(prediction, ground_truths):
    return max([exact_match_score(prediction, k, v) for k, v in ground_truths.items()])


def includ_ems(prediction, ground_truths):
    return max([includ_match_score(prediction, k, v) for k, v in ground_truths.items()])


def stem_ems(prediction, ground_truths, tokenizer, stemmer, dele_sw=False, stop_words=None):
    stem_ans = {stemmer.stem(t) for t in tokenizer.tokenize(normalize_answer(prediction, dele_sw, stop_words))}
    for ground_truth, value in sorted(ground_truths.items(), key=lambda kv: kv[1], reverse=True):
        if any(stemmer.stem(t) in stem_ans for t in tokenizer.tokenize(normalize_answer(ground_truth))):
            return value
    return 0


# ---- ranking metrics of the retriever's evaluation (src/evaluation.py:200-232, train_retriever.py:142) ----------------------
def count_inversions(arr):
    return sum(1 for i in range(len(arr)) for j in range(i + 1, len(arr)) if arr[i] > arr[j])


def score(x, inversions, avg_topk, idx_topk):
    """x: the gold ranks in predicted order.  inversions; per k: the share of the predicted top-k that is in the gold top-k, and the
    number of predicted passages needed to cover the gold top-k."""
    x = list(x)
    inversions.append(count_inversions(x))
    for k in avg_topk:
        avg_topk[k].append(sum(1 for v in x[:k] if v < k) / float(len(x[:k])))
    for k in idx_topk:
        below = [v < k for v in x]
        last = max((i for i, b in enumerate(below) if b), default=len(x) - 1)       # numpy: len − argmax(reversed) = last True + 1
        idx_topk[k].append(last + 1 if any(below) else len(x))


def eval_batch(scores, inversions, avg_topk, idx_topk):
    for s in scores:
        s = [float(v) for v in s]
        order = sorted(range(len(s)), key=lambda i: -s[i])
        score(order, inversions, avg_topk, idx_topk)
