"""Answer metrics of the reader drivers (reference: src/evaluation.py:130-194, itself the SQuAD / DPR answer
normalisation).  They run on the host over detokenised strings after greedy decoding (train_reader.py:147-160) and
are not on the throughput path; restated here so that `evaluate()` reports the reference's number.

  normalize_answer(s)            lower → strip punctuation → drop the articles a/an/the → squeeze whitespace
  exact_match_score(p, g, v)     v if the normalised strings are equal else 0
  includ_match_score(p, g, v)    v if either normalised string contains the other else 0
  ems / includ_ems(p, golds)     max over the gold dict {answer: soft score}  (OKVQA: score = min(1, #annotators/3))
  stem_ems(p, golds, tok, stem)  first gold (by descending score) that shares a stemmed token with the prediction

`normalize_answer(..., dele_sw=True)` needs the reference's ad-hoc stop-word table; pass it as `stop_words=` (the
reference only uses that mode inside stem_ems, which train_reader.py keeps commented out).
"""
from __future__ import annotations

import re
import string

_ARTICLES = re.compile(r"\b(a|an|the)\b")
_PUNCT = set(string.punctuation)


def normalize_answer(s: str, dele_sw: bool = False, stop_words=None) -> str:
    text = s.lower()
    text = "".join(ch for ch in text if ch not in _PUNCT)
    text = _ARTICLES.sub(" ", text)
    if dele_sw:
        if stop_words is None:
            raise ValueError("dele_sw=True needs the stop-word table (stop_words=...)")
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
