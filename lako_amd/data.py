"""Reader input pipeline — restatement of the reader parts of the reference's `src/data.py`
(Dataset :14-85, encode_passages :88-104, Collator :107-148).

Example dict (the JSON schema `data_process/vqa2_deal.py:132-143` writes):
  {question, target | answers, answer{str: weight}, img_id, caption, fact[{sentence, id[, score]}]}
→ prefixed strings `question: …`, `context: …` (image caption/OCR), `fact: s1 s2 … ` → token tensors
  (index [B], target_ids [B,T] int64 with −100 on padding, target_mask bool, passage_ids [B,N,L] int64,
   passage_masks [B,N,L] bool) — exactly what `FiDT5.forward/generate` consume.

stream 1: one passage  (question + caption + facts);  stream 2: two passages [question + caption, facts]
(src/data.py:130-139).  `fact_use_way="separate"` is a TODO in the reference (the collator returns None,
:139-141); here it yields FiD's classic layout — one passage per fact, each prefixed with the question —
which is the N = 20 … 100 regime of the BASELINE configs.

The tokenizer is duck-typed: anything callable like a HuggingFace tokenizer
(`tok(texts, max_length=…, padding=…, truncation=…, return_tensors="pt")`) or exposing the 3.0.2-era
`batch_encode_plus(texts, max_length=…, pad_to_max_length=True, return_tensors="pt", truncation=…)`.
"""
from __future__ import annotations

import random

import torch


class Dataset(torch.utils.data.Dataset):
    def __init__(self, data, opt, question_prefix="question:", caption_prefix="context:", fact_prefix="fact:"):
        self.data = data
        self.n_context = opt.n_context
        self.question_prefix, self.caption_prefix, self.fact_prefix = question_prefix, caption_prefix, fact_prefix
        self.fact_use_way = opt.fact_use_way
        self.use_fact = opt.use_fact

    def __len__(self):
        return len(self.data)

    def get_target(self, example):
        # the 3.0.2 T5 tokenizer did not append EOS, hence the explicit ' </s>' (src/data.py:34-41)
        if "target" in example:
            return example["target"] + " </s>"
        if "answers" in example:
            return random.choice(example["answers"]) + " </s>"
        return None

    def __getitem__(self, index):
        ex = self.data[index]
        question = self.question_prefix + " " + ex["question"]
        caption = self.caption_prefix + " " + ex["caption"]
        fact, scores = None, None
        if self.use_fact == "yes":
            contexts = ex["fact"][:self.n_context]
            sentences = [c["sentence"] for c in contexts]
            if self.fact_use_way == "concate":
                fact = self.fact_prefix + " " + " ".join(sentences) + " "
            else:
                fact = sentences
            if contexts and "score" in contexts[0]:
                scores = torch.tensor([float(c["score"]) for c in contexts])
        return {"index": index, "question": question, "caption": caption, "target": self.get_target(ex),
                "answer": ex["answer"], "fact": fact, "score": scores}

    def get_example(self, index):
        return self.data[index]


def _encode(tokenizer, texts, max_length, truncation):
    if hasattr(tokenizer, "batch_encode_plus") and getattr(tokenizer, "legacy_api", False):
        return tokenizer.batch_encode_plus(texts, max_length=max_length, pad_to_max_length=True, return_tensors="pt",
                                           truncation=truncation)
    try:
        return tokenizer(texts, max_length=max_length, padding="max_length" if max_length else "longest",
                         truncation=truncation, return_tensors="pt")
    except TypeError:
        return tokenizer.batch_encode_plus(texts, max_length=max_length, pad_to_max_length=True, return_tensors="pt",
                                           truncation=truncation)


def encode_passages(batch_text_passages, tokenizer, max_length):
    ids, masks = [], []
    for passages in batch_text_passages:
        p = _encode(tokenizer, passages, max_length, True)
        ids.append(p["input_ids"][None])
        masks.append(p["attention_mask"][None])
    return torch.cat(ids, dim=0), torch.cat(masks, dim=0).bool()


class Collator:
    def __init__(self, text_maxlength, tokenizer, answer_maxlength=20, stream=2, fact_prefix="fact:"):
        self.tokenizer, self.text_maxlength, self.answer_maxlength, self.stream = (tokenizer, text_maxlength,
                                                                                   answer_maxlength, stream)
        self.fact_prefix = fact_prefix

    def passages_of(self, ex):
        head = ex["question"] + " " + ex["caption"]
        if ex["fact"] is None:
            return [head]
        if isinstance(ex["fact"], str):
            return [head + " " + ex["fact"]] if self.stream == 1 else [head, ex["fact"]]
        # one passage per fact (FiD layout; unimplemented in the reference, src/data.py:139-141)
        return [head] + [ex["question"] + " " + self.fact_prefix + " " + s for s in ex["fact"]]

    def __call__(self, batch):
        index = torch.tensor([ex["index"] for ex in batch])
        tgt = _encode(self.tokenizer, [ex["target"] for ex in batch],
                      self.answer_maxlength if self.answer_maxlength > 0 else None, self.answer_maxlength > 0)
        target_mask = tgt["attention_mask"].bool()
        target_ids = tgt["input_ids"].masked_fill(~target_mask, -100)
        passages = [self.passages_of(ex) for ex in batch]
        n = max(len(p) for p in passages)
        passages = [p + [""] * (n - len(p)) for p in passages]          # ragged "separate" batches: empty passages
        passage_ids, passage_masks = encode_passages(passages, self.tokenizer, self.text_maxlength)
        return index, target_ids, target_mask, passage_ids, passage_masks


class RetrieverCollator:
    """src/data.py:178-211: question + caption → question ids / mask; the example's fact sentences → passage ids / masks [B, n, L];
    the reader's per-fact scores as gold scores.  (index, question_ids, question_mask, passage_ids, passage_masks, scores) —
    the batch train_retriever.py:57-66 feeds to `Retriever.forward`."""

    def __init__(self, tokenizer, passage_maxlength=140, question_maxlength=140):
        self.tokenizer, self.passage_maxlength, self.question_maxlength = tokenizer, passage_maxlength, question_maxlength

    def __call__(self, batch):
        index = torch.tensor([ex["index"] for ex in batch])
        q = _encode(self.tokenizer, [ex["question"] + " " + ex["caption"] for ex in batch], self.question_maxlength, True)
        question_ids, question_mask = q["input_ids"], q["attention_mask"].bool()
        if batch[0]["score"] is None or batch[0]["fact"] is None:
            return index, question_ids, question_mask, None, None, None
        scores = torch.stack([ex["score"] for ex in batch], dim=0)
        passage_ids, passage_masks = encode_passages([ex["fact"] for ex in batch], self.tokenizer, self.passage_maxlength)
        return index, question_ids, question_mask, passage_ids, passage_masks, scores
