"""Deterministic whitespace tokenizer used to pin the collator (no SentencePiece model is available offline).
Implements both the 3.0.2-era `batch_encode_plus(..., pad_to_max_length=True)` the reference calls and the
modern `__call__(..., padding=...)`.  ids: 0 pad, 1 '</s>', 5 '.', 10 ':' (the conventions src/model.py relies on,
SURVEY.md A.4), other words 11 + crc32 % 50."""
import zlib

import torch


class StubTokenizer:
    legacy_api = True

    def _ids(self, text):
        out = []
        for w in text.replace(":", " : ").replace(".", " . ").split():
            out.append({"</s>": 1, ".": 5, ":": 10}.get(w, 11 + zlib.crc32(w.encode()) % 50))
            self._seen[out[-1]] = w
        return out

    _seen: dict = {}

    def batch_decode(self, sequences, skip_special_tokens=True):
        """ids → text through the words seen so far (unknown ids print as <id>); 0 / 1 are pad / EOS."""
        texts = []
        for seq in sequences:
            words = []
            for t in (seq.tolist() if hasattr(seq, "tolist") else seq):
                if skip_special_tokens and t in (0, 1):
                    continue
                words.append(self._seen.get(int(t), f"<{int(t)}>"))
            texts.append(" ".join(words))
        return texts

    def batch_encode_plus(self, texts, max_length=None, pad_to_max_length=True, return_tensors="pt", truncation=False):
        seqs = [self._ids(t) for t in texts]
        if truncation and max_length:
            seqs = [s[:max_length] for s in seqs]
        width = max_length if max_length else max(len(s) for s in seqs)
        ids = torch.zeros(len(seqs), width, dtype=torch.long)
        mask = torch.zeros(len(seqs), width, dtype=torch.long)
        for i, s in enumerate(seqs):
            ids[i, :len(s)] = torch.tensor(s, dtype=torch.long)
            mask[i, :len(s)] = 1
        return {"input_ids": ids, "attention_mask": mask}

    def __call__(self, texts, max_length=None, padding="longest", truncation=False, return_tensors="pt"):
        return self.batch_encode_plus(texts, max_length=max_length if padding == "max_length" else None,
                                      truncation=truncation)
