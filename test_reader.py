#!/usr/bin/env python
"""Reader evaluation driver — the counterpart of the reference's test_reader.py (same flags): greedy decode of the
eval set, soft exact match / include match / stem match (test_reader.py:84-91), optional per-example results JSON
(`--write_results`, :93-105,125-127) and the LaKo "late injection" output: step-0 cross-attention scores
aggregated per fact, softmaxed and written back into every example's `fact[j]['score']`
(`--write_crossattention_scores`, :36-38,62-76,107-122,197-214) — the file the retriever distillation consumes.

    python test_reader.py --model_path CKPT_DIR --eval_data dev.json --per_gpu_batch_size 16 --n_context 10 \
        --text_maxlength 200 --stream 2 --write_results --write_crossattention_scores [--tokenizer PATH]

The model path is a directory written by `FiDT5.save_pretrained` / `lako_amd.util.save`.  The stem metric uses
nltk's WordPunctTokenizer + PorterStemmer like the reference when nltk is importable; otherwise a regex tokenizer
and a light suffix stemmer stand in and the log line says so."""
from __future__ import annotations

import json
import logging
import os
import re
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from lako_amd import FiDT5  # noqa: E402
from lako_amd import evaluation as E  # noqa: E402
from lako_amd import util as U  # noqa: E402
from lako_amd.options import Options  # noqa: E402

logger = logging.getLogger("test_reader")


class _RegexTokenizer:
    """nltk.tokenize.WordPunctTokenizer's pattern: runs of word characters or runs of non-space punctuation."""
    _pat = re.compile(r"\w+|[^\w\s]+")

    def tokenize(self, s):
        return self._pat.findall(s)


class _SuffixStemmer:
    exact = False

    def stem(self, w):
        for suf in ("ingly", "edly", "ing", "ies", "ed", "es", "ly", "s"):
            if w.endswith(suf) and len(w) - len(suf) >= 3:
                return w[:-len(suf)]
        return w


def stem_tools():
    try:
        import nltk.stem.porter as pt
        import nltk.tokenize as tk
        st = pt.PorterStemmer()
        st.exact = True
        return tk.WordPunctTokenizer(), st
    except Exception:
        return _RegexTokenizer(), _SuffixStemmer()


def evaluate(model, dataset, dataloader, tokenizer, opt, dir_path, stop_words=None):
    """test_reader.py:31-132.  Returns (em, stem_em, include_em, total)."""
    model.eval()
    if hasattr(model, "module"):
        model = model.module
    if opt.write_crossattention_scores:
        model.overwrite_forward_crossattention()
        model.reset_score_storage()
    exactmatch, stem_exactmatch, include_exactmatch = [], [], []
    result_json = []
    tk_tokenizer, stemmer = stem_tools()
    sw = stop_words          # None → the reference's table (lako_amd.evaluation.STOP_WORDS)
    device = next(model.parameters()).device
    with torch.no_grad():
        for batch in dataloader:
            idx, _, _, context_ids, context_mask = batch
            if opt.write_crossattention_scores:
                model.reset_score_storage()
            outputs = model.generate(input_ids=context_ids.to(device), attention_mask=context_mask.to(device), max_length=50)
            if opt.write_crossattention_scores:
                scores = model.get_crossattention_scores(opt, context_ids, tokenizer, context_mask.to(device))
                scores = scores.cpu() if opt.ans_attention == "yes" else torch.softmax(scores, dim=-1)
            for k, ans in enumerate(tokenizer.batch_decode(outputs, skip_special_tokens=True)):
                example = dataset.data[int(idx[k])]
                gold = example["answer"]
                score = E.ems(ans, gold)
                include_score = E.includ_ems(ans, gold)
                stem_score = E.stem_ems(ans, gold, tk_tokenizer, stemmer, dele_sw=True, stop_words=sw)
                exactmatch.append(float(score))
                stem_exactmatch.append(float(stem_score))
                include_exactmatch.append(float(include_score))
                if opt.write_results:
                    result_json.append({"question": example["question"], "img_id": example.get("img_id"), "answer": ans,
                                        "target": example.get("target"), "real answers": gold,
                                        "fact": example.get("fact", [])[:50], "include_score": include_score, "score": score,
                                        "stem_score": stem_score})
                if opt.write_crossattention_scores:
                    facts = example.get("fact", [])
                    n = min(opt.n_context, len(facts))
                    if opt.ans_attention == "yes":     # answer-aware prior added before the softmax (:109-117)
                        prior = [max(E.includ_ems(f["sentence"], gold),
                                     E.stem_ems(f["sentence"], gold, tk_tokenizer, stemmer, dele_sw=True, stop_words=sw))
                                 for f in facts[:n]]
                        scores[k, :n] += torch.tensor(prior, dtype=scores.dtype)
                        scores[k, :n] = torch.softmax(scores[k, :n], dim=-1)
                    for j in range(n):
                        facts[j]["score"] = scores[k, j].item()
    if opt.write_results:
        now = time.strftime("%m-%d-%H", time.localtime(time.time()))
        fact_para = f"_stream_{opt.stream}_content_{opt.n_context}_" if opt.use_fact == "yes" else ""
        name = (f"{opt.dataset}_{opt.model_size}_batch_{opt.per_gpu_batch_size}_maxLen_{opt.text_maxlength}"
                f"{fact_para}{now}.json")
        os.makedirs(os.path.join(dir_path, "test_results"), exist_ok=True)
        with open(os.path.join(dir_path, "test_results", name), "w") as f:
            json.dump(result_json, f)
    total = len(exactmatch)
    mean = lambda xs: sum(xs) / max(len(xs), 1)   # noqa: E731
    em, _ = U.weighted_average(mean(exactmatch), total, opt)
    stem_em, _ = U.weighted_average(mean(stem_exactmatch), total, opt)
    inc_em, total = U.weighted_average(mean(include_exactmatch), total, opt)
    return em, stem_em, inc_em, total


def scores_file_name(opt):
    """test_reader.py:199-211."""
    prefix = os.path.basename(opt.eval_data).replace(".json", "")
    half = "last_half_layer_attention" if opt.use_last_half_layer_attention == "yes" else "full_attention"
    return f"{prefix}_{half}_of_{opt.model_size}_with_{opt.attention_score_style}_{opt.version}.json"


def main():
    opt = Options().add_reader_options().add_eval_options().add_optim_options().parse()
    opt.world_size, opt.global_rank, opt.is_distributed, opt.is_main = 1, 0, False, True
    torch.cuda.set_device(opt.gpu)
    opt.device = torch.device("cuda", opt.gpu)
    logging.basicConfig(level=logging.INFO, format="[%(asctime)s] %(message)s")
    import transformers
    from torch.utils.data import DataLoader, SequentialSampler

    from lako_amd.data import Collator, Dataset
    try:
        tokenizer = transformers.T5Tokenizer.from_pretrained(opt.tokenizer or ("t5-" + opt.model_size))
    except Exception as e:
        raise SystemExit(f"cannot load a T5 tokenizer ({e}); pass --tokenizer /path/to/t5-tokenizer")
    dir_path = opt.checkpoint_dir
    model = FiDT5.from_pretrained(os.path.realpath(opt.model_path), dtype=torch.bfloat16 if opt.dtype == "bf16" else torch.float32,
                                  legacy_cross_bias=True if opt.legacy_cross_bias else None)
    model = model.cuda(opt.gpu)
    with open(opt.eval_data) as f:
        dataset = Dataset(json.load(f), opt)
    collator = Collator(opt.text_maxlength, tokenizer, answer_maxlength=opt.answer_maxlength, stream=opt.stream)
    loader = DataLoader(dataset, sampler=SequentialSampler(dataset), batch_size=opt.per_gpu_batch_size, num_workers=2,
                        collate_fn=collator)
    logger.info("Start eval")
    em, stem_em, inc_em, total = evaluate(model, dataset, loader, tokenizer, opt, dir_path)
    exact = getattr(stem_tools()[1], "exact", False)
    logger.info(f"evaluation: {100 * em:.2f}EM | include: {100 * inc_em:.2f}EM| stem: {100 * stem_em:.2f}EM"
                f"{'' if exact else ' (suffix stemmer: nltk absent)'} , Total number of example {total}|")
    if opt.write_crossattention_scores:
        out_dir = os.path.join(dir_path, "tmp_dir")
        os.makedirs(out_dir, exist_ok=True)
        save_name = os.path.join(out_dir, scores_file_name(opt))
        with open(save_name, "w") as fw:
            json.dump(dataset.data, fw)
        logger.info(f"finish. save to: {save_name}")


if __name__ == "__main__":
    main()
