#!/usr/bin/env python
"""Retriever training driver — the counterpart of the reference's `train_retriever.py`: KL distillation of the
reader's per-fact cross-attention scores (the `score` fields test_reader.py --write_crossattention_scores leaves in the data) into
the BERT bi-encoder.  Same flags (src/options.py: base + retriever + optim), same step order (train_retriever.py:57-71: forward
with gold scores → backward → clip → optimizer → scheduler → zero_grad), same evaluation (KL loss, inversions, top-k overlap:
:113-153), same checkpoint layout (src/util.save) — on `lako_amd.Retriever`, whose forward AND backward run on the HIP kernels.

  python train_retriever.py --train_data train.json --eval_data dev.json --tokenizer <local bert tokenizer dir> …
  python train_retriever.py --synthetic 8,20,40,130 --steps 20          # B, n_context, question length, passage length (no files)
"""
from __future__ import annotations

import json
import logging
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from lako_amd import util as U  # noqa: E402
from lako_amd.evaluation import eval_batch  # noqa: E402
from lako_amd.options import Options  # noqa: E402
from lako_amd.retriever import Retriever, RetrieverConfig  # noqa: E402

logger = logging.getLogger("train_retriever")


def synthetic_loader(cfg, shape, n_batches, seed):
    """(idx, question_ids, question_mask, passage_ids, passage_mask, gold_score) batches of random ids, valid-first masks and softmax
    gold scores — the shapes RetrieverCollator produces"""
    B, n, ql, pl = shape
    g = torch.Generator().manual_seed(seed)

    def one(rows, L):
        ids = torch.randint(1, cfg.vocab_size, (rows, L), generator=g)
        lens = torch.randint(max(1, L // 3), L + 1, (rows,), generator=g)
        m = torch.arange(L)[None, :] < lens[:, None]
        return ids * m, m
    out = []
    for _ in range(n_batches):
        qi, qm = one(B, ql)
        pi, pm = one(B * n, pl)
        out.append((torch.arange(B), qi, qm, pi.view(B, n, pl), pm.view(B, n, pl), torch.softmax(torch.randn(B, n, generator=g) * 2, -1)))
    return out


def evaluate(model, loader, opt):
    """train_retriever.py:113-153 (the reference returns the LAST batch's loss; here the mean over the batches)"""
    model.eval()
    avg_topk = {k: [] for k in (1, 2, 5) if k <= opt.n_context}
    idx_topk = {k: [] for k in (1, 2, 5) if k <= opt.n_context}
    inversions, losses = [], []
    with torch.no_grad():
        for (_, qi, qm, pi, pm, gold) in loader:
            _, _, scores, loss = model(qi.cuda(), qm.cuda(), pi.cuda(), pm.cuda(), gold_score=gold.cuda())
            eval_batch(scores.cpu().tolist(), inversions, avg_topk, idx_topk)
            losses.append(float(loss))
    return float(np.mean(losses)), float(np.mean(inversions)), {k: float(np.mean(v)) for k, v in avg_topk.items()}, \
        {k: float(np.mean(v)) for k, v in idx_topk.items()}


def main(argv=None):
    options = Options()
    options.add_retriever_options()
    options.add_optim_options()
    opt = options.parse(argv)
    logging.basicConfig(level=logging.INFO, format="[%(asctime)s] %(message)s", stream=sys.stderr)
    torch.manual_seed(opt.seed)
    torch.cuda.set_device(opt.gpu)
    opt.device = opt.gpu
    opt.is_main, opt.is_distributed, opt.world_size = True, False, 1
    if opt.asymmetric_retri == "yes":
        opt.no_projection = True
    dtype = torch.bfloat16 if opt.dtype == "bf16" else torch.float32
    cfg = RetrieverConfig(indexing_dimension=opt.indexing_dimension, apply_question_mask=not opt.no_question_mask,
                          apply_passage_mask=not opt.no_passage_mask, extract_cls=opt.extract_cls, projection=not opt.no_projection,
                          asymmetric_retri=opt.asymmetric_retri, num_hidden_layers=opt.retriever_layers,
                          question_maxlength=opt.question_maxlength, passage_maxlength=opt.passage_maxlength)
    dir_path = os.path.join(opt.checkpoint_dir, opt.name)
    os.makedirs(dir_path, exist_ok=True)
    global_step, best_eval_loss = 0, float("inf")
    if opt.model_path == "none":
        model = Retriever(cfg, dtype=dtype, seed=opt.seed)            # (the reference starts from bert-base-uncased: load it with
        model.set_dropout(opt.dropout)                                #  model.load_state_dict(...) when the weights are on disk)
        model = model.cuda()
    else:
        model, _, _, _, global_step, best_eval_loss = U.load(Retriever, opt.model_path, opt, reset_params=True, dtype=dtype)
        model.set_dropout(opt.dropout)
        logger.info("model loaded from %s (step %d)", opt.model_path, global_step)

    if opt.synthetic:
        shape = tuple(int(v) for v in opt.synthetic.split(","))
        opt.n_context = shape[1]
        train_batches = synthetic_loader(cfg, shape, 16, opt.seed)
        dev_batches = synthetic_loader(cfg, shape, 2, opt.seed + 1)
    else:
        from transformers import AutoTokenizer
        from lako_amd.data import Dataset, RetrieverCollator
        tok = AutoTokenizer.from_pretrained(opt.tokenizer or "bert-base-uncased")
        coll = RetrieverCollator(tok, passage_maxlength=opt.passage_maxlength, question_maxlength=opt.question_maxlength)
        opt.fact_use_way = "separate"                                  # the retriever sees one passage per fact sentence
        with open(opt.train_data) as f:
            train_ds = Dataset(json.load(f), opt)
        with open(opt.eval_data) as f:
            dev_ds = Dataset(json.load(f), opt)
        mk = lambda ds, shuffle: torch.utils.data.DataLoader(ds, batch_size=opt.per_gpu_batch_size, shuffle=shuffle, drop_last=shuffle,   # noqa: E731
                                                             num_workers=4, collate_fn=coll)
        train_batches, dev_batches = mk(train_ds, True), mk(dev_ds, False)
    steps_per_epoch = max(1, len(train_batches))
    opt.total_steps = opt.steps if opt.steps else steps_per_epoch * opt.epochs
    opt.warmup_steps = int(opt.total_steps * 0.06)                      # train_retriever.py:261-262
    optimizer, scheduler = U.set_optim(opt, model)

    model.train()
    epoch, patience, curr_loss, t0, seen = 0, 0, 0.0, time.time(), 0
    done = False
    while epoch < opt.epochs and not done:
        epoch += 1
        for (_, qi, qm, pi, pm, gold) in train_batches:
            global_step += 1
            loss = model(question_ids=qi.cuda(), question_mask=qm.cuda(), passage_ids=pi.cuda(), passage_mask=pm.cuda(),
                         gold_score=gold.cuda())[3]
            loss.backward()
            U.clip_grad_norm_(model, opt.clip)
            optimizer.step()
            scheduler.step()
            model.zero_grad()
            curr_loss += float(loss)
            seen += qi.shape[0]
            if opt.steps and global_step >= opt.steps:
                done = True
                break
        patience += 1
        eval_loss, inv, avg_topk, idx_topk = evaluate(model, dev_batches, opt)
        dt = time.time() - t0
        log = f"epoch {epoch} step {global_step} / {opt.total_steps} -- train: {curr_loss / steps_per_epoch:.6f}, eval: {eval_loss:.6f}, " \
              f"inv: {inv:.1f}, lr: {scheduler.get_last_lr()[0]:.6f}, {seen / dt:.1f} questions/s"
        log += "".join(f" | avg top{k}: {100 * v:.1f}" for k, v in avg_topk.items())
        log += "".join(f" | idx top{k}: {v:.1f}" for k, v in idx_topk.items())
        logger.info(log)
        curr_loss = 0.0
        if eval_loss < best_eval_loss:
            patience, best_eval_loss = 0, eval_loss
            U.save(model, optimizer, scheduler, global_step, best_eval_loss, opt, dir_path, "best_dev")
        if patience > opt.early_stop:
            logger.info("early stop in epoch %d", epoch)
            break
        model.train()
    logger.info("stop epoch %d | best_eval_loss: %.4f", epoch, best_eval_loss)
    return best_eval_loss


if __name__ == "__main__":
    main()
