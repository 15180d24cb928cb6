#!/usr/bin/env python
"""Reader training driver — the counterpart of the reference's train_reader.py (same flags, same step order:
train_reader.py:62-84  fwd → bwd → clip_grad_norm_(opt.clip) → optimizer.step → scheduler.step → zero_grad,
per-epoch greedy-decode evaluation :123-169, warm-up = 6 % of total steps :259-261, seed rule :39), running the
lako_amd.FiDT5 HIP path.  One process per GPU; under `python -m torch.distributed.run --nproc-per-node N`
gradients are all-reduced over RCCL (the reference itself never synchronises gradients — SURVEY.md §0.3-4).

    python train_reader.py --model_size base --per_gpu_batch_size 16 --n_context 10 --text_maxlength 200 \
        --optim adamw --scheduler linear --weight_decay 1e-4 --lr 4e-5 --epochs 1 --synthetic 16,20,200,8 --steps 20

Data: `--synthetic B,N,L,T` or the reference's JSON files through lako_amd.data (Dataset/Collator
restated from src/data.py; needs a T5 tokenizer on disk: `--tokenizer PATH`, default `t5-<model_size>`)."""
from __future__ import annotations

import logging
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from lako_amd import FiDConfig, FiDT5  # noqa: E402
from lako_amd import util as U  # noqa: E402
from lako_amd.options import Options  # noqa: E402

logger = logging.getLogger("train_reader")


def synthetic_loader(opt, cfg, device, n_batches):
    from bench import synthetic_batch
    B, N, L, T = (int(x) for x in opt.synthetic.split(","))
    assert B == opt.per_gpu_batch_size, "--synthetic B must equal --per_gpu_batch_size"
    for i in range(n_batches):
        ids, mask, labels, lens = synthetic_batch(B, N, L, T, cfg.vocab_size, seed=opt.seed + opt.global_rank * 7919 + i,
                                                  device=device, with_lengths=True)
        yield dict(idx=None, ids=ids, mask=mask, labels=labels, lens=lens)


def json_dataset(opt, path):
    import json

    from lako_amd.data import Dataset
    with open(path) as f:
        return Dataset(json.load(f), opt)


def json_loader(opt, ds, tokenizer, device, shuffle, with_index=False):
    """train_reader.py:40-48 / :123-131: DataLoader over the reference's JSON examples."""
    from torch.utils.data import DataLoader, RandomSampler, SequentialSampler

    from lako_amd.data import Collator
    col = Collator(opt.text_maxlength, tokenizer, answer_maxlength=opt.answer_maxlength, stream=opt.stream)
    dl = DataLoader(ds, sampler=RandomSampler(ds) if shuffle else SequentialSampler(ds),
                    batch_size=opt.per_gpu_batch_size, drop_last=shuffle, num_workers=2, collate_fn=col)
    for idx, labels, _, ids, mask in dl:
        # the mask is born on the host (src/data.py:88-104: pad to text_maxlength): its per-passage lengths go along, so the
        # unpadded encoder needs no device→host read-back; a mask that is not "valid tokens first" gets no lengths
        lens = mask.sum(-1).to(torch.int32)
        prefix = bool((mask == (torch.arange(mask.shape[-1])[None, None, :] < lens[..., None])).all())
        yield dict(idx=idx if with_index else None, ids=ids.to(device, non_blocking=True),
                   mask=mask.to(device, non_blocking=True), labels=labels.to(device, non_blocking=True),
                   lens=lens if prefix else None)


def evaluate(model, batches, opt, tokenizer=None, dataset=None):
    """Greedy decode (max_length 50) and score (train_reader.py:123-169).  With a tokenizer and the JSON dataset:
    the reference's soft exact match `ems(decoded answer, example['answer'])` (src/evaluation.py:158).  On synthetic
    batches (no tokenizer, no strings): exact match of the generated ids against the label ids."""
    from lako_amd import evaluation as E
    model.eval()
    scores = []
    with torch.no_grad():
        for batch in batches:
            idx, ids, mask, labels = batch["idx"], batch["ids"], batch["mask"], batch["labels"]
            out = model.generate(input_ids=ids, attention_mask=mask, max_length=50, passage_lengths=batch["lens"])
            if tokenizer is not None and dataset is not None:
                for k, ans in enumerate(tokenizer.batch_decode(out, skip_special_tokens=True)):
                    scores.append(float(E.ems(ans, dataset.get_example(int(idx[k]))["answer"])))
            else:
                for b in range(ids.shape[0]):
                    gold = [t for t in labels[b].tolist() if t not in (-100, 0, 1)]
                    pred = [t for t in out[b].tolist() if t not in (0, 1)]
                    scores.append(float(gold == pred))
    model.train()
    score, _ = U.weighted_average(sum(scores) / max(len(scores), 1), len(scores), opt)
    return score


def main():
    opt = Options().add_reader_options().add_optim_options().parse()
    opt.world_size = int(os.environ.get("WORLD_SIZE", "1"))
    opt.global_rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", opt.gpu if opt.local_rank < 0 else opt.local_rank))
    opt.is_distributed = opt.world_size > 1
    opt.is_main = opt.global_rank == 0
    torch.cuda.set_device(local_rank)
    opt.device = torch.device("cuda", local_rank)
    if opt.is_distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=opt.device)
    logging.basicConfig(level=logging.INFO if opt.is_main else logging.WARN, format="[%(asctime)s] %(message)s")
    torch.manual_seed(opt.seed)

    tokenizer = None
    if opt.synthetic is None:
        if opt.train_data == "none":
            raise SystemExit("pass --synthetic B,N,L,T or --train_data/--eval_data JSON files")
        import transformers
        try:
            tokenizer = transformers.T5Tokenizer.from_pretrained(opt.tokenizer or ("t5-" + opt.model_size))
        except Exception as e:      # no network in the build container: a local tokenizer directory is required
            raise SystemExit(f"cannot load a T5 tokenizer ({e}); pass --tokenizer /path/to/t5-tokenizer or use --synthetic")
    cfg = FiDConfig.named(opt.model_size, dropout_rate=opt.dropout, legacy_cross_bias=opt.legacy_cross_bias)
    legacy = True if opt.legacy_cross_bias else None       # (None: a checkpoint's own config.json decides)
    dtype = torch.bfloat16 if opt.dtype == "bf16" else torch.float32
    train_ds = eval_ds = None
    if tokenizer is not None:
        train_ds, eval_ds = json_dataset(opt, opt.train_data), json_dataset(opt, opt.eval_data)

    def train_batches(n):
        if tokenizer is None:
            return synthetic_loader(opt, cfg, opt.device, n)
        return json_loader(opt, train_ds, tokenizer, opt.device, True)

    def eval_batches():
        if tokenizer is None:
            return list(synthetic_loader(opt, cfg, opt.device, 2))
        return json_loader(opt, eval_ds, tokenizer, opt.device, False, with_index=True)

    # train_reader.py:255-262: steps/epoch from the dataset, warm-up = 6 % of all steps
    steps_per_epoch = opt.steps or (len(train_ds) // opt.per_gpu_batch_size if train_ds is not None else 100)
    opt.total_steps = steps_per_epoch * opt.epochs
    opt.warmup_steps = int(opt.total_steps * 0.06)
    step, best = 0, 0.0
    if opt.model_path == "none":
        model = FiDT5(cfg, dtype=dtype, seed=opt.seed + opt.global_rank)
        with torch.no_grad():
            model._params_by_plain["shared.weight"].mul_(0.05)      # random-init stand-in for t5-* weights
        model = model.cuda(local_rank)
        optimizer, scheduler = U.set_optim(opt, model)
    elif os.path.exists(os.path.join(os.path.realpath(opt.model_path), "optimizer.pth.tar")):
        # a checkpoint directory written by util.save: weights kept, fresh optimizer / scheduler (train_reader.py:255)
        model, optimizer, scheduler, _, step, best = U.load(FiDT5, opt.model_path, opt, reset_params=True, dtype=dtype,
                                                            seed=opt.seed + opt.global_rank, legacy_cross_bias=legacy)
        model = model.cuda(local_rank)
        logger.info(f"model loaded from {opt.model_path} (was at step {step}, best {best})")
        step, best = 0, 0.0       # train_reader.py:266: a warm start, not a resume — counters restart like the reference's
    else:
        model = FiDT5.from_pretrained(opt.model_path, dtype=dtype, seed=opt.seed + opt.global_rank, legacy_cross_bias=legacy).cuda(local_rank)
        optimizer, scheduler = U.set_optim(opt, model)
    model.set_checkpoint(opt.use_checkpoint)
    if opt.is_distributed:
        from lako_amd.dist import GradSync, broadcast_parameters
        broadcast_parameters(model)
        GradSync(model)

    torch.manual_seed(opt.global_rank + opt.seed)
    model.train()
    patience, epoch = 0, 0
    for epoch in range(1, opt.epochs + 1):
        curr_loss = torch.zeros((), device=opt.device)
        t0 = time.time()
        n = 0
        for batch in train_batches(steps_per_epoch):
            step += 1
            ids = batch["ids"]
            train_loss = model(input_ids=ids, attention_mask=batch["mask"], labels=batch["labels"],
                               passage_lengths=batch["lens"])[0]
            train_loss.backward()
            U.clip_grad_norm_(model, opt.clip)
            optimizer.step()
            scheduler.step()
            model.zero_grad()
            train_loss = U.average_main(train_loss.detach(), opt)
            curr_loss += train_loss
            n += 1
            if opt.steps and step >= opt.steps:
                break
        torch.cuda.synchronize()
        dt = time.time() - t0
        patience += 1                       # train_reader.py:86: epochs since the dev metric last improved
        dev_em = evaluate(model, eval_batches(), opt, tokenizer, eval_ds)
        stop = False
        if opt.is_main:
            logger.info(f"epoch {epoch} |step {step} |train loss: {curr_loss.item() / max(n, 1):.3f} |"
                        f"evaluation: {100 * dev_em:.2f}EM |lr: {scheduler.get_last_lr()[0]:.5f} |"
                        f"{n * ids.shape[0] * opt.world_size / dt:.1f} samples/s")
            if dev_em > best:      # train_reader.py:99-108
                patience = 0
                best = dev_em
                U.save(model, optimizer, scheduler, step, best, opt, os.path.join(opt.checkpoint_dir, opt.name), "best_dev")
            if patience > opt.early_stop:       # train_reader.py:112-114
                logger.info(f"early stop in epoch {epoch}")
                stop = True
        if opt.is_distributed:                  # rank 0 decides (only it knows `best`); everyone leaves together
            flag = torch.tensor([int(stop)], device=opt.device)
            dist.broadcast(flag, 0)
            stop = bool(flag.item())
        if stop or (opt.steps and step >= opt.steps):
            break
    if opt.is_main:
        logger.info(f"stop epoch {epoch} |evaluation: {100 * best:.2f}EM |")
    if opt.is_distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
