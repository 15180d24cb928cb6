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
        yield synthetic_batch(B, N, L, T, cfg.vocab_size, seed=opt.seed + opt.global_rank * 7919 + i, device=device)


def json_loader(opt, path, tokenizer, device, shuffle):
    """train_reader.py:40-48 / :123-131: DataLoader over the reference's JSON examples."""
    import json

    from torch.utils.data import DataLoader, RandomSampler, SequentialSampler

    from lako_amd.data import Collator, Dataset
    with open(path) as f:
        ds = Dataset(json.load(f), opt)
    col = Collator(opt.text_maxlength, tokenizer, answer_maxlength=opt.answer_maxlength, stream=opt.stream)
    dl = DataLoader(ds, sampler=RandomSampler(ds) if shuffle else SequentialSampler(ds),
                    batch_size=opt.per_gpu_batch_size, drop_last=shuffle, num_workers=2, collate_fn=col)
    for _, labels, _, ids, mask in dl:
        yield ids.to(device, non_blocking=True), mask.to(device, non_blocking=True), labels.to(device, non_blocking=True)


def evaluate(model, batches, opt):
    """Greedy decode + exact match of the generated ids against the label ids (train_reader.py:123-169; the string
    metrics of src/evaluation.py operate on detokenised text and are out of the hot path)."""
    model.eval()
    hit = total = 0
    with torch.no_grad():
        for ids, mask, labels in batches:
            out = model.generate(input_ids=ids, attention_mask=mask, max_length=50)
            for b in range(ids.shape[0]):
                gold = [t for t in labels[b].tolist() if t not in (-100, 0, 1)]
                pred = [t for t in out[b].tolist() if t not in (0, 1)]
                hit += int(gold == pred)
                total += 1
    model.train()
    score, total = U.weighted_average(hit / max(total, 1), total, opt)
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
    cfg = FiDConfig.named(opt.model_size, dropout_rate=opt.dropout)
    dtype = torch.bfloat16 if opt.dtype == "bf16" else torch.float32
    if opt.model_path == "none":
        model = FiDT5(cfg, dtype=dtype, seed=opt.seed + opt.global_rank)
        with torch.no_grad():
            model._params_by_plain["shared.weight"].mul_(0.05)      # random-init stand-in for t5-* weights
    else:
        model = FiDT5.from_pretrained(opt.model_path, dtype=dtype, seed=opt.seed + opt.global_rank)
    model = model.cuda(local_rank)
    model.set_checkpoint(opt.use_checkpoint)

    def train_batches(n):
        if tokenizer is None:
            return synthetic_loader(opt, cfg, opt.device, n)
        return json_loader(opt, opt.train_data, tokenizer, opt.device, True)

    def eval_batches():
        if tokenizer is None:
            return list(synthetic_loader(opt, cfg, opt.device, 2))
        return json_loader(opt, opt.eval_data, tokenizer, opt.device, False)

    steps_per_epoch = opt.steps or 100
    opt.total_steps = steps_per_epoch * opt.epochs
    opt.warmup_steps = int(opt.total_steps * 0.06)
    optimizer, scheduler = U.set_optim(opt, model)
    if opt.is_distributed:
        from lako_amd.dist import GradSync, broadcast_parameters
        broadcast_parameters(model)
        GradSync(model)

    torch.manual_seed(opt.global_rank + opt.seed)
    model.train()
    step, best = 0, 0.0
    for epoch in range(1, opt.epochs + 1):
        curr_loss = torch.zeros((), device=opt.device)
        t0 = time.time()
        n = 0
        for ids, mask, labels in train_batches(steps_per_epoch):
            step += 1
            train_loss = model(input_ids=ids, attention_mask=mask, labels=labels)[0]
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
        dev_em = evaluate(model, eval_batches(), opt)
        if opt.is_main:
            logger.info(f"epoch {epoch} |step {step} |train loss: {curr_loss.item() / max(n, 1):.3f} |"
                        f"evaluation: {100 * dev_em:.2f}EM |lr: {scheduler.get_last_lr()[0]:.5f} |"
                        f"{n * ids.shape[0] * opt.world_size / dt:.1f} samples/s")
            if dev_em > best:
                best = dev_em
                path = os.path.join(opt.checkpoint_dir, opt.name, "checkpoint", "best_dev")
                model.save_pretrained(path)
        if opt.steps and step >= opt.steps:
            break
    if opt.is_distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
