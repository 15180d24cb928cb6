"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference, which never travels to the GPU box):

    python oracle/make_fixtures.py

What runs: the reference's own `src/model.py` (`FiDT5`, `EncoderWrapper`, `CheckpointWrapper`,
`get_crossattention_scores`) and `src/util.py` (`set_optim`, `WarmupLinearScheduler`) imported from
/root/reference, on top of the installed transformers 5.15.0 `modeling_t5.py`.  The reference was
written against transformers 3.0.2 (README.md:21), which cannot be installed offline, so three
harness-side monkeypatches (no edits to the reference) adapt call signatures:
  (i)   CheckpointWrapper.forward accepts the extra positional args HF5 passes to a block,
  (ii)  EncoderWrapper.forward re-wraps its tuple in BaseModelOutput,
  (iii) EncoderWrapper.main_input_name = "input_ids" (for generate()).
`transformers.AdamW` (removed in HF 5) is provided by a harness stub with HF<=4 semantics so that the
reference's own `set_optim` can construct it (`correct_bias=False`, src/util.py:225).

Only DATA is written (inputs and expected outputs); no reference source is copied.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import transformers  # noqa: E402
from transformers.modeling_outputs import BaseModelOutput  # noqa: E402

from oracle import fid_t5_oracle as O  # noqa: E402


class _HFAdamW(torch.optim.Optimizer):
    """Harness stub for the removed `transformers.AdamW` (HF <= 4 semantics)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                      correct_bias=correct_bias))

    @torch.no_grad()
    def step(self, closure=None):
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                b1, b2 = group["betas"]
                st["step"] += 1
                st["exp_avg"].mul_(b1).add_(p.grad, alpha=1.0 - b1)
                st["exp_avg_sq"].mul_(b2).addcmul_(p.grad, p.grad, value=1.0 - b2)
                denom = st["exp_avg_sq"].sqrt().add_(group["eps"])
                step_size = group["lr"]
                if group["correct_bias"]:
                    step_size = step_size * (1.0 - b2 ** st["step"]) ** 0.5 / (1.0 - b1 ** st["step"])
                p.addcdiv_(st["exp_avg"], denom, value=-step_size)
                if group["weight_decay"] > 0.0:
                    p.add_(p, alpha=-group["lr"] * group["weight_decay"])


import src.model as rm  # noqa: E402  (reference)

sys.modules["transformers"].AdamW = _HFAdamW   # after src.model: importing it swaps the lazy module object
import src.util as ru  # noqa: E402  (reference)


def _cw_forward(self, hidden_states, attention_mask=None, position_bias=None, *args, **kwargs):
    return self.module(hidden_states, attention_mask, position_bias, *args, **kwargs)


_ew_orig = rm.EncoderWrapper.forward


def _ew_forward(self, input_ids=None, attention_mask=None, **kwargs):
    out = _ew_orig(self, input_ids=input_ids, attention_mask=attention_mask, **kwargs)
    return BaseModelOutput(last_hidden_state=out[0])


rm.CheckpointWrapper.forward = _cw_forward
rm.EncoderWrapper.forward = _ew_forward
rm.EncoderWrapper.main_input_name = "input_ids"


def build_reference(dims: O.T5Dims, weights: dict[str, torch.Tensor], dropout: float = 0.0):
    cfg = transformers.T5Config(
        vocab_size=dims.vocab_size, d_model=dims.d_model, d_kv=dims.d_kv, d_ff=dims.d_ff,
        num_layers=dims.num_layers, num_decoder_layers=dims.num_decoder_layers, num_heads=dims.num_heads,
        relative_attention_num_buckets=dims.num_buckets, relative_attention_max_distance=dims.max_distance,
        dropout_rate=dropout, layer_norm_epsilon=dims.eps, feed_forward_proj="relu",
        decoder_start_token_id=0, pad_token_id=0, eos_token_id=1)
    cfg._attn_implementation = "eager"
    model = rm.FiDT5(cfg)
    sd = {k: v.clone() for k, v in weights.items()}
    sd["encoder.embed_tokens.weight"] = sd["shared.weight"]
    sd["decoder.embed_tokens.weight"] = sd["shared.weight"]
    sd["lm_head.weight"] = sd["shared.weight"]
    model.load_t5(sd)                                   # src/model.py:79-82
    assert model.lm_head.weight.data_ptr() == model.shared.weight.data_ptr()
    return model


def plain_name(wrapped: str) -> str | None:
    """Wrapped FiD key (encoder.encoder.block.i.module.layer…) → plain T5 key."""
    if wrapped in ("lm_head.weight", "encoder.encoder.embed_tokens.weight", "decoder.embed_tokens.weight"):
        return None
    if wrapped.startswith("encoder.encoder."):
        wrapped = "encoder." + wrapped[len("encoder.encoder."):]
    return wrapped.replace(".module.layer.", ".layer.")


def ref_grads(model) -> dict[str, np.ndarray]:
    out = {}
    for k, p in model.named_parameters():
        pk = plain_name(k)
        if pk is not None and p.grad is not None:
            out[pk] = p.grad.detach().numpy().copy()
    return out


def ref_weights(model) -> dict[str, np.ndarray]:
    out = {}
    for k, p in model.named_parameters():
        pk = plain_name(k)
        if pk is not None:
            out[pk] = p.detach().numpy().copy()
    return out


def make_case(name: str, dims: O.T5Dims, B, N, L, T, seed, full_pad=None, fact_case=False, pretrain=0):
    torch.manual_seed(seed)
    weights = O.init_weights(dims, seed=seed)
    ids, mask, labels = O.synthetic_batch(B, N, L, T, dims.vocab_size, seed=seed + 100)
    if full_pad is not None:                             # a fully padded passage
        b, n = full_pad
        mask[b, n] = False
        ids[b, n] = 0
    if fact_case:                                        # stream-2 layout: passage 1 = "fact : s1 . s2 . …"
        g = torch.Generator().manual_seed(seed + 7)
        for b in range(B):
            row = torch.randint(11, dims.vocab_size, (L,), generator=g)
            row[0], row[1] = 7, 10
            seps = sorted(torch.randperm(L - 4, generator=g)[: 3 + b].add(3).tolist())
            for s in seps:
                row[s] = 5
            valid = int(mask[b, 1].sum())
            if b == 0:
                valid = L                                # unterminated final span, no padding
                row[L - 1] = 9
            ids[b, 1] = row
            mask[b, 1] = torch.arange(L) < valid
            ids[b, 1] = ids[b, 1].masked_fill(~mask[b, 1], 0)

    if pretrain:
        # Overfit the (data-only) weights to ragged targets with the oracle's trainer so that greedy decode
        # reproduces them: rows then hit EOS at different steps, which pins the EOS/pad bookkeeping.  The
        # expected outputs below still come from the reference run on these weights.
        for b in range(B):
            tl = 2 + (b * 3) % (T - 1)
            labels[b] = torch.randint(2, dims.vocab_size, (T,), generator=torch.Generator().manual_seed(seed + b))
            labels[b, tl - 1] = 1
            labels[b, tl:] = -100
        st = {}
        for k in range(pretrain):
            O.train_step(weights, dims, st, ids, mask, labels, k, 3e-3, 0.0, 1.0, 5, pretrain + 50, training=False)

    out: dict[str, np.ndarray] = {}
    out.update({"w/" + k: v.numpy() for k, v in weights.items()})
    out.update(input_ids=ids.numpy(), attention_mask=mask.numpy(), labels=labels.numpy(),
               dims=np.array([dims.vocab_size, dims.d_model, dims.d_kv, dims.d_ff, dims.num_layers,
                              dims.num_decoder_layers, dims.num_heads, dims.num_buckets, dims.max_distance]))

    # ---- forward / backward through the reference -------------------------------------------
    model = build_reference(dims, weights)
    model.train()                                        # dropout_rate = 0 → deterministic
    acts = {}
    blk0 = model.encoder.encoder.block[0].module
    hooks = [
        blk0.layer[0].layer_norm.register_forward_hook(lambda m, i, o: acts.__setitem__("enc0_xn", o.detach())),
        blk0.layer[0].SelfAttention.q.register_forward_hook(lambda m, i, o: acts.__setitem__("enc0_q", o.detach())),
        blk0.layer[0].SelfAttention.k.register_forward_hook(lambda m, i, o: acts.__setitem__("enc0_k", o.detach())),
        blk0.layer[0].SelfAttention.v.register_forward_hook(lambda m, i, o: acts.__setitem__("enc0_v", o.detach())),
        blk0.layer[0].SelfAttention.o.register_forward_hook(
            lambda m, i, o: acts.__setitem__("enc0_ctx", i[0].detach())),
        blk0.layer[1].DenseReluDense.register_forward_hook(
            lambda m, i, o: acts.__setitem__("enc0_ffn_out", o.detach())),
        blk0.register_forward_hook(lambda m, i, o: acts.__setitem__("enc0_out", o[0].detach())),
    ]
    res = model(input_ids=ids, attention_mask=mask, labels=labels)
    for h in hooks:
        h.remove()
    loss = res[0]
    loss.backward()
    out.update(loss=np.array(loss.item(), dtype=np.float64), logits=res.logits.detach().numpy(),
               enc_out=res.encoder_last_hidden_state.detach().numpy())
    out.update({"act/" + k: v.numpy() for k, v in acts.items()})
    grads = ref_grads(model)
    out.update({"g/" + k: v for k, v in grads.items()})
    gn = float(np.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads.values())))
    out["grad_norm"] = np.array(gn)
    bias = blk0.layer[0].SelfAttention.compute_bias(L, L).detach()
    out["act/enc0_bias"] = bias.numpy()
    model.zero_grad()

    # ---- 3 optimizer steps through the reference's own set_optim / scheduler ---------------
    opt = types.SimpleNamespace(optim="adamw", lr=1e-2, weight_decay=1e-2, scheduler="linear", scheduler_steps=None,
                                total_steps=10, warmup_steps=2, fixed_lr=False)
    optimizer, scheduler = ru.set_optim(opt, model)      # src/util.py:230-245
    losses, norms = [], []
    for k in range(3):
        bi, bm, bl = O.synthetic_batch(B, N, L, T, dims.vocab_size, seed=seed + 200 + k)
        tl = model(input_ids=bi, attention_mask=bm, labels=bl)[0]
        tl.backward()
        norms.append(float(torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)))   # train_reader.py:76
        optimizer.step()
        scheduler.step()
        model.zero_grad()
        losses.append(tl.item())
        if k in (0, 2):
            out.update({f"w_step{k + 1}/" + n: v for n, v in ref_weights(model).items()})
    out["train_losses"] = np.array(losses)
    out["train_gnorms"] = np.array(norms)
    out["train_hparams"] = np.array([opt.lr, opt.weight_decay, 1.0, opt.warmup_steps, opt.total_steps])

    # ---- greedy generate + step-0 cross-attention scores -----------------------------------
    model = build_reference(dims, weights)
    model.eval()
    qk = {}
    hooks = []
    for li, blk in enumerate(model.decoder.block):
        att = blk.layer[1].EncDecAttention
        def _keep(kind, li):
            def hook(m, i, o):                           # first call = decode step 0; must return None
                qk.setdefault((kind, li), o.detach())
            return hook
        hooks.append(att.q.register_forward_hook(_keep("q", li)))
        hooks.append(att.k.register_forward_hook(_keep("k", li)))
    with torch.no_grad():
        for ml in (4, 12):
            toks = model.generate(input_ids=ids, attention_mask=mask, max_length=ml)
            out[f"gen_{ml}"] = toks.numpy()
    for h in hooks:
        h.remove()
    H, dk = dims.num_heads, dims.d_kv
    flat_mask = mask.view(B, -1)
    per_layer = []
    for li in range(dims.num_decoder_layers):
        q = qk[("q", li)].view(B, -1, H, dk).transpose(1, 2)
        k = qk[("k", li)].view(B, -1, H, dk).transpose(1, 2)
        s = torch.einsum("bnqd,bnkd->bnqk", q, k)        # the quantity src/model.py:316-329 stores (step 0)
        s = s + torch.zeros(B, 1, 1, N * L).masked_fill(~flat_mask[:, None, None, :], torch.finfo(torch.float32).min)
        per_layer.append(s)
        blk = model.decoder.block[li]
        blk.layer[1].EncDecAttention.score_storage = s
    out["cross_scores"] = torch.cat(per_layer, dim=2).numpy()       # [B,H,n_layers,N*L]
    if N == 2:
        for style in ("mean", "max", "21mean"):
            for half in ("no", "yes"):
                o2 = types.SimpleNamespace(stream=2, n_context=5, use_last_half_layer_attention=half,
                                           attention_score_style=style)
                fs = model.get_crossattention_scores(o2, ids, None, mask)         # src/model.py:143-204
                out[f"fact_scores_{style}_{half}"] = fs.numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", name + ".npz"), **out)
    print(name, "loss", loss.item(), "gnorm", gn, "gen", out["gen_12"].tolist())


def make_tables():
    from transformers.models.t5.modeling_t5 import T5Attention
    rel = torch.arange(-260, 261)
    enc = T5Attention._relative_position_bucket(rel, bidirectional=True, num_buckets=32, max_distance=128)
    dec = T5Attention._relative_position_bucket(rel, bidirectional=False, num_buckets=32, max_distance=128)
    sched = ru.WarmupLinearScheduler(torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0),
                                     warmup_steps=6, scheduler_steps=100, min_ratio=0.0, fixed_lr=False)
    lam = np.array([sched.lr_lambda(s) for s in range(0, 111)])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "tables.npz"), rel=rel.numpy(), enc_bucket=enc.numpy(),
                        dec_bucket=dec.numpy(), lr_lambda_w6_t100=lam)
    print("tables ok")


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


def make_collate():
    """Golden outputs of the reference's own Dataset + Collator (src/data.py) with the stub tokenizer."""
    import src.data as rd
    from tests.stub_tokenizer import StubTokenizer
    out = {}
    for stream in (1, 2):
        for use_fact in ("yes", "no"):
            for ans_len in (-1, 3):
                opt = types.SimpleNamespace(n_context=2, fact_use_way="concate", use_fact=use_fact)
                ds = rd.Dataset(EXAMPLES, opt)
                col = rd.Collator(12, StubTokenizer(), answer_maxlength=ans_len, stream=stream)
                index, tid, tmask, pid, pmask = col([ds[i] for i in range(len(ds))])
                key = f"s{stream}_{use_fact}_{ans_len}"
                out.update({key + "/index": index.numpy(), key + "/target_ids": tid.numpy(),
                            key + "/target_mask": tmask.numpy(), key + "/passage_ids": pid.numpy(),
                            key + "/passage_masks": pmask.numpy()})
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "collate.npz"), **out)
    print("collate ok", sorted({k.split("/")[0] for k in out}))


EVAL_STRINGS = [
    # (prediction, gold dict {answer: soft score})
    ("The Eiffel Tower", {"eiffel tower": 1.0, "tower": 0.6}),
    ("a red, fire-truck!", {"red fire truck": 0.3, "firetruck": 1.0, "red firetruck": 0.6}),
    ("an apple", {"apple pie": 1.0, "banana": 0.3}),
    ("ski", {"skiing": 1.0, "snowboarding": 0.6}),
    ("", {"nothing": 1.0}),
    ("New   York\tCity", {"new york city": 1.0, "nyc": 0.6}),
    ("theatre", {"the theatre": 0.6, "a theater": 1.0}),
    ("it's 3.5%", {"its 35": 1.0}),
    ("Café au lait", {"cafe au lait": 1.0, "café au lait": 0.3}),
    ("frisbee", {"frisbee": 1.0, "Frisbee.": 0.6, "disc": 0.3}),
    ("ÅNGSTRÖM unit", {"ångström unit": 1.0}),
    ("the", {"a": 1.0}),
]


from tests.util_golden import FixStem as _FixStem, FixTok as _FixTok  # noqa: E402


STEM_STRINGS = [
    ("what is the dog doing", {"dogs": 1.0, "running": 0.6}),
    ("he is skiing on the snow", {"ski": 1.0, "snowboard": 0.6}),
    ("the man rides a wave", {"surfing": 1.0, "waves": 0.3}),
    ("yes", {"yes": 1.0, "no": 0.3}),
    ("which and this", {"which": 1.0}),
    ("a frisbee and the dogs", {"frisbees": 0.6, "dog": 1.0}),
    ("bananas with apples", {"apple": 0.6, "banana": 0.6, "pear": 1.0}),
    ("washing machine", {"wash": 0.3, "machines": 1.0}),
    ("north america", {"america": 1.0}),
    ("", {"nothing": 1.0}),
]


def make_evaluation():
    """Golden values of the reference's answer metrics (src/evaluation.py: normalize_answer, ems, includ_ems, stem_ems incl.
    the stop-word mode) on fixed strings; the predictions / golds are data of this repository.  Also writes
    lako_amd/stop_words.json: the VALUES of the reference's module-level `stop_words` table (src/evaluation.py:21-28), which
    `normalize_answer(dele_sw=True)` deletes as substrings — a data table of the metric, de-duplicated and sorted."""
    import json
    import src.evaluation as rev
    rows = []
    for pred, golds in EVAL_STRINGS:
        rows.append({"prediction": pred, "golds": golds,
                     "normalized": rev.normalize_answer(pred),
                     "golds_normalized": {k: rev.normalize_answer(k) for k in golds},
                     "ems": float(rev.ems(pred, golds)), "includ_ems": float(rev.includ_ems(pred, golds))})
    for pred, golds in STEM_STRINGS + EVAL_STRINGS:
        rows.append({"prediction": pred, "golds": golds, "stem": True,
                     "normalized_sw": rev.normalize_answer(pred, True),
                     "stem_ems": float(rev.stem_ems(pred, golds, _FixTok(), _FixStem())),
                     "stem_ems_sw": float(rev.stem_ems(pred, golds, _FixTok(), _FixStem(), dele_sw=True))})
    with open(os.path.join(ROOT, "tests", "golden", "evaluation.json"), "w") as f:
        json.dump(rows, f, indent=1, ensure_ascii=False)
    with open(os.path.join(ROOT, "lako_amd", "stop_words.json"), "w") as f:
        json.dump(sorted(set(rev.stop_words)), f, ensure_ascii=False)
    print("wrote evaluation.json", len(rows), "stop words", len(set(rev.stop_words)))


RETRIEVER_CASES = {   # name → RetrieverConfig overrides; one tiny BERT shared by all
    "proj_mean": dict(projection=True),
    "proj_mask": dict(projection=True, apply_question_mask=True, apply_passage_mask=True),
    "proj_cls": dict(projection=True, extract_cls=True, apply_passage_mask=True),
    "raw_mean": dict(projection=False),
    "asym_mask": dict(projection=False, asymmetric_retri="yes", apply_question_mask=True, apply_passage_mask=True),
}
RETRIEVER_TINY = dict(vocab_size=96, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                      max_position_embeddings=40, indexing_dimension=64)


def make_retriever():
    """Outputs of the reference's own Retriever (src/model.py:375-483 on transformers' BertModel, eval mode) for the
    weights oracle/retriever_oracle.init_weights generates → tests/golden/retriever.npz (inputs are regenerated from seeds)."""
    import src.model as rm
    from lako_amd.retriever import RetrieverConfig
    from oracle import retriever_oracle as RO
    out = {}
    for ci, (name, over) in enumerate(RETRIEVER_CASES.items()):
        kw = dict(RETRIEVER_TINY, **over)
        ref_cfg = rm.RetrieverConfig(**kw)
        cfg = RetrieverConfig.from_hf(ref_cfg)
        w = RO.init_weights(cfg, seed=40 + ci)
        ref = rm.Retriever(ref_cfg)
        sd = ref.state_dict()
        assert set(w) <= set(sd), sorted(set(w) - set(sd))
        missing, unexpected = ref.load_state_dict(w, strict=False)
        assert not unexpected and all("position_ids" in k or "token_type_ids" in k for k in missing), (missing, unexpected)
        ref.eval()
        qi, qm, pi, pm, gold = RO.synthetic_batch(cfg, 3, 4, 9, 14, seed=70 + ci)
        with torch.no_grad():
            q, p, score, loss = ref(qi, qm, pi, pm, gold_score=gold)
        for k, v in (("q", q), ("p", p), ("score", score), ("loss", loss)):
            out[f"{name}.{k}"] = v.detach().double().numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "retriever.npz"), **out)
    print("retriever.npz", {k: v.shape for k, v in out.items() if k.startswith("proj_mean")})


RANK_SCORES = [[0.9, 0.1, 0.5, 0.3, 0.2, 0.05, 0.7], [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7], [3.0, 2.0, 1.0, 0.5, 0.4, 0.3, 0.2],
               [0.0, 5.0, -1.0, 4.0, 4.5, 2.0, 1.0], [1.0, 0.5]]


def make_rank_metrics():
    """Golden values of the retriever evaluation's ranking metrics (src/evaluation.py:200-232: eval_batch → inversions, share of the
    predicted top-k in the gold top-k, predicted passages needed to cover the gold top-k) on fixed score rows → tests/golden/rank_metrics.json"""
    import json
    import src.evaluation as rev
    rows = []
    for sc in RANK_SCORES:
        ks = [k for k in (1, 2, 5) if k <= len(sc)]
        inv, avg, idx = [], {k: [] for k in ks}, {k: [] for k in ks}
        rev.eval_batch(torch.tensor([sc]), inv, avg, idx)
        rows.append({"scores": sc, "inversions": [int(v) for v in inv], "avg_topk": {str(k): [float(x) for x in v] for k, v in avg.items()},
                     "idx_topk": {str(k): [int(x) for x in v] for k, v in idx.items()}})
    with open(os.path.join(ROOT, "tests", "golden", "rank_metrics.json"), "w") as f:
        json.dump(rows, f, indent=1)
    print("wrote rank_metrics.json", len(rows))


def make_retriever_train():
    """Gradients of the reference's own Retriever (src/model.py:413-483: KLDivLoss of log_softmax(q·p/√d) against the gold
    scores, train_retriever.py:57-66) for every parameter, plus the weights after TWO steps of the reference's optimizer
    (src/util.set_optim → the HF<=4 AdamW stub of this harness + WarmupLinearScheduler, clip_grad_norm_ 1.0 as in
    train_retriever.py:68-71) → tests/golden/retriever_train.npz.  eval() mode (dropout off), same weights / batches as
    make_retriever."""
    import types
    import src.model as rm
    import src.util as ru
    from lako_amd.retriever import RetrieverConfig
    from oracle import retriever_oracle as RO
    out = {}
    for ci, (name, over) in enumerate(RETRIEVER_CASES.items()):
        if name == "proj_mean":              # (covered by proj_mask / raw_mean)
            continue
        kw = dict(RETRIEVER_TINY, **over)
        ref_cfg = rm.RetrieverConfig(**kw)
        cfg = RetrieverConfig.from_hf(ref_cfg)
        w = RO.init_weights(cfg, seed=40 + ci)
        ref = rm.Retriever(ref_cfg)
        ref.load_state_dict(w, strict=False)
        ref.eval()
        qi, qm, pi, pm, gold = RO.synthetic_batch(cfg, 3, 4, 9, 14, seed=70 + ci)
        _, _, _, loss = ref(qi, qm, pi, pm, gold_score=gold)
        loss.backward()
        out[f"{name}.loss"] = loss.detach().double().numpy()
        for k, prm in ref.named_parameters():
            if k in w:
                g = prm.grad if prm.grad is not None else torch.zeros_like(prm)
                out[f"{name}.g/{k}"] = g.detach().numpy().astype(np.float32)
        ref.zero_grad()
        if name != "asym_mask":              # (fixture size: the optimizer steps of one configuration, gradients of all)
            continue
        opt = types.SimpleNamespace(optim="adamw", lr=1e-3, weight_decay=0.01, scheduler="linear", scheduler_steps=None, total_steps=10,
                                    warmup_steps=1, fixed_lr=False)
        optimizer, scheduler = ru.set_optim(opt, ref)
        for step in range(2):
            _, _, _, loss = ref(qi, qm, pi, pm, gold_score=gold)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
            optimizer.step()
            scheduler.step()
            ref.zero_grad()
            out[f"{name}.loss_step{step}"] = loss.detach().double().numpy()
        for k, prm in ref.named_parameters():
            if k in w:
                out[f"{name}.w2/{k}"] = prm.detach().numpy().astype(np.float32)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "retriever_train.npz"), **out)
    print("retriever_train.npz", len(out), "arrays")


def make_rerank():
    """Golden outputs of the reference's OWN re-rank, fact_retrieval_small_range.py:64-89 (`resort_facts`: every example's candidate facts
    re-ordered by the inner product of their embeddings with the example's question embedding, `sorted(zip(score, id), reverse=True)`),
    imported from /root/reference with `faiss` stubbed in sys.modules (the module imports src.index, which imports faiss at its top; the
    function itself is plain torch — nothing of faiss runs) → tests/golden/rerank.json: seeded embeddings, the examples' candidate ids, the
    order and scores the reference produced.  Includes exact score ties (duplicated embedding rows: descending id wins, the tuple sort) and
    an example with one candidate."""
    import copy
    import json
    sys.modules.setdefault("faiss", types.ModuleType("faiss"))
    import fact_retrieval_small_range as frs
    g = np.random.default_rng(11)
    n, d, nq = 60, 16, 5
    emb = (g.integers(-8, 9, size=(n, d)) / 8.0).astype(np.float32)      # multiples of 1/8: the products are exact in fp32 on any path
    emb[7] = emb[3]
    emb[41] = emb[3]                                                     # ties inside one candidate list
    q = (g.integers(-8, 9, size=(nq, d)) / 8.0).astype(np.float32)
    cand = [sorted(set([3, 7, 41] + g.choice(n, size=12, replace=False).tolist())), g.choice(n, size=20, replace=False).tolist(),
            [5], g.permutation(n).tolist(), [41, 3, 7, 0]]
    dic = {str(i): f"fact number {i}" for i in range(n)}
    examples = [{"question": f"q{k}", "fact": [{"id": str(i), "sentence": dic[str(i)], "score": 0} for i in cand[k]]} for k in range(nq)]
    ref = copy.deepcopy(examples)
    frs.resort_facts(ref, dic, torch.from_numpy(q), torch.from_numpy(emb))
    out = {"embeddings": emb.tolist(), "questions": q.tolist(), "candidates": cand,
           "expected": [{"ids": [int(f["id"]) for f in ex["fact"]], "scores": [float(f["score"]) for f in ex["fact"]],
                         "sentences": [f["sentence"] for f in ex["fact"]]} for ex in ref]}
    with open(os.path.join(ROOT, "tests", "golden", "rerank.json"), "w") as f:
        json.dump(out, f)
    print("wrote rerank.json", [len(e["ids"]) for e in out["expected"]])


def make_legacy_cross():
    """The reference's OWN cross-attention forward (src/model.py:286-349) called as it was written to be called — the
    transformers-3.0.2 argument list, `position_bias=None` on a layer with `has_relative_attention_bias`, so that line 323-325
    computes `self.compute_bias(qlen, klen)` and adds it — on an installed-transformers `T5Attention(is_decoder=True,
    has_relative_attention_bias=True)` (its q / k / v / o / relative_attention_bias / compute_bias are what the function touches;
    `d_kv` is set beside transformers' newer name for it).  Pins what `legacy_cross_bias` means: WHICH bias (the decoder's one-sided
    buckets of key position − query position over (qlen, klen)), WHERE (before the softmax, inside the stored score), and that a
    decode step of query length 1 gets the row of position 0.  → tests/golden/legacy_cross.npz"""
    from transformers.models.t5.modeling_t5 import T5Attention
    g = torch.Generator().manual_seed(11)
    B, T, S, d, H, dk = 2, 7, 45, 48, 3, 16
    cfg = transformers.T5Config(vocab_size=32, d_model=d, d_kv=dk, d_ff=64, num_layers=1, num_decoder_layers=1, num_heads=H,
                                relative_attention_num_buckets=32, relative_attention_max_distance=128, dropout_rate=0.0,
                                is_decoder=True)
    try:
        att = T5Attention(cfg, has_relative_attention_bias=True, layer_idx=0)
    except TypeError:
        att = T5Attention(cfg, has_relative_attention_bias=True)
    att.eval()
    att.is_decoder = True
    att.d_kv = dk
    w = {n: torch.randn(sh, generator=g) * sc for n, sh, sc in (("q", (H * dk, d), 0.12), ("k", (H * dk, d), 0.3), ("v", (H * dk, d), 0.3),
                                                               ("o", (d, H * dk), 0.2), ("table", (32, H), 0.7))}
    with torch.no_grad():
        for n in "qkvo":
            getattr(att, n).weight.copy_(w[n])
        att.relative_attention_bias.weight.copy_(w["table"])
    x = torch.randn(B, T, d, generator=g)
    kv = torch.randn(B, S, d, generator=g)
    keep = torch.ones(B, S, dtype=torch.bool)
    keep[0, 30:] = False
    keep[1, 3:9] = False         # a hole INSIDE the first positions: the bias is by position, not by count of valid keys
    mask = torch.zeros(B, 1, 1, S).masked_fill(~keep[:, None, None, :], -1e9)
    out = {k: v.numpy() for k, v in w.items()}
    out.update(x=x.numpy(), kv=kv.numpy(), keep=keep.numpy())
    with torch.no_grad():
        for tag, xin in (("full", x), ("step", x[:, 3:4])):          # the training call (qlen = T) and one cached decode step (qlen = 1)
            att.score_storage = None
            res = rm.cross_attention_forward(att, xin, mask=mask.clone(), kv=kv, position_bias=None)
            out[f"out_{tag}"] = res[0].numpy()
            out[f"scores_{tag}"] = att.score_storage.numpy()
            out[f"bias_{tag}"] = res[-1].numpy()                      # (position_bias is returned when the layer owns the table, :346-347)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "legacy_cross.npz"), **out)
    print("legacy_cross ok", out["bias_full"].shape, out["bias_step"].shape)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "legacy_cross":
        make_legacy_cross()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "rerank":
        make_rerank()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "retriever_train":
        make_retriever_train()
        make_rank_metrics()
        sys.exit(0)
    tiny = O.T5Dims.named("tiny")
    make_tables()
    make_collate()
    make_retriever()
    make_retriever_train()
    make_rank_metrics()
    make_rerank()
    make_evaluation()
    make_legacy_cross()
    make_case("tiny_a", tiny, B=3, N=3, L=12, T=5, seed=1, full_pad=(1, 2))
    make_case("tiny_fact", tiny, B=3, N=2, L=24, T=4, seed=2, fact_case=True)
    make_case("tiny_eos", tiny, B=4, N=3, L=12, T=6, seed=5, pretrain=150)
    # a slightly wider case: odd L, more heads/layers, exercises the log-spaced buckets (L > 16)
    mid = O.T5Dims(vocab_size=96, d_model=64, d_kv=32, d_ff=128, num_layers=3, num_decoder_layers=2, num_heads=4)
    make_case("mid_a", mid, B=2, N=4, L=37, T=7, seed=3)
