"""SURVEY.md §8 f4, the bi-encoder part: `lako_amd.Retriever` forward (src/model.py:375-483 over HF BertModel) against
golden outputs of the reference's own Retriever (tests/golden/retriever.npz, oracle/make_fixtures.py::make_retriever) and
against the restatement in oracle/retriever_oracle.py.  Host logic on the test double (CPU); kernels and the class on the GPU."""
import numpy as np
import pytest
import torch

from lako_amd.retriever import Retriever, RetrieverConfig
from oracle import retriever_oracle as RO
from tests.ref_ops import RefOps

CASES = {
    "proj_mean": dict(projection=True),
    "proj_mask": dict(projection=True, apply_question_mask=True, apply_passage_mask=True),
    "proj_cls": dict(projection=True, extract_cls=True, apply_passage_mask=True),
    "raw_mean": dict(projection=False),
    "asym_mask": dict(projection=False, asymmetric_retri="yes", apply_question_mask=True, apply_passage_mask=True),
}
TINY = dict(vocab_size=96, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
            max_position_embeddings=40, indexing_dimension=64)
GOLD = np.load(__file__.rsplit("/", 1)[0] + "/golden/retriever.npz")


def case(name):
    ci = list(CASES).index(name)
    cfg = RetrieverConfig(**dict(TINY, **CASES[name]))
    return cfg, RO.init_weights(cfg, seed=40 + ci), RO.synthetic_batch(cfg, 3, 4, 9, 14, seed=70 + ci)


def check_against_gold(name, q, p, score, loss, tol):
    for k, v in (("q", q), ("p", p), ("score", score), ("loss", loss)):
        np.testing.assert_allclose(v.detach().double().cpu().numpy(), GOLD[f"{name}.{k}"], rtol=tol, atol=tol, err_msg=f"{name}.{k}")


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_matches_reference_outputs(name):
    cfg, w, (qi, qm, pi, pm, gold) = case(name)
    check_against_gold(name, *RO.forward(cfg, w, qi, qm, pi, pm, gold), tol=2e-5)


@pytest.mark.parametrize("name", list(CASES))
def test_host_logic_matches_reference_outputs(name):
    cfg, w, (qi, qm, pi, pm, gold) = case(name)
    model = Retriever(cfg, dtype=torch.float32, _ops=RefOps())
    model.load_state_dict(w)
    check_against_gold(name, *model(qi, qm, pi, pm, gold_score=gold), tol=5e-5)
    q, p, score, loss = model(qi, qm, pi, pm)
    assert loss is None and score.shape == (3, 4) and q.shape == (3, 64 if cfg.projection or cfg.asymmetric_retri == "yes" else 128)


def test_state_dict_names_and_checkpoint_roundtrip(tmp_path):
    cfg, w, (qi, qm, pi, pm, gold) = case("proj_mask")
    model = Retriever(cfg, dtype=torch.float32, _ops=RefOps())
    assert set(model.state_dict()) == set(w)
    model.load_state_dict(dict(w, **{"model.embeddings.position_ids": torch.arange(40)[None]}))     # old HF buffer: ignored
    with pytest.raises(KeyError):
        model.load_state_dict({k: v for k, v in w.items() if k != "proj.weight"})
    with pytest.raises(ValueError):
        model.load_state_dict(dict(w, **{"proj.weight": torch.zeros(3, 3)}))
    ref = model(qi, qm, pi, pm)[2]
    model.save_pretrained(tmp_path / "ckpt")
    again = Retriever.from_pretrained(str(tmp_path / "ckpt"), dtype=torch.float32, _ops=RefOps())
    assert again.config == cfg
    assert torch.equal(again(qi, qm, pi, pm)[2], ref)
    with torch.no_grad():                                   # an in-place weight edit must invalidate the packed layout
        again.proj.bias.add_(1.0)
    assert not torch.equal(again(qi, qm, pi, pm)[2], ref)


def test_config_from_reference_style_dict_and_guards():
    cfg = RetrieverConfig.from_hf(dict(TINY, projection=False, asymmetric_retri="yes", model_type="bert", unused_field=1))
    assert cfg.asymmetric_retri == "yes" and not cfg.projection and cfg.layer_norm_eps == 1e-12
    from lako_amd._lib import LakoError
    with pytest.raises(LakoError):
        Retriever(RetrieverConfig(**dict(TINY, hidden_act="relu")))
    with pytest.raises(LakoError):
        Retriever(RetrieverConfig(**dict(TINY, num_attention_heads=8)))       # head size 16


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only check")
def test_product_path_needs_the_gpu():
    cfg, w, (qi, qm, pi, pm, _) = case("raw_mean")
    model = Retriever(cfg)
    model.load_state_dict(w)
    with pytest.raises(Exception):
        model(qi, qm, pi, pm)


# ---------------------------------------------------------------------------------------------------------------
gpu = pytest.mark.gpu


@gpu
@pytest.mark.parametrize("name", list(CASES))
def test_gpu_fp32_matches_reference_outputs(name):
    cfg, w, (qi, qm, pi, pm, gold) = case(name)
    model = Retriever(cfg, dtype=torch.float32)
    model.load_state_dict(w)
    model = model.cuda()
    check_against_gold(name, *model(qi.cuda(), qm.cuda(), pi.cuda(), pm.cuda(), gold_score=gold.cuda()), tol=2e-4)


@gpu
@pytest.mark.parametrize("name", ["proj_mask", "raw_mean"])
def test_gpu_bf16_close_to_reference_outputs(name):
    cfg, w, (qi, qm, pi, pm, gold) = case(name)
    model = Retriever(cfg, dtype=torch.bfloat16)
    model.load_state_dict(w)
    model = model.cuda()
    q, p, score, loss = model(qi.cuda(), qm.cuda(), pi.cuda(), pm.cuda(), gold_score=gold.cuda())
    # bf16 activations through 2 post-LN blocks: embeddings are O(1) after the LayerNorm → absolute tolerance
    np.testing.assert_allclose(q.cpu().double().numpy(), GOLD[f"{name}.q"], atol=6e-2, rtol=0)
    np.testing.assert_allclose(score.cpu().double().numpy(), GOLD[f"{name}.score"], atol=0.25, rtol=0)
    gs = GOLD[f"{name}.score"]                         # the top-scored fact is the reference's, up to near ties
    assert np.all(gs.max(1) - gs[np.arange(len(gs)), score.cpu().numpy().argmax(1)] < 0.05)


@gpu
def test_gpu_bert_base_shape_against_oracle():
    """BERT-base dimensions (768 / 12 heads / 3072, 3 layers to keep the CPU oracle quick), L = 130 (passage_maxlength)."""
    cfg = RetrieverConfig(num_hidden_layers=3, vocab_size=3000, apply_passage_mask=True, apply_question_mask=True)
    w = RO.init_weights(cfg, seed=9)
    qi, qm, pi, pm, gold = RO.synthetic_batch(cfg, 2, 5, 40, 130, seed=10)
    model = Retriever(cfg, dtype=torch.float32)
    model.load_state_dict(w)
    model = model.cuda()
    q, p, score, loss = model(qi.cuda(), qm.cuda(), pi.cuda(), pm.cuda(), gold_score=gold.cuda())
    oq, op, os_, ol = RO.forward(cfg, w, qi, qm, pi, pm, gold)
    np.testing.assert_allclose(q.cpu().double().numpy(), oq.numpy(), atol=3e-4, rtol=3e-4)
    np.testing.assert_allclose(p.cpu().double().numpy(), op.numpy(), atol=3e-4, rtol=3e-4)
    np.testing.assert_allclose(score.cpu().double().numpy(), os_.numpy(), atol=1e-3, rtol=1e-3)
    np.testing.assert_allclose(float(loss), float(ol), atol=1e-4, rtol=1e-3)


@gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gpu_rowwise_kernels_match_test_double(dtype):
    from lako_amd.ops import HipOps
    hip, ref = HipOps(), RefOps()
    g = torch.Generator().manual_seed(3)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    for rows, d in ((7, 64), (130, 768), (33, 1024), (5, 8)):
        x, r = torch.randn(rows, d, generator=g).to(dtype), torch.randn(rows, d, generator=g).to(dtype)
        lb, ga, be = torch.randn(d, generator=g), 1 + 0.1 * torch.randn(d, generator=g), torch.randn(d, generator=g)
        for use_b, use_r in ((True, True), (False, False), (True, False)):
            want = torch.empty(rows, d, dtype=dtype)
            ref.layernorm_fwd(x, ga, be, want, lin_bias=lb if use_b else None, resid=r if use_r else None, eps=1e-5)
            got = torch.empty(rows, d, dtype=dtype, device="cuda")
            hip.layernorm_fwd(x.cuda(), ga.cuda(), be.cuda(), got, lin_bias=lb.cuda() if use_b else None,
                              resid=r.cuda() if use_r else None, eps=1e-5)
            torch.testing.assert_close(got.cpu().float(), want.float(), atol=tol, rtol=tol)
        for gelu in (False, True):
            want = torch.empty(rows, d, dtype=dtype)
            ref.bias_act(x, lb, want, gelu=gelu)
            got = x.cuda().clone()
            hip.bias_act(got, lb.cuda(), got, gelu=gelu)                # in place, as the retriever uses it
            torch.testing.assert_close(got.cpu().float(), want.float(), atol=tol, rtol=tol)
    for B, L, d in ((3, 9, 64), (4, 130, 256), (2, 17, 768), (2, 5, 1032)):
        x = torch.randn(B, L, d, generator=g).to(dtype)
        m = torch.rand(B, L, generator=g) < 0.6
        m[:, 0] = True
        for mask in (None, m):
            want = torch.empty(B, d)
            ref.seq_mean(x, mask, want)
            got = torch.empty(B, d, device="cuda")
            hip.seq_mean(x.cuda(), None if mask is None else mask.cuda(), got)
            torch.testing.assert_close(got.cpu(), want, atol=1e-5 if dtype == torch.float32 else 1e-4, rtol=1e-5)
    if dtype == torch.float32:
        for B, n, d in ((3, 4, 64), (5, 100, 256), (1, 1, 768)):
            q, p = torch.randn(B, d, generator=g), torch.randn(B, n, d, generator=g)
            want, got = torch.empty(B, n), torch.empty(B, n, device="cuda")
            ref.bi_score(q, p, want, 0.125)
            hip.bi_score(q.cuda(), p.cuda(), got, 0.125)
            torch.testing.assert_close(got.cpu(), want, atol=1e-4, rtol=1e-5)
            gold = torch.softmax(torch.randn(B, n, generator=g), -1)
            gold[0, 0] = 0.0
            wl, gl = torch.empty(1), torch.empty(1, device="cuda")
            ref.kldiv_fwd(want, gold, wl)
            hip.kldiv_fwd(got, gold.cuda(), gl)
            torch.testing.assert_close(gl.cpu(), wl, atol=1e-6, rtol=1e-4)
    V, L, d = 50, 12, 128
    word, pos, ty = torch.randn(V, d, generator=g), torch.randn(L + 3, d, generator=g), torch.randn(d, generator=g)
    ga, be = 1 + 0.1 * torch.randn(d, generator=g), torch.randn(d, generator=g)
    ids = torch.randint(0, V, (5 * L,), generator=g)
    want = torch.empty(5 * L, d, dtype=dtype)
    ref.bert_embed(ids, word, pos, ty, ga, be, want, L)
    got = torch.empty(5 * L, d, dtype=dtype, device="cuda")
    hip.bert_embed(ids.cuda(), word.cuda(), pos.cuda(), ty.cuda(), ga.cuda(), be.cuda(), got, L)
    torch.testing.assert_close(got.cpu().float(), want.float(), atol=tol, rtol=tol)


# ---- training (round 3): gradients and optimizer steps of the reference's own Retriever (tests/golden/retriever_train.npz,
# oracle/make_fixtures.py::make_retriever_train) ----------------------------------------------------------------------------------
TRAIN = np.load(__file__.rsplit("/", 1)[0] + "/golden/retriever_train.npz")
TRAIN_CASES = [n for n in CASES if f"{n}.loss" in TRAIN]


def ref_grads(name):
    return {k.split("/", 1)[1]: torch.from_numpy(TRAIN[k]) for k in TRAIN.files if k.startswith(name + ".g/")}


def check_grads(name, got, tol):
    ref = ref_grads(name)
    scale = max(float(v.abs().max()) for v in ref.values())
    assert set(ref) <= set(got)
    for k, r in ref.items():
        # (relative to the tensor's own scale, but not below 1e-3 of the largest gradient: the key biases have a ZERO true gradient —
        #  softmax ignores a shift of all scores of a query — and the reference's fp32 autograd leaves noise there)
        err = float((got[k].double().cpu() - r.double()).abs().max()) / max(1e-3 * scale, float(r.abs().max()))
        assert err < tol, (name, k, err)


@pytest.mark.parametrize("name", TRAIN_CASES)
def test_oracle_gradients_match_reference(name):
    """the restatement's autograd (float64) against the gradients of the reference's Retriever: pins the training oracle"""
    cfg, w, (qi, qm, pi, pm, gold) = case(name)
    leaves = {k: v.double().clone().requires_grad_(True) for k, v in w.items()}
    loss = RO.forward(cfg, leaves, qi, qm, pi, pm, gold)[3]
    loss.backward()
    assert abs(float(loss) - float(TRAIN[f"{name}.loss"])) < 1e-6
    check_grads(name, {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}, tol=2e-4)


@pytest.mark.parametrize("name", TRAIN_CASES)
def test_host_backward_matches_reference_gradients(name):
    """the hand-written backward schedule (lako_amd/retriever_train.py) on the fp32 test double against the reference's gradients"""
    cfg, w, (qi, qm, pi, pm, gold) = case(name)
    model = Retriever(cfg, dtype=torch.float32, _ops=RefOps())
    model.load_state_dict(w)
    model.eval()                                            # dropout off, like the fixture
    loss = model(qi, qm, pi, pm, gold_score=gold)[3]
    assert abs(float(loss) - float(TRAIN[f"{name}.loss"])) < 1e-5
    loss.backward()
    check_grads(name, {n: p.grad for n, p in model.named_parameters()}, tol=2e-3)
    with torch.no_grad():                                   # under no_grad the forward is the inference schedule: no graph
        assert model(qi, qm, pi, pm, gold_score=gold)[3].grad_fn is None


def train_two_steps(model, batch):
    import types
    from lako_amd import util as U
    qi, qm, pi, pm, gold = batch
    opt = types.SimpleNamespace(optim="adamw", lr=1e-3, weight_decay=0.01, scheduler="linear", scheduler_steps=None, total_steps=10,
                                warmup_steps=1, fixed_lr=False)
    optimizer, scheduler = U.set_optim(opt, model)
    losses = []
    for _ in range(2):                                      # train_retriever.py:57-71
        loss = model(qi, qm, pi, pm, gold_score=gold)[3]
        loss.backward()
        U.clip_grad_norm_(model, 1.0)
        optimizer.step()
        scheduler.step()
        model.zero_grad()
        losses.append(float(loss))
    return losses


def test_host_two_optimizer_steps_vs_reference():
    """two steps of train_retriever.py's loop (clip 1.0, AdamW without bias correction, linear warm-up) against the weights the
    reference's own loop leaves — incl. the unused pooler, which its optimizer never touches"""
    name = "asym_mask"
    cfg, w, batch = case(name)
    model = Retriever(cfg, dtype=torch.float32, _ops=RefOps())
    model.load_state_dict(w)
    model.eval()
    losses = train_two_steps(model, batch)
    for i, l in enumerate(losses):
        assert abs(l - float(TRAIN[f"{name}.loss_step{i}"])) < 2e-5, (i, l)
    for k in w:
        ref = torch.from_numpy(TRAIN[f"{name}.w2/{k}"])
        torch.testing.assert_close(model.state_dict()[k].cpu(), ref, atol=2e-5, rtol=1e-4, msg=lambda m, k=k: f"{k}: {m}")


def test_dropout_training_is_deterministic_and_differs_between_steps():
    cfg, w, (qi, qm, pi, pm, gold) = case("proj_mask")
    res = []
    for _ in range(2):
        model = Retriever(cfg, dtype=torch.float32, _ops=RefOps(), seed=3)
        model.load_state_dict(w)
        model.train()
        model.set_dropout(0.2)
        out = []
        for step in range(2):
            loss = model(qi, qm, pi, pm, gold_score=gold)[3]
            loss.backward()
            out.append((float(loss), model._get_engine().G.clone()))
            model.zero_grad()
        res.append(out)
    assert res[0][0][0] == res[1][0][0] and torch.equal(res[0][0][1], res[1][0][1])       # same seed, same step: same masks
    assert res[0][0][0] != res[0][1][0]                                                     # next step: new masks
    model.eval()
    assert abs(float(model(qi, qm, pi, pm, gold_score=gold)[3]) - float(TRAIN["proj_mask.loss"])) < 1e-5


@gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gpu_training_kernels_match_test_double(dtype):
    """csrc/bertbwd.hip against the torch-autograd doubles of tests/ref_ops.py: LayerNorm backward (with / without linear bias and
    residual, parameter-gradient accumulation), bias / GELU backward on column slices of a fused buffer with a bias-gradient scale,
    mean-pooling backward (all rows, masked, CLS), bi-encoder score and KLDiv backward, embedding backward (scatter + pad ids)."""
    from lako_amd.ops import HipOps
    hip, ref = HipOps(), RefOps()
    g = torch.Generator().manual_seed(13)
    tol = 3e-5 if dtype == torch.float32 else 3e-2
    c = lambda t: t.cuda()        # noqa: E731
    for rows, d in ((7, 64), (700, 768), (33, 1024), (5, 8)):
        x, r, dy = (torch.randn(rows, d, generator=g).to(dtype) for _ in range(3))
        lb, ga = torch.randn(d, generator=g), 1 + 0.1 * torch.randn(d, generator=g)
        for use_b, use_r in ((True, True), (False, False), (False, True)):
            acc0 = [torch.randn(d, generator=g) for _ in range(3)]
            want = torch.empty(rows, d, dtype=dtype)
            wacc = [a.clone() for a in acc0]
            ref.layernorm_bwd(dy, x, ga, want, wacc[0], wacc[1], lin_bias=lb if use_b else None, resid=r if use_r else None,
                              dbias=wacc[2] if use_b else None, eps=1e-5)
            got = torch.empty(rows, d, dtype=dtype, device="cuda")
            gacc = [c(a.clone()) for a in acc0]
            hip.layernorm_bwd(c(dy), c(x), c(ga), got, gacc[0], gacc[1], lin_bias=c(lb) if use_b else None, resid=c(r) if use_r else None,
                              dbias=gacc[2] if use_b else None, eps=1e-5)
            torch.testing.assert_close(got.cpu().float(), want.float(), atol=tol * 3, rtol=tol)
            for a, b in zip(gacc, wacc):
                torch.testing.assert_close(a.cpu(), b, atol=tol * max(1.0, float(b.abs().max())) * 3, rtol=tol)
    for rows, n, blocks in ((9, 64, 1), (300, 768, 3), (1000, 3072, 1)):
        buf, xb = torch.randn(rows, n * blocks, generator=g).to(dtype), torch.randn(rows, n * blocks, generator=g).to(dtype)
        bias = torch.randn(n * blocks, generator=g)
        for gelu in (False, True):
            for j in range(blocks):
                sl = slice(j * n, (j + 1) * n)
                db0 = torch.randn(n, generator=g)
                wdx, wdb = torch.empty(rows, n, dtype=dtype), db0.clone()
                ref.bias_act_bwd(buf[:, sl], xb[:, sl] if gelu else None, bias[sl], wdx, wdb, gelu=gelu, dbias_scale=0.5)
                gbuf, gdb = c(buf.clone()), c(db0.clone())
                hip.bias_act_bwd(gbuf[:, sl], c(xb)[:, sl] if gelu else None, c(bias)[sl], gbuf[:, sl] if gelu else None, gdb, gelu=gelu,
                                 dbias_scale=0.5)
                if gelu:
                    torch.testing.assert_close(gbuf[:, sl].cpu().float(), wdx.float(), atol=tol * 3, rtol=tol)
                    keep = torch.ones(n * blocks, dtype=torch.bool)
                    keep[sl] = False
                    assert torch.equal(gbuf.cpu()[:, keep], buf[:, keep])                # nothing outside the column block is written
                torch.testing.assert_close(gdb.cpu(), wdb, atol=tol * max(1.0, float(wdb.abs().max())) * 10, rtol=tol)
    for B, L, d in ((3, 9, 64), (4, 130, 256), (2, 5, 1032)):
        dout = torch.randn(B, d, generator=g)
        m = torch.rand(B, L, generator=g) < 0.6
        m[:, 0] = True
        cls = torch.zeros(B, L, dtype=torch.bool)
        cls[:, 0] = True
        for mask in (None, m, cls):
            want, got = torch.empty(B, L, d, dtype=dtype), torch.empty(B, L, d, dtype=dtype, device="cuda")
            ref.seq_mean_bwd(dout, mask, want)
            hip.seq_mean_bwd(c(dout), None if mask is None else c(mask.to(torch.uint8)), got)
            torch.testing.assert_close(got.cpu().float(), want.float(), atol=tol, rtol=tol)
    if dtype == torch.float32:
        for B, n, d in ((3, 4, 64), (5, 100, 256), (1, 1, 768)):
            q, p, ds = torch.randn(B, d, generator=g), torch.randn(B, n, d, generator=g), torch.randn(B, n, generator=g)
            wq, wp, gq, gp = torch.empty(B, d), torch.empty(B, n, d), torch.empty(B, d, device="cuda"), torch.empty(B, n, d, device="cuda")
            ref.bi_score_bwd(ds, q, p, wq, wp, 0.125)
            hip.bi_score_bwd(c(ds), c(q), c(p), gq, gp, 0.125)
            torch.testing.assert_close(gq.cpu(), wq, atol=1e-4, rtol=1e-5)
            torch.testing.assert_close(gp.cpu(), wp, atol=1e-5, rtol=1e-5)
            sc, gold = torch.randn(B, n, generator=g) * 2, torch.softmax(torch.randn(B, n, generator=g), -1)
            gold[0, 0] = 0.0
            up = torch.tensor([0.7])
            wd, gd = torch.empty(B, n), torch.empty(B, n, device="cuda")
            ref.kldiv_bwd(sc, gold, wd, up)
            hip.kldiv_bwd(c(sc), c(gold), gd, c(up))
            torch.testing.assert_close(gd.cpu(), wd, atol=1e-7, rtol=1e-4)
    V, L, d = 50, 12, 128
    word, pos, ty = torch.randn(V, d, generator=g), torch.randn(L + 3, d, generator=g), torch.randn(d, generator=g)
    ga = 1 + 0.1 * torch.randn(d, generator=g)
    ids = torch.randint(0, V, (5 * L,), generator=g)
    ids[3], ids[17] = -4, V + 2                              # out-of-range ids read (and credit) row 0, like the forward
    dy = torch.randn(5 * L, d, generator=g).to(dtype)
    wacc = [torch.zeros(V, d), torch.zeros(L + 3, d), torch.zeros(d), torch.zeros(d), torch.zeros(d)]
    gacc = [c(torch.zeros_like(a)) for a in wacc]
    ref.bert_embed_bwd(ids, word, pos, ty, ga, dy, *wacc, L)
    hip.bert_embed_bwd(c(ids), c(word), c(pos), c(ty), c(ga), c(dy), *gacc, L)
    for a, b in zip(gacc, wacc):
        torch.testing.assert_close(a.cpu(), b, atol=tol * max(1.0, float(b.abs().max())) * 3, rtol=tol)


@gpu
@pytest.mark.parametrize("name", TRAIN_CASES)
def test_gpu_fp32_training_gradients_vs_reference(name):
    """`loss.backward()` of the HIP retriever (fp32 compute) against the gradients of the reference's own Retriever: within 2e-3 of
    each tensor's scale (the 1e-3 class of the reader's fp32 parity, with the float-atomic column sums of the bias / LayerNorm
    gradients on top)"""
    cfg, w, (qi, qm, pi, pm, gold) = case(name)
    model = Retriever(cfg, dtype=torch.float32)
    model.load_state_dict(w)
    model = model.cuda().eval()
    loss = model(qi.cuda(), qm.cuda(), pi.cuda(), pm.cuda(), gold_score=gold.cuda())[3]
    assert abs(float(loss) - float(TRAIN[f"{name}.loss"])) < 1e-4
    loss.backward()
    torch.cuda.synchronize()
    check_grads(name, {n: p.grad for n, p in model.named_parameters()}, tol=2e-3)


@gpu
def test_gpu_two_optimizer_steps_vs_reference_and_bf16_training():
    name = "asym_mask"
    cfg, w, batch = case(name)
    model = Retriever(cfg, dtype=torch.float32)
    model.load_state_dict(w)
    model = model.cuda().eval()
    losses = train_two_steps(model, tuple(t.cuda() for t in batch))
    for i, l in enumerate(losses):
        assert abs(l - float(TRAIN[f"{name}.loss_step{i}"])) < 1e-4, (i, l)
    for k in w:
        ref = torch.from_numpy(TRAIN[f"{name}.w2/{k}"])
        torch.testing.assert_close(model.state_dict()[k].cpu(), ref, atol=5e-5, rtol=1e-3, msg=lambda m, k=k: f"{k}: {m}")
    # bf16 compute: the gradient as one vector within 5 % of the reference's, loss within 1 %; dropout on: finite, deterministic
    m16 = Retriever(cfg, dtype=torch.bfloat16, seed=5)
    m16.load_state_dict(w)
    m16 = m16.cuda().eval()
    loss = m16(*(t.cuda() for t in batch[:4]), gold_score=batch[4].cuda())[3]
    loss.backward()
    ref = ref_grads(name)
    got = {n: p.grad for n, p in m16.named_parameters()}
    num = sum(float(((got[k].double().cpu() - r.double()) ** 2).sum()) for k, r in ref.items())
    den = sum(float((r.double() ** 2).sum()) for r in ref.values())
    assert abs(float(loss) - float(TRAIN[f"{name}.loss"])) < 0.01 * float(TRAIN[f"{name}.loss"]) + 1e-3
    assert (num / den) ** 0.5 < 0.05, (num / den) ** 0.5
    m16.zero_grad()
    m16.train()
    m16.set_dropout(0.1)
    l1 = m16(*(t.cuda() for t in batch[:4]), gold_score=batch[4].cuda())[3]
    l1.backward()
    assert torch.isfinite(m16._get_engine().G).all() and float(m16._get_engine().G.abs().max()) > 0


@gpu
def test_gpu_bert_base_training_step_against_oracle():
    """BERT-base width (768 / 12 heads / 3072; 2 layers keep the float64 oracle quick), 130-token passages, masks on both towers:
    fp32 gradients of the HIP retriever against the oracle's autograd"""
    cfg = RetrieverConfig(num_hidden_layers=2, vocab_size=3000, apply_passage_mask=True, apply_question_mask=True)
    w = RO.init_weights(cfg, seed=9)
    qi, qm, pi, pm, gold = RO.synthetic_batch(cfg, 2, 5, 40, 130, seed=10)
    leaves = {k: v.double().clone().requires_grad_(True) for k, v in w.items()}
    ol = RO.forward(cfg, leaves, qi, qm, pi, pm, gold)[3]
    ol.backward()
    model = Retriever(cfg, dtype=torch.float32)
    model.load_state_dict(w)
    model = model.cuda().eval()
    loss = model(qi.cuda(), qm.cuda(), pi.cuda(), pm.cuda(), gold_score=gold.cuda())[3]
    loss.backward()
    assert abs(float(loss) - float(ol)) < 1e-4 + 1e-3 * abs(float(ol))
    scale = max(float(v.grad.abs().max()) for v in leaves.values() if v.grad is not None)
    for n, p in model.named_parameters():
        r = leaves[n].grad if leaves[n].grad is not None else torch.zeros_like(leaves[n])
        err = float((p.grad.double().cpu() - r).abs().max()) / max(1e-3 * scale, float(r.abs().max()))
        assert err < 3e-3, (n, err)
