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
