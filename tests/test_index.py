"""SURVEY.md §8 f4, the index part: exact inner-product search (src/index.py) and the per-example re-ranking
(fact_retrieval_small_range.py:64-89) against the numpy restatement in oracle/index_oracle.py.  Host logic on the
test double (CPU); kernels and the full class on the GPU."""
import copy

import numpy as np
import pytest
import torch

from lako_amd.index import Indexer, resort_facts
from oracle import index_oracle as IO
from tests.ref_ops import RefOps


def _data(n=700, d=32, nq=9, seed=0):
    g = np.random.default_rng(seed)
    return g.standard_normal((n, d)).astype(np.float32), g.standard_normal((nq, d)).astype(np.float32), g.permutation(10 * n)[:n]


def _check_search(ix, emb, q, ids, k, tol):
    res = ix.search_knn(q, k, index_batch_size=4)
    vals, order = IO.flat_ip_search(q, emb, min(k, len(emb)))
    assert len(res) == len(q)
    for r, (db_ids, scores) in enumerate(res):
        np.testing.assert_allclose(scores, vals[r], rtol=tol, atol=tol)
        assert all(isinstance(s, str) for s in db_ids)
        # ids may swap only where the reference scores are closer than the tolerance
        for c, got in enumerate(db_ids):
            if int(got) != ids[order[r, c]]:
                j = list(ids).index(int(got))
                assert abs((emb[j] @ q[r]) - vals[r, c]) < 10 * tol


def test_indexer_host_logic_on_double(tmp_path):
    emb, q, ids = _data()
    ix = Indexer(32, device="cpu", ops=RefOps())
    ix.index_data(ids[:300], emb[:300])
    ix.index_data(ids[300:], emb[300:])          # appended batches, n not a multiple of 4 in between
    assert ix.ntotal == 700
    _check_search(ix, emb, q, ids, 50, 1e-5)
    _check_search(ix, emb, q, ids, 5000, 1e-5)   # top_docs > ntotal is clamped
    ix.serialize(tmp_path)
    ix2 = Indexer(32, device="cpu", ops=RefOps())
    ix2.deserialize_from(tmp_path)
    assert ix2.ntotal == 700 and np.array_equal(ix2.index_id_to_db_id, ix.index_id_to_db_id)
    with pytest.raises(NotImplementedError):
        Indexer(32, n_subquantizers=8, device="cpu", ops=RefOps())


def _examples(n_facts, nq, seed=1):
    g = np.random.default_rng(seed)
    return [{"question": f"q{i}", "fact": [{"id": int(j), "sentence": "?"} for j in g.choice(n_facts, size=int(g.integers(0, 40)), replace=False)]}
            for i in range(nq)]


def test_resort_facts_on_double():
    emb, q, _ = _data(n=500, d=32, nq=6)
    dic = {str(i): f"fact {i}" for i in range(500)}
    ex_a, ex_b = _examples(500, 6), None
    ex_b = copy.deepcopy(ex_a)
    resort_facts(ex_a, dic, q, emb, ops=RefOps(), device="cpu")
    IO.resort_facts(ex_b, dic, q, emb)
    for a, b in zip(ex_a, ex_b):
        assert [f["id"] for f in a["fact"]] == [f["id"] for f in b["fact"]]
        assert [f["sentence"] for f in a["fact"]] == [f["sentence"] for f in b["fact"]]
        np.testing.assert_allclose([f["score"] for f in a["fact"]], [f["score"] for f in b["fact"]], rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,n,k", [(3, 1000, 10), (2, 300600, 500), (5, 4099, 1024), (1, 7, 7), (4, 2048, 1)])
def test_topk_kernel(rows, n, k):
    from lako_amd.ops import HipOps
    ops, ref = HipOps(), RefOps()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(n)
    ld = (n + 3) // 4 * 4
    buf = torch.randn(rows, ld, generator=g).to(dev)
    cases = [buf[:, :n]]
    tied = torch.randint(0, 7, (rows, ld), generator=g).float().to(dev)[:, :n]     # massive ties: index order decides
    cases += [tied, torch.zeros(rows, ld, device=dev)[:, :n], -buf[:, :n].abs()]
    for sc in cases:
        v, i = torch.empty(rows, k, device=dev), torch.empty(rows, k, dtype=torch.int64, device=dev)
        vr, ir = torch.empty_like(v), torch.empty_like(i)
        ops.topk(sc, k, v, i)
        ref.topk(sc, k, vr, ir)
        assert torch.equal(v, vr) and torch.equal(i, ir)


@pytest.mark.gpu
def test_indexer_and_resort_on_gpu(tmp_path):
    emb, q, ids = _data(n=30061, d=256, nq=7, seed=3)       # the reference's vector size, a tenth of its fact count
    ix = Indexer(256)
    ix.index_data(ids, emb)
    _check_search(ix, emb, q, ids, 500, 2e-4)
    ix.serialize(tmp_path)
    ix2 = Indexer(256)
    ix2.deserialize_from(tmp_path)
    _check_search(ix2, emb, q, ids, 20, 2e-4)
    dic = {str(i): f"fact {i}" for i in range(len(emb))}
    ex_a = _examples(len(emb), len(q), seed=5)
    ex_b = copy.deepcopy(ex_a)
    resort_facts(ex_a, dic, q, emb)
    IO.resort_facts(ex_b, dic, q, emb)
    for a, b in zip(ex_a, ex_b):
        sa, sb = [f["score"] for f in a["fact"]], [f["score"] for f in b["fact"]]
        np.testing.assert_allclose(sa, sb, rtol=2e-4, atol=2e-4)
        assert sorted(f["id"] for f in a["fact"]) == sorted(f["id"] for f in b["fact"])
