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
    with pytest.raises(ValueError):
        Indexer(32, n_subquantizers=5, device="cpu", ops=RefOps())       # 32 is not a multiple of 5


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


def test_failed_add_leaves_the_id_map_alone_and_legacy_files_fail_clearly(tmp_path):
    """ADVICE (round 3): (1) index_data appends the ids only after the vectors are in — a PQ train that raises (fewer vectors than
    centroids) must not leave ids without codes behind them; (2) an index.pt whose id map is a pickled numpy array (files written before
    the id map became a tensor) is refused with a message that says what to do, not unpickled."""
    emb, _, ids = _data(n=40, d=32)
    ix = Indexer(32, n_subquantizers=4, n_bits=8, device="cpu", ops=RefOps())
    with pytest.raises(Exception):
        ix.index_data(ids, emb)                       # 40 vectors cannot train 256 centroids per sub-quantiser
    assert len(ix.index_id_to_db_id) == 0 and ix.ntotal == 0
    flat = Indexer(32, device="cpu", ops=RefOps())
    flat.index_data(ids, emb)
    flat.serialize(tmp_path)
    back = Indexer(32, device="cpu", ops=RefOps())
    back.deserialize_from(tmp_path)
    assert back.ntotal == 40 and list(back.index_id_to_db_id) == list(ids)
    legacy = tmp_path / "legacy"
    legacy.mkdir()
    torch.save({"embeddings": torch.from_numpy(emb), "index_id_to_db_id": np.asarray(ids), "vector_sz": 32}, legacy / "index.pt")
    with pytest.raises(ValueError, match="rebuild the index"):
        Indexer(32, device="cpu", ops=RefOps()).deserialize_from(legacy)


def _rerank_case():
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "rerank.json")) as f:
        z = json.load(f)
    emb, q = np.asarray(z["embeddings"], np.float32), np.asarray(z["questions"], np.float32)
    dic = {str(i): f"fact number {i}" for i in range(len(emb))}
    ex = [{"question": f"q{k}", "fact": [{"id": str(i), "sentence": dic[str(i)], "score": 0} for i in c]} for k, c in enumerate(z["candidates"])]
    return emb, q, dic, ex, z["expected"]


def _check_rerank(examples, expected):
    for ex, want in zip(examples, expected):
        assert [f["id"] for f in ex["fact"]] == want["ids"]                      # the order, ties included (descending id)
        assert [f["sentence"] for f in ex["fact"]] == want["sentences"]
        assert [f["score"] for f in ex["fact"]] == want["scores"]                # exact: the fixture's products are exact in fp32


def test_resort_facts_vs_reference_outputs():
    """PIN of the re-rank: tests/golden/rerank.json holds what the reference's own `resort_facts`
    (fact_retrieval_small_range.py:64-89, imported by oracle/make_fixtures.py::make_rerank with faiss stubbed) returned for seeded
    embeddings; lako_amd.index.resort_facts (host logic on the test double) and the numpy restatement must reproduce ids, order — exact
    ties included — sentences and scores.  (The faiss half of src/index.py stays unpinned: faiss is absent from the image.)"""
    emb, q, dic, ex, want = _rerank_case()
    ex_o = copy.deepcopy(ex)
    resort_facts(ex, dic, q, emb, ops=RefOps(), device="cpu")
    _check_rerank(ex, want)
    IO.resort_facts(ex_o, dic, q, emb)
    _check_rerank(ex_o, want)


@pytest.mark.gpu
def test_resort_facts_vs_reference_outputs_on_gpu():
    emb, q, dic, ex, want = _rerank_case()
    resort_facts(ex, dic, q, emb)
    _check_rerank(ex, want)


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


# ---- product-quantised index: Indexer(vector_sz, n_subquantizers > 0, n_bits) = faiss.IndexPQ, src/index.py:21-23 -------------------
def _clustered(n, d, nq, seed, n_centers=40, spread=0.25):
    """facts around a few centres (what makes a PQ index useful), queries near facts"""
    g = np.random.default_rng(seed)
    centers = g.standard_normal((n_centers, d)).astype(np.float32)
    emb = (centers[g.integers(0, n_centers, n)] + spread * g.standard_normal((n, d))).astype(np.float32)
    q = (emb[g.integers(0, n, nq)] + 0.05 * g.standard_normal((nq, d))).astype(np.float32)
    return emb, q, g.permutation(10 * n)[:n]


def test_pq_oracle_is_self_consistent():
    """the numpy restatement: training lowers the quantisation error monotonically (Lloyd), codes are the nearest centroids, and the
    table-sum score equals the inner product with the RECONSTRUCTED vector (what asymmetric distance computation means)"""
    emb, q, _ = _clustered(1500, 32, 6, seed=2)
    cent, errs = IO.pq_train(emb, M=8, nbits=4, niter=12)
    assert all(b <= a * (1 + 1e-6) for a, b in zip(errs, errs[1:])) and errs[-1] < 0.5 * errs[0]
    codes = IO.pq_encode(emb, cent)
    recon = np.concatenate([cent[m][codes[:, m]] for m in range(8)], axis=1)
    assert ((emb - recon) ** 2).sum(1).mean() < 1.05 * errs[-1] + 1e-6
    np.testing.assert_allclose(IO.pq_scores(q, cent, codes), q @ recon.T, rtol=1e-5, atol=1e-5)


def _check_pq_index(ix, emb, q, ids, k, M, nbits, tol):
    """the class against the oracle on the class's OWN codebooks (the k-means trajectories of fp32 atomics and float64 numpy part
    ways at the first near-tie): codes = the oracle's nearest centroids, scores = the oracle's table sums, ids / order as for the
    flat index; the codebooks themselves are as good as the oracle's (quantisation error within 3 %)."""
    cent = ix.pq.centroids.cpu().numpy()
    codes = ix.codes.cpu().numpy()
    want_codes = IO.pq_encode(emb, cent)
    differ = np.nonzero(codes != want_codes)
    for i, m in zip(*differ):                      # only exact / last-bit ties may differ
        dsub = cent.shape[2]
        xm = emb[i, m * dsub:(m + 1) * dsub]
        d_got, d_want = ((xm - cent[m, codes[i, m]]) ** 2).sum(), ((xm - cent[m, want_codes[i, m]]) ** 2).sum()
        assert abs(d_got - d_want) <= 1e-5 * max(1.0, d_want), (i, m, d_got, d_want)
    assert len(differ[0]) <= 1e-3 * codes.size
    vals, order = IO.pq_search(q, cent, codes, min(k, len(emb)))
    res = ix.search_knn(q, k, index_batch_size=4)
    scores_all = IO.pq_scores(q, cent, codes)
    for r, (db_ids, scores) in enumerate(res):
        np.testing.assert_allclose(scores, vals[r], rtol=tol, atol=tol)
        for c, got in enumerate(db_ids):
            if int(got) != ids[order[r, c]]:
                j = list(ids).index(int(got))
                assert abs(scores_all[r, j] - vals[r, c]) < 10 * tol
    _, oerrs = IO.pq_train(emb, M, nbits)
    assert abs(ix.pq.train_error[-1] - oerrs[-1]) < 0.03 * oerrs[-1], (ix.pq.train_error[-1], oerrs[-1])
    assert all(b <= a * (1 + 1e-4) for a, b in zip(ix.pq.train_error, ix.pq.train_error[1:]))
    # recall of the true top-10 inside the PQ top-100 (clustered data, 4 bits x 8 sub-quantisers is coarse: a sanity bound only)
    _, exact = IO.flat_ip_search(q, emb, 10)
    top = [set(int(x) for x in res[r][0][:100]) for r in range(len(q))]
    recall = np.mean([len({int(ids[j]) for j in exact[r]} & top[r]) / 10.0 for r in range(len(q))])
    return recall


def test_pq_indexer_host_logic_on_double(tmp_path):
    emb, q, ids = _clustered(1200, 32, 7, seed=4)
    ix = Indexer(32, n_subquantizers=8, n_bits=4, device="cpu", ops=RefOps())
    ix.index_data(ids[:700], emb[:700])            # trains on the first batch (src/index.py:31-33), then adds
    ix.index_data(ids[700:], emb[700:])
    assert ix.ntotal == 1200 and ix.codes.dtype == torch.uint8 and tuple(ix.codes.shape) == (1200, 8) and ix.embeddings.shape[0] == 0
    cent = ix.pq.centroids.numpy()
    codes = ix.codes.numpy()
    assert np.array_equal(codes, IO.pq_encode(emb, cent))
    vals, order = IO.pq_search(q, cent, codes, 50)
    res = ix.search_knn(q, 50, index_batch_size=4)
    for r, (db_ids, scores) in enumerate(res):
        np.testing.assert_allclose(scores, vals[r], rtol=1e-5, atol=1e-5)
    _, oerrs = IO.pq_train(emb[:700], 8, 4)
    assert abs(ix.pq.train_error[-1] - oerrs[-1]) < 0.03 * oerrs[-1]
    ix.serialize(tmp_path)
    ix2 = Indexer(32, device="cpu", ops=RefOps())  # the file decides the index type, like faiss.read_index
    ix2.deserialize_from(tmp_path)
    assert ix2.pq is not None and ix2.ntotal == 1200 and torch.equal(ix2.codes, ix.codes)
    r2 = ix2.search_knn(q, 50, index_batch_size=4)
    assert all(a[0] == b[0] for a, b in zip(res, r2))
    with pytest.raises(ValueError):
        Indexer(32, n_subquantizers=8, n_bits=9, device="cpu", ops=RefOps())
    small = Indexer(32, n_subquantizers=8, n_bits=8, device="cpu", ops=RefOps())
    with pytest.raises(ValueError):
        small.index_data(ids[:100], emb[:100])     # fewer training vectors than centroids (faiss refuses too)


@pytest.mark.gpu
@pytest.mark.parametrize("n,d,M,nbits,nq", [(8000, 256, 16, 8, 9), (3000, 64, 16, 4, 3), (70000, 256, 32, 8, 5), (2500, 768, 12, 6, 4)])
def test_pq_indexer_on_gpu(tmp_path, n, d, M, nbits, nq):
    """lako_pq_assign / lako_pq_lut / lako_pq_scan + lako_topk behind Indexer(d, M, nbits): the reference's vector size (256) with 16
    and 32 one-byte sub-quantisers (the second case trains on a 65 536-vector subsample), a 4-bit case with sub-vectors of 4, and
    BERT's 768 with sub-vectors of 64 and 6-bit codes."""
    emb, q, ids = _clustered(n, d, nq, seed=n)
    ix = Indexer(d, n_subquantizers=M, n_bits=nbits)
    half = n // 2 // 4 * 4 + 1
    ix.index_data(ids[:half], emb[:half])
    ix.index_data(ids[half:], emb[half:])
    assert ix.ntotal == n
    if n <= 8000:
        # the oracle's codebook comparison trains on everything; the class trained on the first batch: re-train for the comparison
        ix = Indexer(d, n_subquantizers=M, n_bits=nbits)
        ix.index_data(ids, emb)
        recall = _check_pq_index(ix, emb, q, ids, 200, M, nbits, 2e-4)
        assert recall > 0.5, recall
    else:
        cent, codes = ix.pq.centroids.cpu().numpy(), ix.codes.cpu().numpy()
        vals, order = IO.pq_search(q, cent, codes, 100)
        res = ix.search_knn(q, 100)
        for r, (db_ids, scores) in enumerate(res):
            np.testing.assert_allclose(scores, vals[r], rtol=2e-4, atol=2e-4)
    ix.serialize(tmp_path)
    ix2 = Indexer(d)
    ix2.deserialize_from(tmp_path)
    a, b = ix.search_knn(q, 20), ix2.search_knn(q, 20)
    assert all(x[0] == y[0] and np.array_equal(x[1], y[1]) for x, y in zip(a, b))


@pytest.mark.gpu
def test_pq_kernels_bitwise(tmp_path):
    """the scan adds the table entries in ascending sub-quantiser order in fp32 — the same bits as the double's loop —, odd row
    strides / M not a multiple of 16 / fewer queries than a query group included; assignment ties go to the lowest index"""
    from lako_amd.ops import HipOps
    ops, ref = HipOps(), RefOps()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    for (n, nq, M, ksub, dsub) in [(5000, 9, 16, 256, 16), (777, 3, 12, 64, 8), (1030, 1, 32, 256, 8), (64, 17, 4, 16, 2), (300, 5, 48, 256, 4)]:
        cent = torch.randn(M, ksub, dsub, generator=g)
        x = torch.randn(n, M * dsub + 4, generator=g)[:, :M * dsub]          # row stride != M * dsub
        q = torch.randn(nq, M * dsub, generator=g)
        cent[:, 1] = cent[:, 0]                                              # an exact tie: index 0 must win
        codes, codes_r = torch.empty(n, M, dtype=torch.uint8, device=dev), torch.empty(n, M, dtype=torch.uint8)
        sums, counts, err = torch.zeros(M, ksub, dsub, device=dev), torch.zeros(M, ksub, dtype=torch.int32, device=dev), torch.zeros(1, device=dev)
        sums_r, counts_r, err_r = torch.zeros(M, ksub, dsub), torch.zeros(M, ksub, dtype=torch.int32), torch.zeros(1)
        ops.pq_assign(x.to(dev), cent.to(dev), codes, sums, counts, err)
        ref.pq_assign(x, cent, codes_r, sums_r, counts_r, err_r)
        mism = (codes.cpu() != codes_r)
        assert mism.float().mean() < 2e-3                                    # last-bit distance ties between different summation orders
        assert not (codes.cpu() == 1).any()
        if not mism.any():
            assert torch.equal(counts.cpu(), counts_r)
            torch.testing.assert_close(sums.cpu(), sums_r, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(err.cpu(), err_r, rtol=1e-4, atol=1e-3)
        lut, lut_r = torch.empty(nq, M, ksub, device=dev), torch.empty(nq, M, ksub)
        ops.pq_lut(q.to(dev), cent.to(dev), lut)
        ref.pq_lut(q, cent, lut_r)
        torch.testing.assert_close(lut.cpu(), lut_r, rtol=1e-5, atol=1e-5)
        ld = (n + 3) // 4 * 4
        sc, sc_r = torch.full((nq, ld), 7.0, device=dev), torch.empty(nq, n)
        ops.pq_scan(lut, codes, sc[:, :n])
        ref.pq_scan(lut.cpu(), codes.cpu(), sc_r)
        assert torch.equal(sc[:, :n].cpu(), sc_r)
        assert bool((sc[:, n:] == 7.0).all())                                # nothing written past a row's end
