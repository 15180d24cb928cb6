"""TEST INFRASTRUCTURE — CPU restatement (numpy) of the reference's exact inner-product fact search, for checking
`lako_amd/index.py`; never imported by the product path.

  flat_ip_search      src/index.py:19-50: faiss.IndexFlatIP.search = top-k of queries · embeddingsᵀ, scores descending.
                      faiss is not installed in this image, so this restates IndexFlatIP's published definition (exact
                      maximum inner product search, no training, no quantisation); ties are broken towards the lower index
                      (faiss leaves the order of equal scores unspecified) — parity unpinned by reference outputs for this row.
  resort_facts        fact_retrieval_small_range.py:64-89, line by line: torch.matmul(fact_embedding, question) →
                      sorted(zip(score, id), reverse=True).
  pq_*                src/index.py:21-23 with n_subquantizers > 0: faiss.IndexPQ(d, M, nbits, METRIC_INNER_PRODUCT), restated from
                      its published definition (faiss absent — parity unpinned): k-means codebooks per sub-vector (Lloyd
                      iterations from ksub random training points; numpy RandomState(1234) stands in for faiss' generator),
                      codes = nearest centroid in L2 (lowest index on ties), search = per-query inner-product tables summed over
                      the sub-quantisers in ascending order in fp32, k largest.
"""
import numpy as np


def flat_ip_search(queries: np.ndarray, embeddings: np.ndarray, k: int):
    scores = queries.astype(np.float32) @ embeddings.astype(np.float32).T
    order = np.argsort(-scores, axis=1, kind="stable")[:, :k]
    return np.take_along_axis(scores, order, axis=1), order


def resort_facts(examples, all_id_to_facts_dic, questions_embedding, allembeddings):
    q = np.asarray(questions_embedding, dtype=np.float32)
    e = np.asarray(allembeddings, dtype=np.float32)
    for num, ex in enumerate(examples):
        fact_ids = [int(f["id"]) for f in ex["fact"]]
        if not fact_ids:
            continue
        score_list = (e[fact_ids] @ q[num]).tolist()
        score_list, fact_ids = (list(t) for t in zip(*sorted(zip(score_list, fact_ids), reverse=True)))
        ex["fact"] = [{"sentence": all_id_to_facts_dic[str(fact_ids[c])], "id": fact_ids[c], "score": score_list[c]}
                      for c in range(len(fact_ids))]


def pq_train(x: np.ndarray, M: int, nbits: int, niter: int = 25, seed: int = 1234, max_points_per_centroid: int = 256):
    """→ centroids [M, ksub, dsub] float32 and the mean squared quantisation error per iteration."""
    x = np.asarray(x, dtype=np.float32)
    n, d = x.shape
    ksub, dsub = 1 << nbits, d // M
    rs = np.random.RandomState(seed)
    if n > ksub * max_points_per_centroid:
        x = x[rs.permutation(n)[:ksub * max_points_per_centroid]]
        n = x.shape[0]
    first = rs.permutation(n)[:ksub]
    cent = x[first].reshape(ksub, M, dsub).transpose(1, 0, 2).copy()
    errs = []
    for _ in range(niter):
        e = 0.0
        for m in range(M):
            xm = x[:, m * dsub:(m + 1) * dsub]
            d2 = ((xm[:, None, :].astype(np.float64) - cent[m][None].astype(np.float64)) ** 2).sum(-1)
            a = d2.argmin(1)
            e += d2[np.arange(n), a].sum()
            cnt = np.bincount(a, minlength=ksub)
            sm = np.zeros((ksub, dsub), np.float64)
            np.add.at(sm, a, xm)
            new = np.where(cnt[:, None] > 0, sm / np.maximum(cnt, 1)[:, None], cent[m])
            cnt = cnt.astype(np.int64)
            for c in np.nonzero(cnt == 0)[0]:
                big = int(cnt.argmax())
                sign = np.where(np.arange(dsub) % 2 == 0, 1.0, -1.0) / 1024.0
                new[c] = new[big] * (1.0 + sign)
                new[big] = new[big] * (1.0 - sign)
                cnt[c] = cnt[big] // 2
                cnt[big] -= cnt[c]
            cent[m] = new.astype(np.float32)
        errs.append(e / n)
    return cent, errs


def pq_encode(x: np.ndarray, cent: np.ndarray) -> np.ndarray:
    M, ksub, dsub = cent.shape
    x = np.asarray(x, dtype=np.float32)
    codes = np.empty((x.shape[0], M), np.uint8)
    for m in range(M):
        t = x[:, None, m * dsub:(m + 1) * dsub] - cent[m][None]
        d2 = np.zeros(t.shape[:2], np.float32)
        for j in range(dsub):                     # the kernel's summation order (ascending j, fp32)
            d2 += t[:, :, j] * t[:, :, j]
        codes[:, m] = d2.argmin(1)
    return codes


def pq_scores(queries: np.ndarray, cent: np.ndarray, codes: np.ndarray) -> np.ndarray:
    M, ksub, dsub = cent.shape
    q = np.asarray(queries, dtype=np.float32)
    out = np.zeros((q.shape[0], codes.shape[0]), np.float32)
    for m in range(M):
        lut = np.zeros((q.shape[0], ksub), np.float32)
        for j in range(dsub):
            lut += q[:, m * dsub + j, None] * cent[m][None, :, j]
        out += lut[:, codes[:, m]]
    return out


def pq_search(queries, cent, codes, k):
    s = pq_scores(queries, cent, codes)
    order = np.argsort(-s, axis=1, kind="stable")[:, :k]
    return np.take_along_axis(s, order, axis=1), order
