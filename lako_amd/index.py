"""Exact inner-product index of the knowledge facts — the GPU counterpart of the reference's `src/index.py:19-75`
(`Indexer` over `faiss.IndexFlatIP`) and of the per-example re-ranking `fact_retrieval_small_range.py:64-89`
("FAISS IndexFlatIP top-500 over [300600, 256] is one GEMM + top-k on GPU").

Same surface as the reference class: `Indexer(vector_sz)`, `index_data(ids, embeddings)`, `search_knn(query_vectors,
top_docs, index_batch_size)` → `[(db_ids as str, scores), …]`, `serialize(dir)` / `deserialize_from(dir)`.  The scores are
one fp32 `lako_gemm_nt` (exact-fp32 MFMA) of a query batch against the resident embeddings, the selection is `lako_topk`
(descending; equal scores in ascending insertion order — faiss leaves that order unspecified).  With `n_subquantizers > 0`
(`faiss.IndexPQ(d, M, n_bits, METRIC_INNER_PRODUCT)`, src/index.py:21-23; LaKo's own scripts construct the flat index) the vectors
are stored as M one-byte codes each: `ProductQuantizer` below trains the M codebooks by k-means (`lako_pq_assign` = assignment +
accumulation, the loop around it is host logic), `index_data` encodes, and a search is a per-query lookup table (`lako_pq_lut`), a
scan of the codes (`lako_pq_scan`) and the same `lako_topk`.  faiss is absent from the image: the definition followed is the
published one (csrc/pq.hip header), the random choices (training subsample, initial centroids) come from numpy's RandomState(1234)
rather than faiss' generator, so codebooks are comparable in quality, not in bits — parity unpinned by reference outputs, like the
flat index.  On-disk format: `index.pt` (embeddings or codebooks + codes, ids), not faiss' binary format."""
from __future__ import annotations

import os
from typing import List, Tuple

import numpy as np
import torch


PQ_SUBVECTOR_LENGTHS = (1, 2, 4, 6, 8, 12, 16, 24, 32, 48, 64)      # instantiations of pq_assign_kernel (csrc/pq.hip)


class ProductQuantizer:
    """faiss.ProductQuantizer(d, M, nbits) as IndexPQ uses it: M independent k-means codebooks of ksub = 2^nbits centroids over
    the M sub-vectors of length d / M (faiss.Clustering defaults: 25 iterations, at most 256 training points per centroid —
    a random subsample beyond that —, initial centroids = ksub random training points, an empty cluster re-seeded from the most
    populated one with a ±1/1024 relative perturbation)."""
    NITER, MAX_POINTS_PER_CENTROID, SEED, EPS = 25, 256, 1234, 1.0 / 1024.0

    def __init__(self, d: int, M: int, nbits: int, ops, device):
        if M <= 0 or d % M:
            raise ValueError(f"the vector size {d} must be a multiple of n_subquantizers {M}")
        if not 1 <= nbits <= 8:
            raise ValueError("n_bits must be in 1..8 (one byte per sub-quantiser code)")
        self.d, self.M, self.nbits, self.ksub, self.dsub = int(d), int(M), int(nbits), 1 << int(nbits), d // M
        if self.dsub not in PQ_SUBVECTOR_LENGTHS:
            raise ValueError(f"sub-vector length {self.dsub} (= vector_sz / n_subquantizers) must be one of {PQ_SUBVECTOR_LENGTHS}")
        if self.M * self.ksub * 4 > 128 * 1024:
            raise ValueError("n_subquantizers * 2^n_bits * 4 bytes (one query's lookup table) must fit 128 KiB of LDS")
        self.ops, self.device = ops, torch.device(device)
        self.centroids = None                    # [M, ksub, dsub] fp32
        self.train_error = []                    # mean squared quantisation error per iteration (whole vectors)

    @property
    def is_trained(self) -> bool:
        return self.centroids is not None

    def train(self, x: torch.Tensor):
        n = x.shape[0]
        if n < self.ksub:
            raise ValueError(f"{n} training vectors for {self.ksub} centroids per sub-quantiser")
        rs = np.random.RandomState(self.SEED)
        if n > self.ksub * self.MAX_POINTS_PER_CENTROID:
            x = x[torch.from_numpy(rs.permutation(n)[:self.ksub * self.MAX_POINTS_PER_CENTROID]).to(x.device)].contiguous()
            n = x.shape[0]
        first = torch.from_numpy(rs.permutation(n)[:self.ksub]).to(x.device)
        M, ksub, dsub = self.M, self.ksub, self.dsub
        cent = x[first].view(ksub, M, dsub).transpose(0, 1).contiguous()
        sums = torch.empty(M, ksub, dsub, dtype=torch.float32, device=self.device)
        counts = torch.empty(M, ksub, dtype=torch.int32, device=self.device)
        err = torch.empty(1, dtype=torch.float32, device=self.device)
        self.train_error = []
        for _ in range(self.NITER):
            sums.zero_(); counts.zero_(); err.zero_()
            self.ops.pq_assign(x, cent, None, sums, counts, err)
            cent = torch.where(counts[..., None] > 0, sums / counts.clamp(min=1)[..., None].float(), cent)
            cnt = counts.cpu().numpy().astype(np.int64)
            self.train_error.append(float(err) / n)
            for m, c in zip(*np.nonzero(cnt == 0)):           # empty clusters (host logic; rare)
                big = int(cnt[m].argmax())
                sign = torch.where(torch.arange(dsub, device=self.device) % 2 == 0, 1.0, -1.0) * self.EPS
                cent[m, c] = cent[m, big] * (1.0 + sign)
                cent[m, big] = cent[m, big] * (1.0 - sign)
                cnt[m, c] = cnt[m, big] // 2
                cnt[m, big] -= cnt[m, c]
        self.centroids = cent.contiguous()

    def compute_codes(self, x: torch.Tensor) -> torch.Tensor:
        codes = torch.empty(x.shape[0], self.M, dtype=torch.uint8, device=self.device)
        self.ops.pq_assign(x, self.centroids, codes)
        return codes

    def decode(self, codes: torch.Tensor) -> torch.Tensor:
        """reconstruction (faiss `ProductQuantizer.decode`): host-side convenience for tests and inspection"""
        m = torch.arange(self.M, device=codes.device)[None, :]
        return self.centroids[m, codes.long()].reshape(codes.shape[0], self.d)


class Indexer:
    def __init__(self, vector_sz: int, n_subquantizers: int = 0, n_bits: int = 8, device="cuda", ops=None):
        if vector_sz % 4:
            raise ValueError("vector_sz must be a multiple of 4")
        self.vector_sz = int(vector_sz)
        self.device = torch.device(device)
        if ops is None:
            from .ops import HipOps
            ops = HipOps()          # raises when the HIP library is missing: there is no CPU search path
        self.ops = ops
        self.pq = ProductQuantizer(self.vector_sz, n_subquantizers, n_bits, ops, self.device) if n_subquantizers > 0 else None
        self.embeddings = torch.empty(0, self.vector_sz, dtype=torch.float32, device=self.device)       # flat index
        self.codes = torch.empty(0, n_subquantizers, dtype=torch.uint8, device=self.device)              # PQ index
        self.index_id_to_db_id = np.empty((0,), dtype=np.int64)

    @property
    def ntotal(self) -> int:
        return self.codes.shape[0] if self.pq is not None else self.embeddings.shape[0]

    def index_data(self, ids, embeddings):
        """src/index.py:28-35: append `embeddings` [m, vector_sz] under the external ids `ids`."""
        emb = torch.as_tensor(np.asarray(embeddings, dtype=np.float32) if not torch.is_tensor(embeddings) else embeddings)
        emb = emb.to(self.device, torch.float32).reshape(-1, self.vector_sz)
        if len(ids) != emb.shape[0]:
            raise ValueError("ids and embeddings disagree in length")
        new_ids = np.array(ids, dtype=np.int64)
        if self.pq is not None:                  # src/index.py:31-33: `if not self.index.is_trained: self.index.train(embeddings)`, then add
            emb = emb.contiguous()
            if not self.pq.is_trained:
                self.pq.train(emb)               # (may raise: too few vectors for 2^n_bits centroids, unsupported sub-vector width)
            self.codes = torch.cat([self.codes, self.pq.compute_codes(emb)], 0).contiguous()
        else:
            self.embeddings = torch.cat([self.embeddings, emb], 0).contiguous()
        # the id map grows only AFTER the vectors are in: a failed train / encode must not leave ids without vectors behind them
        self.index_id_to_db_id = np.concatenate((self.index_id_to_db_id, new_ids), axis=0)

    def search_scores(self, query_vectors) -> torch.Tensor:
        """fp32 inner products [nq, ntotal] of the queries with every stored vector (one GEMM)."""
        q = torch.as_tensor(np.asarray(query_vectors, dtype=np.float32) if not torch.is_tensor(query_vectors) else query_vectors)
        q = q.to(self.device, torch.float32).reshape(-1, self.vector_sz).contiguous()
        n = self.ntotal
        ld = (n + 3) // 4 * 4                                 # 16-byte rows for the store epilogue / the top-k loads
        buf = torch.empty(q.shape[0], ld, dtype=torch.float32, device=self.device)
        if self.pq is not None:                  # asymmetric distance computation: table per query, then one pass over the codes
            lut = torch.empty(q.shape[0], self.pq.M, self.pq.ksub, dtype=torch.float32, device=self.device)
            self.ops.pq_lut(q, self.pq.centroids, lut)
            self.ops.pq_scan(lut, self.codes, buf[:, :n])
            return buf[:, :n]
        if ld != n:                                           # pad the embedding rows seen by the GEMM, never the result
            emb = torch.zeros(ld, self.vector_sz, dtype=torch.float32, device=self.device)
            emb[:n] = self.embeddings
        else:
            emb = self.embeddings
        self.ops.gemm_nt(q, emb, buf)
        return buf[:, :n]

    def search_knn(self, query_vectors, top_docs: int, index_batch_size: int = 1024) -> List[Tuple[List[str], np.ndarray]]:
        """src/index.py:37-50."""
        q_all = torch.as_tensor(np.asarray(query_vectors, dtype=np.float32) if not torch.is_tensor(query_vectors) else query_vectors)
        q_all = q_all.reshape(-1, self.vector_sz)
        k = min(int(top_docs), self.ntotal)
        result = []
        for s in range(0, q_all.shape[0], index_batch_size):
            scores = self.search_scores(q_all[s:s + index_batch_size])
            vals = torch.empty(scores.shape[0], k, dtype=torch.float32, device=self.device)
            idx = torch.empty(scores.shape[0], k, dtype=torch.int64, device=self.device)
            self.ops.topk(scores, k, vals, idx)
            vals_h, idx_h = vals.cpu().numpy(), idx.cpu().numpy()
            for r in range(idx_h.shape[0]):
                result.append(([str(self.index_id_to_db_id[i]) for i in idx_h[r]], vals_h[r]))
        return result

    def serialize(self, dir_path):
        os.makedirs(str(dir_path), exist_ok=True)
        # tensors and plain numbers only: the file loads with torch.load(weights_only=True) (the reference pickles its id map,
        # src/index.py:58-59 — an index file from an untrusted source must not be able to run code here)
        d = {"embeddings": self.embeddings.cpu(), "index_id_to_db_id": torch.from_numpy(np.ascontiguousarray(self.index_id_to_db_id)),
             "vector_sz": self.vector_sz}
        if self.pq is not None:
            d.update(codes=self.codes.cpu(), centroids=None if self.pq.centroids is None else self.pq.centroids.cpu(),
                     n_subquantizers=self.pq.M, n_bits=self.pq.nbits)
        torch.save(d, os.path.join(str(dir_path), "index.pt"))

    def deserialize_from(self, dir_path):
        path = os.path.join(str(dir_path), "index.pt")
        try:
            d = torch.load(path, map_location="cpu", weights_only=True)
        except Exception as e:      # (pickle.UnpicklingError for anything but tensors / numbers: e.g. the numpy id map older files carried)
            raise ValueError(f"{path} does not load as tensors and plain numbers only (torch.load(weights_only=True): {type(e).__name__}). "
                             "Index files written before the id map became a tensor stored it as a pickled numpy array: rebuild the index "
                             "(index_data + serialize) — such a file is not unpickled here, whatever its origin") from e
        assert d["vector_sz"] == self.vector_sz
        if "codes" in d:                         # (like faiss.read_index: the file decides the index type)
            self.pq = ProductQuantizer(self.vector_sz, d["n_subquantizers"], d["n_bits"], self.ops, self.device)
            self.pq.centroids = None if d["centroids"] is None else d["centroids"].to(self.device)
            self.codes = d["codes"].to(self.device)
        else:
            self.pq = None
            self.embeddings = d["embeddings"].to(self.device)
        self.index_id_to_db_id = d["index_id_to_db_id"].numpy().astype(np.int64)
        assert len(self.index_id_to_db_id) == self.ntotal, "Deserialized index_id_to_db_id should match the index size"


def resort_facts(examples, all_id_to_facts_dic, questions_embedding, allembeddings, ops=None, device="cuda"):
    """fact_retrieval_small_range.py:64-89: re-rank every example's own candidate facts by the inner product of their
    embeddings with the example's question embedding (descending; equal scores keep descending id order like the
    reference's `sorted(zip(score, id), reverse=True)`), rewriting `ex['fact']` in place."""
    assert len(examples) == questions_embedding.shape[0]
    if ops is None:
        from .ops import HipOps
        ops = HipOps()
    dev = torch.device(device)
    emb = torch.as_tensor(allembeddings).to(dev, torch.float32).contiguous()
    qs = torch.as_tensor(questions_embedding).to(dev, torch.float32).contiguous()
    for num, ex in enumerate(examples):
        fact_ids = [int(f["id"]) for f in ex["fact"]]
        if not fact_ids:
            continue
        m = (len(fact_ids) + 3) // 4 * 4
        sel = torch.zeros(m, emb.shape[1], dtype=torch.float32, device=dev)
        sel[:len(fact_ids)] = emb[torch.tensor(fact_ids, device=dev)]
        out = torch.empty(1, m, dtype=torch.float32, device=dev)
        ops.gemm_nt(qs[num:num + 1].contiguous(), sel, out)
        scores = out[0, :len(fact_ids)].cpu().tolist()
        pairs = sorted(zip(scores, fact_ids), reverse=True)
        ex["fact"] = [{"sentence": all_id_to_facts_dic[str(i)], "id": i, "score": s} for s, i in pairs]
