"""Exact inner-product index of the knowledge facts — the GPU counterpart of the reference's `src/index.py:19-75`
(`Indexer` over `faiss.IndexFlatIP`) and of the per-example re-ranking `fact_retrieval_small_range.py:64-89`
("FAISS IndexFlatIP top-500 over [300600, 256] is one GEMM + top-k on GPU").

Same surface as the reference class: `Indexer(vector_sz)`, `index_data(ids, embeddings)`, `search_knn(query_vectors,
top_docs, index_batch_size)` → `[(db_ids as str, scores), …]`, `serialize(dir)` / `deserialize_from(dir)`.  The scores are
one fp32 `lako_gemm_nt` (exact-fp32 MFMA) of a query batch against the resident embeddings, the selection is `lako_topk`
(descending; equal scores in ascending insertion order — faiss leaves that order unspecified).  Product quantisation
(`n_subquantizers > 0`, unused by LaKo's scripts) is not implemented.  On-disk format: `index.pt` (embeddings + ids), not
faiss' binary format.  The BERT bi-encoder that produces the embeddings (src/model.py:375-483) is the remaining part of f4."""
from __future__ import annotations

import os
from typing import List, Tuple

import numpy as np
import torch


class Indexer:
    def __init__(self, vector_sz: int, n_subquantizers: int = 0, n_bits: int = 8, device="cuda", ops=None):
        if n_subquantizers > 0:
            raise NotImplementedError("IndexPQ is not implemented (LaKo's scripts use the flat index)")
        if vector_sz % 4:
            raise ValueError("vector_sz must be a multiple of 4")
        self.vector_sz = int(vector_sz)
        self.device = torch.device(device)
        if ops is None:
            from .ops import HipOps
            ops = HipOps()          # raises when the HIP library is missing: there is no CPU search path
        self.ops = ops
        self.embeddings = torch.empty(0, self.vector_sz, dtype=torch.float32, device=self.device)
        self.index_id_to_db_id = np.empty((0,), dtype=np.int64)

    @property
    def ntotal(self) -> int:
        return self.embeddings.shape[0]

    def index_data(self, ids, embeddings):
        """src/index.py:28-35: append `embeddings` [m, vector_sz] under the external ids `ids`."""
        emb = torch.as_tensor(np.asarray(embeddings, dtype=np.float32) if not torch.is_tensor(embeddings) else embeddings)
        emb = emb.to(self.device, torch.float32).reshape(-1, self.vector_sz)
        if len(ids) != emb.shape[0]:
            raise ValueError("ids and embeddings disagree in length")
        self.index_id_to_db_id = np.concatenate((self.index_id_to_db_id, np.array(ids, dtype=np.int64)), axis=0)
        self.embeddings = torch.cat([self.embeddings, emb], 0).contiguous()

    def search_scores(self, query_vectors) -> torch.Tensor:
        """fp32 inner products [nq, ntotal] of the queries with every stored vector (one GEMM)."""
        q = torch.as_tensor(np.asarray(query_vectors, dtype=np.float32) if not torch.is_tensor(query_vectors) else query_vectors)
        q = q.to(self.device, torch.float32).reshape(-1, self.vector_sz).contiguous()
        n = self.ntotal
        ld = (n + 3) // 4 * 4                                 # 16-byte rows for the store epilogue / the top-k loads
        buf = torch.empty(q.shape[0], ld, dtype=torch.float32, device=self.device)
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
        torch.save({"embeddings": self.embeddings.cpu(), "index_id_to_db_id": self.index_id_to_db_id, "vector_sz": self.vector_sz},
                   os.path.join(str(dir_path), "index.pt"))

    def deserialize_from(self, dir_path):
        d = torch.load(os.path.join(str(dir_path), "index.pt"), map_location="cpu", weights_only=False)
        assert d["vector_sz"] == self.vector_sz and len(d["index_id_to_db_id"]) == d["embeddings"].shape[0]
        self.embeddings = d["embeddings"].to(self.device)
        self.index_id_to_db_id = d["index_id_to_db_id"]


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
