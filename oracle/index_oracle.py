"""TEST INFRASTRUCTURE — CPU restatement (numpy) of the reference's exact inner-product fact search, for checking
`lako_amd/index.py`; never imported by the product path.

  flat_ip_search      src/index.py:19-50: faiss.IndexFlatIP.search = top-k of queries · embeddingsᵀ, scores descending.
                      faiss is not installed in this image, so this restates IndexFlatIP's published definition (exact
                      maximum inner product search, no training, no quantisation); ties are broken towards the lower index
                      (faiss leaves the order of equal scores unspecified) — parity unpinned by reference outputs for this row.
  resort_facts        fact_retrieval_small_range.py:64-89, line by line: torch.matmul(fact_embedding, question) →
                      sorted(zip(score, id), reverse=True).
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
