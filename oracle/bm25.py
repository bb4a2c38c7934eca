"""BM25 oracle (numpy, float64) -- test infrastructure, see oracle/__init__.py.

Follows the reference op-for-op so results are bit-identical to its numpy:
  * build : genmodel.py:51-99  (gen_and_save_bm25_index)
  * score : webui.py:119-172   (compute_bm25_scores)
Pinned by tests/golden/g1_bm25_build.json and g2_bm25_score.json.
"""
from typing import Dict, List, Sequence

import numpy as np

K1 = 1.5   # webui.py:126
B = 0.75   # webui.py:127
REQUIRE_TAG_MAGIC_NUMBER = 1000  # webui.py:60


def bm25_build(corpus: Sequence[Sequence[str]], token2id: Dict[str, int]):
    """genmodel.py:51-82.  Returns (bm25_corpus, idf, avgdl, D, doc_lengths) with the
    reference's Python/numpy types: list[dict[int,int]], dict[int,np.float64],
    np.float64, int, int64 ndarray."""
    bm25_corpus: List[Dict[int, int]] = []
    doc_lengths: List[int] = []
    term_doc_freq: Dict[int, int] = {}
    D = len(corpus)
    for tags in corpus:                                   # genmodel.py:57
        term_ids = [token2id[t] for t in tags if t in token2id]   # :59-61
        tf: Dict[int, int] = {}
        for t in term_ids:                                # :64-66
            tf[t] = tf.get(t, 0) + 1
        bm25_corpus.append(tf)
        doc_lengths.append(len(term_ids))                 # :69
        for t in tf.keys():                               # :72-73
            term_doc_freq[t] = term_doc_freq.get(t, 0) + 1
    dl = np.array(doc_lengths)                            # :75 (int64)
    avgdl = np.mean(dl)                                   # :76
    idf = {}
    for t, df in term_doc_freq.items():                   # :80-82
        idf[t] = np.log(1 + (D - df + 0.5) / (df + 0.5))
    return bm25_corpus, idf, avgdl, D, dl


def bm25_score(bm25_corpus, idf, avgdl, D, dl, query_weights: Dict[int, float]) -> np.ndarray:
    """webui.py:136-172 with query_weights given (the only form find_similar_documents
    uses, webui.py:374).  Same numpy expression order => bit-identical float64."""
    scores = np.zeros(D)
    for term_id in list(query_weights.keys()):            # webui.py:131,139
        idf_t = idf.get(term_id, 0)                       # :140
        tfs = np.array([doc.get(term_id, 0) for doc in bm25_corpus])   # :142
        denom = tfs + K1 * (1 - B + B * (dl / avgdl))     # :144
        numer = tfs * (K1 + 1)                            # :145
        score = idf_t * (numer / denom)                   # :146
        weight = query_weights.get(term_id, 1.0)          # :150
        present = np.array([term_id in doc for doc in bm25_corpus], dtype=bool)
        if weight < 0:                                    # :154-160
            scores[present] = -np.inf
        elif weight > REQUIRE_TAG_MAGIC_NUMBER:           # :161-168
            scores += (weight - REQUIRE_TAG_MAGIC_NUMBER) * score
            scores[~present] = -np.inf
        else:                                             # :169-170
            scores += weight * score
    return scores


# ---------------------------------------------------------------------------
# Vectorised restatement over CSR arrays (same arithmetic, no Python dicts):
# used for the 100k-document parity cases and as the cpu_baseline "port".
# Checked bit-for-bit against bm25_score() above in tests/test_oracle_golden.py.
# ---------------------------------------------------------------------------
def to_csr(bm25_corpus: List[Dict[int, int]]):
    """doc-major CSR in the dict's insertion order (= first occurrence order)."""
    ptr = np.zeros(len(bm25_corpus) + 1, dtype=np.int64)
    for i, d in enumerate(bm25_corpus):
        ptr[i + 1] = ptr[i] + len(d)
    terms = np.zeros(int(ptr[-1]), dtype=np.int32)
    tfs = np.zeros(int(ptr[-1]), dtype=np.int32)
    for i, d in enumerate(bm25_corpus):
        s = int(ptr[i])
        for j, (t, f) in enumerate(d.items()):
            terms[s + j] = t
            tfs[s + j] = f
    return ptr, terms, tfs


def bm25_score_csr(ptr, terms, tfs, idf_arr, avgdl, dl, q_terms, q_weights) -> np.ndarray:
    """idf_arr: dense float64[V] (0 where the term has no idf entry)."""
    D = len(ptr) - 1
    scores = np.zeros(D)
    rows = np.repeat(np.arange(D), np.diff(ptr))
    for term_id, weight in zip(q_terms, q_weights):
        idf_t = idf_arr[term_id] if 0 <= term_id < len(idf_arr) else 0.0
        m = terms == term_id
        tf_full = np.zeros(D, dtype=np.int64)
        tf_full[rows[m]] = tfs[m]
        present = np.zeros(D, dtype=bool)
        present[rows[m]] = True
        denom = tf_full + K1 * (1 - B + B * (dl / avgdl))
        numer = tf_full * (K1 + 1)
        score = idf_t * (numer / denom)
        if weight < 0:
            scores[present] = -np.inf
        elif weight > REQUIRE_TAG_MAGIC_NUMBER:
            scores += (weight - REQUIRE_TAG_MAGIC_NUMBER) * score
            scores[~present] = -np.inf
        else:
            scores += weight * score
    return scores
